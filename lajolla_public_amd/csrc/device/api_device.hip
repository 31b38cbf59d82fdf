// C-ABI entry points that own the GPU: context, scene upload, the wavefront render loop, batched ray queries.
// One context = one HIP device + one stream; one process per GPU (the multi-GPU layer above this is
// torch.distributed over RCCL, see bench.py).  Every HIP call is checked; failures surface as LJ_ERR_DEVICE with
// the HIP error string — there is no CPU fallback behind any of these functions.
#include "api_internal.h"

namespace {

void set_device(lj_context *ctx) { HIP_CHECK(hipSetDevice(ctx->device)); }

ljd::DQueue carve_queue(void *base, uint32_t cap) {
    // one allocation, eight arrays of 16-byte records; hipMalloc returns 256-byte aligned memory and cap is a
    // multiple of 64, so every array starts on a 1 KiB boundary (one full wave instruction)
    ljd::DQueue q{};
    ljd::Rec4 *p = (ljd::Rec4 *)base;
    auto take = [&]() { ljd::Rec4 *r = p; p += cap; return r; };
    q.ro = take(); q.rd = take(); q.rs = take(); q.rh = take(); q.rw = take(); q.rl = take(); q.rn = take(); q.rg = take();
    return q;
}
size_t queue_bytes(uint32_t cap) { return (size_t)cap * kQueueSlotBytes; }

void ensure_queues(lj_context *ctx, uint32_t cap) {
    if (ctx->queue_capacity >= cap) return;
    cap = (cap + 63u) & ~63u;
    ctx->queue_mem.alloc(queue_bytes(cap));
    ctx->queue_capacity = cap;
}

// overflow stack storage for a launch of `grid` workgroups over a scene whose BVH needs `levels` levels beyond LDS
int *ensure_spill(lj_context *ctx, int levels, uint32_t grid) {
    if (levels <= 0) return nullptr;
    const size_t need = (size_t)levels * grid * 256 * sizeof(int);
    if (ctx->spill.bytes < need) ctx->spill.alloc(need);
    return (int *)ctx->spill.p;
}

struct RenderPlan {
    int spp; uint32_t pool; uint64_t seed;
    std::shared_ptr<const std::vector<uint32_t>> pixels;  // rendered pixels in tile order (cached per scene: same share -> same list)
    uint64_t pixels_key = 0;                               // identifies the list on the device (lj_context::pixel_list_key)
    int max_depth;
};

RenderPlan make_plan(const lj_scene *sc, const LjRenderArgs *a) {
    RenderPlan p;
    const int w = sc->flat.cam.width, h = sc->flat.cam.height;
    p.spp = (a && a->spp > 0) ? a->spp : sc->flat.spp;
    if (p.spp <= 0) throw LjError(LJ_ERR_INVALID_ARG, "samples per pixel must be positive");
    p.seed = (a && a->seed) ? a->seed : 0x853c49e6748fea9bULL;
    p.max_depth = (a && a->max_depth != INT32_MIN) ? a->max_depth : sc->flat.max_depth;
    // 128 M paths in flight = 16 GiB of queue records (of the 288 GB a MI355X has; allocated only as far as a pass has samples): long
    // per-wave slices keep the extend kernel's lanes refilled and every launch's drain phase is amortised over more paths
    // (tools/pool_big.sh, sponza 1024 spp: 2^25 828 ms, 2^26 811, 2^27 794, 2^28 793; disney_bsdf 256 spp: 97.3 / 93.4 / 91.8 / 91.5)
    p.pool = (a && a->pool_paths) ? a->pool_paths : (1u << 27);
    p.pool = std::max<uint32_t>(p.pool, 4096);
    if (a && a->rng_mode != LJ_RNG_SAMPLE) throw LjError(LJ_ERR_UNSUPPORTED, "only LJ_RNG_SAMPLE exists on the device (a per-tile sequential stream cannot be parallelised, SURVEY §0.2)");
    int rank = a ? a->rank : 0, world = (a && a->world_size > 0) ? a->world_size : 1;
    if (rank < 0 || rank >= world) throw LjError(LJ_ERR_INVALID_ARG, "rank must be in [0, world_size)");
    bool crop = a && a->crop_x1 > a->crop_x0 && a->crop_y1 > a->crop_y0;
    int cx0 = crop ? a->crop_x0 : 0, cy0 = crop ? a->crop_y0 : 0, cx1 = crop ? a->crop_x1 : w, cy1 = crop ? a->crop_y1 : h;
    if (cx0 < 0 || cy0 < 0 || cx1 > w || cy1 > h) throw LjError(LJ_ERR_INVALID_ARG, "crop window outside the film");
    const int tile = 16, ntx = (w + tile - 1) / tile, nty = (h + tile - 1) / tile;  // render.cpp:75-77
    // the pixel list depends on the share only (rank, world, crop): a render loop asks for the same one every frame
    const uint64_t key_parts[9] = {(uint64_t)rank, (uint64_t)world, (uint64_t)crop, (uint64_t)cx0, (uint64_t)cy0, (uint64_t)cx1, (uint64_t)cy1, (uint64_t)w, (uint64_t)h};
    uint64_t key = 1469598103934665603ull;
    for (uint64_t v : key_parts) { key ^= v + 0x9e3779b97f4a7c15ull; key *= 1099511628211ull; }
    key ^= (uint64_t)(uintptr_t)sc; key |= 1ull;
    lj_scene *msc = const_cast<lj_scene *>(sc);
    if (msc->plan_pixels && msc->plan_pixels_key == key) { p.pixels = msc->plan_pixels; p.pixels_key = key; return p; }
    auto pixels = std::make_shared<std::vector<uint32_t>>();
    if (crop) {  // crop windows are enumerated row-major (lj_render_samples layout)
        for (int y = cy0; y < cy1; y++) for (int x = cx0; x < cx1; x++) {
            int t = (y / tile) * ntx + (x / tile);
            if (t % world == rank) pixels->push_back((uint32_t)(y * w + x));
        }
    } else {
        for (int t = 0; t < ntx * nty; t++) {
            if (t % world != rank) continue;
            int tx = t % ntx, ty = t / ntx;
            int x0 = tx * tile, x1 = std::min(x0 + tile, w), y0 = ty * tile, y1 = std::min(y0 + tile, h);
            for (int y = y0; y < y1; y++) for (int x = x0; x < x1; x++) pixels->push_back((uint32_t)(y * w + x));
        }
    }
    p.pixels = pixels; p.pixels_key = key;
    msc->plan_pixels = pixels; msc->plan_pixels_key = key;
    return p;
}

// the plan's pixel list on the device; the copy is skipped when the context still holds this very list
void upload_pixel_list(lj_context *ctx, const RenderPlan &plan, hipStream_t stream) {
    const size_t n = plan.pixels->size();
    if (ctx->pixel_list.bytes < n * 4) { ctx->pixel_list.alloc(n * 4); ctx->pixel_list_key = 0; }
    if (ctx->pixel_list_key == plan.pixels_key && plan.pixels_key != 0) return;
    HIP_CHECK(hipMemcpyAsync(ctx->pixel_list.p, plan.pixels->data(), n * 4, hipMemcpyHostToDevice, stream));
    ctx->pixel_list_key = plan.pixels_key;
}

// Renders plan.pixels; if rgb_dev != null writes radiance/spp there (other pixels untouched), if samples_host != null
// the per-sample radiance of every pass is copied to host memory in pixel-list order.
void run_render(lj_scene *sc, const RenderPlan &plan, float *rgb_dev, float *samples_host, hipStream_t stream, bool timing) {
    lj_context *ctx = sc->ctx;
    set_device(ctx);
    ljd::DScene ds = sc->dscene;
    ds.max_depth = plan.max_depth;
    const uint64_t n_pix = plan.pixels->size();
    LjStats &st = sc->stats; st = LjStats{};
    if (n_pix == 0) return;
    if (sc->flat.integrator < LJ_INTEGRATOR_PATH) {   // auxiliary buffers: one primary ray per pixel, no queue
        if (samples_host) throw LjError(LJ_ERR_UNSUPPORTED, "the auxiliary integrators have one deterministic value per pixel, no per-sample values");
        upload_pixel_list(ctx, plan, stream);
        HIP_CHECK(hipEventRecord(ctx->ev_begin, stream));
        const int grid = (int)std::min<uint64_t>((n_pix + 255) / 256, (uint64_t)ctx->n_cus * 4);
        ljd::launch_aux(sc->dscene, (const uint32_t *)ctx->pixel_list.p, (uint32_t)n_pix, sc->flat.integrator, rgb_dev, sc->ecfg,
                        ensure_spill(ctx, sc->ecfg.spill_levels, (uint32_t)grid), grid, stream);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipEventRecord(ctx->ev_end, stream));
        HIP_CHECK(hipEventSynchronize(ctx->ev_end));
        float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
        st.render_ms = ms; st.samples = n_pix; st.rays_closest = n_pix;
        return;
    }
    // pass size: keep the per-sample radiance buffer <= ~1.5 GiB and sample ids in 32 bits
    const uint64_t max_samples_pass = (uint64_t)1 << 27;
    uint64_t pix_per_pass = std::max<uint64_t>(1, max_samples_pass / (uint64_t)plan.spp);
    pix_per_pass = std::min<uint64_t>(pix_per_pass, n_pix);
    if (sc->flat.integrator == LJ_INTEGRATOR_VOLPATH) {   // volumetric path tracer: one lane per sample, whole path (dvol.h)
        const uint64_t pass_samples_max = pix_per_pass * (uint64_t)plan.spp;
        if (ctx->sample_rgb.bytes < pass_samples_max * 12) ctx->sample_rgb.alloc(pass_samples_max * 12);
        if (!ctx->chunk_counter.p) ctx->chunk_counter.alloc(128 * kMaxLanes);
        upload_pixel_list(ctx, plan, stream);
        HIP_CHECK(hipMemsetAsync(ctx->chunk_counter.p, 0, 256, stream));
        HIP_CHECK(hipEventRecord(ctx->ev_begin, stream));
        for (uint64_t p0 = 0; p0 < n_pix; p0 += pix_per_pass) {
            const uint64_t np = std::min<uint64_t>(pix_per_pass, n_pix - p0);
            const uint64_t total = np * (uint64_t)plan.spp;
            ljd::DPass pass{};
            pass.pixel_list = (const uint32_t *)ctx->pixel_list.p + p0; pass.n_pixels = (uint32_t)np; ljd::set_pass_divisors(pass, (uint32_t)plan.spp, (uint32_t)sc->flat.cam.width);
            pass.seed = plan.seed; pass.sample_rgb = (float *)ctx->sample_rgb.p;
            // persistent grid (k_volpath regenerates paths): as many workgroups as stay resident, fewer for a small pass
            const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((total + 255) / 256, (uint64_t)ctx->n_cus * (uint64_t)ljd::volpath_blocks_per_cu(ds)));
            HIP_CHECK(hipMemsetAsync((char *)ctx->chunk_counter.p + 8, 0, 4, stream));   // the launch's sample counter
            ljd::launch_volpath(ds, pass, (uint32_t)total, (uint32_t *)ctx->chunk_counter.p, sc->ecfg, sc->scfg.variant,
                                /* plain: */ (sc->feat_kinds & ~7u) == 0u && !sc->feat_textured && !sc->feat_envmap && !sc->feat_sphere_lights,
                                ensure_spill(ctx, sc->ecfg.spill_levels, (uint32_t)grid), grid, stream);
            HIP_CHECK(hipGetLastError());
            st.samples += total; st.wavefront_steps++;
            if (rgb_dev) ljd::launch_resolve(pass, (uint32_t)np, rgb_dev, stream);
            if (samples_host) {
                HIP_CHECK(hipMemcpyAsync(samples_host + p0 * (uint64_t)plan.spp * 3, ctx->sample_rgb.p, total * 12, hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipStreamSynchronize(stream));
            }
        }
        HIP_CHECK(hipEventRecord(ctx->ev_end, stream));
        HIP_CHECK(hipEventSynchronize(ctx->ev_end));
        float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
        unsigned long long nb = 0;
        HIP_CHECK(hipMemcpy(&nb, ctx->chunk_counter.p, 8, hipMemcpyDeviceToHost));
        st.render_ms = ms; st.bounce_iterations = nb;
        if (getenv("LJ_VOLPATH_STATS")) {   // developer build only (-DLJ_VOLPATH_STATS=1): events and lane-events per part of the tracer
            unsigned long long h[32];
            HIP_CHECK(hipMemcpy(h, ctx->chunk_counter.p, sizeof(h), hipMemcpyDeviceToHost));
            const char *names[8] = {"node iterations", "leaf steps", "closest calls", "tracking (bounce ray)", "tracking (shadow segment)", "shadow segments", "path steps", "-"};
            for (int k = 0; k < 7; k++) fprintf(stderr, "[volpath stats] %-26s wave events %12llu  lanes active %5.1f %%  per sample %.2f\n", names[k], h[2 + 2 * k], h[2 + 2 * k] ? 100.0 * h[3 + 2 * k] / (64.0 * h[2 + 2 * k]) : 0.0, (double)h[3 + 2 * k] / (double)st.samples);
        }
        return;
    }
    // ---- tiny scenes (flat leaf table, shading tables in LDS): the fused persistent kernel, one launch per pass (mega.hip)
    const size_t mega_lds = ljd::mega_smem(sc->dscene, sc->scfg);
    const bool mega_off = getenv("LJ_TUNE_MEGA") && atoi(getenv("LJ_TUNE_MEGA")) == 0;
    if (mega_lds > 0 && !mega_off) {
        const uint64_t pass_samples_max = pix_per_pass * (uint64_t)plan.spp;
        if (ctx->sample_rgb.bytes < pass_samples_max * 12) ctx->sample_rgb.alloc(pass_samples_max * 12);
        if (!ctx->mega_state.p) ctx->mega_state.alloc(64);
        upload_pixel_list(ctx, plan, stream);
        HIP_CHECK(hipEventRecord(ctx->ev_begin, stream));
        int per_cu = ljd::mega_blocks_per_cu(sc->scfg);
        if (const char *e = getenv("LJ_TUNE_MEGA_BLOCKS_PER_CU")) per_cu = std::max(1, atoi(e));
        // camera samples a wave takes off the counter at a time: a multiple of 64 so that a wave's lanes share pixels, chosen so that
        // every wave draws ~16 times and the grid ends together — between 192 (below that the counter traffic shows) and 768:
        // the full bench frame 13.28 ms at 1024 per draw, 13.06 at 768; a 1/8 share of it (one rank of eight: tools/shard_time.py)
        // 2.50 ms at 1024, 2.00 at 192 (ideal 1.65)
        uint32_t grab_max = 768, grab_min = 192;
        bool grab_fixed = false;
        if (const char *e = getenv("LJ_TUNE_MEGA_GRAB")) { grab_max = (uint32_t)std::max(64, atoi(e)) & ~63u; grab_fixed = true; }
        double mega_ms = 0;
        unsigned long long hstats[5] = {};
        for (uint64_t p0 = 0; p0 < n_pix; p0 += pix_per_pass) {
            const uint64_t np = std::min<uint64_t>(pix_per_pass, n_pix - p0);
            const uint64_t total = np * (uint64_t)plan.spp;
            ljd::DPass pass{};
            pass.pixel_list = (const uint32_t *)ctx->pixel_list.p + p0; pass.n_pixels = (uint32_t)np; ljd::set_pass_divisors(pass, (uint32_t)plan.spp, (uint32_t)sc->flat.cam.width);
            pass.seed = plan.seed; pass.sample_rgb = (float *)ctx->sample_rgb.p;
            HIP_CHECK(hipMemsetAsync(ctx->mega_state.p, 0, 64, stream));
            uint32_t grab = grab_max;
            if (!grab_fixed) {
                uint64_t draws = 16;
                if (const char *e = getenv("LJ_TUNE_MEGA_DRAWS")) draws = (uint64_t)std::max(1, atoi(e));
                const uint64_t waves = (uint64_t)ctx->n_cus * per_cu * 4, per_draw = total / (waves * draws);
                grab = (uint32_t)std::min<uint64_t>(grab_max, std::max<uint64_t>(grab_min, per_draw & ~(uint64_t)63));
            }
            // persistent grid: as many workgroups as stay resident, but no more waves than there are `grab`-sized pieces of work
            const uint64_t pieces = (total + grab - 1) / grab;
            const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)ctx->n_cus * per_cu, (pieces + 3) / 4));
            if (timing) HIP_CHECK(hipEventRecord(ctx->ev_k0, stream));
            ljd::launch_mega(ds, pass, sc->scfg, sc->flat.n_spheres > 0, (uint32_t)total, grab, (uint32_t *)ctx->mega_state.p,
                             (unsigned long long *)((char *)ctx->mega_state.p + 8), grid, stream);
            HIP_CHECK(hipGetLastError());
            if (timing) {
                HIP_CHECK(hipEventRecord(ctx->ev_k1, stream)); HIP_CHECK(hipEventSynchronize(ctx->ev_k1));
                float a = 0; HIP_CHECK(hipEventElapsedTime(&a, ctx->ev_k0, ctx->ev_k1)); mega_ms += a;
            }
            st.mega_launches++; st.wavefront_steps++;
            if (rgb_dev) ljd::launch_resolve(pass, (uint32_t)np, rgb_dev, stream);
            unsigned long long *h = ctx->stats_host;
            HIP_CHECK(hipMemcpyAsync(h, (char *)ctx->mega_state.p + 8, 5 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
            if (samples_host) HIP_CHECK(hipMemcpyAsync(samples_host + p0 * (uint64_t)plan.spp * 3, ctx->sample_rgb.p, total * 12, hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            if (h[3] != total)
                throw LjError(LJ_ERR_INTERNAL, "mega launch ended with " + std::to_string(h[3]) + " of " + std::to_string(total) + " samples finished");
            for (int k = 0; k < 5; k++) hstats[k] += h[k];
            st.samples += total;
        }
        HIP_CHECK(hipEventRecord(ctx->ev_end, stream));
        HIP_CHECK(hipEventSynchronize(ctx->ev_end));
        float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
        st.render_ms = ms; st.mega_ms = mega_ms;
        st.bounce_iterations = hstats[0]; st.rays_closest = hstats[1]; st.rays_shadow = hstats[2]; st.path_steps = hstats[4];
        // algorithmic HBM bytes of a mega launch: the finished radiance of every sample (12 B) and its pixel's list entry — the path state
        // never leaves the registers
        st.mega_bytes = st.samples * 12ull + n_pix * 4ull;
        st.queue_bytes = st.mega_bytes;
        return;
    }
    // queue geometry: n_blocks workgroups x seg slots; workgroup b owns slots [b*seg, (b+1)*seg)
    const uint64_t pass_samples_max = pix_per_pass * (uint64_t)plan.spp;
    uint32_t pool = (uint32_t)std::min<uint64_t>(plan.pool, std::max<uint64_t>(pass_samples_max, 256));
    uint32_t blocks_per_cu = 8;
    if (const char *e = getenv("LJ_TUNE_BLOCKS_PER_CU")) blocks_per_cu = (uint32_t)std::max(1, atoi(e));
    // Lanes: the pool, its workgroup segments and the pass's samples are split into four parts that advance
    // independently on four streams.  The extend kernel is bound by VALU issue and moves few bytes, the shade kernel
    // streams the queue and leaves the VALUs mostly idle; with four staggered lanes the GPU always has some of each to
    // run side by side (two lanes: +13 % time on cbox; six or eight: worse again — more streams than hardware queues).
    // (One lane when kernels are timed individually or for tiny renders, two for small ones, or on request.)
    // developer instrumentation of the extend kernel (utilisation counters printed to stderr); off unless asked for
    unsigned long long *xstats = nullptr;
    DevBuf xstats_buf;
    if (getenv("LJ_EXTEND_STATS")) { xstats_buf.alloc(64); HIP_CHECK(hipMemsetAsync(xstats_buf.p, 0, 64, stream)); xstats = (unsigned long long *)xstats_buf.p; }
    uint32_t n_lanes = (timing || xstats || pool < (1u << 20)) ? 1u : (pool < (1u << 22) ? 2u : 4u);
    // the Disney / all-features shade kernels (164 VGPRs, three waves per SIMD) leave an extend kernel no registers to run beside
    // them: lanes would only time-slice the GPU (disney_bsdf 256 spp: 104 ms with one lane, 116 with two; a 64-spp render, which fits
    // the pool at once, is the other way round by 7 %)
    if (sc->scfg.variant >= 4) n_lanes = 1;   // (FeatDisney, FeatAll)
    // a tree beyond the LDS image (nodes fetched through L2): one lane.  Its extend launches fill every CU (five workgroups of 30 KiB
    // LDS and 92 VGPRs each), so the lanes' kernels queue up behind one another instead of running side by side (tools/lanes_sweep.sh,
    // sponza 256 spp: 202.5 ms with one lane, 208.8 with four)
    const bool large_tree = ds.n_nodes > sc->ecfg.lds_nodes;
    if (large_tree) n_lanes = 1;
    if (const char *e = getenv("LJ_TUNE_LANES")) n_lanes = (uint32_t)std::min((int)kMaxLanes, std::max(1, atoi(e)));
    uint32_t n_blocks = std::min<uint32_t>(kMaxBlocks, std::max<uint32_t>(n_lanes, std::min<uint32_t>((uint32_t)ctx->n_cus * blocks_per_cu, pool / 256)));
    n_blocks = (n_blocks / n_lanes) * n_lanes;
    const uint32_t lane_blocks = n_blocks / n_lanes;
    uint32_t seg = ((pool + n_blocks - 1) / n_blocks + 255u) & ~255u;
    const uint32_t n_slots = n_blocks * seg, lane_slots = lane_blocks * seg;
    ensure_queues(ctx, n_slots);
    // feature sets that sort a segment as a whole shade from one set of queue records into another (kernels.hip shade_sorted_segment)
    const bool sort_segments = ljd::shade_sorts_segments(sc->scfg);
    if (sort_segments) {
        const uint32_t cap = (n_slots + 63u) & ~63u;
        if (ctx->queue2_capacity < cap) { ctx->queue_mem2.alloc(queue_bytes(cap)); ctx->queue2_capacity = cap; }
        if (ctx->sort_perm.bytes < (size_t)cap * 4) ctx->sort_perm.alloc((size_t)cap * 4);
        if (ctx->sort_keys.bytes < (size_t)cap) ctx->sort_keys.alloc((size_t)cap);
    }
    // extend: persistent workgroups that draw 256-slot chunks of the queue.  One lane: more workgroups than fit at once, so
    // the hardware keeps every CU as full as registers and LDS allow; several lanes: few enough that the other lanes'
    // shade workgroups find registers beside them (8 extend waves per CU and lane).
    uint32_t ext_per_cu = n_lanes == 1 ? 8 : (n_lanes == 2 ? 4 : 2);
    if (const char *e = getenv("LJ_TUNE_EXTEND_BLOCKS_PER_CU")) ext_per_cu = (uint32_t)std::max(1, atoi(e));
    const uint32_t lane_chunks = lane_slots / 256u;
    const uint32_t ext_grid = std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)ctx->n_cus * ext_per_cu, (lane_chunks + 3) / 4));
    // (the fused tail launches one workgroup per segment, so its overflow stacks are sized for the shade grid)
    // (not for a large tree: the tail's lane-by-lane traversal — one workgroup per segment, leaves tested unpooled — is slower there than
    // the extend / shade launches it replaces: tools/tail_sweep.sh, sponza 64 spp 60.4 -> 56.2 ms, disney_bsdf 256 spp 98.7 -> 97.2 without it)
    bool use_tail = !timing && !xstats && !large_tree && ljd::tail_smem(sc->ecfg, sc->scfg) > 0;
    if (const char *e = getenv("LJ_TUNE_TAIL")) use_tail = use_tail && atoi(e) != 0;
    uint64_t tail_num = 1, tail_den = 4;   // fuse once fewer than tail_num / tail_den of the slots hold a path
    if (const char *e = getenv("LJ_TUNE_TAIL_FRAC")) { tail_num = (uint64_t)std::max(1, atoi(e)); tail_den = 16; }
    const uint32_t spill_grid = std::max<uint32_t>(ext_grid, use_tail ? lane_blocks : 0u);
    int *spill_base = ensure_spill(ctx, sc->ecfg.spill_levels, spill_grid * n_lanes);
    // per lane: work[0] = the extend kernel's draw counter, work[1 + parity] = number of listed chunks; two chunk lists, used
    // alternately, so that a shade launch can append to one while nothing reads it
    if (!ctx->chunk_counter.p) ctx->chunk_counter.alloc(128 * kMaxLanes);
    if (ctx->chunk_list.bytes < (size_t)lane_chunks * 8 * n_lanes) ctx->chunk_list.alloc((size_t)lane_chunks * 8 * n_lanes);
    const ljd::DQueue q_all = carve_queue(ctx->queue_mem.p, ctx->queue_capacity);
    const ljd::DQueue q2_all = sort_segments ? carve_queue(ctx->queue_mem2.p, ctx->queue2_capacity) : q_all;
    struct Lane {
        hipStream_t stream; ljd::DQueue q, q2; uint32_t *sort_perm; uint8_t *sort_keys; ljd::DBlockState *dblocks; ljd::DBlockState *hblocks;
        uint32_t *work; uint32_t *lists[2]; uint32_t parity; int *spill; int *spill_tail; bool done; int batch;
    } lanes[kMaxLanes];
    for (uint32_t l = 0; l < n_lanes; l++) {
        Lane &L = lanes[l];
        L.stream = l == 0 ? stream : ctx->lane_streams[l - 1];
        const size_t o = (size_t)l * lane_slots;
        L.q = q_all; L.q.ro += o; L.q.rd += o; L.q.rs += o; L.q.rh += o; L.q.rw += o; L.q.rl += o; L.q.rn += o; L.q.rg += o;
        L.q2 = q2_all; L.q2.ro += o; L.q2.rd += o; L.q2.rs += o; L.q2.rh += o; L.q2.rw += o; L.q2.rl += o; L.q2.rn += o; L.q2.rg += o;
        L.sort_perm = sort_segments ? (uint32_t *)ctx->sort_perm.p + o : nullptr; L.sort_keys = sort_segments ? (uint8_t *)ctx->sort_keys.p + o : nullptr;
        L.dblocks = (ljd::DBlockState *)ctx->blocks.p + (size_t)l * lane_blocks; L.hblocks = ctx->blocks_host + (size_t)l * lane_blocks;
        L.work = (uint32_t *)ctx->chunk_counter.p + 32 * l;
        L.lists[0] = (uint32_t *)ctx->chunk_list.p + (size_t)l * 2 * lane_chunks; L.lists[1] = L.lists[0] + lane_chunks;
        L.spill = spill_base ? spill_base + (size_t)l * sc->ecfg.spill_levels * spill_grid * 256 : nullptr;
        L.spill_tail = L.spill;
    }
    if (ctx->sample_rgb.bytes < pass_samples_max * 12) ctx->sample_rgb.alloc(pass_samples_max * 12);
    upload_pixel_list(ctx, plan, stream);
    HIP_CHECK(hipEventRecord(ctx->ev_begin, stream));
    double extend_ms = 0, shade_ms = 0;
    for (uint64_t p0 = 0; p0 < n_pix; p0 += pix_per_pass) {
        const uint64_t np = std::min<uint64_t>(pix_per_pass, n_pix - p0);
        const uint64_t total = np * (uint64_t)plan.spp;
        ljd::DPass pass{};
        pass.pixel_list = (const uint32_t *)ctx->pixel_list.p + p0; pass.n_pixels = (uint32_t)np; ljd::set_pass_divisors(pass, (uint32_t)plan.spp, (uint32_t)sc->flat.cam.width);
        pass.seed = plan.seed; pass.sample_rgb = (float *)ctx->sample_rgb.p;
        // contiguous sample ranges per workgroup, multiples of 64 so that a wave's first samples share a pixel
        {
            uint64_t per = ((total + n_blocks - 1) / n_blocks + 63) & ~(uint64_t)63;
            for (uint32_t b = 0; b < n_blocks; b++) {
                ljd::DBlockState bs{};
                uint64_t lo = std::min<uint64_t>(total, (uint64_t)b * per), hi = std::min<uint64_t>(total, (uint64_t)(b + 1) * per);
                bs.next_sample = (uint32_t)lo; bs.end_sample = (uint32_t)hi;
                ctx->blocks_host[b] = bs;
            }
            HIP_CHECK(hipMemcpyAsync(ctx->blocks.p, ctx->blocks_host, sizeof(ljd::DBlockState) * n_blocks, hipMemcpyHostToDevice, stream));
        }
        HIP_CHECK(hipMemsetAsync(ctx->chunk_counter.p, 0, 128 * kMaxLanes, stream));
        if (n_lanes > 1) {  // the second lane starts once the pass's inputs are in place
            HIP_CHECK(hipEventRecord(ctx->ev_fork, stream));
            for (uint32_t l = 1; l < n_lanes; l++) HIP_CHECK(hipStreamWaitEvent(ctx->lane_streams[l - 1], ctx->ev_fork, 0));
        }
        // step 0 is a shade over empty segments: it only generates camera rays
        for (uint32_t l = 0; l < n_lanes; l++) {
            Lane &L = lanes[l];
            L.parity = 0; L.done = false; L.batch = 8;  // steps per host round trip; all per-step state lives on the device
            // stagger: lane l starts when lane l-1 has generated its camera rays, so its shade launches fall on the other
            // lane's extend launches (the two lanes have equal work per step, so the phase offset persists)
            if (l > 0) HIP_CHECK(hipStreamWaitEvent(L.stream, ctx->ev_join, 0));
            ljd::launch_shade(ds, pass, L.q, L.q2, L.sort_perm, L.sort_keys, L.dblocks, lane_blocks, seg, sc->scfg, L.work, L.lists[0], 0, ext_grid * 4u, L.stream);
            if (sort_segments) std::swap(L.q, L.q2);   // (L.q: the set the paths are in now)
            if (l + 1 < n_lanes) HIP_CHECK(hipEventRecord(ctx->ev_join, L.stream));
        }
        // Round-robin over the lanes: look at a lane's block states only when its previous batch has drained, and give it
        // its next batch at once, so the other lane's kernels keep the GPU busy during this lane's host round trip.
        bool pending[kMaxLanes] = {};
        for (int guard = 0; guard < (1 << 21); guard++) {
            bool any = false;
            for (uint32_t l = 0; l < n_lanes; l++) {
                Lane &L = lanes[l];
                if (L.done) continue;
                if (pending[l]) {
                    HIP_CHECK(hipStreamSynchronize(L.stream));
                    pending[l] = false;
                    uint64_t alive = 0;
                    for (uint32_t b = 0; b < lane_blocks; b++) alive += L.hblocks[b].count;
                    L.done = alive == 0;
                    // the long tail (a few deep paths left): launches are nearly empty, so look less often
                    L.batch = alive * 64ull < (uint64_t)lane_slots ? 16 : 8;
                    if (L.done) continue;
                    // ... or not at all: once every camera sample has been started and the segments are mostly empty, the
                    // rest of the lane's paths finish inside one fused launch (k_tail)
                    uint64_t to_start = 0;
                    for (uint32_t b = 0; b < lane_blocks; b++) to_start += L.hblocks[b].end_sample - L.hblocks[b].next_sample;
                    if (use_tail && to_start == 0 && alive * tail_den < (uint64_t)lane_slots * tail_num) {
                        ljd::launch_tail(ds, pass, L.q, L.dblocks, lane_blocks, seg, sc->ecfg, sc->scfg, L.spill_tail, L.stream);
                        HIP_CHECK(hipGetLastError());
                        HIP_CHECK(hipMemcpyAsync(L.hblocks, L.dblocks, sizeof(ljd::DBlockState) * lane_blocks, hipMemcpyDeviceToHost, L.stream));
                        pending[l] = true; any = true;
                        continue;
                    }
                }
                any = true;
                for (int b = 0; b < L.batch; b++) {
                    if (timing) HIP_CHECK(hipEventRecord(ctx->ev_k0, L.stream));
                    ljd::launch_extend(ds, L.q, L.dblocks, ext_grid, seg, L.work, L.lists[L.parity], L.parity, sc->ecfg, L.spill, xstats, L.stream);
                    if (timing) HIP_CHECK(hipEventRecord(ctx->ev_k1, L.stream));
                    L.parity ^= 1u;
                    ljd::launch_shade(ds, pass, L.q, L.q2, L.sort_perm, L.sort_keys, L.dblocks, lane_blocks, seg, sc->scfg, L.work, L.lists[L.parity], L.parity, ext_grid * 4u, L.stream);
                    if (sort_segments) std::swap(L.q, L.q2);
                    if (timing) {
                        HIP_CHECK(hipEventRecord(ctx->ev_end, L.stream));
                        HIP_CHECK(hipEventSynchronize(ctx->ev_end));
                        float a = 0, c = 0;
                        HIP_CHECK(hipEventElapsedTime(&a, ctx->ev_k0, ctx->ev_k1)); HIP_CHECK(hipEventElapsedTime(&c, ctx->ev_k1, ctx->ev_end));
                        extend_ms += a; shade_ms += c;
                    }
                    st.extend_launches++; st.shade_launches++;
                    if (l == 0) st.wavefront_steps++;
                }
                HIP_CHECK(hipGetLastError());
                HIP_CHECK(hipMemcpyAsync(L.hblocks, L.dblocks, sizeof(ljd::DBlockState) * lane_blocks, hipMemcpyDeviceToHost, L.stream));
                pending[l] = true;
            }
            if (!any) break;
        }
        uint64_t samples_done = 0, path_steps = 0, shadow_rays = 0;
        for (uint32_t b = 0; b < n_blocks; b++) {
            const ljd::DBlockState &bs = ctx->blocks_host[b];
            samples_done += bs.samples_done; path_steps += bs.path_steps; shadow_rays += bs.rays_shadow;
            st.bounce_iterations += bs.bounce_iterations; st.rays_closest += bs.rays_closest; st.rays_shadow += bs.rays_shadow;
        }
        if (samples_done != total)
            throw LjError(LJ_ERR_INTERNAL, "wavefront loop ended with " + std::to_string(samples_done) + " of " + std::to_string(total) + " samples finished");
        st.samples += total;
        // algorithmic queue traffic per live path-step: extend reads ro, rd (32 B), rs (16 B) when a shadow ray
        // is pending, and writes rh (16 B); shade reads 7 records (112 B) and writes 7 (112 B), plus 12 B per finished sample
        st.path_steps += path_steps;
        st.extend_bytes += path_steps * 48ull + shadow_rays * 16ull;
        st.shade_bytes += path_steps * 224ull + total * 12ull;
        // (both lanes were synchronised with the host above, so the caller's stream may read what the second lane wrote)
        if (rgb_dev) ljd::launch_resolve(pass, (uint32_t)np, rgb_dev, stream);
        if (samples_host) {
            HIP_CHECK(hipMemcpyAsync(samples_host + p0 * (uint64_t)plan.spp * 3, ctx->sample_rgb.p, total * 12, hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
        }
    }
    HIP_CHECK(hipEventRecord(ctx->ev_end, stream));
    HIP_CHECK(hipEventSynchronize(ctx->ev_end));
    float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
    st.render_ms = ms; st.extend_ms = extend_ms; st.shade_ms = shade_ms;
    if (xstats) {
        unsigned long long h[8];
        HIP_CHECK(hipMemcpy(h, xstats, 64, hipMemcpyDeviceToHost));
        fprintf(stderr, "[extend stats] rays %llu | outer iters %llu busy lanes %.1f%% | node steps %llu (%.2f/ray) lane use %.1f%% | prim rounds %llu (%.2f tests/ray) lane use %.1f%% | refills %llu\n",
                h[7], h[0], 100.0 * h[1] / (64.0 * h[0]), h[2], (double)h[3] / h[7], 100.0 * h[3] / (64.0 * h[2]), h[4], (double)h[5] / h[7], 100.0 * h[5] / (64.0 * h[4]), h[6]);
    }
    st.queue_bytes = st.extend_bytes + st.shade_bytes;
}

} // namespace

extern "C" {

int lj_context_create(int device_id, lj_context **out) {
    return lj::guard([&]() {
        if (!out) throw LjError(LJ_ERR_INVALID_ARG, "lj_context_create: null out");
        *out = nullptr;
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n == 0) throw LjError(LJ_ERR_DEVICE, std::string("no HIP device available (") + hipGetErrorString(e) + "); this library has no CPU path");
        if (device_id < 0 || device_id >= n) throw LjError(LJ_ERR_INVALID_ARG, "device id out of range");
        auto ctx = std::make_unique<lj_context>();
        ctx->device = device_id;
        HIP_CHECK(hipSetDevice(device_id));
        hipDeviceProp_t prop; HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
        if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
            throw LjError(LJ_ERR_DEVICE, std::string("device is ") + prop.gcnArchName + "; this build contains gfx950 code objects only");
        ctx->n_cus = prop.multiProcessorCount;
        HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        for (auto &ls : ctx->lane_streams) HIP_CHECK(hipStreamCreateWithFlags(&ls, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_caller, hipEventDisableTiming));
        ctx->blocks.alloc(sizeof(ljd::DBlockState) * kMaxBlocks);
        HIP_CHECK(hipHostMalloc((void **)&ctx->blocks_host, sizeof(ljd::DBlockState) * kMaxBlocks, hipHostMallocDefault));
        HIP_CHECK(hipHostMalloc((void **)&ctx->stats_host, 64, hipHostMallocDefault));
        HIP_CHECK(hipEventCreate(&ctx->ev_begin)); HIP_CHECK(hipEventCreate(&ctx->ev_end));
        HIP_CHECK(hipEventCreate(&ctx->ev_k0)); HIP_CHECK(hipEventCreate(&ctx->ev_k1));
        *out = ctx.release();
    });
}

void lj_context_destroy(lj_context *ctx) {
    if (!ctx) return;
    if (ctx->live_scenes > 0) { ctx->doomed = true; return; }   // released by the last lj_scene_destroy
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
    for (auto &ls : ctx->lane_streams) if (ls) { (void)hipStreamSynchronize(ls); (void)hipStreamDestroy(ls); }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->ev_caller) (void)hipEventDestroy(ctx->ev_caller);
    if (ctx->blocks_host) (void)hipHostFree(ctx->blocks_host);
    if (ctx->stats_host) (void)hipHostFree(ctx->stats_host);
    for (hipEvent_t e : {ctx->ev_begin, ctx->ev_end, ctx->ev_k0, ctx->ev_k1}) if (e) (void)hipEventDestroy(e);
    delete ctx;
}

int lj_scene_upload(lj_context *ctx, const LjSceneDesc *desc, lj_scene **out) {
    return lj::guard([&]() {
        if (!ctx || !desc || !out) throw LjError(LJ_ERR_INVALID_ARG, "lj_scene_upload: null argument");
        *out = nullptr;
        set_device(ctx);
        auto sc = std::make_unique<lj_scene>();
        sc->ctx = ctx;
        sc->flat = lj::flatten_scene(*desc);
        lj::FlatScene &F = sc->flat;
        hipStream_t s = ctx->stream;
        upload(sc->nodes, F.nodes, s);
        // BVH8 nodes: 80 bytes each; LJ_TUNE_NODE8_STRIDE=128 lays them out one per 128-byte cache line (a node then never straddles two lines)
        int node8_stride = (int)sizeof(ljd::DNode8);
        if (const char *e = getenv("LJ_TUNE_NODE8_STRIDE")) node8_stride = atoi(e) >= 128 ? 128 : 80;
        std::vector<unsigned char> nodes8_padded;
        if (node8_stride == (int)sizeof(ljd::DNode8)) upload(sc->nodes8, F.nodes8, s);
        else {
            nodes8_padded.assign(F.nodes8.size() * (size_t)node8_stride, 0);
            for (size_t i = 0; i < F.nodes8.size(); i++) memcpy(&nodes8_padded[i * (size_t)node8_stride], &F.nodes8[i], sizeof(ljd::DNode8));
            upload(sc->nodes8, nodes8_padded, s);
        }
        upload(sc->leaf_prims, F.leaf_prims, s); upload(sc->prims, F.prims, s); upload(sc->spheres, F.spheres, s);
        upload(sc->materials, F.materials, s); upload(sc->lights, F.lights, s); upload(sc->light_cdf, F.light_cdf, s);
        upload(sc->light_tris, F.light_tris, s); upload(sc->light_tri_cdf, F.light_tri_cdf, s);
        upload(sc->images3, F.images3, s); upload(sc->images1, F.images1, s); upload(sc->texels, F.texels, s); upload(sc->env_tables, F.env_tables, s);
        upload(sc->media, F.media, s); upload(sc->volume_data, F.volume_data, s); upload(sc->shape_media, F.shape_media, s);
        upload(sc->scan_leaves, F.scan_leaves, s);
        HIP_CHECK(hipStreamSynchronize(s));
        ljd::DScene d = F.host_view();
        d.nodes = (const ljd::DNode4 *)sc->nodes.p; d.nodes8 = (const ljd::DNode8 *)sc->nodes8.p; d.node8_stride = node8_stride; d.leaf_prims = (const ljd::DPrim *)sc->leaf_prims.p; d.prims = (const ljd::DPrimShade *)sc->prims.p;
        d.spheres = (const ljd::DSphere *)sc->spheres.p; d.materials = (const ljd::DMaterial *)sc->materials.p; d.lights = (const ljd::DLight *)sc->lights.p;
        d.light_cdf = (const float *)sc->light_cdf.p; d.light_tris = (const ljd::DLightTri *)sc->light_tris.p; d.light_tri_cdf = (const float *)sc->light_tri_cdf.p;
        d.images3 = (const ljd::DImage *)sc->images3.p; d.images1 = (const ljd::DImage *)sc->images1.p; d.texels = (const float *)sc->texels.p; d.env_tables = (const float *)sc->env_tables.p; d.env_marg = d.env_tables + F.env_marg_first;
        d.media = (const ljd::DMedium *)sc->media.p; d.volume_data = (const float *)sc->volume_data.p; d.shape_media = (const int32_t *)sc->shape_media.p;
        d.scan_leaves = F.scan_leaves.empty() ? nullptr : (const ljd::DScanLeaf *)sc->scan_leaves.p;
        sc->dscene = d;
        {   int64_t g = 0;
            for (int si = 0; si < desc->n_shapes; si++) { sc->shape_first_gprim.push_back(g); g += desc->shapes[si].kind == LJ_SHAPE_SPHERE ? 1 : desc->shapes[si].n_triangles; }
            sc->shape_first_gprim.push_back(g);
        }
        if (F.bvh_depth > ljd::max_stack_depth())
            throw LjError(LJ_ERR_INTERNAL, "BVH depth " + std::to_string(F.bvh_depth) + " exceeds the traversal stack");
        sc->ecfg = ljd::extend_config((int)F.nodes.size(), (int)F.leaf_prims.size(), F.bvh_depth, (int)F.n_spheres, (int)F.nodes8.size(), F.bvh8_depth, /* open scene: */ F.envmap_light_id >= 0);
        if (const char *e = getenv("LJ_TUNE_REFILL")) sc->ecfg.refill_min = (uint32_t)atoi(e);
        if (const char *e = getenv("LJ_TUNE_MINDESC")) sc->ecfg.min_descending = (uint32_t)atoi(e);
        sc->scfg = ljd::shade_config(F.prims.size(), F.materials.size(), F.lights.size(), F.light_tris.size(), F.light_tri_cdf.size(), F.images3.size(), F.images1.size(), (size_t)F.env_marg_count);
        {   // which Material / Texture / Light alternatives the scene holds decides the shade kernel instantiation
            uint32_t kinds = 0; bool textured = false, sphere_lights = false;
            for (const auto &m : F.materials) { kinds |= 1u << m.kind; for (int t = 0; t < 12; t++) textured = textured || m.tex[t].kind != 0; }
            for (const auto &l : F.lights) sphere_lights = sphere_lights || (l.kind == 0 && l.is_sphere);
            sc->feat_kinds = kinds; sc->feat_textured = textured; sc->feat_envmap = F.envmap_light_id >= 0; sc->feat_sphere_lights = sphere_lights;
            sc->scfg.variant = ljd::shade_variant(kinds, textured, F.envmap_light_id >= 0, sphere_lights);
            if (const char *e = getenv("LJ_TUNE_SHADE_VARIANT")) sc->scfg.variant = std::max(sc->scfg.variant, atoi(e));
            if (sc->scfg.smem == 0) sc->scfg.variant = ljd::kShadeVariantAll;   // tables too large to stage: one instantiation serves that case
        }
        ctx->live_scenes++;
        *out = sc.release();
    });
}

void lj_scene_destroy(lj_scene *scene) {
    if (!scene) return;
    lj_context *ctx = scene->ctx;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    delete scene;
    if (--ctx->live_scenes == 0 && ctx->doomed) lj_context_destroy(ctx);
}

int lj_render_device(lj_scene *scene, const LjRenderArgs *args, float *rgb_device, void *hip_stream) {
    return lj::guard([&]() {
        if (!scene || !rgb_device) throw LjError(LJ_ERR_INVALID_ARG, "lj_render_device: null argument");
        lj_context *ctx = scene->ctx;
        set_device(ctx);
        // The render always runs on the context's own streams — its lanes were created together and sit on distinct hardware
        // queues; a caller's stream created later may share a queue with one of them, which serialises two lanes (measured:
        // 26 -> 32 ms per cbox render) — and is ordered against the caller's stream by events on both sides.
        hipStream_t caller = (hipStream_t)hip_stream, s = ctx->stream;
        if (caller) { HIP_CHECK(hipEventRecord(ctx->ev_caller, caller)); HIP_CHECK(hipStreamWaitEvent(s, ctx->ev_caller, 0)); }
        RenderPlan plan = make_plan(scene, args);
        const size_t fb = (size_t)scene->flat.cam.width * scene->flat.cam.height * 3 * sizeof(float);
        HIP_CHECK(hipMemsetAsync(rgb_device, 0, fb, s));
        run_render(scene, plan, rgb_device, nullptr, s, args && (args->flags & 1u));
        if (caller) { HIP_CHECK(hipEventRecord(ctx->ev_caller, s)); HIP_CHECK(hipStreamWaitEvent(caller, ctx->ev_caller, 0)); }
    });
}

int lj_render(lj_scene *scene, const LjRenderArgs *args, float *rgb_host) {
    return lj::guard([&]() {
        if (!scene || !rgb_host) throw LjError(LJ_ERR_INVALID_ARG, "lj_render: null argument");
        lj_context *ctx = scene->ctx;
        set_device(ctx);
        const size_t fb = (size_t)scene->flat.cam.width * scene->flat.cam.height * 3 * sizeof(float);
        if (ctx->frame.bytes < fb) ctx->frame.alloc(fb);
        RenderPlan plan = make_plan(scene, args);
        HIP_CHECK(hipMemsetAsync(ctx->frame.p, 0, fb, ctx->stream));
        run_render(scene, plan, (float *)ctx->frame.p, nullptr, ctx->stream, args && (args->flags & 1u));
        HIP_CHECK(hipMemcpyAsync(rgb_host, ctx->frame.p, fb, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int lj_render_samples(lj_scene *scene, const LjRenderArgs *args, float *radiance_host) {
    return lj::guard([&]() {
        if (!scene || !radiance_host || !args) throw LjError(LJ_ERR_INVALID_ARG, "lj_render_samples: null argument");
        if (!(args->crop_x1 > args->crop_x0 && args->crop_y1 > args->crop_y0)) throw LjError(LJ_ERR_INVALID_ARG, "lj_render_samples needs a crop window");
        if (args->world_size > 1) throw LjError(LJ_ERR_INVALID_ARG, "lj_render_samples is single-rank");
        set_device(scene->ctx);
        RenderPlan plan = make_plan(scene, args);
        run_render(scene, plan, nullptr, radiance_host, scene->ctx->stream, false);
    });
}

static int trace_batch(lj_scene *scene, int64_t n, const LjRay *rays_host, LjHit *hits_host, uint8_t *occ_host) {
    return lj::guard([&]() {
        if (!scene || !rays_host || n < 0) throw LjError(LJ_ERR_INVALID_ARG, "trace: bad argument");
        if (n == 0) return;
        lj_context *ctx = scene->ctx;
        set_device(ctx);
        DevBuf rays, out;
        rays.alloc((size_t)n * sizeof(LjRay));
        out.alloc((size_t)n * (hits_host ? sizeof(LjHit) : 1));
        HIP_CHECK(hipMemcpyAsync(rays.p, rays_host, (size_t)n * sizeof(LjRay), hipMemcpyHostToDevice, ctx->stream));
        int grid = (int)std::min<int64_t>((n + 255) / 256, ctx->n_cus * 4);
        // a tiny scene is traced by the leaf scan of mega.hip in a render, so that is what answers here too (LJ_TUNE_MEGA=0: the BVH
        // traversal of k_extend, as for every other scene)
        const bool scan = scene->dscene.n_scan_leaves > 0 && !(getenv("LJ_TUNE_MEGA") && atoi(getenv("LJ_TUNE_MEGA")) == 0);
        HIP_CHECK(hipEventRecord(ctx->ev_begin, ctx->stream));
        if (scan) ljd::launch_trace_rays_scan(scene->dscene, rays.p, n, hits_host ? out.p : nullptr, hits_host ? nullptr : (unsigned char *)out.p, grid, ctx->stream);
        else ljd::launch_trace_rays(scene->dscene, rays.p, n, hits_host ? out.p : nullptr, hits_host ? nullptr : (unsigned char *)out.p, scene->ecfg, ensure_spill(ctx, scene->ecfg.spill_levels, (uint32_t)grid), grid, ctx->stream);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipEventRecord(ctx->ev_end, ctx->stream));
        if (hits_host) HIP_CHECK(hipMemcpyAsync(hits_host, out.p, (size_t)n * sizeof(LjHit), hipMemcpyDeviceToHost, ctx->stream));
        else HIP_CHECK(hipMemcpyAsync(occ_host, out.p, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
        scene->stats = LjStats{}; scene->stats.render_ms = ms; scene->stats.rays_closest = hits_host ? (uint64_t)n : 0; scene->stats.rays_shadow = hits_host ? 0 : (uint64_t)n;   // (the query kernel's device time)
    });
}
int lj_intersect(lj_scene *scene, int64_t n, const LjRay *rays_host, LjHit *hits_host) {
    if (!hits_host) { lj::set_last_error("lj_intersect: null output"); return LJ_ERR_INVALID_ARG; }
    return trace_batch(scene, n, rays_host, hits_host, nullptr);
}
int lj_occluded(lj_scene *scene, int64_t n, const LjRay *rays_host, uint8_t *occluded_host) {
    if (!occluded_host) { lj::set_last_error("lj_occluded: null output"); return LJ_ERR_INVALID_ARG; }
    return trace_batch(scene, n, rays_host, nullptr, occluded_host);
}

int lj_get_stats(const lj_scene *scene, LjStats *out) {
    if (!scene || !out) { lj::set_last_error("lj_get_stats: null argument"); return LJ_ERR_INVALID_ARG; }
    *out = scene->stats;
    return LJ_OK;
}

int lj_scene_info(const lj_scene *scene, LjSceneInfo *out) {
    if (!scene || !out) { lj::set_last_error("lj_scene_info: null argument"); return LJ_ERR_INVALID_ARG; }
    const lj::FlatScene &F = scene->flat;
    LjSceneInfo i{};
    i.width = F.cam.width; i.height = F.cam.height; i.spp = F.spp; i.max_depth = F.max_depth; i.rr_depth = F.rr_depth; i.integrator = F.integrator;
    i.n_triangles = F.n_triangles; i.n_spheres = F.n_spheres; i.n_bvh_nodes = (int64_t)F.nodes.size();
    i.bounds_radius = F.bounds_radius; for (int k = 0; k < 3; k++) i.bounds_center[k] = F.bounds_center[k];
    i.shadow_epsilon = F.shadow_epsilon;
    *out = i;
    return LJ_OK;
}

} // extern "C"
