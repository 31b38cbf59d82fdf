// Batched per-object queries on the device (include/lajolla_hip.h, "per-object queries"): the reference's free functions
// on Material / Light / Shape / Camera / Filter / Texture / Frame / pcg32, answered by the very device functions the shade
// kernels are built from (dshade.h, dmath.h), one query per lane.  They exist so that the HIP shading code can be held value
// for value against vectors the reference's own functions produced (tests/golden/*.json) and so that the reference's
// unit tests (src/tests/materials.cpp, filter.cpp, frame.cpp, mipmap.cpp) can be re-run against the GPU — instead of
// meeting the reference only inside whole paths.  No oracle, no CPU arithmetic: the host code here only copies.
#include "api_internal.h"
#include "dshade.h"

namespace ljd {
namespace {

constexpr int kQBlock = 256;

__device__ __forceinline__ void st3(float *o, f3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

__device__ __forceinline__ DVertex vertex_in(const LjVertex &v) {
    DVertex vx;
    vx.position = ld3(v.position); vx.gn = ld3(v.geometry_normal);
    vx.frame.x = ld3(v.frame_x); vx.frame.y = ld3(v.frame_y); vx.frame.n = ld3(v.frame_n);
    vx.u = v.uv[0]; vx.v = v.uv[1]; vx.uv_screen_size = v.uv_screen_size;
    vx.material_id = v.material_id; vx.light_id = v.light_id; vx.gprim = 0; vx.is_sphere = false;
    return vx;
}

template <class Ft>
__global__ void __launch_bounds__(kQBlock) k_bsdf_queries(DScene sc, const LjBsdfQuery *q, LjBsdfResult *r, long long n) {
    for (long long i = (long long)blockIdx.x * kQBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kQBlock) {
        const LjBsdfQuery qi = q[i];
        const DVertex vx = vertex_in(qi.vertex);
        const DMaterial &m = sc.materials[vx.material_id];
        const f3 din = ld3(qi.dir_in), dout = ld3(qi.dir_out);
        f3 f; float pdf;
        bsdf_eval_pdf<Ft>(sc, m, din, dout, vx, f, pdf);
        const BsdfSample bs = bsdf_sample<Ft>(sc, m, din, vx, qi.rnd_uv[0], qi.rnd_uv[1], qi.rnd_w);
        LjBsdfResult o;
        st3(o.eval, f); o.pdf = pdf; st3(o.sample_dir, bs.dir_out); o.sample_eta = bs.eta; o.sample_roughness = bs.roughness; o.sample_valid = bs.valid ? 1 : 0;
        r[i] = o;
    }
}

template <class Ft>
__global__ void __launch_bounds__(kQBlock) k_light_queries(DScene sc, const LjLightQuery *q, LjLightResult *r, long long n) {
    for (long long i = (long long)blockIdx.x * kQBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kQBlock) {
        const LjLightQuery qi = q[i];
        const DLight &L = sc.lights[qi.light_id];
        const f3 ref = ld3(qi.ref);
        const LightSample ls = sample_point_on_light<Ft>(sc, L, ref, qi.rnd_uv[0], qi.rnd_uv[1], qi.rnd_w);
        LjLightResult o;
        st3(o.position, ls.position); st3(o.normal, ls.normal);
        o.pdf = pdf_point_on_light<Ft>(sc, L, ls.position, ls.normal, ref);
        st3(o.emission, light_emission<Ft>(sc, L, ld3(qi.view_dir), ls.normal));
        o.pmf = L.pmf; o._pad = 0;
        o.position_d[0] = ls.dpos[0]; o.position_d[1] = ls.dpos[1]; o.position_d[2] = ls.dpos[2];
        r[i] = o;
    }
}

__global__ void __launch_bounds__(kQBlock) k_sample_light_queries(DScene sc, const float *u, int32_t *id, long long n) {
    for (long long i = (long long)blockIdx.x * kQBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kQBlock)
        id[i] = sample_cdf(sc.light_cdf, sc.n_lights, u[i]);
}

// gprim_of_shape: global primitive id of each shape's first primitive
template <class Ft>
__global__ void __launch_bounds__(kQBlock) k_vertex_queries(DScene sc, const long long *gprim_of_shape, const LjHitQuery *q, LjHitResult *r, long long n) {
    for (long long i = (long long)blockIdx.x * kQBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kQBlock) {
        const LjHitQuery qi = q[i];
        const int gprim = (int)(gprim_of_shape[qi.shape_id] + qi.primitive_id);
        const f3 org = ld3(qi.org), dir = ld3(qi.dir);
        const DVertex vx = build_vertex(sc, org, dir, qi.t, qi.u, qi.v, gprim, qi.ray_spread);
        LjHitResult o;
        st3(o.vertex.position, vx.position); st3(o.vertex.geometry_normal, vx.gn);
        st3(o.vertex.frame_x, vx.frame.x); st3(o.vertex.frame_y, vx.frame.y); st3(o.vertex.frame_n, vx.frame.n);
        o.vertex.uv_screen_size = vx.uv_screen_size;
        // (mean curvature: the value the MeanCurvature auxiliary integrator reports, triangle_mesh.inl:128-146 / sphere.inl:258)
        DScene s1 = sc; s1.init_spread = qi.ray_spread;
        o.vertex.mean_curvature = aux_value(s1, 2, org, dir, qi.t, qi.u, qi.v, gprim).x;
        o.vertex.uv[0] = vx.u; o.vertex.uv[1] = vx.v;
        o.vertex.material_id = vx.material_id; o.vertex.light_id = vx.light_id;
        o.vertex.shape_id = sc.prims[gprim].shape_id; o.vertex.primitive_id = sc.prims[gprim].prim_id;
        f3 em = mk3(0, 0, 0);
        if (vx.light_id >= 0) em = light_emission<Ft>(sc, sc.lights[vx.light_id], -dir, vx.gn);
        st3(o.emission, em); o._pad = 0;
        r[i] = o;
    }
}

__global__ void __launch_bounds__(kQBlock) k_primary_queries(DScene sc, const LjPrimaryQuery *q, LjPrimaryResult *r, long long n) {
    for (long long i = (long long)blockIdx.x * kQBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kQBlock) {
        LjPrimaryResult o;
        st3(o.org, ld3(sc.cam.org));
        st3(o.dir, camera_primary_dir(sc.cam, q[i].x, q[i].y, q[i].jx, q[i].jy));
        r[i] = o;
    }
}

__global__ void __launch_bounds__(kQBlock) k_filter_queries(const LjFilterQuery *q, float *out, long long n) {
    for (long long i = (long long)blockIdx.x * kQBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kQBlock) {
        float ox, oy;
        filter_sample(q[i].kind, q[i].param, q[i].rnd[0], q[i].rnd[1], ox, oy);
        out[2 * i] = ox; out[2 * i + 1] = oy;
    }
}

__global__ void __launch_bounds__(kQBlock) k_pcg32_queries(const uint64_t *streams, uint64_t seed, int count, uint32_t *u32, float *real, long long n) {
    for (long long i = (long long)blockIdx.x * kQBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kQBlock) {
        const uint64_t inc = pcg32_inc(streams[i]);
        uint64_t st = pcg32_init(streams[i], seed);
        uint64_t st2 = st;
        for (int k = 0; k < count; k++) {
            u32[i * count + k] = pcg32_next(st, inc);
            if (real) real[i * count + k] = pcg32_real(st2, inc);
        }
    }
}

struct DTexQuery { DTexture tex; double u, v; float footprint; int32_t spectrum; };
__global__ void __launch_bounds__(kQBlock) k_texture_queries(DScene sc, const DTexQuery *q, float *out, long long n) {
    for (long long i = (long long)blockIdx.x * kQBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kQBlock) {
        const f3 c = eval_texture<FeatAll>(sc, q[i].tex, q[i].spectrum != 0, q[i].u, q[i].v, q[i].footprint);
        st3(out + 3 * i, c);
    }
}

__global__ void __launch_bounds__(kQBlock) k_frame_queries(const LjFrameQuery *q, LjFrameResult *r, long long n) {
    for (long long i = (long long)blockIdx.x * kQBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kQBlock) {
        const Frame3 f = make_frame(ld3(q[i].n));
        const f3 v = ld3(q[i].v);
        LjFrameResult o;
        st3(o.x, f.x); st3(o.y, f.y); st3(o.to_local, to_local(f, v)); st3(o.to_world, to_world(f, v));
        r[i] = o;
    }
}

} // namespace
} // namespace ljd

namespace {

int grid_for(const lj_context *ctx, int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + ljd::kQBlock - 1) / ljd::kQBlock, (int64_t)ctx->n_cus * 8)); }

// copy in, launch, copy out; `launch(in_dev, out_dev, stream)`
template <class In, class Out, class Launch>
void run_queries(lj_context *ctx, int64_t n, const In *in_host, size_t in_per, Out *out_host, size_t out_per, Launch &&launch) {
    if (n < 0) throw LjError(LJ_ERR_INVALID_ARG, "negative query count");
    if (n == 0) return;
    if (!in_host || !out_host) throw LjError(LJ_ERR_INVALID_ARG, "null query / result array");
    HIP_CHECK(hipSetDevice(ctx->device));
    DevBuf in, out;
    in.alloc((size_t)n * in_per * sizeof(In)); out.alloc((size_t)n * out_per * sizeof(Out));
    HIP_CHECK(hipMemcpyAsync(in.p, in_host, (size_t)n * in_per * sizeof(In), hipMemcpyHostToDevice, ctx->stream));
    launch((const In *)in.p, (Out *)out.p, ctx->stream);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(out_host, out.p, (size_t)n * out_per * sizeof(Out), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
}

int resolve_variant(const lj_scene *sc, int variant) {
    if (variant < 0) return sc->scfg.variant;
    if (variant >= ljd::kNumShadeVariants) throw LjError(LJ_ERR_INVALID_ARG, "shade variant out of range");
    if (!ljd::variant_covers(variant, sc->feat_kinds, sc->feat_textured, sc->feat_envmap, sc->feat_sphere_lights))
        throw LjError(LJ_ERR_INVALID_ARG, "shade variant " + std::to_string(variant) + " does not cover the Material / Texture / Light alternatives of this scene");
    return variant;
}

} // namespace

extern "C" {

int lj_shade_variant_count(void) { return ljd::kNumShadeVariants; }
int lj_scene_shade_variant(const lj_scene *scene) { return scene ? scene->scfg.variant : LJ_ERR_INVALID_ARG; }

int lj_bsdf_queries(lj_scene *scene, int variant, int64_t n, const LjBsdfQuery *q, LjBsdfResult *r) {
    return lj::guard([&]() {
        if (!scene) throw LjError(LJ_ERR_INVALID_ARG, "lj_bsdf_queries: null scene");
        const int v = resolve_variant(scene, variant);
        for (int64_t i = 0; i < n && q; i++)
            if (q[i].vertex.material_id < 0 || q[i].vertex.material_id >= (int)scene->flat.materials.size()) throw LjError(LJ_ERR_INVALID_ARG, "lj_bsdf_queries: material_id out of range");
        run_queries(scene->ctx, n, q, 1, r, 1, [&](const LjBsdfQuery *qd, LjBsdfResult *rd, hipStream_t s) {
            ljd::with_shade_variant(v, [&](auto ft) {
                hipLaunchKernelGGL(ljd::k_bsdf_queries<decltype(ft)>, dim3(grid_for(scene->ctx, n)), dim3(ljd::kQBlock), 0, s, scene->dscene, qd, rd, (long long)n);
            });
        });
    });
}

int lj_light_queries(lj_scene *scene, int variant, int64_t n, const LjLightQuery *q, LjLightResult *r) {
    return lj::guard([&]() {
        if (!scene) throw LjError(LJ_ERR_INVALID_ARG, "lj_light_queries: null scene");
        const int v = resolve_variant(scene, variant);
        for (int64_t i = 0; i < n && q; i++)
            if (q[i].light_id < 0 || q[i].light_id >= (int)scene->flat.lights.size()) throw LjError(LJ_ERR_INVALID_ARG, "lj_light_queries: light_id out of range");
        run_queries(scene->ctx, n, q, 1, r, 1, [&](const LjLightQuery *qd, LjLightResult *rd, hipStream_t s) {
            ljd::with_shade_variant(v, [&](auto ft) {
                hipLaunchKernelGGL(ljd::k_light_queries<decltype(ft)>, dim3(grid_for(scene->ctx, n)), dim3(ljd::kQBlock), 0, s, scene->dscene, qd, rd, (long long)n);
            });
        });
    });
}

int lj_sample_light_queries(lj_scene *scene, int64_t n, const float *u, int32_t *id) {
    return lj::guard([&]() {
        if (!scene) throw LjError(LJ_ERR_INVALID_ARG, "lj_sample_light_queries: null scene");
        if (scene->flat.lights.empty()) throw LjError(LJ_ERR_INVALID_ARG, "lj_sample_light_queries: the scene has no lights");
        run_queries(scene->ctx, n, u, 1, id, 1, [&](const float *ud, int32_t *idd, hipStream_t s) {
            hipLaunchKernelGGL(ljd::k_sample_light_queries, dim3(grid_for(scene->ctx, n)), dim3(ljd::kQBlock), 0, s, scene->dscene, ud, idd, (long long)n);
        });
    });
}

int lj_vertex_queries(lj_scene *scene, int variant, int64_t n, const LjHitQuery *q, LjHitResult *r) {
    return lj::guard([&]() {
        if (!scene) throw LjError(LJ_ERR_INVALID_ARG, "lj_vertex_queries: null scene");
        const int v = resolve_variant(scene, variant);
        const int n_shapes = (int)scene->shape_first_gprim.size() - 1;
        for (int64_t i = 0; i < n && q; i++) {
            const int sid = q[i].shape_id;
            if (sid < 0 || sid >= n_shapes) throw LjError(LJ_ERR_INVALID_ARG, "lj_vertex_queries: shape_id out of range");
            if (q[i].primitive_id < 0 || q[i].primitive_id >= scene->shape_first_gprim[sid + 1] - scene->shape_first_gprim[sid])
                throw LjError(LJ_ERR_INVALID_ARG, "lj_vertex_queries: primitive_id out of range");
        }
        DevBuf first;
        HIP_CHECK(hipSetDevice(scene->ctx->device));
        first.alloc(scene->shape_first_gprim.size() * sizeof(int64_t));
        HIP_CHECK(hipMemcpy(first.p, scene->shape_first_gprim.data(), scene->shape_first_gprim.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        run_queries(scene->ctx, n, q, 1, r, 1, [&](const LjHitQuery *qd, LjHitResult *rd, hipStream_t s) {
            ljd::with_shade_variant(v, [&](auto ft) {
                hipLaunchKernelGGL(ljd::k_vertex_queries<decltype(ft)>, dim3(grid_for(scene->ctx, n)), dim3(ljd::kQBlock), 0, s, scene->dscene, (const long long *)first.p, qd, rd, (long long)n);
            });
        });
    });
}

int lj_primary_ray_queries(lj_scene *scene, int64_t n, const LjPrimaryQuery *q, LjPrimaryResult *r) {
    return lj::guard([&]() {
        if (!scene) throw LjError(LJ_ERR_INVALID_ARG, "lj_primary_ray_queries: null scene");
        run_queries(scene->ctx, n, q, 1, r, 1, [&](const LjPrimaryQuery *qd, LjPrimaryResult *rd, hipStream_t s) {
            hipLaunchKernelGGL(ljd::k_primary_queries, dim3(grid_for(scene->ctx, n)), dim3(ljd::kQBlock), 0, s, scene->dscene, qd, rd, (long long)n);
        });
    });
}

int lj_filter_queries(lj_context *ctx, int64_t n, const LjFilterQuery *q, float *offsets) {
    return lj::guard([&]() {
        if (!ctx) throw LjError(LJ_ERR_INVALID_ARG, "lj_filter_queries: null context");
        for (int64_t i = 0; i < n && q; i++)
            if (q[i].kind < LJ_FILTER_BOX || q[i].kind > LJ_FILTER_GAUSSIAN) throw LjError(LJ_ERR_INVALID_ARG, "lj_filter_queries: not a Filter alternative");
        run_queries(ctx, n, q, 1, offsets, 2, [&](const LjFilterQuery *qd, float *od, hipStream_t s) {
            hipLaunchKernelGGL(ljd::k_filter_queries, dim3(grid_for(ctx, n)), dim3(ljd::kQBlock), 0, s, qd, od, (long long)n);
        });
    });
}

int lj_pcg32_queries(lj_context *ctx, int64_t n, const uint64_t *streams, uint64_t seed, int32_t count, uint32_t *u32, float *real) {
    return lj::guard([&]() {
        if (!ctx) throw LjError(LJ_ERR_INVALID_ARG, "lj_pcg32_queries: null context");
        if (count <= 0 || count > 4096) throw LjError(LJ_ERR_INVALID_ARG, "lj_pcg32_queries: count must be in [1, 4096]");
        if (n <= 0) return;
        DevBuf rb;
        HIP_CHECK(hipSetDevice(ctx->device));
        if (real) rb.alloc((size_t)n * count * sizeof(float));
        run_queries(ctx, n, streams, 1, u32, (size_t)count, [&](const uint64_t *sd, uint32_t *ud, hipStream_t s) {
            hipLaunchKernelGGL(ljd::k_pcg32_queries, dim3(grid_for(ctx, n)), dim3(ljd::kQBlock), 0, s, sd, seed ? seed : 0x853c49e6748fea9bULL, (int)count, ud, (float *)rb.p, (long long)n);
        });
        if (real) HIP_CHECK(hipMemcpy(real, rb.p, (size_t)n * count * sizeof(float), hipMemcpyDeviceToHost));
    });
}

int lj_texture_queries(lj_scene *scene, int64_t n, const LjTextureQuery *q, float *rgb) {
    return lj::guard([&]() {
        if (!scene) throw LjError(LJ_ERR_INVALID_ARG, "lj_texture_queries: null scene");
        if (n > 0 && !q) throw LjError(LJ_ERR_INVALID_ARG, "lj_texture_queries: null queries");
        std::vector<ljd::DTexQuery> dq((size_t)std::max<int64_t>(n, 0));
        for (int64_t i = 0; i < n; i++) {
            const LjTexture &t = q[i].texture;
            if (t.kind < LJ_TEX_CONSTANT || t.kind > LJ_TEX_CHECKERBOARD) throw LjError(LJ_ERR_INVALID_ARG, "lj_texture_queries: not a Texture alternative");
            if (t.kind == LJ_TEX_IMAGE) {
                const size_t pool = q[i].spectrum ? scene->flat.images3.size() : scene->flat.images1.size();
                if (t.texture_id < 0 || (size_t)t.texture_id >= pool) throw LjError(LJ_ERR_INVALID_ARG, "lj_texture_queries: texture_id outside the image pool");
            }
            ljd::DTexQuery &d = dq[i];
            d.tex.kind = t.kind; d.tex.texture_id = t.texture_id;
            for (int k = 0; k < 3; k++) { d.tex.value[k] = (float)t.value[k]; d.tex.color1[k] = (float)t.color1[k]; }
            d.tex.uscale = (float)t.uscale; d.tex.vscale = (float)t.vscale; d.tex.uoffset = (float)t.uoffset; d.tex.voffset = (float)t.voffset;
            d.u = q[i].uv[0]; d.v = q[i].uv[1]; d.footprint = q[i].footprint; d.spectrum = q[i].spectrum;
        }
        run_queries(scene->ctx, n, dq.data(), 1, rgb, 3, [&](const ljd::DTexQuery *qd, float *od, hipStream_t s) {
            hipLaunchKernelGGL(ljd::k_texture_queries, dim3(grid_for(scene->ctx, n)), dim3(ljd::kQBlock), 0, s, scene->dscene, qd, od, (long long)n);
        });
    });
}

int lj_frame_queries(lj_context *ctx, int64_t n, const LjFrameQuery *q, LjFrameResult *r) {
    return lj::guard([&]() {
        if (!ctx) throw LjError(LJ_ERR_INVALID_ARG, "lj_frame_queries: null context");
        run_queries(ctx, n, q, 1, r, 1, [&](const LjFrameQuery *qd, LjFrameResult *rd, hipStream_t s) {
            hipLaunchKernelGGL(ljd::k_frame_queries, dim3(grid_for(ctx, n)), dim3(ljd::kQBlock), 0, s, qd, rd, (long long)n);
        });
    });
}

} // extern "C"
