// Scene flattening: the host half of Scene::Scene (scene.cpp:3-53).  All derived quantities are computed in
// double exactly as the reference computes them and narrowed to float once, at the end.
#include "flatten.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace lj {

namespace {

inline V3 v3(const double *p) { return {p[0], p[1], p[2]}; }
inline void st3(float *d, const V3 &v) { d[0] = (float)v.x; d[1] = (float)v.y; d[2] = (float)v.z; }
inline double luminance(const V3 &s) { return s.x * 0.212671 + s.y * 0.715160 + s.z * 0.072169; }

void frisvad(const V3 &n, V3 &a, V3 &b) {  // frame.h:11-22
    if (n.z < -1 + 1e-6) { a = {0, -1, 0}; b = {-1, 0, 0}; }
    else { double k = 1 / (1 + n.z), m = -n.x * n.y * k; a = {1 - n.x * n.x * k, m, -n.x}; b = {m, 1 - n.y * n.y * k, -n.y}; }
}

// make_table_dist_1d (table_dist.cpp:3-25): cdf has n+1 entries, entries 0..n-1 normalised, the last one left = total
void table_1d(const std::vector<double> &f, std::vector<double> &pmf, std::vector<double> &cdf) {
    size_t n = f.size();
    pmf = f; cdf.assign(n + 1, 0.0);
    for (size_t i = 0; i < n; i++) cdf[i + 1] = cdf[i] + pmf[i];
    double total = cdf.back();
    if (total > 0) { for (size_t i = 0; i < n; i++) { pmf[i] /= total; cdf[i] /= total; } }
    else { for (size_t i = 0; i < n; i++) { pmf[i] = 1.0 / (double)n; cdf[i] = (double)i / (double)n; } cdf.back() = 1; }
}

// guide table of sample_cdf_guided (device/dshade.h) for a cdf of n + 1 float entries: for bin b (u * n truncates to b) the index the full
// search returns — the first entry greater than u, or n + 1 — lies between the results for (b - 1) / n and (b + 2) / n: one bin of slack
// on either side covers the rounding of u * n and of b / n in float.  Packed lo | hi << 16, stored as a float bit pattern.
void make_cdf_guide(const float *cdf, int n, std::vector<float> &out) {
    if (n + 1 > 0xffff) throw LjError(LJ_ERR_UNSUPPORTED, "environment map wider or taller than 65534 texels");
    auto upper = [&](float u) { int lo = 0, hi = n + 1; while (lo < hi) { int mid = (lo + hi) >> 1; if (cdf[mid] > u) hi = mid; else lo = mid + 1; } return lo; };
    for (int b = 0; b < n; b++) {
        const uint32_t lo = b >= 1 ? (uint32_t)upper((float)(b - 1) / (float)n) : 0u;
        const uint32_t hi = b + 2 <= n - 1 ? (uint32_t)upper((float)(b + 2) / (float)n) : (uint32_t)(n + 1);
        union { uint32_t u; float f; } c; c.u = lo | (hi << 16);
        out.push_back(c.f);
    }
}

ljd::DTexture conv_tex(const LjTexture &t) {
    ljd::DTexture d{};
    d.kind = t.kind; d.texture_id = t.texture_id;
    for (int i = 0; i < 3; i++) { d.value[i] = (float)t.value[i]; d.color1[i] = (float)t.color1[i]; }
    d.uscale = (float)t.uscale; d.vscale = (float)t.vscale; d.uoffset = (float)t.uoffset; d.voffset = (float)t.voffset;
    return d;
}

// mip chain of mipmap.h:25-48, computed in double from the float level 0, appended to `texels` as float
ljd::DImage build_mips(const LjImage &img, std::vector<float> &texels, std::vector<double> *level0_d) {
    ljd::DImage out{};
    const int ch = img.channels >= 3 ? 3 : 1;
    std::vector<double> cur((size_t)img.width * img.height * ch);
    for (size_t i = 0; i < (size_t)img.width * img.height; i++) for (int c = 0; c < ch; c++) cur[i * ch + c] = img.data[i * img.channels + c];
    if (level0_d) *level0_d = cur;
    int size = std::max(img.width, img.height);
    int num_levels = std::min((int)std::ceil(std::log2((double)size) + 1), 8);
    int w = img.width, h = img.height;
    out.levels = num_levels; out.channels = ch;
    for (int l = 0; l < num_levels; l++) {
        // colour texels are stored as 16-byte RGB0 quads (one dwordx4 gather per texel instead of three dword loads: a trilinear lookup is
        // eight texels), grey ones as single floats; every level starts on a 16-byte boundary
        while (texels.size() % 4) texels.push_back(0.0f);
        out.lv[l].w = w; out.lv[l].h = h; out.lv[l].offset = (int64_t)texels.size();
        if (ch == 3) { for (size_t i = 0; i + 2 < cur.size(); i += 3) { texels.push_back((float)cur[i]); texels.push_back((float)cur[i + 1]); texels.push_back((float)cur[i + 2]); texels.push_back(0.0f); } }
        else for (double v : cur) texels.push_back((float)v);
        if (l == num_levels - 1) break;
        int nw = std::max(w / 2, 1), nh = std::max(h / 2, 1);
        std::vector<double> next((size_t)nw * nh * ch);
        auto at = [&](int x, int y, int c) { size_t idx = ((size_t)y * w + x); idx = std::min(idx, (size_t)w * h - 1); return cur[idx * ch + c]; };
        for (int y = 0; y < nh; y++) for (int x = 0; x < nw; x++) for (int c = 0; c < ch; c++)
            next[((size_t)y * nw + x) * ch + c] = (((at(2 * x, 2 * y, c) + at(2 * x + 1, 2 * y, c)) + at(2 * x, 2 * y + 1, c)) + at(2 * x + 1, 2 * y + 1, c)) * (1.0 / 4.0);
        cur.swap(next); w = nw; h = nh;
    }
    return out;
}

} // namespace

ljd::DScene FlatScene::host_view() const {
    ljd::DScene s{};
    s.cam = cam;
    s.nodes = nodes.data(); s.n_nodes = (int)nodes.size();
    s.nodes8 = nodes8.data(); s.n_nodes8 = (int)nodes8.size(); s.node8_stride = (int32_t)sizeof(ljd::DNode8);
    s.leaf_prims = leaf_prims.data(); s.n_prims = (int)leaf_prims.size();
    s.prims = prims.data(); s.spheres = spheres.data(); s.n_spheres = (int32_t)n_spheres;
    s.materials = materials.data(); s.n_materials = (int)materials.size();
    s.lights = lights.data(); s.n_lights = (int)lights.size();
    s.light_cdf = light_cdf.data(); s.light_tris = light_tris.data(); s.light_tri_cdf = light_tri_cdf.data();
    s.images3 = images3.data(); s.images1 = images1.data(); s.texels = texels.data(); s.env_tables = env_tables.data();
    s.env_marg = env_tables.data() + env_marg_first; s.env_marg_first = env_marg_first; s.env_marg_count = env_marg_count;
    s.n_images3 = (int32_t)images3.size(); s.n_images1 = (int32_t)images1.size();
    s.envmap_light_id = envmap_light_id; s.max_depth = max_depth; s.rr_depth = rr_depth;
    s.eps = (float)shadow_epsilon;
    s.init_spread = 0.25f / (float)std::max(cam.width, cam.height);
    s.media = media.data(); s.n_media = (int)media.size(); s.volume_data = volume_data.data(); s.shape_media = shape_media.data();
    s.cam_medium = cam_medium; s.max_null_collisions = max_null_collisions;
    s.vol_path_version = vol_path_version;
    s.has_heterogeneous_medium = 0;
    for (const auto &m : media) if (m.kind == LJ_MEDIUM_HETEROGENEOUS) s.has_heterogeneous_medium = 1;
    s.scan_leaves = scan_leaves.empty() ? nullptr : scan_leaves.data(); s.n_scan_leaves = (int32_t)scan_leaves.size(); s.n_scan_used = scan_leaves.empty() ? 0 : n_scan_used;
    return s;
}

FlatScene flatten_scene(const LjSceneDesc &d) {
    FlatScene F;
    if (d.options.integrator < LJ_INTEGRATOR_DEPTH || d.options.integrator > LJ_INTEGRATOR_VOLPATH)
        throw LjError(LJ_ERR_UNSUPPORTED, "integrator id " + std::to_string(d.options.integrator) + " is not implemented");
    const bool volumetric = d.options.integrator == LJ_INTEGRATOR_VOLPATH;
    // ---- numbers the kernels index memory with must be finite: a NaN in a camera matrix, a vertex or a light transform turns into NaN
    // directions and from there into texel / table indices on the device (the reference has the same hole; a GPU fault is a worse failure)
    {
        auto finite = [](const double *v, size_t n) { for (size_t i = 0; i < n; i++) if (!std::isfinite(v[i])) return false; return true; };
        if (!finite(d.camera.cam_to_world, 16) || !finite(d.camera.sample_to_cam, 16) || !std::isfinite(d.camera.filter_param))
            throw LjError(LJ_ERR_INVALID_ARG, "camera: matrix or filter parameter is not finite");
        if (d.camera.width <= 0 || d.camera.height <= 0 || (long long)d.camera.width * d.camera.height > (1ll << 28)) throw LjError(LJ_ERR_INVALID_ARG, "camera: bad film size");
        if (d.n_vertices > 0 && (!d.positions || !finite(d.positions, (size_t)d.n_vertices * 3))) throw LjError(LJ_ERR_INVALID_ARG, "a vertex position is not finite");
        for (int i = 0; i < d.n_shapes; i++) {
            const LjShape &sh = d.shapes[i];
            if (sh.kind == LJ_SHAPE_SPHERE && (!finite(sh.position, 3) || !std::isfinite(sh.radius))) throw LjError(LJ_ERR_INVALID_ARG, "sphere " + std::to_string(i) + ": centre or radius is not finite");
            if (sh.kind != LJ_SHAPE_SPHERE && sh.n_vertices > 0) {
                if (sh.first_vertex < 0 || sh.first_vertex + sh.n_vertices > d.n_vertices) throw LjError(LJ_ERR_INVALID_ARG, "shape " + std::to_string(i) + ": vertex range outside the pools");
                if (sh.has_normals && d.normals && !finite(d.normals + 3 * sh.first_vertex, (size_t)sh.n_vertices * 3)) throw LjError(LJ_ERR_INVALID_ARG, "shape " + std::to_string(i) + ": a vertex normal is not finite");
                if (sh.has_uvs && d.uvs && !finite(d.uvs + 2 * sh.first_vertex, (size_t)sh.n_vertices * 2)) throw LjError(LJ_ERR_INVALID_ARG, "shape " + std::to_string(i) + ": a texture coordinate is not finite");
            }
        }
        for (int i = 0; i < d.n_lights; i++)
            if (!finite(d.lights[i].intensity, 3) || !std::isfinite(d.lights[i].scale) || (d.lights[i].kind == LJ_LIGHT_ENVMAP && (!finite(d.lights[i].to_world, 16) || !finite(d.lights[i].to_local, 16))))
                throw LjError(LJ_ERR_INVALID_ARG, "light " + std::to_string(i) + ": intensity, scale or transform is not finite");
    }
    F.cam_medium = d.camera.medium_id; F.max_null_collisions = d.options.max_null_collisions; F.vol_path_version = d.options.vol_path_version;
    // ---- participating media (only the volumetric integrator looks at them)
    auto medium_ok = [&](int id) { return id >= -1 && id < d.n_media; };
    if (!medium_ok(d.camera.medium_id)) throw LjError(LJ_ERR_INVALID_ARG, "camera references a missing medium");
    for (int i = 0; i < d.n_media; i++) {
        const LjMedium &m = d.media[i];
        ljd::DMedium o{};
        if (m.kind != LJ_MEDIUM_HOMOGENEOUS && m.kind != LJ_MEDIUM_HETEROGENEOUS) throw LjError(LJ_ERR_UNSUPPORTED, "medium kind " + std::to_string(m.kind) + " is not implemented");
        if (m.phase_kind != LJ_PHASE_ISOTROPIC && m.phase_kind != LJ_PHASE_HG) throw LjError(LJ_ERR_UNSUPPORTED, "phase function kind " + std::to_string(m.phase_kind) + " is not implemented");
        o.kind = m.kind; o.phase_kind = m.phase_kind; o.g = (float)m.g;
        for (int k = 0; k < 3; k++) { o.sigma_a[k] = (float)m.sigma_a[k]; o.sigma_s[k] = (float)m.sigma_s[k]; }
        auto conv = [&](const LjVolume &v) {
            ljd::DVolume r{};
            r.kind = v.kind; r.scale = (float)v.scale;
            for (int k = 0; k < 3; k++) { r.res[k] = v.resolution[k]; r.value[k] = (float)v.value[k]; r.p_min[k] = (float)v.p_min[k]; r.p_max[k] = (float)v.p_max[k]; r.max_data[k] = (float)v.max_data[k]; }
            if (m.kind == LJ_MEDIUM_HETEROGENEOUS && v.kind == LJ_VOLUME_GRID) {
                if (!v.data || v.resolution[0] <= 0 || v.resolution[1] <= 0 || v.resolution[2] <= 0) throw LjError(LJ_ERR_INVALID_ARG, "grid volume without voxels");
                const size_t nvox = (size_t)v.resolution[0] * v.resolution[1] * v.resolution[2], n = nvox * 3;
                r.offset = (int64_t)F.volume_data.size();
                // a grid whose three channels agree in every voxel (a one-channel .vol file, volume.cpp) is stored as one float per voxel
                bool mono = true;
                for (size_t i = 0; i < nvox && mono; i++) mono = memcmp(&v.data[3 * i], &v.data[3 * i + 1], 4) == 0 && memcmp(&v.data[3 * i], &v.data[3 * i + 2], 4) == 0;
                r.mono = mono ? 1 : 0;
                if (mono) { F.volume_data.reserve(F.volume_data.size() + nvox); for (size_t i = 0; i < nvox; i++) F.volume_data.push_back(v.data[3 * i]); }
                else F.volume_data.insert(F.volume_data.end(), v.data, v.data + n);
            }
            return r;
        };
        o.albedo = conv(m.albedo); o.density = conv(m.density);
        F.media.push_back(o);
    }
    if (F.volume_data.empty()) F.volume_data.push_back(0.0f);
    F.integrator = d.options.integrator; F.spp = d.options.samples_per_pixel; F.max_depth = d.options.max_depth; F.rr_depth = d.options.rr_depth;
    F.envmap_light_id = d.envmap_light_id;
    // ---- camera
    for (int i = 0; i < 16; i++) { F.cam.sample_to_cam[i] = (float)d.camera.sample_to_cam[i]; F.cam.cam_to_world[i] = (float)d.camera.cam_to_world[i]; }
    {   // xform_point(cam_to_world, 0) (camera.cpp:44)
        const double *m = d.camera.cam_to_world; double inv_w = 1.0 / m[15];
        F.cam.org[0] = (float)(m[3] * inv_w); F.cam.org[1] = (float)(m[7] * inv_w); F.cam.org[2] = (float)(m[11] * inv_w);
    }
    F.cam.width = d.camera.width; F.cam.height = d.camera.height; F.cam.filter_kind = d.camera.filter_kind; F.cam.filter_param = (float)d.camera.filter_param;

    // ---- materials
    for (int i = 0; i < d.n_materials; i++) {
        const LjMaterial &m = d.materials[i];
        ljd::DMaterial dm{}; dm.kind = m.kind; dm.n_tex = m.n_tex; dm.eta = (float)m.eta;
        if (m.n_tex < 0 || m.n_tex > LJ_MAX_TEX_SLOTS) throw LjError(LJ_ERR_INVALID_ARG, "material " + std::to_string(i) + ": n_tex out of range");
        // image textures index the TexturePool (texture.h:13-19): slot 0 of every alternative but DisneyClearcoat, and slot 1
        // of RoughPlastic / RoughDielectric, are Texture<Spectrum> (image3s); every other slot is a Texture<Real> (image1s)
        for (int t = 0; t < m.n_tex; t++) {
            if (m.tex[t].kind != LJ_TEX_IMAGE) continue;
            const bool spectrum = (t == 0 && m.kind != LJ_MAT_DISNEYCLEARCOAT) || (t == 1 && (m.kind == LJ_MAT_ROUGHPLASTIC || m.kind == LJ_MAT_ROUGHDIELECTRIC));
            const int n = spectrum ? d.n_images3 : d.n_images1;
            if (m.tex[t].texture_id < 0 || m.tex[t].texture_id >= n)
                throw LjError(LJ_ERR_INVALID_ARG, "material " + std::to_string(i) + " slot " + std::to_string(t) + ": texture_id " + std::to_string(m.tex[t].texture_id) +
                              " outside the " + (spectrum ? "3" : "1") + "-channel image pool (" + std::to_string(n) + " images)");
        }
        for (int t = 0; t < LJ_MAX_TEX_SLOTS; t++) dm.tex[t] = conv_tex(m.tex[t]);
        F.materials.push_back(dm);
    }
    // ---- texture pool
    std::vector<std::vector<double>> level0_d(d.n_images3);
    for (int i = 0; i < d.n_images3; i++) F.images3.push_back(build_mips(d.images3[i], F.texels, &level0_d[i]));
    for (int i = 0; i < d.n_images1; i++) F.images1.push_back(build_mips(d.images1[i], F.texels, nullptr));

    // ---- geometry: global primitive order = shape order, then triangle order; spheres are one primitive
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    std::vector<BuildPrim> bprims;
    std::vector<ljd::DPrim> gprims;  // in global order; reordered into leaf order after the build
    std::vector<double> mesh_area(d.n_shapes, 0.0);
    for (int si = 0; si < d.n_shapes; si++) {
        const LjShape &sh = d.shapes[si];
        if (sh.material_id >= 0) {
            if (sh.material_id >= d.n_materials) throw LjError(LJ_ERR_INVALID_ARG, "shape references a missing material");
            int kind = d.materials[sh.material_id].kind;
            if (kind < LJ_MAT_LAMBERTIAN || kind > LJ_MAT_DISNEYBSDF)
                throw LjError(LJ_ERR_UNSUPPORTED, "material alternative " + std::to_string(kind) + " (material.h:102-110) is not implemented on the device");
        } else if (!volumetric) throw LjError(LJ_ERR_INVALID_ARG, "shape " + std::to_string(si) + " has no material (the reference asserts material_id >= 0, path_tracing.h:165)");
        // (a shape without a material is an index-matched medium boundary for the volumetric integrator, vol_path_tracing.h:702-712)
        if (!medium_ok(sh.interior_medium_id) || !medium_ok(sh.exterior_medium_id)) throw LjError(LJ_ERR_INVALID_ARG, "shape references a missing medium");
        F.shape_media.push_back(sh.interior_medium_id); F.shape_media.push_back(sh.exterior_medium_id);
        if (sh.kind == LJ_SHAPE_SPHERE) {
            ljd::DSphere ds{}; for (int k = 0; k < 3; k++) ds.center[k] = sh.position[k]; ds.radius = sh.radius; ds.gprim = (int32_t)F.prims.size();
            int slot = (int)F.spheres.size(); F.spheres.push_back(ds);
            ljd::DPrimShade ps{}; ps.shape_id = si; ps.prim_id = 0; ps.material_id = sh.material_id; ps.light_id = sh.area_light_id;
            ps.flags = 1; ps.sphere_slot = slot;
            for (int k = 0; k < 3; k++) ps.n0[k] = (float)sh.position[k];
            ps.n1[0] = (float)sh.radius;
            ljd::DPrim p{}; p.gprim = (int)F.prims.size(); p.kind = 1; p.sphere_slot = slot;
            BuildPrim bp{};
            for (int k = 0; k < 3; k++) {  // sphere_bounds_func (sphere.inl:1-10): double arithmetic stored into float bounds
                float l = (float)(sh.position[k] - sh.radius), h = (float)(sh.position[k] + sh.radius);
                lo[k] = std::min(lo[k], l); hi[k] = std::max(hi[k], h);
                float pad = 1e-5f * (std::fabs(l) + std::fabs(h)) + 1e-7f * (h - l) + 1e-30f;
                bp.lo[k] = l - pad; bp.hi[k] = h + pad;
            }
            F.prims.push_back(ps); gprims.push_back(p); bprims.push_back(bp); F.n_spheres++;
            continue;
        }
        const double *P = d.positions + 3 * sh.first_vertex, *N = d.normals + 3 * sh.first_vertex, *UV = d.uvs + 2 * sh.first_vertex;
        const int32_t *I = d.indices + 3 * sh.first_triangle;
        for (int64_t t = 0; t < sh.n_triangles; t++) {
            int i0 = I[3 * t], i1 = I[3 * t + 1], i2 = I[3 * t + 2];
            if (i0 < 0 || i1 < 0 || i2 < 0 || i0 >= sh.n_vertices || i1 >= sh.n_vertices || i2 >= sh.n_vertices)
                throw LjError(LJ_ERR_INVALID_ARG, "triangle index out of range in shape " + std::to_string(si));
            V3 p0 = v3(P + 3 * i0), p1 = v3(P + 3 * i1), p2 = v3(P + 3 * i2);
            ljd::DPrim p{}; p.gprim = (int)F.prims.size(); p.kind = 0; p.sphere_slot = 0;
            st3(p.v0, p0); st3(p.v1, p1); st3(p.v2, p2);  // triangle_mesh.inl:11-14
            BuildPrim bp{};
            bp.tri = 1;
            for (int k = 0; k < 3; k++) { bp.v[0][k] = p.v0[k]; bp.v[1][k] = p.v1[k]; bp.v[2][k] = p.v2[k]; }
            for (int k = 0; k < 3; k++) {
                float l = std::min(p.v0[k], std::min(p.v1[k], p.v2[k])), h = std::max(p.v0[k], std::max(p.v1[k], p.v2[k]));
                lo[k] = std::min(lo[k], l); hi[k] = std::max(hi[k], h);
                float pad = 1e-5f * (std::fabs(l) + std::fabs(h)) + 1e-7f * (h - l) + 1e-30f;
                bp.lo[k] = l - pad; bp.hi[k] = h + pad;
            }
            // Embree-convention geometry normal on the float vertices, in float, as the hit would report it
            float e1[3] = {p.v1[0] - p.v0[0], p.v1[1] - p.v0[1], p.v1[2] - p.v0[2]}, e2[3] = {p.v2[0] - p.v0[0], p.v2[1] - p.v0[1], p.v2[2] - p.v0[2]};
            V3 Ng{(double)(e1[1] * e2[2] - e1[2] * e2[1]), (double)(e1[2] * e2[0] - e1[0] * e2[2]), (double)(e1[0] * e2[1] - e1[1] * e2[0])};
            V3 gn = normalize(Ng);
            // compute_shading_info constants (triangle_mesh.inl:65-127)
            V2 uv0{0, 0}, uv1{1, 0}, uv2{1, 1};
            if (sh.has_uvs) { uv0 = {UV[2 * i0], UV[2 * i0 + 1]}; uv1 = {UV[2 * i1], UV[2 * i1 + 1]}; uv2 = {UV[2 * i2], UV[2 * i2 + 1]}; }
            V2 duvds{uv2.x - uv0.x, uv2.y - uv0.y}, duvdt{uv2.x - uv1.x, uv2.y - uv1.y};
            double det = duvds.x * duvdt.y - duvdt.x * duvds.y;
            V3 dpdu, dpdv;
            if (std::fabs(det) > 1e-8f) {
                double dsdu = duvdt.y / det, dtdu = -duvds.y / det, dsdv = duvdt.x / det, dtdv = -duvds.x / det;
                V3 dpds = p2 - p0, dpdt = p2 - p1;
                dpdu = dpds * dsdu + dpdt * dtdu; dpdv = dpds * dsdv + dpdt * dtdv;
            } else frisvad(gn, dpdu, dpdv);
            ljd::DPrimShade ps{};
            if (sh.has_normals) { st3(ps.n0, v3(N + 3 * i0)); st3(ps.n1, v3(N + 3 * i1)); st3(ps.n2, v3(N + 3 * i2)); }
            ps.uv0[0] = (float)uv0.x; ps.uv0[1] = (float)uv0.y; ps.uv1[0] = (float)uv1.x; ps.uv1[1] = (float)uv1.y; ps.uv2[0] = (float)uv2.x; ps.uv2[1] = (float)uv2.y;
            st3(ps.dpdu, dpdu); st3(ps.gn, gn);
            ps.inv_uv_size = (float)std::max(length(dpdu), length(dpdv));
            ps.shape_id = si; ps.prim_id = (int)t; ps.material_id = sh.material_id; ps.light_id = sh.area_light_id;
            ps.flags = sh.has_normals ? 2 : 0; ps.sphere_slot = -1;
            F.prims.push_back(ps); gprims.push_back(p); bprims.push_back(bp);
            mesh_area[si] += length(cross(p1 - p0, p2 - p0)) / 2;  // init_sampling_dist (triangle_mesh.inl:48-63)
            F.n_triangles++;
        }
    }
    // ---- bounds sphere from the float scene bounds (scene.cpp:30-34), epsilons (scene.h:99-105)
    if (F.prims.empty()) { for (int k = 0; k < 3; k++) lo[k] = hi[k] = 0.0f; }
    V3 lb{lo[0], lo[1], lo[2]}, ub{hi[0], hi[1], hi[2]};
    F.bounds_radius = length(ub - lb) / 2;
    V3 ctr = (lb + ub) / 2.0;
    F.bounds_center[0] = ctr.x; F.bounds_center[1] = ctr.y; F.bounds_center[2] = ctr.z;
    F.shadow_epsilon = std::min(F.bounds_radius * 1e-5, 0.01);

    // ---- BVH (replaces rtcCommitScene)
    std::vector<int> order;
    int max_leaf = 4;
    if (const char *e = getenv("LJ_TUNE_MAX_LEAF")) max_leaf = std::min(4, std::max(1, atoi(e)));
    build_bvh(bprims, max_leaf, 38, F.nodes, F.nodes8, order, F.bvh_depth, F.bvh8_depth);
    F.leaf_prims.resize(order.size());   // (>= gprims.size(): a primitive cut by a spatial split sits in a leaf on either side)
    for (size_t i = 0; i < order.size(); i++) F.leaf_prims[i] = gprims[order[i]];
    // ---- flat leaf table of a tiny scene (device/dscan.h): the leaves of the tree with their (padded) boxes, one 32-byte
    // record each, for the scan-based traversal that tests every leaf box of the scene for every ray (no stack, no node
    // fetches, every lane busy).  Only for scenes of at most 32 leaves and 256 primitives; padded to a multiple of four with
    // boxes that are never entered (a far-away point) and hold no primitive.
    {
        std::vector<ljd::DScanLeaf> leaves;
        for (const auto &nd : F.nodes) for (int k = 0; k < 4; k++) {
            if (nd.child[k] >= 0 || !(nd.lox[k] <= nd.hix[k])) continue;
            ljd::DScanLeaf L{};
            L.lo[0] = nd.lox[k]; L.lo[1] = nd.loy[k]; L.lo[2] = nd.loz[k]; L.hi[0] = nd.hix[k]; L.hi[1] = nd.hiy[k]; L.hi[2] = nd.hiz[k];
            L.first = (~nd.child[k]) >> 3; L.count = ((~nd.child[k]) & 7) + 1;
            leaves.push_back(L);
        }
        if (leaves.size() <= 32 && F.leaf_prims.size() <= 256 && !leaves.empty()) {
            F.n_scan_used = (int)leaves.size();
            while (leaves.size() % 4) { ljd::DScanLeaf L{}; for (int k = 0; k < 3; k++) L.lo[k] = L.hi[k] = 1e18f; L.first = 0; L.count = 0; leaves.push_back(L); }
            F.scan_leaves = leaves;
        }
    }

    // ---- lights
    std::vector<double> power(d.n_lights, 0.0);
    for (int li = 0; li < d.n_lights; li++) {
        const LjLight &l = d.lights[li];
        ljd::DLight dl{};
        dl.kind = l.kind; dl.shape_id = l.shape_id; dl.scale = (float)l.scale;
        for (int k = 0; k < 3; k++) dl.intensity[k] = (float)l.intensity[k];
        if (l.kind == LJ_LIGHT_AREA) {
            if (l.shape_id < 0 || l.shape_id >= d.n_shapes) throw LjError(LJ_ERR_INVALID_ARG, "area light references a missing shape");
            const LjShape &sh = d.shapes[l.shape_id];
            double area;
            if (sh.kind == LJ_SHAPE_SPHERE) {
                dl.is_sphere = 1; for (int k = 0; k < 3; k++) dl.center[k] = (float)sh.position[k]; dl.radius = (float)sh.radius;
                area = 4 * kPi * sh.radius * sh.radius;  // sphere.inl:206-208
            } else {
                dl.is_sphere = 0; dl.tri_first = (int)F.light_tris.size(); dl.tri_count = (int)sh.n_triangles; dl.cdf_first = (int)F.light_tri_cdf.size();
                const double *P = d.positions + 3 * sh.first_vertex; const int32_t *I = d.indices + 3 * sh.first_triangle;
                std::vector<double> areas(sh.n_triangles);
                for (int64_t t = 0; t < sh.n_triangles; t++) {
                    V3 v0 = v3(P + 3 * I[3 * t]), v1 = v3(P + 3 * I[3 * t + 1]), v2 = v3(P + 3 * I[3 * t + 2]);
                    V3 e1 = v1 - v0, e2 = v2 - v0;
                    areas[t] = length(cross(e1, e2)) / 2;
                    ljd::DLightTri T{}; st3(T.v0, v0); st3(T.e1, e1); st3(T.e2, e2); st3(T.n, normalize(cross(e1, e2)));
                    F.light_tris.push_back(T);
                }
                std::vector<double> pmf, cdf; table_1d(areas, pmf, cdf);
                for (double c : cdf) F.light_tri_cdf.push_back((float)c);
                area = mesh_area[l.shape_id];
                dl.total_area = (float)area;
            }
            power[li] = luminance(V3{l.intensity[0], l.intensity[1], l.intensity[2]}) * area * kPi;  // diffuse_area_light.inl:1-3
        } else {
            if (l.values.kind != LJ_TEX_IMAGE || l.values.texture_id < 0 || l.values.texture_id >= d.n_images3)
                throw LjError(LJ_ERR_UNSUPPORTED, "environment maps must be image textures");
            dl.values = conv_tex(l.values);
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { dl.to_world[r * 3 + c] = (float)l.to_world[r * 4 + c]; dl.to_local[r * 3 + c] = (float)l.to_local[r * 4 + c]; }
            // init_sampling_dist (envmap.inl:75-98) + make_table_dist_2d (table_dist.cpp:40-114)
            const LjImage &img = d.images3[l.values.texture_id];
            const std::vector<double> &tex = level0_d[l.values.texture_id];
            const int w = img.width, h = img.height;
            std::vector<double> cdf_rows((size_t)h * (w + 1)), pdf_rows((size_t)h * w), cdf_marg(h + 1), pdf_marg(h);
            for (int y = 0; y < h; y++) {
                double sin_el = std::sin(kPi * ((y + 0.5) / (double)h));
                double *cdf = &cdf_rows[(size_t)y * (w + 1)];
                cdf[0] = 0;
                std::vector<double> frow(w);
                for (int x = 0; x < w; x++) {
                    const double *t3 = &tex[((size_t)y * w + x) * 3];
                    frow[x] = luminance(V3{t3[0], t3[1], t3[2]}) * sin_el;
                    cdf[x + 1] = cdf[x] + frow[x];
                }
                double integral = cdf[w];
                if (integral > 0) { for (int x = 0; x < w; x++) { cdf[x] /= integral; pdf_rows[(size_t)y * w + x] = frow[x] / integral; } }
                else { for (int x = 0; x < w; x++) { pdf_rows[(size_t)y * w + x] = 1.0 / w; cdf[x] = (double)x / w; } cdf[w] = 1; }
            }
            cdf_marg[0] = 0;
            for (int y = 0; y < h; y++) cdf_marg[y + 1] = cdf_marg[y] + cdf_rows[(size_t)y * (w + 1) + w];
            double total = cdf_marg[h];
            if (total > 0) { for (int y = 0; y < h; y++) { pdf_marg[y] = cdf_rows[(size_t)y * (w + 1) + w] / total; cdf_marg[y] /= total; } cdf_marg[h] = 1; }
            else { for (int y = 0; y < h; y++) { pdf_marg[y] = 1.0 / h; cdf_marg[y] = (double)y / h; } cdf_marg[h] = 1; }
            for (int y = 0; y < h; y++) cdf_rows[(size_t)y * (w + 1) + w] = 1;
            dl.env_w = w; dl.env_h = h;
            auto push = [&](const std::vector<double> &v) { int off = (int)F.env_tables.size(); for (double x : v) F.env_tables.push_back((float)x); return off; };
            dl.env_cdf_rows = push(cdf_rows); dl.env_pdf_rows = push(pdf_rows);
            // guide tables over the float tables the device searches
            dl.env_guide_rows = (int)F.env_tables.size();
            for (int y = 0; y < h; y++) { std::vector<float> g; make_cdf_guide(&F.env_tables[dl.env_cdf_rows + (size_t)y * (w + 1)], w, g); F.env_tables.insert(F.env_tables.end(), g.begin(), g.end()); }
            // the marginal tables, contiguous and 16-byte aligned: the shade kernels keep a copy of them in LDS
            while (F.env_tables.size() % 4) F.env_tables.push_back(0.0f);
            F.env_marg_first = (int)F.env_tables.size();
            dl.env_cdf_marg = push(cdf_marg); dl.env_pdf_marg = push(pdf_marg);
            dl.env_guide_marg = (int)F.env_tables.size();
            { std::vector<float> g; make_cdf_guide(&F.env_tables[dl.env_cdf_marg], h, g); F.env_tables.insert(F.env_tables.end(), g.begin(), g.end()); }
            F.env_marg_count = (int)F.env_tables.size() - F.env_marg_first;
            power[li] = kPi * F.bounds_radius * F.bounds_radius * total / ((double)w * h);  // envmap.inl:1-5
        }
        F.lights.push_back(dl);
    }
    if (d.n_lights > 0) {
        table_1d(power, F.light_pmf_d, F.light_cdf_d);  // scene.cpp:47-52
        for (double c : F.light_cdf_d) F.light_cdf.push_back((float)c);
        for (int li = 0; li < d.n_lights; li++) F.lights[li].pmf = (float)F.light_pmf_d[li];
    } else {
        // (the auxiliary integrators never look at a light — the reference's own intersection test builds a Scene without any,
        // src/tests/intersection.cpp:18-26 — but the path tracers index the table unconditionally)
        if (d.options.integrator >= LJ_INTEGRATOR_PATH)
            throw LjError(LJ_ERR_UNSUPPORTED, "scene has no light: the reference would index an empty light table (path_tracing.h:101-102)");
        F.light_cdf.push_back(0.0f);
    }
    F.light_power_d = power;
    if (F.light_tris.empty()) F.light_tris.push_back(ljd::DLightTri{});
    if (F.light_tri_cdf.empty()) F.light_tri_cdf.push_back(0.0f);
    if (F.spheres.empty()) F.spheres.push_back(ljd::DSphere{});
    if (F.texels.empty()) F.texels.push_back(0.0f);
    if (F.env_tables.empty()) F.env_tables.push_back(0.0f);
    if (F.images3.empty()) F.images3.push_back(ljd::DImage{});
    if (F.images1.empty()) F.images1.push_back(ljd::DImage{});
    if (F.materials.empty()) F.materials.push_back(ljd::DMaterial{});
    return F;
}

} // namespace lj
