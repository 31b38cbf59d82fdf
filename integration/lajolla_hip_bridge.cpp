// lajolla_hip_bridge.cpp — the reference-side binding of liblajolla_hip.so.
//
// This file belongs in the REFERENCE's source tree (src/lajolla_hip_bridge.cpp, added to lajolla_lib in CMakeLists.txt, linked
// with -llajolla_hip); it is kept here so that it can be checked against the reference's headers where they exist:
//     g++ -std=c++17 -fsyntax-only -I<reference>/src -I<reference>/embree/include -I<repo>/include integration/lajolla_hip_bridge.cpp
// (tests/test_integration_bridge.py does exactly that when /root/reference is present).  It uses nothing but the reference's own
// types and the C ABI of include/lajolla_hip.h: a maintainer replaces the body of path_render() (src/render.cpp:71-101) by
// render_hip(scene) and keeps parse_scene(), Scene, Image3 and imwrite() as they are.
#include "lajolla_hip.h"
#include "scene.h"
#include "image.h"
#include "flexception.h"

#include <climits>
#include <type_traits>
#include <vector>

namespace lajolla_hip_bridge {

// ---- Texture<T> (texture.h:76-108): the variant index is LJ_TEX_*
static void put(double *d, const Spectrum &s) { d[0] = s.x; d[1] = s.y; d[2] = s.z; }
static void put(double *d, Real s) { d[0] = d[1] = d[2] = s; }

template <typename T> static LjTexture to_lj(const Texture<T> &t) {
    LjTexture o{};
    o.kind = (int32_t)t.index(); o.texture_id = -1; o.uscale = o.vscale = 1;
    if (auto *c = std::get_if<ConstantTexture<T>>(&t)) put(o.value, c->value);
    else if (auto *i = std::get_if<ImageTexture<T>>(&t)) { o.texture_id = i->texture_id; o.uscale = i->uscale; o.vscale = i->vscale; o.uoffset = i->uoffset; o.voffset = i->voffset; }
    else { const auto &k = std::get<CheckerboardTexture<T>>(t); put(o.value, k.color0); put(o.color1, k.color1); o.uscale = k.uscale; o.vscale = k.vscale; o.uoffset = k.uoffset; o.voffset = k.voffset; }
    return o;
}

// ---- Material (material.h:12-110): the variant index is LJ_MAT_*; slots in the reference's field order
static LjMaterial to_lj(const Material &m) {
    LjMaterial o{};
    o.kind = (int32_t)m.index();
    int n = 0;
    auto slot = [&](const auto &tex) { o.tex[n++] = to_lj(tex); };
    if (auto *p = std::get_if<Lambertian>(&m)) { slot(p->reflectance); }
    else if (auto *p = std::get_if<RoughPlastic>(&m)) { slot(p->diffuse_reflectance); slot(p->specular_reflectance); slot(p->roughness); o.eta = p->eta; }
    else if (auto *p = std::get_if<RoughDielectric>(&m)) { slot(p->specular_reflectance); slot(p->specular_transmittance); slot(p->roughness); o.eta = p->eta; }
    else if (auto *p = std::get_if<DisneyDiffuse>(&m)) { slot(p->base_color); slot(p->roughness); slot(p->subsurface); }
    else if (auto *p = std::get_if<DisneyMetal>(&m)) { slot(p->base_color); slot(p->roughness); slot(p->anisotropic); }
    else if (auto *p = std::get_if<DisneyGlass>(&m)) { slot(p->base_color); slot(p->roughness); slot(p->anisotropic); o.eta = p->eta; }
    else if (auto *p = std::get_if<DisneyClearcoat>(&m)) { slot(p->clearcoat_gloss); }
    else if (auto *p = std::get_if<DisneySheen>(&m)) { slot(p->base_color); slot(p->sheen_tint); }
    else {
        const DisneyBSDF &d = std::get<DisneyBSDF>(m);
        slot(d.base_color); slot(d.specular_transmission); slot(d.metallic); slot(d.subsurface); slot(d.specular); slot(d.roughness);
        slot(d.specular_tint); slot(d.anisotropic); slot(d.sheen); slot(d.sheen_tint); slot(d.clearcoat); slot(d.clearcoat_gloss);
        o.eta = d.eta;
    }
    o.n_tex = n;
    return o;
}

static void put(double *d, const Matrix4x4 &m) { for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) d[4 * i + j] = m(i, j); }   // row-major (matrix.h)

// ---- VolumeSpectrum (volume.h:13-30); the voxel data are narrowed to float (what the .vol files hold, volume.cpp:62-98)
static LjVolume to_lj(const VolumeSpectrum &v, std::vector<std::vector<float>> &keep) {
    LjVolume o{};
    o.kind = (int32_t)v.index();
    if (auto *c = std::get_if<ConstantVolume<Spectrum>>(&v)) { put(o.value, c->value); o.scale = 1; return o; }
    const GridVolume<Spectrum> &g = std::get<GridVolume<Spectrum>>(v);
    o.resolution[0] = g.resolution[0]; o.resolution[1] = g.resolution[1]; o.resolution[2] = g.resolution[2];
    put(o.p_min, g.p_min); put(o.p_max, g.p_max); put(o.max_data, g.max_data); o.scale = g.scale;
    keep.emplace_back(); std::vector<float> &f = keep.back();
    f.reserve(g.data.size() * 3);
    for (const Spectrum &s : g.data) { f.push_back((float)s.x); f.push_back((float)s.y); f.push_back((float)s.z); }
    o.data = f.data();
    return o;
}

// The description with everything it points to (kept alive until the upload returns: lj_scene_upload deep-copies).
struct Description {
    std::vector<LjShape> shapes; std::vector<LjMaterial> mats; std::vector<LjLight> lights; std::vector<LjMedium> media;
    std::vector<LjImage> images3, images1;
    std::vector<double> P, N, UV; std::vector<int32_t> I;
    std::vector<std::vector<float>> keep;   // float copies of voxels and texels
    LjSceneDesc d{};
};

// Scene (scene.h:42-81) -> LjSceneDesc.  Only the constructor-argument view is transferred: bounds, sampling tables and the
// acceleration structure are rebuilt by lj_scene_upload, as Scene::Scene does (scene.cpp:30-52).
inline void describe(const Scene &s, Description &D) {
    for (const Shape &sh : s.shapes) {      // shape.h:26-53: the variant index is LJ_SHAPE_*
        LjShape o{};
        o.kind = (int32_t)sh.index();
        o.material_id = get_material_id(sh); o.area_light_id = get_area_light_id(sh);
        o.interior_medium_id = get_interior_medium_id(sh); o.exterior_medium_id = get_exterior_medium_id(sh);
        if (auto *sp = std::get_if<Sphere>(&sh)) { put(o.position, sp->position); o.radius = sp->radius; }
        else {
            const TriangleMesh &m = std::get<TriangleMesh>(sh);
            o.first_vertex = (int64_t)(D.P.size() / 3); o.n_vertices = (int64_t)m.positions.size();
            o.first_triangle = (int64_t)(D.I.size() / 3); o.n_triangles = (int64_t)m.indices.size();
            o.has_normals = !m.normals.empty(); o.has_uvs = !m.uvs.empty();
            for (size_t v = 0; v < m.positions.size(); v++) {
                D.P.push_back(m.positions[v].x); D.P.push_back(m.positions[v].y); D.P.push_back(m.positions[v].z);
                const Vector3 n = o.has_normals ? m.normals[v] : Vector3{0, 0, 0};
                D.N.push_back(n.x); D.N.push_back(n.y); D.N.push_back(n.z);
                const Vector2 uv = o.has_uvs ? m.uvs[v] : Vector2{0, 0};
                D.UV.push_back(uv.x); D.UV.push_back(uv.y);
            }
            for (const Vector3i &t : m.indices) { D.I.push_back(t[0]); D.I.push_back(t[1]); D.I.push_back(t[2]); }
        }
        D.shapes.push_back(o);
    }
    for (const Material &m : s.materials) D.mats.push_back(to_lj(m));
    for (const Light &l : s.lights) {       // light.h:15-34: the variant index is LJ_LIGHT_*
        LjLight o{};
        o.kind = (int32_t)l.index(); o.shape_id = -1; o.scale = 1;
        if (auto *a = std::get_if<DiffuseAreaLight>(&l)) { o.shape_id = a->shape_id; put(o.intensity, a->intensity); }
        else { const Envmap &e = std::get<Envmap>(l); o.values = to_lj(e.values); put(o.to_world, e.to_world); put(o.to_local, e.to_local); o.scale = e.scale; }
        D.lights.push_back(o);
    }
    for (const Medium &m : s.media) {       // medium.h:10-21: the variant index is LJ_MEDIUM_*; phase_function.h:10-17: LJ_PHASE_*
        LjMedium o{};
        o.kind = (int32_t)m.index();
        const PhaseFunction &ph = std::visit([](const auto &x) -> const PhaseFunction & { return x.phase_function; }, m);
        o.phase_kind = (int32_t)ph.index();
        if (auto *hg = std::get_if<HenyeyGreenstein>(&ph)) o.g = hg->g;
        if (auto *hm = std::get_if<HomogeneousMedium>(&m)) { put(o.sigma_a, hm->sigma_a); put(o.sigma_s, hm->sigma_s); }
        else { const HeterogeneousMedium &het = std::get<HeterogeneousMedium>(m); o.albedo = to_lj(het.albedo, D.keep); o.density = to_lj(het.density, D.keep); }
        D.media.push_back(o);
    }
    // TexturePool (texture.h:13-19): level 0 of every mip chain, narrowed to float (what the image loaders produced, image.cpp:44,96)
    for (const Mipmap3 &mm : s.texture_pool.image3s) {
        const Image3 &img = mm.images[0];
        D.keep.emplace_back(); std::vector<float> &f = D.keep.back();
        for (const Vector3 &p : img.data) { f.push_back((float)p.x); f.push_back((float)p.y); f.push_back((float)p.z); }
        LjImage o{}; o.width = img.width; o.height = img.height; o.channels = 3; o.data = f.data();
        D.images3.push_back(o);
    }
    for (const Mipmap1 &mm : s.texture_pool.image1s) {
        const Image1 &img = mm.images[0];
        D.keep.emplace_back(); std::vector<float> &f = D.keep.back();
        for (Real p : img.data) f.push_back((float)p);
        LjImage o{}; o.width = img.width; o.height = img.height; o.channels = 1; o.data = f.data();
        D.images1.push_back(o);
    }
    LjSceneDesc &d = D.d;
    const Camera &c = s.camera;             // camera.h:10-24; filter.h:45: the variant index is LJ_FILTER_*
    put(d.camera.cam_to_world, c.cam_to_world); put(d.camera.world_to_cam, c.world_to_cam);
    put(d.camera.sample_to_cam, c.sample_to_cam); put(d.camera.cam_to_sample, c.cam_to_sample);
    d.camera.width = c.width; d.camera.height = c.height; d.camera.medium_id = c.medium_id;
    d.camera.filter_kind = (int32_t)c.filter.index();
    d.camera.filter_param = std::visit([](const auto &f) -> Real { if constexpr (std::is_same_v<std::decay_t<decltype(f)>, Gaussian>) return f.stddev; else return f.width; }, c.filter);
    d.options.integrator = (int32_t)s.options.integrator;   // scene.h:14-31: enum order is LJ_INTEGRATOR_*
    d.options.samples_per_pixel = s.options.samples_per_pixel; d.options.max_depth = s.options.max_depth; d.options.rr_depth = s.options.rr_depth;
    d.options.vol_path_version = s.options.vol_path_version; d.options.max_null_collisions = s.options.max_null_collisions;
    d.n_shapes = (int32_t)D.shapes.size(); d.shapes = D.shapes.data();
    d.n_materials = (int32_t)D.mats.size(); d.materials = D.mats.data();
    d.n_lights = (int32_t)D.lights.size(); d.lights = D.lights.data();
    d.n_media = (int32_t)D.media.size(); d.media = D.media.data();
    d.n_images3 = (int32_t)D.images3.size(); d.images3 = D.images3.data();
    d.n_images1 = (int32_t)D.images1.size(); d.images1 = D.images1.data();
    d.envmap_light_id = s.envmap_light_id;
    d.n_vertices = (int64_t)(D.P.size() / 3); d.n_triangles = (int64_t)(D.I.size() / 3);
    d.positions = D.P.data(); d.normals = D.N.data(); d.uvs = D.UV.data(); d.indices = D.I.data();
    d.output_filename = s.output_filename.c_str();
}

static Image3 to_image3(const std::vector<float> &rgb, int w, int hgt) {
    Image3 img(w, hgt);                     // image.h:28-34: data[y * w + x], y = 0 at the top
    for (size_t i = 0; i < (size_t)w * hgt; i++) img(i) = Vector3{rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
    return img;                             // main.cpp:43 imwrite() unchanged
}

// Replaces path_render() / vol_path_render() behind `Image3 render(const Scene&)` (render.cpp:71-170) on one device.
inline Image3 render_hip(const Scene &scene, int device = 0) {
    Description D;
    describe(scene, D);
    lj_context *ctx = nullptr; lj_scene *sc = nullptr;
    if (lj_context_create(device, &ctx) != LJ_OK || lj_scene_upload(ctx, &D.d, &sc) != LJ_OK) Error(lj_last_error());   // flexception.h:8-24
    const int w = scene.camera.width, hgt = scene.camera.height;
    std::vector<float> rgb((size_t)w * hgt * 3);
    LjRenderArgs a{};
    a.max_depth = INT32_MIN;                // spp, max_depth and the seed default to the scene's RenderOptions / pcg.h:33
    const int rc = lj_render(sc, &a, rgb.data());
    lj_scene_destroy(sc); lj_context_destroy(ctx);
    if (rc != LJ_OK) Error(lj_last_error());
    return to_image3(rgb, w, hgt);
}

// The same across the first n_devices GPUs of the node, from this process: the tile loop's parallel_for (render.cpp:78,
// parallel.cpp:183-237) becomes tiles t % N per device and one RCCL reduce of the frames (lajolla_hip.h "device groups").
inline Image3 render_hip_group(const Scene &scene, int n_devices) {
    Description D;
    describe(scene, D);
    lj_device_group *g = nullptr; lj_group_scene *gs = nullptr;
    if (lj_group_create(n_devices, nullptr, &g) != LJ_OK || lj_group_scene_upload(g, &D.d, &gs) != LJ_OK) Error(lj_last_error());
    const int w = scene.camera.width, hgt = scene.camera.height;
    std::vector<float> rgb((size_t)w * hgt * 3);
    LjRenderArgs a{};
    a.max_depth = INT32_MIN;
    const int rc = lj_group_render(gs, &a, rgb.data());
    lj_group_scene_destroy(gs); lj_group_destroy(g);
    if (rc != LJ_OK) Error(lj_last_error());
    return to_image3(rgb, w, hgt);
}

} // namespace lajolla_hip_bridge
