// TEST INFRASTRUCTURE — golden-vector generator (authoring container only).
//
// This program is compiled against the *unmodified* reference headers where they lie
// (/root/reference/src, never copied) and linked against oracle/_ref/libljref.so, the
// Embree-free subset of the reference built by oracle/ref_build.sh.  It calls the
// reference's own functions on seeded inputs and writes inputs + outputs as JSON
// fixtures under tests/golden/.  Those fixtures pin oracle/lj_oracle.cpp (the CPU
// restatement) and the front end; they travel to the GPU box, the reference does not.
//
// What CANNOT be pinned this way: anything that executes Embree (intersect(), occluded(),
// Scene::Scene, hence path_tracing() end-to-end) — the Embree binary is absent from
// /root/reference (.MISSING_LARGE_BLOBS) and we do not write stand-ins for it.
//
// The only logic in this file that is not a call into the reference is glue that the
// reference performs inside Embree-dependent functions:
//   * parse loop       (parse_scene.cpp:1050-1121, minus the Scene construction)
//   * scene tables     (scene.cpp:30-52: bounds sphere, sampling dists, light table)
//   * PathVertex fill  (intersection.cpp:38-62, given (shape, prim, u, v, t) instead of an Embree hit)
// Each is marked GLUE below.
#include "parse_scene.h"
#include "medium.h"
#include "phase_function.h"
#include "volume.h"
#include "parse_obj.h"
#include "load_serialized.h"
#include "transform.h"
#include "pcg.h"
#include "3rdparty/pugixml.hpp"
#include <cstdio>
#include <map>
#include <string>
#include <vector>
#include <unistd.h>

// ---- declarations of non-static reference functions that have no header (parse_scene.cpp) ----
struct ParsedSampler { int sample_count = 4; };                               // parse_scene.cpp:16-18
enum class TextureType { BITMAP, CHECKERBOARD };                              // parse_scene.cpp:20-23
struct ParsedTexture {                                                        // parse_scene.cpp:25-31
    TextureType type; fs::path filename; Spectrum color0, color1;
    Real uscale = 1, vscale = 1; Real uoffset = 0, voffset = 0;
};
RenderOptions parse_integrator(pugi::xml_node node);
std::tuple<Camera, std::string, ParsedSampler> parse_sensor(
    pugi::xml_node node, std::vector<Medium> &media, std::map<std::string, int> &medium_map);
std::tuple<std::string, Material> parse_bsdf(
    pugi::xml_node node, const std::map<std::string, ParsedTexture> &texture_map, TexturePool &texture_pool);
Shape parse_shape(pugi::xml_node node, std::vector<Material> &materials,
    std::map<std::string, int> &material_map, const std::map<std::string, ParsedTexture> &texture_map,
    TexturePool &texture_pool, std::vector<Medium> &media, std::map<std::string, int> &medium_map,
    std::vector<Light> &lights, const std::vector<Shape> &shapes);
ParsedTexture parse_texture(pugi::xml_node node);
std::tuple<std::string, Medium> parse_medium(pugi::xml_node node);            // parse_scene.cpp:407
Matrix4x4 parse_transform(pugi::xml_node node);
Spectrum parse_color(pugi::xml_node node);

// ---------------------------------------------------------------- tiny JSON writer
struct J {
    FILE *f; std::vector<int> first;
    explicit J(const std::string &path) { f = fopen(path.c_str(), "w"); if (!f) { perror(path.c_str()); exit(1); } first.push_back(1); }
    ~J() { fputc('\n', f); fclose(f); }
    void sep() { if (!first.back()) fputc(',', f); first.back() = 0; }
    void key(const char *k) { sep(); fprintf(f, "\"%s\":", k); first.back() = 1; }
    void obj() { sep(); fputc('{', f); first.push_back(1); }
    void eobj() { first.pop_back(); fputc('}', f); first.back() = 0; }
    void arr() { sep(); fputc('[', f); first.push_back(1); }
    void earr() { first.pop_back(); fputc(']', f); first.back() = 0; }
    void num(double v) {
        sep();
        if (std::isnan(v)) fputs("\"nan\"", f);
        else if (std::isinf(v)) fputs(v > 0 ? "\"inf\"" : "\"-inf\"", f);
        else fprintf(f, "%.17g", v);
    }
    void inum(long long v) { sep(); fprintf(f, "%lld", v); }
    void str(const std::string &s) { sep(); fprintf(f, "\"%s\"", s.c_str()); }
    void kv(const char *k, double v) { key(k); num(v); }
    void ki(const char *k, long long v) { key(k); inum(v); }
    void ks(const char *k, const std::string &s) { key(k); str(s); }
    void v3(const Vector3 &v) { arr(); num(v.x); num(v.y); num(v.z); earr(); }
    void v2(const Vector2 &v) { arr(); num(v.x); num(v.y); earr(); }
    void kv3(const char *k, const Vector3 &v) { key(k); v3(v); }
    void kv2(const char *k, const Vector2 &v) { key(k); v2(v); }
    void mat(const Matrix4x4 &m) { arr(); for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) num(m(i, j)); earr(); }
    void kmat(const char *k, const Matrix4x4 &m) { key(k); mat(m); }
    void vec(const std::vector<Real> &v) { arr(); for (Real x : v) num(x); earr(); }
    void kvec(const char *k, const std::vector<Real> &v) { key(k); vec(v); }
};

static pcg32_state g_rng;
static Real rnd() { return next_pcg32_real<Real>(g_rng); }
static Vector3 rnd_dir() {
    // uniform direction on the sphere (inputs only; any distribution would do)
    Real z = 1 - 2 * rnd(), r = sqrt(fmax(Real(0), 1 - z * z)), p = c_TWOPI * rnd();
    return Vector3{r * cos(p), r * sin(p), z};
}

// ---------------------------------------------------------------- scene via the reference's parsers
struct Parsed {
    RenderOptions options; Camera camera; std::string filename;
    std::vector<Material> materials; std::vector<Shape> shapes; std::vector<Light> lights;
    std::vector<Medium> media; TexturePool texture_pool; int envmap_light_id = -1;
    Scene *scene = nullptr; // table-bearing fake Scene (never destroyed: ~Scene would call Embree)
};

// GLUE: parse_scene.cpp:1032-1121 with the final `return Scene{...}` removed.
static void parse_with_reference(const fs::path &xml, Parsed &P) {
    pugi::xml_document doc;
    if (!doc.load_file(xml.c_str())) { fprintf(stderr, "cannot load %s\n", xml.c_str()); exit(1); }
    fs::path old_path = fs::current_path();
    fs::current_path(xml.parent_path());
    pugi::xml_node node = doc.child("scene");
    std::map<std::string, int> material_map, medium_map;
    std::map<std::string, ParsedTexture> texture_map;
    P.camera = Camera(Matrix4x4::identity(), 45.0, 256, 256, Box{Real(1)}, -1);
    for (auto child : node.children()) {
        std::string name = child.name();
        if (name == "integrator") {
            P.options = parse_integrator(child);
        } else if (name == "sensor") {
            ParsedSampler sampler;
            std::tie(P.camera, P.filename, sampler) = parse_sensor(child, P.media, medium_map);
            P.options.samples_per_pixel = sampler.sample_count;
        } else if (name == "bsdf") {
            std::string material_name; Material m;
            std::tie(material_name, m) = parse_bsdf(child, texture_map, P.texture_pool);
            if (!material_name.empty()) {
                material_map[material_name] = P.materials.size();
                P.materials.push_back(m);
            }
        } else if (name == "shape") {
            Shape s = parse_shape(child, P.materials, material_map, texture_map, P.texture_pool,
                                  P.media, medium_map, P.lights, P.shapes);
            P.shapes.push_back(s);
        } else if (name == "texture") {
            texture_map[child.attribute("id").value()] = parse_texture(child);
        } else if (name == "emitter") {
            std::string type = child.attribute("type").value();
            if (type == "envmap") {
                std::string filename; Real scale = 1; Matrix4x4 to_world = Matrix4x4::identity();
                for (auto gc : child.children()) {
                    std::string n = gc.attribute("name").value();
                    if (n == "filename") filename = gc.attribute("value").value();
                    else if (n == "toWorld") to_world = parse_transform(gc);
                    else if (n == "scale") scale = std::stof(gc.attribute("value").value());
                }
                Texture<Spectrum> t = make_image_spectrum_texture("__envmap_texture__", filename, P.texture_pool, 1, 1);
                P.lights.push_back(Envmap{t, to_world, inverse(to_world), scale});
                P.envmap_light_id = (int)P.lights.size() - 1;
            }
        } else if (name == "medium") {  // parse_scene.cpp:1111-1119
            std::string medium_name; Medium m;
            std::tie(medium_name, m) = parse_medium(child);
            if (!medium_name.empty()) { medium_map[medium_name] = P.media.size(); P.media.push_back(m); }
        }
    }
    fs::current_path(old_path);

    // GLUE: scene.cpp:30-52 — the part of Scene::Scene that does not touch Embree.
    Scene *S = new Scene();
    S->camera = P.camera;
    const_cast<std::vector<Material>&>(S->materials) = P.materials;
    const_cast<std::vector<Shape>&>(S->shapes) = P.shapes;
    const_cast<std::vector<Light>&>(S->lights) = P.lights;
    const_cast<TexturePool&>(S->texture_pool) = P.texture_pool;
    S->envmap_light_id = P.envmap_light_id;
    S->options = P.options;
    // Embree's rtcGetSceneBounds: float AABB over the float-narrowed vertices (triangle_mesh.inl:11-14)
    // and the sphere user bounds (sphere.inl:1-10, double arithmetic stored into float fields).
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (const Shape &sh : S->shapes) {
        if (auto *m = std::get_if<TriangleMesh>(&sh)) {
            for (const Vector3 &p : m->positions) for (int k = 0; k < 3; k++) {
                lo[k] = fminf(lo[k], (float)p[k]); hi[k] = fmaxf(hi[k], (float)p[k]);
            }
        } else if (auto *s = std::get_if<Sphere>(&sh)) {
            for (int k = 0; k < 3; k++) {
                lo[k] = fminf(lo[k], (float)(s->position[k] - s->radius));
                hi[k] = fmaxf(hi[k], (float)(s->position[k] + s->radius));
            }
        }
    }
    Vector3 lb{lo[0], lo[1], lo[2]}, ub{hi[0], hi[1], hi[2]};
    S->bounds = BSphere{distance(ub, lb) / 2, (lb + ub) / Real(2)};
    for (Shape &shape : const_cast<std::vector<Shape>&>(S->shapes)) init_sampling_dist(shape);
    for (Light &light : const_cast<std::vector<Light>&>(S->lights)) init_sampling_dist(light, *S);
    std::vector<Real> power(S->lights.size());
    for (int i = 0; i < (int)S->lights.size(); i++) power[i] = light_power(S->lights[i], *S);
    S->light_dist = make_table_dist_1d(power);
    P.scene = S;
}

template <typename T> static void dump_texture(J &j, const Texture<T> &t);
static void put(J &j, Real v) { j.num(v); }
static void put(J &j, const Vector3 &v) { j.v3(v); }
template <typename T> static void dump_texture(J &j, const Texture<T> &t) {
    j.obj();
    if (auto *c = std::get_if<ConstantTexture<T>>(&t)) { j.ks("kind", "constant"); j.key("value"); put(j, c->value); }
    else if (auto *i = std::get_if<ImageTexture<T>>(&t)) {
        j.ks("kind", "image"); j.ki("texture_id", i->texture_id);
        j.kv("uscale", i->uscale); j.kv("vscale", i->vscale); j.kv("uoffset", i->uoffset); j.kv("voffset", i->voffset);
    } else if (auto *k = std::get_if<CheckerboardTexture<T>>(&t)) {
        j.ks("kind", "checkerboard"); j.key("color0"); put(j, k->color0); j.key("color1"); put(j, k->color1);
        j.kv("uscale", k->uscale); j.kv("vscale", k->vscale); j.kv("uoffset", k->uoffset); j.kv("voffset", k->voffset);
    }
    j.eobj();
}

static void dump_material(J &j, const Material &m) {
    j.obj();
    static const char *names[] = {"lambertian", "roughplastic", "roughdielectric", "disneydiffuse", "disneymetal",
                                  "disneyglass", "disneyclearcoat", "disneysheen", "disneybsdf"};
    j.ks("kind", names[m.index()]);
#define TEX(field) j.key(#field); dump_texture(j, p->field)
    if (auto *p = std::get_if<Lambertian>(&m)) { TEX(reflectance); }
    else if (auto *p = std::get_if<RoughPlastic>(&m)) { TEX(diffuse_reflectance); TEX(specular_reflectance); TEX(roughness); j.kv("eta", p->eta); }
    else if (auto *p = std::get_if<RoughDielectric>(&m)) { TEX(specular_reflectance); TEX(specular_transmittance); TEX(roughness); j.kv("eta", p->eta); }
    else if (auto *p = std::get_if<DisneyDiffuse>(&m)) { TEX(base_color); TEX(roughness); TEX(subsurface); }
    else if (auto *p = std::get_if<DisneyMetal>(&m)) { TEX(base_color); TEX(roughness); TEX(anisotropic); }
    else if (auto *p = std::get_if<DisneyGlass>(&m)) { TEX(base_color); TEX(roughness); TEX(anisotropic); j.kv("eta", p->eta); }
    else if (auto *p = std::get_if<DisneyClearcoat>(&m)) { TEX(clearcoat_gloss); }
    else if (auto *p = std::get_if<DisneySheen>(&m)) { TEX(base_color); TEX(sheen_tint); }
    else if (auto *p = std::get_if<DisneyBSDF>(&m)) {
        TEX(base_color); TEX(specular_transmission); TEX(metallic); TEX(subsurface); TEX(specular); TEX(roughness);
        TEX(specular_tint); TEX(anisotropic); TEX(sheen); TEX(sheen_tint); TEX(clearcoat); TEX(clearcoat_gloss); j.kv("eta", p->eta);
    }
#undef TEX
    j.eobj();
}

// Full dump for small meshes; counts + checksums + strided samples for large ones.
static void dump_shape(J &j, const Shape &sh, bool full) {
    j.obj();
    j.ki("material_id", get_material_id(sh)); j.ki("area_light_id", get_area_light_id(sh));
    if (auto *s = std::get_if<Sphere>(&sh)) {
        j.ks("kind", "sphere"); j.kv3("position", s->position); j.kv("radius", s->radius);
    } else if (auto *m = std::get_if<TriangleMesh>(&sh)) {
        j.ks("kind", "trimesh");
        j.ki("n_positions", m->positions.size()); j.ki("n_indices", m->indices.size());
        j.ki("n_normals", m->normals.size()); j.ki("n_uvs", m->uvs.size());
        j.kv("total_area", m->total_area);
        Vector3 sp{0, 0, 0}, sn{0, 0, 0}; Vector2 su{0, 0}; long long si = 0;
        for (auto &p : m->positions) sp += p;
        for (auto &n : m->normals) sn += n;
        for (auto &u : m->uvs) su = su + u;
        for (size_t t = 0; t < m->indices.size(); t++) si += (long long)(t % 7 + 1) * (m->indices[t][0] + 2LL * m->indices[t][1] + 3LL * m->indices[t][2]);
        j.kv3("sum_positions", sp); j.kv3("sum_normals", sn); j.kv2("sum_uvs", su); j.ki("index_checksum", si);
        size_t stride = full ? 1 : std::max<size_t>(1, m->positions.size() / 16);
        j.ki("vertex_stride", stride);
        j.key("positions"); j.arr(); for (size_t i = 0; i < m->positions.size(); i += stride) j.v3(m->positions[i]); j.earr();
        j.key("normals"); j.arr(); for (size_t i = 0; i < m->normals.size(); i += stride) j.v3(m->normals[i]); j.earr();
        j.key("uvs"); j.arr(); for (size_t i = 0; i < m->uvs.size(); i += stride) j.v2(m->uvs[i]); j.earr();
        size_t tstride = full ? 1 : std::max<size_t>(1, m->indices.size() / 16);
        j.ki("index_stride", tstride);
        j.key("indices"); j.arr();
        for (size_t i = 0; i < m->indices.size(); i += tstride) { j.arr(); j.inum(m->indices[i][0]); j.inum(m->indices[i][1]); j.inum(m->indices[i][2]); j.earr(); }
        j.earr();
        if (full && m->triangle_sampler.cdf.size() <= 64) { j.kvec("tri_pmf", m->triangle_sampler.pmf); j.kvec("tri_cdf", m->triangle_sampler.cdf); }
    }
    j.eobj();
}

// GLUE: intersection.cpp:38-62 with the Embree hit replaced by explicit (shape, prim, u, v, t, Ng).
static PathVertex make_vertex(const Scene &scene, const Ray &ray, const RayDifferential &rd,
                              int shape_id, int prim_id, float t, float u, float v, const Vector3 &Ng) {
    PathVertex vertex;
    vertex.position = ray.org + ray.dir * Real(t);
    vertex.geometry_normal = normalize(Ng);
    vertex.shape_id = shape_id; vertex.primitive_id = prim_id;
    const Shape &shape = scene.shapes[shape_id];
    vertex.material_id = get_material_id(shape);
    vertex.st = Vector2{u, v};
    ShadingInfo si = compute_shading_info(shape, vertex);
    vertex.shading_frame = si.shading_frame; vertex.uv = si.uv; vertex.mean_curvature = si.mean_curvature;
    vertex.ray_radius = transfer(rd, distance(ray.org, vertex.position));
    vertex.uv_screen_size = vertex.ray_radius / si.inv_uv_size;
    if (dot(vertex.geometry_normal, vertex.shading_frame.n) < 0) vertex.geometry_normal = -vertex.geometry_normal;
    return vertex;
}

static void dump_vertex(J &j, const PathVertex &v) {
    j.obj();
    j.kv3("position", v.position); j.kv3("geometry_normal", v.geometry_normal);
    j.kv3("frame_x", v.shading_frame.x); j.kv3("frame_y", v.shading_frame.y); j.kv3("frame_n", v.shading_frame.n);
    j.kv2("st", v.st); j.kv2("uv", v.uv); j.kv("uv_screen_size", v.uv_screen_size);
    j.kv("mean_curvature", v.mean_curvature); j.kv("ray_radius", v.ray_radius);
    j.ki("shape_id", v.shape_id); j.ki("primitive_id", v.primitive_id); j.ki("material_id", v.material_id);
    j.eobj();
}

// ---------------------------------------------------------------- per-scene fixture
static void gen_scene(const std::string &outdir, const std::string &tag, const fs::path &xml, bool full_meshes) {
    Parsed P; parse_with_reference(xml, P);
    const Scene &S = *P.scene;
    J j(outdir + "/scene_" + tag + ".json");
    j.obj();
    j.ks("generator", "oracle/gen_golden.cpp on the reference's own parse_*/Scene-table/light/shape functions");
    j.key("options"); j.obj();
    j.ki("integrator", (int)P.options.integrator); j.ki("samples_per_pixel", P.options.samples_per_pixel);
    j.ki("max_depth", P.options.max_depth); j.ki("rr_depth", P.options.rr_depth); j.eobj();
    j.key("camera"); j.obj();
    j.ki("width", P.camera.width); j.ki("height", P.camera.height);
    j.kmat("cam_to_world", P.camera.cam_to_world); j.kmat("world_to_cam", P.camera.world_to_cam);
    j.kmat("sample_to_cam", P.camera.sample_to_cam); j.kmat("cam_to_sample", P.camera.cam_to_sample);
    if (auto *b = std::get_if<Box>(&P.camera.filter)) { j.ks("filter", "box"); j.kv("filter_param", b->width); }
    if (auto *b = std::get_if<Tent>(&P.camera.filter)) { j.ks("filter", "tent"); j.kv("filter_param", b->width); }
    if (auto *b = std::get_if<Gaussian>(&P.camera.filter)) { j.ks("filter", "gaussian"); j.kv("filter_param", b->stddev); }
    j.eobj();
    j.key("materials"); j.arr(); for (auto &m : S.materials) dump_material(j, m); j.earr();
    j.key("shapes"); j.arr(); for (auto &s : S.shapes) dump_shape(j, s, full_meshes); j.earr();
    j.key("lights"); j.arr();
    for (auto &l : S.lights) {
        j.obj();
        if (auto *a = std::get_if<DiffuseAreaLight>(&l)) { j.ks("kind", "area"); j.ki("shape_id", a->shape_id); j.kv3("intensity", a->intensity); }
        else if (auto *e = std::get_if<Envmap>(&l)) {
            j.ks("kind", "envmap"); j.kmat("to_world", e->to_world); j.kmat("to_local", e->to_local); j.kv("scale", e->scale);
            j.key("values"); dump_texture(j, e->values);
            const TableDist2D &d = e->sampling_dist;
            j.ki("dist_width", d.width); j.ki("dist_height", d.height); j.kv("dist_total", d.total_values);
            j.kvec("pdf_marginals", d.pdf_marginals); j.kvec("cdf_marginals", d.cdf_marginals);
            // one full row + row checksums
            int r = d.height / 3;
            j.ki("row", r);
            j.kvec("cdf_row", std::vector<Real>(d.cdf_rows.begin() + r * (d.width + 1), d.cdf_rows.begin() + (r + 1) * (d.width + 1)));
            j.kvec("pdf_row", std::vector<Real>(d.pdf_rows.begin() + r * d.width, d.pdf_rows.begin() + (r + 1) * d.width));
        }
        j.kv("power", light_power(l, S));
        j.eobj();
    }
    j.earr();
    j.ki("envmap_light_id", S.envmap_light_id);
    j.kvec("light_pmf", S.light_dist.pmf); j.kvec("light_cdf", S.light_dist.cdf);
    j.kv("bounds_radius", S.bounds.radius); j.kv3("bounds_center", S.bounds.center);
    j.kv("shadow_epsilon", get_shadow_epsilon(S));
    // texture pool: dims + mip checksums + a few texels (images themselves do not travel as goldens)
    j.key("image3s"); j.arr();
    for (auto &mm : S.texture_pool.image3s) {
        j.obj(); j.ki("levels", mm.images.size());
        j.key("dims"); j.arr(); for (auto &im : mm.images) { j.arr(); j.inum(im.width); j.inum(im.height); j.earr(); } j.earr();
        j.key("level_sums"); j.arr(); for (auto &im : mm.images) { Vector3 s{0, 0, 0}; for (auto &p : im.data) s += p; j.v3(s); } j.earr();
        j.key("texels0"); j.arr(); { auto &im = mm.images[0]; size_t st = std::max<size_t>(1, im.data.size() / 32); for (size_t i = 0; i < im.data.size(); i += st) j.v3(im.data[i]); } j.earr();
        j.eobj();
    }
    j.earr();

    // ---- camera rays: sample_primary (camera.cpp:23-47)
    g_rng = init_pcg32(101);
    j.key("primary"); j.arr();
    for (int i = 0; i < 24; i++) {
        Vector2 sp{rnd(), rnd()};
        if (i == 0) sp = Vector2{0, 0};
        if (i == 1) sp = Vector2{0.5, 0.5};
        Ray r = sample_primary(P.camera, sp);
        j.obj(); j.kv2("screen_pos", sp); j.kv3("org", r.org); j.kv3("dir", r.dir); j.eobj();
    }
    j.earr();

    // ---- light selection (scene.cpp:73-79)
    j.key("sample_light"); j.arr();
    for (int i = 0; i < 24; i++) { Real u = (i == 0) ? 0 : (i == 1 ? Real(0.999999999) : rnd()); j.obj(); j.kv("u", u); j.ki("id", sample_light(S, u)); j.eobj(); }
    j.earr();

    // ---- light sampling (light.cpp:52-81): sample_point_on_light / pdf_point_on_light / emission
    j.key("light_samples"); j.arr();
    for (int i = 0; i < 96 && !S.lights.empty(); i++) {
        int lid = i % (int)S.lights.size();
        const Light &L = S.lights[lid];
        Vector3 ref = S.bounds.center + S.bounds.radius * Real(0.6) * Vector3{2 * rnd() - 1, 2 * rnd() - 1, 2 * rnd() - 1};
        if (auto *a = std::get_if<DiffuseAreaLight>(&L)) {
            if (auto *sp = std::get_if<Sphere>(&S.shapes[a->shape_id])) {
                if (i % 5 == 4) ref = sp->position + Real(0.5) * sp->radius * rnd_dir();      // inside
                else if (i % 5 == 3) ref = sp->position + Real(1.5) * sp->radius * rnd_dir(); // close
            }
        }
        Vector2 uv{rnd(), rnd()}; Real w = rnd();
        PointAndNormal pn = sample_point_on_light(L, ref, uv, w, S);
        Real pdf = pdf_point_on_light(L, pn, ref, S);
        Vector3 view = is_envmap(L) ? pn.normal : normalize(ref - pn.position);
        Real fp = (i % 3 == 0) ? Real(0) : Real(0.01) * rnd();
        Spectrum Le = emission(L, view, fp, pn, S);
        j.obj(); j.ki("light_id", lid); j.kv3("ref", ref); j.kv2("uv", uv); j.kv("w", w);
        j.kv3("position", pn.position); j.kv3("normal", pn.normal); j.kv("pdf", pdf);
        j.kv3("view_dir", view); j.kv("footprint", fp); j.kv3("emission", Le); j.eobj();
    }
    j.earr();

    // ---- shading info + BSDF at surface points (shape.cpp:77; material.cpp:90-119)
    // A vertex is synthesised on a random primitive with random barycentrics / sphere point; the ray comes from a
    // random origin.  Ng follows Embree's convention (v1-v0)x(v2-v0) (triangles) / p - c (sphere.inl:87-91).
    j.key("vertices"); j.arr();
    for (int i = 0; i < 64; i++) {
        int sid = (int)(rnd() * S.shapes.size()) % (int)S.shapes.size();
        const Shape &sh = S.shapes[sid];
        Vector3 hit, Ng; int prim = 0; float u, v;
        if (auto *m = std::get_if<TriangleMesh>(&sh)) {
            prim = (int)(rnd() * m->indices.size()) % (int)m->indices.size();
            Real a = sqrt(rnd()), b = rnd(); Real b1 = (1 - a), b2 = a * b; // weights of v1, v2 -> (u, v)
            u = (float)b1; v = (float)b2;
            Vector3i id = m->indices[prim];
            Vector3 p0 = m->positions[id[0]], p1 = m->positions[id[1]], p2 = m->positions[id[2]];
            hit = (1 - Real(u) - Real(v)) * p0 + Real(u) * p1 + Real(v) * p2;
            Ng = cross(p1 - p0, p2 - p0);
        } else {
            auto *s = std::get_if<Sphere>(&sh);
            Vector3 d = rnd_dir(); hit = s->position + s->radius * d; Ng = hit - s->position;
            Vector3 c = Ng / s->radius;
            u = (float)(atan2(c.z, c.x) / c_TWOPI); v = (float)(acos(std::clamp(c.y, Real(-1), Real(1))) / c_PI); // sphere.inl:92-95
        }
        Vector3 org = S.bounds.center + S.bounds.radius * Real(0.5) * rnd_dir();
        Vector3 dir = normalize(hit - org);
        float t = (float)distance(hit, org);
        Ray ray{org, dir, Real(0), infinity<Real>()};
        RayDifferential rd = (i % 2) ? init_ray_differential(P.camera.width, P.camera.height) : RayDifferential{};
        PathVertex vx = make_vertex(S, ray, rd, sid, prim, t, u, v, Ng);
        j.obj();
        j.kv3("ray_org", org); j.kv3("ray_dir", dir); j.kv("t", t); j.kv("u", u); j.kv("v", v); j.kv3("Ng", Ng);
        j.kv("rd_radius", rd.radius); j.kv("rd_spread", rd.spread);
        j.key("vertex"); dump_vertex(j, vx);
        if (vx.material_id >= 0) {
            const Material &mat = S.materials[vx.material_id];
            Vector3 dir_in = -dir;
            j.key("bsdf"); j.arr();
            for (int k = 0; k < 4; k++) {
                Vector3 dir_out = rnd_dir();
                if (k < 2 && dot(dir_out, vx.geometry_normal) * dot(dir_in, vx.geometry_normal) < 0) dir_out = -dir_out;
                Vector2 ruv{rnd(), rnd()}; Real rw = rnd();
                Spectrum f = eval(mat, dir_in, dir_out, vx, S.texture_pool);
                Real pdf = pdf_sample_bsdf(mat, dir_in, dir_out, vx, S.texture_pool);
                auto rec = sample_bsdf(mat, dir_in, vx, S.texture_pool, ruv, rw);
                j.obj(); j.kv3("dir_in", dir_in); j.kv3("dir_out", dir_out); j.kv3("eval", f); j.kv("pdf", pdf);
                j.kv2("rnd_uv", ruv); j.kv("rnd_w", rw); j.ki("sample_valid", rec ? 1 : 0);
                if (rec) {
                    j.kv3("sample_dir", rec->dir_out); j.kv("sample_eta", rec->eta); j.kv("sample_roughness", rec->roughness);
                    j.kv3("sample_eval", eval(mat, dir_in, rec->dir_out, vx, S.texture_pool));
                    j.kv("sample_pdf", pdf_sample_bsdf(mat, dir_in, rec->dir_out, vx, S.texture_pool));
                }
                j.eobj();
            }
            j.earr();
        }
        if (is_light(sh)) { j.kv3("emission", emission(vx, -dir, S)); }
        j.eobj();
    }
    j.earr();
    j.eobj();
}

// ---------------------------------------------------------------- scene-independent KATs
static void gen_core(const std::string &outdir) {
    J j(outdir + "/core.json");
    j.obj();
    j.ks("generator", "oracle/gen_golden.cpp calling pcg.h, filter.cpp, frame.h, table_dist.cpp, microfacet.h, ray.h, spectrum.h");
    // pcg32 (pcg.h:22-68)
    j.key("pcg32"); j.arr();
    for (uint64_t stream : {0ULL, 1ULL, 2ULL, 1023ULL, 67108863ULL, 12345678901ULL}) {
        pcg32_state s = init_pcg32(stream);
        j.obj(); j.ks("stream", std::to_string(stream));
        j.ks("state0", std::to_string(s.state)); j.ks("inc", std::to_string(s.inc));
        j.key("u32"); j.arr(); for (int i = 0; i < 16; i++) j.inum(next_pcg32(s)); j.earr();
        j.key("f64"); j.arr(); for (int i = 0; i < 8; i++) j.num(next_pcg32_real<double>(s)); j.earr();
        j.key("f32"); j.arr(); for (int i = 0; i < 8; i++) j.num(next_pcg32_real<float>(s)); j.earr();
        j.eobj();
    }
    j.earr();
    // filters (filter.cpp:16; filters/*.inl)
    g_rng = init_pcg32(7);
    j.key("filters"); j.arr();
    for (int i = 0; i < 36; i++) {
        Vector2 r{rnd(), rnd()};
        if (i < 3) r = Vector2{0.3, 0.4};              // the reference's own test point (tests/filter.cpp)
        if (i >= 3 && i < 6) r = Vector2{0.0, 0.25};     // exercises the 1e-8 clamp of gaussian.inl:4
        Filter f; const char *kind; Real param;
        switch (i % 3) {
            case 0: param = 1 + (i / 3) * Real(0.5); f = Box{param}; kind = "box"; break;
            case 1: param = 2 + (i / 3) * Real(0.25); f = Tent{param}; kind = "tent"; break;
            default: param = Real(0.5) + (i / 3) * Real(0.125); f = Gaussian{param}; kind = "gaussian"; break;
        }
        Vector2 o = sample(f, r);
        j.obj(); j.ks("kind", kind); j.kv("param", param); j.kv2("rnd", r); j.kv2("out", o); j.eobj();
    }
    j.earr();
    // frames (frame.h:11-57)
    j.key("frames"); j.arr();
    for (int i = 0; i < 12; i++) {
        Vector3 n = rnd_dir();
        if (i == 0) n = Vector3{0.0, 0.0, -1.0};
        if (i == 1) n = normalize(Vector3{0.3, 0.4, 0.5}); // tests/frame.cpp
        Frame fr(n); Vector3 v = rnd_dir();
        j.obj(); j.kv3("n", n); j.kv3("x", fr.x); j.kv3("y", fr.y); j.kv3("v", v);
        j.kv3("to_local", to_local(fr, v)); j.kv3("to_world", to_world(fr, v)); j.eobj();
    }
    j.earr();
    // table dists (table_dist.cpp)
    {
        std::vector<Real> f1{1, 2, 3, 0.5, 4, 0.25};
        TableDist1D d = make_table_dist_1d(f1);
        j.key("table1d"); j.obj(); j.kvec("f", f1); j.kvec("pmf", d.pmf); j.kvec("cdf", d.cdf);
        j.key("samples"); j.arr();
        for (int i = 0; i < 24; i++) { Real u = i == 0 ? 0 : (i == 1 ? 1 : (i == 2 ? d.cdf[2] : rnd())); j.obj(); j.kv("u", u); j.ki("id", sample(d, u)); j.eobj(); }
        j.earr(); j.eobj();
        int w = 5, h = 4; std::vector<Real> f2(w * h);
        for (int i = 0; i < w * h; i++) f2[i] = (i % 7 == 3) ? 0 : rnd() * 3;
        for (int x = 0; x < w; x++) f2[2 * w + x] = 0; // an all-zero row (uniform fallback branch)
        TableDist2D d2 = make_table_dist_2d(f2, w, h);
        j.key("table2d"); j.obj(); j.ki("width", w); j.ki("height", h); j.kvec("f", f2);
        j.kvec("cdf_rows", d2.cdf_rows); j.kvec("pdf_rows", d2.pdf_rows);
        j.kvec("cdf_marginals", d2.cdf_marginals); j.kvec("pdf_marginals", d2.pdf_marginals); j.kv("total_values", d2.total_values);
        j.key("samples"); j.arr();
        for (int i = 0; i < 32; i++) {
            Vector2 r{rnd(), rnd()}; if (i == 0) r = Vector2{0, 0}; if (i == 1) r = Vector2{0.999999, 0.999999};
            Vector2 xy = sample(d2, r);
            j.obj(); j.kv2("rnd", r); j.kv2("xy", xy); j.kv("pdf", pdf(d2, xy)); j.eobj();
        }
        j.earr(); j.eobj();
    }
    // ray differentials (ray.h:35-66)
    j.key("raydiff"); j.arr();
    for (int i = 0; i < 12; i++) {
        RayDifferential rd{rnd() * Real(0.01), rnd() * Real(0.01)};
        Real dist = rnd() * 100, curv = (rnd() - Real(0.5)) * 4, rough = rnd(), eta = Real(0.5) + rnd() * 1.5;
        j.obj(); j.kv("radius", rd.radius); j.kv("spread", rd.spread); j.kv("dist", dist); j.kv("curv", curv);
        j.kv("rough", rough); j.kv("eta", eta);
        j.kv("transfer", transfer(rd, dist)); j.kv("reflect", reflect(rd, curv, rough)); j.kv("refract", refract(rd, curv, eta, rough)); j.eobj();
    }
    j.earr();
    // spectra (spectrum.h:68-125)
    j.key("spectra"); j.arr();
    {
        std::vector<std::vector<std::pair<Real, Real>>> specs = {
            {{400, 0}, {500, 8}, {600, 15.6}, {700, 18.4}},
            {{400, 0.78}, {500, 0.78}, {600, 0.78}, {700, 0.78}},
            {{450, 1}, {550, 0.5}},
            {{380, 0.1}, {420, 0.9}, {560, 0.3}, {780, 0.7}}};
        for (auto &s : specs) {
            Vector3 xyz = integrate_XYZ(s);
            j.obj(); j.key("data"); j.arr(); for (auto &p : s) { j.arr(); j.num(p.first); j.num(p.second); j.earr(); } j.earr();
            j.kv3("xyz", xyz); j.kv3("rgb", XYZ_to_RGB(xyz)); j.eobj();
        }
    }
    j.earr();
    j.key("srgb"); j.arr();
    for (int i = 0; i < 6; i++) { Vector3 c{rnd(), rnd() * Real(0.05), rnd()}; j.obj(); j.kv3("in", c); j.kv3("out", sRGB_to_RGB(c)); j.eobj(); }
    j.earr();
    // transforms (transform.cpp; matrix.h inverse)
    j.key("transforms"); j.arr();
    for (int i = 0; i < 6; i++) {
        Vector3 a{rnd() * 4 - 2, rnd() * 4 - 2, rnd() * 4 - 2}, b = rnd_dir(), up{0, 1, 0};
        Real ang = rnd() * 360;
        Matrix4x4 m = translate(a) * rotate(ang, b) * scale(Vector3{1 + rnd(), 1 + rnd(), 1 + rnd()});
        Matrix4x4 la = look_at(a, a + b, up);
        Vector3 p{rnd(), rnd(), rnd()};
        j.obj(); j.kv3("a", a); j.kv3("b", b); j.kv("angle", ang);
        j.kmat("translate", translate(a)); j.kmat("rotate", rotate(ang, b)); j.kmat("look_at", la);
        j.kmat("m", m); j.kmat("inv_m", inverse(m)); j.kmat("perspective", perspective(30 + ang / 8));
        j.kv3("p", p); j.kv3("xform_point", xform_point(m, p)); j.kv3("xform_vector", xform_vector(m, p));
        j.kv3("xform_normal", xform_normal(inverse(m), p)); j.eobj();
    }
    j.earr();
    j.eobj();
}

// ---------------------------------------------------------------- BSDF KATs for all 9 material alternatives
static Texture<Spectrum> rnd_spec_tex(int i) {
    if (i % 5 == 4) return make_checkerboard_spectrum_texture(Vector3{rnd(), rnd(), rnd()}, Vector3{rnd(), rnd(), rnd()}, 4, 3, 0.1, 0.2);
    return make_constant_spectrum_texture(Vector3{rnd(), rnd(), rnd()});
}
static Texture<Real> rnd_f(Real lo = 0, Real hi = 1) { return make_constant_float_texture(lo + (hi - lo) * rnd()); }

static void gen_materials(const std::string &outdir) {
    J j(outdir + "/materials.json");
    j.obj();
    j.ks("generator", "oracle/gen_golden.cpp calling eval / pdf_sample_bsdf / sample_bsdf (material.cpp:90-119) with TexturePool()");
    j.key("cases"); j.arr();
    g_rng = init_pcg32(2024);
    TexturePool pool;
    for (int kind = 0; kind < 9; kind++) {
        for (int i = 0; i < 40; i++) {
            Material m;
            switch (kind) {
                case 0: m = Lambertian{rnd_spec_tex(i)}; break;
                case 1: m = RoughPlastic{rnd_spec_tex(i), rnd_spec_tex(0), rnd_f(), Real(1.1) + rnd()}; break;
                case 2: m = RoughDielectric{rnd_spec_tex(0), rnd_spec_tex(0), rnd_f(), Real(1.1) + rnd()}; break;
                case 3: m = DisneyDiffuse{rnd_spec_tex(i), rnd_f(), rnd_f()}; break;
                case 4: m = DisneyMetal{rnd_spec_tex(i), rnd_f(), rnd_f()}; break;
                case 5: m = DisneyGlass{rnd_spec_tex(i), rnd_f(), rnd_f(), Real(1.1) + rnd()}; break;
                case 6: m = DisneyClearcoat{rnd_f()}; break;
                case 7: m = DisneySheen{rnd_spec_tex(i), rnd_f()}; break;
                default: m = DisneyBSDF{rnd_spec_tex(i), rnd_f(), rnd_f(), rnd_f(), rnd_f(), rnd_f(), rnd_f(),
                                        rnd_f(), rnd_f(), rnd_f(), rnd_f(), rnd_f(), Real(1.1) + rnd()}; break;
            }
            if (i == 0 && kind == 1) m = RoughPlastic{make_constant_spectrum_texture(Vector3{.5, .5, .5}), make_constant_spectrum_texture(Vector3{1, 1, 1}), make_constant_float_texture(0.1), Real(1.5)}; // tests/materials.cpp
            PathVertex vx;
            Vector3 n = rnd_dir();
            vx.geometry_normal = n;
            // shading normal = perturbed geometry normal (same hemisphere), frame from coordinate_system
            Vector3 sn = normalize(n + Real(0.3) * rnd_dir());
            if (i % 4 == 0) sn = n;
            vx.shading_frame = Frame(sn);
            if (i == 0) { vx.geometry_normal = Vector3{0, 0, 1}; vx.shading_frame = Frame(Vector3{1, 0, 0}, Vector3{0, 1, 0}, Vector3{0, 0, 1}); } // tests/materials.cpp
            vx.uv = Vector2{rnd() * 3 - 1, rnd() * 3 - 1}; vx.uv_screen_size = rnd() * Real(0.01);
            vx.position = Vector3{0, 0, 0}; vx.st = Vector2{0, 0}; vx.mean_curvature = 0; vx.ray_radius = 0;
            Vector3 dir_in = rnd_dir();
            if (i == 0) dir_in = normalize(Vector3{0.3, 0.4, 0.5});
            // mostly on the upper side; every 8th case comes from below (inside / back-face branches)
            if ((i % 8 != 7) && dot(dir_in, vx.geometry_normal) < 0) dir_in = -dir_in;
            if ((i % 8 == 7) && dot(dir_in, vx.geometry_normal) > 0) dir_in = -dir_in;
            j.obj();
            j.key("material"); dump_material(j, m);
            j.kv3("geometry_normal", vx.geometry_normal);
            j.kv3("frame_x", vx.shading_frame.x); j.kv3("frame_y", vx.shading_frame.y); j.kv3("frame_n", vx.shading_frame.n);
            j.kv2("uv", vx.uv); j.kv("uv_screen_size", vx.uv_screen_size);
            j.kv3("dir_in", dir_in);
            j.key("queries"); j.arr();
            for (int k = 0; k < 6; k++) {
                Vector3 dir_out = rnd_dir();
                // k<3: reflection side; k>=3: transmission side w.r.t. dir_in
                Real s = dot(dir_out, vx.geometry_normal) * dot(dir_in, vx.geometry_normal);
                if ((k < 3 && s < 0) || (k >= 3 && s > 0)) dir_out = -dir_out;
                Vector2 ruv{rnd(), rnd()}; Real rw = rnd();
                if (i == 0 && k == 0) { ruv = Vector2{0.3, 0.4}; rw = 0.6; }
                for (int td = 0; td < 2; td++) {
                    TransportDirection dir = td == 0 ? TransportDirection::TO_LIGHT : TransportDirection::TO_VIEW;
                    if (td == 1 && kind != 2) continue; // only RoughDielectric reads it (roughdielectric.inl:64)
                    Spectrum f = eval(m, dir_in, dir_out, vx, pool, dir);
                    Real pdf = pdf_sample_bsdf(m, dir_in, dir_out, vx, pool, dir);
                    auto rec = sample_bsdf(m, dir_in, vx, pool, ruv, rw, dir);
                    j.obj(); j.ki("to_view", td); j.kv3("dir_out", dir_out); j.kv3("eval", f); j.kv("pdf", pdf);
                    j.kv2("rnd_uv", ruv); j.kv("rnd_w", rw); j.ki("sample_valid", rec ? 1 : 0);
                    if (rec) { j.kv3("sample_dir", rec->dir_out); j.kv("sample_eta", rec->eta); j.kv("sample_roughness", rec->roughness); }
                    j.eobj();
                }
            }
            j.earr();
            j.eobj();
        }
    }
    j.earr();
    j.eobj();
}

// ---------------------------------------------------------------- participating media
static void dump_volume(J &j, const VolumeSpectrum &v) {
    j.obj();
    if (auto *c = std::get_if<ConstantVolume<Spectrum>>(&v)) { j.ks("kind", "constant"); j.kv3("value", c->value); }
    else {
        const auto &g = std::get<GridVolume<Spectrum>>(v);
        j.ks("kind", "grid");
        j.key("resolution"); j.arr(); j.inum(g.resolution.x); j.inum(g.resolution.y); j.inum(g.resolution.z); j.earr();
        j.kv3("p_min", g.p_min); j.kv3("p_max", g.p_max); j.kv3("max_data", g.max_data); j.kv("scale", g.scale);
        Vector3 sum{0, 0, 0}; for (auto &d : g.data) sum += d;
        j.kv3("data_sum", sum);
        j.key("samples"); j.arr();
        const size_t st = std::max<size_t>(1, g.data.size() / 17);
        for (size_t i = 0; i < g.data.size(); i += st) j.v3(g.data[i]);
        j.earr();
    }
    j.eobj();
}
static void gen_media(const std::string &outdir, const fs::path &ref) {
    J j(outdir + "/media.json");
    j.obj();
    j.ks("generator", "oracle/gen_golden.cpp on the reference's parse_medium / get_majorant / get_sigma_* / phase function code");
    g_rng = init_pcg32(11);
    // phase functions (phase_functions/*.inl)
    j.key("phase"); j.arr();
    const Real gs[] = {Real(-1), Real(-0.7), Real(0.0005), Real(0.3), Real(0.9)};   // -1: isotropic
    for (Real g : gs) {
        PhaseFunction pf = g == Real(-1) ? PhaseFunction{IsotropicPhase{}} : PhaseFunction{HenyeyGreenstein{g}};
        j.obj(); j.kv("g", g); j.ki("isotropic", g == Real(-1) ? 1 : 0);
        j.key("cases"); j.arr();
        for (int i = 0; i < 24; i++) {
            Vector3 din = rnd_dir(), dout = rnd_dir(); Vector2 uv{rnd(), rnd()};
            j.obj(); j.kv3("dir_in", din); j.kv3("dir_out", dout); j.kv2("uv", uv);
            j.kv3("eval", eval(pf, din, dout)); j.kv("pdf", pdf_sample_phase(pf, din, dout));
            auto smp = sample_phase_function(pf, din, uv);
            j.kv3("sample", *smp);
            j.eobj();
        }
        j.earr(); j.eobj();
    }
    j.earr();
    // scenes: media as parsed + known answers of the medium queries
    j.key("scenes"); j.obj();
    for (const char *name : {"hetvol", "hetvol_colored", "vol_cbox_teapot", "volpath_test6", "volpath_test5", "volpath_test1"}) {
        Parsed P; parse_with_reference(ref / "scenes/volpath_test" / (std::string(name) + ".xml"), P);
        j.key(name); j.obj();
        j.ki("camera_medium_id", P.camera.medium_id);
        j.ki("integrator", (int)P.options.integrator); j.ki("vol_path_version", P.options.vol_path_version);
        j.ki("max_null_collisions", P.options.max_null_collisions); j.ki("max_depth", P.options.max_depth); j.ki("rr_depth", P.options.rr_depth);
        j.key("shape_media"); j.arr();
        for (auto &sh : P.shapes) { j.arr(); j.inum(get_material_id(sh)); j.inum(get_interior_medium_id(sh)); j.inum(get_exterior_medium_id(sh)); j.earr(); }
        j.earr();
        j.key("media"); j.arr();
        for (auto &m : P.media) {
            j.obj();
            PhaseFunction pf = get_phase_function(m);
            if (auto *hg = std::get_if<HenyeyGreenstein>(&pf)) { j.ks("phase", "hg"); j.kv("g", hg->g); } else { j.ks("phase", "isotropic"); j.kv("g", 0); }
            Vector3 lo{-1, -1, -1}, hi{1, 1, 1};
            if (auto *h = std::get_if<HomogeneousMedium>(&m)) { j.ks("kind", "homogeneous"); j.kv3("sigma_a", h->sigma_a); j.kv3("sigma_s", h->sigma_s); }
            else {
                const auto &h2 = std::get<HeterogeneousMedium>(m);
                j.ks("kind", "heterogeneous");
                j.key("albedo"); dump_volume(j, h2.albedo); j.key("density"); dump_volume(j, h2.density);
                if (auto *g = std::get_if<GridVolume<Spectrum>>(&h2.density)) { lo = g->p_min; hi = g->p_max; }
            }
            // queries: points in and a little around the volume's box; rays towards and past it
            j.key("points"); j.arr();
            for (int i = 0; i < 40; i++) {
                Vector3 p{lo.x + (hi.x - lo.x) * (rnd() * Real(1.2) - Real(0.1)), lo.y + (hi.y - lo.y) * (rnd() * Real(1.2) - Real(0.1)), lo.z + (hi.z - lo.z) * (rnd() * Real(1.2) - Real(0.1))};
                j.obj(); j.kv3("p", p); j.kv3("sigma_s", get_sigma_s(m, p)); j.kv3("sigma_a", get_sigma_a(m, p)); j.eobj();
            }
            j.earr();
            j.key("rays"); j.arr();
            for (int i = 0; i < 24; i++) {
                Vector3 org{lo.x + (hi.x - lo.x) * (rnd() * 3 - 1), lo.y + (hi.y - lo.y) * (rnd() * 3 - 1), lo.z + (hi.z - lo.z) * (rnd() * 3 - 1)};
                Ray ray{org, rnd_dir(), Real(0), i % 3 == 0 ? (hi.x - lo.x) * rnd() : infinity<Real>()};
                j.obj(); j.kv3("org", ray.org); j.kv3("dir", ray.dir); j.kv("tfar", ray.tfar); j.kv3("majorant", get_majorant(m, ray)); j.eobj();
            }
            j.earr();
            j.eobj();
        }
        j.earr();
        j.eobj();
    }
    j.eobj();
    j.eobj();
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: gen_golden <reference_root> <outdir>\n"); return 1; }
    fs::path ref = fs::absolute(argv[1]); std::string out = fs::absolute(argv[2]).string();
    gen_core(out);
    gen_materials(out);
    gen_scene(out, "cbox", ref / "scenes/cbox/cbox.xml", true);
    gen_scene(out, "veach_mi", ref / "scenes/veach_mi/mi.xml", true);
    gen_scene(out, "disney_bsdf", ref / "scenes/disney_bsdf_test/disney_bsdf.xml", false);
    gen_scene(out, "sponza", ref / "scenes/sponza/sponza.xml", false);
    gen_media(out, ref);
    // _exit: static destructors of the leaked fake Scenes must never run (they would call into Embree).
    fflush(nullptr); _exit(0);
}
