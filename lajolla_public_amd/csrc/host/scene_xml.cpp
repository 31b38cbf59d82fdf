// Mitsuba-0.x XML front end: file -> HostScene (the constructor arguments of the reference's Scene).
//
// Behavioural contract = parse_scene.cpp in the reference; each block cites the lines it follows.
// Things that change pixels and are therefore reproduced on purpose:
//   * every number goes through std::stof, i.e. is rounded to float first      (parse_scene.cpp:51-57,110,477)
//   * transforms compose as  new * accumulated                                 (parse_scene.cpp:130-163)
//   * an <integrator> node resets samples_per_pixel to the default (4) if it comes after the <sensor>
//     (options = parse_integrator(...) at parse_scene.cpp:1053 overwrites the whole struct)
//   * a top-level <bsdf> without id is parsed and dropped                      (parse_scene.cpp:1063-1066)
//   * a one-entry spectrum is white for BSDFs but whitepoint*value for emitters (parse_scene.cpp:180-181,944-950)
#include "host_scene.h"
#include "xml.h"
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <sstream>

namespace lj {

namespace {

std::string dirname_of(const std::string &p) {
    size_t s = p.find_last_of('/');
    return s == std::string::npos ? std::string(".") : (s == 0 ? std::string("/") : p.substr(0, s));
}
std::string join_path(const std::string &dir, const std::string &f) {
    if (!f.empty() && f[0] == '/') return f;
    return dir + "/" + f;
}
std::string lower(std::string s) { for (auto &c : s) c = (char)std::tolower((unsigned char)c); return s; }

[[noreturn]] void parse_error(const std::string &msg) { throw LjError(LJ_ERR_PARSE, msg); }

double stof_d(const std::string &s) {
    try { return (double)std::stof(s); } catch (const std::exception &) { parse_error("not a number: '" + s + "'"); }
}
int stoi_i(const std::string &s) {
    try { return std::stoi(s); } catch (const std::exception &) { parse_error("not an integer: '" + s + "'"); }
}

// std::sregex_token_iterator(..., "(,| )+", -1) semantics (parse_scene.cpp:41-45): runs of ',' / ' ' separate
// tokens; a leading run yields an empty first token; a trailing run yields nothing.
std::vector<std::string> split_list(const std::string &str, const char *delims = ", ") {
    std::vector<std::string> out;
    auto is_delim = [&](char c) { for (const char *d = delims; *d; d++) if (*d == c) return true; return false; };
    size_t i = 0, n = str.size();
    if (n == 0) { return out; }
    std::string cur;
    bool in_delim = false;
    for (; i < n; i++) {
        if (is_delim(str[i])) { if (!in_delim) { out.push_back(cur); cur.clear(); in_delim = true; } }
        else { cur.push_back(str[i]); in_delim = false; }
    }
    if (!in_delim) out.push_back(cur);
    return out;
}

V3 parse_vector3(const std::string &value) {  // parse_scene.cpp:47-62
    auto list = split_list(value);
    if (list.size() == 1) { double v = stof_d(list[0]); return {v, v, v}; }
    if (list.size() == 3) return {stof_d(list[0]), stof_d(list[1]), stof_d(list[2])};
    parse_error("parse_vector3 failed");
}

V3 parse_srgb(const std::string &value) {  // parse_scene.cpp:64-80
    if (value.size() == 7 && value[0] == '#') {
        char *end = nullptr;
        long enc = strtol(value.c_str() + 1, &end, 16);
        if (*end != '\0') parse_error("Invalid SRGB value: " + value);
        return {(double)(((enc & 0xFF0000) >> 16) / 255.0f), (double)(((enc & 0x00FF00) >> 8) / 255.0f), (double)((enc & 0x0000FF) / 255.0f)};
    }
    parse_error("Unknown SRGB format: " + value);
}

std::vector<std::pair<double, double>> parse_spectrum(const std::string &value) {  // parse_scene.cpp:82-98
    auto list = split_list(value);
    std::vector<std::pair<double, double>> s;
    if (list.size() == 1 && list[0].find(':') == std::string::npos) {
        s.emplace_back(-1.0, stof_d(list[0]));
    } else {
        for (auto &tok : list) {
            size_t c = tok.find(':');
            if (c == std::string::npos) parse_error("parse_spectrum failed");
            // the reference splits on ':' and takes fields 0 and 1
            std::string a = tok.substr(0, c), rest = tok.substr(c + 1);
            size_t c2 = rest.find(':');
            std::string b = c2 == std::string::npos ? rest : rest.substr(0, c2);
            s.emplace_back(stof_d(a), stof_d(b));
        }
    }
    return s;
}

// ---- colour science (spectrum.h:44-125): Wyman et al. analytic CIE fits, 1 nm Riemann sum over [400,700]
double xfit(double w) {
    double t1 = (w - 442.0) * ((w < 442.0) ? 0.0624 : 0.0374);
    double t2 = (w - 599.8) * ((w < 599.8) ? 0.0264 : 0.0323);
    double t3 = (w - 501.1) * ((w < 501.1) ? 0.0490 : 0.0382);
    return 0.362 * std::exp(-0.5 * t1 * t1) + 1.056 * std::exp(-0.5 * t2 * t2) - 0.065 * std::exp(-0.5 * t3 * t3);
}
double yfit(double w) {
    double t1 = (w - 568.8) * ((w < 568.8) ? 0.0213 : 0.0247);
    double t2 = (w - 530.9) * ((w < 530.9) ? 0.0613 : 0.0322);
    return 0.821 * std::exp(-0.5 * t1 * t1) + 0.286 * std::exp(-0.5 * t2 * t2);
}
double zfit(double w) {
    double t1 = (w - 437.0) * ((w < 437.0) ? 0.0845 : 0.0278);
    double t2 = (w - 459.0) * ((w < 459.0) ? 0.0385 : 0.0725);
    return 1.217 * std::exp(-0.5 * t1 * t1) + 0.681 * std::exp(-0.5 * t2 * t2);
}
V3 integrate_XYZ(const std::vector<std::pair<double, double>> &data) {
    const double cie_y_integral = 106.856895, w0 = 400, w1 = 700;
    if (data.empty()) return {0, 0, 0};
    V3 acc{0, 0, 0};
    int pos = 0, n = (int)data.size();
    for (double w = w0; w <= w1; w += 1.0) {
        while (pos < n - 1 && !((data[pos].first <= w && data[pos + 1].first > w) || data[0].first > w)) pos++;
        double m;
        if (pos < n - 1 && data[0].first <= w) {
            double cd = data[pos].second, nd = data[std::min(pos + 1, n - 1)].second;
            double cw = data[pos].first, nw = data[std::min(pos + 1, n - 1)].first;
            m = cd * (nw - w) / (nw - cw) + nd * (w - cw) / (nw - cw);
        } else {
            m = data[pos].second;
        }
        acc = acc + V3{xfit(w), yfit(w), zfit(w)} * m;
    }
    double span = w1 - w0;
    return acc * (span / (cie_y_integral * (w1 - w0)));
}
V3 xyz_to_rgb(const V3 &c) {
    return {3.240479 * c.x - 1.537150 * c.y - 0.498535 * c.z,
            -0.969256 * c.x + 1.875991 * c.y + 0.041556 * c.z,
            0.055648 * c.x - 0.204043 * c.y + 1.057311 * c.z};
}
V3 srgb_to_rgb(V3 c) {
    for (int i = 0; i < 3; i++) c[i] = c[i] <= 0.04045 ? c[i] / 12.92 : std::pow((c[i] + 0.055) / 1.055, 2.4);
    return c;
}

M4 parse_matrix4x4(const std::string &value) {  // parse_scene.cpp:100-114
    auto list = split_list(value);
    if (list.size() != 16) parse_error("parse_matrix4x4 failed");
    M4 m; int k = 0;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m(i, j) = stof_d(list[k++]);
    return m;
}

M4 parse_transform(const XmlNode &node) {  // parse_scene.cpp:116-167
    M4 tform = M4::identity();
    for (auto &cp : node.children) {
        const XmlNode &c = *cp;
        std::string name = lower(c.name);
        auto get = [&](const char *k, double dflt) { return c.has(k) ? stof_d(c.attr(k)) : dflt; };
        if (name == "scale") tform = scale({get("x", 1), get("y", 1), get("z", 1)}) * tform;
        else if (name == "translate") tform = translate({get("x", 0), get("y", 0), get("z", 0)}) * tform;
        else if (name == "rotate") { V3 ax{get("x", 0), get("y", 0), get("z", 0)}; tform = rotate(get("angle", 0), ax) * tform; }
        else if (name == "lookat") tform = look_at(parse_vector3(c.attr("origin")), parse_vector3(c.attr("target")), parse_vector3(c.attr("up"))) * tform;
        else if (name == "matrix") tform = parse_matrix4x4(c.attr("value")) * tform;
    }
    return tform;
}

enum class TexType { Bitmap, Checkerboard };
struct ParsedTexture {  // parse_scene.cpp:25-31
    TexType type = TexType::Bitmap;
    std::string filename;
    V3 color0, color1;
    double uscale = 1, vscale = 1, uoffset = 0, voffset = 0;
};

LjTexture const_tex3(const V3 &v) {
    LjTexture t{}; t.kind = LJ_TEX_CONSTANT; t.texture_id = -1;
    t.value[0] = v.x; t.value[1] = v.y; t.value[2] = v.z; t.uscale = t.vscale = 1; return t;
}
LjTexture const_tex1(double v) { return const_tex3({v, v, v}); }

struct Parser {
    HostScene &hs;
    std::string base_dir;
    std::map<std::string, int> material_map;
    std::map<std::string, ParsedTexture> texture_map;
    std::map<std::string, int> medium_map;

    // TexturePool::insert_image3 / insert_image1 (texture.h:21-63): keyed by texture name, first insert wins.
    int insert_image3(const std::string &name, const std::string &file) {
        auto it = hs.image3s_map.find(name);
        if (it != hs.image3s_map.end()) return it->second;
        int id = (int)hs.images3.size();
        hs.image3s_map[name] = id;
        hs.images3.push_back(read_image(join_path(base_dir, file), 3));
        return id;
    }
    int insert_image1(const std::string &name, HostImage img) {
        auto it = hs.image1s_map.find(name);
        if (it != hs.image1s_map.end()) return it->second;
        int id = (int)hs.images1.size();
        hs.image1s_map[name] = id;
        hs.images1.push_back(std::move(img));
        return id;
    }
    const ParsedTexture &find_texture(const std::string &id) {
        auto it = texture_map.find(id);
        if (it == texture_map.end()) parse_error("Texture not found. ID = " + id);
        return it->second;
    }

    LjTexture parse_spectrum_texture(const XmlNode &n) {  // parse_scene.cpp:169-213
        const std::string &type = n.name;
        if (type == "spectrum") {
            auto spec = parse_spectrum(n.attr("value"));
            if (spec.size() > 1) return const_tex3(xyz_to_rgb(integrate_XYZ(spec)));
            if (spec.size() == 1) return const_tex3({1, 1, 1});
            return const_tex3({0, 0, 0});
        } else if (type == "rgb") {
            return const_tex3(parse_vector3(n.attr("value")));
        } else if (type == "srgb") {
            return const_tex3(srgb_to_rgb(parse_srgb(n.attr("value"))));
        } else if (type == "ref") {
            const std::string &ref_id = n.attr("id");
            const ParsedTexture &t = find_texture(ref_id);
            LjTexture out{};
            out.uscale = t.uscale; out.vscale = t.vscale; out.uoffset = t.uoffset; out.voffset = t.voffset;
            if (t.type == TexType::Bitmap) {
                out.kind = LJ_TEX_IMAGE; out.texture_id = insert_image3(ref_id, t.filename);
            } else {
                out.kind = LJ_TEX_CHECKERBOARD; out.texture_id = -1;
                for (int i = 0; i < 3; i++) { out.value[i] = t.color0[i]; out.color1[i] = t.color1[i]; }
            }
            return out;
        }
        parse_error("Unknown spectrum texture type:" + type);
    }

    LjTexture parse_float_texture(const XmlNode &n) {  // parse_scene.cpp:215-237
        const std::string &type = n.name;
        if (type == "ref") {
            const std::string &ref_id = n.attr("id");
            const ParsedTexture &t = find_texture(ref_id);
            LjTexture out{};
            out.kind = LJ_TEX_IMAGE;
            out.texture_id = insert_image1(ref_id, read_image(join_path(base_dir, t.filename), 1));
            out.uscale = t.uscale; out.vscale = t.vscale;  // offsets are not forwarded here (parse_scene.cpp:228-229)
            return out;
        } else if (type == "float") {
            return const_tex1(stof_d(n.attr("value")));
        }
        parse_error("Unknown float texture type:" + type);
    }

    // "alpha" -> roughness = sqrt(alpha) (parse_scene.cpp:592-617, 644-667)
    LjTexture parse_alpha_as_roughness(const XmlNode &n) {
        const std::string &type = n.name;
        if (type == "ref") {
            const std::string &ref_id = n.attr("id");
            const ParsedTexture &t = find_texture(ref_id);
            HostImage img = read_image(join_path(base_dir, t.filename), 1);
            for (auto &v : img.data) v = (float)std::sqrt((double)v);
            LjTexture out{};
            out.kind = LJ_TEX_IMAGE; out.texture_id = insert_image1(ref_id, std::move(img));
            out.uscale = t.uscale; out.vscale = t.vscale;
            return out;
        } else if (type == "float") {
            return const_tex1(std::sqrt(stof_d(n.attr("value"))));
        }
        parse_error("Unknown float texture type:" + type);
    }

    V3 parse_color(const XmlNode &n) {  // parse_scene.cpp:239-263
        const std::string &type = n.name;
        if (type == "spectrum") {
            auto spec = parse_spectrum(n.attr("value"));
            if (spec.size() > 1) return xyz_to_rgb(integrate_XYZ(spec));
            if (spec.size() == 1) return {1, 1, 1};
            return {0, 0, 0};
        } else if (type == "rgb") return parse_vector3(n.attr("value"));
        else if (type == "srgb") return srgb_to_rgb(parse_srgb(n.attr("value")));
        else if (type == "float") { double v = stof_d(n.attr("value")); return {v, v, v}; }
        parse_error("Unknown color type:" + type);
    }

    LjRenderOptions parse_integrator(const XmlNode &n) {  // parse_scene.cpp:265-309; defaults scene.h:24-31
        LjRenderOptions o{};
        o.integrator = LJ_INTEGRATOR_PATH; o.samples_per_pixel = 4; o.max_depth = -1; o.rr_depth = 5;
        o.vol_path_version = 0; o.max_null_collisions = 1000;
        const std::string &type = n.attr("type");
        if (type == "path" || type == "volpath") {
            o.integrator = type == "path" ? LJ_INTEGRATOR_PATH : LJ_INTEGRATOR_VOLPATH;
            for (auto &cp : n.children) {
                const std::string &name = cp->attr("name");
                if (name == "maxDepth") o.max_depth = stoi_i(cp->attr("value"));
                else if (name == "rrDepth") o.rr_depth = stoi_i(cp->attr("value"));
                else if (type == "volpath" && name == "version") o.vol_path_version = stoi_i(cp->attr("value"));
                else if (type == "volpath" && name == "maxNullCollisions") o.max_null_collisions = stoi_i(cp->attr("value"));
            }
        } else if (type == "direct") { o.integrator = LJ_INTEGRATOR_PATH; o.max_depth = 2; }
        else if (type == "depth") o.integrator = LJ_INTEGRATOR_DEPTH;
        else if (type == "shadingNormal") o.integrator = LJ_INTEGRATOR_SHADING_NORMAL;
        else if (type == "meanCurvature") o.integrator = LJ_INTEGRATOR_MEAN_CURVATURE;
        else if (type == "rayDifferential") o.integrator = LJ_INTEGRATOR_RAY_DIFFERENTIAL;
        else if (type == "mipmapLevel") o.integrator = LJ_INTEGRATOR_MIPMAP_LEVEL;
        else parse_error("Unsupported integrator: " + type);
        return o;
    }

    struct Film { int width = 256, height = 256; std::string filename = "image.exr"; int filter_kind = LJ_FILTER_BOX; double filter_param = 1; };

    Film parse_film(const XmlNode &n) {  // parse_scene.cpp:311-357
        Film f;
        for (auto &cp : n.children) {
            const XmlNode &c = *cp;
            const std::string &name = c.attr("name");
            if (name == "width") f.width = stoi_i(c.attr("value"));
            else if (name == "height") f.height = stoi_i(c.attr("value"));
            else if (name == "filename") f.filename = c.attr("value");
            if (c.name == "rfilter") {
                const std::string &ft = c.attr("type");
                auto grand = [&](const char *key, double dflt) {
                    double v = dflt;
                    for (auto &g : c.children) if (g->attr("name") == key) v = stof_d(g->attr("value"));
                    return v;
                };
                if (ft == "box") { f.filter_kind = LJ_FILTER_BOX; f.filter_param = grand("width", 1); }
                else if (ft == "tent") { f.filter_kind = LJ_FILTER_TENT; f.filter_param = grand("width", 2); }
                else if (ft == "gaussian") { f.filter_kind = LJ_FILTER_GAUSSIAN; f.filter_param = grand("stddev", 0.5); }
            }
        }
        return f;
    }

    static void store(double dst[16], const M4 &m) { for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) dst[i * 4 + j] = m(i, j); }

    // Camera::Camera (camera.cpp:7-21)
    static LjCamera make_camera(const M4 &cam_to_world, double fov, int width, int height, int filter_kind, double filter_param) {
        LjCamera cam{};
        double aspect = (double)width / (double)height;
        M4 cam_to_sample = scale({-0.5, -0.5 * aspect, 1.0}) * translate({-1.0, -1.0 / aspect, 0.0}) * perspective(fov);
        store(cam.cam_to_world, cam_to_world); store(cam.world_to_cam, inverse(cam_to_world));
        store(cam.cam_to_sample, cam_to_sample); store(cam.sample_to_cam, inverse(cam_to_sample));
        cam.width = width; cam.height = height; cam.filter_kind = filter_kind; cam.filter_param = filter_param; cam.medium_id = -1;
        return cam;
    }

    int parse_sensor(const XmlNode &n) {  // parse_scene.cpp:459-556; returns the sampler's sampleCount
        double fov = 45.0; M4 to_world = M4::identity(); Film film; int sample_count = 4; int sensor_medium_id = -1;
        enum { X, Y, DIAGONAL, SMALLER, LARGER } axis = X;
        const std::string &type = n.attr("type");
        if (type != "perspective") parse_error("Unsupported sensor: " + type);
        for (auto &cp : n.children) {
            const XmlNode &c = *cp;
            const std::string &name = c.attr("name");
            if (name == "fov") fov = stof_d(c.attr("value"));
            else if (name == "toWorld") to_world = parse_transform(c);
            else if (name == "fovAxis") {
                const std::string &v = c.attr("value");
                if (v == "x") axis = X; else if (v == "y") axis = Y; else if (v == "diagonal") axis = DIAGONAL;
                else if (v == "smaller") axis = SMALLER; else if (v == "larger") axis = LARGER;
                else parse_error("Unknown fovAxis value: " + v);
            }
        }
        for (auto &cp : n.children) {
            const XmlNode &c = *cp;
            if (c.name == "film") film = parse_film(c);
            else if (c.name == "sampler") {
                if (c.attr("type") != "independent")
                    fprintf(stderr, "Warning: the renderer currently only supports independent samplers.\n");
                for (auto &g : c.children) if (g->attr("name") == "sampleCount") sample_count = stoi_i(g->attr("value"));
            }
            else if (c.name == "ref") {  // a reference to a medium (parse_scene.cpp:517-527)
                if (!c.has("id")) parse_error("Medium reference not specified.");
                auto mt = medium_map.find(c.attr("id"));
                if (mt == medium_map.end()) parse_error("Medium reference " + c.attr("id") + " not found.");
                sensor_medium_id = mt->second;
            } else if (c.name == "medium") {
                LjMedium m; std::string mname = parse_medium(c, m);
                if (!mname.empty()) medium_map[mname] = (int)hs.media.size();
                sensor_medium_id = (int)hs.media.size();
                hs.media.push_back(m);
            }
        }
        int width = film.width, height = film.height;
        if (axis == Y || (axis == SMALLER && height < width) || (axis == LARGER && width < height)) {
            double aspect = width / (double)height;
            fov = degrees(2 * std::atan(std::tan(radians(fov) / 2) * aspect));
        } else if (axis == DIAGONAL) {
            double aspect = width / (double)height;
            double diagonal = 2 * std::tan(radians(fov) / 2);
            double wd = diagonal / std::sqrt(1 + 1 / (aspect * aspect));
            fov = degrees(2 * std::atan(wd / 2));
        }
        hs.camera = make_camera(to_world, fov, width, height, film.filter_kind, film.filter_param);
        hs.camera.medium_id = sensor_medium_id;
        hs.output_filename = film.filename;
        return sample_count;
    }

    // parse_bsdf (parse_scene.cpp:558-809).  Table-driven: slot names per alternative + defaults.
    struct SlotSpec { const char *xml_name; bool spectrum; double dflt; };
    bool parse_bsdf(const XmlNode &n, std::string &id_out, LjMaterial &m) {
        const std::string &type = n.attr("type");
        id_out = n.has("id") ? n.attr("id") : std::string();
        m = LjMaterial{};
        std::vector<SlotSpec> slots;
        bool has_ior_pair = false, has_eta = false;
        double int_ior = 1.49, ext_ior = 1.000277, eta = 1.5;
        if (type == "diffuse") { m.kind = LJ_MAT_LAMBERTIAN; slots = {{"reflectance", true, 0.5}}; }
        else if (type == "roughplastic" || type == "plastic") {
            m.kind = LJ_MAT_ROUGHPLASTIC; has_ior_pair = true; int_ior = 1.49;
            slots = {{"diffuseReflectance", true, 0.5}, {"specularReflectance", true, 1.0}, {"roughness", false, type == "plastic" ? 0.01 : 0.1}};
        } else if (type == "roughdielectric" || type == "dielectric") {
            m.kind = LJ_MAT_ROUGHDIELECTRIC; has_ior_pair = true; int_ior = 1.5046;
            slots = {{"specularReflectance", true, 1.0}, {"specularTransmittance", true, 1.0}, {"roughness", false, type == "dielectric" ? 0.01 : 0.1}};
        } else if (type == "disneydiffuse") { m.kind = LJ_MAT_DISNEYDIFFUSE; slots = {{"baseColor", true, 0.5}, {"roughness", false, 0.5}, {"subsurface", false, 0}}; }
        else if (type == "disneymetal") { m.kind = LJ_MAT_DISNEYMETAL; slots = {{"baseColor", true, 0.5}, {"roughness", false, 0.5}, {"anisotropic", false, 0}}; }
        else if (type == "disneyglass") { m.kind = LJ_MAT_DISNEYGLASS; has_eta = true; slots = {{"baseColor", true, 0.5}, {"roughness", false, 0.5}, {"anisotropic", false, 0}}; }
        else if (type == "disneyclearcoat") { m.kind = LJ_MAT_DISNEYCLEARCOAT; slots = {{"clearcoatGloss", false, 1.0}}; }
        else if (type == "disneysheen") { m.kind = LJ_MAT_DISNEYSHEEN; slots = {{"baseColor", true, 0.5}, {"sheenTint", false, 0.5}}; }
        else if (type == "disneybsdf") {
            m.kind = LJ_MAT_DISNEYBSDF; has_eta = true;
            slots = {{"baseColor", true, 0.5}, {"specularTransmission", false, 0}, {"metallic", false, 0}, {"subsurface", false, 0},
                     {"specular", false, 0.5}, {"roughness", false, 0.5}, {"specularTint", false, 0}, {"anisotropic", false, 0},
                     {"sheen", false, 0}, {"sheenTint", false, 0.5}, {"clearcoat", false, 0}, {"clearcoatGloss", false, 1}};
        } else parse_error("Unknown BSDF: " + type);
        m.n_tex = (int)slots.size();
        for (int i = 0; i < m.n_tex; i++) m.tex[i] = slots[i].spectrum ? const_tex3({slots[i].dflt, slots[i].dflt, slots[i].dflt}) : const_tex1(slots[i].dflt);
        for (auto &cp : n.children) {
            const XmlNode &c = *cp;
            const std::string &name = c.attr("name");
            bool matched = false;
            for (int i = 0; i < m.n_tex && !matched; i++) {
                if (name == slots[i].xml_name) { m.tex[i] = slots[i].spectrum ? parse_spectrum_texture(c) : parse_float_texture(c); matched = true; }
            }
            if (matched) continue;
            if (has_ior_pair && name == "alpha") m.tex[2] = parse_alpha_as_roughness(c);
            else if (has_ior_pair && name == "intIOR") int_ior = stof_d(c.attr("value"));
            else if (has_ior_pair && name == "extIOR") ext_ior = stof_d(c.attr("value"));
            else if (has_eta && name == "eta") eta = stof_d(c.attr("value"));
        }
        m.eta = has_ior_pair ? int_ior / ext_ior : eta;
        return true;
    }

    // parse_volume_spectrum (parse_scene.cpp:359-386)
    LjVolume parse_volume(const XmlNode &n, int medium_index, int which) {
        LjVolume v{};
        const std::string &type = n.attr("type");
        if (type == "constvolume") {
            v.kind = LJ_VOLUME_CONSTANT;
            for (auto &cp : n.children) if (cp->attr("name") == "value") { V3 c = parse_color(*cp); v.value[0] = c.x; v.value[1] = c.y; v.value[2] = c.z; }
            v.scale = 1;
        } else if (type == "gridvolume") {
            std::string filename;
            for (auto &cp : n.children) if (cp->attr("name") == "filename") filename = cp->attr("value");
            if (filename.empty()) parse_error("Empty filename for a gridvolume.");
            hs.volume_data.emplace_back();
            hs.volume_owner.emplace_back(medium_index, which);
            load_grid_volume(join_path(base_dir, filename), v, hs.volume_data.back());
        } else parse_error("Unknown volume type:" + type);
        return v;
    }

    // parse_medium (parse_scene.cpp:407-457) with parse_phase_function (:388-405); returns the medium's id attribute
    std::string parse_medium(const XmlNode &n, LjMedium &m) {
        m = LjMedium{};
        m.phase_kind = LJ_PHASE_ISOTROPIC;
        auto parse_phase = [&](const XmlNode &c) {
            const std::string &pt = c.attr("type");
            if (pt == "isotropic") { m.phase_kind = LJ_PHASE_ISOTROPIC; m.g = 0; }
            else if (pt == "hg") {
                m.phase_kind = LJ_PHASE_HG; m.g = 0;
                for (auto &g : c.children) if (g->attr("name") == "g") m.g = stof_d(g->attr("value"));
            } else parse_error("Unrecognized phase function:" + pt);
        };
        const std::string &type = n.attr("type");
        const int index = (int)hs.media.size();
        if (type == "homogeneous") {
            m.kind = LJ_MEDIUM_HOMOGENEOUS;
            V3 sa{0.5, 0.5, 0.5}, ss{0.5, 0.5, 0.5}; double scl = 1;
            for (auto &cp : n.children) {
                const XmlNode &c = *cp; const std::string &name = c.attr("name");
                if (name == "sigmaA") sa = parse_color(c);
                else if (name == "sigmaS") ss = parse_color(c);
                else if (name == "scale") scl = stof_d(c.attr("value"));
                else if (c.name == "phase") parse_phase(c);
            }
            sa = sa * scl; ss = ss * scl;
            m.sigma_a[0] = sa.x; m.sigma_a[1] = sa.y; m.sigma_a[2] = sa.z; m.sigma_s[0] = ss.x; m.sigma_s[1] = ss.y; m.sigma_s[2] = ss.z;
        } else if (type == "heterogeneous") {
            m.kind = LJ_MEDIUM_HETEROGENEOUS;
            m.albedo.kind = LJ_VOLUME_CONSTANT; m.albedo.value[0] = m.albedo.value[1] = m.albedo.value[2] = 1; m.albedo.scale = 1;
            m.density = m.albedo;
            double scl = 1;
            for (auto &cp : n.children) {
                const XmlNode &c = *cp; const std::string &name = c.attr("name");
                if (name == "albedo") m.albedo = parse_volume(c, index, 0);
                else if (name == "density") m.density = parse_volume(c, index, 1);
                else if (name == "scale") scl = stof_d(c.attr("value"));
                else if (c.name == "phase") parse_phase(c);
            }
            // "scale only applies to density" (parse_scene.cpp:449-450; set_scale, volume.h:100-111)
            if (m.density.kind == LJ_VOLUME_CONSTANT) for (int k = 0; k < 3; k++) m.density.value[k] *= scl;
            else m.density.scale = scl;
        } else parse_error("Unknown medium type:" + type);
        return n.has("id") ? n.attr("id") : std::string();
    }

    void parse_shape(const XmlNode &n) {  // parse_scene.cpp:811-970
        int material_id = -1, interior_medium_id = -1, exterior_medium_id = -1;
        for (auto &cp : n.children) {
            const XmlNode &c = *cp;
            if (c.name == "ref") {
                const std::string &nv = c.attr("name");
                if (!c.has("id")) parse_error("Material/medium reference id not specified.");
                if (nv == "interior" || nv == "exterior") {
                    auto mt = medium_map.find(c.attr("id"));
                    if (mt == medium_map.end()) parse_error("Medium reference " + c.attr("id") + " not found.");
                    (nv == "interior" ? interior_medium_id : exterior_medium_id) = mt->second;
                    continue;
                }
                auto it = material_map.find(c.attr("id"));
                if (it == material_map.end()) parse_error("Material reference " + c.attr("id") + " not found.");
                material_id = it->second;
            } else if (c.name == "bsdf") {
                std::string mname; LjMaterial m;
                parse_bsdf(c, mname, m);
                if (!mname.empty()) material_map[mname] = (int)hs.materials.size();
                material_id = (int)hs.materials.size();
                hs.materials.push_back(m);
            } else if (c.name == "medium") {
                LjMedium m; std::string mname = parse_medium(c, m);
                if (!mname.empty()) medium_map[mname] = (int)hs.media.size();
                const std::string &nv = c.attr("name");
                if (nv == "interior") interior_medium_id = (int)hs.media.size();
                else if (nv == "exterior") exterior_medium_id = (int)hs.media.size();
                else parse_error("Unrecognized medium name: " + nv);
                hs.media.push_back(m);
            }
        }
        const std::string &type = n.attr("type");
        LjShape shape{};
        auto mesh_params = [&](std::string &filename, M4 &to_world, int *shape_index) {
            for (auto &cp : n.children) {
                const XmlNode &c = *cp;
                const std::string &name = c.attr("name");
                if (name == "filename") filename = c.attr("value");
                else if (name == "toWorld") { if (c.name == "transform") to_world = parse_transform(c); }
                else if (shape_index && name == "shapeIndex") *shape_index = stoi_i(c.attr("value"));
            }
        };
        if (type == "obj") {
            std::string filename; M4 to_world = M4::identity();
            mesh_params(filename, to_world, nullptr);
            shape = load_obj_mesh(hs, join_path(base_dir, filename), to_world);
        } else if (type == "serialized") {
            std::string filename; M4 to_world = M4::identity(); int shape_index = 0;
            mesh_params(filename, to_world, &shape_index);
            shape = load_serialized_mesh(hs, join_path(base_dir, filename), shape_index, to_world);
        } else if (type == "sphere") {
            shape.kind = LJ_SHAPE_SPHERE; shape.radius = 1;
            for (auto &cp : n.children) {
                const XmlNode &c = *cp;
                const std::string &name = c.attr("name");
                if (name == "center") { shape.position[0] = stof_d(c.attr("x")); shape.position[1] = stof_d(c.attr("y")); shape.position[2] = stof_d(c.attr("z")); }
                else if (name == "radius") shape.radius = stof_d(c.attr("value"));
            }
        } else parse_error("Unknown shape:" + type);
        shape.material_id = material_id; shape.area_light_id = -1;
        shape.interior_medium_id = interior_medium_id; shape.exterior_medium_id = exterior_medium_id;
        for (auto &cp : n.children) {
            const XmlNode &c = *cp;
            if (c.name != "emitter") continue;
            V3 radiance{1, 1, 1};
            for (auto &g : c.children) {
                if (g->attr("name") != "radiance") continue;
                const std::string &rt = g->name;
                if (rt == "spectrum") {
                    auto spec = parse_spectrum(g->attr("value"));
                    if (spec.size() == 1) radiance = xyz_to_rgb(V3{0.9505, 1.0, 1.0888} * spec[0].second);
                    else radiance = xyz_to_rgb(integrate_XYZ(spec));
                } else if (rt == "rgb") radiance = parse_vector3(g->attr("value"));
                else if (rt == "srgb") radiance = srgb_to_rgb(parse_srgb(g->attr("value")));
            }
            shape.area_light_id = (int)hs.lights.size();
            LjLight l{}; l.kind = LJ_LIGHT_AREA; l.shape_id = (int)hs.shapes.size();
            l.intensity[0] = radiance.x; l.intensity[1] = radiance.y; l.intensity[2] = radiance.z; l.scale = 1;
            hs.lights.push_back(l);
        }
        hs.shapes.push_back(shape);
    }

    ParsedTexture parse_texture(const XmlNode &n) {  // parse_scene.cpp:973-1030
        const std::string &type = n.attr("type");
        ParsedTexture t;
        if (type == "bitmap") t.type = TexType::Bitmap;
        else if (type == "checkerboard") { t.type = TexType::Checkerboard; t.color0 = {0.4, 0.4, 0.4}; t.color1 = {0.2, 0.2, 0.2}; }
        else parse_error("Unknown texture type: " + type);
        for (auto &cp : n.children) {
            const XmlNode &c = *cp;
            const std::string &name = c.attr("name");
            if (t.type == TexType::Bitmap && name == "filename") t.filename = c.attr("value");
            else if (t.type == TexType::Checkerboard && name == "color0") t.color0 = parse_color(c);
            else if (t.type == TexType::Checkerboard && name == "color1") t.color1 = parse_color(c);
            else if (name == "uvscale") t.uscale = t.vscale = stof_d(c.attr("value"));
            else if (name == "uscale") t.uscale = stof_d(c.attr("value"));
            else if (name == "vscale") t.vscale = stof_d(c.attr("value"));
            else if (name == "uoffset") t.uoffset = stof_d(c.attr("value"));
            else if (name == "voffset") t.voffset = stof_d(c.attr("value"));
        }
        return t;
    }

    void parse_root(const XmlNode &root) {  // parse_scene.cpp:1032-1121
        hs.options = LjRenderOptions{LJ_INTEGRATOR_PATH, 4, -1, 5, 0, 1000};
        hs.camera = make_camera(M4::identity(), 45.0, 256, 256, LJ_FILTER_BOX, 1.0);
        for (auto &cp : root.children) {
            const XmlNode &c = *cp;
            if (c.name == "integrator") hs.options = parse_integrator(c);
            else if (c.name == "sensor") hs.options.samples_per_pixel = parse_sensor(c);
            else if (c.name == "bsdf") {
                std::string mname; LjMaterial m;
                parse_bsdf(c, mname, m);
                if (!mname.empty()) { material_map[mname] = (int)hs.materials.size(); hs.materials.push_back(m); }
            } else if (c.name == "shape") parse_shape(c);
            else if (c.name == "texture") {
                const std::string &id = c.attr("id");
                if (texture_map.count(id)) parse_error("Duplicated texture ID:" + id);
                texture_map[id] = parse_texture(c);
            } else if (c.name == "emitter") {
                const std::string &type = c.attr("type");
                if (type != "envmap") parse_error("Unknown emitter type:" + type);
                std::string filename; double scl = 1; M4 to_world = M4::identity();
                for (auto &g : c.children) {
                    const std::string &name = g->attr("name");
                    if (name == "filename") filename = g->attr("value");
                    else if (name == "toWorld") to_world = parse_transform(*g);
                    else if (name == "scale") scl = stof_d(g->attr("value"));
                }
                if (filename.empty()) parse_error("Filename unspecified for envmap.");
                LjLight l{}; l.kind = LJ_LIGHT_ENVMAP; l.shape_id = -1;
                l.values.kind = LJ_TEX_IMAGE; l.values.texture_id = insert_image3("__envmap_texture__", filename);
                l.values.uscale = l.values.vscale = 1; l.values.uoffset = l.values.voffset = 0;
                store(l.to_world, to_world); store(l.to_local, inverse(to_world)); l.scale = scl;
                hs.lights.push_back(l);
                hs.envmap_light_id = (int)hs.lights.size() - 1;
            }
            else if (c.name == "medium") {  // parse_scene.cpp:1111-1119: only named top-level media are kept
                LjMedium m; std::string mname = parse_medium(c, m);
                if (!mname.empty()) { medium_map[mname] = (int)hs.media.size(); hs.media.push_back(m); }
                else { while (!hs.volume_owner.empty() && hs.volume_owner.back().first == (int)hs.media.size()) { hs.volume_owner.pop_back(); hs.volume_data.pop_back(); } }
            }
        }
    }
};

} // namespace

HostScene *parse_scene_xml(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw LjError(LJ_ERR_IO, "cannot open scene file: " + path);
    std::stringstream ss; ss << f.rdbuf();
    std::string text = ss.str();
    std::unique_ptr<XmlNode> doc;
    try { doc = XmlParser(text).parse_document(); }
    catch (const std::runtime_error &e) { throw LjError(LJ_ERR_PARSE, std::string(e.what()) + " in " + path); }
    if (doc->name != "scene") throw LjError(LJ_ERR_PARSE, "root element is <" + doc->name + ">, expected <scene>");
    auto hs = std::make_unique<HostScene>();
    Parser p{*hs, dirname_of(path), {}, {}, {}};
    p.parse_root(*doc);
    hs->finalize();
    return hs.release();
}

void HostScene::finalize() {
    image3_views.clear(); image1_views.clear();
    for (auto &im : images3) image3_views.push_back(LjImage{im.width, im.height, im.channels, 0, im.data.data()});
    for (auto &im : images1) image1_views.push_back(LjImage{im.width, im.height, im.channels, 0, im.data.data()});
    desc = LjSceneDesc{};
    desc.camera = camera; desc.options = options;
    desc.n_shapes = (int)shapes.size(); desc.n_materials = (int)materials.size(); desc.n_lights = (int)lights.size();
    desc.n_images3 = (int)images3.size(); desc.n_images1 = (int)images1.size(); desc.envmap_light_id = envmap_light_id;
    desc.shapes = shapes.data(); desc.materials = materials.data(); desc.lights = lights.data();
    desc.images3 = image3_views.data(); desc.images1 = image1_views.data();
    desc.n_vertices = (int64_t)positions.size() / 3; desc.n_triangles = (int64_t)indices.size() / 3;
    desc.positions = positions.data(); desc.normals = normals.data(); desc.uvs = uvs.data(); desc.indices = indices.data();
    desc.output_filename = output_filename.c_str();
    for (size_t i = 0; i < volume_data.size(); i++) {
        LjMedium &m = media[volume_owner[i].first];
        (volume_owner[i].second == 0 ? m.albedo : m.density).data = volume_data[i].data();
    }
    desc.n_media = (int)media.size(); desc.media = media.data();
}

HostScene *host_scene_from_desc(const LjSceneDesc &d) {
    auto hs = std::make_unique<HostScene>();
    hs->camera = d.camera; hs->options = d.options;
    hs->shapes.assign(d.shapes, d.shapes + d.n_shapes);
    hs->materials.assign(d.materials, d.materials + d.n_materials);
    hs->lights.assign(d.lights, d.lights + d.n_lights);
    auto copy_images = [](const LjImage *src, int n, std::vector<HostImage> &dst) {
        for (int i = 0; i < n; i++) {
            HostImage im; im.width = src[i].width; im.height = src[i].height; im.channels = src[i].channels;
            im.data.assign(src[i].data, src[i].data + (size_t)im.width * im.height * im.channels);
            dst.push_back(std::move(im));
        }
    };
    copy_images(d.images3, d.n_images3, hs->images3);
    copy_images(d.images1, d.n_images1, hs->images1);
    hs->positions.assign(d.positions, d.positions + d.n_vertices * 3);
    if (d.normals) hs->normals.assign(d.normals, d.normals + d.n_vertices * 3); else hs->normals.assign(d.n_vertices * 3, 0.0);
    if (d.uvs) hs->uvs.assign(d.uvs, d.uvs + d.n_vertices * 2); else hs->uvs.assign(d.n_vertices * 2, 0.0);
    hs->indices.assign(d.indices, d.indices + d.n_triangles * 3);
    hs->envmap_light_id = d.envmap_light_id;
    if (d.output_filename) hs->output_filename = d.output_filename;
    for (int i = 0; i < d.n_media; i++) {
        hs->media.push_back(d.media[i]);
        LjVolume *vols[2] = {&hs->media.back().albedo, &hs->media.back().density};
        for (int w = 0; w < 2; w++) {
            LjVolume &v = *vols[w];
            if (d.media[i].kind != LJ_MEDIUM_HETEROGENEOUS || v.kind != LJ_VOLUME_GRID) { v.data = nullptr; continue; }
            const size_t n = (size_t)v.resolution[0] * v.resolution[1] * v.resolution[2] * 3;
            hs->volume_data.emplace_back(v.data, v.data + n);
            hs->volume_owner.emplace_back(i, w);
        }
    }
    hs->finalize();
    return hs.release();
}

} // namespace lj
