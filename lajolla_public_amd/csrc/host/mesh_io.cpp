// Mesh loaders of the front end: Wavefront OBJ and Mitsuba .serialized -> pooled TriangleMesh arrays.
// Behaviour follows parse_obj.cpp:137-234 and load_serialized.cpp:103-256 of the reference; quirks kept
// because they change vertex data: `vt` is stored as (s, 1-t) (parse_obj.cpp:166); quads fan as (0,1,2),(0,2,3)
// (parse_obj.cpp:199-212); transformed normals are NOT renormalised (parse_obj.cpp:131, load_serialized.cpp:217);
// when an OBJ has no `vn`, angle-weighted normals are synthesised, including the reference's obtuse-angle
// expression `(pi - 2) * asin(...)` (parse_obj.cpp:50-55,57-92).
#include "host_scene.h"
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <zlib.h>

namespace lj {

namespace {

struct ObjKey {
    int v, vt, vn;
    bool operator<(const ObjKey &o) const {
        if (v != o.v) return v < o.v;
        if (vt != o.vt) return vt < o.vt;
        return vn < o.vn;
    }
};

// "a/b/c" -> {a, b, c}; missing / empty fields are 0 (parse_obj.cpp:31-46)
void split_face(const std::string &s, int out[3]) {
    out[0] = out[1] = out[2] = 0;
    size_t b = 0; int k = 0;
    while (k < 3 && b <= s.size()) {
        size_t e = s.find('/', b);
        std::string f = s.substr(b, e == std::string::npos ? std::string::npos : e - b);
        if (!f.empty()) {
            try { out[k] = std::stoi(f); } catch (const std::exception &) { throw LjError(LJ_ERR_PARSE, "bad face index '" + s + "'"); }
        }
        k++;
        if (e == std::string::npos) break;
        b = e + 1;
    }
}

double unit_angle(const V3 &u, const V3 &v) {  // parse_obj.cpp:50-55, as written
    if (dot(u, v) < 0) return (kPi - 2) * std::asin(0.5 * length(v + u));
    return 2 * std::asin(0.5 * length(v - u));
}

LjShape append_mesh(HostScene &hs, const std::vector<V3> &pos, const std::vector<V3> &nor, const std::vector<V2> &uv,
                    const std::vector<int32_t> &idx) {
    if (!nor.empty() && nor.size() != pos.size()) throw LjError(LJ_ERR_PARSE, "mesh has normals on only some vertices");
    if (!uv.empty() && uv.size() != pos.size()) throw LjError(LJ_ERR_PARSE, "mesh has uvs on only some vertices");
    for (int32_t i : idx) if (i < 0 || (size_t)i >= pos.size()) throw LjError(LJ_ERR_PARSE, "mesh index out of range");
    LjShape s{};
    s.kind = LJ_SHAPE_TRIMESH;
    s.material_id = s.area_light_id = s.interior_medium_id = s.exterior_medium_id = -1;
    s.has_normals = !nor.empty(); s.has_uvs = !uv.empty();
    s.first_vertex = (int64_t)hs.positions.size() / 3; s.n_vertices = (int64_t)pos.size();
    s.first_triangle = (int64_t)hs.indices.size() / 3; s.n_triangles = (int64_t)idx.size() / 3;
    for (size_t i = 0; i < pos.size(); i++) {
        hs.positions.insert(hs.positions.end(), {pos[i].x, pos[i].y, pos[i].z});
        if (s.has_normals) hs.normals.insert(hs.normals.end(), {nor[i].x, nor[i].y, nor[i].z}); else hs.normals.insert(hs.normals.end(), {0.0, 0.0, 0.0});
        if (s.has_uvs) hs.uvs.insert(hs.uvs.end(), {uv[i].x, uv[i].y}); else hs.uvs.insert(hs.uvs.end(), {0.0, 0.0});
    }
    hs.indices.insert(hs.indices.end(), idx.begin(), idx.end());
    return s;
}

} // namespace

LjShape load_obj_mesh(HostScene &hs, const std::string &filename, const M4 &to_world) {
    std::ifstream ifs(filename);
    if (!ifs.is_open()) throw LjError(LJ_ERR_IO, "Unable to open the obj file: " + filename);
    std::vector<V3> pos_pool, nor_pool; std::vector<V2> st_pool;
    std::vector<V3> pos, nor; std::vector<V2> st; std::vector<int32_t> idx;
    std::map<ObjKey, int32_t> vertex_map;
    const M4 inv_world = inverse(to_world);
    auto vertex_id = [&](const int f[3]) -> int32_t {
        ObjKey key{f[0] - 1, f[1] - 1, f[2] - 1};
        auto it = vertex_map.find(key);
        if (it != vertex_map.end()) return it->second;
        if (key.v < 0 || (size_t)key.v >= pos_pool.size()) throw LjError(LJ_ERR_PARSE, "obj face references a missing vertex in " + filename);
        int32_t id = (int32_t)pos.size();
        pos.push_back(xform_point(to_world, pos_pool[key.v]));
        if (key.vt != -1) {
            if ((size_t)key.vt >= st_pool.size()) throw LjError(LJ_ERR_PARSE, "obj face references a missing vt in " + filename);
            st.push_back(st_pool[key.vt]);
        }
        if (key.vn != -1) {
            if ((size_t)key.vn >= nor_pool.size()) throw LjError(LJ_ERR_PARSE, "obj face references a missing vn in " + filename);
            nor.push_back(xform_normal(inv_world, nor_pool[key.vn]));
        }
        vertex_map[key] = id;
        return id;
    };
    std::string line;
    while (std::getline(ifs, line)) {
        size_t b = line.find_first_not_of(" \t\r\n\v\f"), e = line.find_last_not_of(" \t\r\n\v\f");
        if (b == std::string::npos) continue;
        line = line.substr(b, e - b + 1);
        if (line[0] == '#') continue;
        std::stringstream ss(line);
        std::string token; ss >> token;
        if (token == "v") {
            double x = 0, y = 0, z = 0, w = 1;
            ss >> x >> y >> z >> w;  // a missing w leaves 1 (the extraction fails before touching it)
            pos_pool.push_back(V3{x, y, z} / w);
        } else if (token == "vt") {
            double s = 0, t = 0, w = 0;
            ss >> s >> t >> w;
            st_pool.push_back(V2{s, 1 - t});
        } else if (token == "vn") {
            double x = 0, y = 0, z = 0;
            ss >> x >> y >> z;
            nor_pool.push_back(normalize(V3{x, y, z}));
        } else if (token == "f") {
            std::string t0, t1, t2, t3, t4;
            ss >> t0 >> t1 >> t2;
            int f0[3], f1[3], f2[3];
            split_face(t0, f0); split_face(t1, f1); split_face(t2, f2);
            int32_t a = vertex_id(f0), bb = vertex_id(f1), c = vertex_id(f2);
            idx.insert(idx.end(), {a, bb, c});
            if (ss >> t3) {
                int f3[3]; split_face(t3, f3);
                int32_t d = vertex_id(f3);
                idx.insert(idx.end(), {a, c, d});
            }
            if (ss >> t4) throw LjError(LJ_ERR_PARSE, "The object file contains n-gon (n>4) that we do not support.");
        }
    }
    if (nor.empty()) {
        // Nelson Max, "Computing Vertex Normals from Facet Normals" (parse_obj.cpp:57-92)
        nor.assign(pos.size(), V3{0, 0, 0});
        for (size_t t = 0; t + 2 < idx.size(); t += 3) {
            V3 n{0, 0, 0};
            for (int i = 0; i < 3; i++) {
                const V3 &v0 = pos[idx[t + i]], &v1 = pos[idx[t + (i + 1) % 3]], &v2 = pos[idx[t + (i + 2) % 3]];
                V3 side1 = v1 - v0, side2 = v2 - v0;
                if (i == 0) {
                    n = cross(side1, side2);
                    double l = length(n);
                    if (l == 0) break;
                    n = n / l;
                }
                double angle = unit_angle(normalize(side1), normalize(side2));
                nor[idx[t + i]] = nor[idx[t + i]] + n * angle;
            }
        }
        for (auto &n : nor) { double l = length(n); n = (l != 0) ? n / l : V3{0, 0, 0}; }
    }
    return append_mesh(hs, pos, nor, st, idx);
}

LjShape load_serialized_mesh(HostScene &hs, const std::string &filename, int shape_index, const M4 &to_world) {
    std::ifstream fs(filename, std::ios::binary);
    if (!fs.is_open()) throw LjError(LJ_ERR_IO, "Unable to open the serialized file: " + filename);
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(fs)), std::istreambuf_iterator<char>());
    if (file.size() < 8) throw LjError(LJ_ERR_PARSE, "serialized file too short: " + filename);
    auto rd16 = [&](size_t o) { return (uint16_t)(file[o] | (file[o + 1] << 8)); };
    auto rd32 = [&](size_t o) { uint32_t v; memcpy(&v, &file[o], 4); return v; };
    auto rd64 = [&](size_t o) { uint64_t v; memcpy(&v, &file[o], 8); return v; };
    uint16_t version = rd16(2);
    if (version != 3 && version != 4) throw LjError(LJ_ERR_PARSE, "unsupported .serialized version in " + filename);
    size_t offset = 4;  // magic + version
    if (shape_index > 0) {  // end-of-file dictionary (load_serialized.cpp:103-121)
        uint32_t count = rd32(file.size() - 4);
        if ((uint32_t)shape_index >= count) throw LjError(LJ_ERR_PARSE, "shapeIndex out of range in " + filename);
        // (the dictionary — one offset per sub-mesh, then the count — has to lie inside the file)
        const unsigned long long back = version == 4 ? 8ull * (count - (uint32_t)shape_index) + 4ull : 4ull * ((unsigned long long)(count - (uint32_t)shape_index) + 1ull);
        if (back + 4ull > file.size()) throw LjError(LJ_ERR_PARSE, "sub-mesh dictionary reaches outside the file: " + filename);
        const unsigned long long sub = version == 4 ? rd64(file.size() - (size_t)back) : rd32(file.size() - (size_t)back);
        if (sub >= file.size()) throw LjError(LJ_ERR_PARSE, "bad sub-mesh offset in " + filename);
        offset = (size_t)sub + 4;
    }
    if (offset >= file.size()) throw LjError(LJ_ERR_PARSE, "bad sub-mesh offset in " + filename);

    z_stream zs{};
    if (inflateInit2(&zs, 15) != Z_OK) throw LjError(LJ_ERR_INTERNAL, "Could not initialize ZLIB");
    zs.next_in = file.data() + offset;
    zs.avail_in = (uInt)std::min<size_t>(file.size() - offset, 0xFFFFFFFFu);
    auto zread = [&](void *ptr, size_t size) {
        zs.next_out = (Bytef *)ptr; zs.avail_out = (uInt)size;
        while (zs.avail_out > 0) {
            int r = inflate(&zs, Z_NO_FLUSH);
            if (r == Z_STREAM_END && zs.avail_out > 0) { inflateEnd(&zs); throw LjError(LJ_ERR_PARSE, "inflate(): attempting to read past the end of the stream!"); }
            if (r != Z_OK && r != Z_STREAM_END) { inflateEnd(&zs); throw LjError(LJ_ERR_PARSE, "inflate(): data error in " + filename); }
        }
    };
    enum { HasNormals = 0x0001, HasTexcoords = 0x0002, HasColors = 0x0008, DoublePrecision = 0x2000 };
    uint32_t flags; zread(&flags, 4);
    if (version == 4) { char c; do { zread(&c, 1); } while (c != '\0'); }
    uint64_t vertex_count, triangle_count;
    zread(&vertex_count, 8); zread(&triangle_count, 8);
    // (deflate expands at most ~1030-fold: counts the rest of the file could not hold are a damaged header, not a reason to allocate gigabytes)
    const unsigned long long can_hold = (unsigned long long)(file.size() - offset) * 1100ull + 4096ull;
    if (vertex_count > (1ull << 31) || triangle_count > (1ull << 31) || vertex_count * 12ull > can_hold || triangle_count * 12ull > can_hold) { inflateEnd(&zs); throw LjError(LJ_ERR_PARSE, "implausible mesh size in " + filename); }
    bool dbl = flags & DoublePrecision;
    auto read_reals = [&](size_t n) {
        std::vector<double> out(n);
        if (dbl) zread(out.data(), n * 8);
        else { std::vector<float> tmp(n); zread(tmp.data(), n * 4); for (size_t i = 0; i < n; i++) out[i] = tmp[i]; }
        return out;
    };
    std::vector<V3> pos(vertex_count), nor; std::vector<V2> uv;
    {
        auto p = read_reals(vertex_count * 3);
        for (size_t i = 0; i < vertex_count; i++) pos[i] = xform_point(to_world, V3{p[3 * i], p[3 * i + 1], p[3 * i + 2]});
    }
    if (flags & HasNormals) {
        auto p = read_reals(vertex_count * 3);
        M4 inv = inverse(to_world);
        nor.resize(vertex_count);
        for (size_t i = 0; i < vertex_count; i++) nor[i] = xform_normal(inv, V3{p[3 * i], p[3 * i + 1], p[3 * i + 2]});
    }
    if (flags & HasTexcoords) {
        auto p = read_reals(vertex_count * 2);
        uv.resize(vertex_count);
        for (size_t i = 0; i < vertex_count; i++) uv[i] = V2{p[2 * i], p[2 * i + 1]};
    }
    if (flags & HasColors) read_reals(vertex_count * 3);
    std::vector<int32_t> idx(triangle_count * 3);
    zread(idx.data(), idx.size() * 4);
    inflateEnd(&zs);
    return append_mesh(hs, pos, nor, uv, idx);
}

// load_volume_from_file<Spectrum> (volume.cpp:6-104): Mitsuba's gridvolume file — "VOL", version 3, type 1 (float32),
// resolution, channel count (1 or 3), bounding box as six floats, then the voxels, x fastest.  A one-channel file is
// replicated to three (volume.cpp:84-87); max_data is the per-channel maximum, starting from zero.
void load_grid_volume(const std::string &filename, LjVolume &v, std::vector<float> &data) {
    std::ifstream f(filename, std::ios::binary);
    if (!f) throw LjError(LJ_ERR_IO, "cannot open volume: " + filename);
    char header[4];
    f.read(header, 4);
    if (!f || header[0] != 'V' || header[1] != 'O' || header[2] != 'L' || header[3] != 3)
        throw LjError(LJ_ERR_PARSE, "Error loading volume from a file (incorrect header). Filename:" + filename);
    int32_t type = 0, res[3] = {0, 0, 0}, channels = 0;
    f.read((char *)&type, 4);
    if (type != 1) throw LjError(LJ_ERR_UNSUPPORTED, "Unsupported volume format (only support Float32). Filename:" + filename);
    f.read((char *)res, 12); f.read((char *)&channels, 4);
    if (channels != 1 && channels != 3) throw LjError(LJ_ERR_UNSUPPORTED, "Unsupported volume format (wrong number of channels). Filename:" + filename);
    float box[6];
    f.read((char *)box, 24);
    if (!f || res[0] <= 0 || res[1] <= 0 || res[2] <= 0) throw LjError(LJ_ERR_PARSE, "malformed volume header: " + filename);
    const size_t n = (size_t)res[0] * res[1] * res[2];
    {   // a damaged header must not make us allocate terabytes: the grid may be larger than the file (missing voxels read as zero, below),
        // but not out of all proportion to it
        const std::streampos at = f.tellg(); f.seekg(0, std::ios::end); const unsigned long long bytes = (unsigned long long)f.tellg(); f.seekg(at);
        if ((unsigned long long)res[0] * (unsigned long long)res[1] > (1ull << 40) || n > (1ull << 31) || (unsigned long long)n * channels * 4ull > std::max<unsigned long long>(64ull << 20, 64ull * bytes))
            throw LjError(LJ_ERR_PARSE, "volume resolution in the header does not fit the file: " + filename);
    }
    std::vector<float> raw(n * channels, 0.0f);
    f.read((char *)raw.data(), (std::streamsize)(raw.size() * 4));   // a short file leaves zeros, as the reference's read does
    v = LjVolume{};
    v.kind = LJ_VOLUME_GRID; v.scale = 1;
    for (int k = 0; k < 3; k++) { v.resolution[k] = res[k]; v.p_min[k] = box[k]; v.p_max[k] = box[3 + k]; v.max_data[k] = 0; }
    data.resize(n * 3);
    for (size_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            const float x = channels == 1 ? raw[i] : raw[3 * i + k];
            data[3 * i + k] = x;
            if ((double)x > v.max_data[k]) v.max_data[k] = x;
        }
    v.data = data.data();
}

} // namespace lj
