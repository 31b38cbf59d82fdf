// Host-side scene container behind `lj_host_scene`: owns the storage an LjSceneDesc points into.
// Mirrors what parse_scene() hands to Scene::Scene in the reference (parse_scene.cpp:1122-1131).
#pragma once
#include "../../../include/lajolla_hip.h"
#include "hmath.h"
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace lj {

struct LjError : std::runtime_error {
    int code;
    LjError(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};

// A decoder's sanity check of the size a file header claims, before anything is allocated for it: at most 2^24 texels a side (the
// reference loader's limit), 2^28 in all, and not more than a file of this length could possibly encode (no format packs 8192 texels a byte).
inline void check_image_size(long long w, long long h, size_t file_bytes, const std::string &name) {
    if (w <= 0 || h <= 0 || w > (1 << 24) || h > (1 << 24) || w * h > (1ll << 28) || (unsigned long long)(w * h) > (unsigned long long)file_bytes * 8192ull + 65536ull)
        throw LjError(LJ_ERR_PARSE, "image size in the header does not fit the file (" + std::to_string(w) + " x " + std::to_string(h) + "): " + name);
}

struct HostImage {
    int width = 0, height = 0, channels = 0;
    std::vector<float> data;
};

struct HostScene {
    LjCamera camera{};
    LjRenderOptions options{};
    std::vector<LjShape> shapes;
    std::vector<LjMaterial> materials;
    std::vector<LjLight> lights;
    std::vector<LjMedium> media;                       // LjVolume::data point into volume_data (set by finalize)
    std::vector<std::vector<float>> volume_data;       // one entry per grid volume, 3 floats per voxel
    std::vector<std::pair<int, int>> volume_owner;     // (medium index, 0 = albedo / 1 = density) of each volume_data entry
    std::vector<HostImage> images3, images1;
    std::map<std::string, int> image3s_map, image1s_map;  // TexturePool maps (texture.h:13-19)
    std::vector<double> positions, normals, uvs;
    std::vector<int32_t> indices;
    int envmap_light_id = -1;
    std::string output_filename = "image.exr";

    // filled by finalize()
    std::vector<LjImage> image3_views, image1_views;
    LjSceneDesc desc{};
    void finalize();
};

// deep copy of a caller-provided description
HostScene *host_scene_from_desc(const LjSceneDesc &d);

// scene_xml.cpp
HostScene *parse_scene_xml(const std::string &path);

// mesh_io.cpp — both append one TriangleMesh to the pools and return its LjShape (ids unset)
LjShape load_obj_mesh(HostScene &hs, const std::string &filename, const M4 &to_world);
LjShape load_serialized_mesh(HostScene &hs, const std::string &filename, int shape_index, const M4 &to_world);

// mesh_io.cpp — load_volume_from_file<Spectrum> (volume.cpp:6-104): Mitsuba gridvolume, float32, 1 or 3 channels
void load_grid_volume(const std::string &filename, LjVolume &v, std::vector<float> &data);

// image_io.cpp — imread3 / imread1 (image.cpp:28-133)
HostImage read_image(const std::string &filename, int channels);
// imwrite (image.cpp:135-173): .pfm (the reference's top-down layout) or .exr (HALF, scan-line)
void write_image(const std::string &filename, int width, int height, const float *rgb);

} // namespace lj
