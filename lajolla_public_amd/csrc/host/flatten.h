// LjSceneDesc -> flat, pointer-free arrays in the device layouts of device/dtypes.h.
// This is the host half of the reference's Scene::Scene (scene.cpp:3-53): bounds sphere, per-mesh area tables,
// envmap table, light table — plus the BVH that replaces rtcCommitScene.  Pure C++ (no HIP), so the CPU test
// suite can check the flattened tables against the golden vectors without a GPU.
#pragma once
#include "../device/dtypes.h"
#include "host_scene.h"
#include <vector>

namespace lj {

struct FlatScene {
    ljd::DCamera cam{};
    std::vector<ljd::DNode4> nodes;
    std::vector<ljd::DNode8> nodes8;   // the same tree eight wide with quantised boxes (trees beyond the extend kernel's LDS image)
    std::vector<ljd::DPrim> leaf_prims;
    std::vector<ljd::DPrimShade> prims;
    std::vector<ljd::DSphere> spheres;
    std::vector<ljd::DMaterial> materials;
    std::vector<ljd::DLight> lights;
    std::vector<float> light_cdf;
    std::vector<ljd::DLightTri> light_tris;
    std::vector<float> light_tri_cdf;
    std::vector<ljd::DImage> images3, images1;
    std::vector<float> texels;
    std::vector<float> env_tables;
    int env_marg_first = 0, env_marg_count = 0;   // the environment map's marginal tables inside env_tables (dtypes.h DScene::env_marg)
    std::vector<ljd::DMedium> media;
    std::vector<float> volume_data;
    std::vector<int32_t> shape_media;
    int n_scan_used = 0;                       // leaves among scan_leaves (the rest pads the table to a multiple of four)
    std::vector<ljd::DScanLeaf> scan_leaves;   // tiny scenes only (else empty): the flat leaf table of device/dscan.h
    int cam_medium = -1, max_null_collisions = 1000, vol_path_version = 0;
    int envmap_light_id = -1, max_depth = -1, rr_depth = 5, spp = 4, integrator = LJ_INTEGRATOR_PATH;
    int bvh_depth = 0, bvh8_depth = 0;
    int64_t n_triangles = 0, n_spheres = 0;
    double bounds_radius = 0, bounds_center[3] = {0, 0, 0}, shadow_epsilon = 0;
    // double-precision tables kept for inspection by tests (scene.cpp:47-52)
    std::vector<double> light_pmf_d, light_cdf_d, light_power_d;
    // a DScene whose pointers refer to the vectors above (host memory)
    ljd::DScene host_view() const;
};

// Throws LjError(LJ_ERR_UNSUPPORTED) for variant alternatives the device path does not implement.
FlatScene flatten_scene(const LjSceneDesc &d);

// bvh.cpp — binned-SAH tree with spatial splits (SBVH) over padded float boxes, collapsed to a BVH4; fills nodes (breadth-first)
// and leaf order.
// depth_out = number of inner (wide) levels; a traversal needs at most 3 * depth_out stack entries.
// `tri`: the primitive is the triangle v[0..2] (float vertices as the device tests them) — what a spatial split clips; else
// only its box is known (spheres).  leaf_order lists the primitives of the leaves one leaf after the other; with spatial
// splits a primitive that straddles a split plane is referenced by a leaf on either side, so the list may be longer than `prims`.
struct BuildPrim { float lo[3], hi[3]; float v[3][3]; int tri; };
// nodes8 / depth8_out: the same binary tree collapsed eight wide (DNode8); a traversal of it holds at most depth8_out node groups.
void build_bvh(const std::vector<BuildPrim> &prims, int max_leaf, int max_depth,
               std::vector<ljd::DNode4> &nodes, std::vector<ljd::DNode8> &nodes8, std::vector<int> &leaf_order, int &depth_out, int &depth8_out);

} // namespace lj
