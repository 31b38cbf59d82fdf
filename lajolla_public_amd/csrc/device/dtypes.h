// Device-resident scene and wavefront queue layouts (plain PODs shared by host flattening code and HIP kernels).
// See DESIGN.md §3 for the HBM layout rationale.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define LJ_HD __host__ __device__ __forceinline__
#else
#define LJ_HD inline
#endif

namespace ljd {

// ---- BVH4: the boxes of all four children live in the parent, structure-of-arrays by axis, so one traversal step is
// one 128-byte line and the ray's direction signs pick the "near" and "far" quarter of every axis by address alone.
// child >= 0: inner node index.  child < 0: leaf; ~child = first * 8 + (count - 1) addresses prims
// [first, first + count) of the leaf-ordered prim array (count <= 8).
// An empty slot has lo = +inf, hi = -inf: with sign-selected slabs its entry distance is +inf and its exit -inf.
struct DNode4 {
    float lox[4], loy[4], loz[4];   // quarters 0-2: lower corners of the four children, one axis per 16-byte quarter
    float hix[4], hiy[4], hiz[4];   // quarters 3-5: upper corners
    int32_t child[4];               // quarter 6: >= 0 inner node index, < 0 leaf code ~(first * 8 + count - 1)
    int32_t pad[4];                 // quarter 7: keeps a node on exactly one 128-byte line
};
static_assert(sizeof(DNode4) == 128, "DNode4 must be 128 bytes");

// ---- BVH8 with quantised child boxes (80 bytes = five 16-byte quarters), for trees beyond the LDS image.  A traversal of such a
// tree is bound by the bytes each step gathers through the texture-address path and by the number of dependent fetches per ray, not
// by arithmetic: eight children per step make the chain ~0.6x as long as a BVH4's, 8-bit boxes on a per-node grid
// (origin p, per-axis scale 2^(e-127): plane = p + q * 2^(e-127), lower planes rounded down, upper planes up — the quantised box always
// contains the child's box) make a step 80 bytes instead of 112.  Children sit in octant-ordered slots: slot s holds the child that lies
// towards corner s (bit k set: the high side of axis k) of the node, so that a ray whose direction signs are `oct` (bit k set: negative
// along axis k) meets the slots roughly front to back in ascending (s ^ oct) — no distance sort (Ylitie, Karras, Laine: "Efficient
// Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs", HPG 2017, whose node this follows).
// Inner children are consecutive nodes in slot order from child_base; the primitives of the leaf children are consecutive in slot order
// from prim_base (at most 8 leaves x 4 primitives: a 5-bit offset).
struct DNode8 {
    float p[3];                 // quarter 0: grid origin = lower corner of the union of the children
    uint8_t e[3];               //            biased exponents of the grid steps, one per axis
    uint8_t imask;              //            slots that hold inner nodes
    uint32_t child_base;        // quarter 1: node index of the first inner child
    uint32_t prim_base;         //            leaf-ordered primitive index of the first leaf child's first primitive
    uint8_t meta[8];            //            per slot: leaf = 0x80 | (count - 1) << 5 | offset from prim_base; inner or empty = 0
    uint8_t qlo_x[8], qlo_y[8]; // quarter 2
    uint8_t qlo_z[8], qhi_x[8]; // quarter 3
    uint8_t qhi_y[8], qhi_z[8]; // quarter 4    (an empty slot has qlo = 255, qhi = 0 and is in neither imask nor meta)
};
static_assert(sizeof(DNode8) == 80, "DNode8 must be 80 bytes");

// Leaf-ordered primitive for the intersection tests (48 bytes = three float4).
// kind 0: triangle, float vertices exactly as the reference hands them to Embree (triangle_mesh.inl:11-14).
// kind 1: sphere, `gprim` indexes DScene::prims and the sphere parameters are read from there (double maths).
struct DPrim {
    float v0[3]; int32_t gprim;
    float v1[3]; int32_t kind;
    float v2[3]; int32_t sphere_slot;  // kind 1: index into DScene::spheres
};
static_assert(sizeof(DPrim) == 48, "DPrim must be 48 bytes");

// Per-primitive shading record, indexed by global primitive id (shape order, then triangle order).
// Everything compute_shading_info (triangle_mesh.inl:65-157 / sphere.inl:235-260) derives from per-primitive
// constants is precomputed on the host in double and narrowed once.
struct DPrimShade {
    float n0[3], n1[3], n2[3];  // vertex normals (triangles with normals); sphere: centre in n0, radius in n1[0]
    float uv0[2], uv1[2], uv2[2];
    float dpdu[3];              // dp/du of the triangle (or Frisvad tangent of Ng when the uv map is degenerate)
    float gn[3];                // normalize(Ng), Ng = (p1-p0)x(p2-p0) on the float vertices
    float inv_uv_size;          // max(|dpdu|, |dpdv|)
    int32_t shape_id, prim_id;  // what intersect() reports (intersection.cpp:42-43)
    int32_t material_id, light_id;
    int32_t flags;              // bit0: sphere, bit1: has vertex normals
    int32_t sphere_slot;
};
static_assert(sizeof(DPrimShade) == 112, "DPrimShade layout");

struct DSphere { double center[3]; double radius; int32_t gprim, _pad; };   // gprim: the sphere's global primitive id

// One leaf of a tiny scene's flat leaf table (dscan.h): padded box + its primitives [first, first + count) of the leaf-ordered
// primitive array.  32 bytes: one s_load_dwordx8.
struct DScanLeaf { float lo[3], hi[3]; int32_t first, count; };
static_assert(sizeof(DScanLeaf) == 32, "DScanLeaf must be 32 bytes");

struct DTexture {
    int32_t kind, texture_id;
    float value[3], color1[3];
    float uscale, vscale, uoffset, voffset;
};
struct DMaterial {
    int32_t kind, n_tex;
    float eta, _pad;
    DTexture tex[12];
};

// Emissive triangle for sample_point_on_shape (triangle_mesh.inl:24-38)
struct DLightTri { float v0[3], e1[3], e2[3], n[3]; };
struct DLight {
    int32_t kind, shape_id;
    float intensity[3];
    int32_t is_sphere;
    float center[3], radius;          // sphere lights
    int32_t tri_first, tri_count;     // into light_tris / light_tri_cdf (cdf has tri_count+1 entries starting at tri_first + light index)
    int32_t cdf_first;
    float total_area;
    float pmf;                        // light_pmf(scene, id)
    // envmap
    float to_world[9], to_local[9];   // upper-left 3x3, row-major
    float scale;
    int32_t env_w, env_h, env_cdf_rows, env_pdf_rows, env_cdf_marg, env_pdf_marg;  // offsets into env_tables
    int32_t env_guide_rows, env_guide_marg;   // guide tables of the two cdf searches (dshade.h sample_cdf_guided): env_h x env_w and env_h entries, bit patterns
    DTexture values;
};

struct DMipLevel { int32_t w, h; int64_t offset; };  // offset in floats into texels
struct DImage { int32_t levels, channels; DMipLevel lv[8]; };

struct DCamera {
    float sample_to_cam[16];
    float cam_to_world[16];
    float org[3];
    int32_t width, height, filter_kind;
    float filter_param;
};

// ---- participating media (medium.h:10-21, volume.h:13-30); grid voxels live in DScene::volume_data, 3 floats each
struct DVolume {
    int32_t kind;                    // 0 constant, 1 grid
    int32_t res[3];
    float value[3];                  // constant (medium scale applied)
    float p_min[3], p_max[3], max_data[3];
    float scale;
    int64_t offset;                  // of the first voxel, in floats
    int32_t mono, _pad;              // 1: the grid's three channels are equal everywhere and ONE float per voxel is stored (a third of the gathers)
};
struct DMedium {
    int32_t kind, phase_kind;        // 0 homogeneous / 1 heterogeneous; 0 isotropic / 1 Henyey-Greenstein
    float g;
    float sigma_a[3], sigma_s[3];
    DVolume albedo, density;
};

struct DScene {
    DCamera cam;
    const DNode4 *nodes; int32_t n_nodes;
    const DNode8 *nodes8; int32_t n_nodes8;   // the same binary tree collapsed eight wide (same leaves, same leaf-ordered primitives)
    int32_t node8_stride;                     // bytes between two nodes of nodes8 on the device (80, or 128: one node per cache line)
    const DPrim *leaf_prims; int32_t n_prims;
    const DPrimShade *prims;
    const DSphere *spheres;          // indexed by sphere_slot
    int32_t n_spheres;
    const DMaterial *materials; int32_t n_materials;
    const DLight *lights; int32_t n_lights;
    const float *light_cdf;          // n_lights + 1
    const DLightTri *light_tris;
    const float *light_tri_cdf;
    const DImage *images3, *images1;
    const float *texels;
    const float *env_tables;
    // the environment map's marginal tables (cdf, pdf and the cdf's guide: 3 h + 1 floats, contiguous from env_tables[env_marg_first]) are
    // read through this pointer, which the shade kernels redirect to an LDS copy: offsets into env_tables minus env_marg_first index it
    const float *env_marg; int32_t env_marg_first, env_marg_count;
    int32_t n_images3, n_images1;
    int32_t envmap_light_id;
    int32_t max_depth, rr_depth;
    float eps;                       // get_shadow_epsilon == get_intersection_epsilon (scene.h:99-105)
    float init_spread;               // 0.25 / max(w, h)  (ray.h:35-37)
    // volumetric path tracer only (dvol.h)
    const DMedium *media; int32_t n_media;
    const float *volume_data;
    const int32_t *shape_media;      // per shape: interior, exterior medium id (-1: none)
    int32_t cam_medium, max_null_collisions;
    int32_t has_heterogeneous_medium;   // some medium is a grid volume (picks the k_volpath instantiation)
    int32_t vol_path_version;           // RenderOptions::vol_path_version (render.cpp:111-123): 1 and 2 are estimators of their own
    // tiny scenes only (mega.hip): the flat leaf table; n_scan_leaves (0 when the scene has none) is a multiple of 4, the first n_scan_used
    // entries are leaves, the rest padding that is never entered
    const DScanLeaf *scan_leaves; int32_t n_scan_leaves, n_scan_used;
};

// ---- wavefront path queue: one slot per in-flight path, stored as eight arrays of 16-byte records so that every
// access is a full-width dwordx4 load/store (1 KiB per wave instruction) and each kernel touches only the
// records it needs.
struct alignas(16) Rec4 { float x, y, z, w; };
struct DQueue {
    Rec4 *ro;   // ray origin xyz = position of the vertex the path left | w: tfar of the pending NEE shadow ray (<= 0: none)
    Rec4 *rd;   // extension-ray direction xyz                          | w: flags (bit pattern, see PF_*)
    Rec4 *rs;   // pending NEE shadow-ray direction xyz                 | w: unused
    Rec4 *rh;   // hit of the extension ray: t, u, v                    | w: code = (global prim id + 1) | HIT_VIS_BIT
    Rec4 *rw;   // W = throughput * f / p2(solid angle)                 | w: rr, the Russian-roulette survival probability
    Rec4 *rl;   // accumulated radiance                                 | w: eta_scale (path_tracing.h:53)
    Rec4 *rn;   // pending NEE contribution throughput * C1 * w1        | w: ray_diff.spread
    Rec4 *rg;   // x: sample id, y/z: pcg32 state (lo, hi)  (bit patterns) | w: p2, solid-angle pdf of the sampled direction (< 0: camera ray)
};
// bytes per slot: 128.  extend reads ro, rd, rs (48 B) and writes rh (16 B); shade reads ro, rd, rh, rw, rl, rn, rg
// (112 B) and writes ro, rd, rs, rw, rl, rn, rg (112 B) for every surviving path.

// flags: bits 0-15 num_vertices of the iteration that sampled the ray (2 on camera rays); bit 16: Russian roulette
// said stop (finish after the hit accounting); bit 17: no extension ray (only the pending NEE remains)
enum : uint32_t { PF_DYING = 1u << 16, PF_NO_EXT = 1u << 17 };
enum : int32_t { HIT_VIS_BIT = 1 << 30 };

// Per-workgroup bookkeeping.  Workgroup b owns queue slots [b * seg, (b + 1) * seg) — of which the first `count` hold
// live paths — and the camera samples [next_sample, end_sample).  The shade kernel compacts its survivors to the
// front of the segment (stable, in place) and appends the workgroup's next camera samples behind them, so every
// segment stays dense and full without any grid-wide atomic.  Only workgroup b touches entry b.
struct DBlockState {
    uint32_t next_sample, end_sample;   // sample ids inside the pass
    uint32_t count;                     // live paths at the front of the segment
    uint32_t _pad;
    unsigned long long bounce_iterations, rays_closest, rays_shadow, samples_done, path_steps;
};

// Division of a 32-bit number by a divisor that is fixed for a whole launch (samples per pixel, image width), as a multiplication
// (Granlund & Montgomery 1994, fig. 4.1): q = (t + ((n - t) >> s1)) >> s2 with t = mulhi(m, n) — exact for every n < 2^32 and every
// d >= 1; five instructions where the general division takes ~30.
struct DFastDiv { uint32_t m, s1, s2, d; };
inline DFastDiv make_fast_div(uint32_t d) {
    DFastDiv f; uint32_t l = 0;
    while (l < 32 && (1ull << l) < (unsigned long long)d) l++;   // l = ceil(log2 d)
    f.m = (uint32_t)((((1ull << l) - d) << 32) / d) + 1u;
    f.s1 = l < 1u ? l : 1u; f.s2 = l > 1u ? l - 1u : 0u; f.d = d;
    return f;
}

struct DPass {
    const uint32_t *pixel_list;  // linear pixel index (y*w+x) of the p-th rendered pixel
    uint32_t n_pixels, spp;
    DFastDiv by_spp, by_width;   // (set_pass_divisors)
    uint64_t seed;
    float *sample_rgb;           // 3 floats per sample of the pass: per-sample radiance (written once per sample)
};
inline void set_pass_divisors(DPass &p, uint32_t spp, uint32_t width) { p.spp = spp; p.by_spp = make_fast_div(spp); p.by_width = make_fast_div(width); }

} // namespace ljd
