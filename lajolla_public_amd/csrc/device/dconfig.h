// Launch plans decided at scene upload (kernels.hip, mega.hip) and carried by the scene object (api_internal.h).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace ljd {

// feature-set instantiations of the shading code (dshade.h lists them, smallest first)
constexpr int kNumShadeVariants = 6;
constexpr int kShadeVariantAll = kNumShadeVariants - 1;

// LDS image of the extend kernel: stack levels, staged nodes / primitives, which instantiation
struct ExtendConfig {
    int stack; int spill_levels; int lds_nodes; int lds_prims; int resident; int spheres; size_t smem; uint32_t refill_min, min_descending;
    // trees beyond the LDS image: the BVH8 kernels (k_extend8); group-stack levels in LDS, staged nodes, bytes of that image
    int wide; int stack8; int lds_nodes8; size_t smem8;
};
// LDS staging plan of the shade kernel (sizes rounded up to 16 bytes) and the feature-set instantiation (dshade.h)
struct ShadeConfig { uint32_t prims_bytes, materials_bytes, lights_bytes, light_cdf_bytes, light_tris_bytes, light_tri_cdf_bytes, stage_prims, images3_bytes, images1_bytes, env_marg_bytes; int variant; size_t smem; };

} // namespace ljd
