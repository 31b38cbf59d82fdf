// Device helpers shared by the kernel translation units (kernels.hip: the wavefront kernels; mega.hip: the fused kernel of tiny
// scenes): LDS address-space shorthands, bit casts, a wave-wide sum and the per-workgroup LDS staging of the shading tables.
#pragma once
#include <hip/hip_runtime.h>
#include "dshade.h"

namespace ljd {

constexpr int kBlock = 256;
#define LJ_LDS __attribute__((address_space(3)))
typedef float v4f __attribute__((ext_vector_type(4)));  // builtin vector: assignable across address spaces

__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }

extern __shared__ __attribute__((aligned(16))) v4f lj_smem[];

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// The shading tables every path touches in a data-dependent order (per-primitive shading records, materials, lights
// and their cdfs) are copied to LDS once per workgroup when they fit: a chain of five or six dependent L2-latency
// gathers per path-step becomes LDS-latency reads.  Pointers stay generic, so dshade.h is unchanged.
// (and, when they fit beside those: the image descriptors every texture lookup starts from and the environment map's marginal tables)
struct ShadeStage { uint32_t prims_bytes, materials_bytes, lights_bytes, light_cdf_bytes, light_tris_bytes, light_tri_cdf_bytes, stage_prims, images3_bytes, images1_bytes, env_marg_bytes; };
inline ShadeStage make_shade_stage(const ShadeConfig &c) {
    ShadeStage st;
    st.prims_bytes = c.prims_bytes; st.materials_bytes = c.materials_bytes; st.lights_bytes = c.lights_bytes; st.light_cdf_bytes = c.light_cdf_bytes;
    st.light_tris_bytes = c.light_tris_bytes; st.light_tri_cdf_bytes = c.light_tri_cdf_bytes; st.stage_prims = c.stage_prims;
    st.images3_bytes = c.images3_bytes; st.images1_bytes = c.images1_bytes; st.env_marg_bytes = c.env_marg_bytes;
    return st;
}

__device__ __forceinline__ void lds_copy16(void *dst, const void *src, uint32_t bytes) {
    const v4f *s4 = (const v4f *)src; v4f *d4 = (v4f *)dst;
    for (uint32_t i = threadIdx.x; i < bytes / 16; i += kBlock) d4[i] = s4[i];
}

// LDS staging of the shading tables at byte offset `at` of the dynamic segment; rewrites the pointers of `sc`
// STAGE — what the launch stages: 0 nothing (the tables do not fit, shade_config), 1 materials, lights and their cdfs,
// 2 the per-primitive shading records as well, -1 decided at run time from `stg`.  With a compile-time STAGE the staged
// pointers are LDS pointers by construction, which the compiler sees: their reads become ds_read instead of flat loads
// (flat loads wait on the vector-memory counter together with the queue records).
template <int STAGE>
__device__ __forceinline__ void stage_shade_tables(DScene &sc, const ShadeStage &stg, uint32_t at) {
    char *p = (char *)lj_smem + at;
    if (STAGE == 0 || (STAGE < 0 && stg.materials_bytes == 0u)) return;   // the pointers stay global
    if (STAGE == 2 || (STAGE < 0 && stg.stage_prims)) { lds_copy16(p, sc.prims, stg.prims_bytes); sc.prims = (const DPrimShade *)p; p += stg.prims_bytes; }
    lds_copy16(p, sc.materials, stg.materials_bytes); sc.materials = (const DMaterial *)p; p += stg.materials_bytes;
    lds_copy16(p, sc.lights, stg.lights_bytes); sc.lights = (const DLight *)p; p += stg.lights_bytes;
    lds_copy16(p, sc.light_cdf, stg.light_cdf_bytes); sc.light_cdf = (const float *)p; p += stg.light_cdf_bytes;
    lds_copy16(p, sc.light_tris, stg.light_tris_bytes); sc.light_tris = (const DLightTri *)p; p += stg.light_tris_bytes;
    lds_copy16(p, sc.light_tri_cdf, stg.light_tri_cdf_bytes); sc.light_tri_cdf = (const float *)p; p += stg.light_tri_cdf_bytes;
    if (stg.images3_bytes) { lds_copy16(p, sc.images3, stg.images3_bytes); sc.images3 = (const DImage *)p; p += stg.images3_bytes; }
    if (stg.images1_bytes) { lds_copy16(p, sc.images1, stg.images1_bytes); sc.images1 = (const DImage *)p; p += stg.images1_bytes; }
    if (stg.env_marg_bytes) { lds_copy16(p, sc.env_marg, stg.env_marg_bytes); sc.env_marg = (const float *)p; }
}

} // namespace ljd
