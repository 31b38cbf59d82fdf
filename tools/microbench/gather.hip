// Microbenchmark: cost of fetching one 128-byte BVH node per ray from a table that sits in L2 / MALL, for three
// lane-to-data mappings.  Decides the node layout of the extend kernel.
//   A: lane-per-ray, SoA node: every lane issues 7 x 16 B loads from its own random line          (7 loads / ray)
//   B: quad-per-ray, AoS node: lane k of a quad issues 2 x 16 B loads from child k's 32 B          (2 loads / lane, 4 lanes / ray)
//   C: lane-per-ray, but the 8 quarters of a line are fetched by 8 neighbouring lanes and handed over through LDS
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

__global__ void __launch_bounds__(256, 4) k_a(const char *nodes, uint32_t n_nodes, int iters, float *out) {
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 1u;
    float acc = 0;
    uint32_t idx = lcg(s) % n_nodes;
    for (int it = 0; it < iters; it++) {
        const char *p = nodes + (size_t)idx * 128;
        v4f a = *(const v4f *)(p), b = *(const v4f *)(p + 16), c = *(const v4f *)(p + 32), d = *(const v4f *)(p + 48), e = *(const v4f *)(p + 64), f = *(const v4f *)(p + 80), g = *(const v4f *)(p + 96);
        float r = a.x + b.y + c.z + d.w + e.x + f.y + g.z;
        acc += r;
        idx = (lcg(s) + (uint32_t)(r != 12345.0f)) % n_nodes;   // dependent on the loaded data, like a traversal
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void __launch_bounds__(256, 4) k_b(const char *nodes, uint32_t n_nodes, int iters, float *out) {
    const uint32_t ray = (blockIdx.x * 256 + threadIdx.x) >> 2, sub = threadIdx.x & 3;
    uint32_t s = ray * 2654435761u + 1u;
    float acc = 0;
    uint32_t idx = lcg(s) % n_nodes;
    for (int it = 0; it < iters; it++) {
        const char *p = nodes + (size_t)idx * 128 + sub * 32;
        v4f a = *(const v4f *)(p), b = *(const v4f *)(p + 16);
        float r = a.x + b.y;
        r += __shfl_xor(r, 1, 64); r += __shfl_xor(r, 2, 64);
        acc += r;
        idx = (lcg(s) + (uint32_t)(r != 12345.0f)) % n_nodes;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void __launch_bounds__(256, 4) k_c(const char *nodes, uint32_t n_nodes, int iters, float *out) {
    __shared__ v4f stage[4][64 * 8];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 1u;
    float acc = 0;
    uint32_t idx = lcg(s) % n_nodes;
    for (int it = 0; it < iters; it++) {
        // round j: lanes 8m..8m+7 fetch the eight quarters of the node wanted by lane 8j+m
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t owner = 8 * j + (lane >> 3);
            const uint32_t oi = __shfl(idx, owner, 64);
            const v4f q = *(const v4f *)(nodes + (size_t)oi * 128 + (lane & 7) * 16);
            stage[wave][(lane & 7) * 64 + owner] = q;   // transposed: quarter-major
        }
        const v4f a = stage[wave][lane], b = stage[wave][64 + lane], c = stage[wave][128 + lane], d = stage[wave][192 + lane], e = stage[wave][256 + lane], f = stage[wave][320 + lane], g = stage[wave][384 + lane];
        float r = a.x + b.y + c.z + d.w + e.x + f.y + g.z;
        acc += r;
        idx = (lcg(s) + (uint32_t)(r != 12345.0f)) % n_nodes;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    const uint32_t n_nodes = argc > 1 ? atoi(argv[1]) : 16384;   // 2 MiB table
    const int iters = 256, grid = 256 * 4 * 4;
    char *nodes; float *out;
    CHECK(hipMalloc(&nodes, (size_t)n_nodes * 128)); CHECK(hipMalloc(&out, (size_t)grid * 256 * 4));
    std::vector<float> h((size_t)n_nodes * 32, 1.0f);
    CHECK(hipMemcpy(nodes, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int variant = 0; variant < 3; variant++) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            CHECK(hipEventRecord(e0));
            if (variant == 0) hipLaunchKernelGGL(k_a, dim3(grid), dim3(256), 0, 0, nodes, n_nodes, iters, out);
            if (variant == 1) hipLaunchKernelGGL(k_b, dim3(grid), dim3(256), 0, 0, nodes, n_nodes, iters, out);
            if (variant == 2) hipLaunchKernelGGL(k_c, dim3(grid), dim3(256), 0, 0, nodes, n_nodes, iters, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double rays = (double)grid * 256 / (variant == 1 ? 4 : 1);
        printf("variant %c  nodes %u (%.1f MiB): %.3f ms, %.2f G node-fetches/s\n", 'A' + variant, n_nodes, n_nodes * 128.0 / 1048576, best, rays * iters / best / 1e6);
    }
    return 0;
}
