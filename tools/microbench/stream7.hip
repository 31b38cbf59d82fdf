// Microbenchmark: the shade kernel's memory skeleton — every path-step reads seven 16-byte records and writes seven,
// workgroup b walking its own segment of the queue chunk by chunk — with no shading work in between.
// Gives the HBM-side ceiling for k_shade's 224 B per path-step at the launch geometry the renderer uses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int OCC>
__global__ void __launch_bounds__(256, OCC) k_stream(v4f *q, size_t n_slots, unsigned seg, int sync) {
    const size_t base = (size_t)blockIdx.x * seg;
    for (unsigned c0 = 0; c0 < seg; c0 += 256) {
        const size_t i = base + c0 + threadIdx.x;
        v4f r[7];
#pragma unroll
        for (int k = 0; k < 7; k++) r[k] = q[(size_t)k * n_slots + i];
#pragma unroll
        for (int k = 0; k < 7; k++) r[k] = r[k] * 1.0001f + 1.0f;
        if (sync) __syncthreads();
#pragma unroll
        for (int k = 0; k < 7; k++) q[(size_t)(k + (k >= 2 ? 1 : 0)) * n_slots + i] = r[k];   // writes skip record 2, like shade
    }
}

int main(int argc, char **argv) {
    const unsigned n_blocks = argc > 1 ? atoi(argv[1]) : 2048, seg = argc > 2 ? atoi(argv[2]) : 8192;
    const size_t n_slots = (size_t)n_blocks * seg;
    v4f *q; CHECK(hipMalloc(&q, n_slots * 16 * 8)); CHECK(hipMemset(q, 0, n_slots * 16 * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int sync = 0; sync < 2; sync++) for (int occ = 4; occ <= 8; occ += 4) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            CHECK(hipEventRecord(e0));
            if (occ == 4) hipLaunchKernelGGL(k_stream<4>, dim3(n_blocks), dim3(256), 0, 0, q, n_slots, seg, sync);
            else hipLaunchKernelGGL(k_stream<8>, dim3(n_blocks), dim3(256), 0, 0, q, n_slots, seg, sync);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("blocks %u seg %u occ %d sync %d: %.3f ms for %.1f M path-steps -> %.2f TB/s (224 B per path-step)\n", n_blocks, seg, occ, sync, best, n_slots / 1e6, n_slots * 224.0 / best / 1e9);
    }
    return 0;
}
