// Micro-benchmark: issue cadence of a single wave64 on one gfx950 SIMD (what bounds k_step at 4096 envs: one wave per CU).
// Measures ns per VALU instruction for dependent chains vs independent streams, plus transcendental / DPP / LDS costs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int CHAINS> __global__ void k_fma(float *out, int iters, float a, float b) {
    float v[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) v[c] = threadIdx.x * 1e-3f + c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) v[c] = __builtin_fmaf(v[c], a, b);
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += v[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_rcp(float *out, int iters, float a) {
    float v = threadIdx.x + 1.5f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) v = __builtin_amdgcn_rcpf(v) + a;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
__global__ void k_dpp(float *out, int iters, float a) {
    float v = threadIdx.x + 1.5f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) v = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)) * a;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
__global__ void k_lds(float *out, int iters) {
    __shared__ float t[256];
    t[threadIdx.x] = threadIdx.x; t[threadIdx.x + 64] = 1; t[threadIdx.x + 128] = 2; t[threadIdx.x + 192] = 3;
    __syncthreads();
    int idx = threadIdx.x & 3;
    float v = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) { v += t[idx]; idx = (idx + (int)v) & 255; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v + idx;
}
template <class F> static double time_ms(F launch, int reps = 5) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    double best = 1e9;
    for (int r = 0; r < reps; r++) { hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    return best;
}
int main() {
    float *out; CHECK(hipMalloc(&out, 1 << 22));
    const int iters = 20000; const double n = iters * 16.0;
    for (int grid : {1, 256, 1024}) {
        for (int waves : {1, 2, 4}) {
            int block = 64 * waves;
            double d1 = time_ms([&] { k_fma<1><<<grid, block>>>(out, iters, 0.999f, 0.001f); });
            double d2 = time_ms([&] { k_fma<2><<<grid, block>>>(out, iters / 2, 0.999f, 0.001f); });
            double d4 = time_ms([&] { k_fma<4><<<grid, block>>>(out, iters / 4, 0.999f, 0.001f); });
            double d8 = time_ms([&] { k_fma<8><<<grid, block>>>(out, iters / 8, 0.999f, 0.001f); });
            printf("grid %4d block %3d: ns per fma instr: dependent %.3f | 2 chains %.3f | 4 chains %.3f | 8 chains %.3f\n", grid, block,
                   d1 * 1e6 / n, d2 * 1e6 / n, d4 * 1e6 / n, d8 * 1e6 / n);
        }
    }
    double r = time_ms([&] { k_rcp<<<256, 64>>>(out, iters, 0.5f); });
    double p = time_ms([&] { k_dpp<<<256, 64>>>(out, iters, 0.999f); });
    double l = time_ms([&] { k_lds<<<256, 64>>>(out, iters); });
    printf("grid 256 block 64: ns per (rcp+add) %.3f | (dpp mov + mul) %.3f | dependent LDS read + 2 valu %.3f\n", r * 1e6 / n, p * 1e6 / n, l * 1e6 / n);
    return 0;
}
