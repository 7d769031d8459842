// Micro-benchmark: can one wave overlap v_mfma_f32_32x32x2_f32 with independent VALU work on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int NV> __global__ void k(float *out, int iters, float a, float b) {
    f32x16 acc = {0}, acc2 = {0};
    f32x4 c4 = {0, 0, 0, 0};
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; c++) v[c] = threadIdx.x * 1e-3f + c;
    float x = threadIdx.x * 0.01f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE == 0 || MODE == 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, x, acc, 0, 0, 0);       // dependent chain
            if (MODE == 3) { if (u & 1) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, x, acc, 0, 0, 0); else acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, x, acc2, 0, 0, 0); }
            if (MODE == 4) c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, x, c4, 0, 0, 0);
            if (MODE == 1 || MODE == 2) {
#pragma unroll
                for (int r = 0; r < NV; r++) v[r & 7] = __builtin_fmaf(v[r & 7], a, b);
            }
            if (MODE == 2) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, NV, 0); }
        }
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) s += v[c];
#pragma unroll
    for (int c = 0; c < 16; c++) s += acc[c] + acc2[c];
    s += c4[0] + c4[1] + c4[2] + c4[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F> static double time_ms(F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    double best = 1e9;
    for (int r = 0; r < 5; r++) { hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    return best;
}
int main() {
    float *out; if (hipMalloc(&out, 1 << 22) != hipSuccess) return 1;
    const int iters = 20000; const double n = iters * 8.0;
    double m = time_ms([&] { k<0, 0><<<256, 64>>>(out, iters, 0.5f, 0.1f); });
    double m2 = time_ms([&] { k<3, 0><<<256, 64>>>(out, iters, 0.5f, 0.1f); });
    double m4 = time_ms([&] { k<4, 0><<<256, 64>>>(out, iters, 0.5f, 0.1f); });
    printf("mfma 32x32x2 f32 dependent chain: %.2f ns each; two alternating accumulators: %.2f ns each; 16x16x4 dependent: %.2f ns each\n", m * 1e6 / n, m2 * 1e6 / n, m4 * 1e6 / n);
    double v8 = time_ms([&] { k<1, 8><<<256, 64>>>(out, iters, 0.5f, 0.1f); });
    double v16 = time_ms([&] { k<1, 16><<<256, 64>>>(out, iters, 0.5f, 0.1f); });
    printf("valu only: 8 fma per slot %.2f ns; 16 fma per slot %.2f ns\n", v8 * 1e6 / n, v16 * 1e6 / n);
    double b8 = time_ms([&] { k<2, 8><<<256, 64>>>(out, iters, 0.5f, 0.1f); });
    double b16 = time_ms([&] { k<2, 16><<<256, 64>>>(out, iters, 0.5f, 0.1f); });
    printf("mfma + valu interleaved: 1 mfma + 8 fma %.2f ns per slot; 1 mfma + 16 fma %.2f ns per slot\n", b8 * 1e6 / n, b16 * 1e6 / n);
    return 0;
}
