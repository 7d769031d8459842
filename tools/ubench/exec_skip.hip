// Micro-benchmark: does a gfx950 SIMD skip the quarter-wave passes of a VALU instruction whose lanes are all inactive?
// (If it did, two half-filled rigid-body waves on two SIMDs would halve k_step's serial chain.)  ns per instruction of 4 independent
// fma chains (issue-bound) with 64 / 32 / 16 active lanes, by branch (EXEC mask) and by a partial workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_fma(float *out, int iters, float a, float b, int active, unsigned long long mask = ~0ull) {
    float v0 = threadIdx.x * 1e-3f, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
    if ((int)threadIdx.x < active && ((mask >> (threadIdx.x & 63)) & 1)) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 16; u++) { v0 = __builtin_fmaf(v0, a, b); v1 = __builtin_fmaf(v1, a, b); v2 = __builtin_fmaf(v2, a, b); v3 = __builtin_fmaf(v3, a, b); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3;
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
    const int iters = 20000; const double n = iters * 64.0;
    for (int active : {64, 32, 16}) {
        double d = time_ms([&] { k_fma<<<256, 64>>>(out, iters, 0.999f, 0.001f, active); });
        double e = time_ms([&] { k_fma<<<256, active>>>(out, iters, 0.999f, 0.001f, 64); });
        printf("active lanes %2d: ns per fma instr %.3f (EXEC mask in a full wave) | %.3f (workgroup of %d threads)\n", active, d * 1e6 / n, e * 1e6 / n, active);
    }
    for (unsigned long long m : {0xFFFFFFFFFFFFFFFEull, 0x7FFFFFFFFFFFFFFFull, 0x5555555555555555ull, 0x0000FFFFFFFFFFFFull, 0xFFFFFFFF00000000ull, 0x00000000000000FFull}) {
        double d = time_ms([&] { k_fma<<<256, 64>>>(out, iters, 0.999f, 0.001f, 64, m); });
        printf("EXEC %016llx: ns per fma instr %.3f\n", m, d * 1e6 / n);
    }
    return 0;
}
