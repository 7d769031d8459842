// Micro-benchmark: cost of s_barrier hand-overs between the four waves of a workgroup (one per SIMD), as k_step uses them:
// (a) bare barriers, (b) producer wave writes LDS -> barrier -> consumer waves read/compute/write -> barrier -> producer reads,
// (c) the same hand-over with LDS flags polled instead of barriers.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_bare(float *out, int iters) {
    float v = threadIdx.x;
    for (int i = 0; i < iters; i++) { __syncthreads(); v += 1.0f; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
__global__ void k_handover(float *out, int iters) {
    __shared__ float x[64], y[3][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float v = lane;
    for (int i = 0; i < iters; i++) {
        if (wave == 0) x[lane] = v;
        __syncthreads();
        if (wave > 0) y[wave - 1][lane] = x[lane] * 1.0001f + wave;
        __syncthreads();
        if (wave == 0) v = y[0][lane] + y[1][lane] + y[2][lane];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
__global__ void k_flags(float *out, int iters) {
    __shared__ float x[64], y[3][64];
    __shared__ volatile int fx, fy[3];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) { fx = 0; fy[0] = fy[1] = fy[2] = 0; }
    __syncthreads();
    float v = lane;
    for (int i = 1; i <= iters; i++) {
        if (wave == 0) {
            x[lane] = v;
            __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): data written before the flag
            if (lane == 0) fx = i;
            while (fy[0] < i || fy[1] < i || fy[2] < i) {}
            v = y[0][lane] + y[1][lane] + y[2][lane];
        } else {
            while (fx < i) {}
            y[wave - 1][lane] = x[lane] * 1.0001f + wave;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (lane == 0) fy[wave - 1] = i;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
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
    const int iters = 20000;
    double a = time_ms([&] { k_bare<<<256, 256>>>(out, iters); });
    double b = time_ms([&] { k_handover<<<256, 256>>>(out, iters); });
    double c = time_ms([&] { k_flags<<<256, 256>>>(out, iters); });
    printf("4 waves/workgroup, 256 workgroups: bare s_barrier %.1f ns; LDS hand-over round trip with 2 barriers %.1f ns; with polled LDS flags %.1f ns\n",
           a * 1e6 / iters, b * 1e6 / iters, c * 1e6 / iters);
    return 0;
}
