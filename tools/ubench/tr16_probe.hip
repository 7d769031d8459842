// ds_read_b64_tr_b16 semantics probe (gfx950): hipcc --offload-arch=gfx950 -O2 -o tr16_probe tr16_probe.hip && ./tr16_probe
// Image [k][f] of 16-bit values k*256+f; every 16-lane group reads a 4-row x 16-column block.  Lane 4q+p supplies the address of
// (row q, columns 4p..4p+3); the probe prints what lane i receives: expected column i of the four rows (MI355X guide, T10).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short *out) {
    __shared__ short img[32][160];
    for (int i = threadIdx.x; i < 32 * 160; i += 64) (&img[0][0])[i] = (short)((i / 160) * 256 + (i % 160));
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const short *addr = &img[8 * (g >> 1) + q][16 * (g & 1) + 4 * p];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)addr);
    *reinterpret_cast<s16x4 *>(out + 4 * lane) = v;
}
int main() {
    short *d; hipMalloc(&d, 64 * 4 * sizeof(short));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; lane++) {
        const int g = lane >> 4, i = lane & 15;
        for (int e = 0; e < 4; e++) {
            const int want = (8 * (g >> 1) + e) * 256 + 16 * (g & 1) + i;
            if (h[4 * lane + e] != want) { if (bad < 8) printf("lane %d elem %d: got (k %d, f %d) want (k %d, f %d)\n", lane, e, h[4 * lane + e] / 256, h[4 * lane + e] % 256, want / 256, want % 256); bad++; }
        }
    }
    printf("tr16 probe: %s (%d mismatches)\n", bad ? "DIFFERENT from the documented mapping" : "matches the documented mapping", bad);
    return bad != 0;
}
