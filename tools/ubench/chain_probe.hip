// Stand-alone timing of the wide learner's chain forward (csrc/lg_policy.h: k_mlp_chain_fwd64) on one PPO mini-batch of the rough tasks
// (24 576 rows, 235-512-256-128-12 | 1, actor + critic per launch).  Kernel experiments are tried here first (seconds to compile):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fno-hip-fp32-correctly-rounded-divide-sqrt -mllvm -amdgpu-mfma-vgpr-form \
//         -mllvm -amdgpu-spill-vgpr-to-agpr=0 [-DLG_CHAIN_PROF] -o chain_probe chain_probe.hip && ./chain_probe      (the library's flags; -DLG_CHAIN_PROF: phase stamps)
#include "../../legged_games_gym_amd/csrc/lg_policy.h"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static float *dev_random(size_t n, unsigned seed, float scale) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; h[i] = scale * ((int)(s >> 8) / 8388608.0f - 1.0f); }
    float *d; CK(hipMalloc(&d, n * sizeof(float))); CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return d;
}
int main(int argc, char **argv) {
    const int mb = argc > 1 ? atoi(argv[1]) : 24576, d[5] = {235, 512, 256, 128, 12}, k0p = 236, k0s = 15;
    lg::ChainPackArgs pk; memset(&pk, 0, sizeof pk);
    lg::ChainArgs c; memset(&c, 0, sizeof c);
    c.mb = mb;
    for (int n = 0; n < 2; n++) {
        lg::ChainNet &cn = c.net[n];
        cn.x = dev_random((size_t)mb * k0p, 1 + n, 1.0f); cn.ldx = k0p; cn.num_in = d[0];
        for (int l = 0; l < 4; l++) {
            const int ks = l == 0 ? k0s : d[l] / 16, ot = (d[l + 1] + 31) / 32;
            pk.W[n][l] = dev_random((size_t)d[l + 1] * d[l], 10 + 4 * n + l, 0.08f); pk.b[n][l] = dev_random(d[l + 1], 30 + l, 0.1f);
            void *wp; CK(hipMalloc(&wp, (size_t)ot * ks * 2 * 64 * 8 * sizeof(__bf16))); float *bp; CK(hipMalloc(&bp, ot * 32 * sizeof(float)));
            pk.wp[n][l] = (__bf16 *)wp; pk.bp[n][l] = bp;
            pk.in_dim[n][l] = d[l]; pk.out_dim[n][l] = d[l + 1]; pk.KS[n][l] = ks; pk.OT[n][l] = ot;
            cn.wb[l] = (const lg::bf16x8g *)wp; cn.bb[l] = bp;
        }
        for (int l = 0; l < 3; l++) { CK(hipMalloc(&cn.act[l], (size_t)mb * d[l + 1] * sizeof(float))); cn.lda[l] = d[l + 1]; }
        CK(hipMalloc(&cn.out, (size_t)mb * d[4] * sizeof(float))); cn.out_dim = d[4];
    }
    hipLaunchKernelGGL(lg::k_chain_pack, dim3(512, 4, 2), dim3(256), 0, 0, pk);
    CK(hipDeviceSynchronize());
    const dim3 grid((mb + 63) / 64, 2), block(64 * LG_PW_WAVES);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((lg::k_mlp_chain_fwd64<15>), grid, block, 0, 0, c);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 20; r++) hipLaunchKernelGGL((lg::k_mlp_chain_fwd64<15>), grid, block, 0, 0, c);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    std::vector<float> out((size_t)mb * d[4]);
    CK(hipMemcpy(out.data(), c.net[0].out, out.size() * sizeof(float), hipMemcpyDeviceToHost));
    double cs = 0; for (float v : out) cs += v;
    printf("chain forward, %d rows x 2 nets: %.1f us per launch  (checksum %.6f)\n", mb, ms * 1e3 / 20, cs);
#ifdef LG_CHAIN_PROF
    {   // phase stamps of wave 0 of every workgroup of net 0 (s_memtime: 100 MHz), last launch
        std::vector<unsigned long long> st(2048 * 16);
        CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(lg::g_chain_prof), st.size() * sizeof(unsigned long long)));
        const int nb = (mb + 63) / 64;
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int b = 0; b < nb; b++) { if (st[b * 16] < t0) t0 = st[b * 16]; if (st[b * 16 + 12] > t1) t1 = st[b * 16 + 12]; }
        printf("first start -> last end: %.1f us\n", (t1 - t0) * 0.01);
        const char *name[12] = {"x0 load + barrier", "L0 tiles 0-7 MFMA", "epilogue 0", "barrier", "L1 k 0-15 MFMA + barrier", "L0 tiles 8-15 MFMA", "epilogue 0'", "barrier",
                                "L1 k 16-31 MFMA + barrier", "epilogue 1 + barrier", "L2 MFMA + epilogue + barrier", "L3 + output"};
        double sum[12] = {0}, tot = 0;
        for (int b = 0; b < nb; b++) for (int i = 0; i < 12; i++) sum[i] += (double)(st[b * 16 + i + 1] - st[b * 16 + i]);
        for (int i = 0; i < 12; i++) { printf("  %-30s %6.2f us\n", name[i], sum[i] / nb * 0.01); tot += sum[i] / nb * 0.01; }
        printf("  workgroup total %.2f us; start times of workgroups 0, 100, 255, 256, 300, 383: %.1f %.1f %.1f %.1f %.1f %.1f us\n", tot,
               (st[0] - t0) * 0.01, (st[100 * 16] - t0) * 0.01, (st[255 * 16] - t0) * 0.01, (st[256 * 16] - t0) * 0.01, (st[300 * 16] - t0) * 0.01, (st[383 * 16] - t0) * 0.01);
    }
#endif
    return 0;
}
