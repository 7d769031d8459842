// Stand-alone timing + phase stamps of the wide rollout actor (csrc/lg_policy.h: k_policy_act_wide, 235-512-256-128-12, 32 envs per workgroup).
// Build with the flags in chain_probe.hip (-DLG_CHAIN_PROF for the stamps):  ./actor_probe [num_envs]
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
    const int N = argc > 1 ? atoi(argv[1]) : 4096, d[5] = {235, 512, 256, 128, 12}, k0s = 15;
    lg::PolicyWideArgs w; memset(&w, 0, sizeof w);
    w.base.obs = dev_random((size_t)N * d[0], 1, 1.0f); w.base.std = dev_random(12, 2, 0.5f);
    CK(hipMalloc(&w.base.actions, (size_t)N * 12 * 4)); CK(hipMalloc(&w.base.mean, (size_t)N * 12 * 4));
    w.base.step = 3; w.base.seed = 7; w.base.num_envs = N; w.base.num_obs = d[0]; w.base.num_actions = 12;
    for (int l = 0; l < 4; l++) {
        const int ks = l == 0 ? k0s : d[l] / 16, ot = (d[l + 1] + 31) / 32;
        float *W = dev_random((size_t)d[l + 1] * d[l], 10 + l, 0.08f), *b = dev_random(d[l + 1], 30 + l, 0.1f);
        void *wp; CK(hipMalloc(&wp, (size_t)ot * ks * 2 * 64 * 8 * sizeof(__bf16))); float *bp; CK(hipMalloc(&bp, ot * 32 * sizeof(float)));
        hipLaunchKernelGGL(lg::k_policy_pack_wide, dim3(512), dim3(256), 0, 0, W, b, d[l], d[l + 1], ks, ot, l == 0 ? 1 : 0, (__bf16 *)wp, bp);
        w.wb[l] = (const lg::bf16x8g *)wp; w.bb[l] = bp;
    }
    CK(hipDeviceSynchronize());
    const dim3 grid((N + LG_PW_ENVS - 1) / LG_PW_ENVS), block(64 * LG_PW_WAVES);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL((lg::k_policy_act_wide<15, 16, 8, 4>), grid, block, 0, 0, w);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < 50; r++) hipLaunchKernelGGL((lg::k_policy_act_wide<15, 16, 8, 4>), grid, block, 0, 0, w);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    std::vector<float> out((size_t)N * 12);
    CK(hipMemcpy(out.data(), w.base.mean, out.size() * 4, hipMemcpyDeviceToHost));
    double cs = 0; for (float v : out) cs += v;
    printf("wide actor, %d envs (%d workgroups): %.1f us per launch  (checksum %.6f)\n", N, grid.x, ms * 1e3 / 50, cs);
#ifdef LG_CHAIN_PROF
    std::vector<unsigned long long> st(2048 * 16);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(lg::g_chain_prof), st.size() * sizeof(unsigned long long)));
    const char *name[9] = {"prime + obs load + barrier", "L0 MFMA (2 tiles)", "prime L1 + epilogue 0", "barrier", "L1 MFMA", "prime L2 + epilogue 1 + barrier", "L2 MFMA + epilogue + barrier", "L3 MFMA (wave 0)", "sampling + stores"};
    double sum[9] = {0}, tot = 0;
    for (int g = 0; g < (int)grid.x; g++) for (int i = 0; i < 9; i++) sum[i] += (double)(st[g * 16 + i + 1] - st[g * 16 + i]);
    for (int i = 0; i < 9; i++) { printf("  %-34s %7.0f cycles\n", name[i], sum[i] / grid.x); tot += sum[i] / grid.x; }
    printf("  workgroup total %.0f cycles\n", tot);
#endif
    return 0;
}
