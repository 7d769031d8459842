// Stand-alone timing + spot check of the wide learner's GEMMs (csrc/lg_gemm.h) on the shapes of one PPO mini-batch of the rough tasks
// (24 576 rows, 235-512-256-128 nets, actor + critic per launch).  Compiles in seconds (the full library takes minutes), so kernel
// experiments are tried here first (build with the flags in chain_probe.hip)
#include "../../legged_games_gym_amd/csrc/lg_gemm.h"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <class F> static double time_us(F launch, int reps = 20) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; r++) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3 / reps;
}
static float *dev_random(size_t n, unsigned seed, float scale, std::vector<float> *keep = nullptr) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; h[i] = scale * ((int)(s >> 8) / 8388608.0f - 1.0f); }
    float *d; CK(hipMalloc(&d, n * sizeof(float))); CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    if (keep) keep->swap(h);
    return d;
}

static void run_dw(int mb, int M, int N, int ldb) {      // dW[M][N] = G[mb][M]^T X[mb][N]
    const int tiles = ((M + LG_GT - 1) / LG_GT) * ((N + LG_GT - 1) / LG_GT);
    int sp = (384 + tiles - 1) / tiles; if (sp > LG_WIDE_MAX_SPLITS) sp = LG_WIDE_MAX_SPLITS;
    int chunk = (mb + sp - 1) / sp; chunk = ((chunk + LG_BK - 1) / LG_BK) * LG_BK; sp = (mb + chunk - 1) / chunk;
    const int ldc = (N + 1 + 3) & ~3;
    std::vector<float> hG, hX;
    lg::GemmArgs a; memset(&a, 0, sizeof a);
    float *part[2];
    for (int n = 0; n < 2; n++) {
        lg::GemmNet &g = a.net[n];
        g.A = dev_random((size_t)mb * M, 3 + n, 0.05f, n == 0 ? &hG : nullptr); g.lda = M;
        g.B = dev_random((size_t)mb * ldb, 7 + n, 1.0f, n == 0 ? &hX : nullptr); g.ldb = ldb;
        CK(hipMalloc(&part[n], (size_t)sp * M * ldc * sizeof(float))); g.C = part[n]; g.ldc = ldc;
        g.M = M; g.N = N; g.K = mb; g.splits = sp; g.k_chunk = chunk;
        g.tiles_m = (M + LG_GT - 1) / LG_GT; g.tiles_n = (N + LG_GT - 1) / LG_GT;
    }
    a.mb = mb;
    const dim3 grid(a.net[0].tiles_m, a.net[0].tiles_n * sp, 2);
    const double us = time_us([&] { hipLaunchKernelGGL((lg::k_gemm_wide_bf16x3<lg::GEMM_DW>), grid, dim3(256), 0, 0, a); });
    std::vector<float> hp((size_t)sp * M * ldc);
    CK(hipMemcpy(hp.data(), part[0], hp.size() * sizeof(float), hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int t = 0; t < 24; t++) {
        const int m = (t * 37 + 5) % M, n = t < 4 ? N : (t * 101 + 3) % N;      // n == N: the bias-gradient column
        double ref = 0.0, got = 0.0, mag = 0.0;
        for (int r = 0; r < mb; r++) { const double v = (double)hG[(size_t)r * M + m] * (n == N ? 1.0 : (double)hX[(size_t)r * ldb + n]); ref += v; mag += fabs(v); }
        for (int s = 0; s < sp; s++) got += hp[((size_t)s * M + m) * ldc + n];
        worst = fmax(worst, fabs(got - ref) / mag);
    }
    const double bytes = 2.0 * ((double)mb * (M + N) * 4 + (double)sp * M * ldc * 4);
    printf("dW  %4d x %4d over %d rows: %7.1f us  (%d splits, %d workgroups x 2 nets)  %.2f TB/s of operands+partials, %.1f TFLOP/s  err/|sum| %.1e\n",
           M, N, mb, us, sp, tiles * sp, bytes / us * 1e-6, 2.0 * 2.0 * mb * M * N / us * 1e-6, worst);
}

static void run_dx(int mb, int N, int K) {               // G_l[mb][N] = (G_{l+1}[mb][K] W[K][N]) * elu'(X_l[mb][N])
    std::vector<float> hG, hW, hX;
    lg::GemmArgs a; memset(&a, 0, sizeof a);
    float *out[2];
    for (int n = 0; n < 2; n++) {
        lg::GemmNet &g = a.net[n];
        g.A = dev_random((size_t)mb * K, 13 + n, 0.05f, n == 0 ? &hG : nullptr); g.lda = K;
        g.B = dev_random((size_t)K * N, 17 + n, 0.1f, n == 0 ? &hW : nullptr); g.ldb = N;
        g.act = dev_random((size_t)mb * N, 19 + n, 1.0f, n == 0 ? &hX : nullptr);
        CK(hipMalloc(&out[n], (size_t)mb * N * sizeof(float))); g.C = out[n]; g.ldc = N;
        g.M = mb; g.N = N; g.K = K; g.splits = 1; g.k_chunk = K;
        g.tiles_m = (mb + LG_GT - 1) / LG_GT; g.tiles_n = (N + LG_GT - 1) / LG_GT;
    }
    a.mb = mb;
    const dim3 grid(a.net[0].tiles_m, a.net[0].tiles_n, 2);
    const double us = time_us([&] { hipLaunchKernelGGL((lg::k_gemm_wide_bf16x3<lg::GEMM_DX>), grid, dim3(256), 0, 0, a); });
    std::vector<float> ho((size_t)mb * N);
    CK(hipMemcpy(ho.data(), out[0], ho.size() * sizeof(float), hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int t = 0; t < 24; t++) {
        const int r = (t * 1009 + 11) % mb, n = (t * 53 + 1) % N;
        double ref = 0.0, mag = 0.0;
        for (int k = 0; k < K; k++) { const double v = (double)hG[(size_t)r * K + k] * hW[(size_t)k * N + n]; ref += v; mag += fabs(v); }
        const double x = hX[(size_t)r * N + n], d = x > 0 ? 1.0 : x + 1.0;
        worst = fmax(worst, fabs(ho[(size_t)r * N + n] - ref * d) / mag);
    }
    const double bytes = 2.0 * ((double)mb * K * 4 + 2.0 * mb * N * 4);
    printf("dX  %d rows, %4d -> %4d: %7.1f us  %.2f TB/s of operands, %.1f TFLOP/s  err/|sum| %.1e\n", mb, K, N, us, bytes / us * 1e-6,
           2.0 * 2.0 * mb * N * K / us * 1e-6, worst);
}

static void run_outbwd(int mb, int chunks) {            // the <= 16-wide output layer: G_2 = (dZ_3 W_3) * elu'(A_2), dW_3 / db_3 partials
    lg::OutBwdArgs o; memset(&o, 0, sizeof o);
    o.mb = mb;
    std::vector<float> hdz, hw, ha;
    for (int n = 0; n < 2; n++) {
        lg::OutBwdNet &q = o.net[n];
        q.N = n == 0 ? 12 : 1; q.K = 128; q.ld = (128 + 1 + 3) & ~3; q.chunks = chunks; q.rows_per_chunk = (mb + chunks - 1) / chunks;
        q.dz = dev_random((size_t)mb * q.N, 41 + n, 0.05f, n == 0 ? &hdz : nullptr);
        q.w = dev_random((size_t)q.N * 128, 43 + n, 0.1f, n == 0 ? &hw : nullptr);
        q.act = dev_random((size_t)mb * 128, 45 + n, 1.0f, n == 0 ? &ha : nullptr);
        CK(hipMalloc(&q.g, (size_t)mb * 128 * 4)); CK(hipMalloc(&q.part, (size_t)chunks * q.N * q.ld * 4));
    }
    const double us = time_us([&] { hipLaunchKernelGGL((lg::k_wide_out_bwd<LG_OUT_MAXN>), dim3(chunks, 2), dim3(256), 0, 0, o); });
    std::vector<float> hg((size_t)mb * 128), hp((size_t)chunks * 12 * o.net[0].ld);
    CK(hipMemcpy(hg.data(), o.net[0].g, hg.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hp.data(), o.net[0].part, hp.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int t = 0; t < 16; t++) {
        const int r = (t * 1531 + 7) % mb, k = (t * 29 + 3) % 128;
        double ref = 0.0, mag = 1e-30;
        for (int n = 0; n < 12; n++) { const double v = (double)hdz[(size_t)r * 12 + n] * hw[(size_t)n * 128 + k]; ref += v; mag += fabs(v); }
        const double x = ha[(size_t)r * 128 + k];
        worst = fmax(worst, fabs(hg[(size_t)r * 128 + k] - ref * (x > 0 ? 1.0 : x + 1.0)) / mag);
        const int n = t % 12;                                                       // dW_3[n][k] over all chunks
        double refw = 0.0, gotw = 0.0, magw = 1e-30;
        for (int rr = 0; rr < mb; rr++) { const double v = (double)hdz[(size_t)rr * 12 + n] * ha[(size_t)rr * 128 + k]; refw += v; magw += fabs(v); }
        for (int c = 0; c < chunks; c++) gotw += hp[((size_t)c * 12 + n) * o.net[0].ld + k];
        worst = fmax(worst, fabs(gotw - refw) / magw);
    }
    printf("out-layer backward, %d rows, %d chunks: %6.1f us  %.2f TB/s of A_2 read + G_2 written  err %.1e\n", mb, chunks, us, 2.0 * 2.0 * mb * 128 * 4 / us * 1e-6, worst);
}

int main(int argc, char **argv) {
    const int mb = argc > 1 ? atoi(argv[1]) : 24576;
    run_outbwd(mb, 256); run_outbwd(mb, 512); run_outbwd(mb, 1024);
    run_dw(mb, 512, 235, 236);
    run_dw(mb, 256, 512, 512);
    run_dw(mb, 128, 256, 256);
    run_dx(mb, 512, 256);
    run_dx(mb, 256, 128);
    return 0;
}
