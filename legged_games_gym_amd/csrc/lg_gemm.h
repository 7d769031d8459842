// lg_gemm.h -- learner kernels for the WIDE actor / critic MLPs ([235 | 169, 512, 256, 128, 12 | 1]: reference
// legged_robot_config.py:205-208, the networks of anymal_c_rough, cassie, a1, anymal_b).  Their 1.1 MB of weights per net do not
// fit the LDS-resident scheme of lg_train.h, so every layer of the mini-batch pass is a tiled GEMM on the matrix cores with the
// layer's element-wise work fused into the epilogue:
//   forward   X_{l+1} = elu(X_l W_l^T + b_l)                  C[m][n] = sum_k A[m][k] B[n][k]      "NT", epilogue bias (+ ELU)
//   backward  G_l     = (G_{l+1} W_l) * elu'(X_l)             C[m][k] = sum_n A[m][n] B[n][k]      "NN", epilogue ELU' from the stored post-activation
//   weights   dW_l    = G_{l+1}^T X_l,  db_l = sum_m G_{l+1}  C[n][k] = sum_m A[m][n] B[m][k]      "TN", split over the 24 576 rows + fixed-order reduce
// (db rides along as one more output column: the X operand is extended by a column of ones).
// Arithmetic: v_mfma_f32_32x32x2_f32 -- exact f32 (a k-ordered fmaf chain, cdna_hip_programming.md section 3), f32 operands and
// accumulators, so the gradients agree with autograd to rounding and runs are bit-reproducible (no atomics).
// Tiling: 128 x 128 output tile per workgroup of 4 waves (2 x 2, each 64 x 64 = 2 x 2 MFMA tiles, 64 accumulator registers),
// k depth 16 per LDS stage, two stages (the next tile's global loads are in flight while the MFMAs of the current one run).  An
// f32 MFMA operand is ONE float per lane (A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31]), so the LDS image is simply
// [k][row] for both operands of all three forms; a 32x32x2 MFMA takes 64 cycles, the four ds_read_b32 that feed four of them are noise.
// Global loads run with the lanes along the memory-contiguous direction of each operand (full 64-byte row segments), transposed
// into the [k][row] image by the LDS write where needed; the mini-batch row gather (rows[]) is part of the address.
#pragma once
#include "lg_policy.h"

namespace lg {

#define LG_GT 128                      // output tile edge
#define LG_GK 16                       // k depth of one LDS stage
#define LG_GLD (LG_GT + 4)             // row stride of the LDS images (floats): bank = (4 k + row) % 32
#define LG_WIDE_MAX_SPLITS 64

enum { GEMM_FWD = 0, GEMM_DX = 1, GEMM_DW = 2 };

struct GemmNet {                       // one GEMM of one net; all sizes in the GEMM's own terms: C[M x N] = sum over K
    const float *A, *B;
    float *C;
    const float *bias;                 // FWD: [N]
    const float *act;                  // DX: post-activation X_l [M x N] (ldc) whose ELU' scales the result
    int M, N, K, lda, ldb, ldc;
    int elu;                           // FWD: apply ELU
    int tiles_m, tiles_n, splits;      // DW: `splits` chunks of the reduction (K = mini-batch rows), chunk length k_chunk
    int k_chunk;
    int ones_col;                      // DW: B column index that reads as 1.0 (the bias gradient), -1 none
};
struct GemmArgs {
    GemmNet net[2];
    const int64_t *rows;               // mini-batch row list (gather) or null; applies to the operand(s) named below
    int gather_a_rows;                 // FWD layer 0: A row m is input row rows[m]
    int gather_b_k;                    // DW layer 0: B row k (a mini-batch row) is input row rows[k]
    int mb;
};

typedef float f32x16g __attribute__((ext_vector_type(16)));

// One LDS stage of both operands, each [LG_GK][LG_GLD]
struct GemmStage { float a[LG_GK][LG_GLD]; float b[LG_GK][LG_GLD]; };

// ---- global -> register staging.  A tile element is addressed as (r, k): r = row of the output tile side (0..127), k = reduction
// index inside the stage (0..15).  KC: the operand is k-contiguous in memory (lanes along k); otherwise r-contiguous (lanes along r).
template <bool KC> struct TileMap {
    // element e (0..7) of thread t
    LG_DEV static void rk(int t, int e, int &r, int &k) {
        if (KC) { k = t & 15; r = (t >> 4) + 16 * e; }          // 16 lanes cover 64 contiguous bytes of one row; 8 passes cover 128 rows
        else { r = t & 127; k = (t >> 7) + 2 * e; }             // 128 lanes along r; 8 passes cover 16 k
    }
};

template <int MODE>
__global__ void __launch_bounds__(256) k_gemm_wide(const GemmArgs G) {
    const GemmNet &N = G.net[blockIdx.z];
    const int tile_m = blockIdx.x, tile_n = blockIdx.y % N.tiles_n, split = MODE == GEMM_DW ? blockIdx.y / N.tiles_n : 0;
    if (tile_m >= N.tiles_m || blockIdx.y >= N.tiles_n * (MODE == GEMM_DW ? N.splits : 1)) return;      // the grid is sized for the larger net
    __shared__ GemmStage st[2];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, wy = wave >> 1, wx = wave & 1;
    const int m0 = tile_m * LG_GT, n0 = tile_n * LG_GT;
    // reduction range of this workgroup
    const int k_begin = MODE == GEMM_DW ? split * N.k_chunk : 0;
    const int k_end = MODE == GEMM_DW ? min(N.K, k_begin + N.k_chunk) : N.K;
    // operand element fetchers (zero outside the matrices)
    auto fetch_a = [&](int r, int k) -> float {                  // A side: output row m0 + r, reduction index k
        const int m = m0 + r;
        if (MODE == GEMM_DW) {                                   // A[k][m]: G_{l+1}[row k][feature m]
            return (k < k_end && m < N.M) ? N.A[(size_t)k * N.lda + m] : 0.0f;
        } else {
            if (!(m < N.M && k < k_end)) return 0.0f;
            const size_t row = (MODE == GEMM_FWD && G.gather_a_rows) ? (size_t)G.rows[m] : (size_t)m;
            return N.A[row * N.lda + k];
        }
    };
    auto fetch_b = [&](int r, int k) -> float {                  // B side: output column n0 + r, reduction index k
        const int n = n0 + r;
        if (MODE == GEMM_FWD) return (n < N.N && k < k_end) ? N.B[(size_t)n * N.ldb + k] : 0.0f;        // W[n][k]
        if (MODE == GEMM_DX) return (n < N.N && k < k_end) ? N.B[(size_t)k * N.ldb + n] : 0.0f;         // W[k][n]
        if (!(k < k_end)) return 0.0f;                                                                    // DW: X_l[row k][feature n] (+ ones column)
        if (n == N.ones_col) return 1.0f;
        if (!(n < N.N)) return 0.0f;
        const size_t row = G.gather_b_k ? (size_t)G.rows[k] : (size_t)k;
        return N.B[row * N.ldb + n];
    };
    constexpr bool A_KC = MODE != GEMM_DW, B_KC = MODE == GEMM_FWD;     // which operands are k-contiguous in memory
    float ra[8], rb[8];
    auto load_stage = [&](int kb) {
#pragma unroll
        for (int e = 0; e < 8; e++) {
            int r, k;
            TileMap<A_KC>::rk(t, e, r, k); ra[e] = fetch_a(r, kb + k);
            TileMap<B_KC>::rk(t, e, r, k); rb[e] = fetch_b(r, kb + k);
        }
    };
    auto store_stage = [&](GemmStage &s) {
#pragma unroll
        for (int e = 0; e < 8; e++) {
            int r, k;
            TileMap<A_KC>::rk(t, e, r, k); s.a[k][r] = ra[e];
            TileMap<B_KC>::rk(t, e, r, k); s.b[k][r] = rb[e];
        }
    };
    f32x16g acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int c = 0; c < 16; c++) acc[i][j][c] = 0.0f;

    const int n_stage = (k_end - k_begin + LG_GK - 1) / LG_GK;
    if (n_stage > 0) { load_stage(k_begin); store_stage(st[0]); }
    __syncthreads();
    const int li = lane & 31, lh = lane >> 5;
    for (int s = 0; s < n_stage; s++) {
        const bool more = s + 1 < n_stage;
        if (more) load_stage(k_begin + (s + 1) * LG_GK);                  // in flight during this stage's MFMAs
        const GemmStage &cur = st[s & 1];
#pragma unroll
        for (int kk = 0; kk < LG_GK; kk += 2) {
            const float a0 = cur.a[kk + lh][64 * wy + li], a1 = cur.a[kk + lh][64 * wy + 32 + li];
            const float b0 = cur.b[kk + lh][64 * wx + li], b1 = cur.b[kk + lh][64 * wx + 32 + li];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) store_stage(st[(s + 1) & 1]);
        __syncthreads();
    }
    // ---- epilogue.  C/D map of the 32 x 32 tile: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float *C = N.C + (MODE == GEMM_DW ? (size_t)split * N.M * N.ldc : 0);
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int col = n0 + 64 * wx + 32 * j + li;
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const int row = m0 + 64 * wy + 32 * i + (c & 3) + 8 * (c >> 2) + 4 * lh;
                if (row < N.M && col < (MODE == GEMM_DW ? N.ldc : N.N)) {
                    float v = acc[i][j][c];
                    if (MODE == GEMM_FWD) { v += N.bias[col]; if (N.elu) v = elu1(v); }
                    if (MODE == GEMM_DX) { const float x = N.act[(size_t)row * N.ldc + col]; v *= (x > 0.0f ? 1.0f : x + 1.0f); }
                    C[(size_t)row * N.ldc + col] = v;
                }
            }
        }
}

// dW / db from the per-split partials [splits][N][K + 1] in a fixed order: gw[n][k] (torch layout), gb[n]
struct WideReduceArgs { const float *part[2]; float *gw[2], *gb[2]; int N[2], K[2], splits[2]; };
__global__ void __launch_bounds__(256) k_wide_reduce(const WideReduceArgs R) {
    const int z = blockIdx.y, N = R.N[z], K = R.K[z], ld = K + 1;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * ld) return;
    float s = 0.0f;
    for (int p = 0; p < R.splits[z]; p++) s += R.part[z][(size_t)p * N * ld + i];
    const int n = i / ld, k = i % ld;
    if (k < K) R.gw[z][(size_t)n * K + k] = s;
    else R.gb[z][n] = s;
}

}  // namespace lg
