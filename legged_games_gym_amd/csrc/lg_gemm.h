// lg_gemm.h -- learner kernels for the WIDE actor / critic MLPs ([235 | 169, 512, 256, 128, 12 | 1]: reference
// legged_robot_config.py:205-208, the networks of anymal_c_rough, cassie, a1, anymal_b).  Their 1.1 MB of weights per net do not
// fit the LDS-resident scheme of lg_train.h, so every layer of the mini-batch pass is a tiled GEMM on the matrix cores with the
// layer's element-wise work fused into the epilogue:
//   forward   X_{l+1} = elu(X_l W_l^T + b_l)                  C[m][n] = sum_k A[m][k] B[n][k]      "NT", epilogue bias (+ ELU)
//   backward  G_l     = (G_{l+1} W_l) * elu'(X_l)             C[m][k] = sum_n A[m][n] B[n][k]      "NN", epilogue ELU' from the stored post-activation
//   weights   dW_l    = G_{l+1}^T X_l,  db_l = sum_m G_{l+1}  C[n][k] = sum_m A[m][n] B[m][k]      "TN", split over the 24 576 rows + fixed-order reduce
// (db rides along: the workgroups of the first tile column also sum their G^T tiles over the rows and store that as output column K).
// Arithmetic: v_mfma_f32_32x32x2_f32 -- exact f32 (a k-ordered fmaf chain, cdna_hip_programming.md section 3), f32 operands and
// accumulators, so the gradients agree with autograd to rounding and runs are bit-reproducible (no atomics).
// Tiling: 128 x 128 output tile per workgroup of 4 waves (2 x 2, each 64 x 64 = 2 x 2 MFMA tiles, 64 accumulator registers),
// k depth 16 per LDS stage, two stages (the next tile's global loads are in flight while the MFMAs of the current one run).  An
// f32 MFMA operand is ONE float per lane (A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31]), so the LDS image is simply
// [k][row] for both operands of all three forms; a 32x32x2 MFMA takes 64 cycles, the four ds_read_b32 that feed four of them are noise.
// Global loads are 16-byte vectors along the memory-contiguous direction of each operand (the scalar-load version of this kernel
// was bound by the texture addresser: 50 TFLOP/s), transposed into the [k][row] image by the LDS write where needed; the
// mini-batch row gather (rows[]) is part of the address.
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

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // 4 floats at float alignment (row strides like 235 are not 16-byte multiples)

// up to 4 consecutive floats at p (nv of them exist, the rest read as 0)
LG_DEV float4 load4(const float *p, int nv) {
    if (nv >= 4) { const f32x4u v = *reinterpret_cast<const f32x4u *>(p); return make_float4(v.x, v.y, v.z, v.w); }
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nv > 0) r.x = p[0];
    if (nv > 1) r.y = p[1];
    if (nv > 2) r.z = p[2];
    return r;
}

// ---- global -> register staging: 8 floats per thread and operand as two 4-vectors along the operand's memory-contiguous direction.
// A tile element is (r, k): r = index on the output-tile side (0..127), k = reduction index inside the stage (0..15).
//   k-contiguous operand (A of FWD / DX, B of FWD):  thread t holds k = 4 (t & 3) .. +3 of rows r = (t >> 2) + 64 e,  e = 0, 1
//   r-contiguous operand (B of DX, A and B of DW):   thread t holds r = 4 (t & 31) .. +3 of k = (t >> 5) + 8 e,      e = 0, 1
template <int MODE>
__global__ void __launch_bounds__(256) k_gemm_wide(const GemmArgs G) {
    const GemmNet &N = G.net[blockIdx.z];
    const int tile_m = blockIdx.x, tile_n = blockIdx.y % N.tiles_n, split = MODE == GEMM_DW ? blockIdx.y / N.tiles_n : 0;
    if (tile_m >= N.tiles_m || blockIdx.y >= N.tiles_n * (MODE == GEMM_DW ? N.splits : 1)) return;      // the grid is sized for the larger net
    __shared__ GemmStage st[2];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, wy = wave >> 1, wx = wave & 1;
    const int m0 = tile_m * LG_GT, n0 = tile_n * LG_GT;
    const int k_begin = MODE == GEMM_DW ? split * N.k_chunk : 0;
    const int k_end = MODE == GEMM_DW ? min(N.K, k_begin + N.k_chunk) : N.K;
    constexpr bool A_KC = MODE != GEMM_DW, B_KC = MODE == GEMM_FWD;     // which operands are k-contiguous in memory

    // ---- A side
    const float *pa[2]; int na[2];          // KC: row pointer (at k = 0) and 1 / 0 row validity; RC: unused / number of valid columns from this thread's first
    if (A_KC) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int m = m0 + (t >> 2) + 64 * e;
            na[e] = m < N.M;
            const size_t row = (MODE == GEMM_FWD && G.gather_a_rows && na[e]) ? (size_t)G.rows[m] : (size_t)(na[e] ? m : 0);
            pa[e] = N.A + row * N.lda;
        }
    } else {                                                      // DW: A[k][m] = G_{l+1}[row k][feature m]
        const int m = m0 + 4 * (t & 31);
        na[0] = na[1] = max(0, min(4, N.M - m));
        pa[0] = pa[1] = N.A + m;
    }
    // ---- B side
    const float *pb[2]; int nb[2];
    if (B_KC) {                                                   // FWD: W[n][k]
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int n = n0 + (t >> 2) + 64 * e;
            nb[e] = n < N.N;
            pb[e] = N.B + (size_t)(nb[e] ? n : 0) * N.ldb;
        }
    } else {                                                      // DX: W[k][n];  DW: X_l[row k][feature n] (+ ones column)
        const int n = n0 + 4 * (t & 31);
        nb[0] = nb[1] = max(0, min(4, N.N - n));
        pb[0] = pb[1] = N.B + n;
    }
    int64_t brow_next[2] = {0, 0};                                // DW with a row gather: storage row of this thread's k in the NEXT stage to load
    if (MODE == GEMM_DW && G.gather_b_k) {
#pragma unroll
        for (int e = 0; e < 2; e++) brow_next[e] = G.rows[min(k_begin + (t >> 5) + 8 * e, max(k_end - 1, 0))];
    }

    float4 ra[2], rb[2];
    auto load_stage = [&](int kb) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
            if (A_KC) {
                const int k = kb + 4 * (t & 3);
                ra[e] = load4(pa[e] + k, na[e] ? k_end - k : 0);
            } else {
                const int k = kb + (t >> 5) + 8 * e;
                ra[e] = load4(pa[e] + (size_t)min(k, k_end - 1) * N.lda, k < k_end ? na[e] : 0);
            }
            if (B_KC) {
                const int k = kb + 4 * (t & 3);
                rb[e] = load4(pb[e] + k, nb[e] ? k_end - k : 0);
            } else {
                const int k = kb + (t >> 5) + 8 * e;
                const int kc = min(k, k_end - 1);
                size_t row = (size_t)kc;
                if (MODE == GEMM_DW && G.gather_b_k) {            // gathered rows: the index was requested a stage ahead (two dependent loads otherwise)
                    row = (size_t)brow_next[e];
                    brow_next[e] = G.rows[min(k + LG_GK, k_end - 1)];
                }
                rb[e] = load4(pb[e] + row * N.ldb, k < k_end ? nb[e] : 0);
            }
        }
    };
    auto store_stage = [&](GemmStage &s) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
            if (A_KC) {
                const int r = (t >> 2) + 64 * e, k = 4 * (t & 3);
                s.a[k][r] = ra[e].x; s.a[k + 1][r] = ra[e].y; s.a[k + 2][r] = ra[e].z; s.a[k + 3][r] = ra[e].w;
            } else {
                *reinterpret_cast<float4 *>(&s.a[(t >> 5) + 8 * e][4 * (t & 31)]) = ra[e];
            }
            if (B_KC) {
                const int r = (t >> 2) + 64 * e, k = 4 * (t & 3);
                s.b[k][r] = rb[e].x; s.b[k + 1][r] = rb[e].y; s.b[k + 2][r] = rb[e].z; s.b[k + 3][r] = rb[e].w;
            } else {
                *reinterpret_cast<float4 *>(&s.b[(t >> 5) + 8 * e][4 * (t & 31)]) = rb[e];
            }
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
    float colsum = 0.0f;                                                  // DW, first tile column: db[m0 + t] partial (threads < 128)
    for (int s = 0; s < n_stage; s++) {
        const bool more = s + 1 < n_stage;
        if (more) load_stage(k_begin + (s + 1) * LG_GK);                  // in flight during this stage's MFMAs
        const GemmStage &cur = st[s & 1];
        if (MODE == GEMM_DW && tile_n == 0 && t < LG_GT) {
#pragma unroll
            for (int kk = 0; kk < LG_GK; kk++) colsum += cur.a[kk][t];
        }
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
    if (MODE == GEMM_DW && tile_n == 0 && t < LG_GT && m0 + t < N.M) C[(size_t)(m0 + t) * N.ldc + N.N] = colsum;      // bias-gradient column
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int col = n0 + 64 * wx + 32 * j + li;
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const int row = m0 + 64 * wy + 32 * i + (c & 3) + 8 * (c >> 2) + 4 * lh;
                if (row < N.M && col < N.N) {
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
