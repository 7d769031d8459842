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

// Epilogue of both builds.  C/D map of a 32 x 32 MFMA tile: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) -- one
// column per lane, so storing straight from the accumulators is 64 dword stores per thread, and the kernels were store-ISSUE
// bound (~1 TB/s of output whatever the K depth).  Each wave therefore turns its 64 x 64 result, 32 rows at a time, through a
// private 8 KB LDS patch ([32][64] floats, carved from the stage buffers after the main loop) and writes it row-wise: a lane owns
// 4 consecutive columns (16 lanes = one 256-byte row segment), 16 dwordx4 stores per thread.  Bias / stored post-activations
// are read the same way (float4) BEFORE the stores of their batch.
template <int MODE>
LG_DEV void gemm_epilogue(const GemmNet &N, float *C, const f32x16g (&acc)[2][2], int m0, int n0, int wy, int wx, int lane, float *lds_scratch) {
    const int li = lane & 31, lh = lane >> 5, wave = 2 * wy + wx;
    float (*ep)[64] = reinterpret_cast<float (*)[64]>(lds_scratch + wave * 32 * 64);
    const int cq = 4 * (lane & 15), col = n0 + 64 * wx + cq;              // this lane's 4 columns in the row-wise phase
    const int nv = max(0, min(4, N.N - col));
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == GEMM_FWD) bias = load4(N.bias + col, nv);
    const bool vec_ok = (N.ldc & 3) == 0;                                  // rows of C start 16-byte aligned
#pragma unroll
    for (int i = 0; i < 2; i++) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int c = 0; c < 16; c++) ep[(c & 3) + 8 * (c >> 2) + 4 * lh][32 * j + li] = acc[i][j][c];
        __builtin_amdgcn_wave_barrier();                                   // one wave, LDS in order: its own writes are visible to its reads
        float4 v[8], xa[8];
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const int rr = 4 * p + (lane >> 4), row = m0 + 64 * wy + 32 * i + rr;
            v[p] = *reinterpret_cast<const float4 *>(&ep[rr][cq]);
            if (MODE == GEMM_DX) xa[p] = load4(N.act + (size_t)min(row, N.M - 1) * N.ldc + col, row < N.M ? nv : 0);
        }
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const int rr = 4 * p + (lane >> 4), row = m0 + 64 * wy + 32 * i + rr;
            if (row >= N.M || nv <= 0) continue;
            float4 o = v[p];
            if (MODE == GEMM_FWD) {
                o.x += bias.x; o.y += bias.y; o.z += bias.z; o.w += bias.w;
                if (N.elu) { o.x = elu1(o.x); o.y = elu1(o.y); o.z = elu1(o.z); o.w = elu1(o.w); }
            }
            if (MODE == GEMM_DX) {
                o.x *= (xa[p].x > 0.f ? 1.f : xa[p].x + 1.f); o.y *= (xa[p].y > 0.f ? 1.f : xa[p].y + 1.f);
                o.z *= (xa[p].z > 0.f ? 1.f : xa[p].z + 1.f); o.w *= (xa[p].w > 0.f ? 1.f : xa[p].w + 1.f);
            }
            float *dst = C + (size_t)row * N.ldc + col;
            if (nv == 4 && vec_ok) *reinterpret_cast<float4 *>(dst) = o;
            else { dst[0] = o.x; if (nv > 1) dst[1] = o.y; if (nv > 2) dst[2] = o.z; if (nv > 3) dst[3] = o.w; }
        }
    }
}

// ---- global -> register staging: 8 floats per thread and operand as two 4-vectors along the operand's memory-contiguous direction.
// A tile element is (r, k): r = index on the output-tile side (0..127), k = reduction index inside the stage (0..15).
//   k-contiguous operand (A of FWD / DX, B of FWD):  thread t holds k = 4 (t & 3) .. +3 of rows r = (t >> 2) + 64 e,  e = 0, 1
//   r-contiguous operand (B of DX, A and B of DW):   thread t holds r = 4 (t & 31) .. +3 of k = (t >> 5) + 8 e,      e = 0, 1
// blockIdx -> (tile_m, tile_n, split), XCD-aware.  Workgroups are handed to the 8 XCDs round-robin by linear id, each XCD has its own
// 4 MB L2, and the tiles that read the same operand rows -- the tiles_m x tiles_n tiles of one dW row-chunk, the tiles_n column tiles of one
// row tile in FWD / DX -- were 4-192 ids apart: on different XCDs or at different times, so every re-read went to the Infinity Cache /
// HBM (dW of layer 0: 386 MB of traffic for 146 MB of operands, 4.6 TB/s).  Here the ids of one residue class mod 8 (= one XCD, in
// dispatch order) are mapped to CONSECUTIVE logical tiles, so the sharers run side by side on one L2 and walk their k-slabs in step.
template <int MODE> LG_DEV bool tile_coords(const GemmNet &N, int &tile_m, int &tile_n, int &split) {
    const int T = gridDim.x * gridDim.y, i = blockIdx.x + gridDim.x * blockIdx.y;
    const int q = T >> 3, r = T & 7, c = i & 7, j = i >> 3;
    const int L = (c < r ? c * (q + 1) : r * (q + 1) + (c - r) * q) + j;         // bijection [0, T) -> [0, T)
    const int per = N.tiles_m * N.tiles_n;
    if (MODE == GEMM_DW) {
        split = L / per;
        const int tile = L - split * per;
        tile_m = tile % N.tiles_m; tile_n = tile / N.tiles_m;
        return split < N.splits;
    }
    split = 0;
    tile_m = L / N.tiles_n; tile_n = L - tile_m * N.tiles_n;
    return tile_m < N.tiles_m;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_gemm_wide(const GemmArgs G) {
    const GemmNet &N = G.net[blockIdx.z];
    int tile_m, tile_n, split;
    if (!tile_coords<MODE>(N, tile_m, tile_n, split)) return;                    // the grid is sized for the larger net
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
    float *C = N.C + (MODE == GEMM_DW ? (size_t)split * N.M * N.ldc : 0);
    if (MODE == GEMM_DW && tile_n == 0 && t < LG_GT && m0 + t < N.M) C[(size_t)(m0 + t) * N.ldc + N.N] = colsum;      // bias-gradient column
    gemm_epilogue<MODE>(N, C, acc, m0, n0, wy, wx, lane, reinterpret_cast<float *>(&st));
}

// ------------------------------------------------------------------ split-bf16 ("bf16x3") build of the same three GEMMs
// v_mfma_f32_32x32x16_bf16 runs at 16 x the f32-MFMA rate.  Every f32 operand x is split ONCE, when its tile is written to LDS, into
// hi = bf16(x) and lo = bf16(x - hi) (16 significand bits together), and each product is formed as hi*hi + hi*lo + lo*hi with f32
// accumulation: three MFMAs instead of one at 16 x the rate, i.e. ~5 x the f32 path; the dropped lo*lo term and the 16-bit
// operands bound the relative error of a product by ~2^-15.  HBM traffic, tiling and epilogues are those of k_gemm_wide; the
// LDS images become [row][k] bf16 (k-contiguous, 80-byte rows: conflict-free ds_read_b128 fragments of 8 k-values), one pair
// (hi, lo) per operand; stage depth 32.  Selected by lg_mlp_wide_set_precision (default) -- the exact-f32 kernels stay available.
#define LG_BK 32                       // k depth of one stage
#define LG_BLD 40                      // bf16 per LDS image row (32 + 8 pad): 80 bytes
typedef __bf16 bf16x4g __attribute__((ext_vector_type(4)));
typedef float f32x4g __attribute__((ext_vector_type(4)));
struct GemmStageB { __bf16 ahi[LG_GT][LG_BLD], alo[LG_GT][LG_BLD], bhi[LG_GT][LG_BLD], blo[LG_GT][LG_BLD]; };      // 40 KB

#define LG_TRS 160                     // bf16 per row of a [k][feature] image: 128 features + 32 pad = 320 bytes (= 64 mod 256: the four
                                       // rows a transposed read gathers sit on disjoint bank groups); 32 x 320 B = the 10 240 B of a [128][40] image
static_assert(LG_BK * LG_TRS == LG_GT * LG_BLD, "the two image layouts share one buffer");
typedef short s16x4g __attribute__((ext_vector_type(4)));
// MFMA 32x32x16 operand fragment of rows r0 .. r0+31, k = ks .. ks+15 from a [k][feature] image: lane l needs feature r0 + (l & 31),
// k = ks + 8 (l >> 5) + 0..7.  ds_read_b64_tr_b16 (gfx950): every 16-lane group reads a 4 (k) x 16 (feature) block and lane i of the
// group receives column i; lane 4q + p of the group supplies the address of row q, columns 4p .. 4p+3 (tools/ubench/tr16_probe.hip
// checks this mapping on the device).  EXEC must be full: called from the uniform main loop only.
LG_DEV bf16x8g frag_tr(const __bf16 *img, int r0, int ks, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const __bf16 *a = img + (ks + 8 * (g >> 1) + q) * LG_TRS + r0 + 16 * (g & 1) + 4 * p;
    const s16x4g v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4g *)a);
    const s16x4g v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4g *)(a + 4 * LG_TRS));
    union { struct { s16x4g lo, hi; } s; bf16x8g v; } u;
    u.s.lo = v0; u.s.hi = v1;
    return u.v;
}
LG_DEV void split4(float4 x, bf16x4g &hi, bf16x4g &lo) {
    const f32x4g v = {x.x, x.y, x.z, x.w};
    hi = __builtin_convertvector(v, bf16x4g);
    lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4g), bf16x4g);
}

template <int MODE>
__global__ void __launch_bounds__(256) k_gemm_wide_bf16x3(const GemmArgs G) {
    const GemmNet &N = G.net[blockIdx.z];
    int tile_m, tile_n, split;
    if (!tile_coords<MODE>(N, tile_m, tile_n, split)) return;                    // the grid is sized for the larger net
    __shared__ GemmStageB st;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, wy = wave >> 1, wx = wave & 1;
    const int m0 = tile_m * LG_GT, n0 = tile_n * LG_GT;
    const int k_begin = MODE == GEMM_DW ? split * N.k_chunk : 0;
    const int k_end = MODE == GEMM_DW ? min(N.K, k_begin + N.k_chunk) : N.K;
    constexpr bool A_KC = MODE != GEMM_DW, B_KC = MODE == GEMM_FWD;
    // thread -> tile elements.  k-contiguous operand: row r = t >> 1, 16 consecutive k from 16 (t & 1).
    //                           r-contiguous operand: the 4 x 4 block rows r = 4 (t & 31) .. +3, k = 4 (t >> 5) .. +3.
    const float *pa; int na;
    if (A_KC) {
        const int m = m0 + (t >> 1);
        na = m < N.M;
        const size_t row = (MODE == GEMM_FWD && G.gather_a_rows && na) ? (size_t)G.rows[m] : (size_t)(na ? m : 0);
        pa = N.A + row * N.lda;
    } else {
        const int m = m0 + 4 * (t & 31);
        na = max(0, min(4, N.M - m));
        pa = N.A + m;
    }
    const float *pb; int nb;
    if (B_KC) {
        const int n = n0 + (t >> 1);
        nb = n < N.N;
        pb = N.B + (size_t)(nb ? n : 0) * N.ldb;
    } else {
        const int n = n0 + 4 * (t & 31);
        nb = max(0, min(4, N.N - n));
        pb = N.B + n;
    }
    int64_t brow_next[4] = {0, 0, 0, 0};
    if (MODE == GEMM_DW && G.gather_b_k) {
#pragma unroll
        for (int c = 0; c < 4; c++) brow_next[c] = G.rows[min(k_begin + 4 * (t >> 5) + c, max(k_end - 1, 0))];
    }
    float4 ra[4], rb[4];
    // Fast path of a stage's loads: the whole k-slab is inside the range and every lane's four elements are either all valid or all
    // outside the matrix (widths that are multiples of 4) -- one predicated 16-byte load per operand quad.  The general load4() with its
    // per-element tails costs ~25 executed instructions per quad even when no tail exists: with it a stage was ~700 instructions per
    // wave for 24 MFMAs and the kernels were instruction-issue bound.
    //
    // The fast loads are UNCONDITIONAL and land in the registers the next store_stage() reads: a quad outside the matrix is fetched from a
    // clamped, valid address instead of being predicated off (its values only reach output rows / columns the epilogue never stores, and
    // the bias column sum of rows it never stores).  With a predicated load (`r = 0; if (ok) r = load`) or a fast / general branch inside
    // the loop the compiler loads into temporaries and merges them with v_mov behind an `s_waitcnt vmcnt(0)` placed BEFORE the MFMA block:
    // every stage then waited for its own prefetch and the kernels ran at memory time PLUS matrix time (52.9 us for dW of layer 1 =
    // 37.3 us without its MFMAs + 15.4 us of MFMAs) instead of the larger of the two.  Hence two copies of the main loop, chosen once.
    // dW of layer 0: the input rows are 235 / 169 wide but stored zero-padded to a multiple of 4 (k_wide_prep), so the last quad is
    // readable too (its extra output column is never stored).
    const int Npad = (MODE == GEMM_DW && ((N.N + 3) & ~3) <= N.ldb) ? ((N.N + 3) & ~3) : N.N;
    const bool quads = (N.M & 3) == 0 && (Npad & 3) == 0 && !(MODE == GEMM_DW && G.gather_b_k) && !(MODE == GEMM_FWD && G.gather_a_rows);
    const bool fast = quads && k_end > k_begin && ((k_end - k_begin) % LG_BK) == 0;
    const float *pa_f = pa, *pb_f = pb;                   // clamped bases of the fast loads
    if (!A_KC && na != 4) pa_f = N.A;
    if (!B_KC && n0 + 4 * (t & 31) + 4 > Npad) pb_f = N.B;
    auto ldq = [&](const float *p) { const f32x4u v = *reinterpret_cast<const f32x4u *>(p); return make_float4(v.x, v.y, v.z, v.w); };
    auto load_fast = [&](int kb) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (A_KC) ra[c] = ldq(pa_f + kb + 16 * (t & 1) + 4 * c);
            else ra[c] = ldq(pa_f + (size_t)(kb + 4 * (t >> 5) + c) * N.lda);
            if (B_KC) rb[c] = ldq(pb_f + kb + 16 * (t & 1) + 4 * c);
            else rb[c] = ldq(pb_f + (size_t)(kb + 4 * (t >> 5) + c) * N.ldb);
        }
    };
    auto load_stage = [&](int kb) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (A_KC) { const int k = kb + 16 * (t & 1) + 4 * c; ra[c] = load4(pa + k, na ? k_end - k : 0); }
            else { const int k = kb + 4 * (t >> 5) + c; ra[c] = load4(pa + (size_t)min(k, k_end - 1) * N.lda, k < k_end ? na : 0); }
            if (B_KC) { const int k = kb + 16 * (t & 1) + 4 * c; rb[c] = load4(pb + k, nb ? k_end - k : 0); }
            else {
                const int k = kb + 4 * (t >> 5) + c;
                size_t row = (size_t)min(k, k_end - 1);
                if (MODE == GEMM_DW && G.gather_b_k) { row = (size_t)brow_next[c]; brow_next[c] = G.rows[min(k + LG_BK, k_end - 1)]; }
                rb[c] = load4(pb + row * N.ldb, k < k_end ? nb : 0);
            }
        }
    };
    float colsum = 0.0f;                                  // DW, first tile column: this thread's share of db (its 4 features x 4 rows per stage)
    float cs4[4] = {0.f, 0.f, 0.f, 0.f};
    auto store_stage = [&]() {
        auto put_kc = [&](__bf16 (*hi)[LG_BLD], __bf16 (*lo)[LG_BLD], const float4 (&v)[4]) {
            const int r = t >> 1, k = 16 * (t & 1);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                bf16x4g h0, l0, h1, l1;
                split4(v[2 * h], h0, l0); split4(v[2 * h + 1], h1, l1);
                *reinterpret_cast<bf16x8g *>(&hi[r][k + 8 * h]) = bf16x8g{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                *reinterpret_cast<bf16x8g *>(&lo[r][k + 8 * h]) = bf16x8g{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
            }
        };
        // operand that is NOT k-contiguous in memory (dW: both; dX: the weights): image [k][feature], rows of LG_TRS bf16.  A thread's
        // float4 (4 consecutive features of one k) is one 8-byte write and a half-wave's 32 writes are 256 contiguous bytes -- the
        // earlier [feature][k] image put the lanes 4 rows = 320 B apart: 2 of 32 banks, every write 16-way conflicted (the GEMMs ran
        // at a fifth of the matrix-core rate).  The MFMA fragments are read back transposed by ds_read_b64_tr_b16 (frag_tr).
        auto put_rc = [&](__bf16 *hi, __bf16 *lo, const float4 (&v)[4]) {
            const int r = 4 * (t & 31), k = 4 * (t >> 5);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                bf16x4g h, l;
                split4(v[c], h, l);
                *reinterpret_cast<bf16x4g *>(hi + (k + c) * LG_TRS + r) = h;
                *reinterpret_cast<bf16x4g *>(lo + (k + c) * LG_TRS + r) = l;
            }
        };
        if (A_KC) put_kc(st.ahi, st.alo, ra); else put_rc(&st.ahi[0][0], &st.alo[0][0], ra);
        if (B_KC) put_kc(st.bhi, st.blo, rb); else put_rc(&st.bhi[0][0], &st.blo[0][0], rb);
        if (MODE == GEMM_DW && tile_n == 0) {             // bias gradient: exact f32 column sums of the G tile this thread just staged
#pragma unroll
            for (int c = 0; c < 4; c++) { cs4[0] += ra[c].x; cs4[1] += ra[c].y; cs4[2] += ra[c].z; cs4[3] += ra[c].w; }
        }
    };
    f32x16g acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int c = 0; c < 16; c++) acc[i][j][c] = 0.0f;
    const int n_stage = (k_end - k_begin + LG_BK - 1) / LG_BK;
    const int li = lane & 31, lh = lane >> 5;
    auto mfma_stage = [&]() {
#pragma unroll
        for (int ks = 0; ks < LG_BK; ks += 16) {
            bf16x8g ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                if (A_KC) {
                    ah[i] = *reinterpret_cast<const bf16x8g *>(&st.ahi[64 * wy + 32 * i + li][ks + 8 * lh]);
                    al[i] = *reinterpret_cast<const bf16x8g *>(&st.alo[64 * wy + 32 * i + li][ks + 8 * lh]);
                } else {
                    ah[i] = frag_tr(&st.ahi[0][0], 64 * wy + 32 * i, ks, lane);
                    al[i] = frag_tr(&st.alo[0][0], 64 * wy + 32 * i, ks, lane);
                }
                if (B_KC) {
                    bh[i] = *reinterpret_cast<const bf16x8g *>(&st.bhi[64 * wx + 32 * i + li][ks + 8 * lh]);
                    bl[i] = *reinterpret_cast<const bf16x8g *>(&st.blo[64 * wx + 32 * i + li][ks + 8 * lh]);
                } else {
                    bh[i] = frag_tr(&st.bhi[0][0], 64 * wx + 32 * i, ks, lane);
                    bl[i] = frag_tr(&st.blo[0][0], 64 * wx + 32 * i, ks, lane);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);     // small terms first
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    };
    if (fast) {
        load_fast(k_begin);
        for (int s = 0; s < n_stage; s++) {
            store_stage();                                        // registers of stage s -> LDS (the previous stage's readers passed the barrier below)
            __syncthreads();
            load_fast(k_begin + min(s + 1, n_stage - 1) * LG_BK); // in flight during this stage's MFMAs (the last stage is fetched twice: no branch)
            __builtin_amdgcn_sched_barrier(0);                    // ... all of them: the scheduler otherwise sinks the loads behind the MFMAs
            mfma_stage();
            __syncthreads();                                      // all fragment reads of this stage done before the next store
        }
    } else {
        if (n_stage > 0) load_stage(k_begin);
        for (int s = 0; s < n_stage; s++) {
            store_stage();
            __syncthreads();
            if (s + 1 < n_stage) load_stage(k_begin + (s + 1) * LG_BK);
            mfma_stage();
            __syncthreads();
        }
    }
    float *C = N.C + (MODE == GEMM_DW ? (size_t)split * N.M * N.ldc : 0);
    if (MODE == GEMM_DW && tile_n == 0) {                         // fold the 8 k-groups (t >> 5) of each feature through LDS, fixed order
        float *red = reinterpret_cast<float *>(&st) + 8192;       // beyond the 32 KB the epilogue's LDS patches use
#pragma unroll
        for (int i = 0; i < 4; i++) red[(t >> 5) * LG_GT + 4 * (t & 31) + i] = cs4[i];
        __syncthreads();
        if (t < LG_GT) {
#pragma unroll
            for (int g = 0; g < 8; g++) colsum += red[g * LG_GT + t];
            if (m0 + t < N.M) C[(size_t)(m0 + t) * N.ldc + N.N] = colsum;
        }
    }
    gemm_epilogue<MODE>(N, C, acc, m0, n0, wy, wx, lane, reinterpret_cast<float *>(&st));
}

// Layer-0 operands in aligned, gather-free form: x0p[m][0 .. K0p) = x[rows[m]][0 .. d0) (zero padded to K0p = round-up of d0 to 4
// floats) and w0p[n][0 .. K0p) likewise.  The input rows of the rough tasks are 235 / 169 floats: 940-byte strides leave every
// 16-byte load of the GEMMs unaligned, and the row gather adds a dependent load; one copy pass per mini-batch removes both from
// the two GEMMs that read the inputs (forward layer 0, dW of layer 0).
struct WidePrepArgs { const float *x[2], *w[2]; float *xp[2], *wp[2]; const int64_t *rows; int d0[2], d1[2], k0p[2], mb; int skip_x[2]; /* the rows of this net are another net's copy */ };
__global__ void __launch_bounds__(256) k_wide_prep(const WidePrepArgs P) {
    const int z = blockIdx.y, kp = P.k0p[z], d0 = P.d0[z], q = kp / 4;
    const size_t nx = (size_t)P.mb * q, nw = (size_t)P.d1[z] * q;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nx + nw; i += (size_t)gridDim.x * 256) {
        const bool isx = i < nx;
        if (isx && P.skip_x[z]) continue;
        const size_t j = isx ? i : i - nx;
        const int r = (int)(j / q), c = 4 * (int)(j % q);
        const float *src = isx ? P.x[z] + (size_t)(P.rows ? P.rows[r] : r) * d0 : P.w[z] + (size_t)r * d0;
        float4 v = load4(src + c, d0 - c);
        *reinterpret_cast<float4 *>((isx ? P.xp[z] : P.wp[z]) + (size_t)r * kp + c) = v;
    }
}

// dW / db from the per-split partials [splits][N][K + 1] in a fixed order: gw[n][k] (torch layout), gb[n].  `group` lanes share one
// element (lane q sums the splits q, q + group, ... in order, then a fixed xor butterfly): 1 for the big layers (coalesced, bandwidth-bound),
// 8 for the narrow output layer, whose 256 chunks x 1.6 k elements are pure load latency (12.9 us with one lane per element and 8 loads
// in flight, 8.0 us with 16).  Either way the order of the additions is fixed: the gradients are bit-reproducible.
#define LG_REDUCE_JOBS 8               // 4 layers x 2 nets: ONE launch at the end of the backward pass (four launches of 5-13 us, one of them pure
                                       // latency, were 30 us per mini-batch)
struct WideReduceJob { const float *part; float *gw, *gb; int N, K, ld, splits, group; };     // ld: row stride of a partial (K + 1 rounded up to 4)
struct WideReduceArgs { WideReduceJob job[LG_REDUCE_JOBS]; };
__global__ void __launch_bounds__(256) k_wide_reduce(const WideReduceArgs R) {
    const WideReduceJob &J = R.job[blockIdx.y];
    const int N = J.N, K = J.K, ld = J.ld, G = J.group;
    if (G <= 0) return;
    const int t = blockIdx.x * 256 + threadIdx.x, i = t / G, q = t % G;
    if ((blockIdx.x * 256) / G >= N * ld) return;   // whole workgroup beyond this job (the grid is sized for the largest)
    const bool in = i < N * ld;                     // (whole groups are in or out: 256 is a multiple of G)
    const int n = in ? i / ld : 0, k = in ? i % ld : 0;
    float s = 0.0f;
    if (in && k <= K) {
        const size_t stride = (size_t)N * ld;
        const float *src = J.part + i;
        const int S = J.splits;
        int p = q;
        for (; p + 15 * G < S; p += 16 * G) {       // 16 independent loads in flight
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = src[(size_t)(p + u * G) * stride];
#pragma unroll
            for (int u = 0; u < 16; u++) s += v[u];
        }
        for (; p < S; p += G) s += src[(size_t)p * stride];
    }
    for (int o = 1; o < G; o <<= 1) s += __shfl_xor(s, o);
    if (!in || k > K || q != 0) return;
    if (k < K) J.gw[(size_t)n * K + k] = s;
    else J.gb[n] = s;
}


// ------------------------------------------------------------------ backward of a NARROW output layer (12 actions / 1 value) in one pass
// G_prev = (dZ W) * elu'(A_prev) and the dW / db partials from ONE read of A_prev: the tiled GEMMs pad the <= 16 outputs to a 128-wide tile
// (dW 25 us + dX 13 us for 0.4 GFLOP).  Plain FMAs, bound by the 2 x 12.6 MB per net it moves.  A workgroup owns a contiguous chunk of rows;
// 32 lanes cover a row's K = 128-wide previous layer (4 features each), 8 rows per pass.  Partials leave in k_wide_reduce's layout
// ([chunk][N][ld], column K = the bias gradient), summed there in a fixed order.
#define LG_OUT_MAXN 16
struct OutBwdNet { const float *dz; const float *w; const float *act; float *g; float *part; int N, K, ld, chunks, rows_per_chunk; };
struct OutBwdArgs { OutBwdNet net[2]; int mb; };
template <int NMAX>
__global__ void __launch_bounds__(256) k_wide_out_bwd(const OutBwdArgs O) {
    const OutBwdNet &Q = O.net[blockIdx.y];
    if ((int)blockIdx.x >= Q.chunks) return;
    const int t = threadIdx.x, c = t & 31, rg = t >> 5;            // c: features 4c .. 4c+3; rg: which of the 8 rows of a pass
    __shared__ float red[8][NMAX][132];
    float wreg[NMAX][4], acc[NMAX][4], accb[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; n++) {
        const float4 w = n < Q.N ? *reinterpret_cast<const float4 *>(Q.w + (size_t)n * Q.K + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
        wreg[n][0] = w.x; wreg[n][1] = w.y; wreg[n][2] = w.z; wreg[n][3] = w.w;
        acc[n][0] = acc[n][1] = acc[n][2] = acc[n][3] = 0.0f; accb[n] = 0.0f;
    }
    const int r_begin = blockIdx.x * Q.rows_per_chunk, r_end = min(O.mb, r_begin + Q.rows_per_chunk);
    auto fetch = [&](int r, float4 &a, float (&dz)[NMAX]) {        // row r of this thread's pass (clamped: the tail re-reads a valid row, unused)
        const int rr = min(r, r_end - 1);
        a = *reinterpret_cast<const float4 *>(Q.act + (size_t)rr * Q.K + 4 * c);
#pragma unroll
        for (int n = 0; n < NMAX; n++) dz[n] = n < Q.N ? Q.dz[(size_t)rr * Q.N + n] : 0.0f;
    };
    float4 a_nx; float dz_nx[NMAX];
    if (r_begin + rg < r_end) fetch(r_begin + rg, a_nx, dz_nx);
    for (int r = r_begin + rg; r < r_end; r += 8) {
        const float4 a = a_nx;
        float dz[NMAX];
#pragma unroll
        for (int n = 0; n < NMAX; n++) dz[n] = dz_nx[n];
        fetch(r + 8, a_nx, dz_nx);                                 // the next pass's row is in flight while this one is computed
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int n = 0; n < NMAX; n++) {
            s0 = fmaf(dz[n], wreg[n][0], s0); s1 = fmaf(dz[n], wreg[n][1], s1); s2 = fmaf(dz[n], wreg[n][2], s2); s3 = fmaf(dz[n], wreg[n][3], s3);
            acc[n][0] = fmaf(dz[n], a.x, acc[n][0]); acc[n][1] = fmaf(dz[n], a.y, acc[n][1]);
            acc[n][2] = fmaf(dz[n], a.z, acc[n][2]); acc[n][3] = fmaf(dz[n], a.w, acc[n][3]);
            accb[n] += dz[n];
        }
        // elu'(x) from the stored post-activation a: 1 for a > 0, a + 1 otherwise
        *reinterpret_cast<float4 *>(Q.g + (size_t)r * Q.K + 4 * c) =
            make_float4(s0 * (a.x > 0.f ? 1.f : a.x + 1.f), s1 * (a.y > 0.f ? 1.f : a.y + 1.f), s2 * (a.z > 0.f ? 1.f : a.z + 1.f), s3 * (a.w > 0.f ? 1.f : a.w + 1.f));
    }
#pragma unroll
    for (int n = 0; n < NMAX; n++) {
        *reinterpret_cast<float4 *>(&red[rg][n][4 * c]) = make_float4(acc[n][0], acc[n][1], acc[n][2], acc[n][3]);
        if (c == 0) red[rg][n][128] = accb[n];
    }
    __syncthreads();
    float *part = Q.part + (size_t)blockIdx.x * Q.N * Q.ld;
    for (int i = t; i < Q.N * (Q.K + 1); i += 256) {
        const int n = i / (Q.K + 1), k = i - n * (Q.K + 1);
        float v = 0.0f;
#pragma unroll
        for (int g8 = 0; g8 < 8; g8++) v += red[g8][n][k];
        part[(size_t)n * Q.ld + k] = v;
    }
}

}  // namespace lg
