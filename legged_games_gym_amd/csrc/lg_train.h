// lg_train.h -- the learner half of the rollout+update loop: PPO mini-batch MLP forward / loss / backward on the matrix cores
// (k_mlp_train, k_mlp_reduce), gradient clip + Adam (k_adam_*), rollout bookkeeping (k_rollout_record).
//
// Stands in for the autograd pass over rsl_rl's ActorCritic MLPs ([EXTERNAL]; Linear/ELU x3 + Linear, dims from reference
// legged_robot_config.py:204-209) inside PPO.update(): y = net(x[rows]) and, given dL/dy, the gradients of all weights
// and biases.  The weights are read in torch's own [out, in] layout (they change every optimiser step, so nothing is
// re-packed): each persistent workgroup copies them ONCE into LDS (72 KB, rows padded by 4 floats against bank conflicts)
// and then walks its row tiles with every operand coming from LDS -- a wave per SIMD cannot hide L2 latency eight times
// per row tile.  Activations never leave LDS, and the weight gradients are accumulated in MFMA accumulators across all
// the row tiles a workgroup walks, then written once as a per-workgroup partial that k_mlp_reduce sums in a fixed order
// (deterministic, no atomics).  Builds of the one kernel template: forward only (lg_mlp_forward), backward from a given dL/dy
// (lg_mlp_backward), and forward + PPO loss + backward in one pass (lg_ppo_minibatch, the one PPO.update uses).
//
// Tile algebra (v_mfma_f32_16x16x4_f32, D[i][j] += A[i][k] B[k][j]; A: lane l holds A[l&15][l>>4], B: lane l holds
// B[l>>4][l&15], D: lane l holds D[4(l>>4)+c][l&15]):
//   forward   x_{l+1}[f][row] = W[f][k] x_l[k][row]           A = W (float4 per lane per input tile), B = x_l tile
//   backward  dx_l[i][row]    = W^T[i][f] g_{l+1}[f][row]     A = W read column-wise,                  B = g tile
//   weights   dW[f][i]        = g_{l+1}[f][row] x_l[i][row]   A = g^T, B = x^T (feature-major copies kept in LDS, padded)
//   biases    db[f]           = g_{l+1}[f][row] * 1           same as dW with an all-ones B operand (an extra "input tile")
// where g is the gradient w.r.t. the pre-activation (ELU' folded in from the stored post-activation: x > 0 ? 1 : x + 1).
#pragma once
#include <type_traits>
#include "lg_policy.h"

namespace lg {

#define LG_TRAIN_WAVES 4
#define LG_TT 20                       // row stride of the feature-major tile copies: 16-byte aligned quads, <= 2-way bank conflicts
#define LG_WPAD 4                      // padding floats per weight row in LDS

struct MlpNetArgs {
    const float *w[4], *b[4];          // torch Linear weights [out, in] / biases
    const float *x;                    // [R, dims[0]] input rows
    float *y;                          // forward: [mb, dims[4]]
    const float *dy;                   // backward: [mb, dims[4]]
    float *partial;                    // backward: [groups][part_stride] per-group partial sums (gradients, then LG_PPO_EXTRA loss terms)
    int32_t dims[5];
    int32_t grad_floats;               // sum over layers of out*in + out
    int32_t part_stride;
};
// PPO loss terms of rsl_rl PPO.update ([EXTERNAL]; same expressions as k_ppo_loss), evaluated inside the backward kernel
#define LG_PPO_EXTRA 20                // per-group partials after the gradients: d_std[16], then sums of {surrogate, value loss, KL, -}
struct PpoArgs {
    const float *actions, *old_lp, *old_mu, *old_sigma, *adv, *old_values, *returns;   // rollout storage, indexed by rows[i]
    const float *std;                  // [A] policy std parameter
    float clip, vcoef, inv_n;
    int32_t clipped_value;
};
struct MlpArgs {
    MlpNetArgs net[2];
    const int64_t *rows;               // [mb] row indices into x, or null for 0..mb-1
    int32_t mb, n_tiles;
    PpoArgs ppo;                       // LOSS build only
    unsigned long long *trace;         // diagnostic: s_memtime stamps of workgroup (0, 0), thread 0 (lg_mlp_trace); null in normal use
};

// LDS image of one Linear layer: rows 16*OUT_T (zero beyond out_dim), IN_T*16 columns (zero beyond in_dim) + LG_WPAD, then the bias
template <int IN_T, int OUT_T> struct LdsLayer {
    static constexpr int stride = 16 * IN_T + LG_WPAD, w_floats = 16 * OUT_T * stride, floats = w_floats + 16 * OUT_T;
};
// The copy is split in two so that the global loads of ALL layers are in flight before the first LDS store waits for one:
// quads i = tid, tid + NT, ... of the padded [16*OUT_T][16*IN_T] image (+ one bias element per thread).
template <int IN_T, int OUT_T, int NT> struct LdsFill {
    using L = LdsLayer<IN_T, OUT_T>;
    static constexpr int QPR = 4 * IN_T, NQ = 16 * OUT_T * QPR, PER = (NQ + NT - 1) / NT;   // quads per row / per layer / per thread
    static_assert(16 * OUT_T <= NT, "one bias element per thread");
    float4 q[PER];
    float bias;
    LG_DEV void load(const float *__restrict__ W, const float *__restrict__ b, int in_dim, int out_dim, int tid) {
        const bool vec = (in_dim & 3) == 0;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int i = tid + u * NT, row = i / QPR, col = 4 * (i % QPR);
            q[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < NQ && row < out_dim) {
                const float *src = W + (size_t)row * in_dim + col;
                if (vec && col + 3 < in_dim) q[u] = *reinterpret_cast<const float4 *>(src);
                else {
                    if (col + 0 < in_dim) q[u].x = src[0];
                    if (col + 1 < in_dim) q[u].y = src[1];
                    if (col + 2 < in_dim) q[u].z = src[2];
                    if (col + 3 < in_dim) q[u].w = src[3];
                }
            }
        }
        bias = tid < out_dim ? b[tid] : 0.0f;
    }
    LG_DEV void store(float *wl, int tid) const {
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int i = tid + u * NT, row = i / QPR, col = 4 * (i % QPR);
            if (i < NQ) *reinterpret_cast<float4 *>(wl + row * L::stride + col) = q[u];
        }
        if (tid < 16 * OUT_T) wl[L::w_floats + tid] = bias;
    }
};

// feature-major copy of a tile: xt[feature 16][LG_TT] <- lane (row l&15, group g) holds features 4g..4g+3
LG_DEV void tile_store_t(float (*xt)[16][LG_TT], int tile, int lane, float4 v) {
    const int g = lane >> 4, r = lane & 15;
    xt[tile][4 * g + 0][r] = v.x; xt[tile][4 * g + 1][r] = v.y; xt[tile][4 * g + 2][r] = v.z; xt[tile][4 * g + 3][r] = v.w;
}

// MFMA B operand / D-layout view of a feature-major tile: lane (row r = l&15, group g) <- features 4g..4g+3 of row r
// (4 ds_read_b32 at stride LG_TT: within a 32-lane half the banks are r and 16 + r -- conflict-free)
LG_DEV float4 tile_load_b(const float (*xt)[16][LG_TT], int tile, int lane) {
    const int g = lane >> 4, r = lane & 15;
    return make_float4(xt[tile][4 * g + 0][r], xt[tile][4 * g + 1][r], xt[tile][4 * g + 2][r], xt[tile][4 * g + 3][r]);
}

// backward build: activations exist only feature-major (the B-operand copies would not leave room for two row tiles in flight)
template <int IN_T, int OUT_T>
LG_DEV void train_forward_layer_t(const float *wl, const float (*xinT)[16][LG_TT], float (*xoutT)[16][LG_TT], int wave, int lane) {
    using L = LdsLayer<IN_T, OUT_T>;
    const int g = lane >> 4;
#pragma unroll
    for (int o = wave; o < OUT_T; o += LG_TRAIN_WAVES) {
        const float *wr = wl + (16 * o + (lane & 15)) * L::stride + 4 * g;
        const float4 bv = *reinterpret_cast<const float4 *>(wl + L::w_floats + 16 * o + 4 * g);
        f32x4 acc = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int t = 0; t < IN_T; t++) {
            const float4 wv = *reinterpret_cast<const float4 *>(wr + 16 * t);
            const float4 xv = tile_load_b(xinT, t, lane);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, xv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, xv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, xv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, xv.w, acc, 0, 0, 0);
        }
        tile_store_t(xoutT, o, lane, make_float4(elu1(acc[0]), elu1(acc[1]), elu1(acc[2]), elu1(acc[3])));
    }
}

// x_out = act(W x_in + b), all operands in LDS
template <int IN_T, int OUT_T, bool ACT>
LG_DEV void train_forward_layer(const float *wl, const float4 (*xin)[64], float4 (*xout)[64], int wave, int lane) {
    using L = LdsLayer<IN_T, OUT_T>;
    const int g = lane >> 4;
#pragma unroll
    for (int o = wave; o < OUT_T; o += LG_TRAIN_WAVES) {
        const float *wr = wl + (16 * o + (lane & 15)) * L::stride + 4 * g;
        const float4 bv = *reinterpret_cast<const float4 *>(wl + L::w_floats + 16 * o + 4 * g);
        f32x4 acc = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int t = 0; t < IN_T; t++) {
            const float4 wv = *reinterpret_cast<const float4 *>(wr + 16 * t);
            const float4 xv = xin[t][lane];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, xv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, xv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, xv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, xv.w, acc, 0, 0, 0);
        }
        const float4 res = ACT ? make_float4(elu1(acc[0]), elu1(acc[1]), elu1(acc[2]), elu1(acc[3])) : make_float4(acc[0], acc[1], acc[2], acc[3]);
        xout[o][lane] = res;
    }
}

// g_in = (W^T g_out) * elu'(x_in): OUT_T gradient tiles -> IN_T gradient tiles (x_in = stored post-activation of the layer input);
// all tiles feature-major
template <int IN_T, int OUT_T>
LG_DEV void train_backward_layer(const float *wl, const float (*goutT)[16][LG_TT], const float (*xinT)[16][LG_TT],
                                 float (*ginT)[16][LG_TT], int wave, int lane) {
    using L = LdsLayer<IN_T, OUT_T>;
    const int g = lane >> 4;
#pragma unroll
    for (int ti = wave; ti < IN_T; ti += LG_TRAIN_WAVES) {
        const float *wc = wl + 4 * g * L::stride + 16 * ti + (lane & 15);      // column of W = row of W^T
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < OUT_T; o++) {
            const float4 gv = tile_load_b(goutT, o, lane);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[(16 * o + 0) * L::stride], gv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[(16 * o + 1) * L::stride], gv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[(16 * o + 2) * L::stride], gv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[(16 * o + 3) * L::stride], gv.w, acc, 0, 0, 0);
        }
        const float4 xv = tile_load_b(xinT, ti, lane);
        tile_store_t(ginT, ti, lane, make_float4(acc[0] * (xv.x > 0.f ? 1.f : xv.x + 1.f), acc[1] * (xv.y > 0.f ? 1.f : xv.y + 1.f),
                                                 acc[2] * (xv.z > 0.f ? 1.f : xv.z + 1.f), acc[3] * (xv.w > 0.f ? 1.f : xv.w + 1.f)));
    }
}

// number of (out tile, in tile | ones) accumulator pairs of a layer one wave owns
template <int IN_T, int OUT_T> struct PairCount { static constexpr int total = OUT_T * (IN_T + 1), per_wave = (total + LG_TRAIN_WAVES - 1) / LG_TRAIN_WAVES; };

// dW[o][t] += g^T x^T over this row tile; pair p = wave + 4 i lives in acc[i]
template <int IN_T, int OUT_T>
LG_DEV void train_weight_grad(const float (*gT)[16][LG_TT], const float (*xT)[16][LG_TT], f32x4 *acc, int wave, int lane) {
    using PC = PairCount<IN_T, OUT_T>;
    const int f = lane & 15, r0 = 4 * (lane >> 4);
#pragma unroll
    for (int i = 0; i < PC::per_wave; i++) {
        const int p = wave + LG_TRAIN_WAVES * i;
        if (p < PC::total) {
            const int o = p / (IN_T + 1), t = p % (IN_T + 1);
            const float4 ga = *reinterpret_cast<const float4 *>(&gT[o][f][r0]);
            float4 xb = make_float4(1.f, 1.f, 1.f, 1.f);
            if (t < IN_T) xb = *reinterpret_cast<const float4 *>(&xT[t][f][r0]);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga.x, xb.x, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga.y, xb.y, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga.z, xb.z, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga.w, xb.w, acc[i], 0, 0, 0);
        }
    }
}

// write a layer's accumulators to the partial buffer in torch layout: [W (out x in) | b (out)]
template <int IN_T, int OUT_T>
LG_DEV void train_flush(const f32x4 *acc, float *__restrict__ part, int in_dim, int out_dim, int wave, int lane) {
    using PC = PairCount<IN_T, OUT_T>;
#pragma unroll
    for (int i = 0; i < PC::per_wave; i++) {
        const int p = wave + LG_TRAIN_WAVES * i;
        if (p < PC::total) {
            const int o = p / (IN_T + 1), t = p % (IN_T + 1);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int fo = 16 * o + 4 * (lane >> 4) + c;
                if (fo >= out_dim) continue;
                if (t < IN_T) {
                    const int fi = 16 * t + (lane & 15);
                    if (fi < in_dim) part[(size_t)fo * in_dim + fi] = acc[i][c];
                } else if ((lane & 15) == 0) {
                    part[(size_t)out_dim * in_dim + fo] = acc[i][c];
                }
            }
        }
    }
}

// LDS budget of k_mlp_train in floats (dynamic shared memory; 72 KB of weights + 17 KB (forward) / 40 KB (backward) per row tile in flight)
template <int D0T, int D1T, int D2T, int D3T, bool BWD, int SLOTS> struct TrainLds {
    static constexpr int XT = D0T + D1T + D2T + D3T, GT = D1T + D2T + D3T + 1;
    static constexpr int w0 = 0, w1 = w0 + LdsLayer<D0T, D1T>::floats, w2 = w1 + LdsLayer<D1T, D2T>::floats, w3 = w2 + LdsLayer<D2T, D3T>::floats,
                         slots = w3 + LdsLayer<D3T, 1>::floats;                  // weights, then SLOTS row-tile work areas:
    // forward build: activations as MFMA B operands [tile][lane] float4; backward build: activations and gradients feature-major
    static constexpr int x = 0, xT = 0, gT = XT * 16 * LG_TT,
                         slot_floats = BWD ? (XT + GT) * 16 * LG_TT : XT * 256, floats = slots + SLOTS * slot_floats;
};

// One persistent workgroup per (slice blockIdx.x, net blockIdx.y): weights -> LDS once, then it walks its row tiles.  SLOTS
// groups of LG_TRAIN_WAVES waves each work on their own row tile (own activation area, shared weights): with several waves
// per SIMD one group's MFMAs overlap another's LDS traffic and ELUs.  All groups run the same barrier sequence.
// LOSS (with BWD): net 0 is the actor, net 1 the critic; dL/dy is not read but computed from the PPO loss after the forward pass.
template <int D0T, int D1T, int D2T, int D3T, bool BWD, int SLOTS, bool LOSS = false>
__global__ void __launch_bounds__(64 * LG_TRAIN_WAVES * SLOTS) k_mlp_train(const MlpArgs A) {
    static_assert(!LOSS || BWD, "the fused loss feeds the backward pass");
    using S = TrainLds<D0T, D1T, D2T, D3T, BWD, SLOTS>;
    constexpr int X0 = 0, X1 = D0T, X2 = D0T + D1T, X3 = D0T + D1T + D2T;        // activation tile offsets
    constexpr int G1 = 0, G2 = D1T, G3 = D1T + D2T, G4 = D1T + D2T + D3T;        // gradient tiles w.r.t. x1, x2, x3 pre-acts, and y
    extern __shared__ float4 lds_raw[];
    float *lds = reinterpret_cast<float *>(lds_raw);
    float *wl0 = lds + S::w0, *wl1 = lds + S::w1, *wl2 = lds + S::w2, *wl3 = lds + S::w3;
    const int slot = threadIdx.x / (64 * LG_TRAIN_WAVES);
    float *area = lds + S::slots + slot * S::slot_floats;
    float4 (*x)[64] = reinterpret_cast<float4 (*)[64]>(area + S::x);
    float (*xT)[16][LG_TT] = reinterpret_cast<float (*)[16][LG_TT]>(area + S::xT);
    float (*gT)[16][LG_TT] = reinterpret_cast<float (*)[16][LG_TT]>(area + S::gT);
    // phase stamps of workgroup (0, 0), thread 0: -DLG_PROFILE builds only (tools/profile_sections.py build; TRACE=1 tools/mlp_probe.py)
#ifdef LG_PROFILE
    unsigned long long *tr = (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) ? A.trace : nullptr;
    int tri = 0;
#define LG_TR() do { if (tr && tri < 60) tr[tri++] = __builtin_readcyclecounter(); } while (0)
#else
#define LG_TR() do { } while (0)
#endif
#ifdef LG_PROFILE
    // (diagnostic build: these few registers cost the fused kernel 2.4 us) every workgroup also leaves its start / end on the 100 MHz wall clock
    // behind the 64 stamps: [64 + 2 * wg], [65 + 2 * wg]; tools/mlp_probe.py prints the spread
    unsigned long long *wgclk = (A.trace && threadIdx.x == 0) ? A.trace + 64 + 2 * (blockIdx.y * gridDim.x + blockIdx.x) : nullptr;
    if (wgclk) wgclk[0] = wall_clock64();
#endif
    LG_TR();
    const MlpNetArgs &N = A.net[blockIdx.y];
    // role of this wave within its group, rotated by the group index: the thin layers (2 and 1 output tiles) land on different SIMDs
    const int wave = ((threadIdx.x >> 6) + slot) % LG_TRAIN_WAVES, lane = threadIdx.x & 63, g = lane >> 4;
    const int d0 = N.dims[0], d1 = N.dims[1], d2 = N.dims[2], d3 = N.dims[3], d4 = N.dims[4];
    using P0 = PairCount<D0T, D1T>; using P1 = PairCount<D1T, D2T>; using P2 = PairCount<D2T, D3T>; using P3 = PairCount<D3T, 1>;
    f32x4 a0[BWD ? P0::per_wave : 1], a1[BWD ? P1::per_wave : 1], a2[BWD ? P2::per_wave : 1], a3[BWD ? P3::per_wave : 1];
    if (BWD) {
#pragma unroll
        for (int i = 0; i < P0::per_wave; i++) a0[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < P1::per_wave; i++) a1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < P2::per_wave; i++) a2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < P3::per_wave; i++) a3[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // The inputs of the group's next row tile are requested while the current one is computed, and the row indices one tile
    // further ahead: nothing else hides the gather's two dependent global loads (row index, then the row).
    static_assert(D0T <= LG_TRAIN_WAVES, "one input tile per wave");
    float4 xv_next = make_float4(0.f, 0.f, 0.f, 0.f), dy_next = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t src_next = 0;                                          // storage row of this lane in the tile requested NEXT
    // LOSS: the last role of each group evaluates the loss; its per-row inputs are gathered with the rows, one tile ahead
    const bool actor = blockIdx.y == 0;
    float4 act_next = dy_next, omu_next = dy_next, osg_next = dy_next;
    float s0_next = 0.f, s1_next = 0.f;                            // actor: old log-prob, advantage; critic: old value, return
    float gstd[4] = {0.f, 0.f, 0.f, 0.f}, sum_sur = 0.f, sum_kl = 0.f, sum_val = 0.f;
    float sg[4] = {1.f, 1.f, 1.f, 1.f}, isg[4] = {1.f, 1.f, 1.f, 1.f}, lsg[4] = {0.f, 0.f, 0.f, 0.f};
    if (LOSS && actor && wave == LG_TRAIN_WAVES - 1) {
#pragma unroll
        for (int c = 0; c < 4; c++) if (4 * g + c < d4) { sg[c] = A.ppo.std[4 * g + c]; isg[c] = 1.0f / sg[c]; lsg[c] = __logf(sg[c]); }
    }
    const bool vec_in = (d0 & 3) == 0;
    auto row_index = [&](int rt) -> int64_t {                      // rows past the batch re-read the last row (their dL/dy is zero)
        const int r = min(rt * 16 + (lane & 15), A.mb - 1);
        return A.rows ? A.rows[r] : (int64_t)r;
    };
    auto request = [&](int rt, int64_t src) {
        if (wave < D0T) {
            const float *xr = N.x + (size_t)src * d0 + 16 * wave + 4 * g;
            const int k = 16 * wave + 4 * g;
            if (vec_in && k + 3 < d0) xv_next = *reinterpret_cast<const float4 *>(xr);
            else xv_next = make_float4(k < d0 ? xr[0] : 0.f, k + 1 < d0 ? xr[1] : 0.f, k + 2 < d0 ? xr[2] : 0.f, k + 3 < d0 ? xr[3] : 0.f);
        }
        if (BWD && !LOSS && wave == LG_TRAIN_WAVES - 1) {          // dL/dy tile (zero for rows past the batch: they contribute nothing)
            const int r = rt * 16 + (lane & 15);
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; c++) { const int k = 4 * g + c; v[c] = (r < A.mb && k < d4) ? N.dy[(size_t)r * d4 + k] : 0.0f; }
            dy_next = make_float4(v[0], v[1], v[2], v[3]);
        }
        if (LOSS && wave == LG_TRAIN_WAVES - 1) {
            if (actor) {
                const size_t o = (size_t)src * d4 + 4 * g;
                auto quad = [&](const float *p) {
                    return ((d4 & 3) == 0 && 4 * g + 3 < d4) ? *reinterpret_cast<const float4 *>(p + o)
                         : make_float4(4 * g < d4 ? p[o] : 0.f, 4 * g + 1 < d4 ? p[o + 1] : 0.f, 4 * g + 2 < d4 ? p[o + 2] : 0.f, 4 * g + 3 < d4 ? p[o + 3] : 0.f);
                };
                act_next = quad(A.ppo.actions); omu_next = quad(A.ppo.old_mu); osg_next = quad(A.ppo.old_sigma);
                s0_next = A.ppo.old_lp[src]; s1_next = A.ppo.adv[src];
            } else {
                s0_next = A.ppo.old_values[src]; s1_next = A.ppo.returns[src];
            }
        }
    };
    const int stride = gridDim.x * SLOTS, n_iter = (A.n_tiles + stride - 1) / stride;     // uniform trip count: barriers inside
    int rt = blockIdx.x * SLOTS + slot;
    // Vector memory loads return in order: the weights are requested FIRST, the first tile's gather (row index, then the row: two dependent
    // trips, the only gather that waits for its index in line) behind them -- the other way round the 72 KB of weights could not be stored
    // to LDS before the gather had come back (prologue 9.7 us of the kernel's 62, phase stamps of the -DLG_PROFILE build).
    {
        constexpr int NT = 64 * LG_TRAIN_WAVES * SLOTS;
        LdsFill<D0T, D1T, NT> f0; LdsFill<D1T, D2T, NT> f1; LdsFill<D2T, D3T, NT> f2; LdsFill<D3T, 1, NT> f3;
        f0.load(N.w[0], N.b[0], d0, d1, threadIdx.x); f1.load(N.w[1], N.b[1], d1, d2, threadIdx.x);
        f2.load(N.w[2], N.b[2], d2, d3, threadIdx.x); f3.load(N.w[3], N.b[3], d3, d4, threadIdx.x);
        const int64_t src0 = rt < A.n_tiles ? row_index(rt) : 0;
        if (rt + stride < A.n_tiles) src_next = row_index(rt + stride);
        LG_TR();
        f0.store(wl0, threadIdx.x); f1.store(wl1, threadIdx.x); f2.store(wl2, threadIdx.x); f3.store(wl3, threadIdx.x);
        if (rt < A.n_tiles) request(rt, src0);
    }
    LG_TR();
    // The row-tile loop is compiled once per role: with the role a compile-time constant every tile index and LDS tile address
    // below is an immediate instead of per-access integer arithmetic (the kernel is instruction-issue bound: VALU : MFMA was 3 : 1).
    // The role branch is wave-uniform and every copy runs the same barrier sequence.
    auto row_tiles = [&](auto role) {
        constexpr int wave = decltype(role)::value;                // shadows the runtime role
    #pragma unroll 1
        for (int it = 0; it < n_iter; it++, rt += stride) {
            const bool active = rt < A.n_tiles;
            const int r = rt * 16 + (lane & 15);
            const bool live = active && r < A.mb;
            const float4 dyv = active ? dy_next : make_float4(0.f, 0.f, 0.f, 0.f);   // an idle group still runs the barriers; it adds zeros
            const float4 actv = act_next, omuv = omu_next, osgv = osg_next;
            const float s0v = s0_next, s1v = s1_next;
            if (wave < D0T) {
                if (BWD) tile_store_t(xT + X0, wave, lane, xv_next); else x[X0 + wave][lane] = xv_next;
            }
            LG_TR();
            if (rt + stride < A.n_tiles) request(rt + stride, src_next);           // its index was loaded an iteration ago
            if (rt + 2 * stride < A.n_tiles) src_next = row_index(rt + 2 * stride);
            LG_TR();
            __syncthreads();                                           // (first pass: also the weights are in LDS)
            LG_TR();
            if (BWD) train_forward_layer_t<D0T, D1T>(wl0, xT + X0, xT + X1, wave, lane);
            else train_forward_layer<D0T, D1T, true>(wl0, x + X0, x + X1, wave, lane);
            LG_TR();
            __syncthreads();
            LG_TR();
            if (BWD) train_forward_layer_t<D1T, D2T>(wl1, xT + X1, xT + X2, wave, lane);
            else train_forward_layer<D1T, D2T, true>(wl1, x + X1, x + X2, wave, lane);
            LG_TR();
            __syncthreads();
            LG_TR();
            if (BWD) train_forward_layer_t<D2T, D3T>(wl2, xT + X2, xT + X3, wave, lane);
            else train_forward_layer<D2T, D3T, true>(wl2, x + X2, x + X3, wave, lane);
            if (BWD && !LOSS && wave == LG_TRAIN_WAVES - 1) tile_store_t(gT + G4, 0, lane, dyv);
            LG_TR();
            __syncthreads();
            LG_TR();
            if (LOSS) {
                if (wave == LG_TRAIN_WAVES - 1) {
                    // output layer in registers: lane (row l&15, group g) holds y[4g + c]
                    using L3 = LdsLayer<D3T, 1>;
                    const float4 bv = *reinterpret_cast<const float4 *>(wl3 + L3::w_floats + 4 * g);
                    f32x4 y = {bv.x, bv.y, bv.z, bv.w};
    #pragma unroll
                    for (int t = 0; t < D3T; t++) {
                        const float4 wv = *reinterpret_cast<const float4 *>(wl3 + (lane & 15) * L3::stride + 16 * t + 4 * g);
                        const float4 xv = tile_load_b(xT + X3, t, lane);
                        y = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, xv.x, y, 0, 0, 0);
                        y = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, xv.y, y, 0, 0, 0);
                        y = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, xv.z, y, 0, 0, 0);
                        y = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, xv.w, y, 0, 0, 0);
                    }
                    const PpoArgs &P = A.ppo;
                    float d[4] = {0.f, 0.f, 0.f, 0.f};
                    if (actor) {
                        const float av[4] = {actv.x, actv.y, actv.z, actv.w}, om[4] = {omuv.x, omuv.y, omuv.z, omuv.w}, os[4] = {osgv.x, osgv.y, osgv.z, osgv.w};
                        float z[4], lp = 0.0f, kl = 0.0f;
    #pragma unroll
                        for (int c = 0; c < 4; c++) {
                            z[c] = 0.0f;
                            if (4 * g + c < d4) {
                                z[c] = (av[c] - y[c]) * isg[c];
                                lp += -0.5f * z[c] * z[c] - lsg[c] - 0.918938533f;                     // log N(a; mu, sigma)
                                kl += __logf(sg[c] / os[c] + 1.0e-5f) + (os[c] * os[c] + (om[c] - y[c]) * (om[c] - y[c])) * (0.5f * isg[c] * isg[c]) - 0.5f;
                            }
                        }
                        lp += __shfl_xor(lp, 16); lp += __shfl_xor(lp, 32);                            // the row's actions live in 4 lanes
                        kl += __shfl_xor(kl, 16); kl += __shfl_xor(kl, 32);
                        const float ad = s1v, ratio = __expf(lp - s0v);
                        const float t1 = -ad * ratio, t2 = -ad * fminf(fmaxf(ratio, 1.0f - P.clip), 1.0f + P.clip);
                        const bool inside = ratio >= 1.0f - P.clip && ratio <= 1.0f + P.clip;
                        const float dlp = (t1 > t2 || inside) ? -ad * ratio : (t1 == t2 ? -0.5f * ad * ratio : 0.0f);
                        if (live) {
    #pragma unroll
                            for (int c = 0; c < 4; c++) if (4 * g + c < d4) {
                                d[c] = P.inv_n * dlp * z[c] * isg[c];                                  // d lp / d mu = (a - mu) / sigma^2
                                gstd[c] += P.inv_n * dlp * (z[c] * z[c] - 1.0f) * isg[c];              // d lp / d sigma
                            }
                            if (g == 0) { sum_sur += fmaxf(t1, t2); sum_kl += kl; }
                        }
                    } else if (g == 0) {
                        const float v = y[0], tv = s0v, R = s1v;
                        float dv, vl;
                        if (P.clipped_value) {
                            const float dvt = v - tv, vc = tv + fminf(fmaxf(dvt, -P.clip), P.clip);
                            const float v1 = (v - R) * (v - R), v2 = (vc - R) * (vc - R);
                            const bool in_v = dvt >= -P.clip && dvt <= P.clip;
                            vl = fmaxf(v1, v2);
                            dv = (v1 > v2 || in_v) ? 2.0f * (v - R) : (v1 == v2 ? (v - R) : 0.0f);
                        } else {
                            vl = (R - v) * (R - v);
                            dv = 2.0f * (v - R);
                        }
                        if (live) { d[0] = P.vcoef * P.inv_n * dv; sum_val += vl; }
                    }
                    tile_store_t(gT + G4, 0, lane, make_float4(d[0], d[1], d[2], d[3]));
                }
                LG_TR();
                __syncthreads();
            }
            if (!BWD) {
                if (wave == 0) {
                    // the output layer reuses x tile X0 as scratch (its inputs are no longer needed in forward-only mode)
                    train_forward_layer<D3T, 1, false>(wl3, x + X3, x + X0, 0, lane);
                    const float4 yv = x[X0][lane];
                    const float y4[4] = {yv.x, yv.y, yv.z, yv.w};
    #pragma unroll
                    for (int c = 0; c < 4; c++) { const int k = 4 * g + c; if (live && k < d4) N.y[(size_t)r * d4 + k] = y4[c]; }
                }
                LG_TR();
                __syncthreads();
                LG_TR();
                continue;
            }
            // output layer: dW3 / db3 and g3 = (W3^T dy) * elu'(x3)
            train_backward_layer<D3T, 1>(wl3, gT + G4, xT + X3, gT + G3, wave, lane);
            train_weight_grad<D3T, 1>(gT + G4, xT + X3, a3, wave, lane);
            LG_TR();
            __syncthreads();
            train_backward_layer<D2T, D3T>(wl2, gT + G3, xT + X2, gT + G2, wave, lane);
            train_weight_grad<D2T, D3T>(gT + G3, xT + X2, a2, wave, lane);
            LG_TR();
            __syncthreads();
            train_backward_layer<D1T, D2T>(wl1, gT + G2, xT + X1, gT + G1, wave, lane);
            train_weight_grad<D1T, D2T>(gT + G2, xT + X1, a1, wave, lane);
            LG_TR();
            __syncthreads();
            train_weight_grad<D0T, D1T>(gT + G1, xT + X0, a0, wave, lane);
            LG_TR();
            __syncthreads();
            LG_TR();
        }
    };
    switch (wave) {
        case 0: row_tiles(std::integral_constant<int, 0>{}); break;
        case 1: row_tiles(std::integral_constant<int, 1>{}); break;
        case 2: row_tiles(std::integral_constant<int, 2>{}); break;
        default: row_tiles(std::integral_constant<int, 3>{}); break;
    }
#ifdef LG_PROFILE
    if (tr) { tr[61] = wgclk[0]; tr[62] = wall_clock64(); }      // wall clock at the start and at the end of the row-tile loop
#endif
    // The groups of a workgroup fold their accumulators through LDS (the activation areas are free now) in group order, so each
    // workgroup writes ONE partial: half the workspace traffic and half the work of k_mlp_reduce with 2 groups.
    if (BWD) {
        constexpr int NACC = P0::per_wave + P1::per_wave + P2::per_wave + P3::per_wave + 2;       // + loss terms (2 quads)
        static_assert(SLOTS == 1 || NACC * 64 * LG_TRAIN_WAVES * 4 <= S::floats, "accumulator exchange must fit the LDS");
        float4 *xch = reinterpret_cast<float4 *>(lds) + (size_t)(wave * 64 + lane) * NACC;       // indexed by ROLE: same pairs in every group
        if (LOSS && wave == LG_TRAIN_WAVES - 1) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
                for (int c = 0; c < 4; c++) gstd[c] += __shfl_xor(gstd[c], o);
                sum_sur += __shfl_xor(sum_sur, o); sum_kl += __shfl_xor(sum_kl, o); sum_val += __shfl_xor(sum_val, o);
            }
        }
        for (int s_ = SLOTS - 1; s_ >= 1; s_--) {                  // group s_ hands its sums to group s_ - 1
            __syncthreads();
            if (slot == s_) {
                int n = 0;
#pragma unroll
                for (int i = 0; i < P0::per_wave; i++) xch[n++] = make_float4(a0[i][0], a0[i][1], a0[i][2], a0[i][3]);
#pragma unroll
                for (int i = 0; i < P1::per_wave; i++) xch[n++] = make_float4(a1[i][0], a1[i][1], a1[i][2], a1[i][3]);
#pragma unroll
                for (int i = 0; i < P2::per_wave; i++) xch[n++] = make_float4(a2[i][0], a2[i][1], a2[i][2], a2[i][3]);
#pragma unroll
                for (int i = 0; i < P3::per_wave; i++) xch[n++] = make_float4(a3[i][0], a3[i][1], a3[i][2], a3[i][3]);
                xch[n++] = make_float4(gstd[0], gstd[1], gstd[2], gstd[3]);
                xch[n++] = make_float4(sum_sur, sum_val, sum_kl, 0.0f);
            }
            __syncthreads();
            if (slot == s_ - 1) {
                int n = 0;
                auto add = [&](f32x4 &a) { const float4 v = xch[n++]; a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w; };
#pragma unroll
                for (int i = 0; i < P0::per_wave; i++) add(a0[i]);
#pragma unroll
                for (int i = 0; i < P1::per_wave; i++) add(a1[i]);
#pragma unroll
                for (int i = 0; i < P2::per_wave; i++) add(a2[i]);
#pragma unroll
                for (int i = 0; i < P3::per_wave; i++) add(a3[i]);
                const float4 gs = xch[n++], sm = xch[n++];
                gstd[0] += gs.x; gstd[1] += gs.y; gstd[2] += gs.z; gstd[3] += gs.w;
                sum_sur += sm.x; sum_val += sm.y; sum_kl += sm.z;
            }
        }
    }
    if (LOSS && slot == 0 && wave == LG_TRAIN_WAVES - 1) {         // loss partials: d_std[16] | surrogate, value loss, KL sums
        float *ex = N.partial + (size_t)blockIdx.x * N.part_stride + N.grad_floats;
        if ((lane & 15) == 0) {
#pragma unroll
            for (int c = 0; c < 4; c++) ex[4 * g + c] = gstd[c];
            if (g == 0) { ex[16] = sum_sur * A.ppo.inv_n; ex[17] = sum_val * A.ppo.inv_n; ex[18] = sum_kl * A.ppo.inv_n; ex[19] = 0.0f; }
        }
    }
    if (BWD && slot == 0) {
        float *part = N.partial + (size_t)blockIdx.x * N.part_stride;
        train_flush<D0T, D1T>(a0, part, d0, d1, wave, lane); part += (size_t)d1 * d0 + d1;
        train_flush<D1T, D2T>(a1, part, d1, d2, wave, lane); part += (size_t)d2 * d1 + d2;
        train_flush<D2T, D3T>(a2, part, d2, d3, wave, lane); part += (size_t)d3 * d2 + d3;
        train_flush<D3T, 1>(a3, part, d3, d4, wave, lane);
    }
    LG_TR();
#ifdef LG_PROFILE
    if (wgclk) wgclk[1] = wall_clock64();
    if (tr) tr[63] = wgclk[1];
#endif
#undef LG_TR
}

struct MlpReduceArgs {
    const float *partial[2];
    float *gw[2][4], *gb[2][4];
    int32_t dims[2][5];
    int32_t grad_floats[2], part_stride[2];
    int32_t n_partials;
    // fused-loss launches: the LG_PPO_EXTRA partials behind the gradients become d_std [A] (+ entropy term) and stats[4]
    int32_t loss, num_actions;
    const float *std;
    float ecoef;
    float *d_std, *stats;
    float *loss_acc;           // optional [2]: += {value-loss mean, surrogate mean} (the update's running sums for the log)
};
// grads = sum over workgroups of the partials, in a fixed order.  A workgroup owns 32 consecutive gradient elements; its 8
// slices (threadIdx >> 5) sum partials k = slice, slice + 8, ... (128-byte coalesced rows) and meet in LDS.
__global__ void __launch_bounds__(256) k_mlp_reduce(const MlpReduceArgs A) {
    __shared__ float red[8][32];
    const int n = blockIdx.y, e = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + e;
    const int gf = A.grad_floats[n], ps = A.part_stride[n];
    const bool ok = j < gf + (A.loss ? LG_PPO_EXTRA : 0);
    const float *p = A.partial[n] + (ok ? j : 0);
    float s0 = 0.f, s1 = 0.f;
    int k = slice;
    for (; k + 8 < A.n_partials; k += 16) { s0 += p[(size_t)k * ps]; s1 += p[(size_t)(k + 8) * ps]; }
    if (k < A.n_partials) s0 += p[(size_t)k * ps];
    red[slice][e] = s0 + s1;
    __syncthreads();
    if (slice != 0 || !ok) return;
    const float s = ((red[0][e] + red[1][e]) + (red[2][e] + red[3][e])) + ((red[4][e] + red[5][e]) + (red[6][e] + red[7][e]));
    int off = j;
#pragma unroll
    for (int l = 0; l < 4; l++) {
        const int nw = A.dims[n][l + 1] * A.dims[n][l], nb = A.dims[n][l + 1];
        if (off < nw) { A.gw[n][l][off] = s; return; }
        off -= nw;
        if (off < nb) { A.gb[n][l][off] = s; return; }
        off -= nb;
    }
    // off = index into the loss partials (fused-loss launches only)
    if (n == 0) {
        if (off < 16) { if (off < A.num_actions) A.d_std[off] = s - A.ecoef / A.std[off]; }       // + d(-ecoef * entropy) / d sigma
        else if (off == 16) { A.stats[0] = s; if (A.loss_acc) A.loss_acc[1] += s; }
        else if (off == 18) A.stats[2] = s;
        else if (off == 19) {                          // entropy is row-independent: sum_a (0.5 + 0.5 log 2 pi + log sigma_a)
            float H = 0.0f;
            for (int a = 0; a < A.num_actions; a++) H += 1.418938533f + __logf(A.std[a]);
            A.stats[3] = H;
        }
    } else if (off == 17) { A.stats[1] = s; if (A.loss_acc) A.loss_acc[0] += s; }
}

// ---- gradient-norm clip + Adam + adaptive-KL learning rate: the rest of a PPO mini-batch step in three launches ---------------
#define LG_ADAM_MAX_TENSORS 32
struct AdamTensor { float *param; const float *grad; float *exp_avg, *exp_avg_sq, *step; int64_t numel; };
struct AdamArgs {
    AdamTensor t[LG_ADAM_MAX_TENSORS];
    int32_t n_tensors;
    float *lr;                 // device scalar (read by the update, written by the KL rule)
    const float *kl;           // device scalar or null (fixed schedule)
    float *scratch;            // [LG_ADAM_SCRATCH_FLOATS]: total gradient norm, clip coefficient, then the partial sums of squares
    float beta1, beta2, eps, max_norm, desired_kl;
};

// ||g||^2 in two fixed-order stages: every tensor is cut into LG_ADAM_CHUNKS chunks, workgroup (chunk, tensor) writes its partial
// sum to scratch[2 + tensor * LG_ADAM_CHUNKS + chunk] (the 512-wide networks have 580 k parameters: one workgroup took 200 us)
#define LG_ADAM_CHUNKS 64
__global__ void __launch_bounds__(256) k_adam_sumsq(const AdamArgs A, int64_t chunk_len) {
    __shared__ float red[256];
    const AdamTensor &T = A.t[blockIdx.y];
    const int64_t lo = (int64_t)blockIdx.x * chunk_len, hi = min(lo + chunk_len, T.numel);
    float s = 0.0f;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) s = fmaf(T.grad[i], T.grad[i], s);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) A.scratch[2 + blockIdx.y * LG_ADAM_CHUNKS + blockIdx.x] = red[0];
}

// one workgroup: sum of the partials -> clip coefficient; step counters += 1; the KL rule on lr
// (rsl_rl PPO.update [EXTERNAL]: kl > 2 d -> lr = max(1e-5, lr / 1.5); 0 < kl < d / 2 -> lr = min(1e-2, lr * 1.5))
__global__ void __launch_bounds__(1024) k_adam_prepare(const AdamArgs A) {
    __shared__ float red[1024];
    float s = 0.0f;
    for (int i = threadIdx.x; i < A.n_tensors * LG_ADAM_CHUNKS; i += 1024) s += A.scratch[2 + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if ((int)threadIdx.x < A.n_tensors) A.t[threadIdx.x].step[0] += 1.0f;
    if (threadIdx.x == 0) {
        const float norm = sqrtf(red[0]);
        A.scratch[0] = norm;
        A.scratch[1] = fminf(1.0f, A.max_norm / (norm + 1e-6f));      // torch.nn.utils.clip_grad_norm_
        if (A.kl && A.desired_kl > 0.0f) {
            const float kl = A.kl[0], lr = A.lr[0];
            if (kl > 2.0f * A.desired_kl) A.lr[0] = fmaxf(1e-5f, lr / 1.5f);
            else if (kl < 0.5f * A.desired_kl && kl > 0.0f) A.lr[0] = fminf(1e-2f, lr * 1.5f);
        }
    }
}

// torch.optim.Adam (no weight decay, no amsgrad) on the clipped gradient; tensor blockIdx.y, 256 elements per workgroup
__global__ void __launch_bounds__(256) k_adam_update(const AdamArgs A) {
    const AdamTensor &T = A.t[blockIdx.y];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= T.numel) return;
    const float step = T.step[0], lr = A.lr[0], coef = A.scratch[1];
    const float bc1 = 1.0f - powf(A.beta1, step), bc2 = 1.0f - powf(A.beta2, step);
    const float g = T.grad[i] * coef;
    const float m = T.exp_avg[i] + (g - T.exp_avg[i]) * (1.0f - A.beta1);
    const float v = T.exp_avg_sq[i] * A.beta2 + (1.0f - A.beta2) * g * g;
    T.exp_avg[i] = m; T.exp_avg_sq[i] = v;
    T.param[i] -= (lr / bc1) * m / (sqrtf(v) / sqrtf(bc2) + A.eps);
}

// ---- rollout bookkeeping: one launch per env step instead of ~18 copies / tiny reductions ---------------------------------
struct RecordArgs {
    const float *obs, *actions, *mean, *rewards; const uint8_t *dones, *time_outs;           // this step: [N, .] / [N]
    float *st_obs, *st_actions, *st_mu, *st_rewards; uint8_t *st_dones; float *st_time_outs;  // storage slices of step t
    float *cur_rew, *cur_len, *sums;                                                         // running episode return / length [N]; {sum_rew, sum_len, count}
    const float *std; float *st_sigma, *st_log_prob;                                         // optional: policy std [A] -> sigma[t] [N, A], log pi(a) [N]
    int32_t num_envs, num_obs, num_actions;
};
__global__ void __launch_bounds__(256) k_rollout_record(const RecordArgs A) {
    // blocks [0, nb_copy): one element of the observation copy per thread (+ the action / mean copies on the first num_actions lanes of an env);
    // blocks [nb_copy, ...): the per-env bookkeeping on 16 lanes per env -- lane a owns action a's log-prob term, a 16-lane butterfly sums
    // them (one thread per env walking the actions serially was a ~100-instruction chain of dependent loads: the kernel's 10 us)
    const int64_t n_copy = (int64_t)A.num_envs * A.num_obs;
    const int nb_copy = (int)((n_copy + 255) / 256);
    if ((int)blockIdx.x < nb_copy) {
        const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (i >= n_copy) return;
        const int env = (int)(i / A.num_obs), k = (int)(i % A.num_obs);
        A.st_obs[i] = A.obs[i];
        if (k < A.num_actions) {
            const size_t j = (size_t)env * A.num_actions + k;
            A.st_actions[j] = A.actions[j]; A.st_mu[j] = A.mean[j];
        }
        return;
    }
    const int t = ((int)blockIdx.x - nb_copy) * 256 + threadIdx.x, env_raw = t >> 4, a = t & 15;
    const bool live = env_raw < A.num_envs;
    const int env = live ? env_raw : A.num_envs - 1;
    float term = 0.0f;
    if (A.std && a < A.num_actions) {                  // Normal(mean, std).log_prob(action).sum(-1) and the broadcast std, as PPO.act stores them
        const size_t j = (size_t)env * A.num_actions + a;
        const float sg = A.std[a], z = (A.actions[j] - A.mean[j]) / sg;
        term = -0.5f * z * z - __logf(sg) - 0.918938533f;
        if (live) A.st_sigma[j] = sg;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) term += __shfl_xor(term, o);      // within the env's 16 lanes (aligned: 256 threads = 16 envs per block)
    if (!live || a != 0) return;
    if (A.std) A.st_log_prob[env] = term;
    const float r = A.rewards[env];
    const uint8_t d = A.dones[env];
    A.st_rewards[env] = r; A.st_dones[env] = d;
    if (A.st_time_outs) A.st_time_outs[env] = A.time_outs && A.time_outs[env] ? 1.0f : 0.0f;
    if (A.cur_rew) {                                   // episode statistics for the log (rsl_rl OnPolicyRunner.learn's rewbuffer / lenbuffer)
        const float cr = A.cur_rew[env] + r, cl = A.cur_len[env] + 1.0f;
        if (d) { atomicAdd(A.sums + 0, cr); atomicAdd(A.sums + 1, cl); atomicAdd(A.sums + 2, 1.0f); }
        A.cur_rew[env] = d ? 0.0f : cr; A.cur_len[env] = d ? 0.0f : cl;
    }
}

// ---- the same bookkeeping for a whole rollout segment that lg_rollout_policy wrote straight into the storage: ONE launch ------------
struct RollPostArgs {
    const float *actions, *mean, *rewards; const uint8_t *dones, *time_outs;   // [T][N][A] / [T][N]
    const float *std; float *sigma, *log_prob, *time_outs_f;                   // [A] -> [T][N][A], [T][N], [T][N] (0/1, optional)
    float *cur_rew, *cur_len, *sums;                                            // running episode return / length [N]; {sum_rew, sum_len, count}
    int32_t steps, num_envs, num_actions;
};
__global__ void __launch_bounds__(256) k_rollout_post(const RollPostArgs A) {
    // blocks [0, nb_tr): 16 lanes per transition (t, env): lane a owns action a's log-prob term (as k_rollout_record);
    // blocks [nb_tr, ...): one thread per env walks its T transitions in order for the episode statistics (rsl_rl's rewbuffer / lenbuffer)
    const int64_t n_tr = (int64_t)A.steps * A.num_envs;
    const int nb_tr = (int)((n_tr * 16 + 255) / 256);
    if ((int)blockIdx.x < nb_tr) {
        const int64_t t16 = (int64_t)blockIdx.x * 256 + threadIdx.x;
        const int64_t tr_raw = t16 >> 4; const int a = (int)(t16 & 15);
        const bool live = tr_raw < n_tr;
        const int64_t tr = live ? tr_raw : n_tr - 1;
        float term = 0.0f;
        if (a < A.num_actions) {
            const size_t j = (size_t)tr * A.num_actions + a;
            const float sg = A.std[a], z = (A.actions[j] - A.mean[j]) / sg;
            term = -0.5f * z * z - __logf(sg) - 0.918938533f;
            if (live) A.sigma[j] = sg;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) term += __shfl_xor(term, o);
        if (live && a == 0) {
            A.log_prob[tr] = term;
            if (A.time_outs_f) A.time_outs_f[tr] = A.time_outs && A.time_outs[tr] ? 1.0f : 0.0f;
        }
        return;
    }
    const int env = ((int)blockIdx.x - nb_tr) * 256 + threadIdx.x;
    if (env >= A.num_envs || !A.cur_rew) return;
    float cr = A.cur_rew[env], cl = A.cur_len[env], s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
    for (int t = 0; t < A.steps; t++) {
        cr += A.rewards[(size_t)t * A.num_envs + env]; cl += 1.0f;
        if (A.dones[(size_t)t * A.num_envs + env]) { s0 += cr; s1 += cl; s2 += 1.0f; cr = 0.0f; cl = 0.0f; }
    }
    A.cur_rew[env] = cr; A.cur_len[env] = cl;
    if (s2 > 0.0f) { atomicAdd(A.sums + 0, s0); atomicAdd(A.sums + 1, s1); atomicAdd(A.sums + 2, s2); }
}

}  // namespace lg
