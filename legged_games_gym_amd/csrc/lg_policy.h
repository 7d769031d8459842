// lg_policy.h -- fused actor MLP + Gaussian sampling on the matrix cores (rollout-time policy forward).
//
// Stands in for rsl_rl ActorCritic.act() ([EXTERNAL]; dims from reference legged_robot_config.py:204-209 and
// anymal_c_flat_config.py:62-65): a = actor(obs) + std * eps, actor = Linear/ELU x3 + Linear.
//
// One workgroup = 16 envs on LG_POLICY_WAVES waves (one per SIMD of the CU: a lone wave issues serially, so the layer's
// output tiles are dealt round-robin to the waves and the activations are exchanged through LDS, one barrier per layer).
// v_mfma_f32_16x16x4_f32 computes  D[out 16][env 16] += W[out][k 4] * X[k][env]:
//   A operand (weights): lane l holds W[16*o + (l&15)][k_(l>>4)],   B operand (activations): lane l holds X[k_(l>>4)][env l&15],
//   D: lane l holds rows 4*(l>>4)+r (r = 0..3) of column env l&15.
// So the float4 a lane owns of output tile t (lane group g = l>>4) is features 16t + 4g + (0..3) of env l&15 -- exactly a
// B operand if the next layer's K-steps are enumerated as (t, r) with k_g = 16t + 4g + r.  The host permutes the weight
// columns accordingly (lg_policy_pack), so the LDS exchange is a plain [tile][lane] float4 array: no transposes.
// Weights stream from L2 as one coalesced 256-B load per MFMA (the whole flat actor is 67 KB).
#pragma once
#include "lg_device.h"

namespace lg {

typedef float f32x4 __attribute__((ext_vector_type(4)));

LG_DEV float elu1(float x) { return x > 0.0f ? x : (__builtin_amdgcn_exp2f(1.442695041f * x) - 1.0f); }

#define LG_POLICY_WAVES 4
// one layer: IN_T input tiles (LDS, [tile][lane] float4) -> OUT_T output tiles (LDS); wave w computes tiles w, w + NW, ...
template <int IN_T, int OUT_T, bool ACT>
LG_DEV void mlp_layer(const float *__restrict__ w /* [OUT_T][IN_T*4][64] */, const float *__restrict__ b /* [OUT_T][4][64] */,
                      const float4 (*xin)[64], float4 (*xout)[64], int wave, int lane) {
    // All IN_T*4 weight words of an output tile are requested before the first MFMA consumes one: a lone wave otherwise
    // keeps only ~16 of these L2 loads in flight and the 512-wide actors become load-latency bound (95 us per call).
    constexpr int CH = IN_T > 16 ? 16 : IN_T;                      // K tiles per register batch (64 VGPRs)
#pragma unroll 1
    for (int o = wave; o < OUT_T; o += LG_POLICY_WAVES) {
        f32x4 acc;
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] = b[(o * 4 + r) * 64 + lane];
        const float *wo = w + (size_t)o * IN_T * 4 * 64 + lane;
#pragma unroll
        for (int t0 = 0; t0 < IN_T; t0 += CH) {
            float wr[CH * 4];
#pragma unroll
            for (int i = 0; i < CH * 4; i++) wr[i] = (t0 * 4 + i < IN_T * 4) ? wo[(t0 * 4 + i) * 64] : 0.0f;
            __builtin_amdgcn_sched_barrier(0);                     // keep the scheduler from re-interleaving loads and MFMAs
#pragma unroll
            for (int t = 0; t < CH; t++) {
                if (t0 + t < IN_T) {
                    const float4 xv = xin[t0 + t][lane];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[t * 4 + 0], xv.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[t * 4 + 1], xv.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[t * 4 + 2], xv.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[t * 4 + 3], xv.w, acc, 0, 0, 0);
                }
            }
        }
        xout[o][lane] = ACT ? make_float4(elu1(acc[0]), elu1(acc[1]), elu1(acc[2]), elu1(acc[3])) : make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// One output tile's weights and biases held in registers (the flat actor's layers are small enough for a wave to hold ALL its tiles of all
// four layers: see policy_forward)
template <int IN_T> struct TileW {
    float w[IN_T * 4], b[4];
    LG_DEV void load(const float *__restrict__ wp, const float *__restrict__ bp, int o, int lane) {
#pragma unroll
        for (int i = 0; i < IN_T * 4; i++) w[i] = wp[((size_t)o * IN_T * 4 + i) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; r++) b[r] = bp[(o * 4 + r) * 64 + lane];
    }
    template <bool ACT> LG_DEV float4 run(const float4 (*xin)[64], int lane) const {
        f32x4 acc = {b[0], b[1], b[2], b[3]};
#pragma unroll
        for (int t = 0; t < IN_T; t++) {
            const float4 xv = xin[t][lane];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t * 4 + 0], xv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t * 4 + 1], xv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t * 4 + 2], xv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t * 4 + 3], xv.w, acc, 0, 0, 0);
        }
        return ACT ? make_float4(elu1(acc[0]), elu1(acc[1]), elu1(acc[2]), elu1(acc[3])) : make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
};

struct PolicyArgs {
    const float *obs;      // [N, num_obs]
    const float *w[4];     // packed weights per layer (lg_policy_pack)
    const float *b[4];     // packed biases per layer
    const float *std;      // [num_actions]
    float *actions;        // [N, num_actions]  sampled
    float *mean;           // [N, num_actions]  or null
    const int64_t *step_counter;   // device counter (Philox stream), may be null
    int64_t step;          // used when >= 0
    uint64_t seed;
    int32_t num_envs, num_obs, num_actions, deterministic;
};

// Actor forward + sampling for the 16 envs of workgroup `block`; must be called by all LG_POLICY_WAVES waves of the
// workgroup (contains barriers).  Wave 0 samples and writes actions / mean to global memory and, if `lds_act` is given,
// publishes the sampled actions as lds_act[action][env-in-block] for a consumer in the same workgroup (k_step's fused mode;
// the caller places the barrier).  Layer widths in tiles of 16 (D0T = ceil(num_obs/16)); the output layer is one tile.
template <int D0T, int D1T, int D2T, int D3T>
LG_DEV void policy_forward(const PolicyArgs &A, float4 (*xa)[64], float4 (*xb)[64], float4 (*xy)[64], int block, int wave, int lane,
                           int64_t step, float (*lds_act)[16], const float (*lds_obs)[48] = nullptr /* the 16 envs' observations in LDS instead of A.obs */) {
    const int g = lane >> 4;
    int env = block * 16 + (lane & 15);
    const bool live = env < A.num_envs;
    if (!live) env = A.num_envs - 1;
    // The 48-128-64-32 actor (67 KB of weights): a wave's share of ALL four layers is 100 registers at most, so every weight is requested
    // up front, in layer order -- mlp_layer() per layer starts each of its tiles with a full L2 round trip (five per call, ~40 % of the
    // 3.2 us the actor adds to a policy step of k_step).
    constexpr bool PRELOAD = D0T <= 4 && D1T == 2 * LG_POLICY_WAVES && D2T == LG_POLICY_WAVES && D3T == 2;
    TileW<PRELOAD ? D0T : 1> w0a, w0b;
    TileW<PRELOAD ? D1T : 1> w1;
    TileW<PRELOAD ? D2T : 1> w2;
    TileW<PRELOAD ? D3T : 1> w3;
    if constexpr (PRELOAD) {
        w0a.load(A.w[0], A.b[0], wave, lane); w0b.load(A.w[0], A.b[0], wave + LG_POLICY_WAVES, lane);
        w1.load(A.w[1], A.b[1], wave, lane);
        if (wave < D3T) w2.load(A.w[2], A.b[2], wave, lane);
        if (wave == 0) w3.load(A.w[3], A.b[3], 0, lane);
    }
    // layer-0 B operands from global: k_g = 16t + 4g + r
    const float *o = A.obs + (size_t)env * A.num_obs;
    for (int t = wave; t < D0T; t += LG_POLICY_WAVES) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { int k = 16 * t + 4 * g + r; v[r] = k < A.num_obs ? (lds_obs ? lds_obs[lane & 15][k] : o[k]) : 0.0f; }
        xa[t][lane] = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    if constexpr (PRELOAD) {
        xb[wave][lane] = w0a.template run<true>(xa, lane);
        xb[wave + LG_POLICY_WAVES][lane] = w0b.template run<true>(xa, lane);
    } else mlp_layer<D0T, D1T, true>(A.w[0], A.b[0], xa, xb, wave, lane);
    __syncthreads();
    if constexpr (PRELOAD) xa[wave][lane] = w1.template run<true>(xb, lane);
    else mlp_layer<D1T, D2T, true>(A.w[1], A.b[1], xb, xa, wave, lane);
    __syncthreads();
    if constexpr (PRELOAD) { if (wave < D3T) xb[wave][lane] = w2.template run<true>(xa, lane); }
    else mlp_layer<D2T, D3T, true>(A.w[2], A.b[2], xa, xb, wave, lane);
    // The exploration noise std * eps (Philox -> Box-Muller: two logs, two sincos, four loads of std) is drawn by the LAST wave -- which has
    // no output tile in this layer of the 48-128-64-32 actor -- into the still unused xy tile, not by wave 0 behind the output layer, where
    // everything waits for the actions.
    if (wave == LG_POLICY_WAVES - 1) {
        float u[4];
        rand4(A.seed ^ 0x9E3779B97F4A7C15ull, env, step, 100 + g, 0, u);
        float rad0 = sqrtf(-2.0f * __logf(fmaxf(u[0], 1e-12f))), rad1 = sqrtf(-2.0f * __logf(fmaxf(u[2], 1e-12f)));
        float s0, c0, s1, c1;
        __sincosf(6.2831853f * u[1], &s0, &c0);
        __sincosf(6.2831853f * u[3], &s1, &c1);
        const float eps[4] = {rad0 * c0, rad0 * s0, rad1 * c1, rad1 * s1};
        float ns[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { const int a = 4 * g + r; ns[r] = (a < A.num_actions && !A.deterministic) ? A.std[a] * eps[r] : 0.0f; }
        xy[0][lane] = make_float4(ns[0], ns[1], ns[2], ns[3]);
    }
    __syncthreads();
    if (wave != 0) return;
    const float4 nv = xy[0][lane];                                 // lane (env, g): std * eps of actions 4g .. 4g + 3 (read before the output layer reuses the tile)
    const float ns[4] = {nv.x, nv.y, nv.z, nv.w};
    if constexpr (PRELOAD) xy[0][lane] = w3.template run<false>(xb, lane);
    else mlp_layer<D3T, 1, false>(A.w[3], A.b[3], xb, xy, 0, lane);
    const float4 yv = xy[0][lane];                                 // written by this lane
    const float y[4] = {yv.x, yv.y, yv.z, yv.w};
    // lane (env, g) now holds mean[4g + r]; a = mean + std * eps
#pragma unroll
    for (int r = 0; r < 4; r++) {
        int a = 4 * g + r;
        if (a < A.num_actions) {
            float m = y[r], act = m + ns[r];
            if (lds_act) lds_act[a][lane & 15] = act;
            if (live) {
                if (A.mean) A.mean[(size_t)env * A.num_actions + a] = m;
                A.actions[(size_t)env * A.num_actions + a] = act;
            }
        }
    }
}

template <int D0T, int D1T, int D2T, int D3T>
__global__ void __launch_bounds__(64 * LG_POLICY_WAVES) k_policy_act(const PolicyArgs A) {
    constexpr int TA = D0T > D2T ? D0T : D2T, TB = D1T > D3T ? D1T : D3T;
    __shared__ float4 xa[TA][64], xb[TB][64], xy[1][64];          // ping-pong activations: obs/x2 in xa, x1/x3 in xb
    const int64_t step = A.step >= 0 ? A.step : (A.step_counter ? A.step_counter[0] + 1 : 0);
    policy_forward<D0T, D1T, D2T, D3T>(A, xa, xb, xy, blockIdx.x, threadIdx.x >> 6, threadIdx.x & 63, step, nullptr);
}

// ------------------------------------------------------------------ wide actors (235/169-512-256-128-12): split-bf16 build
// The f32 kernel above is bound by the f32 MFMA rate (4512 v_mfma_f32_16x16x4_f32 per 16 envs = 15 us of matrix-core time per
// workgroup) and streams the 1.1 MB of weights once per 16 envs.  This build works on 32 envs per workgroup with
// v_mfma_f32_32x32x16_bf16 (16 x the f32 rate): every f32 operand x is split into hi = bf16(x), lo = bf16(x - hi) and each
// product is formed as lo*hi + hi*lo + hi*hi with f32 accumulation (the dropped lo*lo term is 2^-16 relative) -- the same
// arithmetic as k_gemm_wide_bf16x3 (lg_gemm.h), selected by the same switch (lg_mlp_wide_set_precision).
//   D[out 32][env 32] += W[out][k 16] * X[k][env]
//   A (weights):      lane l holds W[32o + (l&31)][k = 8(l>>5) + 0..7]      -- pre-split and packed on the device (k_policy_pack_wide),
//                     one coalesced 1-KB global_load_dwordx4 per operand, straight into registers: each weight is used by ONE wave
//   B (activations):  lane l holds X[k = 8(l>>5) + 0..7][env l&31]          -- LDS, [k-step][hi/lo][lane] 16-byte entries
//   D:                lane l, register i holds row 8(i>>2) + 4(l>>5) + (i&3) of column env l&31
// so registers 8jj..8jj+7 of output tile o are exactly the B operand of the next layer's k-step 2o + jj when that layer's K
// dimension is enumerated as  k-step (o, jj), lane half h, element (j', r) -> feature 32o + 8(2jj + j') + 4h + r.
// The pack kernel permutes the weight columns accordingly: the LDS exchange needs no transposes.
typedef __bf16 bf16x8g __attribute__((ext_vector_type(8)));
typedef float f32x16p __attribute__((ext_vector_type(16)));
#define LG_PW_ENVS 32
#define LG_PW_WAVES 8                  // two waves per SIMD: one wave's epilogue / LDS traffic hides under the other's MFMAs
#define LG_PW_INFLIGHT 32              // 1-KB weight loads a wave keeps in flight: the stream is L2-LATENCY bound (measured: 16 loads in
                                       // flight per wave on 4 waves = 18 B/clk/CU of the ~64 the L2 delivers), so 8 x 32 KB per CU

struct PolicyWideArgs {
    PolicyArgs base;
    const bf16x8g *wb[4];  // [out tile][k-step][hi/lo][lane] (k_policy_pack_wide)
    const float *bb[4];    // biases, padded to 32 * out tiles
};

LG_DEV void split8(const float (&v)[8], bf16x8g &hi, bf16x8g &lo) {
#pragma unroll
    for (int i = 0; i < 8; i++) { hi[i] = (__bf16)v[i]; lo[i] = (__bf16)(v[i] - (float)hi[i]); }
}

// The weight stream of one wave through one layer: KS k-steps x TPW output tiles, a ring of PF k-steps of operands in registers.
// Workgroups walk the k-steps from different starting points (rot) so that the chip does not ask the L2 for the same lines at once.
template <int KS, int TPW> struct WideStream {
    static constexpr int PF = (LG_PW_INFLIGHT / (2 * TPW) + 1) < KS + 1 ? (LG_PW_INFLIGHT / (2 * TPW) + 1) : KS + 1;     // ring slots; PF - 1 k-steps ahead
    bf16x8g wh[PF][TPW], wl[PF][TPW];
    LG_DEV void fetch(const bf16x8g *__restrict__ w, int o0, int rot, int lane, int s) {
        int sr = s + rot; sr = sr >= KS ? sr - KS : sr;
#pragma unroll
        for (int t = 0; t < TPW; t++) {
            const bf16x8g *q = w + ((size_t)((o0 + t) * KS + sr) * 2) * 64 + lane;
            wh[s % PF][t] = q[0]; wl[s % PF][t] = q[64];
        }
    }
    LG_DEV void prime(const bf16x8g *__restrict__ w, int o0, int rot, int lane) {
#pragma unroll
        for (int s = 0; s < PF - 1; s++) if (s < KS) fetch(w, o0, rot, lane, s);
    }
    LG_DEV void run(const bf16x8g *__restrict__ w, const bf16x8g (*xin)[2][64], int o0, int rot, int lane, f32x16p (&acc)[TPW]) {
#pragma unroll
        for (int t = 0; t < TPW; t++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[t][i] = 0.0f;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            if (s + PF - 1 < KS) fetch(w, o0, rot, lane, s + PF - 1);
            int sr = s + rot; sr = sr >= KS ? sr - KS : sr;
            const bf16x8g bh = xin[sr][0][lane], bl = xin[sr][1][lane];
#pragma unroll
            for (int t = 0; t < TPW; t++) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[s % PF][t], bh, acc[t], 0, 0, 0);     // small terms first
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[s % PF][t], bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[s % PF][t], bh, acc[t], 0, 0, 0);
            }
        }
    }
};

// bias + ELU + split of TPW finished tiles -> the next layer's k-steps 2o, 2o + 1 in LDS
template <int TPW>
LG_DEV void wide_epilogue(const f32x16p (&acc)[TPW], const float *__restrict__ b, bf16x8g (*xout)[2][64], int o0, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int t = 0; t < TPW; t++) {
        const float *bo = b + 32 * (o0 + t) + 4 * h;
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = elu1(acc[t][8 * jj + i] + bo[8 * (2 * jj + (i >> 2)) + (i & 3)]);
            bf16x8g hi, lo;
            split8(v, hi, lo);
            xout[2 * (o0 + t) + jj][0][lane] = hi; xout[2 * (o0 + t) + jj][1][lane] = lo;
        }
    }
}

// phase stamps of wave 0 of every workgroup (tools/ubench/chain_probe.hip, actor_probe.hip; -DLG_CHAIN_PROF builds only)
#ifdef LG_CHAIN_PROF
__device__ unsigned long long g_chain_prof[2048 * 16];
#define CHAIN_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.y == 0) g_chain_prof[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CHAIN_STAMP(i) do { } while (0)
#endif
// K0S = ceil(num_obs / 16) k-steps of layer 0; H1T/H2T/H3T = hidden widths / 32.  Every wave's weight stream runs ahead of the
// layer barriers: the ring of layer l+1 is primed before layer l's epilogue, so no layer starts with an empty pipeline.
template <int K0S, int H1T, int H2T, int H3T>
__global__ void __launch_bounds__(64 * LG_PW_WAVES) k_policy_act_wide(const PolicyWideArgs W) {
    const PolicyArgs &A = W.base;
    constexpr int NW = LG_PW_WAVES;
    static_assert(H1T % NW == 0 && (H2T % NW == 0 || H2T <= NW) && H3T <= NW, "tiles per wave");
    constexpr int T1 = H1T / NW, T2 = H2T >= NW ? H2T / NW : 1, T3 = 1;
    constexpr int KA = K0S > 2 * H2T ? K0S : 2 * H2T, KB = H1T > H3T ? 2 * H1T : 2 * H3T;
    __shared__ bf16x8g xa[KA][2][64], xb[KB][2][64];               // ping-pong activations: obs / x2 in xa, x1 / x3 in xb (96 KB)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5;
    const int wv = (wave + blockIdx.x) % NW;                       // which share of the output tiles this wave takes (rotates over workgroups)
    const int r0 = (blockIdx.x * 5) % K0S, r1 = (blockIdx.x * 5) % (2 * H1T), r2 = (blockIdx.x * 5) % (2 * H2T);
    const bool on2 = wv * T2 < H2T, on3 = wv * T3 < H3T;
    CHAIN_STAMP(0);
    const int64_t step = A.step >= 0 ? A.step : (A.step_counter ? A.step_counter[0] + 1 : 0);
    int env = blockIdx.x * LG_PW_ENVS + (lane & 31);
    const bool live = env < A.num_envs;
    if (!live) env = A.num_envs - 1;
    const float *o = A.obs + (size_t)env * A.num_obs;
    WideStream<K0S, T1> s1;
    s1.prime(W.wb[0], wv * T1, r0, lane);
    for (int s = wave; s < K0S; s += NW) {                         // layer-0 B operands from global, natural k order (requesting them ahead of
        float v[8];                                                // the ring, as the learner's chain does, changes nothing here: measured)
#pragma unroll
        for (int i = 0; i < 8; i++) { const int k = 16 * s + 8 * h + i; v[i] = k < A.num_obs ? o[k] : 0.0f; }
        bf16x8g hi, lo;
        split8(v, hi, lo);
        xa[s][0][lane] = hi; xa[s][1][lane] = lo;
    }
    // The exploration noise and the output biases of wave 0's lanes are formed HERE, while the first weights are still on their way, not
    // behind the last layer where one wave would walk Philox, two logs, two sincos and four dependent loads alone (5 k of the kernel's 52 k cycles).
    // lane (env, h), register i = 4ii + r (ii = 0, 1): action a = 8ii + 4h + r, i.e. group g = a >> 2 = 2ii + h of the f32 kernel's
    // noise stream (rand4 sub-stream 100 + g): same samples from both builds for the same mean
    float ns[2][4], by[2][4];                                        // std * eps (0 when deterministic), output bias
    if (wave == 0) {
#pragma unroll
        for (int ii = 0; ii < 2; ii++) {
            const int g = 2 * ii + h;
#pragma unroll
            for (int r = 0; r < 4; r++) { ns[ii][r] = 0.0f; by[ii][r] = 0.0f; }
            if (4 * g >= A.num_actions) continue;
            float u[4];
            rand4(A.seed ^ 0x9E3779B97F4A7C15ull, env, step, 100 + g, 0, u);
            const float rad0 = sqrtf(-2.0f * __logf(fmaxf(u[0], 1e-12f))), rad1 = sqrtf(-2.0f * __logf(fmaxf(u[2], 1e-12f)));
            float s0, c0, sn1, c1;
            __sincosf(6.2831853f * u[1], &s0, &c0);
            __sincosf(6.2831853f * u[3], &sn1, &c1);
            const float eps[4] = {rad0 * c0, rad0 * s0, rad1 * c1, rad1 * sn1};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int a = 4 * g + r;
                if (a < A.num_actions) { by[ii][r] = W.bb[3][a]; ns[ii][r] = A.deterministic ? 0.0f : A.std[a] * eps[r]; }
            }
        }
    }
    __syncthreads();
    CHAIN_STAMP(1);
    f32x16p a1[T1];
    s1.run(W.wb[0], xa, wv * T1, r0, lane, a1);
    CHAIN_STAMP(2);
    WideStream<2 * H1T, T2> s2;
    if (on2) s2.prime(W.wb[1], wv * T2, r1, lane);
    __builtin_amdgcn_sched_barrier(0);
    wide_epilogue<T1>(a1, W.bb[0], xb, wv * T1, lane);
    CHAIN_STAMP(3);
    __syncthreads();
    CHAIN_STAMP(4);
    f32x16p a2[T2];
    WideStream<2 * H2T, T3> s3;
    if (on2) s2.run(W.wb[1], xb, wv * T2, r1, lane, a2);
    CHAIN_STAMP(5);
    if (on3) s3.prime(W.wb[2], wv * T3, r2, lane);
    __builtin_amdgcn_sched_barrier(0);
    if (on2) wide_epilogue<T2>(a2, W.bb[1], xa, wv * T2, lane);
    __syncthreads();
    CHAIN_STAMP(6);
    f32x16p a3[T3];
    WideStream<2 * H3T, 1> s4;
    if (on3) s3.run(W.wb[2], xa, wv * T3, r2, lane, a3);
    if (wave == 0) s4.prime(W.wb[3], 0, 0, lane);
    __builtin_amdgcn_sched_barrier(0);
    if (on3) wide_epilogue<T3>(a3, W.bb[2], xb, wv * T3, lane);
    __syncthreads();
    CHAIN_STAMP(7);
    if (wave != 0) return;
    f32x16p y[1];
    s4.run(W.wb[3], xb, 0, 0, lane, y);
    CHAIN_STAMP(8);
#pragma unroll
    for (int ii = 0; ii < 2; ii++) {
        const int g = 2 * ii + h;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int a = 4 * g + r;
            if (a < A.num_actions && live) {
                const float m = y[0][4 * ii + r] + by[ii][r];
                if (A.mean) A.mean[(size_t)env * A.num_actions + a] = m;
                A.actions[(size_t)env * A.num_actions + a] = m + ns[ii][r];
            }
        }
    }
    CHAIN_STAMP(9);
}

// ------------------------------------------------------------------ the same chain as the LEARNER's forward pass
// lg_mlp_wide_forward at precision 1 for the 235/169-512-256-128 shapes: the mini-batch rows walk all four layers with the
// activations in LDS; what the backward pass needs (the post-ELU activations of the three hidden layers, f32, row-major) is
// written out on the way -- 128 contiguous bytes per row and tile -- and nothing is read back.  The layer-by-layer GEMMs this
// replaces moved every activation through HBM twice (forward: 226 us per mini-batch of 24 576 rows, both nets; a first chain
// kernel with 32 rows per workgroup, the actor kernel's structure: 163 us; this one: 130 us).
struct ChainNet {
    const float *x; int ldx, num_in;       // gathered, padded input rows (k_wide_prep's copy)
    const bf16x8g *wb[4]; const float *bb[4];
    float *act[3]; int lda[3];             // hidden activations, [mb][width]
    float *out; int out_dim;               // [mb][out_dim]
};
struct ChainArgs { ChainNet net[2]; int mb; };

// 64 rows per workgroup: every weight fragment feeds TWO 32-row halves, so the L2 -> CU weight stream that bounds the chain
// (1.16 MB per workgroup whatever the rows) is paid once per 64 rows.  A 64-row x1 (512 wide, hi + lo) alone would be 128 KB of LDS, so
// layer 0 runs in two phases of 8 output tiles and layer 1 accumulates over the 16 k-steps each phase leaves in LDS: two 64 KB
// buffers hold x0 | x2 and half of x1 | x3.
#define LG_CHAIN_INFLIGHT 32           // 1-KB weight loads a wave of the chain keeps in flight: 17 ring slots = a whole 16-k-step segment requested by prime().
                                       // (With the tile's biases held between the MFMAs and the epilogue the kernel spills 8 registers; 15 slots: no spill, 130 vs 122 us.)
template <int KS, int KTOT> struct WideStream2 {                  // one output tile, k-steps k0 .. k0+KS-1 of a layer with KTOT k-steps, two row halves
    static constexpr int PF = (LG_CHAIN_INFLIGHT / 2 + 1) < KS + 1 ? (LG_CHAIN_INFLIGHT / 2 + 1) : KS + 1;     // one ring live at a time (two half-depth rings, the next segment
                                                                                                         // requested early, measured slower: 145 vs 130 us)
    bf16x8g wh[PF], wl[PF];
    LG_DEV void fetch(const bf16x8g *__restrict__ w, int tile, int k0, int rot, int lane, int s) {
        int sr = s + rot; sr = sr >= KS ? sr - KS : sr;
        const bf16x8g *q = w + ((size_t)(tile * KTOT + k0 + sr) * 2) * 64 + lane;
        wh[s % PF] = q[0]; wl[s % PF] = q[64];
    }
    LG_DEV void prime(const bf16x8g *__restrict__ w, int tile, int k0, int rot, int lane) {
#pragma unroll
        for (int s = 0; s < PF - 1; s++) if (s < KS) fetch(w, tile, k0, rot, lane, s);
    }
    LG_DEV void run(const bf16x8g *__restrict__ w, const bf16x8g (*xin)[2][2][64], int tile, int k0, int rot, int lane, f32x16p (&acc)[2], bool zero) {
        if (zero) {
#pragma unroll
            for (int i = 0; i < 16; i++) { acc[0][i] = 0.0f; acc[1][i] = 0.0f; }
        }
#pragma unroll
        for (int s = 0; s < KS; s++) {
            if (s + PF - 1 < KS) fetch(w, tile, k0, rot, lane, s + PF - 1);
            int sr = s + rot; sr = sr >= KS ? sr - KS : sr;
            // (the two row halves interleaved or one after the other: the same time)
            const bf16x8g bh0 = xin[sr][0][0][lane], bl0 = xin[sr][0][1][lane], bh1 = xin[sr][1][0][lane], bl1 = xin[sr][1][1][lane];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[s % PF], bh0, acc[0], 0, 0, 0);     // small terms first
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[s % PF], bh1, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[s % PF], bl0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[s % PF], bl1, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[s % PF], bh0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[s % PF], bh1, acc[1], 0, 0, 0);
        }
    }
};
// bias + ELU of one finished tile (both row halves): f32 activations to global (column block `tile_g`), split -> LDS k-steps 2 tile_l, 2 tile_l + 1
// (the lane's 16 biases of the tile are fetched by chain_bias() BEFORE the next layer's weight ring is primed: loads return in order,
// and behind the ring's 32 KB per wave the epilogue started with ~1 k cycles of waiting)
struct ChainBias { float4 q[4]; };
LG_DEV ChainBias chain_bias(const float *__restrict__ b, int tile_g, int lane) {
    const float4 *bo = reinterpret_cast<const float4 *>(b + 32 * tile_g + 4 * (lane >> 5));
    ChainBias r;
#pragma unroll
    for (int c = 0; c < 4; c++) r.q[c] = bo[2 * c];          // floats 8 c .. 8 c + 3 of this lane's half
    return r;
}
LG_DEV void chain_epilogue2(const f32x16p (&acc)[2], const ChainBias &B, bf16x8g (*xout)[2][2][64], int tile_g, int tile_l, int lane,
                            float *__restrict__ act0, float *__restrict__ act1 /* the lane's rows of the two halves, or null */) {
    const int h = lane >> 5;
    const float bo[16] = {B.q[0].x, B.q[0].y, B.q[0].z, B.q[0].w, B.q[1].x, B.q[1].y, B.q[1].z, B.q[1].w,
                          B.q[2].x, B.q[2].y, B.q[2].z, B.q[2].w, B.q[3].x, B.q[3].y, B.q[3].z, B.q[3].w};
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
        float *act_row = hf ? act1 : act0;
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = elu1(acc[hf][8 * jj + i] + bo[4 * (2 * jj + (i >> 2)) + (i & 3)]);
            if (act_row) {
                float *dst = act_row + 32 * tile_g + 16 * jj + 4 * h;
                *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4 *>(dst + 8) = make_float4(v[4], v[5], v[6], v[7]);
            }
            bf16x8g hi, lo;
            split8(v, hi, lo);
            xout[2 * tile_l + jj][hf][0][lane] = hi; xout[2 * tile_l + jj][hf][1][lane] = lo;
        }
    }
}

template <int K0S>                                               // hidden widths 512-256-128 (16 / 8 / 4 tiles), 8 waves
__global__ void __launch_bounds__(64 * LG_PW_WAVES) k_mlp_chain_fwd64(const ChainArgs C) {
    static_assert(LG_PW_WAVES == 8 && K0S <= 16, "one layer-1 tile per wave; x0 fits the 16 k-step buffer");
    const ChainNet &N = C.net[blockIdx.y];
    __shared__ bf16x8g bufA[16][2][2][64], bufB[16][2][2][64];     // [k-step][row half][hi/lo][lane]: x0 | x2 and half of x1 | x3 (64 KB each)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5;
    const int wv = (wave + blockIdx.x) & 7;
    const int r0 = (blockIdx.x * 5) % K0S, r1 = (blockIdx.x * 5) & 15, r2 = (blockIdx.x * 3) & 15;
    int row[2]; bool live[2];
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
        row[hf] = blockIdx.x * 64 + 32 * hf + (lane & 31);
        live[hf] = row[hf] < C.mb;
        if (!live[hf]) row[hf] = C.mb - 1;
    }
    auto arow = [&](int l, int hf) -> float * { return live[hf] ? N.act[l] + (size_t)row[hf] * N.lda[l] : nullptr; };     // (formed at the epilogue: six
                                                                                                                           // pointers held across the MFMA phases cost 12 registers)
    CHAIN_STAMP(0);
    // x0 of both halves -> bufA.  All of a wave's input loads (16-byte vectors: the rows are zero-padded to a multiple of 4 floats by
    // k_wide_prep) are requested BEFORE the weight ring is primed -- loads return in order, and behind 32 KB of weights per wave the
    // inputs arrived last: the first barrier was 16 k cycles into the 82 k of a workgroup (now 11 k; one coalesced request per row
    // instead of these 32-byte gathers: 9.5 k, but 9 spilled registers and no faster overall).
    constexpr int NX = (2 * K0S + LG_PW_WAVES - 1) / LG_PW_WAVES;
    float4 xv[NX][2];
#pragma unroll
    for (int j = 0; j < NX; j++) {
        const int s = min(wave + LG_PW_WAVES * j, 2 * K0S - 1), ks = s >> 1, hf = s & 1, k0 = 16 * ks + 8 * h;
        const float *o = N.x + (size_t)row[hf] * N.ldx + k0;
        xv[j][0] = *reinterpret_cast<const float4 *>(k0 + 4 <= N.ldx ? o : N.x);          // clamped, zeroed below: no predicated load
        xv[j][1] = *reinterpret_cast<const float4 *>(k0 + 8 <= N.ldx ? o + 4 : N.x);
    }
    WideStream2<K0S, K0S> s0;
    s0.prime(N.wb[0], wv, 0, r0, lane);
#pragma unroll
    for (int j = 0; j < NX; j++) {
        const int s = wave + LG_PW_WAVES * j, ks = s >> 1, hf = s & 1, k0 = 16 * ks + 8 * h;
        if (s >= 2 * K0S) break;
        const bool in0 = k0 + 4 <= N.ldx, in1 = k0 + 8 <= N.ldx;
        const float v[8] = {in0 ? xv[j][0].x : 0.f, in0 ? xv[j][0].y : 0.f, in0 ? xv[j][0].z : 0.f, in0 ? xv[j][0].w : 0.f,
                            in1 ? xv[j][1].x : 0.f, in1 ? xv[j][1].y : 0.f, in1 ? xv[j][1].z : 0.f, in1 ? xv[j][1].w : 0.f};
        bf16x8g hi, lo;
        split8(v, hi, lo);
        bufA[ks][hf][0][lane] = hi; bufA[ks][hf][1][lane] = lo;
    }
    __syncthreads();
    CHAIN_STAMP(1);
    f32x16p a0[2], a1[2];
    WideStream2<16, 32> s1;
#pragma unroll                                                     // unrolled: only one weight ring is live at a time (a loop would carry both: 270 spills)
    for (int ph = 0; ph < 2; ph++) {
        s0.run(N.wb[0], bufA, 8 * ph + wv, 0, r0, lane, a0, true);                      // layer-0 tile 8 ph + wv
        CHAIN_STAMP(2 + 4 * ph);
        const ChainBias b0 = chain_bias(N.bb[0], 8 * ph + wv, lane);
        __builtin_amdgcn_sched_barrier(0);
        s1.prime(N.wb[1], wv, 16 * ph, r1, lane);                                       // layer-1 tile wv, this phase's 16 k-steps
        __builtin_amdgcn_sched_barrier(0);
        chain_epilogue2(a0, b0, bufB, 8 * ph + wv, wv, lane, arow(0, 0), arow(0, 1));
        CHAIN_STAMP(3 + 4 * ph);
        __syncthreads();
        CHAIN_STAMP(4 + 4 * ph);
        s1.run(N.wb[1], bufB, wv, 16 * ph, r1, lane, a1, ph == 0);
        if (ph == 0) s0.prime(N.wb[0], 8 + wv, 0, r0, lane);
        __syncthreads();                                                                // bufB free again
        CHAIN_STAMP(5 + 4 * ph);
    }
    WideStream2<16, 16> s2;
    const ChainBias b1 = chain_bias(N.bb[1], wv, lane);
    __builtin_amdgcn_sched_barrier(0);
    if (wv < 4) s2.prime(N.wb[2], wv, 0, r2, lane);
    __builtin_amdgcn_sched_barrier(0);
    chain_epilogue2(a1, b1, bufA, wv, wv, lane, arow(1, 0), arow(1, 1));           // x2 over x0 (every wave is past layer 0)
    __syncthreads();
    CHAIN_STAMP(10);
    f32x16p a2[2];
    WideStream2<8, 8> s3;
    if (wv < 4) s2.run(N.wb[2], bufA, wv, 0, r2, lane, a2, true);
    const ChainBias b2 = chain_bias(N.bb[2], wv & 3, lane);
    float by[4];                                                   // output biases of this lane's action groups (wave 0)
#pragma unroll
    for (int r = 0; r < 4; r++) by[r] = N.bb[3][4 * h + r];
    float by2[4];
#pragma unroll
    for (int r = 0; r < 4; r++) by2[r] = N.bb[3][8 + 4 * h + r];
    __builtin_amdgcn_sched_barrier(0);
    if (wave == 0) s3.prime(N.wb[3], 0, 0, 0, lane);
    __builtin_amdgcn_sched_barrier(0);
    if (wv < 4) chain_epilogue2(a2, b2, bufB, wv, wv, lane, arow(2, 0), arow(2, 1));
    __syncthreads();
    CHAIN_STAMP(11);
    if (wave != 0) return;
    f32x16p y[2];
    s3.run(N.wb[3], bufB, 0, 0, 0, lane, y, true);
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
        if (!live[hf]) continue;
#pragma unroll
        for (int ii = 0; ii < 2; ii++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int a = 8 * ii + 4 * h + r;
                if (a < N.out_dim) N.out[(size_t)row[hf] * N.out_dim + a] = y[hf][4 * ii + r] + (ii ? by2[r] : by[r]);
            }
    }
    CHAIN_STAMP(12);
}

// torch Linear [out, in] f32 -> the operand stream above; `first` selects the natural k order of layer 0
LG_DEV void pack_wide_body(const float *__restrict__ Wt, const float *__restrict__ bias, int in_dim, int out_dim, int KS, int OT, int first,
                           __bf16 *__restrict__ wp, float *__restrict__ bp) {
    const size_t n = (size_t)OT * KS * 64 * 8;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < n + (size_t)OT * 32; idx += (size_t)gridDim.x * blockDim.x) {
        if (idx >= n) { const int r = (int)(idx - n); bp[r] = r < out_dim ? bias[r] : 0.0f; continue; }
        const int i = (int)(idx & 7), l = (int)((idx >> 3) & 63);
        const size_t os = idx >> 9;
        const int s = (int)(os % KS), o = (int)(os / KS), hh = l >> 5;
        const int row = 32 * o + (l & 31);
        const int col = first ? 16 * s + 8 * hh + i : 32 * (s >> 1) + 8 * (2 * (s & 1) + (i >> 2)) + 4 * hh + (i & 3);
        const float v = (row < out_dim && col < in_dim) ? Wt[(size_t)row * in_dim + col] : 0.0f;
        const __bf16 hi = (__bf16)v, lo = (__bf16)(v - (float)hi);
        wp[((((size_t)o * KS + s) * 2 + 0) * 64 + l) * 8 + i] = hi;
        wp[((((size_t)o * KS + s) * 2 + 1) * 64 + l) * 8 + i] = lo;
    }
}
__global__ void k_policy_pack_wide(const float *__restrict__ Wt, const float *__restrict__ bias, int in_dim, int out_dim, int KS, int OT, int first,
                                   __bf16 *__restrict__ wp, float *__restrict__ bp) {
    pack_wide_body(Wt, bias, in_dim, out_dim, KS, OT, first, wp, bp);
}
// all layers of up to two nets in one launch (the learner re-packs after every optimiser step): blockIdx.y = layer, blockIdx.z = net
struct ChainPackArgs { const float *W[2][4], *b[2][4]; __bf16 *wp[2][4]; float *bp[2][4]; int in_dim[2][4], out_dim[2][4], KS[2][4], OT[2][4]; };
__global__ void k_chain_pack(const ChainPackArgs P) {
    const int l = blockIdx.y, z = blockIdx.z;
    pack_wide_body(P.W[z][l], P.b[z][l], P.in_dim[z][l], P.out_dim[z][l], P.KS[z][l], P.OT[z][l], l == 0 ? 1 : 0, P.wp[z][l], P.bp[z][l]);
}

}  // namespace lg
