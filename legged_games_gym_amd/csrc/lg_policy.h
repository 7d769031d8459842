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
                           int64_t step, float (*lds_act)[16]) {
    const int g = lane >> 4;
    int env = block * 16 + (lane & 15);
    const bool live = env < A.num_envs;
    if (!live) env = A.num_envs - 1;
    // layer-0 B operands from global: k_g = 16t + 4g + r
    const float *o = A.obs + (size_t)env * A.num_obs;
    for (int t = wave; t < D0T; t += LG_POLICY_WAVES) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { int k = 16 * t + 4 * g + r; v[r] = k < A.num_obs ? o[k] : 0.0f; }
        xa[t][lane] = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    mlp_layer<D0T, D1T, true>(A.w[0], A.b[0], xa, xb, wave, lane);
    __syncthreads();
    mlp_layer<D1T, D2T, true>(A.w[1], A.b[1], xb, xa, wave, lane);
    __syncthreads();
    mlp_layer<D2T, D3T, true>(A.w[2], A.b[2], xa, xb, wave, lane);
    __syncthreads();
    if (wave != 0) return;
    mlp_layer<D3T, 1, false>(A.w[3], A.b[3], xb, xy, 0, lane);
    const float4 yv = xy[0][lane];                                 // written by this lane
    const float y[4] = {yv.x, yv.y, yv.z, yv.w};
    // lane (env, g) now holds mean[4g + r]; sample a = mean + std * eps  (Philox -> Box-Muller)
    float u[4];
    rand4(A.seed ^ 0x9E3779B97F4A7C15ull, env, step, 100 + g, 0, u);
    float rad0 = sqrtf(-2.0f * __logf(fmaxf(u[0], 1e-12f))), rad1 = sqrtf(-2.0f * __logf(fmaxf(u[2], 1e-12f)));
    float s0, c0, s1, c1;
    __sincosf(6.2831853f * u[1], &s0, &c0);
    __sincosf(6.2831853f * u[3], &s1, &c1);
    float eps[4] = {rad0 * c0, rad0 * s0, rad1 * c1, rad1 * s1};
#pragma unroll
    for (int r = 0; r < 4; r++) {
        int a = 4 * g + r;
        if (a < A.num_actions) {
            float m = y[r], act = A.deterministic ? m : m + A.std[a] * eps[r];
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

}  // namespace lg
