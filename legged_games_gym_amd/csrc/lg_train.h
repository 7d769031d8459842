// lg_train.h -- PPO mini-batch MLP forward / backward on the matrix cores (the learner half of the rollout+update loop).
//
// Stands in for the autograd pass over rsl_rl's ActorCritic MLPs ([EXTERNAL]; Linear/ELU x3 + Linear, dims from reference
// legged_robot_config.py:204-209) inside PPO.update(): y = net(x[rows]) and, given dL/dy, the gradients of all weights
// and biases.  The weights are read in torch's own [out, in] layout (they change every optimiser step, so nothing is
// re-packed), activations never leave LDS, and the weight gradients are accumulated in MFMA accumulators across all the
// row tiles a workgroup walks, then written once as a per-workgroup partial that k_mlp_reduce sums in a fixed order
// (deterministic, no atomics).
//
// Tile algebra (v_mfma_f32_16x16x4_f32, D[i][j] += A[i][k] B[k][j]; A: lane l holds A[l&15][l>>4], B: lane l holds
// B[l>>4][l&15], D: lane l holds D[4(l>>4)+c][l&15]):
//   forward   x_{l+1}[f][row] = W[f][k] x_l[k][row]           A = W (float4 per lane per input tile), B = x_l tile
//   backward  dx_l[i][row]    = W^T[i][f] g_{l+1}[f][row]     A = W read column-wise,                  B = g tile
//   weights   dW[f][i]        = g_{l+1}[f][row] x_l[i][row]   A = g^T, B = x^T (feature-major copies kept in LDS, padded)
//   biases    db[f]           = g_{l+1}[f][row] * 1           same as dW with an all-ones B operand (an extra "input tile")
// where g is the gradient w.r.t. the pre-activation (ELU' folded in from the stored post-activation: x > 0 ? 1 : x + 1).
#pragma once
#include "lg_policy.h"

namespace lg {

#define LG_TRAIN_WAVES 4
#define LG_TT 17                       // padded row stride of the feature-major tile copies (bank-conflict free)

struct MlpNetArgs {
    const float *w[4], *b[4];          // torch Linear weights [out, in] / biases
    const float *x;                    // [R, dims[0]] input rows
    float *y;                          // forward: [mb, dims[4]]
    const float *dy;                   // backward: [mb, dims[4]]
    float *partial;                    // backward: [workgroups][grad_floats] per-workgroup partial sums
    int32_t dims[5];
    int32_t grad_floats;               // sum over layers of out*in + out
};
struct MlpArgs {
    MlpNetArgs net[2];
    const int64_t *rows;               // [mb] row indices into x, or null for 0..mb-1
    int32_t mb, n_tiles;
};

// feature-major copy of a tile: xt[feature 16][LG_TT] <- lane (row l&15, group g) holds features 4g..4g+3
LG_DEV void tile_store(float4 (*x)[64], float (*xt)[16][LG_TT], int tile, int lane, float4 v) {
    x[tile][lane] = v;
    const int g = lane >> 4, r = lane & 15;
    xt[tile][4 * g + 0][r] = v.x; xt[tile][4 * g + 1][r] = v.y; xt[tile][4 * g + 2][r] = v.z; xt[tile][4 * g + 3][r] = v.w;
}

// x_out = act(W x_in + b); W [out_dim, in_dim] row-major.  FIRST: in_dim is arbitrary (guarded scalar loads), otherwise a
// multiple of 16 (float4 loads).
template <int IN_T, int OUT_T, bool ACT, bool FIRST, bool KEEP_T>
LG_DEV void train_forward_layer(const float *__restrict__ W, const float *__restrict__ b, int in_dim, int out_dim,
                                const float4 (*xin)[64], float4 (*xout)[64], float (*xoutT)[16][LG_TT], int wave, int lane) {
    const int g = lane >> 4;
#pragma unroll 1
    for (int o = wave; o < OUT_T; o += LG_TRAIN_WAVES) {
        const int row = 16 * o + (lane & 15);
        const bool rok = row < out_dim;
        const float *wr = W + (size_t)(rok ? row : 0) * in_dim + 4 * g;
        float4 wv[IN_T];
#pragma unroll
        for (int t = 0; t < IN_T; t++) {
            if (FIRST) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; r++) { const int k = 16 * t + 4 * g + r; v[r] = (rok && k < in_dim) ? wr[16 * t + r] : 0.0f; }
                wv[t] = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                wv[t] = rok ? *reinterpret_cast<const float4 *>(wr + 16 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        f32x4 acc;
#pragma unroll
        for (int r = 0; r < 4; r++) { const int f = 16 * o + 4 * g + r; acc[r] = f < out_dim ? b[f] : 0.0f; }
#pragma unroll
        for (int t = 0; t < IN_T; t++) {
            const float4 xv = xin[t][lane];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t].x, xv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t].y, xv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t].z, xv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t].w, xv.w, acc, 0, 0, 0);
        }
        const float4 res = ACT ? make_float4(elu1(acc[0]), elu1(acc[1]), elu1(acc[2]), elu1(acc[3])) : make_float4(acc[0], acc[1], acc[2], acc[3]);
        if (KEEP_T) tile_store(xout, xoutT, o, lane, res); else xout[o][lane] = res;
    }
}

// g_in = (W^T g_out) * elu'(x_in): OUT_T gradient tiles -> IN_T gradient tiles (x_in = stored post-activation of the layer input)
// (KEEP_B: also keep the MFMA B-operand form of the result -- not needed for the first hidden layer, whose input gets no gradient)
template <int IN_T, int OUT_T, bool KEEP_B>
LG_DEV void train_backward_layer(const float *__restrict__ W, int in_dim, int out_dim, const float4 (*gout)[64], const float4 (*xin)[64],
                                 float4 (*gin)[64], float (*ginT)[16][LG_TT], int wave, int lane) {
    const int g = lane >> 4;
#pragma unroll 1
    for (int ti = wave; ti < IN_T; ti += LG_TRAIN_WAVES) {
        const float *wc = W + 16 * ti + (lane & 15);             // column of W = row of W^T; hidden widths are multiples of 16
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < OUT_T; o++) {
            float a[4];
#pragma unroll
            for (int r = 0; r < 4; r++) { const int k = 16 * o + 4 * g + r; a[r] = k < out_dim ? wc[(size_t)k * in_dim] : 0.0f; }
            const float4 gv = gout[o][lane];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], gv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], gv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], gv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], gv.w, acc, 0, 0, 0);
        }
        const float4 xv = xin[ti][lane];
        const float4 res = make_float4(acc[0] * (xv.x > 0.f ? 1.f : xv.x + 1.f), acc[1] * (xv.y > 0.f ? 1.f : xv.y + 1.f),
                                       acc[2] * (xv.z > 0.f ? 1.f : xv.z + 1.f), acc[3] * (xv.w > 0.f ? 1.f : xv.w + 1.f));
        if (KEEP_B) gin[ti][lane] = res;
        const int r = lane & 15;
        ginT[ti][4 * g + 0][r] = res.x; ginT[ti][4 * g + 1][r] = res.y; ginT[ti][4 * g + 2][r] = res.z; ginT[ti][4 * g + 3][r] = res.w;
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
            const float *ga = &gT[o][f][r0];
            if (t < IN_T) {
                const float *xb = &xT[t][f][r0];
#pragma unroll
                for (int s = 0; s < 4; s++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s], xb[s], acc[i], 0, 0, 0);
            } else {
#pragma unroll
                for (int s = 0; s < 4; s++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s], 1.0f, acc[i], 0, 0, 0);
            }
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

// One workgroup walks row tiles blockIdx.x, blockIdx.x + gridDim.x, ... of net blockIdx.y.
template <int D0T, int D1T, int D2T, int D3T, bool BWD>
__global__ void __launch_bounds__(64 * LG_TRAIN_WAVES) k_mlp_train(const MlpArgs A) {
    constexpr int XT = D0T + D1T + D2T + D3T, GT = D1T + D2T + D3T + 1;
    constexpr int X0 = 0, X1 = D0T, X2 = D0T + D1T, X3 = D0T + D1T + D2T;        // activation tile offsets
    constexpr int G1 = 0, G2 = D1T, G3 = D1T + D2T, G4 = D1T + D2T + D3T;        // gradient tiles w.r.t. x1, x2, x3 pre-acts, and y
    constexpr int B2 = 0, B3 = D2T, B4 = D2T + D3T, BT = D2T + D3T + 1;          // B-operand copies exist for g2, g3, dy only
    __shared__ float4 x[XT][64];
    __shared__ float  xT[BWD ? XT : 1][16][LG_TT];
    __shared__ float4 gr[BWD ? BT : 1][64];
    __shared__ float  gT[BWD ? GT : 1][16][LG_TT];
    const MlpNetArgs &N = A.net[blockIdx.y];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4;
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
#pragma unroll 1
    for (int rt = blockIdx.x; rt < A.n_tiles; rt += gridDim.x) {
        const int r = rt * 16 + (lane & 15);
        const bool live = r < A.mb;
        const int64_t src = live ? (A.rows ? A.rows[r] : (int64_t)r) : (A.rows ? A.rows[A.mb - 1] : (int64_t)(A.mb - 1));
        const float *xr = N.x + (size_t)src * d0;
        for (int t = wave; t < D0T; t += LG_TRAIN_WAVES) {
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; c++) { const int k = 16 * t + 4 * g + c; v[c] = k < d0 ? xr[k] : 0.0f; }
            const float4 xv = make_float4(v[0], v[1], v[2], v[3]);
            if (BWD) tile_store(x + X0, xT + X0, t, lane, xv); else x[X0 + t][lane] = xv;
        }
        __syncthreads();
        train_forward_layer<D0T, D1T, true, true, BWD>(N.w[0], N.b[0], d0, d1, x + X0, x + X1, xT + (BWD ? X1 : 0), wave, lane);
        __syncthreads();
        train_forward_layer<D1T, D2T, true, false, BWD>(N.w[1], N.b[1], d1, d2, x + X1, x + X2, xT + (BWD ? X2 : 0), wave, lane);
        __syncthreads();
        train_forward_layer<D2T, D3T, true, false, BWD>(N.w[2], N.b[2], d2, d3, x + X2, x + X3, xT + (BWD ? X3 : 0), wave, lane);
        if (BWD && wave == LG_TRAIN_WAVES - 1) {                   // dL/dy tile (zero for rows past the batch: they contribute nothing)
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; c++) { const int k = 4 * g + c; v[c] = (live && k < d4) ? N.dy[(size_t)r * d4 + k] : 0.0f; }
            tile_store(gr + B4, gT + G4, 0, lane, make_float4(v[0], v[1], v[2], v[3]));
        }
        __syncthreads();
        if (!BWD) {
            if (wave == 0) {
                // the output layer reuses x tile X0 as scratch (its inputs are no longer needed in forward-only mode)
                train_forward_layer<D3T, 1, false, false, false>(N.w[3], N.b[3], d3, d4, x + X3, x + X0, nullptr, 0, lane);
                const float4 yv = x[X0][lane];
                const float y4[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
                for (int c = 0; c < 4; c++) { const int k = 4 * g + c; if (live && k < d4) N.y[(size_t)r * d4 + k] = y4[c]; }
            }
            __syncthreads();
            continue;
        }
        // output layer: dW3 / db3 and g3 = (W3^T dy) * elu'(x3)
        train_backward_layer<D3T, 1, true>(N.w[3], d3, d4, gr + B4, x + X3, gr + B3, gT + G3, wave, lane);
        train_weight_grad<D3T, 1>(gT + G4, xT + X3, a3, wave, lane);
        __syncthreads();
        train_backward_layer<D2T, D3T, true>(N.w[2], d2, d3, gr + B3, x + X2, gr + B2, gT + G2, wave, lane);
        train_weight_grad<D2T, D3T>(gT + G3, xT + X2, a2, wave, lane);
        __syncthreads();
        train_backward_layer<D1T, D2T, false>(N.w[1], d1, d2, gr + B2, x + X1, nullptr, gT + G1, wave, lane);
        train_weight_grad<D1T, D2T>(gT + G2, xT + X1, a1, wave, lane);
        __syncthreads();
        train_weight_grad<D0T, D1T>(gT + G1, xT + X0, a0, wave, lane);
        __syncthreads();
    }
    if (BWD) {
        float *part = N.partial + (size_t)blockIdx.x * N.grad_floats;
        train_flush<D0T, D1T>(a0, part, d0, d1, wave, lane); part += (size_t)d1 * d0 + d1;
        train_flush<D1T, D2T>(a1, part, d1, d2, wave, lane); part += (size_t)d2 * d1 + d2;
        train_flush<D2T, D3T>(a2, part, d2, d3, wave, lane); part += (size_t)d3 * d2 + d3;
        train_flush<D3T, 1>(a3, part, d3, d4, wave, lane);
    }
}

struct MlpReduceArgs {
    const float *partial[2];
    float *gw[2][4], *gb[2][4];
    int32_t dims[2][5];
    int32_t grad_floats[2];
    int32_t n_partials;
};
// grads = sum over workgroups of the partials, in a fixed order; thread j owns flat gradient element j of net blockIdx.y
__global__ void __launch_bounds__(256) k_mlp_reduce(const MlpReduceArgs A) {
    const int n = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int gf = A.grad_floats[n];
    if (j >= gf) return;
    const float *p = A.partial[n] + j;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 4 <= A.n_partials; k += 4) {
        s0 += p[(size_t)(k + 0) * gf]; s1 += p[(size_t)(k + 1) * gf];
        s2 += p[(size_t)(k + 2) * gf]; s3 += p[(size_t)(k + 3) * gf];
    }
    for (; k < A.n_partials; k++) s0 += p[(size_t)k * gf];
    const float s = (s0 + s1) + (s2 + s3);
    int off = j;
#pragma unroll
    for (int l = 0; l < 4; l++) {
        const int nw = A.dims[n][l + 1] * A.dims[n][l], nb = A.dims[n][l + 1];
        if (off < nw) { A.gw[n][l][off] = s; return; }
        off -= nw;
        if (off < nb) { A.gb[n][l][off] = s; return; }
        off -= nb;
    }
}

}  // namespace lg
