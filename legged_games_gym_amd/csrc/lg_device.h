// lg_device.h -- device-side building blocks of the fused legged-robot step (gfx950, wave64).
//
// Layout: K lanes per environment, one lane per limb ("limb-per-lane").  A limb
// is a serial chain of L revolute joints hanging off the floating base; the
// lane owns that chain's kinematics, articulated-body recursion, contacts,
// actuator-net rows, reward partial sums and observation slots.  The base is
// shared: the K lanes of an environment butterfly-sum their limb's articulated
// inertia / bias force with DPP quad permutes and every lane then solves the
// same 6x6 base system redundantly (no broadcast, no LDS, no divergence).
//
// Math follows DESIGN.md "Physics step"; the CPU oracle (oracle/lg_oracle.c)
// is an independent scalar coding of the same equations.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/legged_hip.h"

#define LG_DEV __device__ __forceinline__

namespace lg {

// ------------------------------------------------------------------ small vectors
struct V3 { float x, y, z; };
LG_DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
LG_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
LG_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
LG_DEV V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
LG_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
LG_DEV V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
struct M3 { float m[9]; };                       // row-major
LG_DEV V3 mul(const M3 &R, V3 a) {
    return v3(R.m[0] * a.x + R.m[1] * a.y + R.m[2] * a.z, R.m[3] * a.x + R.m[4] * a.y + R.m[5] * a.z,
              R.m[6] * a.x + R.m[7] * a.y + R.m[8] * a.z);
}
LG_DEV M3 mul(const M3 &A, const M3 &B) {
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
    return C;
}
LG_DEV V3 symv(const float *S, V3 a) {           // (xx,xy,xz,yy,yz,zz)
    return v3(S[0] * a.x + S[1] * a.y + S[2] * a.z, S[1] * a.x + S[3] * a.y + S[4] * a.z, S[2] * a.x + S[4] * a.y + S[5] * a.z);
}
struct S6 { V3 w, v; };                          // spatial motion (w,v) / force (n,f)
LG_DEV S6 operator+(S6 a, S6 b) { S6 r; r.w = a.w + b.w; r.v = a.v + b.v; return r; }
LG_DEV S6 operator*(S6 a, float k) { S6 r; r.w = a.w * k; r.v = a.v * k; return r; }
LG_DEV float dot(S6 a, S6 b) { return dot(a.w, b.w) + dot(a.v, b.v); }

// articulated inertia [[A,H],[H^T,M]] about the base origin, world axes
struct AI { float A[6], H[9], M[6]; };
LG_DEV void ai_add(AI &a, const AI &b) {
#pragma unroll
    for (int i = 0; i < 6; i++) { a.A[i] += b.A[i]; a.M[i] += b.M[i]; }
#pragma unroll
    for (int i = 0; i < 9; i++) a.H[i] += b.H[i];
}
LG_DEV void ai_add_point(AI &I, float m, V3 c) {
    float cc = dot(c, c);
    I.A[0] += m * (cc - c.x * c.x); I.A[1] -= m * c.x * c.y; I.A[2] -= m * c.x * c.z;
    I.A[3] += m * (cc - c.y * c.y); I.A[4] -= m * c.y * c.z; I.A[5] += m * (cc - c.z * c.z);
    I.H[1] -= m * c.z; I.H[2] += m * c.y; I.H[3] += m * c.z; I.H[5] -= m * c.x; I.H[6] -= m * c.y; I.H[7] += m * c.x;
    I.M[0] += m; I.M[3] += m; I.M[5] += m;
}
LG_DEV void ai_add_rank1(AI &I, float k, V3 gw, V3 gv) {
    float w[3] = {gw.x, gw.y, gw.z}, v[3] = {gv.x, gv.y, gv.z};
    int t = 0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = i; j < 3; j++, t++) { I.A[t] += k * w[i] * w[j]; I.M[t] += k * v[i] * v[j]; }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) I.H[3 * i + j] += k * w[i] * v[j];
}
LG_DEV S6 ai_mul(const AI &I, S6 s) {
    S6 r;
    M3 H; for (int i = 0; i < 9; i++) H.m[i] = I.H[i];
    r.w = symv(I.A, s.w) + mul(H, s.v);
    r.v = v3(I.H[0] * s.w.x + I.H[3] * s.w.y + I.H[6] * s.w.z, I.H[1] * s.w.x + I.H[4] * s.w.y + I.H[7] * s.w.z,
             I.H[2] * s.w.x + I.H[5] * s.w.y + I.H[8] * s.w.z) + symv(I.M, s.v);
    return r;
}

// U = I (w, 0): the joint motion subspace is a pure rotation about the body's own reference point
LG_DEV S6 ai_mul_w(const AI &I, V3 w) {
    S6 r;
    r.w = symv(I.A, w);
    r.v = v3(I.H[0] * w.x + I.H[3] * w.y + I.H[6] * w.z, I.H[1] * w.x + I.H[4] * w.y + I.H[7] * w.z,
             I.H[2] * w.x + I.H[5] * w.y + I.H[8] * w.z);
    return r;
}
// Re-express (I, p) given about point P at Q = P - d:  H' = H + [d]x M,  A' = A + [d]x H^T - H'[d]x,  n' = n + d x f
LG_DEV void ai_shift(AI &I, S6 &p, V3 d) {
    const int ix[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    float T[9];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        V3 x = cross(d, v3(I.M[ix[0][j]], I.M[ix[1][j]], I.M[ix[2][j]]));
        T[j] = I.H[j] + x.x; T[3 + j] = I.H[3 + j] + x.y; T[6 + j] = I.H[6 + j] + x.z;
    }
    V3 dxh[3], txd[3];
#pragma unroll
    for (int j = 0; j < 3; j++) dxh[j] = cross(d, v3(I.H[3 * j], I.H[3 * j + 1], I.H[3 * j + 2]));
#pragma unroll
    for (int i = 0; i < 3; i++) txd[i] = cross(v3(T[3 * i], T[3 * i + 1], T[3 * i + 2]), d);
    float A2[6];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = i; j < 3; j++) {
            float a = (i == 0) ? dxh[j].x : (i == 1) ? dxh[j].y : dxh[j].z;
            float b = (j == 0) ? txd[i].x : (j == 1) ? txd[i].y : txd[i].z;
            A2[ix[i][j]] = I.A[ix[i][j]] + a - b;
        }
#pragma unroll
    for (int i = 0; i < 6; i++) I.A[i] = A2[i];
#pragma unroll
    for (int i = 0; i < 9; i++) I.H[i] = T[i];
    p.w = p.w + cross(d, p.v);
}

// SPD 6x6 solve by LDL^T, fully unrolled (registers only)
LG_DEV bool solve6(const AI &I, const float *b, float *x) {
    float m[6][6];
    const int ix[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            m[i][j] = I.A[ix[i][j]]; m[i + 3][j + 3] = I.M[ix[i][j]];
            m[i][j + 3] = I.H[3 * i + j]; m[j + 3][i] = I.H[3 * i + j];
        }
    float Lm[6][6], D[6], rD[6];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        float d = m[j][j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= Lm[j][k] * Lm[j][k] * D[k];
        ok = ok && (d > 0.0f);
        D[j] = d;
        rD[j] = __builtin_amdgcn_rcpf(d);            // 6 v_rcp instead of 21 divisions (1 ulp)
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            float v = m[i][j];
#pragma unroll
            for (int k = 0; k < j; k++) v -= Lm[i][k] * Lm[j][k] * D[k];
            Lm[i][j] = v * rD[j];
        }
    }
    float y[6];
#pragma unroll
    for (int i = 0; i < 6; i++) { float v = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) v -= Lm[i][k] * y[k];
        y[i] = v; }
#pragma unroll
    for (int i = 0; i < 6; i++) y[i] *= rD[i];
#pragma unroll
    for (int i = 5; i >= 0; i--) { float v = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) v -= Lm[k][i] * x[k];
        x[i] = v; }
    return ok;
}

LG_DEV M3 quat_to_mat(const float *q) {
    float x = q[0], y = q[1], z = q[2], w = q[3];
    M3 R;
    R.m[0] = 1 - 2 * (y * y + z * z); R.m[1] = 2 * (x * y - z * w);     R.m[2] = 2 * (x * z + y * w);
    R.m[3] = 2 * (x * y + z * w);     R.m[4] = 1 - 2 * (x * x + z * z); R.m[5] = 2 * (y * z - x * w);
    R.m[6] = 2 * (x * z - y * w);     R.m[7] = 2 * (y * z + x * w);     R.m[8] = 1 - 2 * (x * x + y * y);
    return R;
}
// isaacgym.torch_utils.quat_rotate_inverse / quat_apply ([EXTERNAL]; legged_robot.py:119-121,338)
LG_DEV V3 quat_rotate_inverse(const float *q, V3 v) {
    float qw = q[3]; V3 qv = v3(q[0], q[1], q[2]);
    V3 a = v * (2.0f * qw * qw - 1.0f);
    V3 b = cross(qv, v) * (qw * 2.0f);
    V3 c = qv * (dot(qv, v) * 2.0f);
    return (a - b) + c;
}
LG_DEV V3 quat_apply(const float *q, V3 b) {
    V3 xyz = v3(q[0], q[1], q[2]);
    V3 t = cross(xyz, b) * 2.0f;
    return (b + t * q[3]) + cross(xyz, t);
}

// ------------------------------------------------------------------ cross-lane (DPP quad permutes; K in {1,2,4})
template <int CTRL> LG_DEV float dpp(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xF, 0xF, true));
}
template <int K> LG_DEV float group_sum(float x) {      // butterfly; every lane of the group gets (x0+x1)+(x2+x3)
    if (K >= 2) x = x + dpp<0xB1>(x);                   // quad_perm [1,0,3,2]
    if (K >= 4) x = x + dpp<0x4E>(x);                   // quad_perm [2,3,0,1]
    return x;
}
template <int K> LG_DEV int group_or(int x) {
    if (K >= 2) x |= __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);
    if (K >= 4) x |= __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true);
    return x;
}
template <int K> LG_DEV void group_sum(AI &I, S6 &p) {
#pragma unroll
    for (int i = 0; i < 6; i++) { I.A[i] = group_sum<K>(I.A[i]); I.M[i] = group_sum<K>(I.M[i]); }
#pragma unroll
    for (int i = 0; i < 9; i++) I.H[i] = group_sum<K>(I.H[i]);
    p.w.x = group_sum<K>(p.w.x); p.w.y = group_sum<K>(p.w.y); p.w.z = group_sum<K>(p.w.z);
    p.v.x = group_sum<K>(p.v.x); p.v.y = group_sum<K>(p.v.y); p.v.z = group_sum<K>(p.v.z);
}

// ------------------------------------------------------------------ Philox4x32-10 (same stream as the oracle)
enum { RNG_NOISE = 0, RNG_CMD_STEP = 1, RNG_CMD_RESET = 2, RNG_DOF = 3, RNG_ROOT = 4, RNG_PUSH = 5, RNG_TERRAIN = 6, RNG_NOISE_H = 7 };
LG_DEV void rand4(uint64_t seed, int env, int64_t step, int purpose, int block, float u[4]) {
    uint32_t c0 = (uint32_t)env, c1 = (uint32_t)step, c2 = (uint32_t)purpose, c3 = (uint32_t)block;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    u[0] = (float)(c0 >> 8) * (1.0f / 16777216.0f); u[1] = (float)(c1 >> 8) * (1.0f / 16777216.0f);
    u[2] = (float)(c2 >> 8) * (1.0f / 16777216.0f); u[3] = (float)(c3 >> 8) * (1.0f / 16777216.0f);
}
LG_DEV float urange(float lo, float hi, float u) { return (hi - lo) * u + lo; }

// ------------------------------------------------------------------ ANYdrive actuator net (anymal.py:71-78) on the matrix cores
// TorchScript LSTMsea: x*in_scale -> LSTM(2,8,2 layers; gates i,f,g,o) -> out_scale*Linear(8,1), one row per joint.
//
// MFMA mapping (v_mfma_f32_32x32x2_f32, exact fp32 fma chains): D[gate 0..31][batch col] += A[gate][k] * B[k][col].
//  * A = weights: lane l holds W[gate = l&31][k-th input of the step, k = l>>5]; 15 VGPRs hold the whole net
//    (bias rides along as a K-step against the constant 1).
//  * The batch is the wave itself: 64 lanes = 64 joint rows, processed as two 32-column groups g (lanes 32g..32g+31).
//  * Hidden/cell vectors live "unit-split": register s of group g holds unit s of row (32g + l) on lanes l < 32 and
//    unit s+4 of row (32g + l - 32) on lanes l >= 32.  That is exactly the D layout the MFMA produces (lanes < 32 get
//    gate rows {0-3, 8-11, 16-19, 24-27} = i,f,g,o of units 0-3, lanes >= 32 units 4-7) AND the B layout the next
//    MFMA consumes (k = l>>5 selects unit s vs s+4), so layer 0 -> layer 1 -> next time step needs no data movement.
//  * Per-lane <-> unit-split conversion is one v_permlane32_swap per register pair, at kernel entry / exit only.
typedef float f32x16 __attribute__((ext_vector_type(16)));
enum { LW_B0 = 0, LW_X = 1, LW_H0 = 2, LW_B1 = 6, LW_I1 = 7, LW_H1 = 11, LW_LIN = 15, LW_LB = 19, LW_OUT = 20, LW_ROWS = 21 };
struct LstmLane { float a[15]; float lw[4]; float lb, out_scale; };      // this lane's slice of the net
LG_DEV LstmLane lstm_load(const float *table /* [LW_ROWS][64] */, int lane) {
    LstmLane W;
#pragma unroll
    for (int i = 0; i < 15; i++) W.a[i] = table[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 4; i++) W.lw[i] = table[(LW_LIN + i) * 64 + lane];
    W.lb = table[LW_LB * 64 + lane]; W.out_scale = table[LW_OUT * 64 + lane];
    return W;
}
LG_DEV void swap32(float &a, float &b) {        // a <- [a.lo | b.lo], b <- [a.hi | b.hi]
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
// packed-fp32 pairs (v_pk_add/mul/fma_f32): two hidden units per instruction; exp2 / rcp stay per component
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define LSTM_EXP_CAP 0x1p30f   // sigma saturates at 9e-10 and a product of three (1 + e) terms stays far below FLT_MAX
LG_DEV f32x2 exp2_capped(f32x2 a) {     // capped after the exp (2^x -> inf is fine as a min operand, and needs no input canonicalisation)
    return f32x2{fminf(__builtin_amdgcn_exp2f(a.x), LSTM_EXP_CAP), fminf(__builtin_amdgcn_exp2f(a.y), LSTM_EXP_CAP)};
}
LG_DEV f32x2 rcp2(f32x2 a) { return f32x2{__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)}; }

// per-lane 8-vector (u[0..7] of this lane's row) -> unit-split pair (g0[s], g1[s]); the same call inverts it
LG_DEV void lstm_split(const float (&u)[8], float (&g0)[4], float (&g1)[4]) {
#pragma unroll
    for (int s = 0; s < 4; s++) { g0[s] = u[s]; g1[s] = u[s + 4]; swap32(g0[s], g1[s]); }
}
LG_DEV void lstm_unsplit(const float (&g0)[4], const float (&g1)[4], float (&u)[8]) {
#pragma unroll
    for (int s = 0; s < 4; s++) { u[s] = g0[s]; u[s + 4] = g1[s]; swap32(u[s], u[s + 4]); }
}
struct LstmSplit { float h0[2][4], c0[2][4], h1[2][4], c1[2][4]; };      // [group][s]

// LSTM cell on pre-activations that arrive PRE-SCALED by the weight table (build_lstm_table): rows i, f, o hold
// -log2(e) x and rows g hold -2 log2(e) x, so sigma(x) = 1 / (1 + 2^a) and tanh(x) = (1 - 2^a) / (1 + 2^a) need no
// multiply.  The five activations of a unit share two reciprocals (transcendentals are quarter rate):
//   c' = sigma(f) c + sigma(i) tanh(g) = [c A G + (1 - e_g) F] / (F A G),   A = 1 + e_i, F = 1 + e_f, G = 1 + e_g
//   h  = sigma(o) tanh(c')             = (1 - e_c) / (O C),                 O = 1 + e_o, C = 1 + e_c, e_c = 2^(-2 log2(e) c')
LG_DEV void lstm_cell(const f32x16 &g, float (&h)[4], float (&c)[4]) {
#pragma unroll
    for (int u = 0; u < 4; u += 2) {
        const f32x2 one = {1.0f, 1.0f}, two = {2.0f, 2.0f};
        f32x2 A = one + exp2_capped(f32x2{g[u], g[u + 1]});
        f32x2 F = one + exp2_capped(f32x2{g[4 + u], g[5 + u]});
        f32x2 G = one + exp2_capped(f32x2{g[8 + u], g[9 + u]});
        f32x2 O = one + exp2_capped(f32x2{g[12 + u], g[13 + u]});
        f32x2 cc = {c[u], c[u + 1]};
        f32x2 t = A * G;
        f32x2 cn = (cc * t + (two - G) * F) * rcp2(F * t);
        f32x2 C = one + exp2_capped(cn * -2.885390082f);
        f32x2 hn = (two - C) * rcp2(O * C);
        c[u] = cn.x; c[u + 1] = cn.y; h[u] = hn.x; h[u + 1] = hn.y;
    }
}
// One time step for the wave's 64 rows.  (pos_err, vel) are this lane's row; returns this lane's torque.
// Must be executed with all 64 lanes active.
LG_DEV float actuator_step_mfma(const LstmLane &W, float pos_err, float vel, LstmSplit &st) {
    float xg0 = pos_err, xg1 = vel;               // in_scale is folded into the A operand (exact: 2 and 0.25)
    swap32(xg0, xg1);                             // xg0 = B operand of group 0, xg1 of group 1
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float part[2];
#pragma unroll
    for (int g = 0; g < 2; g++) {
        f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W.a[LW_B0], 1.0f, zero, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W.a[LW_X], g ? xg1 : xg0, acc, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W.a[LW_H0 + s], st.h0[g][s], acc, 0, 0, 0);
        lstm_cell(acc, st.h0[g], st.c0[g]);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W.a[LW_B1], 1.0f, zero, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W.a[LW_I1 + s], st.h0[g][s], acc, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 4; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W.a[LW_H1 + s], st.h1[g][s], acc, 0, 0, 0);
        lstm_cell(acc, st.h1[g], st.c1[g]);
        float p = 0.0f;
#pragma unroll
        for (int u = 0; u < 4; u++) p += W.lw[u] * st.h1[g][u];
        part[g] = p;
    }
    swap32(part[0], part[1]);                     // lanes < 32: (units 0-3, units 4-7) of group 0's row; lanes >= 32: group 1's
    return W.out_scale * ((part[0] + part[1]) + W.lb);
}


}  // namespace lg
