/*
 * lg_oracle.c -- CPU oracle for the legged-robot hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call this file.  The product (legged_games_gym_amd/) never
 * does: it fails loudly when the HIP extension is missing.
 *
 * Two halves:
 *  (1) TORCH-SIDE HALF -- a scalar, one-env-at-a-time restatement of
 *      reference legged_gym/envs/base/legged_robot.py:80-230, 329-444, 831-969,
 *      envs/anymal_c/anymal.py:56-81, envs/cassie/cassie.py:43-46,
 *      utils/math.py:38-48 and of the [EXTERNAL] isaacgym.torch_utils helpers
 *      (quat_rotate_inverse, quat_apply, torch_rand_float) it calls.  PINNED by
 *      tests/golden: G1 (actuator net: the reference's own weights through
 *      ATen's aten::lstm), G4 (post-physics block) and G5 (the reset / RNG half:
 *      post_physics_step as a whole, reset_idx, terrain / command curricula,
 *      command resampling, pushes, observation noise) -- all outputs of the
 *      reference's OWN method bodies, ast-extracted from /root/reference and
 *      executed by tools/make_golden.py, the random draws answered from this
 *      file's Philox stream by an independent numpy Philox4x32-10 (Random123
 *      known answers): tests/test_oracle_torch_side.py, test_oracle_reset_half.py.
 *  (2) PHYSICS HALF -- the reference delegates gym.simulate() to PhysX
 *      (closed source, absent): PARITY UNPINNED against PhysX itself.  The
 *      DYNAMICS of a sub-step are tied to the published equations of motion
 *      by tests/test_equation_of_motion.py (independent float64 recursive
 *      Newton-Euler on all four robots: residual <= 2e-5; single-joint
 *      oscillation period vs 2 pi sqrt(I / Kp)); the CONTACT model is defined
 *      here and validated by invariants (tests/test_oracle_physics.py).  This
 *      file *defines* the rigid-body step the HIP kernels must reproduce:
 *      floating-base articulated-body algorithm in world-aligned coordinates,
 *      every body's spatial quantities about its own joint origin, implicit (backward-Euler) spring-damper contacts
 *      and joint limits folded into the articulated inertias, implicit regularised
 *      Coulomb friction refined over two passes, semi-implicit Euler.
 *      See DESIGN.md "Physics step".
 *
 * Same C-ABI as include/legged_hip.h with prefix lgo_ and HOST pointers.
 * Plain C99, fp32 arithmetic throughout (double only where the reference does).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/legged_hip.h"

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ types */
typedef struct { float x, y, z; } v3;
typedef struct { float A[6], H[9], M[6]; } ai6;     /* [[A,H],[H^T,M]], A,M sym (xx,xy,xz,yy,yz,zz) */
typedef struct { v3 w, v; } sv6;                   /* spatial motion (w, v) or force (n, f) */

struct lgo_sim {
    lg_params      P;
    lg_robot_model R;
    float          W[LG_ACTUATOR_FLOATS];
    int            has_net;
    lg_buffers     B;
    int            threads;
};
typedef struct lgo_sim lgo_sim;

static char g_err[256] = "";
const char *lgo_last_error(void) { return g_err; }
int lgo_abi_version(void) { return LG_ABI_VERSION; }
int lgo_sizeof(int which) {
    switch (which) { case 0: return (int)sizeof(lg_params); case 1: return (int)sizeof(lg_robot_model);
                     case 2: return (int)sizeof(lg_buffers); case 3: return (int)sizeof(lg_point); default: return -1; }
}

/* ------------------------------------------------------------------ small vector algebra */
static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 scl(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 cross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline v3 mv(const float *R, v3 a) {     /* row-major 3x3 times vector */
    return V(R[0] * a.x + R[1] * a.y + R[2] * a.z, R[3] * a.x + R[4] * a.y + R[5] * a.z, R[6] * a.x + R[7] * a.y + R[8] * a.z);
}
static inline v3 symv(const float *S, v3 a) {   /* symmetric (xx,xy,xz,yy,yz,zz) times vector */
    return V(S[0] * a.x + S[1] * a.y + S[2] * a.z, S[1] * a.x + S[3] * a.y + S[4] * a.z, S[2] * a.x + S[4] * a.y + S[5] * a.z);
}
static inline void mm(const float *A, const float *B, float *C) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
        C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
static void quat_to_mat(const float *q, float *R) {   /* q = xyzw, assumed unit */
    float x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}
/* [EXTERNAL] isaacgym.torch_utils.quat_rotate_inverse, used at legged_robot.py:119-121 */
static v3 quat_rotate_inverse(const float *q, v3 v) {
    float qw = q[3]; v3 qv = V(q[0], q[1], q[2]);
    v3 a = scl(v, 2.0f * qw * qw - 1.0f);
    v3 b = scl(cross(qv, v), qw * 2.0f);
    v3 c = scl(qv, dot(qv, v) * 2.0f);
    return add(sub(a, b), c);
}
/* [EXTERNAL] isaacgym.torch_utils.quat_apply, used at legged_robot.py:338 and utils/math.py:42 */
static v3 quat_apply(const float *q, v3 b) {
    v3 xyz = V(q[0], q[1], q[2]);
    v3 t = scl(cross(xyz, b), 2.0f);
    return add(add(b, scl(t, q[3])), cross(xyz, t));
}

/* ------------------------------------------------------------------ Philox4x32-10 counter RNG */
/* The reference draws from torch's global generator in call order; bit parity
 * with that is neither possible nor needed.  Both the oracle and the HIP path
 * use this counter-based stream keyed by (seed; env, step, purpose, block). */
enum { RNG_NOISE = 0, RNG_CMD_STEP = 1, RNG_CMD_RESET = 2, RNG_DOF = 3, RNG_ROOT = 4, RNG_PUSH = 5,
       RNG_TERRAIN = 6, RNG_NOISE_H = 7 };
static inline void philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static inline float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }   /* [0,1), 24 bit like torch.rand */
static inline void rand4(const lgo_sim *s, int env, int64_t step, int purpose, int block, float u[4]) {
    uint32_t o[4];
    philox(s->P.seed, (uint32_t)env, (uint32_t)step, (uint32_t)purpose, (uint32_t)block, o);
    for (int i = 0; i < 4; i++) u[i] = u01(o[i]);
}
static inline float urange(float lo, float hi, float u) { return (hi - lo) * u + lo; }   /* torch_rand_float */

/* ------------------------------------------------------------------ actuator network (anymal.py:71-78) */
static inline float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

/* TorchScript LSTMsea.forward (anydrive_v3_lstm.pt: code/__torch__/models.py):
 *   x*in_scale -> LSTM(2,8,num_layers=2), gate order i,f,g,o -> out_scale*Linear(8,1) */
static float actuator_row(const float *W, float pos_err, float vel, float *h0, float *c0, float *h1, float *c1) {
    const float *in_scale = W, *out_scale = W + 2;
    const float *Wih0 = W + 3, *Whh0 = Wih0 + 64, *bih0 = Whh0 + 256, *bhh0 = bih0 + 32;
    const float *Wih1 = bhh0 + 32, *Whh1 = Wih1 + 256, *bih1 = Whh1 + 256, *bhh1 = bih1 + 32;
    const float *lw = bhh1 + 32, *lb = lw + 8;
    float x0 = pos_err * in_scale[0], x1 = vel * in_scale[1];
    float g[32], hn[8];
    for (int r = 0; r < 32; r++) {
        float a = Wih0[2 * r] * x0 + Wih0[2 * r + 1] * x1 + bih0[r];
        float b = bhh0[r];
        for (int k = 0; k < 8; k++) b += Whh0[8 * r + k] * h0[k];
        g[r] = a + b;
    }
    for (int u = 0; u < 8; u++) {
        float c = sigm(g[8 + u]) * c0[u] + sigm(g[u]) * tanhf(g[16 + u]);
        c0[u] = c; hn[u] = sigm(g[24 + u]) * tanhf(c);
    }
    for (int u = 0; u < 8; u++) h0[u] = hn[u];
    for (int r = 0; r < 32; r++) {
        float a = bih1[r], b = bhh1[r];
        for (int k = 0; k < 8; k++) { a += Wih1[8 * r + k] * h0[k]; b += Whh1[8 * r + k] * h1[k]; }
        g[r] = a + b;
    }
    for (int u = 0; u < 8; u++) {
        float c = sigm(g[8 + u]) * c1[u] + sigm(g[u]) * tanhf(g[16 + u]);
        c1[u] = c; hn[u] = sigm(g[24 + u]) * tanhf(c);
    }
    float y = lb[0];
    for (int u = 0; u < 8; u++) { h1[u] = hn[u]; y += lw[u] * hn[u]; }
    return out_scale[0] * y;
}

int lgo_actuator_forward(lgo_sim *s, const float *pos_err, const float *vel, float *torques,
                         float *hidden, float *cell, int32_t rows, void *stream) {
    (void)stream;
    if (!s->has_net) { snprintf(g_err, sizeof g_err, "no actuator weights"); return -2; }
    for (int r = 0; r < rows; r++)
        torques[r] = actuator_row(s->W, pos_err[r], vel[r], hidden + (size_t)r * 8, cell + (size_t)r * 8,
                                  hidden + ((size_t)rows + r) * 8, cell + ((size_t)rows + r) * 8);
    return 0;
}

/* ------------------------------------------------------------------ articulated-inertia helpers */
static void ai_zero(ai6 *I) { memset(I, 0, sizeof *I); }
static void ai_add(ai6 *a, const ai6 *b) {
    for (int i = 0; i < 6; i++) { a->A[i] += b->A[i]; a->M[i] += b->M[i]; }
    for (int i = 0; i < 9; i++) a->H[i] += b->H[i];
}
/* point mass `m` at c:  A += m(|c|^2 1 - c c^T), H += m [c]x, M += m 1 */
static void ai_add_point(ai6 *I, float m, v3 c) {
    float cc = dot(c, c);
    I->A[0] += m * (cc - c.x * c.x); I->A[1] -= m * c.x * c.y; I->A[2] -= m * c.x * c.z;
    I->A[3] += m * (cc - c.y * c.y); I->A[4] -= m * c.y * c.z; I->A[5] += m * (cc - c.z * c.z);
    I->H[1] -= m * c.z; I->H[2] += m * c.y; I->H[3] += m * c.z; I->H[5] -= m * c.x; I->H[6] -= m * c.y; I->H[7] += m * c.x;
    I->M[0] += m; I->M[3] += m; I->M[5] += m;
}
/* I += k * g g^T with g = (gw, gv) */
static void ai_add_rank1(ai6 *I, float k, v3 gw, v3 gv) {
    float w[3] = {gw.x, gw.y, gw.z}, v[3] = {gv.x, gv.y, gv.z};
    int t = 0;
    for (int i = 0; i < 3; i++) for (int j = i; j < 3; j++, t++) { I->A[t] += k * w[i] * w[j]; I->M[t] += k * v[i] * v[j]; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) I->H[3 * i + j] += k * w[i] * v[j];
}
static sv6 ai_mul(const ai6 *I, sv6 s) {      /* [[A,H],[H^T,M]] (w;v) */
    sv6 r;
    r.w = add(symv(I->A, s.w), mv(I->H, s.v));
    r.v = add(V(I->H[0] * s.w.x + I->H[3] * s.w.y + I->H[6] * s.w.z,
                I->H[1] * s.w.x + I->H[4] * s.w.y + I->H[7] * s.w.z,
                I->H[2] * s.w.x + I->H[5] * s.w.y + I->H[8] * s.w.z), symv(I->M, s.v));
    return r;
}
static inline float sdot(sv6 a, sv6 b) { return dot(a.w, b.w) + dot(a.v, b.v); }
static inline sv6 sadd(sv6 a, sv6 b) { sv6 r = {add(a.w, b.w), add(a.v, b.v)}; return r; }
static inline sv6 sscl(sv6 a, float k) { sv6 r = {scl(a.w, k), scl(a.v, k)}; return r; }

/* solve [[A,H],[H^T,M]] x = b for SPD 6x6 by LDL^T */
static int solve6(const ai6 *I, const float *b, float *x) {
    float m[6][6];
    static const int ix[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        m[i][j] = I->A[ix[i][j]]; m[i + 3][j + 3] = I->M[ix[i][j]];
        m[i][j + 3] = I->H[3 * i + j]; m[j + 3][i] = I->H[3 * i + j];
    }
    float L[6][6], D[6];
    for (int j = 0; j < 6; j++) {
        float d = m[j][j];
        for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k] * D[k];
        if (!(d > 0.0f)) return -1;
        D[j] = d;
        for (int i = j + 1; i < 6; i++) {
            float v = m[i][j];
            for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k] * D[k];
            L[i][j] = v / d;
        }
    }
    float y[6];
    for (int i = 0; i < 6; i++) { float v = b[i]; for (int k = 0; k < i; k++) v -= L[i][k] * y[k]; y[i] = v; }
    for (int i = 0; i < 6; i++) y[i] /= D[i];
    for (int i = 5; i >= 0; i--) { float v = y[i]; for (int k = i + 1; k < 6; k++) v -= L[k][i] * x[k]; x[i] = v; }
    return 0;
}

/* ------------------------------------------------------------------ terrain */
static inline float hf_at(const lgo_sim *s, int ix, int iy) {
    if (ix < 0) ix = 0; if (iy < 0) iy = 0;
    if (ix > s->P.hf_rows - 1) ix = s->P.hf_rows - 1;
    if (iy > s->P.hf_cols - 1) iy = s->P.hf_cols - 1;
    return (float)s->B.height_samples[(size_t)ix * s->P.hf_cols + iy] * s->P.hf_vertical_scale;
}
/* Ground contact of a collision sphere (centre at world (x, y, pz), radius): penetration depth and unit normal.  Plane: z = 0.
 * Height field: bilinear patch of the 4 surrounding samples (the collision surface of the built-in engine; the reference
 * hands the same int16 grid to PhysX, legged_robot.py:619-637).
 * hf_step_threshold > 0 ('trimesh', terrain.py:69-73: slopes above slope_treshold are "corrected to vertical surfaces" when the
 * samples are triangulated): along an axis whose height difference across the cell exceeds the threshold, the ramp is replaced
 * by the LOW side's level up to a vertical face at the HIGH side of the cell; a sphere whose centre is below the top of that face and
 * within reach of it touches the face with a horizontal normal, one whose centre is above the top touches the face's top EDGE (normal
 * from the edge point to the centre: a foot overhanging a stair edge is carried by it instead of dropping to the lower level).  The
 * deepest of {ground, x-face / x-edge, y-face / y-edge} is the point's contact. */
static void ground_contact(const lgo_sim *s, float x, float y, float pz, float radius, float *depth, v3 *n) {
    if (s->P.terrain_type == LG_TERRAIN_PLANE || !s->B.height_samples) { *n = V(0, 0, 1); *depth = radius - pz; return; }
    float inv = 1.0f / s->P.hf_horizontal_scale;
    float gx = (x + s->P.hf_border) * inv, gy = (y + s->P.hf_border) * inv;
    if (!(fabsf(gx) < 1e9f)) gx = 0.0f;            /* non-finite / absurd position: keep float -> int defined (the env resets) */
    if (!(fabsf(gy) < 1e9f)) gy = 0.0f;
    float fx = floorf(gx), fy = floorf(gy);
    int ix = (int)fx, iy = (int)fy;
    float tx = gx - fx, ty = gy - fy;
    float h00 = hf_at(s, ix, iy), h10 = hf_at(s, ix + 1, iy), h01 = hf_at(s, ix, iy + 1), h11 = hf_at(s, ix + 1, iy + 1);
    const float thr = s->P.hf_step_threshold;
    float wall_depth = -1e30f; v3 wall_n = V(0, 0, 1);
    if (thr > 0.0f) {
        const float hs = s->P.hf_horizontal_scale;
        /* x axis: step if either edge along x jumps by more than the threshold; "up" towards the side with the larger sum */
        if (fmaxf(fabsf(h10 - h00), fabsf(h11 - h01)) > thr) {
            const int up = (h10 + h11) > (h00 + h01);              /* rising towards +x */
            const float top = up ? h10 + (h11 - h10) * ty : h00 + (h01 - h00) * ty;       /* level behind the face (y-interpolated) */
            const float dist = (up ? 1.0f - tx : tx) * hs;         /* horizontal distance to the face */
            if (pz - radius < top) {
                float d = radius - dist; v3 fn = V(up ? -1.0f : 1.0f, 0, 0);
                if (pz > top) {              /* centre above the riser's top edge: the sphere touches the EDGE, normal from the edge point to the centre */
                    const float dz = pz - top, len = sqrtf(dist * dist + dz * dz), il = 1.0f / fmaxf(len, 1e-9f);
                    d = radius - len; fn = V((up ? -dist : dist) * il, 0, dz * il);
                }
                if (d > wall_depth) { wall_depth = d; wall_n = fn; }
            }
            if (up) { h10 = h00; h11 = h01; } else { h00 = h10; h01 = h11; }          /* the low level extends to the face */
        }
        if (fmaxf(fabsf(h01 - h00), fabsf(h11 - h10)) > thr) {
            const int up = (h01 + h11) > (h00 + h10);
            const float top = up ? h01 + (h11 - h01) * tx : h00 + (h10 - h00) * tx;
            const float dist = (up ? 1.0f - ty : ty) * hs;
            if (pz - radius < top) {
                float d = radius - dist; v3 fn = V(0, up ? -1.0f : 1.0f, 0);
                if (pz > top) {
                    const float dz = pz - top, len = sqrtf(dist * dist + dz * dz), il = 1.0f / fmaxf(len, 1e-9f);
                    d = radius - len; fn = V(0, (up ? -dist : dist) * il, dz * il);
                }
                if (d > wall_depth) { wall_depth = d; wall_n = fn; }
            }
            if (up) { h01 = h00; h11 = h10; } else { h00 = h01; h10 = h11; }
        }
    }
    float hx0 = h00 + (h10 - h00) * tx, hx1 = h01 + (h11 - h01) * tx;
    float h = hx0 + (hx1 - hx0) * ty;
    float dhdx = ((h10 - h00) + ((h11 - h01) - (h10 - h00)) * ty) * inv;
    float dhdy = (hx1 - hx0) * inv;
    float l = 1.0f / sqrtf(dhdx * dhdx + dhdy * dhdy + 1.0f);
    *n = V(-dhdx * l, -dhdy * l, l);
    *depth = radius - (pz - h) * n->z;
    if (wall_depth > *depth) { *depth = wall_depth; *n = wall_n; }
}

/* Torque a drive can still deliver at joint speed qd: fades linearly to zero over the last 10 % below the URDF velocity
 * limit when it would accelerate the joint further (no-load speed of the motor).  Without it a saturated controller on a
 * light link relies on the post-integration velocity clamp, which removes the link's momentum while the base keeps the
 * reaction: a momentum pump (observed: Cassie's pelvis spun up to 180 rad/s).  The clamp stays as a last-resort guard. */
#define LG_VEL_LIMIT_GAIN       100.0f
#define LG_MAX_LINEAR_VELOCITY  1000.0f     /* asset.max_linear_velocity  (legged_robot_config.py:117) */
#define LG_MAX_ANGULAR_VELOCITY 1000.0f     /* asset.max_angular_velocity (legged_robot_config.py:116) */
static inline v3 clamp_norm(v3 a, float lim) {
    float n2 = dot(a, a);
    if (!(n2 <= lim * lim)) { float k = (n2 > 0.0f && n2 < INFINITY) ? lim / sqrtf(n2) : 0.0f; return scl(a, k); }   /* also maps inf/NaN to 0 */
    return a;
}
static inline float motor_torque(float tau, float qd, float vlim) {
    if (vlim <= 0.0f || tau * qd <= 0.0f) return tau;
    float s = (vlim - fabsf(qd)) / (0.1f * vlim);
    return tau * fminf(fmaxf(s, 0.0f), 1.0f);
}


/* ------------------------------------------------------------------ self-collision (asset.self_collisions = 0, legged_robot.py:683)
 * PhysX collides the links of one articulation with each other when the asset's collision filter is 0 (anymal_c_flat_config.py:42),
 * except directly connected links.  The built-in engine's version (DESIGN.md "Self-collision"), defined here:
 *   - shapes: consecutive collision points on the same body with equal radius form a capsule (segment + radius), a single
 *     point is a sphere (ANYmal-C limb: KFE-drive capsule on the THIGH, shank capsule, foot sphere; base capsule);
 *   - pairs: every limb capsule against the base capsules and against the capsules of every OTHER limb (links of one limb
 *     are adjacent or out of each other's reach); closest points of the two segments, depth = ra + rb - distance.  Per limb
 *     BODY and partner (the base, each other limb) the DEEPEST active pair is the contact of this sub-step;
 *   - when: detection uses the state at the start of the sub-step; the contact terms enter the LAST articulated-body pass
 *     only (the HIP kernel detects on its helper waves while the rigid-body wave runs the first pass);
 *   - force: frictionless implicit spring-damper along the contact normal, bias s = max(0, K d - kappa v_n) with v_n the relative
 *     normal velocity at the start of the sub-step; active while the shapes overlap (d > 0), and in the speculative range
 *     -contact_margin < d <= 0 when K d - kappa v_n > 0 (approaching fast enough to touch within the step);
 *   - coupling: block Jacobi with mass-ratio weights.  The true contact force is implicit in the RELATIVE acceleration,
 *     f = s - kappa dt n.(a_A - a_B) with s = K d - kappa v_n; each body only knows its own acceleration, so body A uses
 *     f = s - kappa (1 + mA/mB) dt n.a_A and body B the mirror image (mA, mB: nominal masses behind the two shapes -- the
 *     limb sub-tree from the carrying joint outwards, the whole robot for the base).  For a single contact this reproduces the
 *     coupled solution exactly when the ratio is right, and for ANY positive ratio the two velocity corrections add up to
 *     exactly the relative normal velocity (no overshoot: plain Jacobi, ratio term dropped, makes stiff contacts swap the two
 *     bodies' velocities every sub-step).  The limb body gets dt kappa (1 + mA/mB) g g^T added to its rigid-body inertia and
 *     -J^T s to its bias force before the articulated-body passes; the base gets the reaction of each limb's DEEPEST base
 *     contact the same way (it rides with that limb's partial sums);
 *   - contact_forces: the exported net contact force of a body includes its self-collision forces (as PhysX's does), estimated
 *     as s / (1 + kappa dt (1/mA + 1/mB)) along the normal -- equal and opposite on the two bodies. */
typedef struct { v3 a0, a1, va0, va1; float rad, mhat; int body, report; } capsule_t;

static void seg_seg_closest(v3 a0, v3 a1, v3 b0, v3 b1, float *ps, float *pt) {
    /* closest points a0 + s (a1 - a0), b0 + t (b1 - b0), s, t in [0,1] (Ericson, Real-Time Collision Detection 5.1.9) */
    const float eps = 1e-12f;
    v3 d1 = sub(a1, a0), d2 = sub(b1, b0), r = sub(a0, b0);
    float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r), s, t;
    if (a <= eps && e <= eps) { s = 0.0f; t = 0.0f; }
    else if (a <= eps) { s = 0.0f; t = fminf(fmaxf(f / e, 0.0f), 1.0f); }
    else {
        float c = dot(d1, r);
        if (e <= eps) { t = 0.0f; s = fminf(fmaxf(-c / a, 0.0f), 1.0f); }
        else {
            float b = dot(d1, d2), denom = a * e - b * b;
            s = (denom > eps) ? fminf(fmaxf((b * f - c * e) / denom, 0.0f), 1.0f) : 0.0f;
            t = (b * s + f) / e;
            if (t < 0.0f) { t = 0.0f; s = fminf(fmaxf(-c / a, 0.0f), 1.0f); }
            else if (t > 1.0f) { t = 1.0f; s = fminf(fmaxf((b - c) / a, 0.0f), 1.0f); }
        }
    }
    *ps = s; *pt = t;
}

/* capsules of one point group (base: g < 0, else limb g); positions relative to the base origin, world axes */
static int build_capsules(const lg_robot_model *M, int g, int L, const float (*Rb)[9], const v3 *rb, const v3 *wb, const v3 *vb, capsule_t *out) {
    int np = (g < 0) ? M->num_base_points : M->num_limb_points[g], n = 0;
    for (int i = 0; i < np; ) {
        const lg_point *p0 = (g < 0) ? &M->base_points[i] : &M->limb_points[g][i];
        int b = (g < 0) ? 0 : 1 + g * L + p0->joint, j = i;
        if (i + 1 < np) {
            const lg_point *p1 = (g < 0) ? &M->base_points[i + 1] : &M->limb_points[g][i + 1];
            if (p1->joint == p0->joint && p1->radius == p0->radius && p1->report_body == p0->report_body) j = i + 1;
        }
        const lg_point *p1 = (g < 0) ? &M->base_points[j] : &M->limb_points[g][j];
        capsule_t *c = &out[n++];
        v3 r0 = mv(Rb[b], V(p0->pos[0], p0->pos[1], p0->pos[2])), r1 = mv(Rb[b], V(p1->pos[0], p1->pos[1], p1->pos[2]));
        c->a0 = add(rb[b], r0); c->a1 = add(rb[b], r1);
        c->va0 = add(vb[b], cross(wb[b], r0)); c->va1 = add(vb[b], cross(wb[b], r1));
        c->rad = p0->radius; c->body = b; c->report = p0->report_body;
        c->mhat = 0.0f;                                   /* nominal mass behind the shape */
        if (g < 0) { c->mhat = M->base_mass; for (int d = 0; d < M->num_limbs * L; d++) c->mhat += M->body_mass[d]; }
        else for (int jj = p0->joint; jj < L; jj++) c->mhat += M->body_mass[g * L + jj];
        i = j + 1;
    }
    return n;
}

typedef struct { int on, body, report; float depth, f0, coef, coef_other, frep; v3 n, pa, pb; } self_rec_t;   /* deepest contact of a limb with one partner */

/* one capsule pair: A (on a limb body) against B.  Returns 1 and fills (n: from B to A, depth, f0, contact points) when active. */
static int capsule_contact(const lg_params *P, const capsule_t *A, const capsule_t *B, float kn, v3 *n, float *depth, float *f0, v3 *pa, v3 *pb) {
    float s, t;
    seg_seg_closest(A->a0, A->a1, B->a0, B->a1, &s, &t);
    v3 ca = add(A->a0, scl(sub(A->a1, A->a0), s)), cb = add(B->a0, scl(sub(B->a1, B->a0), t));
    v3 diff = sub(ca, cb);
    float dist = sqrtf(dot(diff, diff));
    float d = A->rad + B->rad - dist;
    if (!(d > -P->contact_margin)) return 0;
    *n = scl(diff, 1.0f / fmaxf(dist, 1e-9f));
    v3 va = add(A->va0, scl(sub(A->va1, A->va0), s)), vbv = add(B->va0, scl(sub(B->va1, B->va0), t));
    float vn = dot(*n, sub(va, vbv));
    float f = P->contact_stiffness * d - kn * vn;
    /* speculative range (not yet touching): only when approaching fast enough to touch within the step.  Overlapping shapes
     * stay coupled even while they separate (s <= 0 -> no bias force, the implicit term alone): a contact that switched off
     * there would alternate on / off from sub-step to sub-step and let the actuators push the links back in on the off beats */
    if (!(d > 0.0f || f > 0.0f)) return 0;
    *depth = d; *f0 = fmaxf(f, 0.0f);
    *pa = sub(ca, scl(*n, A->rad));              /* contact point on A's surface */
    *pb = add(cb, scl(*n, B->rad));              /* ... on B's surface */
    return 1;
}

/* ------------------------------------------------------------------ physics sub-step (stands in for legged_robot.py:92-96) */
#define NB (1 + LG_MAX_DOF)
#define NPTS (LG_MAX_BASE_POINTS + LG_MAX_LIMBS * LG_MAX_LIMB_POINTS)
#define LG_CONTACT_PASSES 2

typedef struct {
    int   body;        /* dynamic body: 0 base, 1+dof */
    int   report;
    v3    r, n, vc;    /* r: rel. base origin (world axes); n: ground normal; vc: point velocity */
    float depth, kn, bt, mu, vtn;   /* vtn = tangential speed at the start of the step */
    int   on;
    v3    f;           /* resulting force (world) */
    v3    fs;          /* constant sliding friction force of the corrector pass (zero while the damper form is used) */
} contact_t;

/* Re-express an articulated inertia / force given about point P at the point Q = P - d (d = P - Q):
 *   I_Q = E^T I_P E,  E = [[1,0],[-[d]x,1]]  ->  H' = H + [d]x M,  A' = A + [d]x H^T - H' [d]x,  M' = M;   n' = n + d x f. */
static void ai_shift(ai6 *I, sv6 *p, v3 d) {
    static const int ix[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    float T[9], A2[6];
    for (int j = 0; j < 3; j++) {
        v3 mc = V(I->M[ix[0][j]], I->M[ix[1][j]], I->M[ix[2][j]]);
        v3 x = cross(d, mc);
        T[j] = I->H[j] + x.x; T[3 + j] = I->H[3 + j] + x.y; T[6 + j] = I->H[6 + j] + x.z;
    }
    v3 dxh[3], txd[3];          /* dxh[j] = d x row_j(H) ; txd[i] = row_i(T) x d */
    for (int j = 0; j < 3; j++) dxh[j] = cross(d, V(I->H[3 * j], I->H[3 * j + 1], I->H[3 * j + 2]));
    for (int i = 0; i < 3; i++) txd[i] = cross(V(T[3 * i], T[3 * i + 1], T[3 * i + 2]), d);
    for (int i = 0; i < 3; i++) for (int j = i; j < 3; j++) {
        float a = (i == 0) ? dxh[j].x : (i == 1) ? dxh[j].y : dxh[j].z;
        float b = (j == 0) ? txd[i].x : (j == 1) ? txd[i].y : txd[i].z;
        A2[ix[i][j]] = I->A[ix[i][j]] + a - b;
    }
    for (int i = 0; i < 6; i++) I->A[i] = A2[i];
    for (int i = 0; i < 9; i++) I->H[i] = T[i];
    p->w = add(p->w, cross(d, p->v));
}

static void physics_substep_env(const lgo_sim *s, int e, const float *tau, int write_contacts) {
    const lg_robot_model *M = &s->R;
    const lg_params *P = &s->P;
    const int K = M->num_limbs, L = M->chain_len, nd = K * L;
    const float dt = P->sim_dt;
    float *root = s->B.root_states + (size_t)e * 13;
    float *dof = s->B.dof_state + (size_t)e * nd * 2;
    const v3 grav = V(P->gravity[0], P->gravity[1], P->gravity[2]);

    /* ---- kinematics.  World axes throughout; every body's spatial quantities are expressed about ITS OWN
     * joint origin O_b (the base about its origin), so lever arms stay of the order of a link length and
     * fp32 does not lose the small distal inertias (DESIGN.md "Conditioning").
     *   rb = O_b relative to the base origin, db = O_b - O_parent, wb = angular velocity,
     *   vb = velocity of the body point at O_b, S = (axis, 0), C = velocity-product acceleration. */
    float Rb[NB][9];
    v3 rb[NB], db[NB], wb[NB], vb[NB], ax[NB];
    sv6 C[NB];
    quat_to_mat(root + 3, Rb[0]);
    rb[0] = V(0, 0, 0); db[0] = V(0, 0, 0); wb[0] = V(root[10], root[11], root[12]); vb[0] = V(root[7], root[8], root[9]);
    for (int k = 0; k < K; k++) for (int j = 0; j < L; j++) {
        int d = k * L + j, b = 1 + d, par = (j == 0) ? 0 : b - 1;
        float q = dof[2 * d], qd = dof[2 * d + 1];
        db[b] = mv(Rb[par], V(M->joint_pos[d][0], M->joint_pos[d][1], M->joint_pos[d][2]));
        rb[b] = add(rb[par], db[b]);
        float R0[9];
        mm(Rb[par], M->joint_rot[d], R0);
        ax[b] = mv(R0, V(M->joint_axis[d][0], M->joint_axis[d][1], M->joint_axis[d][2]));
        float sn = sinf(q), cs = cosf(q);
        for (int c = 0; c < 3; c++) {       /* Rodrigues on each column of R0 about the world axis */
            v3 col = V(R0[c], R0[3 + c], R0[6 + c]);
            v3 rot = add(add(scl(col, cs), scl(cross(ax[b], col), sn)), scl(ax[b], dot(ax[b], col) * (1.0f - cs)));
            Rb[b][c] = rot.x; Rb[b][3 + c] = rot.y; Rb[b][6 + c] = rot.z;
        }
        wb[b] = add(wb[par], scl(ax[b], qd));
        vb[b] = add(vb[par], cross(wb[par], db[b]));
        C[b].w = scl(cross(wb[b], ax[b]), qd);
        C[b].v = scl(cross(vb[b], ax[b]), qd);
    }

    /* ---- rigid-body inertias about O_b and bias forces (gyroscopic - gravity) */
    ai6 I0[NB];
    sv6 p0[NB];
    for (int b = 0; b <= nd; b++) {
        float m; v3 com; float Il[6];
        if (b == 0) {
            float dm = s->B.base_mass_delta ? s->B.base_mass_delta[e] : 0.0f;
            m = M->base_mass + dm;
            float sc = m / M->base_mass;          /* recomputeInertia=True (legged_robot.py:729): scale with mass */
            for (int i = 0; i < 6; i++) Il[i] = M->base_inertia[i] * sc;
            com = V(M->base_com[0], M->base_com[1], M->base_com[2]);
        } else {
            m = M->body_mass[b - 1];
            for (int i = 0; i < 6; i++) Il[i] = M->body_inertia[b - 1][i];
            com = V(M->body_com[b - 1][0], M->body_com[b - 1][1], M->body_com[b - 1][2]);
        }
        v3 c = mv(Rb[b], com);
        float Ilf[9] = {Il[0], Il[1], Il[2], Il[1], Il[3], Il[4], Il[2], Il[4], Il[5]}, T[9], Rt[9], Ic[9];
        mm(Rb[b], Ilf, T);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rt[3 * i + j] = Rb[b][3 * j + i];
        mm(T, Rt, Ic);
        ai_zero(&I0[b]);
        I0[b].A[0] = Ic[0]; I0[b].A[1] = Ic[1]; I0[b].A[2] = Ic[2]; I0[b].A[3] = Ic[4]; I0[b].A[4] = Ic[5]; I0[b].A[5] = Ic[8];
        ai_add_point(&I0[b], m, c);
        v3 l = scl(add(vb[b], cross(wb[b], c)), m);
        v3 n = add(mv(Ic, wb[b]), cross(c, l));
        v3 fg = scl(grav, m);
        p0[b].w = sub(add(cross(wb[b], n), cross(vb[b], l)), cross(c, fg));
        p0[b].v = sub(cross(wb[b], l), fg);
    }

    /* ---- self-collision: per limb and partner (0 = base, m > 0 = limb k ^ m) the deepest active capsule pair */
    self_rec_t srec[LG_MAX_LIMBS][LG_MAX_LIMBS][LG_MAX_CHAIN];     /* [limb][partner][joint of the carrying body] */
    float self_cf[LG_MAX_BODIES][3];
    memset(self_cf, 0, sizeof self_cf);
    for (int k = 0; k < K; k++) for (int m = 0; m < K; m++) for (int j = 0; j < L; j++) srec[k][m][j].on = 0;
    if (P->self_collision) {
        const float kn_s = P->contact_stiffness * dt + P->contact_damping;
        capsule_t cbase[LG_MAX_BASE_POINTS], climb[LG_MAX_LIMBS][LG_MAX_LIMB_POINTS];
        int nbase = build_capsules(M, -1, L, Rb, rb, wb, vb, cbase), nl[LG_MAX_LIMBS];
        for (int k = 0; k < K; k++) nl[k] = build_capsules(M, k, L, Rb, rb, wb, vb, climb[k]);
        for (int k = 0; k < K; k++) for (int m = 0; m < K; m++) {
            if (m > 0 && (k ^ m) >= K) continue;
            const capsule_t *other = (m == 0) ? cbase : climb[k ^ m];
            const int no = (m == 0) ? nbase : nl[k ^ m];
            for (int i = 0; i < nl[k]; i++) for (int j = 0; j < no; j++) {
                const capsule_t *A = &climb[k][i];
                self_rec_t *r = &srec[k][m][A->body - 1 - k * L];
                v3 n, pa, pb; float depth, f0;
                if (!capsule_contact(P, A, &other[j], kn_s, &n, &depth, &f0, &pa, &pb)) continue;
                if (r->on && !(depth > r->depth)) continue;
                const float mA = A->mhat, mB = other[j].mhat;
                r->on = 1; r->body = A->body; r->report = A->report; r->depth = depth; r->f0 = f0; r->n = n; r->pa = pa; r->pb = pb;
                r->coef = dt * kn_s * (1.0f + mA / mB); r->coef_other = dt * kn_s * (1.0f + mB / mA);
                r->frep = f0 / (1.0f + kn_s * dt * (1.0f / mA + 1.0f / mB));
            }
            for (int jb = 0; jb < L; jb++) if (srec[k][m][jb].on) {
                const self_rec_t *r = &srec[k][m][jb];
                self_cf[r->report][0] += r->n.x * r->frep; self_cf[r->report][1] += r->n.y * r->frep; self_cf[r->report][2] += r->n.z * r->frep;
                if (m == 0) { self_cf[0][0] -= r->n.x * r->frep; self_cf[0][1] -= r->n.y * r->frep; self_cf[0][2] -= r->n.z * r->frep; }
            }
        }
    }

    /* ---- contact candidates (r is relative to the carrying body's O_b) */
    contact_t ct[NPTS];
    int nc = 0;
    float mu_env = 0.5f * ((s->B.friction_coeffs ? s->B.friction_coeffs[e] : 1.0f) + P->ground_friction);
    for (int g = -1; g < K; g++) {
        int np = (g < 0) ? M->num_base_points : M->num_limb_points[g];
        for (int i = 0; i < np; i++) {
            const lg_point *pt = (g < 0) ? &M->base_points[i] : &M->limb_points[g][i];
            int b = (g < 0) ? 0 : 1 + g * L + pt->joint;
            contact_t *c = &ct[nc];
            c->body = b; c->report = pt->report_body;
            c->r = mv(Rb[b], V(pt->pos[0], pt->pos[1], pt->pos[2]));
            v3 pw = add(rb[b], c->r);
            ground_contact(s, root[0] + pw.x, root[1] + pw.y, root[2] + pw.z, pt->radius, &c->depth, &c->n);
            c->on = c->depth > -P->contact_margin;
            c->vc = add(vb[b], cross(wb[b], c->r));
            c->kn = P->contact_stiffness * dt + P->contact_damping;
            c->mu = mu_env;
            {   /* implicit regularised Coulomb friction: tangential impedance = secant of mu f_n / |v_t| at the slip speed
                 * the step starts with, capped by the stick impedance; f_n estimated from the previous sub-step's net
                 * contact force on the point's report body (contact_forces is persistent state). */
                float vn0 = dot(c->n, c->vc);
                v3 vt0 = sub(c->vc, scl(c->n, vn0));
                c->vtn = sqrtf(dot(vt0, vt0));
                const float *fp = s->B.contact_forces + ((size_t)e * M->num_bodies + pt->report_body) * 3;
                float fn_est = fmaxf(dot(c->n, V(fp[0], fp[1], fp[2])), 0.0f);
                c->bt = fminf(P->friction_damping, c->mu * fn_est / fmaxf(c->vtn, P->stick_velocity));
            }
            c->f = V(0, 0, 0);
            c->fs = V(0, 0, 0);
            nc++;
        }
    }

    /* ---- articulated-body passes with implicit contact impedances */
    sv6 U[NB], acc[NB];
    float Dinv[NB], uu[NB];
    float vl[NB];                      /* 0, or +-1: joint speed limit active in that direction (set by the previous pass) */
    for (int b = 0; b <= nd; b++) vl[b] = 0.0f;
    for (int pass = 0; pass < LG_CONTACT_PASSES; pass++) {
        ai6 IA[NB]; sv6 pA[NB];
        if (pass == LG_CONTACT_PASSES - 1)          /* self-collision: limb side folded into the bodies' rigid terms for the final pass */
            for (int k = 0; k < K; k++) for (int m = 0; m < K; m++) for (int jb = 0; jb < L; jb++) if (srec[k][m][jb].on) {
                const self_rec_t *r = &srec[k][m][jb];
                v3 arm = sub(r->pa, rb[r->body]), f = scl(r->n, r->f0);       /* arm about the body's own joint origin */
                ai_add_rank1(&I0[r->body], r->coef, cross(arm, r->n), r->n);
                p0[r->body].w = sub(p0[r->body].w, cross(arm, f));
                p0[r->body].v = sub(p0[r->body].v, f);
            }
        for (int b = 0; b <= nd; b++) { IA[b] = I0[b]; pA[b] = p0[b]; }
        ai6 IBp[LG_MAX_LIMBS]; sv6 pBp[LG_MAX_LIMBS];     /* base point i's terms ride with limb i (HIP: lane i owns base point i) */
        for (int k = 0; k < K; k++) { ai_zero(&IBp[k]); pBp[k].w = V(0, 0, 0); pBp[k].v = V(0, 0, 0); }
        for (int i = 0; i < nc; i++) if (ct[i].on) {
            contact_t *c = &ct[i];
            float vn = dot(c->n, c->vc);
            v3 vt = sub(c->vc, scl(c->n, vn));
            v3 f = add(sub(scl(c->n, P->contact_stiffness * c->depth - c->kn * vn), scl(vt, c->bt)), c->fs);
            ai6 *It = (c->body == 0) ? &IBp[i] : &IA[c->body];      /* base points are the first contacts: index i = point i */
            sv6 *pt_ = (c->body == 0) ? &pBp[i] : &pA[c->body];
            ai_add_point(It, dt * c->bt, c->r);
            ai_add_rank1(It, dt * (c->kn - c->bt), cross(c->r, c->n), c->n);
            pt_->w = sub(pt_->w, cross(c->r, f));
            pt_->v = sub(pt_->v, f);
        }
        if (pass == LG_CONTACT_PASSES - 1)
            for (int k = 0; k < K; k++) for (int jb = 0; jb < L; jb++) if (srec[k][0][jb].on) {      /* reactions of limb k's base contacts: force -n f on the base */
                const self_rec_t *r = &srec[k][0][jb];
                v3 cxn = cross(r->pb, r->n);
                ai_add_rank1(&IBp[k], r->coef_other, cxn, r->n);
                pBp[k].w = add(pBp[k].w, scl(cxn, r->f0));
                pBp[k].v = add(pBp[k].v, scl(r->n, r->f0));
            }
        for (int k = 0; k < K; k++) for (int j = L - 1; j >= 0; j--) {
            int d = k * L + j, b = 1 + d, par = (j == 0) ? 0 : b - 1;
            float q = dof[2 * d], qd = dof[2 * d + 1];
            sv6 Sb = {ax[b], V(0, 0, 0)};
            U[b] = ai_mul(&IA[b], Sb);
            float D = dot(ax[b], U[b].w) + M->dof_armature[d] + dt * M->dof_damping[d];
            float u = motor_torque(tau[d], qd, M->dof_vel_limit[d]) - dot(ax[b], pA[b].w) - M->dof_damping[d] * qd;
            if (M->dof_lower[d] <= M->dof_upper[d]) {       /* implicit joint-limit spring-damper */
                /* active when the explicit prediction leaves [lower, upper]:
                 * tau_l = -k (q' - lim) - b qd'  with q' = q + dt qd', qd' = qd + dt qdd */
                float qp = q + dt * qd;
                int lo = qp < M->dof_lower[d], hi = qp > M->dof_upper[d];
                if (lo || hi) {
                    float viol = q - (lo ? M->dof_lower[d] : M->dof_upper[d]);
                    float kl = P->limit_stiffness * dt + P->limit_damping;
                    D += dt * kl;
                    u += -P->limit_stiffness * viol - kl * qd;
                }
            }
            if (vl[b] != 0.0f) {
                /* joint speed limit as an implicit damper towards +-v_lim, 100x the joint's own articulated inertia:
                 * pins qd' to the limit with the reaction carried by the parent (momentum-conserving, unlike a clamp) */
                float Bv = LG_VEL_LIMIT_GAIN * D / dt;
                u += -Bv * (qd - vl[b] * M->dof_vel_limit[d]);
                D += dt * Bv;
            }
            Dinv[b] = 1.0f / D; uu[b] = u;
            /* Ia = IA - U U^T / D ; pa = pA + Ia c + U u / D ; shift to the parent's origin ; accumulate */
            ai6 Ia = IA[b];
            ai_add_rank1(&Ia, -Dinv[b], U[b].w, U[b].v);
            sv6 pa = sadd(sadd(pA[b], ai_mul(&Ia, C[b])), sscl(U[b], u * Dinv[b]));
            ai_shift(&Ia, &pa, db[b]);
            if (par != 0) { ai_add(&IA[par], &Ia); pA[par] = sadd(pA[par], pa); }
            else { IA[b] = Ia; pA[b] = pa; }      /* stash: the base sums limbs pairwise (mirrors the HIP butterfly) */
        }
        {   /* base: (I0 + contacts) + ((l0+l1)+(l2+l3)) */
            ai6 acc_I[LG_MAX_LIMBS]; sv6 acc_p[LG_MAX_LIMBS];
            for (int k = 0; k < K; k++) {
                acc_I[k] = IA[1 + k * L]; acc_p[k] = pA[1 + k * L];
                ai_add(&acc_I[k], &IBp[k]); acc_p[k] = sadd(acc_p[k], pBp[k]);
            }
            for (int stride = 1; stride < K; stride *= 2)
                for (int k = 0; k + stride < K; k += 2 * stride) { ai_add(&acc_I[k], &acc_I[k + stride]); acc_p[k] = sadd(acc_p[k], acc_p[k + stride]); }
            ai_add(&IA[0], &acc_I[0]); pA[0] = sadd(pA[0], acc_p[0]);
        }
        float rhs[6] = {-pA[0].w.x, -pA[0].w.y, -pA[0].w.z, -pA[0].v.x, -pA[0].v.y, -pA[0].v.z}, a0[6];
        if (solve6(&IA[0], rhs, a0) != 0) memset(a0, 0, sizeof a0);
        acc[0].w = V(a0[0], a0[1], a0[2]); acc[0].v = V(a0[3], a0[4], a0[5]);
        for (int k = 0; k < K; k++) for (int j = 0; j < L; j++) {
            int b = 1 + k * L + j, par = (j == 0) ? 0 : b - 1;
            sv6 ap;                                   /* parent's acceleration re-expressed at O_b, plus C */
            ap.w = add(acc[par].w, C[b].w);
            ap.v = add(add(acc[par].v, cross(acc[par].w, db[b])), C[b].v);
            float qdd = (uu[b] - sdot(U[b], ap)) * Dinv[b];
            acc[b].w = add(ap.w, scl(ax[b], qdd));
            acc[b].v = ap.v;
            uu[b] = qdd;                 /* reuse: joint acceleration */
            {
                int d = b - 1;
                float lim = M->dof_vel_limit[d], qn = dof[2 * d + 1] + dt * qdd;
                if (lim > 0.0f && vl[b] == 0.0f && fabsf(qn) > lim) vl[b] = qn > 0.0f ? 1.0f : -1.0f;
            }
        }
        /* evaluate contacts at the end-of-step velocity and (re)classify */
        for (int i = 0; i < nc; i++) if (ct[i].on) {
            contact_t *c = &ct[i];
            v3 vn_ = add(c->vc, scl(add(acc[c->body].v, cross(acc[c->body].w, c->r)), dt));
            float vn = dot(c->n, vn_);
            v3 vt = sub(vn_, scl(c->n, vn));
            float fn = P->contact_stiffness * c->depth - c->kn * vn;
            if (fn <= 0.0f) { c->on = 0; c->f = V(0, 0, 0); continue; }
            c->f = add(sub(scl(c->n, fn), scl(vt, c->bt)), c->fs);               /* force this pass applied */
            /* corrector for the next pass: if the damper form needed more than the cone allows, the point slides -- apply
             * mu f_n against the predicted slip direction as a constant force; otherwise re-aim the secant at the
             * predicted end-of-step slip speed (never weaker than before: sticking points stay on the stick impedance). */
            float vtm = sqrtf(dot(vt, vt)), cone = c->mu * fn;
            if (c->bt * vtm > cone) { c->fs = scl(vt, -cone / vtm); c->bt = 0.0f; }
            else if (c->bt > 0.0f) c->bt = fminf(P->friction_damping, cone / fmaxf(vtm, P->stick_velocity));
        }
    }

    /* ---- semi-implicit Euler */
    for (int d = 0; d < nd; d++) {
        float qd = dof[2 * d + 1] + dt * uu[1 + d];
        float lim = M->dof_vel_limit[d];
        if (lim > 0.0f) qd = fminf(fmaxf(qd, -lim), lim);
        dof[2 * d + 1] = qd;
        dof[2 * d] += dt * qd;
    }
    v3 w0 = wb[0], v0 = vb[0];
    v3 a_lin = add(acc[0].v, cross(w0, v0));          /* classical acceleration of the base origin */
    v3 w1 = add(w0, scl(acc[0].w, dt)), v1 = add(v0, scl(a_lin, dt));
    w1 = clamp_norm(w1, LG_MAX_ANGULAR_VELOCITY); v1 = clamp_norm(v1, LG_MAX_LINEAR_VELOCITY);
    root[7] = v1.x; root[8] = v1.y; root[9] = v1.z; root[10] = w1.x; root[11] = w1.y; root[12] = w1.z;
    root[0] += dt * v1.x; root[1] += dt * v1.y; root[2] += dt * v1.z;
    {   /* q <- normalize(q + dt/2 * (w,0) (x) q) */
        float x = root[3], y = root[4], z = root[5], w = root[6], hx = 0.5f * dt * w1.x, hy = 0.5f * dt * w1.y, hz = 0.5f * dt * w1.z;
        float nx = x + (hx * w + hy * z - hz * y), ny = y + (hy * w + hz * x - hx * z), nz = z + (hz * w + hx * y - hy * x);
        float nw = w - (hx * x + hy * y + hz * z);
        float inv = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz + nw * nw);
        root[3] = nx * inv; root[4] = ny * inv; root[5] = nz * inv; root[6] = nw * inv;
    }
    /* always written: the net GROUND contact forces also seed the next sub-step's friction estimate.  `write_contacts` (the
     * last sub-step of a policy step, or the sub-step entry point) adds the self-collision forces: that sum is what the
     * net-contact-force tensor exports (termination, collision penalty). */
    {
        float *cf = s->B.contact_forces + (size_t)e * M->num_bodies * 3;
        for (int i = 0; i < M->num_bodies * 3; i++) cf[i] = 0.0f;
        for (int i = 0; i < nc; i++) { cf[3 * ct[i].report] += ct[i].f.x; cf[3 * ct[i].report + 1] += ct[i].f.y; cf[3 * ct[i].report + 2] += ct[i].f.z; }
        if (write_contacts && P->self_collision)
            for (int b = 0; b < M->num_bodies; b++) for (int i = 0; i < 3; i++) cf[3 * b + i] += self_cf[b][i];
    }
}

int lgo_physics_substep(lgo_sim *s, const float *torques, int32_t write_contacts, void *stream) {
    (void)stream;
    const int nd = s->R.num_limbs * s->R.chain_len;
#pragma omp parallel for schedule(static) num_threads(s->threads)
    for (int e = 0; e < s->P.num_envs; e++) physics_substep_env(s, e, torques + (size_t)e * nd, write_contacts);
    return 0;
}

/* ------------------------------------------------------------------ torques (legged_robot.py:371-395, anymal.py:71-81) */
static void compute_torques_env(const lgo_sim *s, int e, float *tau) {
    const lg_params *P = &s->P;
    const int nd = s->R.num_limbs * s->R.chain_len, N = P->num_envs;
    const float *act = s->B.actions + (size_t)e * nd;
    const float *dof = s->B.dof_state + (size_t)e * nd * 2;
    for (int d = 0; d < nd; d++) {
        float a = act[d] * P->action_scale, q = dof[2 * d], qd = dof[2 * d + 1], t;
        if (P->control_type == LG_CTRL_ACTUATOR_NET) {
            size_t row = (size_t)e * nd + d, plane = (size_t)N * nd;
            t = actuator_row(s->W, a + P->default_dof_pos[d] - q, qd,
                             s->B.sea_hidden_state + row * 8, s->B.sea_cell_state + row * 8,
                             s->B.sea_hidden_state + (plane + row) * 8, s->B.sea_cell_state + (plane + row) * 8);
        } else {
            if (P->control_type == LG_CTRL_P) t = P->p_gains[d] * (a + P->default_dof_pos[d] - q) - P->d_gains[d] * qd;
            else if (P->control_type == LG_CTRL_V)
                t = P->p_gains[d] * (a - qd) - P->d_gains[d] * (qd - s->B.last_dof_vel[(size_t)e * nd + d]) / P->sim_dt;
            else t = a;
            t = fminf(fmaxf(t, -P->torque_limits[d]), P->torque_limits[d]);
        }
        tau[d] = t;
    }
}

/* ------------------------------------------------------------------ post-physics (legged_robot.py:106-230, 329-444, 831-969) */
static float pairwise(float *v, int K) {            /* (v0+v1)+(v2+v3): the butterfly order */
    for (int stride = 1; stride < K; stride *= 2) for (int k = 0; k + stride < K; k += 2 * stride) v[k] += v[k + stride];
    return v[0];
}
static void resample_commands(const lgo_sim *s, int e, int64_t step, int purpose) {   /* :347-369 */
    const lg_params *P = &s->P;
    float *cmd = s->B.commands + (size_t)e * 4, u[4];
    rand4(s, e, step, purpose, 0, u);
    cmd[0] = urange(P->cmd_lin_vel_x[0], P->cmd_lin_vel_x[1], u[0]);
    cmd[1] = urange(P->cmd_lin_vel_y[0], P->cmd_lin_vel_y[1], u[1]);
    if (P->heading_command) cmd[3] = urange(P->cmd_heading[0], P->cmd_heading[1], u[2]);
    else cmd[2] = urange(P->cmd_ang_vel_yaw[0], P->cmd_ang_vel_yaw[1], u[2]);
    float keep = (sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]) > 0.2f) ? 1.0f : 0.0f;
    cmd[0] *= keep; cmd[1] *= keep;
}
static float wrap_to_pi(float a) {                 /* utils/math.py:45-48 (float32 tensor, python scalars) */
    const float two_pi = 6.2831855f, pi = 3.14159274f;
    a = fmodf(a, two_pi); if (a < 0.0f) a += two_pi;      /* torch remainder: sign of the divisor */
    if (a > pi) a -= two_pi;
    return a;
}
static void get_heights(const lgo_sim *s, int e, float *out) {   /* :831-869 */
    const lg_params *P = &s->P;
    const float *root = s->B.root_states + (size_t)e * 13;
    if (P->terrain_type == LG_TERRAIN_PLANE || !s->B.height_samples) { for (int i = 0; i < P->num_height_points; i++) out[i] = 0.0f; return; }
    float qy[4] = {0, 0, root[5], root[6]};
    float nrm = sqrtf(qy[2] * qy[2] + qy[3] * qy[3]);
    nrm = fmaxf(nrm, 1e-9f);                       /* torch_utils.normalize: x / norm.clamp(min=eps) */
    qy[2] /= nrm; qy[3] /= nrm;
    for (int i = 0; i < P->num_height_points; i++) {
        v3 p = quat_apply(qy, V(P->height_points[i][0], P->height_points[i][1], 0.0f));
        float px = p.x + root[0] + P->hf_border, py = p.y + root[1] + P->hf_border;
        if (!(fabsf(px) < 1e8f)) px = 0.0f;
        if (!(fabsf(py) < 1e8f)) py = 0.0f;
        long ix = (long)(px / P->hf_horizontal_scale), iy = (long)(py / P->hf_horizontal_scale);   /* .long() truncates */
        if (ix < 0) ix = 0; if (ix > P->hf_rows - 2) ix = P->hf_rows - 2;
        if (iy < 0) iy = 0; if (iy > P->hf_cols - 2) iy = P->hf_cols - 2;
        const int16_t *H = s->B.height_samples;
        int16_t h1 = H[ix * P->hf_cols + iy], h2 = H[(ix + 1) * P->hf_cols + iy], h3 = H[ix * P->hf_cols + iy + 1];
        int16_t h = h1 < h2 ? h1 : h2; h = h < h3 ? h : h3;
        out[i] = (float)h * P->hf_vertical_scale;
    }
}

static void reset_env(const lgo_sim *s, int e, int64_t step) {     /* reset_idx :147-191 + anymal.py:56-60 */
    const lg_params *P = &s->P; const lg_robot_model *M = &s->R;
    const int K = M->num_limbs, nd = K * M->chain_len, N = P->num_envs;
    float *root = s->B.root_states + (size_t)e * 13, *dof = s->B.dof_state + (size_t)e * nd * 2;
    float u[4];
    if (P->terrain_curriculum && s->B.terrain_levels) {           /* _update_terrain_curriculum :446-469 */
        float *org = s->B.env_origins + (size_t)e * 3, *cmd = s->B.commands + (size_t)e * 4;
        float dx = root[0] - org[0], dy = root[1] - org[1], dist = sqrtf(dx * dx + dy * dy);
        int up = dist > P->terrain_env_length / 2;
        int down = (dist < sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]) * P->max_episode_length_s * 0.5f) && !up;
        int lvl = s->B.terrain_levels[e] + up - down;
        if (lvl >= P->terrain_num_rows) { rand4(s, e, step, RNG_TERRAIN, 0, u); lvl = (int)(u[0] * P->terrain_num_rows); if (lvl >= P->terrain_num_rows) lvl = P->terrain_num_rows - 1; }
        else if (lvl < 0) lvl = 0;
        s->B.terrain_levels[e] = lvl;
        const float *to = s->B.terrain_origins + ((size_t)lvl * P->terrain_num_cols + s->B.terrain_types[e]) * 3;
        org[0] = to[0]; org[1] = to[1]; org[2] = to[2];
    }
    for (int d = 0; d < nd; d++) {                                /* _reset_dofs :397-412 */
        if ((d & 3) == 0) rand4(s, e, step, RNG_DOF, d >> 2, u);
        dof[2 * d] = P->default_dof_pos[d] * urange(0.5f, 1.5f, u[d & 3]);
        dof[2 * d + 1] = 0.0f;
    }
    const float *org = s->B.env_origins + (size_t)e * 3;          /* _reset_root_states :414-436 */
    for (int i = 0; i < 13; i++) root[i] = P->base_init_state[i];
    root[0] += org[0]; root[1] += org[1]; root[2] += org[2];
    float v[4];
    rand4(s, e, step, RNG_ROOT, 0, u); rand4(s, e, step, RNG_ROOT, 1, v);
    if (P->custom_origins) { root[0] += urange(-1.0f, 1.0f, u[0]); root[1] += urange(-1.0f, 1.0f, u[1]); }
    root[7] = urange(-0.5f, 0.5f, u[2]); root[8] = urange(-0.5f, 0.5f, u[3]);
    root[9] = urange(-0.5f, 0.5f, v[0]); root[10] = urange(-0.5f, 0.5f, v[1]);
    root[11] = urange(-0.5f, 0.5f, v[2]); root[12] = urange(-0.5f, 0.5f, v[3]);
    resample_commands(s, e, step, RNG_CMD_RESET);
    for (int d = 0; d < nd; d++) { s->B.last_actions[(size_t)e * nd + d] = 0.0f; s->B.last_dof_vel[(size_t)e * nd + d] = 0.0f; }
    for (int k = 0; k < K; k++) s->B.feet_air_time[(size_t)e * K + k] = 0.0f;
    s->B.episode_length_buf[e] = 0;
    s->B.reset_buf[e] = 1;
    /* extras["episode"] accumulation (:179-183) happens in finish_extras(), in env order */
    if (s->B.sea_hidden_state) for (int l = 0; l < 2; l++) for (int d = 0; d < nd; d++) for (int k = 0; k < 8; k++) {
        size_t i = (((size_t)l * N + e) * nd + d) * 8 + k;
        s->B.sea_hidden_state[i] = 0.0f; s->B.sea_cell_state[i] = 0.0f;
    }
}

static void compute_observations_env(const lgo_sim *s, int e, int64_t step) {   /* :212-230 + clip :100-101 */
    const lg_params *P = &s->P;
    const int nd = s->R.num_limbs * s->R.chain_len, K = s->R.num_limbs, Lc = s->R.chain_len;
    const float *root = s->B.root_states + (size_t)e * 13, *dof = s->B.dof_state + (size_t)e * nd * 2;
    float *obs = s->B.obs_buf + (size_t)e * P->num_obs;
    const float *blv = s->B.base_lin_vel + (size_t)e * 3, *bav = s->B.base_ang_vel + (size_t)e * 3, *pg = s->B.projected_gravity + (size_t)e * 3;
    const float *cmd = s->B.commands + (size_t)e * 4;
    float nz[48];
    for (int i = 0; i < 3; i++) {
        obs[i] = blv[i] * P->obs_scale_lin_vel; nz[i] = P->noise_lin_vel;
        obs[3 + i] = bav[i] * P->obs_scale_ang_vel; nz[3 + i] = P->noise_ang_vel;
        obs[6 + i] = pg[i]; nz[6 + i] = P->noise_gravity;
        nz[9 + i] = 0.0f;
    }
    obs[9] = cmd[0] * P->obs_scale_lin_vel; obs[10] = cmd[1] * P->obs_scale_lin_vel; obs[11] = cmd[2] * P->obs_scale_ang_vel;
    for (int d = 0; d < nd; d++) {
        obs[12 + d] = (dof[2 * d] - P->default_dof_pos[d]) * P->obs_scale_dof_pos; nz[12 + d] = P->noise_dof_pos;
        obs[24 + d] = dof[2 * d + 1] * P->obs_scale_dof_vel; nz[24 + d] = P->noise_dof_vel;
        obs[36 + d] = s->B.actions[(size_t)e * nd + d]; nz[36 + d] = 0.0f;
    }
    if (P->measure_heights) {
        const float *mh = s->B.measured_heights + (size_t)e * P->num_height_points;
        for (int i = 0; i < P->num_height_points; i++) {
            float h = root[2] - 0.5f - mh[i];
            obs[48 + i] = fminf(fmaxf(h, -1.0f), 1.0f) * P->obs_scale_height;
        }
    }
    if (P->add_noise) {
        /* element (group g of 4, limb k, j<L) of the first 48 draws from block (g*K+k)*2 + j/4, lane j%4 */
        for (int g = 0; g < 4; g++) for (int k = 0; k < K; k++) for (int j = 0; j < Lc; j++) {
            float u[4];
            rand4(s, e, step, RNG_NOISE, (g * K + k) * 2 + (j >> 2), u);
            int i = g * 12 + k * Lc + j;
            obs[i] += (2.0f * u[j & 3] - 1.0f) * nz[i];
        }
        if (P->measure_heights) for (int i = 0; i < P->num_height_points; i++) {
            float u[4];
            rand4(s, e, step, RNG_NOISE_H, i >> 2, u);
            obs[48 + i] += (2.0f * u[i & 3] - 1.0f) * P->noise_height;
        }
    }
    for (int i = 0; i < P->num_obs; i++) obs[i] = fminf(fmaxf(obs[i], -P->clip_observations), P->clip_observations);
}

static void post_physics_env(const lgo_sim *s, int e, int64_t step) {
    const lg_params *P = &s->P; const lg_robot_model *M = &s->R;
    const int K = M->num_limbs, nd = K * M->chain_len, N = P->num_envs, nb = M->num_bodies;
    float *root = s->B.root_states + (size_t)e * 13, *dof = s->B.dof_state + (size_t)e * nd * 2;
    const float *cf = s->B.contact_forces + (size_t)e * nb * 3;
    float *cmd = s->B.commands + (size_t)e * 4;
    const float *act = s->B.actions + (size_t)e * nd, *tq = s->B.torques + (size_t)e * nd;
    float *lact = s->B.last_actions + (size_t)e * nd, *ldv = s->B.last_dof_vel + (size_t)e * nd;

    s->B.episode_length_buf[e] += 1;                                               /* :114 */
    v3 blv = quat_rotate_inverse(root + 3, V(root[7], root[8], root[9]));          /* :118-121 */
    v3 bav = quat_rotate_inverse(root + 3, V(root[10], root[11], root[12]));
    v3 pg = quat_rotate_inverse(root + 3, V(0, 0, -1));
    float *o;
    o = s->B.base_lin_vel + (size_t)e * 3; o[0] = blv.x; o[1] = blv.y; o[2] = blv.z;
    o = s->B.base_ang_vel + (size_t)e * 3; o[0] = bav.x; o[1] = bav.y; o[2] = bav.z;
    o = s->B.projected_gravity + (size_t)e * 3; o[0] = pg.x; o[1] = pg.y; o[2] = pg.z;

    /* _post_physics_step_callback :329-345 */
    if (s->B.episode_length_buf[e] % P->resample_interval == 0) resample_commands(s, e, step, RNG_CMD_STEP);
    if (P->heading_command) {
        v3 fwd = quat_apply(root + 3, V(1, 0, 0));
        float heading = atan2f(fwd.y, fwd.x);
        cmd[2] = fminf(fmaxf(0.5f * wrap_to_pi(cmd[3] - heading), -1.0f), 1.0f);
    }
    if (P->measure_heights) get_heights(s, e, s->B.measured_heights + (size_t)e * P->num_height_points);
    if (P->push_interval > 0 && step % P->push_interval == 0) {                    /* _push_robots :438-444 */
        float u[4]; rand4(s, e, step, RNG_PUSH, 0, u);
        root[7] = urange(-P->max_push_vel, P->max_push_vel, u[0]);
        root[8] = urange(-P->max_push_vel, P->max_push_vel, u[1]);
    }

    /* check_termination :139-145 */
    int contact_term = 0;
    for (int b = 0; b < nb; b++) if (M->termination_mask >> b & 1u) {
        float n = sqrtf(cf[3 * b] * cf[3 * b] + cf[3 * b + 1] * cf[3 * b + 1] + cf[3 * b + 2] * cf[3 * b + 2]);
        if (n > 1.0f) contact_term = 1;
    }
    int time_out = s->B.episode_length_buf[e] > P->max_episode_length;
    int finite = 1;                                  /* safety net: a non-finite state ends the episode (PhysX never emits NaN) */
    for (int i = 0; i < 13; i++) finite &= isfinite(root[i]) ? 1 : 0;
    for (int d = 0; d < 2 * nd; d++) finite &= isfinite(dof[d]) ? 1 : 0;
    int reset = contact_term || time_out || !finite;
    s->B.time_out_buf[e] = (uint8_t)time_out;
    s->B.reset_buf[e] = (uint8_t)reset;

    /* compute_reward :193-210, terms :872-969 + cassie.py:43-46 */
    float term[LG_NUM_REWARD_TERMS];
    const float *sc = P->reward_scale;
    float cmd_xy = sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]);
    float mh_mean = 0.0f;   /* base_height uses root z - measured heights (0 on the plane / when not measuring, :562) */
    if (P->measure_heights) {
        const float *mh = s->B.measured_heights + (size_t)e * P->num_height_points;
        float acc = 0.0f; for (int i = 0; i < P->num_height_points; i++) acc += root[2] - mh[i];
        mh_mean = acc / (float)P->num_height_points;
    } else mh_mean = root[2];
    /* sums over joints / bodies / feet: partial per limb, then pairwise (l0+l1)+(l2+l3).  torch.sum's own order is
     * unspecified; this one is what the limb-per-lane HIP kernel's butterfly produces, so parity can be tight. */
    float pv[9][LG_MAX_LIMBS];
    memset(pv, 0, sizeof pv);
    const int Lc = M->chain_len;
    for (int k = 0; k < K; k++) for (int j = 0; j < Lc; j++) {
        int d = k * Lc + j;
        float q = dof[2 * d], qd = dof[2 * d + 1];
        float da = lact[d] - act[d]; pv[0][k] += da * da;
        float dd = (ldv[d] - qd) / P->dt_policy; pv[1][k] += dd * dd;
        float ol = -fminf(q - P->soft_pos_lower[d], 0.0f); ol += fmaxf(q - P->soft_pos_upper[d], 0.0f); pv[2][k] += ol;
        pv[3][k] += qd * qd;
        pv[4][k] += fminf(fmaxf(fabsf(qd) - P->dof_vel_limits[d] * P->soft_dof_vel_limit, 0.0f), 1.0f);
        pv[5][k] += fmaxf(fabsf(tq[d]) - P->torque_limits[d] * P->soft_torque_limit, 0.0f);
        pv[6][k] += tq[d] * tq[d];
        pv[7][k] += fabsf(q - P->default_dof_pos[d]);
    }
    for (int b = 1; b < nb; b++) if (M->penalised_mask >> b & 1u) {     /* limb bodies, grouped by owning limb */
        float n = sqrtf(cf[3 * b] * cf[3 * b] + cf[3 * b + 1] * cf[3 * b + 1] + cf[3 * b + 2] * cf[3 * b + 2]);
        pv[8][(b - 1) / ((nb - 1) / K)] += (n > 0.1f) ? 1.0f : 0.0f;
    }
    float s_ar = pairwise(pv[0], K), s_acc = pairwise(pv[1], K), s_lim = pairwise(pv[2], K), s_dv = pairwise(pv[3], K);
    float s_dvl = pairwise(pv[4], K), s_tl = pairwise(pv[5], K), s_tq = pairwise(pv[6], K), s_ss = pairwise(pv[7], K);
    float coll = pairwise(pv[8], K);
    if (M->penalised_mask & 1u) {
        float n = sqrtf(cf[0] * cf[0] + cf[1] * cf[1] + cf[2] * cf[2]);
        coll += (n > 0.1f) ? 1.0f : 0.0f;
    }
    float air = 0.0f, fcf = 0.0f; int stumble = 0;
    float fv[3][LG_MAX_LIMBS];
    memset(fv, 0, sizeof fv);
    for (int k = 0; k < K; k++) {
        const float *f = cf + 3 * M->foot_body[k];
        float fn = sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
        fv[0][k] = fmaxf(fn - P->max_contact_force, 0.0f);
        if (sqrtf(f[0] * f[0] + f[1] * f[1]) > 5.0f * fabsf(f[2])) stumble = 1;
        fv[1][k] = f[2] > 0.1f ? 1.0f : 0.0f;
    }
    fcf = pairwise(fv[0], K);
    float nfly = pairwise(fv[1], K);
    if (sc[LG_REW_FEET_AIR_TIME] != 0.0f) {        /* stateful; only runs when the term is registered (:583-602) */
        for (int k = 0; k < K; k++) {
            const float *f = cf + 3 * M->foot_body[k];
            float *at = s->B.feet_air_time + (size_t)e * K + k; uint8_t *lc = s->B.last_contacts + (size_t)e * K + k;
            int contact = f[2] > 1.0f, filt = contact || *lc;
            *lc = (uint8_t)contact;
            int first = (*at > 0.0f) && filt;
            *at += P->dt_policy;
            fv[2][k] = (*at - 0.5f) * (first ? 1.0f : 0.0f);
            *at *= filt ? 0.0f : 1.0f;
        }
        air = pairwise(fv[2], K) * ((cmd_xy > 0.1f) ? 1.0f : 0.0f);
    }
    term[LG_REW_ACTION_RATE] = s_ar;
    term[LG_REW_ANG_VEL_XY] = bav.x * bav.x + bav.y * bav.y;
    term[LG_REW_BASE_HEIGHT] = (mh_mean - P->base_height_target) * (mh_mean - P->base_height_target);
    term[LG_REW_COLLISION] = coll;
    term[LG_REW_DOF_ACC] = s_acc;
    term[LG_REW_DOF_POS_LIMITS] = s_lim;
    term[LG_REW_DOF_VEL] = s_dv;
    term[LG_REW_DOF_VEL_LIMITS] = s_dvl;
    term[LG_REW_FEET_AIR_TIME] = air;
    term[LG_REW_FEET_CONTACT_FORCES] = fcf;
    term[LG_REW_LIN_VEL_Z] = blv.z * blv.z;
    term[LG_REW_NO_FLY] = (nfly == 1.0f) ? 1.0f : 0.0f;
    term[LG_REW_ORIENTATION] = pg.x * pg.x + pg.y * pg.y;
    term[LG_REW_STAND_STILL] = s_ss * ((cmd_xy < 0.1f) ? 1.0f : 0.0f);
    term[LG_REW_STUMBLE] = stumble ? 1.0f : 0.0f;
    term[LG_REW_TERMINATION] = (reset && !time_out) ? 1.0f : 0.0f;
    term[LG_REW_TORQUE_LIMITS] = s_tl;
    term[LG_REW_TORQUES] = s_tq;
    {
        float ex = cmd[0] - blv.x, ey = cmd[1] - blv.y, ew = cmd[2] - bav.z;
        term[LG_REW_TRACKING_LIN_VEL] = expf(-(ex * ex + ey * ey) / P->tracking_sigma);
        term[LG_REW_TRACKING_ANG_VEL] = expf(-(ew * ew) / P->tracking_sigma);
    }
    float rew = 0.0f;
    for (int t = 0; t < LG_NUM_REWARD_TERMS; t++) {
        if (t == LG_REW_TERMINATION || P->reward_slot[t] < 0) continue;
        float r = term[t] * sc[t];
        rew += r;
        s->B.episode_sums[(size_t)P->reward_slot[t] * N + e] += r;
    }
    if (P->only_positive_rewards) rew = fmaxf(rew, 0.0f);
    if (P->reward_slot[LG_REW_TERMINATION] >= 0) {
        float r = term[LG_REW_TERMINATION] * sc[LG_REW_TERMINATION];
        rew += r;
        s->B.episode_sums[(size_t)P->reward_slot[LG_REW_TERMINATION] * N + e] += r;
    }
    s->B.rew_buf[e] = rew;

    if (reset) reset_env(s, e, step);                                              /* :128-129 */
    compute_observations_env(s, e, step);                                          /* :130 */
    for (int d = 0; d < nd; d++) { lact[d] = act[d]; ldv[d] = dof[2 * d + 1]; }     /* :132-133 */
    for (int i = 0; i < 6; i++) s->B.last_root_vel[(size_t)e * 6 + i] = root[7 + i]; /* :134 */
}

/* extras["episode"] (:179-188): mean over the envs whose reset_buf is set of episode_sums / max_episode_length_s,
 * then zero those sums.  Kept stale when nothing reset (quirk Q4). */
static void finish_extras(const lgo_sim *s, const uint8_t *mask) {
    const lg_params *P = &s->P; const int N = P->num_envs, R = P->num_reward_slots;
    int cnt = 0;
    for (int e = 0; e < N; e++) cnt += mask[e] ? 1 : 0;
    if (cnt > 0) for (int t = 0; t < R; t++) {
        float acc = 0.0f;
        for (int e = 0; e < N; e++) if (mask[e]) { acc += s->B.episode_sums[(size_t)t * N + e]; s->B.episode_sums[(size_t)t * N + e] = 0.0f; }
        s->B.episode_means[t] = acc / (float)cnt / P->max_episode_length_s;
    }
    if (P->terrain_curriculum && s->B.terrain_levels) {
        float acc = 0.0f;
        for (int e = 0; e < N; e++) acc += (float)s->B.terrain_levels[e];
        s->B.episode_means[R] = acc / (float)N;
    }
}

/* ------------------------------------------------------------------ C-ABI */
int lgo_create(const lg_params *params, const lg_robot_model *model, const float *actuator_weights, int device_id, lgo_sim **out) {
    (void)device_id;
    if (!params || !model || !out) { snprintf(g_err, sizeof g_err, "null argument"); return -1; }
    if (params->abi_version != LG_ABI_VERSION) { snprintf(g_err, sizeof g_err, "ABI version %d != %d", params->abi_version, LG_ABI_VERSION); return -3; }
    if (model->num_limbs * model->chain_len != LG_MAX_DOF || model->num_limbs > LG_MAX_LIMBS || model->chain_len > LG_MAX_CHAIN) {
        snprintf(g_err, sizeof g_err, "model must be K limbs x L joints with K*L == 12"); return -4;
    }
    lgo_sim *s = (lgo_sim *)calloc(1, sizeof *s);
    s->P = *params; s->R = *model;
    if (actuator_weights) { memcpy(s->W, actuator_weights, sizeof s->W); s->has_net = 1; }
    if (params->control_type == LG_CTRL_ACTUATOR_NET && !s->has_net) { free(s); snprintf(g_err, sizeof g_err, "actuator net control without weights"); return -2; }
    const char *t = getenv("LGO_THREADS");
    s->threads = t ? atoi(t) : 1;
    if (s->threads < 1) s->threads = 1;
    *out = s;
    return 0;
}
void lgo_destroy(lgo_sim *s) { free(s); }
int lgo_bind(lgo_sim *s, const lg_buffers *b) { s->B = *b; return 0; }
int lgo_set_params(lgo_sim *s, const lg_params *p) { s->P = *p; return 0; }
int lgo_set_obs_buffer(lgo_sim *s, float *obs) { if (!s || !obs) return -1; s->B.obs_buf = obs; return 0; }
int lgo_set_threads(lgo_sim *s, int n) { s->threads = n < 1 ? 1 : n; return 0; }

int lgo_step(lgo_sim *s, const float *actions, int64_t step, void *stream) {
    (void)stream;
    if (step < 0) step = s->B.step_counter[0] + 1;
    if (s->B.step_counter) s->B.step_counter[0] = step;
    const int nd = s->R.num_limbs * s->R.chain_len;
#pragma omp parallel for schedule(static) num_threads(s->threads)
    for (int e = 0; e < s->P.num_envs; e++) {
        float *act = s->B.actions + (size_t)e * nd, *tq = s->B.torques + (size_t)e * nd;
        for (int d = 0; d < nd; d++) act[d] = fminf(fmaxf(actions[(size_t)e * nd + d], -s->P.clip_actions), s->P.clip_actions);  /* :86-87 */
        for (int it = 0; it < s->P.decimation; it++) {                                                                          /* :90-96 */
            compute_torques_env(s, e, tq);
            physics_substep_env(s, e, tq, it == s->P.decimation - 1);
        }
        post_physics_env(s, e, step);                                                                                           /* :97 */
    }
    finish_extras(s, s->B.reset_buf);
    return 0;
}

int lgo_reset_idx(lgo_sim *s, const int32_t *env_ids, int32_t count, int64_t step, void *stream) {
    (void)stream;
    uint8_t *mask = (uint8_t *)calloc((size_t)s->P.num_envs, 1);
    for (int i = 0; i < count; i++) { reset_env(s, env_ids[i], step); mask[env_ids[i]] = 1; }
    finish_extras(s, mask);
    free(mask);
    return 0;
}
int lgo_compute_observations_only(lgo_sim *s, int64_t step, void *stream) {
    (void)stream;
    for (int e = 0; e < s->P.num_envs; e++) compute_observations_env(s, e, step);
    return 0;
}
