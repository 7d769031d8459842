// lg_kernels.hip -- fused LeggedRobot.step() for gfx950 + the C-ABI of include/legged_hip.h.
//
// One launch per policy step does what reference legged_gym/envs/base/legged_robot.py:80-137
// does with ~150 eager torch kernels and 4 PhysX steps:
//   clip actions -> decimation x [actuator net / PD torque -> articulated-body step with
//   implicit contacts] -> post_physics_step (base-frame quantities, command resampling,
//   height sampling, pushes, termination, the _reward_* sum, predicated reset_idx,
//   compute_observations with noise, last_* bookkeeping) -> clip observations.
// Persistent state is read once and written once per env-step.  A workgroup is 64 (env, limb) lanes x up to four waves:
// the rigid-body wave plus helper waves (actuator LSTMs, body inertia terms, height sampling, bookkeeping) -- see k_step.
//
// Build: __graft_entry__.build() (hipcc --offload-arch=gfx950 -O3 -shared -fPIC + the flags explained in DESIGN.md section 5)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <new>

#include "lg_device.h"
#include "lg_policy.h"
#include "lg_train.h"
#include "lg_gemm.h"

using namespace lg;

#define LG_BLOCK 64            // (env, limb) lanes per workgroup = one wave; 4096 envs x 4 limbs = 256 workgroups = one per CU
#define LG_JS 40               // floats per joint in the LDS limb table
#define LG_PASSES 2
#define LG_STEP_WAVES 4          // waves per workgroup of the fused step: 1 rigid-body wave + 3 helpers

// ------------------------------------------------------------------ robot topologies compiled in
struct AnymalTraits {          // quadruped (ANYmal-C/B, A1): base + 4 legs x 3 joints; points: 2 thigh, 2 shank, foot sphere
    static constexpr int K = 4, L = 3, NREP = 4, NPT = 5;
    static constexpr int pt_joint(int i) { return i < 2 ? 1 : 2; }
    static constexpr int pt_rep(int i) { return i < 2 ? 1 : (i < 4 ? 2 : 3); }
    static constexpr int FOOT_REP = 3;
    // self-collision shapes: capsule c = segment between points cap_p0(c), cap_p1(c) (a sphere when equal): KFE-drive capsule on the
    // thigh, shank capsule, foot sphere; bounding groups (culling): g0 = drive (joint 1), g1 = shank + foot (joint 2)
    static constexpr int NCAP = 3, NGRP = 2;
    static constexpr int cap_p0(int c) { return 2 * c; }
    static constexpr int cap_p1(int c) { return c < 2 ? 2 * c + 1 : 4; }
    static constexpr int cap_grp(int c) { return c == 0 ? 0 : 1; }
    static constexpr int grp_c0(int g) { return g == 0 ? 0 : 2; }       // group centre = midpoint of these two points
    static constexpr int grp_c1(int g) { return g == 0 ? 1 : 4; }
    static constexpr int grp_cap_lo(int g) { return g == 0 ? 0 : 1; }   // capsules [lo, hi) of the group (= of one body)
    static constexpr int grp_cap_hi(int g) { return g == 0 ? 1 : 3; }
};
struct CassieTraits {          // pelvis + 2 legs x 6 joints; points: toe capsule
    static constexpr int K = 2, L = 6, NREP = 6, NPT = 2;
    static constexpr int pt_joint(int) { return 5; }
    static constexpr int pt_rep(int) { return 5; }
    static constexpr int FOOT_REP = 5;
    static constexpr int NCAP = 1, NGRP = 1;                            // toe capsule
    static constexpr int cap_p0(int) { return 0; }
    static constexpr int cap_p1(int) { return 1; }
    static constexpr int cap_grp(int) { return 0; }
    static constexpr int grp_c0(int) { return 0; }
    static constexpr int grp_c1(int) { return 1; }
    static constexpr int grp_cap_lo(int) { return 0; }
    static constexpr int grp_cap_hi(int) { return 1; }
};
template <class T> struct Tab { static constexpr int GRP = T::L * LG_JS + 4 * T::NPT;                 // bounding radius of each point group
                          static constexpr int BPT = GRP + T::NGRP + 1;                           // base collision point owned by the lane of this limb (x, y, z, radius)
                          static constexpr int STRIDE = BPT + 4; };
// per-joint offsets inside the limb table
enum { J_POS = 0, J_ROT = 3, J_AXIS = 12, J_MASS = 15, J_COM = 16, J_INERTIA = 19, J_LO = 25, J_HI = 26, J_VLIM = 27,
       J_ARM = 28, J_DAMP = 29, J_KP = 30, J_KD = 31, J_Q0 = 32, J_TLIM = 33, J_SLO = 34, J_SHI = 35, J_DVL = 36,
       J_SUBM = 37 /* mass of the limb from this joint outwards (self-collision weights) */ };

struct BaseTab { float mass, com[3], inertia[6]; float pts[LG_MAX_BASE_POINTS][4]; int32_t num_pts; float mass_robot; };   // num_pts <= K; mass_robot: nominal total mass

// lg_rollout_policy: k_step<..., ROLL> runs `steps` consecutive policy steps in ONE launch (each workgroup walks through the steps of its
// own 16 envs without waiting for the other 255 at every step boundary); per-step outputs go to [t]-indexed rollout storage.
struct RollArgs {
    int      steps;
    float   *obs;              // [steps + 1][N][num_obs]: obs[0] is the input of step 0, step t writes obs[t + 1]
    const float *obs0;         // if set: the input of step 0 lives here (e.g. obs[steps] of the previous segment); every workgroup copies its rows to obs[0]
    float   *actions, *mean;   // [steps][N][num_actions] (mean may be null)
    float   *rew;              // [steps][N]
    uint8_t *done, *time_outs; // [steps][N]
    float   *extras;           // [steps][LG_NUM_REWARD_TERMS + 2] episode accumulators per step (zero at launch; roll_finish publishes and re-zeroes)
};

struct KArgs {                 // passed by value: lives in the kernarg segment -> scalar loads
    lg_params  P;
    lg_buffers B;
    BaseTab    base;
    const float *limb_table;   // [K][STRIDE] device
    const float *hpts;         // [LG_MAX_HEIGHT_POINTS][2] device copy of P.height_points: a lane-dependent index into the
                               // by-value kernarg struct makes the compiler copy all of KArgs to scratch (3 KB per lane)
    const float *weights;      // [LW_ROWS][64] per-lane MFMA operand table of the actuator net, or null
    const float *actions_in;   // [N, ndof]
    const int32_t *env_ids;    // reset kernel only
    int32_t    count;
    uint32_t   penalised_mask, termination_mask;
    unsigned int *done_counter;   // workgroup ticket of k_step (zeroed at create, self-resetting); behind it 2 x gridDim.x terrain-level partial sums
    float *accum_alt;             // deferred extras (lg_set_deferred_extras): the episode accumulators of ODD steps (even ones use B.extras_accum)
    int   flush_parts;            // k_extras as lg_extras_flush: number of level partial sums to add (0: scan terrain_levels)
    int   defer;                  // 1: no finisher in this launch -- workgroup 0 turns the PREVIOUS step's accumulators into episode_means instead
    int64_t    step;
    PolicyArgs pol;               // fused rollout step (k_step<..., POL = true>): the actor that produces this step's actions
    unsigned long long *prof;     // LG_PROFILE builds only: [LG_NPROF] cycle accumulators (tools/profile_sections.py)
    unsigned int *status;         // sticky device status word (host-mapped): LG_STATUS_* bits, see lg_device_status()
    int   spin_limit;             // bound of the LDS hand-over polls (s_sleep rounds); lg_debug_handover() shrinks it
    int   debug_skip;             // test hook: 1 = the rigid-body wave withholds the frame hand-over flag, 2 = the helpers withhold the self-collision flags
    RollArgs roll;                // k_step<..., ROLL> only
};

// A hand-over poll that ran out must not pass silently (rc 0 with wrong physics is the worst failure this library can have): the
// wave ORs a bit into the host-visible status word and goes on (a hung CU would be worse); every later C-ABI call on the handle
// fails with that status until lg_clear_device_status().
LG_DEV void lg_report(unsigned int *status, unsigned int bit) {
    if (status) __hip_atomic_fetch_or(status, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------ section profiler (debug builds: -DLG_PROFILE)
#define LG_NPROF 20
#define LG_NPROF_BLOCKS 1024
#define LG_NPROF_ROLL 33             // rollout kernel (profile builds): wall-clock stamps [step boundary 0 .. 32][workgroup]
#define LG_NPROF_TOTAL (LG_NPROF + LG_NPROF_BLOCKS * 40 + LG_NPROF_ROLL * LG_NPROF_BLOCKS)
#ifdef LG_PROFILE
// lane 0 of each workgroup accumulates s_memtime deltas per section in LDS and adds them to A.prof at the end.
// idx < 0 starts the clock; slot 14 = whole kernel (s_memtime), slot 15 = whole kernel on the 100 MHz wall clock.
__device__ __forceinline__ void lg_prof(int idx, unsigned long long *out, unsigned long long note = 0) {
    __shared__ unsigned long long acc[LG_NPROF], prev, t0, w0;
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        if (idx < 0) { for (int i = 0; i < LG_NPROF; i++) acc[i] = 0; t0 = t; w0 = wall_clock64(); }
        else if (idx >= 18) { if (idx == 18) acc[18] = note; else acc[19] += note; t = prev; }   // user slots: 18 a value, 19 a sum of packed event counts (LG_PROF_COUNT); the clock is not advanced
        else acc[idx] += t - prev;
        if (out) {
            acc[14] = t - t0; acc[15] = wall_clock64() - w0;
            // per workgroup, plain stores to its own record (atomics on shared accumulators stall the workgroups still running):
            // [0] running sums over the launches since the last reset, [1] the LAST launch; slots 16/17 = start/end on the wall clock
            if (blockIdx.x < LG_NPROF_BLOCKS) {
                unsigned long long *a = out + LG_NPROF + (size_t)blockIdx.x * 40, *b = a + 20;
                for (int i = 0; i < 16; i++) { a[i] += acc[i]; b[i] = acc[i]; }
                a[16] = a[16] > acc[14] ? a[16] : acc[14];       // slowest launch of this workgroup
                b[16] = w0; b[17] = w0 + acc[15]; b[18] = acc[18]; b[19] = acc[19];
            }
        }
        prev = idx >= 18 ? prev : __builtin_readcyclecounter();
    }
    __builtin_amdgcn_sched_barrier(0);
}
#ifdef LG_PROFILE_LIGHT       // only each workgroup's start / end on the wall clock: the spread of the PRODUCT kernel's workgroups (no section stamps, no census)
#define LG_PROF(i)
#else
#define LG_PROF(i) lg_prof(i, nullptr)
#endif
#define LG_PROF_NOTE(i, v) lg_prof(i, nullptr, v)
#define LG_PROF_BEGIN() lg_prof(-1, nullptr)
#define LG_PROF_END(i, out) lg_prof(i, out)
// rare-path census of the rigid-body wave (converged code only): field f of slot 19 counts the times ANY lane took the path, field 6 (16 bit) the lanes
#ifdef LG_PROFILE_LIGHT      // the two events of a fallen robot only: trunk contact (1) and self-collision (2) passes
#define LG_PROF_COUNT(f, cond) do { if ((f) == 1 || (f) == 2) { if (__ballot(cond)) lg_prof(19, nullptr, 1ull << (8 * (f))); } } while (0)
#else
#define LG_PROF_COUNT(f, cond) do { const unsigned long long b_ = __ballot(cond); if (b_) lg_prof(19, nullptr, (1ull << (8 * (f))) + ((f) == 0 ? (unsigned long long)__popcll(b_) << 48 : 0ull)); } while (0)
#endif
#else
#define LG_PROF(i)
#define LG_PROF_BEGIN()
#define LG_PROF_END(i, out)
#define LG_PROF_NOTE(i, v)
#define LG_PROF_COUNT(f, cond)
#endif
enum { PF_PROLOGUE = 0, PF_TORQUE, PF_KINEMATICS, PF_INWARD, PF_BASE, PF_OUTWARD, PF_INTEGRATE, PF_POST, PF_EXTRAS,
       PF_POST_HEIGHTS, PF_POST_TERMS, PF_POST_REWARD, PF_POST_RESET, PF_POST_OBS };

// ------------------------------------------------------------------ terrain
// Split in two so the four samples of EVERY collision point of a sub-step are in flight before the first is consumed
// (the table is L2-resident but ~1 us away for a lone wave): hf_fetch issues the loads, hf_contact evaluates the patch.
struct HfFetch { int16_t s00, s10, s01, s11; float tx, ty; };
template <bool HF> LG_DEV HfFetch hf_fetch(const KArgs &A, float x, float y) {
    HfFetch f; f.s00 = f.s10 = f.s01 = f.s11 = 0; f.tx = f.ty = 0.0f;
    if (!HF) return f;
    const lg_params &P = A.P;
    float inv = 1.0f / P.hf_horizontal_scale;
    float gx = (x + P.hf_border) * inv, gy = (y + P.hf_border) * inv;
    if (!(fabsf(gx) < 1e9f)) gx = 0.0f;                 // non-finite / absurd position (the env is reset at the end of the step): no float -> int UB
    if (!(fabsf(gy) < 1e9f)) gy = 0.0f;
    float fx = floorf(gx), fy = floorf(gy);
    int ix = (int)fx, iy = (int)fy;
    f.tx = gx - fx; f.ty = gy - fy;
    auto at = [&](int i, int j) {
        i = min(max(i, 0), P.hf_rows - 1); j = min(max(j, 0), P.hf_cols - 1);
        return A.B.height_samples[(size_t)i * P.hf_cols + j];
    };
    f.s00 = at(ix, iy); f.s10 = at(ix + 1, iy); f.s01 = at(ix, iy + 1); f.s11 = at(ix + 1, iy + 1);
    return f;
}
// Ground contact of a collision sphere (centre height pz, radius) from its fetched samples: depth and unit normal.  Twin of the
// oracle's ground_contact(): bilinear patch; with hf_step_threshold > 0 ('trimesh') a height difference across the cell beyond
// the threshold is a vertical face at the high side (horizontal normal) and the low level extends up to it.
template <bool HF> LG_DEV void hf_contact(const KArgs &A, const HfFetch &f, float pz, float radius, float &depth, V3 &n) {
    if (!HF) { n = v3(0, 0, 1); depth = radius - pz; return; }
    const lg_params &P = A.P;
    float inv = 1.0f / P.hf_horizontal_scale;
    float h00 = (float)f.s00 * P.hf_vertical_scale, h10 = (float)f.s10 * P.hf_vertical_scale;
    float h01 = (float)f.s01 * P.hf_vertical_scale, h11 = (float)f.s11 * P.hf_vertical_scale;
    const float thr = P.hf_step_threshold;
    float wall_depth = -1e30f; V3 wall_n = v3(0, 0, 1);
    if (thr > 0.0f) {                                     // wave-uniform
        const float hs = P.hf_horizontal_scale;
        if (fmaxf(fabsf(h10 - h00), fabsf(h11 - h01)) > thr) {
            const bool up = (h10 + h11) > (h00 + h01);
            const float top = up ? h10 + (h11 - h10) * f.ty : h00 + (h01 - h00) * f.ty;
            const float dist = (up ? 1.0f - f.tx : f.tx) * hs;
            if (pz - radius < top) {
                float d = radius - dist; V3 fn = v3(up ? -1.0f : 1.0f, 0, 0);
                if (pz > top) {              // centre above the riser's top edge: edge contact (oracle comment)
                    const float dz = pz - top, len = sqrtf(dist * dist + dz * dz), il = 1.0f / fmaxf(len, 1e-9f);
                    d = radius - len; fn = v3((up ? -dist : dist) * il, 0, dz * il);
                }
                if (d > wall_depth) { wall_depth = d; wall_n = fn; }
            }
            if (up) { h10 = h00; h11 = h01; } else { h00 = h10; h01 = h11; }
        }
        if (fmaxf(fabsf(h01 - h00), fabsf(h11 - h10)) > thr) {
            const bool up = (h01 + h11) > (h00 + h10);
            const float top = up ? h01 + (h11 - h01) * f.tx : h00 + (h10 - h00) * f.tx;
            const float dist = (up ? 1.0f - f.ty : f.ty) * hs;
            if (pz - radius < top) {
                float d = radius - dist; V3 fn = v3(0, up ? -1.0f : 1.0f, 0);
                if (pz > top) {
                    const float dz = pz - top, len = sqrtf(dist * dist + dz * dz), il = 1.0f / fmaxf(len, 1e-9f);
                    d = radius - len; fn = v3(0, (up ? -dist : dist) * il, dz * il);
                }
                if (d > wall_depth) { wall_depth = d; wall_n = fn; }
            }
            if (up) { h01 = h00; h11 = h10; } else { h00 = h01; h10 = h11; }
        }
    }
    float hx0 = h00 + (h10 - h00) * f.tx, hx1 = h01 + (h11 - h01) * f.tx;
    float h = hx0 + (hx1 - hx0) * f.ty;
    float dhdx = ((h10 - h00) + ((h11 - h01) - (h10 - h00)) * f.ty) * inv;
    float dhdy = (hx1 - hx0) * inv;
    float l = 1.0f / sqrtf(dhdx * dhdx + dhdy * dhdy + 1.0f);
    n = v3(-dhdx * l, -dhdy * l, l);
    depth = radius - (pz - h) * n.z;
    if (wall_depth > depth) { depth = wall_depth; n = wall_n; }
}
// ------------------------------------------------------------------ one 5 ms rigid-body step for (env, limb)
struct Contact { V3 r, n, vc, f, fs; float depth, bt, vtn; bool on; };   // fs: constant sliding force of the corrector pass

// fprev = previous sub-step's net contact force on the point's report body (seeds the friction secant)
LG_DEV void contact_setup(Contact &c, const lg_params &P, float mu, V3 r, V3 n, float depth, V3 vc, V3 fprev) {
    c.r = r; c.n = n; c.depth = depth; c.vc = vc; c.on = depth > -P.contact_margin; c.f = v3(0, 0, 0); c.fs = v3(0, 0, 0);
    float vn0 = dot(n, vc);
    V3 vt0 = vc - n * vn0;
    c.vtn = sqrtf(dot(vt0, vt0));
    float fn_est = fmaxf(dot(n, fprev), 0.0f);
    c.bt = fminf(P.friction_damping, mu * fn_est / fmaxf(c.vtn, P.stick_velocity));
}
LG_DEV void contact_assemble(const Contact &c, const lg_params &P, float kn, AI &IA, S6 &pA) {
    if (c.on) {
        const float dt = P.sim_dt;
        float vn = dot(c.n, c.vc);
        V3 vt = c.vc - c.n * vn;
        V3 f = c.n * (P.contact_stiffness * c.depth - kn * vn) - vt * c.bt + c.fs;
        ai_add_point(IA, dt * c.bt, c.r);
        ai_add_rank1(IA, dt * (kn - c.bt), cross(c.r, c.n), c.n);
        pA.w = pA.w - cross(c.r, f);
        pA.v = pA.v - f;
    }
}
LG_DEV void contact_evaluate(Contact &c, const lg_params &P, float kn, float mu, S6 acc) {
    if (c.on) {
        V3 v1 = c.vc + (acc.v + cross(acc.w, c.r)) * P.sim_dt;
        float vn = dot(c.n, v1);
        V3 vt = v1 - c.n * vn;
        float fn = P.contact_stiffness * c.depth - kn * vn;
        if (fn <= 0.0f) { c.on = false; c.f = v3(0, 0, 0); }
        else {
            c.f = c.n * fn - vt * c.bt + c.fs;                                               // force this pass applied
            // corrector for the next pass: beyond the cone -> slide with mu f_n against the predicted slip direction;
            // inside it -> re-aim the secant at the predicted end-of-step slip speed (sticking points keep the stick impedance)
            float vtm = sqrtf(dot(vt, vt)), cone = mu * fn;
            if (c.bt * vtm > cone) { c.fs = vt * (-cone / vtm); c.bt = 0.0f; }
            else if (c.bt > 0.0f) c.bt = fminf(P.friction_damping, cone / fmaxf(vtm, P.stick_velocity));
        }
    }
}

// torque a drive still delivers at joint speed qd (fades out over the last 10 % below the velocity limit when driving
// the joint faster): keeps saturated controllers from pumping momentum through the post-integration velocity clamp
LG_DEV float motor_torque(float tau, float qd, float vlim) {
    if (vlim <= 0.0f || tau * qd <= 0.0f) return tau;
    float s = (vlim - fabsf(qd)) / (0.1f * vlim);
    return tau * fminf(fmaxf(s, 0.0f), 1.0f);
}
LG_DEV V3 clamp_norm(V3 a, float lim) {           // asset.max_linear/angular_velocity guard; maps inf/NaN to 0
    float n2 = dot(a, a);
    if (!(n2 <= lim * lim)) { float k = (n2 > 0.0f && n2 < INFINITY) ? lim * __builtin_amdgcn_rsqf(n2) : 0.0f; return a * k; }
    return a;
}

// rigid-body inertia about the body's own reference point and bias force (gyroscopic - gravity), world axes
LG_DEV void body_terms(V3 grav, float m, V3 com_l, const float *Il, const M3 &R, V3 w, V3 v, AI &I0, S6 &p0) {
    V3 c = mul(R, com_l);
    M3 Ilf; Ilf.m[0] = Il[0]; Ilf.m[1] = Il[1]; Ilf.m[2] = Il[2]; Ilf.m[3] = Il[1]; Ilf.m[4] = Il[3]; Ilf.m[5] = Il[4];
    Ilf.m[6] = Il[2]; Ilf.m[7] = Il[4]; Ilf.m[8] = Il[5];
    M3 Rt;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j2 = 0; j2 < 3; j2++) Rt.m[3 * i + j2] = R.m[3 * j2 + i];
    M3 Ic = mul(mul(R, Ilf), Rt);
#pragma unroll
    for (int i = 0; i < 9; i++) I0.H[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; i++) I0.M[i] = 0.0f;
    I0.A[0] = Ic.m[0]; I0.A[1] = Ic.m[1]; I0.A[2] = Ic.m[2]; I0.A[3] = Ic.m[4]; I0.A[4] = Ic.m[5]; I0.A[5] = Ic.m[8];
    ai_add_point(I0, m, c);
    V3 l = (v + cross(w, c)) * m;
    V3 n = mul(Ic, w) + cross(c, l);
    V3 fg = grav * m;
    p0.w = (cross(w, n) + cross(v, l)) - cross(c, fg);
    p0.v = cross(w, l) - fg;
}
// one joint of the forward-kinematics chain (world axes): parent frame (Rpar, wpar, vpar) -> this body's frame
struct FkOut { M3 R; V3 db, ax, w, v; };
LG_DEV FkOut fk_joint(const float *tj, const M3 &Rpar, V3 wpar, V3 vpar, float q, float qd) {
    // The host re-frames every body so that its joint axis is the local +x axis (fill_limb_table: "axis alignment"): the
    // joint rotation then mixes columns 1 and 2 of Rpar * Rfix and leaves column 0 = the world axis.
    FkOut o;
    o.db = mul(Rpar, v3(tj[J_POS], tj[J_POS + 1], tj[J_POS + 2]));
    M3 Rfix;
#pragma unroll
    for (int i = 0; i < 9; i++) Rfix.m[i] = tj[J_ROT + i];
    const M3 Rz = mul(Rpar, Rfix);
    o.ax = v3(Rz.m[0], Rz.m[3], Rz.m[6]);
    float sn, cs;
    __sincosf(q, &sn, &cs);               // v_sin/v_cos (|q| stays within a few rad; abs err ~1e-6)
#pragma unroll
    for (int r = 0; r < 3; r++) {
        o.R.m[3 * r] = Rz.m[3 * r];
        o.R.m[3 * r + 1] = cs * Rz.m[3 * r + 1] + sn * Rz.m[3 * r + 2];
        o.R.m[3 * r + 2] = cs * Rz.m[3 * r + 2] - sn * Rz.m[3 * r + 1];
    }
    o.w = wpar + o.ax * qd;
    o.v = vpar + cross(wpar, o.db);
    return o;
}
// LDS hand-over of the limb bodies' (I0, p0) when the helper waves compute them (k_step, quadruped kernels): 28 floats per
// (joint, lane) as 7 float4.
#define LG_BT_QUADS 7
LG_DEV void bt_store(float4 (*dst)[LG_BLOCK], int lane, const AI &I, const S6 &p) {
    dst[0][lane] = make_float4(I.A[0], I.A[1], I.A[2], I.A[3]); dst[1][lane] = make_float4(I.A[4], I.A[5], I.H[0], I.H[1]);
    dst[2][lane] = make_float4(I.H[2], I.H[3], I.H[4], I.H[5]); dst[3][lane] = make_float4(I.H[6], I.H[7], I.H[8], I.M[0]);
    dst[4][lane] = make_float4(I.M[1], I.M[2], I.M[3], I.M[4]); dst[5][lane] = make_float4(I.M[5], p.w.x, p.w.y, p.w.z);
    dst[6][lane] = make_float4(p.v.x, p.v.y, p.v.z, 0.0f);
}
LG_DEV void bt_load(const float4 (*src)[LG_BLOCK], int lane, AI &I, S6 &p) {
    float4 a = src[0][lane], b = src[1][lane], c = src[2][lane], d = src[3][lane], e = src[4][lane], f = src[5][lane], g = src[6][lane];
    I.A[0] = a.x; I.A[1] = a.y; I.A[2] = a.z; I.A[3] = a.w; I.A[4] = b.x; I.A[5] = b.y; I.H[0] = b.z; I.H[1] = b.w;
    I.H[2] = c.x; I.H[3] = c.y; I.H[4] = c.z; I.H[5] = c.w; I.H[6] = d.x; I.H[7] = d.y; I.H[8] = d.z; I.M[0] = d.w;
    I.M[1] = e.x; I.M[2] = e.y; I.M[3] = e.z; I.M[4] = e.w; I.M[5] = f.x; p.w = v3(f.y, f.z, f.w); p.v = v3(g.x, g.y, g.z);
}

// ------------------------------------------------------------------ self-collision (asset.self_collisions = 0; DESIGN.md "Self-collision")
// Twin of oracle/lg_oracle.c "self-collision": limb capsules against the base capsules and against the capsules of the other
// limbs; per limb body and partner the deepest active pair becomes a frictionless implicit spring-damper, coupled by
// mass-ratio weighted block Jacobi, and enters the LAST articulated-body pass.
// Division of labour in the four-wave step kernel: the rigid-body wave publishes every collision sphere (position, velocity),
// the group bounding spheres and the base pose to LDS during its kinematics loop; helper wave w (idle once the sub-step's
// torques / body terms are handed over) does partner m = w for all 64 (env, limb) lanes -- wave 1 also the base -- while the
// rigid-body wave runs its first pass, and leaves one record per (partner, body) in LDS; the rigid-body wave folds the
// records in before its final pass.  Kernels without three helper waves run the same detection on the rigid-body wave.
// Every narrow-phase pair test is skipped wave-uniformly unless some lane of the wave flagged that pair of bounding spheres.
#define LG_SC_REC 4                      // float4 per record: (n, f0) (pa, coef) (pb, coef of the other side) (frep, capsule, -, -)
template <class T> struct SelfLds {
    float4 pos[T::NPT][LG_BLOCK];        // sphere centres relative to the base origin (world axes); w = radius
    float4 vel[T::NPT][LG_BLOCK];        // their velocities
    float4 grp[T::NGRP][LG_BLOCK];       // bounding sphere of each point group (= shapes of one body): centre, radius
    float4 basepose[5][LG_BLOCK];        // rows of the base rotation, base angular and linear velocity (of the lane's env)
    float4 rec[T::K][T::NGRP][LG_SC_REC][LG_BLOCK];   // deepest contact of body g of this lane's limb with partner m (0: base; m: limb k ^ m); coef < 0: none
    alignas(16) int ready[LG_STEP_WAVES];   // helper wave w: sub-step number once its records are complete
    float4 base_pts[LG_MAX_BASE_POINTS];    // A.base.pts, staged once: the detection loops index them with run-time counters, and a run-time
                                            // index into the by-value kernel arguments can make hipcc copy all 3 KB of KArgs to scratch (95 vs 59 us)
};
LG_DEV V3 xyz(float4 a) { return v3(a.x, a.y, a.z); }
// closest points of two segments, parameters in [0,1] (Ericson, RTCD 5.1.9); SA / SB: that "segment" is a point (compile time)
template <bool SA, bool SB> LG_DEV void seg_seg_closest(V3 a0, V3 a1, V3 b0, V3 b1, float &s, float &t) {
    const float eps = 1e-12f;
    V3 d1 = a1 - a0, d2 = b1 - b0, r = a0 - b0;
    float a = SA ? 0.0f : dot(d1, d1), e = SB ? 0.0f : dot(d2, d2), f = SB ? 0.0f : dot(d2, r);
    if ((SA || a <= eps) && (SB || e <= eps)) { s = 0.0f; t = 0.0f; }
    else if (SA || a <= eps) { s = 0.0f; t = fminf(fmaxf(f / e, 0.0f), 1.0f); }
    else {
        float c = dot(d1, r);
        if (SB || e <= eps) { t = 0.0f; s = fminf(fmaxf(-c / a, 0.0f), 1.0f); }
        else {
            float b = dot(d1, d2), denom = a * e - b * b;
            s = (denom > eps) ? fminf(fmaxf((b * f - c * e) / denom, 0.0f), 1.0f) : 0.0f;
            t = (b * s + f) / e;
            if (t < 0.0f) { t = 0.0f; s = fminf(fmaxf(-c / a, 0.0f), 1.0f); }
            else if (t > 1.0f) { t = 1.0f; s = fminf(fmaxf((b - c) / a, 0.0f), 1.0f); }
        }
    }
}
struct SelfHit { V3 n, pa, pb; float depth, f0; };
// One capsule pair: A = segment a0-a1 (radius ra) of this lane's limb, B = the other shape.  Geometry first; the velocities
// (`vel`: callback returning the four end-point velocities) are only fetched for pairs within the contact margin.
template <bool SA, bool SB, class Vel>
LG_DEV bool capsule_contact(const lg_params &P, float kn, V3 a0, V3 a1, float ra, V3 b0, V3 b1, float rb, Vel vel, SelfHit &h) {
    float s, t;
    seg_seg_closest<SA, SB>(a0, a1, b0, b1, s, t);
    V3 ca = a0 + (a1 - a0) * s, cb = b0 + (b1 - b0) * t, diff = ca - cb;
    float dist = sqrtf(dot(diff, diff));
    float d = ra + rb - dist;
    if (!(d > -P.contact_margin)) return false;
    V3 va0, va1, vb0, vb1;
    vel(va0, va1, vb0, vb1);
    h.n = diff * (1.0f / fmaxf(dist, 1e-9f));
    V3 va = va0 + (va1 - va0) * s, vb = vb0 + (vb1 - vb0) * t;
    float f = P.contact_stiffness * d - kn * dot(h.n, va - vb);
    if (!(d > 0.0f || f > 0.0f)) return false;       // speculative range: only when approaching; overlapping shapes stay coupled (oracle comment)
    h.depth = d; h.f0 = fmaxf(f, 0.0f); h.pa = ca - h.n * ra; h.pb = cb + h.n * rb;
    return true;
}
// Detection for partner m (0: the base) of every lane of the calling wave: records sc.rec[m][*][*][ln].  Reads only LDS
// (spheres, group bounds, base pose, limb tables) and kernel arguments, so any wave of the workgroup can run it.
template <class T>
LG_DEV void self_detect(const KArgs &A, const float *lds_tab, int ln, int m, SelfLds<T> &sc) {
    constexpr int K = T::K, NCAP = T::NCAP, NGRP = T::NGRP;
    const lg_params &P = A.P;
    const float dt = P.sim_dt, kn = P.contact_stiffness * dt + P.contact_damping, margin = P.contact_margin;
    const int k = ln & (K - 1), lp = ln ^ m;
    const float *tab = lds_tab + k * Tab<T>::STRIDE, *tabp = lds_tab + (k ^ m) * Tab<T>::STRIDE;
    const int nbc = (A.base.num_pts + 1) >> 1;
    // base pose of the lane's env (m == 0 only)
    M3 R0; V3 w0 = v3(0, 0, 0), v0 = v3(0, 0, 0);
    if (m == 0) {
        const float4 a = sc.basepose[0][ln], b = sc.basepose[1][ln], c = sc.basepose[2][ln];
        R0.m[0] = a.x; R0.m[1] = a.y; R0.m[2] = a.z; R0.m[3] = b.x; R0.m[4] = b.y; R0.m[5] = b.z; R0.m[6] = c.x; R0.m[7] = c.y; R0.m[8] = c.z;
        w0 = xyz(sc.basepose[3][ln]); v0 = xyz(sc.basepose[4][ln]);
    } else {
#pragma unroll
        for (int i = 0; i < 9; i++) R0.m[i] = 0.0f;
    }
    auto base_end = [&](int c, int e) {            // end point e of base capsule c (points come in equal-radius pairs; checked at lg_create)
        const int i0 = 2 * c, i = (e && i0 + 1 < A.base.num_pts) ? i0 + 1 : i0;
        return mul(R0, xyz(sc.base_pts[i]));
    };
    // ---- stage 1: which (my body, their body) bounding spheres touch?  bit ga * NGRP + gb (gb = 0 for the base)
    float4 gme[NGRP];
#pragma unroll
    for (int g = 0; g < NGRP; g++) gme[g] = sc.grp[g][ln];
    unsigned cand = 0;
    if (m == 0) {
#pragma unroll 1
        for (int c = 0; c < nbc; c++) {
            const V3 e0 = base_end(c, 0), d = base_end(c, 1) - e0;
            const float dd = dot(d, d), rdd = dd > 0.0f ? 1.0f / dd : 0.0f, rb = sc.base_pts[2 * c].w;
#pragma unroll
            for (int ga = 0; ga < NGRP; ga++) {
                V3 rel = xyz(gme[ga]) - e0;
                V3 off = rel - d * fminf(fmaxf(dot(rel, d) * rdd, 0.0f), 1.0f);
                float lim = gme[ga].w + rb + margin;
                if (dot(off, off) < lim * lim) cand |= 1u << (ga * NGRP);
            }
        }
    } else {
#pragma unroll
        for (int gb = 0; gb < NGRP; gb++) {
            const float4 o = sc.grp[gb][lp];
#pragma unroll
            for (int ga = 0; ga < NGRP; ga++) {
                V3 off = xyz(gme[ga]) - xyz(o);
                float lim = gme[ga].w + o.w + margin;
                if (dot(off, off) < lim * lim) cand |= 1u << (ga * NGRP + gb);
            }
        }
    }
    const bool any = __builtin_amdgcn_ballot_w64(cand != 0) != 0;
    // ---- stage 2: per body of this limb the deepest active pair
#pragma unroll
    for (int ga = 0; ga < NGRP; ga++) {
        SelfHit best; best.depth = -1e30f; best.f0 = 0.0f; best.n = best.pa = best.pb = v3(0, 0, 0);
        float best_mB = 1.0f; int best_i = 0;
        const unsigned gbits = ((1u << NGRP) - 1u) << (ga * NGRP);
        if (any && __builtin_amdgcn_ballot_w64((cand & gbits) != 0) != 0) {
#pragma unroll 1
            for (int i = T::grp_cap_lo(ga); i < T::grp_cap_hi(ga); i++) {
                const int ip0 = T::cap_p0(i), ip1 = T::cap_p1(i);
                const float4 pa0 = sc.pos[ip0][ln], pa1 = sc.pos[ip1][ln];
                const bool sa = ip0 == ip1;
                const int nother = m == 0 ? nbc : NCAP;
#pragma unroll 1
                for (int j = 0; j < nother; j++) {
                    const unsigned bit = 1u << (ga * NGRP + (m == 0 ? 0 : T::cap_grp(j)));
                    if (__builtin_amdgcn_ballot_w64((cand & bit) != 0) == 0) continue;                  // wave-uniform skip
                    SelfHit h; bool hit = false; float mB;
                    if (m == 0) {
                        const V3 e0 = base_end(j, 0), e1 = base_end(j, 1);
                        auto vel = [&](V3 &va0, V3 &va1, V3 &vb0, V3 &vb1) {
                            va0 = xyz(sc.vel[ip0][ln]); va1 = xyz(sc.vel[ip1][ln]); vb0 = v0 + cross(w0, e0); vb1 = v0 + cross(w0, e1);
                        };
                        mB = A.base.mass_robot;
                        if (cand & bit) hit = sa ? capsule_contact<true, false>(P, kn, xyz(pa0), xyz(pa1), pa0.w, e0, e1, sc.base_pts[2 * j].w, vel, h)
                                                 : capsule_contact<false, false>(P, kn, xyz(pa0), xyz(pa1), pa0.w, e0, e1, sc.base_pts[2 * j].w, vel, h);
                    } else {
                        const int jp0 = T::cap_p0(j), jp1 = T::cap_p1(j);
                        const float4 pb0 = sc.pos[jp0][lp], pb1 = sc.pos[jp1][lp];
                        const bool sb = jp0 == jp1;
                        auto vel = [&](V3 &va0, V3 &va1, V3 &vb0, V3 &vb1) {
                            va0 = xyz(sc.vel[ip0][ln]); va1 = xyz(sc.vel[ip1][ln]); vb0 = xyz(sc.vel[jp0][lp]); vb1 = xyz(sc.vel[jp1][lp]);
                        };
                        mB = tabp[T::pt_joint(jp0) * LG_JS + J_SUBM];
                        if (cand & bit) {              // the shape kinds are wave-uniform (i, j are loop counters): no divergence from the dispatch
                            if (sa && sb) hit = capsule_contact<true, true>(P, kn, xyz(pa0), xyz(pa1), pa0.w, xyz(pb0), xyz(pb1), pb0.w, vel, h);
                            else if (sa) hit = capsule_contact<true, false>(P, kn, xyz(pa0), xyz(pa1), pa0.w, xyz(pb0), xyz(pb1), pb0.w, vel, h);
                            else if (sb) hit = capsule_contact<false, true>(P, kn, xyz(pa0), xyz(pa1), pa0.w, xyz(pb0), xyz(pb1), pb0.w, vel, h);
                            else hit = capsule_contact<false, false>(P, kn, xyz(pa0), xyz(pa1), pa0.w, xyz(pb0), xyz(pb1), pb0.w, vel, h);
                        }
                    }
                    if (hit && h.depth > best.depth) { best = h; best_mB = mB; best_i = i; }
                }
            }
        }
        const bool on = best.depth > -1e29f;
        const float mA = tab[T::pt_joint(T::grp_c0(ga)) * LG_JS + J_SUBM];
        sc.rec[m][ga][0][ln] = make_float4(best.n.x, best.n.y, best.n.z, best.f0);
        sc.rec[m][ga][1][ln] = make_float4(best.pa.x, best.pa.y, best.pa.z, on ? dt * kn * (1.0f + mA / best_mB) : -1.0f);
        sc.rec[m][ga][2][ln] = make_float4(best.pb.x, best.pb.y, best.pb.z, on ? dt * kn * (1.0f + best_mB / mA) : -1.0f);
        sc.rec[m][ga][3][ln] = make_float4(best.f0 / (1.0f + kn * dt * (1.0f / mA + 1.0f / best_mB)), (float)best_i, 0.0f, 0.0f);
    }
}
// Rigid-body wave, before the final pass: fold the limb side of every record into the carrying body's rigid terms.
// Returns (wave-uniform) whether any lane of the wave has a record.
template <class T>
LG_DEV bool self_apply(int ln, const V3 (&db)[T::L], AI (&I0)[T::L], S6 (&p0)[T::L], const SelfLds<T> &sc) {
    float coef[T::K][T::NGRP];                                        // all record headers in flight at once (one LDS latency, not K * NGRP)
    unsigned mask = 0;
#pragma unroll
    for (int m = 0; m < T::K; m++)
#pragma unroll
        for (int g = 0; g < T::NGRP; g++) coef[m][g] = sc.rec[m][g][1][ln].w;
#pragma unroll
    for (int m = 0; m < T::K; m++)
#pragma unroll
        for (int g = 0; g < T::NGRP; g++) if (coef[m][g] >= 0.0f) mask |= 1u << (m * T::NGRP + g);
    if (__builtin_amdgcn_ballot_w64(mask != 0) == 0) return false;
#pragma unroll
    for (int m = 0; m < T::K; m++) {
#pragma unroll
        for (int g = 0; g < T::NGRP; g++) {
            const int jc = T::pt_joint(T::grp_c0(g));                     // compile time after unrolling
            const bool hit = (mask >> (m * T::NGRP + g)) & 1u;
            if (__builtin_amdgcn_ballot_w64(hit) == 0) continue;
            if (hit) {
                const float4 r0 = sc.rec[m][g][0][ln], r1 = sc.rec[m][g][1][ln];
                V3 rj = db[0];
#pragma unroll
                for (int jj = 1; jj <= jc; jj++) rj = rj + db[jj];
                const V3 n = xyz(r0), r = xyz(r1) - rj, f = n * r0.w;
                ai_add_rank1(I0[jc], r1.w, cross(r, n), n);
                p0[jc].w = p0[jc].w - cross(r, f); p0[jc].v = p0[jc].v - f;
            }
        }
    }
    return true;
}

// `torques_ready` runs between the kinematics half (needs no torques) and the articulated-body passes: the fused step
// uses it to join the actuator waves, which compute this sub-step's torques meanwhile (k_step).
struct NoWait { LG_DEV void operator()() const {} };
// OFFLOAD: the limb bodies' (I0, p0) are computed by the helper waves meanwhile and read from `bt` after `torques_ready`.
// SC: self-collision (needs `sc`); `last`: the exported net contact forces (Frep / Fbase after this call) include the
// self-collision forces -- the policy step's last sub-step, or the sub-step entry point.
template <class T, bool HF, class Ready = NoWait, bool OFFLOAD = false, bool SC = false>
LG_DEV void physics_substep(const KArgs &A, const float *tab, int lane_k, float (&root)[13], float (&q)[T::L], float (&qd)[T::L],
                            const float (&tau)[T::L], float base_mass, float mu,
                            float (&Frep)[T::NREP][3], float (&Fbase)[3], Ready torques_ready = Ready(),
                            const float4 (*bt)[LG_BT_QUADS][LG_BLOCK] = nullptr, float4 (*fkout)[4][LG_BLOCK] = nullptr,
                            volatile int *fk_ready = nullptr, int substep_no = 0, SelfLds<T> *sc = nullptr, bool last = true,
                            bool sc_on_helpers = false /* the three helper waves run self_detect (four-wave kernels) */) {
    constexpr int K = T::K, L = T::L, NPT = T::NPT;
    const lg_params &P = A.P;
    const float dt = P.sim_dt;
    const V3 grav = v3(P.gravity[0], P.gravity[1], P.gravity[2]);
    const float kn = P.contact_stiffness * dt + P.contact_damping;

    // ---- per body, fused: kinematics -> inertia / bias -> contact candidates.  World axes; each body's spatial
    // quantities live at ITS OWN joint origin O_j (DESIGN.md "Conditioning"): db = O_j - O_parent, wb = angular
    // velocity, vb = velocity of the body point at O_j, S_j = (ax_j, 0), C_j = velocity-product acceleration.
    const M3 R0 = quat_to_mat(root + 3);
    const V3 w0 = v3(root[10], root[11], root[12]), v0 = v3(root[7], root[8], root[9]);
    V3 db[L], ax[L];
    S6 C[L];
    AI I0[L], I0b;
    S6 p0[L], p0b;
    Contact cb, cl[NPT];
    HfFetch fb, fl[NPT];               // ground samples of all collision points: fetched here, evaluated after the loop
    V3 sgc[T::NGRP];          // self-collision: bounding-sphere centre of each point group (midpoint of its two anchor points)
    {
        // 6-joint chains (Cassie) run out of registers: values that do not change over the sub-steps are re-formed here instead of being
        // hoisted out of the decimation loop and held (or spilled) across it -- the empty asm hides the invariance from LICM.
        float bm = base_mass;
        if (L > 3) asm volatile("" : "+v"(bm));
        float mass_scale = bm / A.base.mass, Il[6];
#pragma unroll
        for (int i = 0; i < 6; i++) Il[i] = A.base.inertia[i] * mass_scale;
        body_terms(grav, bm, v3(A.base.com[0], A.base.com[1], A.base.com[2]), Il, R0, w0, v0, I0b, p0b);
        // base collision points are split over the env's lanes (lane i owns point i); their inertia / bias contribution
        // joins the lane's limb contribution before the butterfly, their force is butterfly-summed afterwards
        static_assert(K <= LG_MAX_BASE_POINTS, "one base point per lane at most");
        {
            // from the lane's limb table (LDS): selecting A.base.pts[lane_k] out of the by-value kernel arguments is a lane-dependent index
            // that hipcc, depending on unrelated code, turns into a scratch copy of the array with VGPR-indexed loads (rough 64.9 -> 68.3 us)
            const float bp[4] = {tab[Tab<T>::BPT], tab[Tab<T>::BPT + 1], tab[Tab<T>::BPT + 2], tab[Tab<T>::BPT + 3]};
            cb.r = mul(R0, v3(bp[0], bp[1], bp[2]));
            cb.vc = v0 + cross(w0, cb.r);
            cb.depth = bp[3];                                            // radius until the ground height arrives
            fb = hf_fetch<HF>(A, root[0] + cb.r.x, root[1] + cb.r.y);
        }
    }
    {
        M3 Rpar = R0;
        V3 rpar = v3(0, 0, 0), wpar = w0, vpar = v0;
#pragma unroll
        for (int j = 0; j < L; j++) {
            const float *tj = tab + j * LG_JS;
            const FkOut fk = fk_joint(tj, Rpar, wpar, vpar, q[j], qd[j]);
            db[j] = fk.db; ax[j] = fk.ax;
            const V3 rj = rpar + db[j];
            const M3 &Rj = fk.R;
            const V3 wj = fk.w, vj = fk.v;
            C[j].w = cross(wj, ax[j]) * qd[j];
            C[j].v = cross(vj, ax[j]) * qd[j];
            if (!OFFLOAD) {
                float Il[6];
#pragma unroll
                for (int i = 0; i < 6; i++) Il[i] = tj[J_INERTIA + i];
                body_terms(grav, tj[J_MASS], v3(tj[J_COM], tj[J_COM + 1], tj[J_COM + 2]), Il, Rj, wj, vj, I0[j], p0[j]);
            } else {                                   // hand the body's frame to helper wave j (it computes I0[j], p0[j])
                const int ln = threadIdx.x % LG_BLOCK;
                fkout[j][0][ln] = make_float4(Rj.m[0], Rj.m[1], Rj.m[2], Rj.m[3]);
                fkout[j][1][ln] = make_float4(Rj.m[4], Rj.m[5], Rj.m[6], Rj.m[7]);
                fkout[j][2][ln] = make_float4(Rj.m[8], wj.x, wj.y, wj.z);
                fkout[j][3][ln] = make_float4(vj.x, vj.y, vj.z, 0.0f);
                if (j == L - 1) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (ln == 0 && A.debug_skip != 1) *fk_ready = substep_no;
                }
            }
#pragma unroll
            for (int i = 0; i < NPT; i++) if (T::pt_joint(i) == j) {
                const float *tp = tab + L * LG_JS + 4 * i;
                cl[i].r = mul(Rj, v3(tp[0], tp[1], tp[2]));
                V3 pw = rj + cl[i].r;
                cl[i].vc = vj + cross(wj, cl[i].r);
                cl[i].depth = tp[3];                                     // radius, for now
                cl[i].vtn = root[2] + pw.z;                              // point height, for now
                fl[i] = hf_fetch<HF>(A, root[0] + pw.x, root[1] + pw.y);
                if (SC) {                                                // publish the sphere to the env's other lanes
                    const int ln = threadIdx.x % LG_BLOCK;
                    sc->pos[i][ln] = make_float4(pw.x, pw.y, pw.z, tp[3]);
                    sc->vel[i][ln] = make_float4(cl[i].vc.x, cl[i].vc.y, cl[i].vc.z, 0.0f);
#pragma unroll
                    for (int g = 0; g < T::NGRP; g++) {
                        if (T::grp_c0(g) == i) sgc[g] = pw;
                        if (T::grp_c1(g) == i) sgc[g] = (sgc[g] + pw) * 0.5f;
                    }
                }
            }
            Rpar = Rj; rpar = rj; wpar = wj; vpar = vj;
        }
    }
    {   // the samples have had the whole kinematics pass to arrive
        float dep; V3 n;
        hf_contact<HF>(A, fb, root[2] + cb.r.z, cb.depth, dep, n);
        contact_setup(cb, P, mu, cb.r, n, dep, cb.vc, v3(Fbase[0], Fbase[1], Fbase[2]));
        cb.on = cb.on && (lane_k < A.base.num_pts);
#pragma unroll
        for (int i = 0; i < NPT; i++) {
            hf_contact<HF>(A, fl[i], cl[i].vtn, cl[i].depth, dep, n);
            contact_setup(cl[i], P, mu, cl[i].r, n, dep, cl[i].vc,
                          v3(Frep[T::pt_rep(i)][0], Frep[T::pt_rep(i)][1], Frep[T::pt_rep(i)][2]));
        }
    }

    if constexpr (SC) {                                    // group bounds and base pose for the detection (helper waves, after the hand-over barrier)
        const int ln = threadIdx.x % LG_BLOCK;
#pragma unroll
        for (int g = 0; g < T::NGRP; g++) sc->grp[g][ln] = make_float4(sgc[g].x, sgc[g].y, sgc[g].z, tab[Tab<T>::GRP + g]);
        sc->basepose[0][ln] = make_float4(R0.m[0], R0.m[1], R0.m[2], 0.0f); sc->basepose[1][ln] = make_float4(R0.m[3], R0.m[4], R0.m[5], 0.0f);
        sc->basepose[2][ln] = make_float4(R0.m[6], R0.m[7], R0.m[8], 0.0f);
        sc->basepose[3][ln] = make_float4(w0.x, w0.y, w0.z, 0.0f); sc->basepose[4][ln] = make_float4(v0.x, v0.y, v0.z, 0.0f);
    }
    LG_PROF(PF_KINEMATICS);
    torques_ready();
    if (OFFLOAD) {
#pragma unroll
        for (int j = 0; j < L; j++) bt_load(bt[j], threadIdx.x % LG_BLOCK, I0[j], p0[j]);
    }
    bool any_self = false;                                 // wave-uniform
    bool sc_missed = false;                                // the self-collision hand-over poll ran out (sticky device status)
    LG_PROF(PF_TORQUE);
    // ---- articulated-body passes with the contact impedances folded in
    S6 U[L], acc0;
    float Dinv[L], uu[L], vl[L];      // vl: 0, or +-1 = joint speed limit active in that direction (set by the previous pass)
#pragma unroll
    for (int j = 0; j < L; j++) vl[j] = 0.0f;
#pragma unroll 1
    for (int pass = 0; pass < LG_PASSES; pass++) {
        AI Ia; S6 pa;
        if constexpr (SC) if (pass == LG_PASSES - 1) {     // self-collision records enter the final pass
            const int ln = threadIdx.x % LG_BLOCK;
            if (sc_on_helpers) {                           // detection ran on the helper waves during the first pass
                volatile int4 *rdy = reinterpret_cast<volatile int4 *>(sc->ready);       // one read covers the three helpers' flags
                bool arrived = false;
                for (int spin = 0; spin < A.spin_limit; spin++) {                          // bounded, like the fk hand-over
                    const int r1 = rdy->y, r2 = rdy->z, r3 = rdy->w;
                    if (min(r1, min(r2, r3)) >= substep_no) { arrived = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                sc_missed |= !arrived;                     // reported after the passes
                __builtin_amdgcn_wave_barrier();
            } else {
                __builtin_amdgcn_wave_barrier();           // this wave's own LDS writes (kinematics loop) precede the reads: one wave, in order
#pragma unroll 1
                for (int m = 0; m < K; m++) self_detect<T>(A, tab - lane_k * Tab<T>::STRIDE, ln, m, *sc);
                __builtin_amdgcn_wave_barrier();
            }
            any_self = self_apply<T>(ln, db, I0, p0, *sc);
        }
#pragma unroll
        for (int j = L - 1; j >= 0; j--) {
            const float *tj = tab + j * LG_JS;
            AI IA = I0[j]; S6 pA = p0[j];
#pragma unroll
            for (int i = 0; i < NPT; i++) if (T::pt_joint(i) == j) { LG_PROF_COUNT(0, cl[i].on); contact_assemble(cl[i], P, kn, IA, pA); }
            if (j < L - 1) { ai_add(IA, Ia); pA = pA + pa; }
            U[j] = ai_mul_w(IA, ax[j]);
            float damp = tj[J_DAMP];
            float D = dot(ax[j], U[j].w) + tj[J_ARM] + dt * damp;
            float u = motor_torque(tau[j], qd[j], tj[J_VLIM]) - dot(ax[j], pA.w) - damp * qd[j];
            float lo = tj[J_LO], hi = tj[J_HI];
            LG_PROF_COUNT(3, lo <= hi && (q[j] + dt * qd[j] < lo || q[j] + dt * qd[j] > hi));
            LG_PROF_COUNT(4, vl[j] != 0.0f);
            if (lo <= hi) {
                float qp = q[j] + dt * qd[j];
                bool blo = qp < lo, bhi = qp > hi;
                if (blo || bhi) {
                    float viol = q[j] - (blo ? lo : hi);
                    float kl = P.limit_stiffness * dt + P.limit_damping;
                    D += dt * kl;
                    u += -P.limit_stiffness * viol - kl * qd[j];
                }
            }
            if (vl[j] != 0.0f) {      // implicit damper towards +-v_lim (100x the joint's articulated inertia), reaction on the parent
                float Bv = 100.0f * D / dt;
                u += -Bv * (qd[j] - vl[j] * tj[J_VLIM]);
                D += dt * Bv;
            }
            Dinv[j] = __builtin_amdgcn_rcpf(D); uu[j] = u;
            Ia = IA;
            ai_add_rank1(Ia, -Dinv[j], U[j].w, U[j].v);
            pa = (pA + ai_mul(Ia, C[j])) + U[j] * (u * Dinv[j]);
            ai_shift(Ia, pa, db[j]);                 // to the parent's origin (the base origin for j == 0)
        }
        LG_PROF(PF_INWARD);
        LG_PROF_COUNT(1, cb.on);
        LG_PROF_COUNT(2, any_self);
        contact_assemble(cb, P, kn, Ia, pa);      // this lane's base point (about the base origin, like Ia after the shift)
        if (SC && any_self) {                     // (final pass only) reactions of this limb's base contacts: force -n f0 on the base, implicit in the base's motion
            const int ln = threadIdx.x % LG_BLOCK;
#pragma unroll
            for (int g = 0; g < T::NGRP; g++) {
                const float4 r2 = sc->rec[0][g][2][ln];
                if (r2.w >= 0.0f) {
                    const float4 r0 = sc->rec[0][g][0][ln];
                    const V3 n = xyz(r0), cxn = cross(xyz(r2), n);
                    ai_add_rank1(Ia, r2.w, cxn, n);
                    pa.w = pa.w + cxn * r0.w; pa.v = pa.v + n * r0.w;
                }
            }
        }
        group_sum<K>(Ia, pa);                     // (limb0+limb1)+(limb2+limb3) on every lane of the env
        AI IAb = I0b; S6 pAb = p0b;
        ai_add(IAb, Ia); pAb = pAb + pa;
        float rhs[6] = {-pAb.w.x, -pAb.w.y, -pAb.w.z, -pAb.v.x, -pAb.v.y, -pAb.v.z}, a0[6];
        bool ok = solve6(IAb, rhs, a0);
        if (!ok) { a0[0] = a0[1] = a0[2] = a0[3] = a0[4] = a0[5] = 0.0f; }
        acc0.w = v3(a0[0], a0[1], a0[2]); acc0.v = v3(a0[3], a0[4], a0[5]);
        contact_evaluate(cb, P, kn, mu, acc0);
        LG_PROF(PF_BASE);
        S6 a = acc0;
#pragma unroll
        for (int j = 0; j < L; j++) {
            S6 ap;
            ap.w = a.w + C[j].w;
            ap.v = (a.v + cross(a.w, db[j])) + C[j].v;
            float qdd = (uu[j] - dot(U[j], ap)) * Dinv[j];
            a.w = ap.w + ax[j] * qdd;
            a.v = ap.v;
            uu[j] = qdd;
            {
                float lim = tab[j * LG_JS + J_VLIM], qn = qd[j] + dt * qdd;
                if (lim > 0.0f && vl[j] == 0.0f && fabsf(qn) > lim) vl[j] = qn > 0.0f ? 1.0f : -1.0f;
            }
#pragma unroll
            for (int i = 0; i < NPT; i++) if (T::pt_joint(i) == j) contact_evaluate(cl[i], P, kn, mu, a);
        }
        LG_PROF(PF_OUTWARD);
    }

    if constexpr (SC) if (sc_missed && threadIdx.x % LG_BLOCK == 0) lg_report(A.status, LG_STATUS_SELF_COLLISION_TIMEOUT);
    // ---- semi-implicit Euler
#pragma unroll
    for (int j = 0; j < L; j++) {
        float v = qd[j] + dt * uu[j];
        float lim = tab[j * LG_JS + J_VLIM];
        if (lim > 0.0f) v = fminf(fmaxf(v, -lim), lim);
        qd[j] = v;
        q[j] += dt * v;
    }
    {
        V3 a_lin = acc0.v + cross(w0, v0);
        V3 w1 = clamp_norm(w0 + acc0.w * dt, 1000.0f), v1 = clamp_norm(v0 + a_lin * dt, 1000.0f);
        root[7] = v1.x; root[8] = v1.y; root[9] = v1.z; root[10] = w1.x; root[11] = w1.y; root[12] = w1.z;
        root[0] += dt * v1.x; root[1] += dt * v1.y; root[2] += dt * v1.z;
        float x = root[3], y = root[4], z = root[5], w = root[6], hx = 0.5f * dt * w1.x, hy = 0.5f * dt * w1.y, hz = 0.5f * dt * w1.z;
        float nx = x + (hx * w + hy * z - hz * y), ny = y + (hy * w + hz * x - hx * z), nz = z + (hz * w + hx * y - hy * x);
        float nw = w - (hx * x + hy * y + hz * z);
        float inv = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz + nw * nw);
        root[3] = nx * inv; root[4] = ny * inv; root[5] = nz * inv; root[6] = nw * inv;
    }
    // ---- net contact force per report body of this limb, and of the base
#pragma unroll
    for (int r = 0; r < T::NREP; r++) {
        V3 f = v3(0, 0, 0);
#pragma unroll
        for (int i = 0; i < NPT; i++) if (T::pt_rep(i) == r) f = f + cl[i].f;
        Frep[r][0] = f.x; Frep[r][1] = f.y; Frep[r][2] = f.z;
    }
    Fbase[0] = group_sum<K>(cb.f.x); Fbase[1] = group_sum<K>(cb.f.y); Fbase[2] = group_sum<K>(cb.f.z);
    if (SC && last && any_self) {                 // exported net contact forces include the self-collision forces (estimate of the record)
        const int ln = threadIdx.x % LG_BLOCK;
        V3 fb = v3(0, 0, 0);
#pragma unroll 1
        for (int m = 0; m < K; m++) {
#pragma unroll
            for (int g = 0; g < T::NGRP; g++) {
                const float4 r1 = sc->rec[m][g][1][ln];
                if (r1.w >= 0.0f) {
                    const float4 r0 = sc->rec[m][g][0][ln], r3 = sc->rec[m][g][3][ln];
                    const V3 f = xyz(r0) * r3.x;
                    const int ci = (int)r3.y;
#pragma unroll
                    for (int c = T::grp_cap_lo(g); c < T::grp_cap_hi(g); c++) if (c == ci) {
                        const int r = T::pt_rep(T::cap_p0(c));
                        Frep[r][0] += f.x; Frep[r][1] += f.y; Frep[r][2] += f.z;
                    }
                    if (m == 0) fb = fb - f;
                }
            }
        }
        Fbase[0] += group_sum<K>(fb.x); Fbase[1] += group_sum<K>(fb.y); Fbase[2] += group_sum<K>(fb.z);
    }
    LG_PROF(PF_INTEGRATE);
}

// ------------------------------------------------------------------ torques (legged_robot.py:371-395)
template <int L> LG_DEV void pd_torques(const lg_params &P, const float *tab, const float (&act)[L], const float (&q)[L],
                                        const float (&qd)[L], const float *last_qd /* this limb's last_dof_vel in memory (V control only) */, float (&tau)[L]) {
#pragma unroll
    for (int j = 0; j < L; j++) {
        const float *tj = tab + j * LG_JS;
        float aj = act[j];
        if (L > 3) asm volatile("" : "+v"(aj));        // long chains: the scaled action is re-formed per sub-step, not held across the loop
        float a = aj * P.action_scale, t;
        if (P.control_type == LG_CTRL_P) t = tj[J_KP] * (a + tj[J_Q0] - q[j]) - tj[J_KD] * qd[j];
        else if (P.control_type == LG_CTRL_V) t = tj[J_KP] * (a - qd[j]) - tj[J_KD] * (qd[j] - last_qd[j]) / P.sim_dt;
        else t = a;
        tau[j] = fminf(fmaxf(t, -tj[J_TLIM]), tj[J_TLIM]);
    }
}

// ------------------------------------------------------------------ commands / heights / reset values
LG_DEV void resample_commands_u(const lg_params &P, const float (&u)[4], float (&cmd)[4]) {   // :347-369, from four uniforms
    cmd[0] = urange(P.cmd_lin_vel_x[0], P.cmd_lin_vel_x[1], u[0]);
    cmd[1] = urange(P.cmd_lin_vel_y[0], P.cmd_lin_vel_y[1], u[1]);
    if (P.heading_command) cmd[3] = urange(P.cmd_heading[0], P.cmd_heading[1], u[2]);
    else cmd[2] = urange(P.cmd_ang_vel_yaw[0], P.cmd_ang_vel_yaw[1], u[2]);
    float keep = (sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]) > 0.2f) ? 1.0f : 0.0f;
    cmd[0] *= keep; cmd[1] *= keep;
}
LG_DEV void resample_commands(const lg_params &P, int e, int64_t step, int purpose, float (&cmd)[4]) {
    float u[4];
    rand4(P.seed, e, step, purpose, 0, u);
    resample_commands_u(P, u, cmd);
}
LG_DEV float wrap_to_pi(float a) {                                                                       // utils/math.py:45-48
    const float two_pi = 6.2831855f, pi = 3.14159274f;
    a = fmodf(a, two_pi); if (a < 0.0f) a += two_pi;
    if (a > pi) a -= two_pi;
    return a;
}
// _get_heights (:831-869) and the height part of compute_observations (:224-226) for the fused step, spread over ALL waves
// of the workgroup: an env's chunks (4 points = one Philox block each) are dealt to K * LG_STEP_WAVES "virtual lanes";
// thread (wave, lane = (env, limb k)) is virtual lane k + K * wave and keeps its chunks' heights in registers between
// sampling (before the reset decision) and the observation (after it, Q7).  All 3*4*NCH gathers are issued before the
// first one is consumed (the samples are L2-resident but ~1 us away for a lone wave).
template <class T, int NW = LG_STEP_WAVES> struct HeightCrew {
    static constexpr int NV = T::K * NW;
    static constexpr int NCH = ((LG_MAX_HEIGHT_POINTS + 3) / 4 + NV - 1) / NV;     // chunks per thread: 3 (K = 4) / 6 (K = 2)
    float h[NCH][4];

    template <bool HF>
    LG_DEV float sample(const KArgs &A, int e, int sub, bool live, float x, float y, float z, float q5, float q6) {
        const lg_params &P = A.P;
        float qy[4] = {0, 0, q5, q6};
        float nrm = fmaxf(sqrtf(qy[2] * qy[2] + qy[3] * qy[3]), 1e-9f);
        qy[2] /= nrm; qy[3] /= nrm;
        float *mh = A.B.measured_heights + (size_t)e * P.num_height_points;
        const int np = P.num_height_points;
        const int16_t *H = A.B.height_samples;
        int16_t s0[NCH][4], s1[NCH][4], s2[NCH][4];
        if (HF) {
#pragma unroll
            for (int b = 0; b < NCH; b++)
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    int i = min(4 * (sub + b * NV) + t, np - 1);                   // clamped: the tail re-reads a valid point
                    V3 p = quat_apply(qy, v3(A.hpts[2 * i], A.hpts[2 * i + 1], 0.0f));
                    float px = p.x + x + P.hf_border, py = p.y + y + P.hf_border;
                    if (!(fabsf(px) < 1e8f)) px = 0.0f;                            // non-finite pose (reset follows): keep the conversion defined
                    if (!(fabsf(py) < 1e8f)) py = 0.0f;
                    int ix = (int)(px / P.hf_horizontal_scale), iy = (int)(py / P.hf_horizontal_scale);   // .long(): truncation
                    ix = min(max(ix, 0), P.hf_rows - 2); iy = min(max(iy, 0), P.hf_cols - 2);
                    const int16_t *hp = H + ix * P.hf_cols + iy;                   // rows x cols < 2^31 (checked at bind)
                    typedef int16_t pair16 __attribute__((ext_vector_type(2), aligned(2)));
                    const pair16 w = *reinterpret_cast<const pair16 *>(hp);        // (ix, iy), (ix, iy + 1): neighbours in memory, one 4-byte gather
                    s0[b][t] = w.x; s2[b][t] = w.y; s1[b][t] = hp[P.hf_cols];
                }
        }
        float hsum = 0.0f;
#pragma unroll
        for (int b = 0; b < NCH; b++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                float hv = 0.0f;
                if (HF) {
                    int16_t m = s0[b][t] < s1[b][t] ? s0[b][t] : s1[b][t];
                    m = m < s2[b][t] ? m : s2[b][t];
                    hv = (float)m * P.hf_vertical_scale;
                }
                h[b][t] = hv;
                const int i = 4 * (sub + b * NV) + t;
                if (i < np) { if (live) mh[i] = hv; hsum += z - hv; }
            }
        return hsum;
    }

    // the height-noise uniforms of this thread's chunks (state-independent): drawn while the thread would otherwise wait
    LG_DEV void draw_noise(const lg_params &P, int e, int sub, int64_t step, float (&un)[NCH][4]) const {
        const int nchunk = (P.num_height_points + 3) >> 2;
#pragma unroll
        for (int b = 0; b < NCH; b++) {
            const int c = sub + b * NV;
            un[b][0] = un[b][1] = un[b][2] = un[b][3] = 0.0f;
            if (c < nchunk && P.add_noise) rand4(P.seed, e, step, RNG_NOISE_H, c, un[b]);
        }
    }
    LG_DEV void write_obs(const KArgs &A, int e, int sub, bool live, int64_t step, float root_z, const float (*un)[4] = nullptr) const {
        const lg_params &P = A.P;
        float *obs = A.B.obs_buf + (size_t)e * P.num_obs;
        const int np = P.num_height_points, nchunk = (np + 3) >> 2;
#pragma unroll
        for (int b = 0; b < NCH; b++) {
            const int c = sub + b * NV;
            if (c < nchunk) {
                float u[4] = {0, 0, 0, 0};
                if (P.add_noise) {
                    if (un) { u[0] = un[b][0]; u[1] = un[b][1]; u[2] = un[b][2]; u[3] = un[b][3]; }
                    else rand4(P.seed, e, step, RNG_NOISE_H, c, u);
                }
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    int i = 4 * c + t;
                    if (i < np) {
                        float hh = root_z - 0.5f - h[b][t];
                        float o = fminf(fmaxf(hh, -1.0f), 1.0f) * P.obs_scale_height;
                        if (P.add_noise) o += (2.0f * u[t] - 1.0f) * P.noise_height;
                        o = fminf(fmaxf(o, -P.clip_observations), P.clip_observations);
                        if (live) obs[48 + i] = o;
                    }
                }
            }
        }
    }
};

// The uniforms of a step (Philox blocks keyed by env / step / purpose: ~0.27 us each on a lone wave) do not depend on the
// state: in the multi-wave kernels the helper waves draw them for EVERY lane during the first sub-step, when they have
// delivered their torques / body terms and would otherwise wait (ResetRand::draw, slots dealt round-robin to the helpers), and
// the rigid-body wave only reads the ones it needs: the observation noise (3 groups; the action group's scale is 0), the
// command resampling, the push, and a reset's dof / root / command / terrain draws -- a workgroup with a reset env no longer
// runs 1.4 us behind the others (the kernel ends with its slowest workgroup), and every workgroup saves ~1 us of noise draws.
template <class T> struct ResetRand {
    static constexpr int NB = (T::L + 2) / 4 + 1;                  // Philox blocks covering the limb's L consecutive dofs
    static constexpr int NZB = T::L > 4 ? 2 : 1;                   // blocks per observation group of the lane
    enum { ROOT0 = NB, ROOT1, CMD, TERRAIN, CMD_STEP, PUSH, NOISE, SLOTS = NOISE + 3 * NZB };
    float4 u[SLOTS][LG_BLOCK];
    LG_DEV void put(const lg_params &P, int e, int64_t step, int lane, int slot, int purpose, int block) {
        float r[4];
        rand4(P.seed, e, step, purpose, block, r);
        u[slot][lane] = make_float4(r[0], r[1], r[2], r[3]);
    }
    // helper `h` of `nh` draws the slots s with s % nh == h (wave-uniform branches)
    LG_DEV void draw(const lg_params &P, int e, int k, int64_t step, int lane, int h, int nh) {
        const int b0 = (k * T::L) >> 2;
#pragma unroll
        for (int s = 0; s < SLOTS; s++) {
            if (s % nh != h) continue;
            if (s < NB) put(P, e, step, lane, s, RNG_DOF, b0 + s);
            else if (s == ROOT0) put(P, e, step, lane, s, RNG_ROOT, 0);
            else if (s == ROOT1) put(P, e, step, lane, s, RNG_ROOT, 1);
            else if (s == CMD) put(P, e, step, lane, s, RNG_CMD_RESET, 0);
            else if (s == TERRAIN) { if (P.terrain_curriculum) put(P, e, step, lane, s, RNG_TERRAIN, 0); }
            else if (s == CMD_STEP) put(P, e, step, lane, s, RNG_CMD_STEP, 0);
            else if (s == PUSH) { if (P.push_interval > 0 && step % P.push_interval == 0) put(P, e, step, lane, s, RNG_PUSH, 0); }
            else if (P.add_noise) put(P, e, step, lane, s, RNG_NOISE, ((s - NOISE) / NZB * T::K + k) * 2 + (s - NOISE) % NZB);
        }
    }
    LG_DEV void get(int slot, int lane, float (&out)[4]) const { const float4 w = u[slot][lane]; out[0] = w.x; out[1] = w.y; out[2] = w.z; out[3] = w.w; }
};

// New state of a reset environment (reset_idx :147-191).  Every lane of the env computes the shared part
// identically; `origin` is in/out (terrain curriculum :446-469), lane-0 writes are done by the caller.
template <class T>
LG_DEV void reset_values(const KArgs &A, const float *tab, int e, int k, int64_t step, float (&root)[13], float (&q)[T::L],
                         float (&qd)[T::L], float (&cmd)[4], float (&origin)[3], int &level, bool &level_changed,
                         const ResetRand<T> *rr = nullptr, int lane = 0) {
    constexpr int L = T::L;
    const lg_params &P = A.P;
    float u[4], v[4];
    auto uniforms = [&](int purpose, int block, int slot, float (&out)[4]) {
        if (rr) rr->get(slot, lane, out);
        else rand4(P.seed, e, step, purpose, block, out);
    };
    level_changed = false;
    if (P.terrain_curriculum && A.B.terrain_levels) {
        float dx = root[0] - origin[0], dy = root[1] - origin[1], dist = sqrtf(dx * dx + dy * dy);
        int up = dist > P.terrain_env_length / 2;
        int down = (dist < sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]) * P.max_episode_length_s * 0.5f) && !up;
        int lvl = A.B.terrain_levels[e] + up - down;
        if (lvl >= P.terrain_num_rows) {
            uniforms(RNG_TERRAIN, 0, ResetRand<T>::TERRAIN, u);
            lvl = (int)(u[0] * P.terrain_num_rows);
            if (lvl >= P.terrain_num_rows) lvl = P.terrain_num_rows - 1;
        } else if (lvl < 0) lvl = 0;
        level = lvl; level_changed = true;
        const float *to = A.B.terrain_origins + ((size_t)lvl * P.terrain_num_cols + A.B.terrain_types[e]) * 3;
        origin[0] = to[0]; origin[1] = to[1]; origin[2] = to[2];
    }
    const int b0 = (k * L) >> 2;
#pragma unroll
    for (int j = 0; j < L; j++) {
        int d = k * L + j;
        float uj;
        if (rr) uj = reinterpret_cast<const float *>(&rr->u[(d >> 2) - b0][lane])[d & 3];
        else {
            rand4(P.seed, e, step, RNG_DOF, d >> 2, u);
            uj = (d & 3) == 0 ? u[0] : ((d & 3) == 1 ? u[1] : ((d & 3) == 2 ? u[2] : u[3]));
        }
        q[j] = tab[j * LG_JS + J_Q0] * urange(0.5f, 1.5f, uj);
        qd[j] = 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 13; i++) root[i] = P.base_init_state[i];
    root[0] += origin[0]; root[1] += origin[1]; root[2] += origin[2];
    uniforms(RNG_ROOT, 0, ResetRand<T>::ROOT0, u); uniforms(RNG_ROOT, 1, ResetRand<T>::ROOT1, v);
    if (P.custom_origins) { root[0] += urange(-1.0f, 1.0f, u[0]); root[1] += urange(-1.0f, 1.0f, u[1]); }
    root[7] = urange(-0.5f, 0.5f, u[2]); root[8] = urange(-0.5f, 0.5f, u[3]);
    root[9] = urange(-0.5f, 0.5f, v[0]); root[10] = urange(-0.5f, 0.5f, v[1]);
    root[11] = urange(-0.5f, 0.5f, v[2]); root[12] = urange(-0.5f, 0.5f, v[3]);
    uniforms(RNG_CMD_RESET, 0, ResetRand<T>::CMD, u);
    resample_commands_u(P, u, cmd);
}

// ------------------------------------------------------------------ observations (legged_robot.py:212-230, :100-101)
// lane k owns slots k*L..k*L+L-1 of each of the four 12-wide groups [base|dof_pos|dof_vel|actions]
template <class T>
LG_DEV void write_observations(const KArgs &A, int e, int k, bool live, int64_t step, const float *root, const float (&q)[T::L],
                               const float (&qd)[T::L], const float (&act)[T::L], const float *tab, V3 blv, V3 bav, V3 pg,
                               const float (&cmd)[4], bool heights_from_buffer /* k_obs: also the height block, from measured_heights */,
                               const ResetRand<T> *rr = nullptr, int lane = 0, float *obs_out = nullptr /* instead of B.obs_buf (rollout storage) */,
                               float *lds_row = nullptr /* rollout kernel: LDS copy of this env's row, the next step's actor input */) {
    constexpr int K = T::K, L = T::L;
    const lg_params &P = A.P;
    float head[12] = {blv.x * P.obs_scale_lin_vel, blv.y * P.obs_scale_lin_vel, blv.z * P.obs_scale_lin_vel,
                      bav.x * P.obs_scale_ang_vel, bav.y * P.obs_scale_ang_vel, bav.z * P.obs_scale_ang_vel,
                      pg.x, pg.y, pg.z,
                      cmd[0] * P.obs_scale_lin_vel, cmd[1] * P.obs_scale_lin_vel, cmd[2] * P.obs_scale_ang_vel};
    float hnz[12] = {P.noise_lin_vel, P.noise_lin_vel, P.noise_lin_vel, P.noise_ang_vel, P.noise_ang_vel, P.noise_ang_vel,
                     P.noise_gravity, P.noise_gravity, P.noise_gravity, 0.0f, 0.0f, 0.0f};
    float val[4][L], nz[4][L];
#pragma unroll
    for (int j = 0; j < L; j++) {
        float hv = head[j], hn = hnz[j];
#pragma unroll
        for (int kk = 1; kk < K; kk++) { hv = (k == kk) ? head[kk * L + j] : hv; hn = (k == kk) ? hnz[kk * L + j] : hn; }
        val[0][j] = hv; nz[0][j] = hn;
        val[1][j] = (q[j] - tab[j * LG_JS + J_Q0]) * P.obs_scale_dof_pos; nz[1][j] = P.noise_dof_pos;
        val[2][j] = qd[j] * P.obs_scale_dof_vel; nz[2][j] = P.noise_dof_vel;
        val[3][j] = act[j]; nz[3][j] = 0.0f;
    }
    float *obs = (obs_out ? obs_out : A.B.obs_buf) + (size_t)e * P.num_obs;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        float u[4], u2[4];
        const bool noisy = P.add_noise && g < 3;                   // the action group's noise scale is 0 (:230): no draw, same value
        if (noisy) {
            if (rr) { rr->get(ResetRand<T>::NOISE + g * ResetRand<T>::NZB, lane, u); if (L > 4) rr->get(ResetRand<T>::NOISE + g * ResetRand<T>::NZB + 1, lane, u2); }
            else {
                rand4(P.seed, e, step, RNG_NOISE, (g * K + k) * 2, u);
                if (L > 4) rand4(P.seed, e, step, RNG_NOISE, (g * K + k) * 2 + 1, u2);
            }
        }
#pragma unroll
        for (int j = 0; j < L; j++) {
            float o = val[g][j];
            if (noisy) {
                float uj = (j < 4) ? u[j & 3] : u2[j & 3];
                o += (2.0f * uj - 1.0f) * nz[g][j];
            }
            o = fminf(fmaxf(o, -P.clip_observations), P.clip_observations);
            if (live) obs[g * 12 + k * L + j] = o;
            if (lds_row) lds_row[g * 12 + k * L + j] = o;
        }
    }
    if (P.measure_heights && heights_from_buffer) {
        const float *mh = A.B.measured_heights + (size_t)e * P.num_height_points;
        const int nchunk = (P.num_height_points + 3) >> 2;
        for (int c = k; c < nchunk; c += K) {          // chunk c = points 4c..4c+3, one Philox block each
            float u[4] = {0, 0, 0, 0}, hv[4];
#pragma unroll
            for (int t = 0; t < 4; t++) hv[t] = mh[min(4 * c + t, P.num_height_points - 1)];
            if (P.add_noise) rand4(P.seed, e, step, RNG_NOISE_H, c, u);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                int i = 4 * c + t;
                if (i < P.num_height_points) {
                    float h = root[2] - 0.5f - hv[t];
                    float o = fminf(fmaxf(h, -1.0f), 1.0f) * P.obs_scale_height;
                    if (P.add_noise) o += (2.0f * u[t] - 1.0f) * P.noise_height;
                    o = fminf(fmaxf(o, -P.clip_observations), P.clip_observations);
                    if (live) obs[48 + i] = o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ LDS staging helpers
template <class T> LG_DEV void stage_base_points(const KArgs &A, SelfLds<T> *sc) {       // before stage_limb_table's barrier; static indices only
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < LG_MAX_BASE_POINTS; i++) sc->base_pts[i] = make_float4(A.base.pts[i][0], A.base.pts[i][1], A.base.pts[i][2], A.base.pts[i][3]);
    }
}
template <class T> LG_DEV void stage_limb_table(const KArgs &A, float *lds_tab) {
    for (int i = threadIdx.x; i < T::K * Tab<T>::STRIDE; i += blockDim.x) lds_tab[i] = A.limb_table[i];
    __syncthreads();
}

// ------------------------------------------------------------------ extras["episode"] finisher (legged_robot.py:179-188)
// Deferred extras: with A.defer the episode accumulators and the level partial sums alternate between two slots by
// step parity; launch s fills slot s & 1 while its workgroup 0 finishes slot (s - 1) & 1, which the previous launch completed.
LG_DEV float *accum_slot(const KArgs &A, int64_t step) { return (A.defer && (step & 1)) ? A.accum_alt : A.B.extras_accum; }
LG_DEV int *level_parts_slot(const KArgs &A, int64_t step, int nwg) { return reinterpret_cast<int *>(A.done_counter + 1) + ((step & 1) ? nwg : 0); }   // (always by parity: the slot of step - 1 is valid whatever mode wrote it)
LG_DEV void finish_extras(const KArgs &A, int t, int64_t step_used, bool publish_step, float *accum, const int *parts = nullptr, int n_parts = 0, bool do_levels = true) {
    const lg_params &P = A.P;
    const int R = P.num_reward_slots;
    __shared__ float level_part[16];
    if (publish_step && t == 0 && A.B.step_counter) A.B.step_counter[0] = step_used;
    // accumulators were updated with device-scope atomics by other workgroups: read them past the L1 (sc1 loads)
    float cnt = __hip_atomic_load(accum + R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float v = (t < R) ? __hip_atomic_load(accum + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
    const bool levels = do_levels && P.terrain_curriculum && A.B.terrain_levels;
    if (levels) {
        // mean terrain level (legged_robot.py:185-186): every thread of the workgroup, eight independent L1-bypassing loads in flight
        // each -- the serial 64-lane scan this replaces took 15 us at 4096 envs, and the kernel ends with this workgroup
        float acc = 0.0f;
        const int nt = blockDim.x;
        constexpr int INFLIGHT = 8;                                // (32 in flight measured no faster)
        // k_step with helper waves: every workgroup left the sum of its own envs' levels behind its ticket (HelperWave, after P3), so the
        // last one adds gridDim.x numbers -- one round trip past the caches instead of num_envs / (8 x threads) of them (2 at 4096 envs,
        // 4 at 8192: the light profile showed this workgroup ending 5 / 9 us after every other one)
        const int n_items = parts ? n_parts : P.num_envs;
        const int *items = parts ? parts : A.B.terrain_levels;
        for (int e0 = t; e0 < n_items; e0 += nt * INFLIGHT) {
            int lv[INFLIGHT];
#pragma unroll
            for (int u = 0; u < INFLIGHT; u++) {
                const int e = e0 + u * nt;
                lv[u] = e < n_items ? __hip_atomic_load(items + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            }
#pragma unroll
            for (int u = 0; u < INFLIGHT; u++) acc += (float)lv[u];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if ((t & 63) == 0) level_part[t >> 6] = acc;
    }
    __syncthreads();
    if (t >= 64) return;                            // blocks wider than one wave (k_step with actuator waves): wave 0 finishes
    if (t < R) {
        if (cnt > 0.0f) A.B.episode_means[t] = v / cnt / P.max_episode_length_s;
        __hip_atomic_store(accum + t, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (t == R) __hip_atomic_store(accum + R, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (levels && t == 0) {
        float acc = 0.0f;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) acc += level_part[w];
        A.B.episode_means[R] = acc / (float)P.num_envs;
    }
}
// ------------------------------------------------------------------ lg_rollout_policy: the segment's finisher
// Run by the first wave of the workgroup that takes the LAST ticket of the multi-step launch (every other workgroup has finished all its
// steps, so all their episode atomics are performed and all of them have read the step counter): extras["episode"] (legged_robot.py:179-188)
// = the sums of the LAST step of the segment in which any env was reset (the dictionary stays stale otherwise, quirk Q4); the device step
// counter moves to the last executed step; the per-step accumulators are left zeroed for the next segment.
LG_DEV void roll_finish(const KArgs &A, int t, int64_t step0) {
    const lg_params &P = A.P;
    const int R = P.num_reward_slots, stride = LG_NUM_REWARD_TERMS + 2;
    int last = -1;
    for (int s = A.roll.steps - 1; s >= 0 && last < 0; s--)
        if (__hip_atomic_load(A.roll.extras + (size_t)s * stride + R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0.0f) last = s;
    if (last >= 0 && t < R) {
        const float v = __hip_atomic_load(A.roll.extras + (size_t)last * stride + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float c = __hip_atomic_load(A.roll.extras + (size_t)last * stride + R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        A.B.episode_means[t] = v / c / P.max_episode_length_s;
    }
    if (t == 0 && A.B.step_counter) A.B.step_counter[0] = step0 + A.roll.steps - 1;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (int i = t; i < A.roll.steps * stride; i += 64) __hip_atomic_store(A.roll.extras + i, 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------ THE fused policy-step kernel
// One workgroup = 64 (env, limb) lanes.  A lone wave issues one instruction every ~6.5 cycles and cannot overlap its own
// MFMA and VALU work (tools/ubench), and at 4096 envs only one wave per CU exists -- so with the actuator net the workgroup
// is WAVE-SPECIALISED across the CU's four SIMDs: wave 0 runs the rigid-body step, waves 1..L own the LSTM of joint 0..L-1
// of every lane (state resident in their registers for the whole step) and compute each sub-step's torques while wave 0
// does the torque-independent kinematics half.  Hand-over through LDS (lds_x -> lds_tau), two barriers per sub-step.
// Post-physics, every variant: the helper waves join the height sampling / height observations (HeightCrew), 4 x the lanes.
template <bool OFF, int L = 3> struct StepSharedT {              // LDS hand-over between the rigid-body wave and the helpers
    float pose[LG_BLOCK][5];                                     // x, y, z, q.z, q.w of the lane's env after the last sub-step
    float hsum[LG_STEP_WAVES][LG_BLOCK];                         // partial sums of (root z - height) per wave
    float root_z[LG_BLOCK];                                      // root z after the reset decision (observation input, Q7)
    int   rst[LG_BLOCK];                                         // reset flag of the lane's env
    float r_t[LG_NUM_REWARD_TERMS][LG_BLOCK];                    // this step's scaled reward terms, for the episode-sum bookkeeping
    // body-terms offload (quadruped kernels, 4 waves): sub-step inputs of the lane and the helpers' (I0, p0) per joint
    float4 fk[OFF ? L : 1][4][LG_BLOCK];                         // per joint: world rotation (9), angular (3) and origin (3) velocity of the body
    float4 bt[OFF ? L : 1][LG_BT_QUADS][LG_BLOCK];
    int    fk_ready;                                             // = sub-step number once the rigid-body wave has published fk[] (polled by the helpers)
};

// episode_sums[name] += term (:203); read + zeroed for reset envs, whose sums feed extras["episode"] (reset_idx :179-183).
// Runs on a helper wave (one lane per env): off the rigid-body wave's critical path.
struct EpisodeSums {
    float sum[LG_NUM_REWARD_TERMS] = {};
    LG_DEV void load(const KArgs &A, int e) {                     // issued early: the running sums do not depend on this step
#pragma unroll
        for (int t = 0; t < LG_NUM_REWARD_TERMS; t++) {
            const int slot = A.P.reward_slot[t];
            sum[t] = slot >= 0 ? A.B.episode_sums[(size_t)slot * A.P.num_envs + e] : 0.0f;
        }
    }
    // keep == this lane keeps an env's sums.  The finished episodes of the workgroup leave as ONE wave instruction: the keeper lanes
    // park their sums in sh.r_t (this wave has consumed it), lane t adds term t over the reset envs and issues the atomic for its
    // slot -- 18 single-lane atomics to one 128-byte line took ~3 us to drain (measured: workgroups with a reset env ended 3 us
    // later in the ticket section), and the kernel ends with its slowest workgroup.
    template <int K, class SH> LG_DEV void update(const KArgs &A, int e, int lane, SH &sh, bool keep, float *accum) {
        const lg_params &P = A.P;
        const bool reset = keep && sh.rst[lane] != 0;
        if (keep) {
#pragma unroll
            for (int t = 0; t < LG_NUM_REWARD_TERMS; t++) {
                const int slot = P.reward_slot[t];
                if (slot >= 0) {
                    sum[t] += sh.r_t[t][lane];
                    A.B.episode_sums[(size_t)slot * P.num_envs + e] = reset ? 0.0f : sum[t];
                }
            }
        }
        const unsigned long long finished = __ballot(reset);
        if (finished == 0) return;                                 // wave-uniform
#pragma unroll
        for (int t = 0; t < LG_NUM_REWARD_TERMS; t++) sh.r_t[t][lane] = reset ? sum[t] : 0.0f;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < LG_NUM_REWARD_TERMS) {
            int slot = -1;                                         // P.reward_slot[lane] without a lane-dependent index into the kernarg struct
#pragma unroll                                                     // (that makes hipcc copy all of KArgs to scratch)
            for (int t = 0; t < LG_NUM_REWARD_TERMS; t++) slot = lane == t ? P.reward_slot[t] : slot;
            float tot = 0.0f;
            for (int i = 0; i < LG_BLOCK; i += K) tot += sh.r_t[lane][i];
            if (slot >= 0) atomicAdd(accum + slot, tot);
        } else if (lane == LG_NUM_REWARD_TERMS) atomicAdd(accum + P.num_reward_slots, (float)__popcll(finished));
    }
};

template <class T, bool NET, bool HF, int NW, bool SC = false> struct HelperWave {
    // wave = 1 .. LG_STEP_WAVES-1; with the actuator net also the LSTM of joint (wave - 1); all 64 lanes active
    static constexpr bool OFF = NW >= 2;                         // the helpers also compute (I0, p0) of the limb bodies each sub-step:
                                                                 // wave w takes bodies w-1, w-1 + (NW-1), ...
    static LG_DEV void run(const KArgs &A, int wave, int lane, int e, int k, int d0, bool live, int64_t step, const float *tab,
                           float2 (*lds_x)[LG_BLOCK], float (*lds_tau)[LG_BLOCK], StepSharedT<OFF, T::L> &sh, SelfLds<T> *sc = nullptr,
                           const float *lds_tab = nullptr, ResetRand<T> *reset_rand = nullptr, int *s_last = nullptr, float4 (*hnoise)[LG_BLOCK] = nullptr,
                           int sub0 = 0 /* sub-steps before this policy step in the launch (rollout kernel) */, float *roll_accum = nullptr /* rollout kernel: this step's episode accumulators */) {
        const lg_buffers &B = A.B;
        const lg_params &P = A.P;
        const int j = wave - 1;
        bool missed = false;                                     // a hand-over poll of this wave ran out (sticky device status, lg_report)
        LstmSplit st;
        float (*part[4])[4] = {st.h0, st.c0, st.h1, st.c1};
        float *row[4] = {nullptr, nullptr, nullptr, nullptr};
        LstmLane LW;
        if (NET) {
            const size_t plane = (size_t)P.num_envs * (T::K * T::L);
            LW = lstm_load(A.weights, lane);
            row[0] = B.sea_hidden_state + (size_t)(d0 + j) * 8; row[1] = B.sea_cell_state + (size_t)(d0 + j) * 8;     // h0, c0, h1, c1 (anymal.py:65-69)
            row[2] = B.sea_hidden_state + (plane + d0 + j) * 8; row[3] = B.sea_cell_state + (plane + d0 + j) * 8;
#pragma unroll
            for (int a = 0; a < 4; a++) {
                float4 lo = reinterpret_cast<const float4 *>(row[a])[0], hi = reinterpret_cast<const float4 *>(row[a])[1];
                float u[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                lstm_split(u, part[a][0], part[a][1]);
            }
        }
        if (NET || OFF) {
            for (int it = 0; it < P.decimation; it++) {
                __syncthreads();                                   // this sub-step has started: actuator inputs are in lds_x
                float tau_j = 0.0f;
                if (NET) {
                    const float2 x = lds_x[j][lane];
                    tau_j = actuator_step_mfma(LW, x.x, x.y, st);
                    lds_tau[j][lane] = tau_j;
                }
                if (OFF) {                                         // the bodies' frames arrive from the rigid-body wave (long before, normally)
                    volatile int *flag = &sh.fk_ready;
                    // bounded: ~0.3 s at most, then go on with whatever is in LDS (wrong numbers beat a hung CU; the rigid-body
                    // wave publishes the flag unconditionally every sub-step, so the bound is never reached in a correct run)
                    int spin = 0;
                    for (; *flag < sub0 + it + 1 && spin < A.spin_limit; spin++) __builtin_amdgcn_s_sleep(1);
                    missed |= spin >= A.spin_limit;                // reported once, at the end of the step (nothing but a compare inside the loop)
#pragma unroll 1
                    for (int b = j; b < T::L; b += NW - 1) {
                        const float4 f0 = sh.fk[b][0][lane], f1 = sh.fk[b][1][lane], f2 = sh.fk[b][2][lane], f3 = sh.fk[b][3][lane];
                        M3 R; R.m[0] = f0.x; R.m[1] = f0.y; R.m[2] = f0.z; R.m[3] = f0.w; R.m[4] = f1.x; R.m[5] = f1.y; R.m[6] = f1.z; R.m[7] = f1.w; R.m[8] = f2.x;
                        const float *tb = tab + b * LG_JS;
                        float Il[6];
#pragma unroll
                        for (int i = 0; i < 6; i++) Il[i] = tb[J_INERTIA + i];
                        AI I0; S6 p0;
                        body_terms(v3(P.gravity[0], P.gravity[1], P.gravity[2]), tb[J_MASS], v3(tb[J_COM], tb[J_COM + 1], tb[J_COM + 2]), Il, R,
                                   v3(f2.y, f2.z, f2.w), v3(f3.x, f3.y, f3.z), I0, p0);
                        bt_store(sh.bt[b], lane, I0, p0);
                    }
                }
                __syncthreads();                                   // torques / body terms published
                if constexpr (SC && NW == LG_STEP_WAVES) {         // self-collision detection in the shadow of the rigid-body wave's first pass:
                    self_detect<T>(A, lds_tab, lane, wave, *sc);   // this wave's partner limb for all 64 lanes ...
                    if (wave == LG_STEP_WAVES - 1) self_detect<T>(A, lds_tab, lane, 0, *sc);      // ... and (the diagonal partner's wave) the base
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0 && A.debug_skip != 2) sc->ready[wave] = sub0 + it + 1;
                }
            }
        }
        // this wave's share of the step's uniforms: after its last hand-over, while the rigid-body wave is still in the last sub-step's
        // passes (kept out of the sub-step loop -- its code would sit in the instruction cache next to the loop's -- and out of the
        // prologue, where it delayed the first torques); published by P1, every reader is behind P1
        reset_rand->draw(P, e, k, step, lane, wave - 1, NW - 1);
        if (P.measure_heights && P.add_noise) {                    // ... and the rigid-body wave's height-noise blocks (its chunks: sub = k)
            const int nchunk = (P.num_height_points + 3) >> 2;
#pragma unroll
            for (int b = 0; b < HeightCrew<T, NW>::NCH; b++) {
                const int c = k + b * HeightCrew<T, NW>::NV;
                if (b % (NW > 1 ? NW - 1 : 1) == wave - 1 && c < nchunk) {
                    float r[4];
                    rand4(P.seed, e, step, RNG_NOISE_H, c, r);
                    hnoise[b][lane] = make_float4(r[0], r[1], r[2], r[3]);
                }
            }
        }
        if (NET) {
            // The actuator state is final after the last sub-step's evaluation: written back HERE, while the rigid-body wave is still in
            // that sub-step's passes, instead of after P3 where the conversion + eight stores per lane were the tail of every workgroup
            // (the rigid-body wave waited ~1 us at the last barrier).  Stores of one lane to one address stay in order, so the zeroing of
            // reset envs after P3 lands on top.
#pragma unroll
            for (int a = 0; a < 4; a++) {                          // unit-split -> per-row 8-vectors
                float u[8];
                lstm_unsplit(part[a][0], part[a][1], u);           // exchanges between lanes: every lane takes part, only live ones store
                if (live) {
                    reinterpret_cast<float4 *>(row[a])[0] = make_float4(u[0], u[1], u[2], u[3]);
                    reinterpret_cast<float4 *>(row[a])[1] = make_float4(u[4], u[5], u[6], u[7]);
                }
            }
        }
        __syncthreads();                                           // P1: final poses published
        HeightCrew<T, NW> hc;
        float hs = 0.0f;
        if (P.measure_heights) hs = hc.template sample<HF>(A, e, k + T::K * wave, live, sh.pose[lane][0], sh.pose[lane][1], sh.pose[lane][2], sh.pose[lane][3], sh.pose[lane][4]);
        sh.hsum[wave][lane] = hs;
        __syncthreads();                                           // P2: partial height sums published
        const bool keeper = wave == 1 && live && k == 0;           // one lane per env keeps the episode sums
        EpisodeSums es;
        if (keeper) es.load(A, e);
        float un[HeightCrew<T, NW>::NCH][4];                       // this thread's height-noise uniforms, drawn while it waits for P3
        if (P.measure_heights) hc.draw_noise(P, e, k + T::K * wave, step, un);
        __syncthreads();                                           // P3: reset flags / post-reset root z / reward terms published
        if (wave == 1) {
            es.template update<T::K>(A, e, lane, sh, keeper, roll_accum ? roll_accum : accum_slot(A, step));
            if (roll_accum) { /* rollout kernel: no per-step ticket, no step counter -- roll_finish, behind the workgroup's last step */ }
            else {
            if (P.terrain_curriculum && A.B.terrain_levels) {      // (wave-uniform) this workgroup's share of the mean terrain level: the reset
                // lanes' new levels were stored and drained by the rigid-body wave before P3; read past this CU's L1, which may hold the old line
                int lv = keeper ? __hip_atomic_load(A.B.terrain_levels + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) lv += __shfl_xor(lv, o);
                if (lane == 0) __hip_atomic_store(level_parts_slot(A, step, gridDim.x) + blockIdx.x, lv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // Workgroup ticket for the extras finisher, taken HERE: what the last workgroup's finisher reads are the episode-sum
            // atomics just issued by this wave (and terrain levels, drained by the rigid-body wave before P3), so the ticket only
            // has to follow their completion -- both round trips (drain, ticket) overlap the rigid-body wave's observations and
            // state write-back instead of standing at the end of the kernel.
            if (A.defer) {
                // deferred extras: the ticket's only job left is the device step counter -- whoever takes the LAST one knows that every workgroup
                // has read the counter and may advance it.  Nothing to drain first, and no finisher behind it.  (Taken here and not in the
                // prologue: an atomic with a return value in front of this wave's first loads delayed the first torques of every workgroup.)
                if (lane == 0 && atomicAdd(A.done_counter, 1u) == gridDim.x - 1) {
                    if (B.step_counter) B.step_counter[0] = step;
                    __hip_atomic_store(A.done_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                const unsigned int ticket = atomicAdd(A.done_counter, 1u);
                const int last = ticket == gridDim.x - 1;
                if (last) __hip_atomic_store(A.done_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // idle until the next launch
                *s_last = last;
            }
            }
            }
        }
        if (P.measure_heights) hc.write_obs(A, e, k + T::K * wave, live, step, sh.root_z[lane], un);
        if (missed && lane == 0) lg_report(A.status, LG_STATUS_FRAME_HANDOVER_TIMEOUT);
        if (NET) {                                                 // reset envs: actuator state zeroed (anymal.py:59-60), over the early write-back above
            const bool reset = sh.rst[lane] != 0 && live;
            if (__ballot(reset) != 0 && reset) {
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    reinterpret_cast<float4 *>(row[a])[0] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    reinterpret_cast<float4 *>(row[a])[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
            }
        }
    }
};

// POL: the flat actor (48-128-64-32) runs first on the same four waves (lg_policy.h) and hands the sampled actions over
// in LDS -- one launch per rollout step instead of policy kernel + step kernel.
// NW = waves per workgroup (1 rigid-body wave + NW-1 helpers).  Every wave of the kernel gets the rigid-body wave's 512
// registers, so a CU holds NW waves = one workgroup at a time: the host picks the largest NW in {4, 2, 1} that still fits
// all workgroups on the chip in one round (256 workgroups -> 4; Cassie at 8192 envs = 512 workgroups -> 2).
template <bool SC, class T> struct SelfStore { char unused; LG_DEV SelfLds<T> *get() { return nullptr; } };
template <class T> struct SelfStore<true, T> { SelfLds<T> lds; LG_DEV SelfLds<T> *get() { return &lds; } };

// SC: self-collision between the robot's own links (asset.self_collisions = 0; compiled for the quadruped layouts)
template <class T, bool NET, bool HF, bool POL = false, int NW = LG_STEP_WAVES, bool SC = false, bool ROLL = false>
__global__ void __launch_bounds__(NW * LG_BLOCK) k_step(const KArgs A) {
    static_assert(!NET || (NW == LG_STEP_WAVES && 1 + T::L == NW), "one actuator wave per joint of the limb");
    static_assert(!ROLL || (POL && !HF), "the multi-step rollout kernel is the fused-actor kernel on the plane");
    static_assert(!POL || (NET && NW == LG_POLICY_WAVES && T::K * T::L <= 16), "fused policy needs the four-wave actuator-net kernel");
    constexpr int K = T::K, L = T::L, ND = K * L, NREP = T::NREP;
    const lg_params &P = A.P;
    const lg_buffers &B = A.B;
    __shared__ float lds_tab[T::K * Tab<T>::STRIDE];
    __shared__ float2 lds_x[NET ? L : 1][LG_BLOCK];             // actuator inputs (pos_err, vel) of the sub-step, [joint][lane]
    __shared__ float lds_tau[NET ? L : 1][LG_BLOCK];            // actuator torques of the sub-step
    constexpr bool OFF = NW >= 2;
    __shared__ StepSharedT<OFF, L> sh;
    __shared__ SelfStore<SC, T> sc_store;
    __shared__ ResetRand<T> reset_rand;                         // NW > 1: the step's state-independent uniforms for every lane, drawn by the helper waves
    __shared__ float4 hnoise[NW > 1 ? HeightCrew<T, NW>::NCH : 1][LG_BLOCK];     // ... and this wave's height-noise blocks
    __shared__ int s_last;
    __shared__ float4 pol_xa[POL ? 4 : 1][64], pol_xb[POL ? 8 : 1][64], pol_xy[1][64];
    __shared__ float lds_act[POL ? 16 : 1][16];                 // sampled actions [action][env in block]
    __shared__ float lds_obs[ROLL ? 16 : 1][48];                // rollout kernel: the observations a step leaves for the next step's actor
    LG_PROF_BEGIN();
    if (threadIdx.x == 0) sh.fk_ready = 0;                      // published before the first use by stage_limb_table's barrier
    if constexpr (SC) {
        if (threadIdx.x < LG_STEP_WAVES) sc_store.get()->ready[threadIdx.x] = 0;
        stage_base_points<T>(A, sc_store.get());
    }
    stage_limb_table<T>(A, lds_tab);

    const int N = P.num_envs;
    const int wave = threadIdx.x / LG_BLOCK, lane = threadIdx.x % LG_BLOCK;
    const int tid = blockIdx.x * LG_BLOCK + lane;
    int e = tid / K;
    const int k = tid % K;
    const bool live = e < N;
    if (!live) e = N - 1;
    const float *tab = lds_tab + k * Tab<T>::STRIDE;
    const int d0 = e * ND + k * L;                                // first dof of this lane
    const int64_t step0 = A.step >= 0 ? A.step : B.step_counter[0] + 1;   // -1: self-advancing (HIP-graph replay)
    // ROLL: `roll.steps` policy steps in this launch.  Every step is the single-step kernel's body on the same state buffers; the
    // workgroup's own stores are ordered before the next step's loads by the barrier that ends the step (workgroup-scope release /
    // acquire; all of an env's data stays inside its workgroup), per-step outputs go to the [t] slice of the rollout storage.
    const int e_wg = e, k_wg = k;
    auto one_step = [&](const int rt) {
    // ROLL: the loop must not become one giant live range -- without this, loop-invariant code motion hoists the limb-table reads, the
    // per-env constants and the actuator weights of EVERY section above the loop and spills ~260 registers per lane.  Laundering the
    // lane's (env, limb) indices through an empty asm makes every address of the step depend on a per-iteration value.
    // (the thread index itself: the actor's per-lane weight addresses were otherwise computed once above the loop, spilled, and reloaded
    // from scratch one by one in front of the loads that need them -- ~3 us per step)
    int tix = threadIdx.x;
    if constexpr (ROLL) asm volatile("" : "+v"(tix));
    const int wave = tix / LG_BLOCK, lane = tix % LG_BLOCK;
    int e = e_wg;
    const int k = ROLL ? (blockIdx.x * LG_BLOCK + lane) % K : k_wg;
    if constexpr (ROLL) { e = (blockIdx.x * LG_BLOCK + lane) / K; if (e >= N) e = N - 1; }
    const float *tab = lds_tab + k * Tab<T>::STRIDE;
    int d0 = e * ND + k * L;
    const int64_t step = step0 + rt;
    const int sub0 = ROLL ? rt * P.decimation : 0;
    float *const roll_accum = ROLL ? A.roll.extras + (size_t)rt * (LG_NUM_REWARD_TERMS + 2) : nullptr;
    float *const roll_obs_out = ROLL ? A.roll.obs + (size_t)(rt + 1) * N * P.num_obs : nullptr;

    if (POL) {
        if (ROLL) {
            PolicyArgs pa = A.pol;
            pa.obs = A.roll.obs + (size_t)rt * N * P.num_obs;
            if (rt == 0 && A.roll.obs0) {                      // (kernel-uniform) first step of a segment that continues the previous one: no copy kernel in front
                pa.obs = A.roll.obs0;
                const size_t row0 = (size_t)blockIdx.x * 16 * P.num_obs;
                const int n = min(16, N - (int)blockIdx.x * 16) * P.num_obs;
                for (int i = tix; i < n; i += NW * LG_BLOCK) A.roll.obs[row0 + i] = A.roll.obs0[row0 + i];
            }
            pa.actions = A.roll.actions + (size_t)rt * N * ND;
            pa.mean = A.roll.mean ? A.roll.mean + (size_t)rt * N * ND : nullptr;
            policy_forward<3, 8, 4, 2>(pa, pol_xa, pol_xb, pol_xy, blockIdx.x, wave, lane, step, lds_act, rt > 0 ? lds_obs : nullptr);
        } else policy_forward<3, 8, 4, 2>(A.pol, pol_xa, pol_xb, pol_xy, blockIdx.x, wave, lane, step, lds_act);
        __syncthreads();
    }
    if (wave > 0) {
        HelperWave<T, NET, HF, NW, SC>::run(A, wave, lane, e, k, d0, live, step, tab, lds_x, lds_tau, sh, sc_store.get(), lds_tab, &reset_rand, &s_last, hnoise, sub0, roll_accum);
    } else {
    // ---- load persistent state (read once per env-step)
    float root[13], q[L], qd[L], act[L], tau[L];
#pragma unroll
    for (int i = 0; i < 13; i++) root[i] = B.root_states[(size_t)e * 13 + i];
#pragma unroll
    for (int j = 0; j < L; j++) {
        float2 s = reinterpret_cast<const float2 *>(B.dof_state)[d0 + j];
        q[j] = s.x; qd[j] = s.y;
        float a = POL ? lds_act[k * L + j][lane / K] : A.actions_in[d0 + j];
        act[j] = fminf(fmaxf(a, -P.clip_actions), P.clip_actions);       // :86-87
        tau[j] = 0.0f;
    }
    const float mu = 0.5f * ((B.friction_coeffs ? B.friction_coeffs[e] : 1.0f) + P.ground_friction);
    const float base_mass = A.base.mass + (B.base_mass_delta ? B.base_mass_delta[e] : 0.0f);
    float last_qd[L];                              // (long chains read it behind the sub-steps instead: L registers less across the physics)
    if (L <= 3) {
#pragma unroll
        for (int j = 0; j < L; j++) last_qd[j] = B.last_dof_vel[d0 + j];
    }

    // ---- decimation x (torque -> physics)   legged_robot.py:90-96
    float Frep[NREP][3], Fbase[3];
    {   // last step's net contact forces: inputs of the friction estimate (and of the post-physics-only test mode)
        const float *cf = B.contact_forces + ((size_t)e * (1 + K * NREP) + 1 + k * NREP) * 3;
#pragma unroll
        for (int r = 0; r < NREP; r++) { Frep[r][0] = cf[3 * r]; Frep[r][1] = cf[3 * r + 1]; Frep[r][2] = cf[3 * r + 2]; }
        const float *c0 = B.contact_forces + (size_t)e * (1 + K * NREP) * 3;
        Fbase[0] = c0[0]; Fbase[1] = c0[1]; Fbase[2] = c0[2];
    }
    if (P.decimation == 0) {      // post-physics only (parity tests): torques are an input
#pragma unroll
        for (int j = 0; j < L; j++) tau[j] = B.torques[d0 + j];
    }
    LG_PROF(PF_PROLOGUE);
#pragma unroll 1
    for (int it = 0; it < P.decimation; it++) {
        if (!NET) pd_torques<L>(P, tab, act, q, qd, B.last_dof_vel + d0, tau);
        if (NET || OFF) {
            if (NET) {
#pragma unroll
                for (int j = 0; j < L; j++) lds_x[j][lane] = make_float2(act[j] * P.action_scale + tab[j * LG_JS + J_Q0] - q[j], qd[j]);   // anymal.py:73-75
            }
            __syncthreads();                                   // helper waves start on this sub-step
            auto join = [&]() {
                __syncthreads();                               // torques / body terms published
                if (NET) {
#pragma unroll
                    for (int j = 0; j < L; j++) tau[j] = lds_tau[j][lane];
                }
            };
            physics_substep<T, HF, decltype(join), OFF, SC>(A, tab, k, root, q, qd, tau, base_mass, mu, Frep, Fbase, join, sh.bt, sh.fk, &sh.fk_ready, sub0 + it + 1,
                                                            sc_store.get(), it == P.decimation - 1, SC && NW == LG_STEP_WAVES);
        } else {
            physics_substep<T, HF, NoWait, false, SC>(A, tab, k, root, q, qd, tau, base_mass, mu, Frep, Fbase, NoWait(), nullptr, nullptr, nullptr, 0,
                                                      sc_store.get(), it == P.decimation - 1);
        }
    }

    // =====================  post_physics_step  (legged_robot.py:106-137)  =====================
    if (L > 3) {      // long chains: the per-env addresses of everything below are formed here, not above the sub-steps and held (spilled) across them
        asm volatile("" : "+v"(e));
        d0 = e * ND + k * L;
    }
    int64_t ep_len = B.episode_length_buf[e] + 1;                                   // :114
    V3 blv = quat_rotate_inverse(root + 3, v3(root[7], root[8], root[9]));          // :118-121
    V3 bav = quat_rotate_inverse(root + 3, v3(root[10], root[11], root[12]));
    V3 pg = quat_rotate_inverse(root + 3, v3(0, 0, -1));
    float cmd[4];
#pragma unroll
    for (int i = 0; i < 4; i++) cmd[i] = B.commands[(size_t)e * 4 + i];
    // Height-field builds of the short chains: everything else the reward / reset block reads from memory is requested here, in front of the
    // two barriers of the height crew, not at its point of use behind them (a barrier is a fence: the compiler cannot lift a load over it,
    // and each was a bare L2 round trip): 66.0 -> 65.3 us on the same box.  Cassie's kernels have no register to hold them that long
    // (+23 spilled) and the plane kernels get slower (51.7 -> 52.2 us, same box: the allocation shifts): they read them where they are used.
    constexpr bool EARLY_POST = L <= 3 && HF;
    float last_act[L], fat, origin[3];
    uint8_t lc;
    if (EARLY_POST) {
#pragma unroll
        for (int j = 0; j < L; j++) last_act[j] = B.last_actions[d0 + j];
        fat = B.feet_air_time[(size_t)e * K + k];
        lc = B.last_contacts[(size_t)e * K + k];
#pragma unroll
        for (int i = 0; i < 3; i++) origin[i] = B.env_origins[(size_t)e * 3 + i];
    }
    // _get_heights :831-869, by all four waves (HeightCrew); this wave is virtual lane k of its env
    sh.pose[lane][0] = root[0]; sh.pose[lane][1] = root[1]; sh.pose[lane][2] = root[2]; sh.pose[lane][3] = root[5]; sh.pose[lane][4] = root[6];
    __syncthreads();                                               // P1 (also publishes the helpers' uniforms)
    // _post_physics_step_callback :329-345
    if (ep_len % P.resample_interval == 0) {
        if (NW > 1) { float u[4]; reset_rand.get(ResetRand<T>::CMD_STEP, lane, u); resample_commands_u(P, u, cmd); }
        else resample_commands(P, e, step, RNG_CMD_STEP, cmd);
    }
    if (P.heading_command) {
        V3 fwd = quat_apply(root + 3, v3(1, 0, 0));
        float heading = atan2f(fwd.y, fwd.x);
        cmd[2] = fminf(fmaxf(0.5f * wrap_to_pi(cmd[3] - heading), -1.0f), 1.0f);
    }
    HeightCrew<T, NW> hc;
    float hsum = 0.0f;
    if (P.measure_heights) hsum = hc.template sample<HF>(A, e, k, live, root[0], root[1], root[2], root[5], root[6]);
    sh.hsum[0][lane] = hsum;
    __syncthreads();                                               // P2
    if (NW == 4) hsum = (sh.hsum[0][lane] + sh.hsum[1][lane]) + (sh.hsum[2][lane] + sh.hsum[3][lane]);
    else if (NW == 2) hsum = sh.hsum[0][lane] + sh.hsum[1][lane];
    if (P.push_interval > 0 && step % P.push_interval == 0) {                     // _push_robots :438-444
        float u[4];
        if (NW > 1) reset_rand.get(ResetRand<T>::PUSH, lane, u);
        else rand4(P.seed, e, step, RNG_PUSH, 0, u);
        root[7] = urange(-P.max_push_vel, P.max_push_vel, u[0]);
        root[8] = urange(-P.max_push_vel, P.max_push_vel, u[1]);
    }

    // contact_forces (net contact force tensor, last sub-step) -- also the inputs of termination / rewards
    const int rep0 = 1 + k * NREP;
    if (live) {
        float *cf = B.contact_forces + ((size_t)e * (1 + K * NREP) + rep0) * 3;
#pragma unroll
        for (int r = 0; r < NREP; r++) { cf[3 * r] = Frep[r][0]; cf[3 * r + 1] = Frep[r][1]; cf[3 * r + 2] = Frep[r][2]; }
        if (k == 0) { float *c0 = B.contact_forces + (size_t)e * (1 + K * NREP) * 3; c0[0] = Fbase[0]; c0[1] = Fbase[1]; c0[2] = Fbase[2]; }
    }
    LG_PROF(PF_POST_HEIGHTS);
    // check_termination :139-145
    int term_local = 0;
    float coll_local = 0.0f;
#pragma unroll
    for (int r = 0; r < NREP; r++) {
        float n = sqrtf(Frep[r][0] * Frep[r][0] + Frep[r][1] * Frep[r][1] + Frep[r][2] * Frep[r][2]);
        if ((A.termination_mask >> (rep0 + r)) & 1u) term_local |= (n > 1.0f);
        if ((A.penalised_mask >> (rep0 + r)) & 1u) coll_local += (n > 0.1f) ? 1.0f : 0.0f;
    }
    float nbase = sqrtf(Fbase[0] * Fbase[0] + Fbase[1] * Fbase[1] + Fbase[2] * Fbase[2]);
    int contact_term = group_or<K>(term_local);
    if (A.termination_mask & 1u) contact_term |= (nbase > 1.0f);
    float coll = group_sum<K>(coll_local);
    if (A.penalised_mask & 1u) coll += (nbase > 0.1f) ? 1.0f : 0.0f;
    const bool time_out = ep_len > P.max_episode_length;
    int bad = 0;                                    // safety net: a non-finite state ends the episode
#pragma unroll
    for (int i = 0; i < 13; i++) bad |= !isfinite(root[i]);
#pragma unroll
    for (int j = 0; j < L; j++) bad |= !isfinite(q[j]) || !isfinite(qd[j]);
    bad = group_or<K>(bad);
    const bool reset = contact_term || time_out || bad;

    // compute_reward :193-210 ; terms :872-969, cassie.py:43-46
    if (!EARLY_POST) {
#pragma unroll
        for (int j = 0; j < L; j++) { last_act[j] = B.last_actions[d0 + j]; if (L > 3) last_qd[j] = B.last_dof_vel[d0 + j]; }
    }
    float s_ar = 0, s_acc = 0, s_lim = 0, s_dv = 0, s_dvl = 0, s_tl = 0, s_tq = 0, s_ss = 0;
#pragma unroll
    for (int j = 0; j < L; j++) {
        const float *tj = tab + j * LG_JS;
        float da = last_act[j] - act[j]; s_ar += da * da;
        float dd = (last_qd[j] - qd[j]) / P.dt_policy; s_acc += dd * dd;
        float ol = -fminf(q[j] - tj[J_SLO], 0.0f); ol += fmaxf(q[j] - tj[J_SHI], 0.0f); s_lim += ol;
        s_dv += qd[j] * qd[j];
        s_dvl += fminf(fmaxf(fabsf(qd[j]) - tj[J_DVL] * P.soft_dof_vel_limit, 0.0f), 1.0f);
        s_tl += fmaxf(fabsf(tau[j]) - tj[J_TLIM] * P.soft_torque_limit, 0.0f);
        s_tq += tau[j] * tau[j];
        s_ss += fabsf(q[j] - tj[J_Q0]);
    }
    s_ar = group_sum<K>(s_ar); s_acc = group_sum<K>(s_acc); s_lim = group_sum<K>(s_lim); s_dv = group_sum<K>(s_dv);
    s_dvl = group_sum<K>(s_dvl); s_tl = group_sum<K>(s_tl); s_tq = group_sum<K>(s_tq); s_ss = group_sum<K>(s_ss);
    const float *ff = Frep[T::FOOT_REP];
    const float cmd_xy = sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]);
    float fnorm = sqrtf(ff[0] * ff[0] + ff[1] * ff[1] + ff[2] * ff[2]);
    float fcf = group_sum<K>(fmaxf(fnorm - P.max_contact_force, 0.0f));
    int stumble = group_or<K>(sqrtf(ff[0] * ff[0] + ff[1] * ff[1]) > 5.0f * fabsf(ff[2]) ? 1 : 0);
    float nfly = group_sum<K>(ff[2] > 0.1f ? 1.0f : 0.0f);
    float air = 0.0f;
    if (!EARLY_POST) { fat = B.feet_air_time[(size_t)e * K + k]; lc = B.last_contacts[(size_t)e * K + k]; }
    if (P.reward_scale[LG_REW_FEET_AIR_TIME] != 0.0f) {
        bool contact = ff[2] > 1.0f, filt = contact || lc;
        lc = (uint8_t)contact;
        bool first = (fat > 0.0f) && filt;
        fat += P.dt_policy;
        air = (fat - 0.5f) * (first ? 1.0f : 0.0f);
        fat *= filt ? 0.0f : 1.0f;
        air = group_sum<K>(air) * ((cmd_xy > 0.1f) ? 1.0f : 0.0f);
    }
    const float base_h = P.measure_heights ? group_sum<K>(hsum) / (float)P.num_height_points : root[2];
    float term[LG_NUM_REWARD_TERMS];
    term[LG_REW_ACTION_RATE] = s_ar;
    term[LG_REW_ANG_VEL_XY] = bav.x * bav.x + bav.y * bav.y;
    term[LG_REW_BASE_HEIGHT] = (base_h - P.base_height_target) * (base_h - P.base_height_target);
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
        term[LG_REW_TRACKING_LIN_VEL] = __expf(-(ex * ex + ey * ey) / P.tracking_sigma);
        term[LG_REW_TRACKING_ANG_VEL] = __expf(-(ew * ew) / P.tracking_sigma);
    }
    LG_PROF(PF_POST_TERMS);
    float rew = 0.0f;
    const bool writer = live && k == 0;
    float r_t[LG_NUM_REWARD_TERMS];                     // scaled terms; exactly 0 for disabled ones (x + 0 == x: the sum order is the oracle's)
#pragma unroll
    for (int t = 0; t < LG_NUM_REWARD_TERMS; t++) r_t[t] = (P.reward_slot[t] >= 0) ? term[t] * P.reward_scale[t] : 0.0f;
#pragma unroll
    for (int t = 0; t < LG_NUM_REWARD_TERMS; t++) if (t != LG_REW_TERMINATION) rew += r_t[t];
    if (P.only_positive_rewards) rew = fmaxf(rew, 0.0f);
    rew += r_t[LG_REW_TERMINATION];                     // added after the clip (:208-210)
#pragma unroll
    for (int t = 0; t < LG_NUM_REWARD_TERMS; t++) sh.r_t[t][lane] = r_t[t];       // episode sums: helper wave, after P3

    LG_PROF(PF_POST_REWARD);
    // reset_idx for terminated envs (predicated epilogue) :128-129, anymal.py:56-60
    if (!EARLY_POST) { origin[0] = B.env_origins[(size_t)e * 3]; origin[1] = B.env_origins[(size_t)e * 3 + 1]; origin[2] = B.env_origins[(size_t)e * 3 + 2]; }
    if (reset) {
        int level = 0; bool level_changed = false;
        reset_values<T>(A, tab, e, k, step, root, q, qd, cmd, origin, level, level_changed, NW > 1 ? &reset_rand : nullptr, lane);
        if (writer && level_changed) {
            __hip_atomic_store(B.terrain_levels + e, level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // read by the finisher
            B.env_origins[(size_t)e * 3] = origin[0]; B.env_origins[(size_t)e * 3 + 1] = origin[1]; B.env_origins[(size_t)e * 3 + 2] = origin[2];
        }
        fat = 0.0f; ep_len = 0;     // the actuator state of reset envs is zeroed at write-back (anymal.py:59-60)
    }

    sh.rst[lane] = reset ? 1 : 0; sh.root_z[lane] = root[2];
    if (NW > 1 && P.terrain_curriculum) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // terrain_levels stores precede the ticket (helper wave 1, after P3)
    __syncthreads();                                               // P3: helpers write the height observations / actuator state
    if (NW == 1) { EpisodeSums es; const bool keep = live && k == 0; if (keep) es.load(A, e); es.template update<K>(A, e, lane, sh, keep, accum_slot(A, step)); }   // no helper wave: keep the sums here

    LG_PROF(PF_POST_RESET);
#ifdef LG_PROFILE
    { int nr = 0; for (int i = 0; i < LG_BLOCK; i += K) nr += sh.rst[i]; LG_PROF_NOTE(18, (unsigned long long)nr); }
#endif
    // compute_observations :130 (stale base-frame quantities for reset envs, as in the reference)
    write_observations<T>(A, e, k, live, step, root, q, qd, act, tab, blv, bav, pg, cmd, false, NW > 1 ? &reset_rand : nullptr, lane, roll_obs_out, ROLL ? lds_obs[lane / K] : nullptr);
    if (P.measure_heights) {
        if (NW > 1) {
            float un[HeightCrew<T, NW>::NCH][4];
#pragma unroll
            for (int b = 0; b < HeightCrew<T, NW>::NCH; b++) { const float4 w = hnoise[b][lane]; un[b][0] = w.x; un[b][1] = w.y; un[b][2] = w.z; un[b][3] = w.w; }
            hc.write_obs(A, e, k, live, step, root[2], un);
        } else hc.write_obs(A, e, k, live, step, root[2]);
    }

    LG_PROF(PF_POST_OBS);
    // ---- write persistent state back (written once per env-step)
    if (live) {
#pragma unroll
        for (int j = 0; j < L; j++) {
            reinterpret_cast<float2 *>(B.dof_state)[d0 + j] = make_float2(q[j], qd[j]);
            B.actions[d0 + j] = act[j];
            B.torques[d0 + j] = tau[j];
            B.last_actions[d0 + j] = act[j];                      // :132 (after the reset zeroing, as in the reference)
            B.last_dof_vel[d0 + j] = qd[j];                       // :133
        }
        B.feet_air_time[(size_t)e * K + k] = fat;
        B.last_contacts[(size_t)e * K + k] = lc;
        if (k == 0) {
#pragma unroll
            for (int i = 0; i < 13; i++) B.root_states[(size_t)e * 13 + i] = root[i];
#pragma unroll
            for (int i = 0; i < 6; i++) B.last_root_vel[(size_t)e * 6 + i] = root[7 + i];     // :134
#pragma unroll
            for (int i = 0; i < 4; i++) B.commands[(size_t)e * 4 + i] = cmd[i];
            B.base_lin_vel[(size_t)e * 3] = blv.x; B.base_lin_vel[(size_t)e * 3 + 1] = blv.y; B.base_lin_vel[(size_t)e * 3 + 2] = blv.z;
            B.base_ang_vel[(size_t)e * 3] = bav.x; B.base_ang_vel[(size_t)e * 3 + 1] = bav.y; B.base_ang_vel[(size_t)e * 3 + 2] = bav.z;
            B.projected_gravity[(size_t)e * 3] = pg.x; B.projected_gravity[(size_t)e * 3 + 1] = pg.y; B.projected_gravity[(size_t)e * 3 + 2] = pg.z;
            B.rew_buf[e] = rew;
            B.reset_buf[e] = (uint8_t)reset;
            B.time_out_buf[e] = (uint8_t)time_out;
            B.episode_length_buf[e] = ep_len;
            if (ROLL) {
                A.roll.rew[(size_t)rt * N + e] = rew;
                A.roll.done[(size_t)rt * N + e] = (uint8_t)reset;
                A.roll.time_outs[(size_t)rt * N + e] = (uint8_t)time_out;
            }
        }
    }
    // ---- the workgroup that finishes last turns the accumulated sums into extras["episode"] (was a second launch).
    // Its inputs are device-scope atomics (performed at the memory side: no cache write-back / invalidate is needed, and an
    // agent-scope release fence per workgroup measured +6 us); the ticket is taken after this wave's own memory operations
    // have drained (s_waitcnt vmcnt(0)), and the finisher reads with device-scope (L1-bypassing) loads.
    }   // physics wave
    };  // one_step
    if constexpr (ROLL) {
#ifdef LG_PROFILE
        if (threadIdx.x == 0 && A.prof && blockIdx.x < LG_NPROF_BLOCKS) A.prof[LG_NPROF + LG_NPROF_BLOCKS * 40 + blockIdx.x] = wall_clock64();
#endif
#pragma unroll 1
        for (int rt = 0; rt < A.roll.steps; rt++) {
            one_step(rt);
            // Step boundary inside the launch.  What the next step reads of this step's GLOBAL stores it reads in the wave that stored it
            // (state, actuator rows, episode sums: same wave instruction stream, in order) -- the one cross-wave item, the observations the
            // actor starts from, goes through lds_obs.  So the boundary is an LDS release + barrier; the stores need not be drained (a full
            // __syncthreads() with its vmcnt(0) cost every workgroup ~2 us per step).
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef LG_PROFILE
            if (threadIdx.x == 0 && A.prof && blockIdx.x < LG_NPROF_BLOCKS && rt + 1 < LG_NPROF_ROLL)
                A.prof[LG_NPROF + LG_NPROF_BLOCKS * 40 + (size_t)(rt + 1) * LG_NPROF_BLOCKS + blockIdx.x] = wall_clock64();
#endif
        }
        // the segment's finisher: the workgroup that takes the last ticket (behind its waves' drained memory operations)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned int ticket = atomicAdd(A.done_counter, 1u);
            s_last = (ticket == gridDim.x - 1);
            if (s_last) __hip_atomic_store(A.done_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (s_last && threadIdx.x < LG_BLOCK) roll_finish(A, threadIdx.x, step0);
        LG_PROF_END(PF_EXTRAS, A.prof);
        return;
    } else one_step(0);
    const int64_t step = step0;
    LG_PROF(PF_POST);
    if (NW == 1 && !A.defer) {                                     // no helper wave: ticket at the end, behind this wave's own memory operations
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) {
            unsigned int ticket = atomicAdd(A.done_counter, 1u);
            s_last = (ticket == gridDim.x - 1);
            if (s_last) __hip_atomic_store(A.done_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // idle until the next launch
        }
    }
    if (A.defer) {                                                 // (kernel-uniform)
        if (NW == 1 && threadIdx.x == 0 && atomicAdd(A.done_counter, 1u) == gridDim.x - 1) {      // (helper-wave kernels: HelperWave::run) every workgroup has read the step counter
            if (B.step_counter) B.step_counter[0] = step;
            __hip_atomic_store(A.done_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // the PREVIOUS step's accumulators (complete: that launch has ended) become episode_means here, next to 255 busy workgroups
        if (blockIdx.x == 0) finish_extras(A, threadIdx.x, step, false, accum_slot(A, step - 1), NW > 1 ? level_parts_slot(A, step - 1, gridDim.x) : nullptr, gridDim.x);
        LG_PROF_END(PF_EXTRAS, A.prof);
        return;
    }
    __syncthreads();                                               // s_last published (NW > 1: helper wave 1 took the ticket after P3)
    if (s_last) finish_extras(A, threadIdx.x, step, true, A.B.extras_accum, NW > 1 ? level_parts_slot(A, step, gridDim.x) : nullptr, gridDim.x);
    LG_PROF_END(PF_EXTRAS, A.prof);
}

// ------------------------------------------------------------------ reset_idx on an id list (base_task.py:114-118)
template <class T, bool NET>
__global__ void __launch_bounds__(LG_BLOCK) k_reset(const KArgs A) {
    constexpr int K = T::K, L = T::L, ND = K * L;
    const lg_params &P = A.P;
    const lg_buffers &B = A.B;
    __shared__ float lds_tab[T::K * Tab<T>::STRIDE];
    stage_limb_table<T>(A, lds_tab);
    const int tid = blockIdx.x * LG_BLOCK + threadIdx.x;
    const int idx = tid / K, k = tid % K;
    if (idx >= A.count) return;
    const int e = A.env_ids[idx];
    const int N = P.num_envs;
    const float *tab = lds_tab + k * Tab<T>::STRIDE;
    const int d0 = e * ND + k * L;
    float root[13], q[L], qd[L], cmd[4], origin[3];
#pragma unroll
    for (int i = 0; i < 13; i++) root[i] = B.root_states[(size_t)e * 13 + i];
#pragma unroll
    for (int i = 0; i < 4; i++) cmd[i] = B.commands[(size_t)e * 4 + i];
#pragma unroll
    for (int i = 0; i < 3; i++) origin[i] = B.env_origins[(size_t)e * 3 + i];
    int level = 0; bool level_changed = false;
    reset_values<T>(A, tab, e, k, A.step, root, q, qd, cmd, origin, level, level_changed);
#pragma unroll
    for (int j = 0; j < L; j++) {
        reinterpret_cast<float2 *>(B.dof_state)[d0 + j] = make_float2(q[j], qd[j]);
        B.last_actions[d0 + j] = 0.0f;
        B.last_dof_vel[d0 + j] = 0.0f;
    }
    B.feet_air_time[(size_t)e * K + k] = 0.0f;
    if (NET) {
        const size_t plane = (size_t)N * ND;
#pragma unroll
        for (int j = 0; j < L; j++)
#pragma unroll
            for (int u = 0; u < 8; u++) {
                B.sea_hidden_state[(size_t)(d0 + j) * 8 + u] = 0.0f; B.sea_cell_state[(size_t)(d0 + j) * 8 + u] = 0.0f;
                B.sea_hidden_state[(plane + d0 + j) * 8 + u] = 0.0f; B.sea_cell_state[(plane + d0 + j) * 8 + u] = 0.0f;
            }
    }
    if (k == 0) {
#pragma unroll
        for (int i = 0; i < 13; i++) B.root_states[(size_t)e * 13 + i] = root[i];
#pragma unroll
        for (int i = 0; i < 4; i++) B.commands[(size_t)e * 4 + i] = cmd[i];
        if (level_changed) {
            B.terrain_levels[e] = level;
#pragma unroll
            for (int i = 0; i < 3; i++) B.env_origins[(size_t)e * 3 + i] = origin[i];
        }
        B.episode_length_buf[e] = 0;
        B.reset_buf[e] = 1;
        for (int t = 0; t < P.num_reward_slots; t++) {
            atomicAdd(B.extras_accum + t, B.episode_sums[(size_t)t * N + e]);
            B.episode_sums[(size_t)t * N + e] = 0.0f;
        }
        atomicAdd(B.extras_accum + P.num_reward_slots, 1.0f);
    }
}

// ------------------------------------------------------------------ extras["episode"] finisher (legged_robot.py:179-188)
// One wave, launched right behind k_step / k_reset on the same stream: turns the accumulated sums + count into the
// means the reference logs (kept stale when nothing reset, quirk Q4), re-zeroes the accumulator, and refreshes the
// mean terrain level.  Keeps env.step() free of per-step torch kernels and host syncs.
__global__ void __launch_bounds__(64) k_extras(const KArgs A) {        // behind k_reset (and available stand-alone)
    // lg_extras_flush: whichever slot the last deferred step filled (the other one is empty); the level partial sums of that step if there are any
    const int64_t last = A.step >= 0 ? A.step : (A.B.step_counter ? A.B.step_counter[0] : 0);
    finish_extras(A, threadIdx.x, A.step, false, A.B.extras_accum, A.flush_parts > 0 ? level_parts_slot(A, last, A.flush_parts) : nullptr, A.flush_parts);
    if (A.accum_alt) finish_extras(A, threadIdx.x, A.step, false, A.accum_alt, nullptr, 0, false);
}

// ------------------------------------------------------------------ sub-path kernels (parity tests drive these)
__global__ void __launch_bounds__(64) k_actuator(const float *table, const float *pos_err, const float *vel, float *torques,
                                                  float *hidden, float *cell, int rows) {
    const int lane = threadIdx.x;
    int r = blockIdx.x * 64 + lane;
    const bool live = r < rows;
    if (!live) r = rows - 1;                      // every lane takes part in the MFMAs / swaps
    const LstmLane W = lstm_load(table, lane);
    float *ptr[4] = {hidden + (size_t)r * 8, cell + (size_t)r * 8, hidden + ((size_t)rows + r) * 8, cell + ((size_t)rows + r) * 8};
    LstmSplit st;
    float (*arr[4])[4] = {st.h0, st.c0, st.h1, st.c1};
#pragma unroll
    for (int a = 0; a < 4; a++) {
        float u[8];
#pragma unroll
        for (int i = 0; i < 8; i++) u[i] = ptr[a][i];
        lstm_split(u, arr[a][0], arr[a][1]);
    }
    float t = actuator_step_mfma(W, pos_err[r], vel[r], st);
#pragma unroll
    for (int a = 0; a < 4; a++) {
        float u[8];
        lstm_unsplit(arr[a][0], arr[a][1], u);
        if (live) {
#pragma unroll
            for (int i = 0; i < 8; i++) ptr[a][i] = u[i];
        }
    }
    if (live) torques[r] = t;
}

template <class T, bool HF, bool SC = false>
__global__ void __launch_bounds__(LG_BLOCK) k_physics(const KArgs A, const float *torques, int write_contacts) {
    constexpr int K = T::K, L = T::L, ND = K * L, NREP = T::NREP;
    const lg_buffers &B = A.B;
    __shared__ float lds_tab[T::K * Tab<T>::STRIDE];
    __shared__ SelfStore<SC, T> sc_store;
    if constexpr (SC) stage_base_points<T>(A, sc_store.get());
    stage_limb_table<T>(A, lds_tab);
    const int N = A.P.num_envs;
    const int tid = blockIdx.x * LG_BLOCK + threadIdx.x;
    int e = tid / K;
    const int k = tid % K;
    const bool live = e < N;
    if (!live) e = N - 1;
    const float *tab = lds_tab + k * Tab<T>::STRIDE;
    const int d0 = e * ND + k * L;
    float root[13], q[L], qd[L], tau[L], Frep[NREP][3], Fbase[3];
#pragma unroll
    for (int i = 0; i < 13; i++) root[i] = B.root_states[(size_t)e * 13 + i];
#pragma unroll
    for (int j = 0; j < L; j++) { q[j] = B.dof_state[2 * (d0 + j)]; qd[j] = B.dof_state[2 * (d0 + j) + 1]; tau[j] = torques[d0 + j]; }
    const float mu = 0.5f * ((B.friction_coeffs ? B.friction_coeffs[e] : 1.0f) + A.P.ground_friction);
    const float base_mass = A.base.mass + (B.base_mass_delta ? B.base_mass_delta[e] : 0.0f);
    {
        const float *cf = B.contact_forces + ((size_t)e * (1 + K * NREP) + 1 + k * NREP) * 3;
#pragma unroll
        for (int r = 0; r < NREP; r++) { Frep[r][0] = cf[3 * r]; Frep[r][1] = cf[3 * r + 1]; Frep[r][2] = cf[3 * r + 2]; }
        const float *c0 = B.contact_forces + (size_t)e * (1 + K * NREP) * 3;
        Fbase[0] = c0[0]; Fbase[1] = c0[1]; Fbase[2] = c0[2];
    }
    write_contacts = 1;           // the net contact forces also seed the next sub-step's friction estimate
    physics_substep<T, HF, NoWait, false, SC>(A, tab, k, root, q, qd, tau, base_mass, mu, Frep, Fbase, NoWait(), nullptr, nullptr, nullptr, 0, sc_store.get(), true);
    if (!live) return;
#pragma unroll
    for (int j = 0; j < L; j++) { B.dof_state[2 * (d0 + j)] = q[j]; B.dof_state[2 * (d0 + j) + 1] = qd[j]; }
    if (write_contacts) {
        float *cf = B.contact_forces + ((size_t)e * (1 + K * NREP) + 1 + k * NREP) * 3;
#pragma unroll
        for (int r = 0; r < NREP; r++) { cf[3 * r] = Frep[r][0]; cf[3 * r + 1] = Frep[r][1]; cf[3 * r + 2] = Frep[r][2]; }
    }
    if (k == 0) {
#pragma unroll
        for (int i = 0; i < 13; i++) B.root_states[(size_t)e * 13 + i] = root[i];
        if (write_contacts) { float *c0 = B.contact_forces + (size_t)e * (1 + K * NREP) * 3; c0[0] = Fbase[0]; c0[1] = Fbase[1]; c0[2] = Fbase[2]; }
    }
}

template <class T>
__global__ void __launch_bounds__(LG_BLOCK) k_obs(const KArgs A) {
    constexpr int K = T::K, L = T::L, ND = K * L;
    const lg_buffers &B = A.B;
    __shared__ float lds_tab[T::K * Tab<T>::STRIDE];
    stage_limb_table<T>(A, lds_tab);
    const int N = A.P.num_envs;
    const int tid = blockIdx.x * LG_BLOCK + threadIdx.x;
    int e = tid / K;
    const int k = tid % K;
    const bool live = e < N;
    if (!live) e = N - 1;
    const float *tab = lds_tab + k * Tab<T>::STRIDE;
    const int d0 = e * ND + k * L;
    float root[13], q[L], qd[L], act[L], cmd[4];
#pragma unroll
    for (int i = 0; i < 13; i++) root[i] = B.root_states[(size_t)e * 13 + i];
#pragma unroll
    for (int j = 0; j < L; j++) { q[j] = B.dof_state[2 * (d0 + j)]; qd[j] = B.dof_state[2 * (d0 + j) + 1]; act[j] = B.actions[d0 + j]; }
#pragma unroll
    for (int i = 0; i < 4; i++) cmd[i] = B.commands[(size_t)e * 4 + i];
    V3 blv = v3(B.base_lin_vel[(size_t)e * 3], B.base_lin_vel[(size_t)e * 3 + 1], B.base_lin_vel[(size_t)e * 3 + 2]);
    V3 bav = v3(B.base_ang_vel[(size_t)e * 3], B.base_ang_vel[(size_t)e * 3 + 1], B.base_ang_vel[(size_t)e * 3 + 2]);
    V3 pg = v3(B.projected_gravity[(size_t)e * 3], B.projected_gravity[(size_t)e * 3 + 1], B.projected_gravity[(size_t)e * 3 + 2]);
    write_observations<T>(A, e, k, live, A.step, root, q, qd, act, tab, blv, bav, pg, cmd, true);
}

// reset_idx calls update_command_curriculum BEFORE _resample_commands (legged_robot.py:159-176): when the host widened the command ranges on a
// curriculum tick, the envs this step reset re-draw their commands from the NEW ranges -- same Philox block as the in-step draw (seed; env,
// step, CMD_RESET), so only the ranges differ -- and the command slots of the step's observations follow (their noise scale is 0, :500).
__global__ void __launch_bounds__(256) k_resample_reset(const KArgs A) {
    const lg_params &P = A.P; const lg_buffers &B = A.B;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= P.num_envs || !B.reset_buf[e]) return;
    float u[4], cmd[4];
#pragma unroll
    for (int i = 0; i < 4; i++) cmd[i] = B.commands[(size_t)e * 4 + i];
    rand4(P.seed, e, A.step, RNG_CMD_RESET, 0, u);
    resample_commands_u(P, u, cmd);
#pragma unroll
    for (int i = 0; i < 4; i++) B.commands[(size_t)e * 4 + i] = cmd[i];
    float *obs = B.obs_buf + (size_t)e * P.num_obs;
    const float c = P.clip_observations;
    obs[9] = fminf(fmaxf(cmd[0] * P.obs_scale_lin_vel, -c), c); obs[10] = fminf(fmaxf(cmd[1] * P.obs_scale_lin_vel, -c), c);
    obs[11] = fminf(fmaxf(cmd[2] * P.obs_scale_ang_vel, -c), c);
}

// ====================================================================  host side: C-ABI  ====================================================================
static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, const char *arg = "") { snprintf(g_err, sizeof g_err, fmt, arg); return code; }
#define HIP_TRY(x) do { hipError_t _e = (x); if (_e != hipSuccess) return fail(-10, "HIP error: %s", hipGetErrorString(_e)); } while (0)

enum RobotKind { ROBOT_ANYMAL = 0, ROBOT_CASSIE = 1 };

struct lg_sim {
    lg_params      P;
    lg_robot_model M;
    lg_buffers     B;
    BaseTab        base;
    RobotKind      kind;
    int            device;
    bool           has_net, bound;
    float         *d_limb_table;
    float         *d_weights;
    unsigned int  *d_done;
    float         *d_accum_alt;      // deferred extras: accumulators of odd steps
    int            defer;            // lg_set_deferred_extras
    unsigned long long *d_prof;
    int            num_cus;
    float         *d_roll_extras;    // lg_rollout_policy: [LG_MAX_ROLL_STEPS][LG_NUM_REWARD_TERMS + 2] per-step episode accumulators
    unsigned int  *h_status;         // host-mapped sticky status word (hipHostMalloc): the kernels OR LG_STATUS_* bits into it
    int            spin_limit, debug_skip;
};

template <class T> static int check_topology(const lg_robot_model *m) {
    if (m->num_limbs != T::K || m->chain_len != T::L) return 0;
    if (m->num_bodies != 1 + T::K * T::NREP || m->num_base_points < 1 || m->num_base_points > T::K) return 0;
    for (int i = 0; i < m->num_base_points; i++) if (m->base_points[i].report_body != 0) return 0;
    for (int k = 0; k < T::K; k++) {
        if (m->num_limb_points[k] != T::NPT) return 0;
        for (int i = 0; i < T::NPT; i++) {
            if (m->limb_points[k][i].joint != T::pt_joint(i)) return 0;
            if (m->limb_points[k][i].report_body != 1 + k * T::NREP + T::pt_rep(i)) return 0;
        }
        if (m->foot_body[k] != 1 + k * T::NREP + T::FOOT_REP) return 0;
    }
    return 1;
}

// 3x3 helpers for the host-side re-framing (row-major doubles)
static void m3_mul(const double *A, const double *B, double *C) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j]; }
static void m3_tr(const double *A, double *B) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) B[3 * i + j] = A[3 * j + i]; }
static void m3_vec(const double *A, const float *v, float *out) { for (int i = 0; i < 3; i++) out[i] = (float)(A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2]); }
// rotation Q with Q e_x = a (a unit)
static void m3_align_x(const float *a, double *Q) {
    double ax = a[0], ay = a[1], az = a[2], n = sqrt(ax * ax + ay * ay + az * az);
    ax /= n; ay /= n; az /= n;
    double bx, by, bz;                              // any unit vector orthogonal to a: cross with the coordinate axis least aligned with it
    if (fabs(ax) <= fabs(ay) && fabs(ax) <= fabs(az)) { bx = 0; by = -az; bz = ay; }
    else if (fabs(ay) <= fabs(az)) { bx = az; by = 0; bz = -ax; }
    else { bx = -ay; by = ax; bz = 0; }
    n = sqrt(bx * bx + by * by + bz * bz); bx /= n; by /= n; bz /= n;
    const double cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
    Q[0] = ax; Q[1] = bx; Q[2] = cx; Q[3] = ay; Q[4] = by; Q[5] = cy; Q[6] = az; Q[7] = bz; Q[8] = cz;
    if (fabs(ax - 1.0) < 1e-12) { for (int i = 0; i < 9; i++) Q[i] = (i % 4 == 0) ? 1.0 : 0.0; }      // already +x: keep the frame
}

template <class T> static void fill_limb_table(const lg_params &P, const lg_robot_model &M, float *t) {
    memset(t, 0, sizeof(float) * T::K * Tab<T>::STRIDE);
    for (int k = 0; k < T::K; k++) {
        float *tk = t + k * Tab<T>::STRIDE;
        // Axis alignment: body j's frame is re-defined as B'_j = B_j Q_j with Q_j e_x = joint axis j, so that the kernel's joint
        // rotation is always about the local +x axis (a 2-column mix instead of a general Rodrigues rotation).  With
        // R_j = R_par Rfix_j Rot(a_j, q) and Rot(a_j, q) Q_j = Q_j Rot(e_x, q):  R'_j = R'_par [Q_par^T Rfix_j Q_j] Rot(e_x, q);
        // joint origins are expressed in the parent's new frame (Q_par^T pos), body-local vectors in the body's (Q_j^T v), the
        // inertia tensor as Q_j^T I Q_j.  World-frame results are unchanged.
        double Q[T::L][9], Qpar[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int j = 0; j < T::L; j++) {
            int d = k * T::L + j;
            float *tj = tk + j * LG_JS;
            m3_align_x(M.joint_axis[d], Q[j]);
            double QparT[9], QjT[9], Rf[9], tmp[9], Rnew[9];
            m3_tr(Qpar, QparT); m3_tr(Q[j], QjT);
            for (int i = 0; i < 9; i++) Rf[i] = M.joint_rot[d][i];
            m3_mul(QparT, Rf, tmp); m3_mul(tmp, Q[j], Rnew);
            m3_vec(QparT, M.joint_pos[d], tj + J_POS);
            for (int i = 0; i < 9; i++) tj[J_ROT + i] = (float)Rnew[i];
            tj[J_AXIS] = 1.0f; tj[J_AXIS + 1] = 0.0f; tj[J_AXIS + 2] = 0.0f;
            tj[J_MASS] = M.body_mass[d];
            m3_vec(QjT, M.body_com[d], tj + J_COM);
            {
                const float *I6 = M.body_inertia[d];
                double If[9] = {I6[0], I6[1], I6[2], I6[1], I6[3], I6[4], I6[2], I6[4], I6[5]}, T1[9], T2[9];
                m3_mul(QjT, If, T1); m3_mul(T1, Q[j], T2);
                tj[J_INERTIA] = (float)T2[0]; tj[J_INERTIA + 1] = (float)T2[1]; tj[J_INERTIA + 2] = (float)T2[2];
                tj[J_INERTIA + 3] = (float)T2[4]; tj[J_INERTIA + 4] = (float)T2[5]; tj[J_INERTIA + 5] = (float)T2[8];
            }
            tj[J_LO] = M.dof_lower[d]; tj[J_HI] = M.dof_upper[d]; tj[J_VLIM] = M.dof_vel_limit[d];
            tj[J_ARM] = M.dof_armature[d]; tj[J_DAMP] = M.dof_damping[d];
            tj[J_KP] = P.p_gains[d]; tj[J_KD] = P.d_gains[d]; tj[J_Q0] = P.default_dof_pos[d]; tj[J_TLIM] = P.torque_limits[d];
            tj[J_SLO] = P.soft_pos_lower[d]; tj[J_SHI] = P.soft_pos_upper[d]; tj[J_DVL] = P.dof_vel_limits[d];
            float sub = 0.0f;
            for (int jj = j; jj < T::L; jj++) sub += M.body_mass[k * T::L + jj];
            tj[J_SUBM] = sub;
            for (int i = 0; i < 9; i++) Qpar[i] = Q[j][i];
        }
        for (int i = 0; i < T::NPT; i++) {
            float *tp = tk + T::L * LG_JS + 4 * i;
            double QjT[9];
            m3_tr(Q[T::pt_joint(i)], QjT);
            m3_vec(QjT, M.limb_points[k][i].pos, tp); tp[3] = M.limb_points[k][i].radius;
        }
        for (int g = 0; g < T::NGRP; g++) {        // bounding sphere of each self-collision point group about the midpoint of its two anchor points
            const float *a = M.limb_points[k][T::grp_c0(g)].pos, *b = M.limb_points[k][T::grp_c1(g)].pos;
            const float c[3] = {0.5f * (a[0] + b[0]), 0.5f * (a[1] + b[1]), 0.5f * (a[2] + b[2])};
            float R = 0.0f;
            for (int cap = 0; cap < T::NCAP; cap++) if (T::cap_grp(cap) == g)
                for (int e = 0; e < 2; e++) {
                    const lg_point &pt = M.limb_points[k][e ? T::cap_p1(cap) : T::cap_p0(cap)];
                    const float dx = pt.pos[0] - c[0], dy = pt.pos[1] - c[1], dz = pt.pos[2] - c[2];
                    R = fmaxf(R, sqrtf(dx * dx + dy * dy + dz * dz) + pt.radius);
                }
            tk[Tab<T>::GRP + g] = R * 1.0001f;
        }
        for (int c = 0; c < 4; c++) tk[Tab<T>::BPT + c] = 0.0f;                  // lane k of an env owns base point k
        if (k < M.num_base_points) { memcpy(tk + Tab<T>::BPT, M.base_points[k].pos, 12); tk[Tab<T>::BPT + 3] = M.base_points[k].radius; }
    }
}

// self-collision shapes: the oracle pairs consecutive points (same body, radius, report body) into capsules at run time; the kernels
// have the pairing compiled in (Traits::cap_p0 / cap_p1).  Accept the model only if both views agree.
template <class T> static int check_capsules(const lg_robot_model *m) {
    auto mergeable = [](const lg_point &a, const lg_point &b) { return a.joint == b.joint && a.radius == b.radius && a.report_body == b.report_body; };
    for (int k = 0; k < T::K; k++) {
        int next = 0;
        for (int c = 0; c < T::NCAP; c++) {
            const int p0 = T::cap_p0(c), p1 = T::cap_p1(c);
            if (p0 != next) return 0;
            if (p1 != p0 && !(p1 == p0 + 1 && mergeable(m->limb_points[k][p0], m->limb_points[k][p1]))) return 0;
            if (p1 == p0 && p0 + 1 < T::NPT && mergeable(m->limb_points[k][p0], m->limb_points[k][p0 + 1])) return 0;
            if (T::pt_joint(p0) != T::pt_joint(p1)) return 0;
            next = p1 + 1;
        }
        if (next != T::NPT) return 0;
    }
    for (int i = 0; i < m->num_base_points; i += 2)        // base: pairs (0,1), (2,3); an odd last point is a sphere
        if (i + 1 < m->num_base_points && m->base_points[i].radius != m->base_points[i + 1].radius) return 0;
    return 1;
}

// Per-lane MFMA operand table of the actuator net (layout: lg_device.h "ANYdrive actuator net").
// `w` is the 972-float blob: in_scale[2], out_scale, Wih0[32][2], Whh0[32][8], bih0, bhh0, Wih1[32][8], Whh1[32][8], bih1, bhh1, lw[8], lb.
static void build_lstm_table(const float *w, float *t /* [LW_ROWS][64] */) {
    const float *in_scale = w, *out_scale = w + 2, *Wih0 = w + 3, *Whh0 = Wih0 + 64, *bih0 = Whh0 + 256, *bhh0 = bih0 + 32;
    const float *Wih1 = bhh0 + 32, *Whh1 = Wih1 + 256, *bih1 = Whh1 + 256, *bhh1 = bih1 + 32, *lw = bhh1 + 32, *lb = lw + 8;
    for (int l = 0; l < 64; l++) {
        const int g = l & 31, hb = l >> 5;
        // gate rows are pre-scaled so lstm_cell can feed v_exp_f32 (= 2^x) directly: i, f, o by -log2(e); g (rows 16-23) by -2 log2(e)
        const double sc = (g >= 16 && g < 24) ? -2.0 * 1.4426950408889634 : -1.4426950408889634;
        auto S = [&](double v) { return (float)(sc * v); };
        t[LW_B0 * 64 + l] = hb ? 0.0f : S((double)bih0[g] + (double)bhh0[g]);
        t[LW_X * 64 + l] = S((double)Wih0[2 * g + hb] * in_scale[hb]);
        t[LW_B1 * 64 + l] = hb ? 0.0f : S((double)bih1[g] + (double)bhh1[g]);
        for (int s = 0; s < 4; s++) {
            t[(LW_H0 + s) * 64 + l] = S(Whh0[8 * g + s + 4 * hb]);
            t[(LW_I1 + s) * 64 + l] = S(Wih1[8 * g + s + 4 * hb]);
            t[(LW_H1 + s) * 64 + l] = S(Whh1[8 * g + s + 4 * hb]);
            t[(LW_LIN + s) * 64 + l] = lw[s + 4 * hb];
        }
        t[LW_LB * 64 + l] = lb[0];
        t[LW_OUT * 64 + l] = out_scale[0];
    }
}

static int upload_tables(lg_sim *s) {
    float host[LG_MAX_LIMBS * (LG_MAX_CHAIN * LG_JS + 4 * LG_MAX_LIMB_POINTS + 1)];
    size_t n;
    if (s->kind == ROBOT_ANYMAL) { fill_limb_table<AnymalTraits>(s->P, s->M, host); n = AnymalTraits::K * Tab<AnymalTraits>::STRIDE; }
    else { fill_limb_table<CassieTraits>(s->P, s->M, host); n = CassieTraits::K * Tab<CassieTraits>::STRIDE; }
    HIP_TRY(hipMemcpy(s->d_limb_table, host, n * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_limb_table + LG_MAX_LIMBS * (LG_MAX_CHAIN * LG_JS + 4 * LG_MAX_LIMB_POINTS + 1), s->P.height_points, sizeof(float) * 2 * LG_MAX_HEIGHT_POINTS, hipMemcpyHostToDevice));
    s->base.mass = s->M.base_mass;
    memcpy(s->base.com, s->M.base_com, 12); memcpy(s->base.inertia, s->M.base_inertia, 24);
    memset(s->base.pts, 0, sizeof s->base.pts);
    s->base.num_pts = s->M.num_base_points;
    s->base.mass_robot = s->M.base_mass;
    for (int d = 0; d < s->M.num_limbs * s->M.chain_len; d++) s->base.mass_robot += s->M.body_mass[d];
    for (int i = 0; i < s->M.num_base_points; i++) { memcpy(s->base.pts[i], s->M.base_points[i].pos, 12); s->base.pts[i][3] = s->M.base_points[i].radius; }
    return 0;
}

static void fill_args(const lg_sim *s, KArgs &a, int64_t step) {
    a.P = s->P; a.B = s->B; a.base = s->base; a.limb_table = s->d_limb_table; a.hpts = s->d_limb_table + LG_MAX_LIMBS * (LG_MAX_CHAIN * LG_JS + 4 * LG_MAX_LIMB_POINTS + 1); a.weights = s->d_weights;
    a.actions_in = nullptr; a.env_ids = nullptr; a.count = 0; a.step = step; memset(&a.pol, 0, sizeof a.pol);
    a.penalised_mask = s->M.penalised_mask; a.termination_mask = s->M.termination_mask; a.done_counter = s->d_done; a.prof = s->d_prof;
    a.accum_alt = s->d_accum_alt; a.defer = s->defer; a.flush_parts = 0;
    a.status = s->h_status; a.spin_limit = s->spin_limit; a.debug_skip = s->debug_skip;
    memset(&a.roll, 0, sizeof a.roll);
}
// Entry check of every call on a handle: a status bit set by an earlier launch is an error from now on (sticky).
static int status_error(const lg_sim *s) {
    const unsigned int st = s->h_status ? __atomic_load_n(s->h_status, __ATOMIC_RELAXED) : 0u;
    if (!st) return 0;
    static thread_local char text[160];
    snprintf(text, sizeof text, "0x%x%s%s", st, (st & LG_STATUS_FRAME_HANDOVER_TIMEOUT) ? " frame hand-over poll timed out" : "",
             (st & LG_STATUS_SELF_COLLISION_TIMEOUT) ? " self-collision hand-over poll timed out" : "");
    return fail(-20, "device status %s: an earlier launch on this handle ran with a missed LDS hand-over -- its physics is invalid (lg_clear_device_status() after re-initialising the state)", text);
}
template <class T> static int grid_for(int n_env_like) { return (n_env_like * T::K + LG_BLOCK - 1) / LG_BLOCK; }

// ------------------------------------------------------------------ fused actor (lg_policy.h): host side
static int g_wide_precision = 1;       /* lg_mlp_wide_set_precision: 0 = f32 MFMA kernels, 1 = split-bf16 (bf16x3) kernels, learner GEMMs and wide actor alike */
struct lg_policy {
    int32_t dims[5];
    int     tiles[4];          // input tiles of layer 0, then hidden widths / 16
    float  *d_w[4], *d_b[4], *d_std;
    int     device;
    // wide actors (hidden 512-256-128): split-bf16 operand stream of k_policy_act_wide next to the f32 one
    bool    wide;
    int     wide_ks[4], wide_ot[4];     // k-steps of 16 / output tiles of 32 per layer
    __bf16 *d_wb[4];
    float  *d_bb[4];
};
// torch Linear [out,in] -> MFMA A-operand stream [out_tile][k_step = (t,r)][lane]: W[16o + (l&15)][16t + 4(l>>4) + r]
static void policy_pack_layer(const float *W, const float *bias, int in_dim, int out_dim, int in_tiles, int out_tiles,
                              float *w_packed, float *b_packed) {
    for (int o = 0; o < out_tiles; o++) {
        for (int t = 0; t < in_tiles; t++) for (int r = 0; r < 4; r++) for (int l = 0; l < 64; l++) {
            int row = 16 * o + (l & 15), col = 16 * t + 4 * (l >> 4) + r;
            w_packed[((size_t)(o * in_tiles + t) * 4 + r) * 64 + l] = (row < out_dim && col < in_dim) ? W[(size_t)row * in_dim + col] : 0.0f;
        }
        for (int r = 0; r < 4; r++) for (int l = 0; l < 64; l++) {       // D layout: lane l, reg r = row 16o + 4(l>>4) + r
            int row = 16 * o + 4 * (l >> 4) + r;
            b_packed[(o * 4 + r) * 64 + l] = row < out_dim ? bias[row] : 0.0f;
        }
    }
}

// device version of policy_pack_layer: one thread per packed element
__global__ void k_policy_pack(const float *__restrict__ W, const float *__restrict__ bias, int in_dim, int out_dim, int in_tiles, int out_tiles,
                              float *__restrict__ w_packed, float *__restrict__ b_packed) {
    const size_t nw = (size_t)out_tiles * in_tiles * 4 * 64, nb = (size_t)out_tiles * 4 * 64;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nw + nb; i += (size_t)gridDim.x * blockDim.x) {
        if (i < nw) {
            const int l = (int)(i & 63), r = (int)((i >> 6) & 3);
            const size_t ot = i >> 8;
            const int t = (int)(ot % in_tiles), o = (int)(ot / in_tiles);
            const int row = 16 * o + (l & 15), col = 16 * t + 4 * (l >> 4) + r;
            w_packed[i] = (row < out_dim && col < in_dim) ? W[(size_t)row * in_dim + col] : 0.0f;
        } else {
            const size_t j = i - nw;
            const int l = (int)(j & 63), r = (int)((j >> 6) & 3), o = (int)(j >> 8);
            const int row = 16 * o + 4 * (l >> 4) + r;
            b_packed[j] = row < out_dim ? bias[row] : 0.0f;
        }
    }
}

// ---- PPO learner: MLP forward / backward for up to two nets (actor, critic) in one launch -------------------------------------
#define LG_TRAIN_WGS 256           /* workgroups per net = partial-sum slices */
static unsigned long long *g_mlp_trace = nullptr;   /* diagnostic, see lg_mlp_trace */
#define LG_FWD_SLOTS 4             /* row tiles in flight per workgroup (forward: 140 KB of LDS) */
#define LG_BWD_SLOTS 2             /* backward: 152 KB */
static int mlp_fill(const lg_mlp_net *nets, int32_t n_nets, const int64_t *rows, int32_t mb, lg::MlpArgs &a, int &wgs, int slots, bool partials) {
    if (!nets || n_nets < 1 || n_nets > 2 || mb <= 0) return fail(-1, "bad argument");
    memset(&a, 0, sizeof a);
    a.rows = rows; a.mb = mb; a.n_tiles = (mb + 15) / 16; a.trace = g_mlp_trace;
    // persistent workgroups, one per CU (the LDS-resident weights allow no more): the CUs are split between the nets
    static int num_cus = 0;
    if (!num_cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || num_cus <= 0)
            num_cus = 256;
    }
    wgs = num_cus / n_nets;
    if (partials && wgs > LG_TRAIN_WGS) wgs = LG_TRAIN_WGS;                       // backward: one partial-sum slice per workgroup
    if (wgs * slots > a.n_tiles) wgs = (a.n_tiles + slots - 1) / slots;
    if (wgs < 1) wgs = 1;
    for (int n = 0; n < n_nets; n++) {
        const lg_mlp_net &s = nets[n];
        lg::MlpNetArgs &d = a.net[n];
        if (!s.input) return fail(-1, "null input");
        if (s.dims[0] <= 0 || s.dims[0] > 48 || s.dims[1] != 128 || s.dims[2] != 64 || s.dims[3] != 32 || s.dims[4] <= 0 || s.dims[4] > 16)
            return fail(-4, "lg_mlp_*: only the <=48-128-64-32-<=16 MLP shape is built");
        int gf = 0;
        for (int l = 0; l < 4; l++) {
            if (!s.weights[l] || !s.biases[l]) return fail(-1, "null layer pointer");
            if (l > 0 && ((uintptr_t)s.weights[l] & 15)) return fail(-4, "weights must be 16-byte aligned");
            d.w[l] = s.weights[l]; d.b[l] = s.biases[l];
            gf += s.dims[l + 1] * s.dims[l] + s.dims[l + 1];
        }
        d.x = s.input; d.y = s.output; d.dy = s.grad_output; d.grad_floats = gf; d.part_stride = gf + LG_PPO_EXTRA;
        memcpy(d.dims, s.dims, sizeof d.dims);
    }
    return 0;
}

extern "C" {

/* diagnostic (tools/mlp_probe.py): later lg_mlp_* launches write up to 60 s_memtime stamps of workgroup (0,0) to `buf` (device, u64[60]); null stops */
void lg_mlp_trace(unsigned long long *buf) { g_mlp_trace = buf; }

int lg_mlp_forward(const lg_mlp_net *nets, int32_t n_nets, const int64_t *rows, int32_t mb, void *stream) {
    lg::MlpArgs a; int wgs;
    if (int rc = mlp_fill(nets, n_nets, rows, mb, a, wgs, LG_FWD_SLOTS, false)) return rc;
    for (int n = 0; n < n_nets; n++) if (!nets[n].output) return fail(-1, "null output");
    constexpr size_t lds_bytes = lg::TrainLds<3, 8, 4, 2, false, LG_FWD_SLOTS>::floats * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void *)lg::k_mlp_train<3, 8, 4, 2, false, LG_FWD_SLOTS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_set = true;
    }
    hipLaunchKernelGGL((lg::k_mlp_train<3, 8, 4, 2, false, LG_FWD_SLOTS>), dim3(wgs, n_nets), dim3(64 * LG_TRAIN_WAVES * LG_FWD_SLOTS), lds_bytes, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

size_t lg_mlp_workspace_bytes(const lg_mlp_net *nets, int32_t n_nets) {
    size_t total = 0;
    if (!nets) return 0;
    for (int n = 0; n < n_nets; n++) {
        size_t gf = 0;
        for (int l = 0; l < 4; l++) gf += (size_t)nets[n].dims[l + 1] * nets[n].dims[l] + nets[n].dims[l + 1];
        total += (gf + LG_PPO_EXTRA) * LG_TRAIN_WGS * sizeof(float);
    }
    return total;
}

// shared by lg_mlp_backward (batch == null: dL/dy from nets[n].grad_output) and lg_ppo_minibatch (dL/dy from the fused PPO loss)
static int mlp_backward_launch(const lg_mlp_net *nets, int32_t n_nets, const int64_t *rows, int32_t mb, const lg_ppo_batch *batch,
                               float *workspace, size_t workspace_bytes, void *stream) {
    lg::MlpArgs a; int wgs;
    if (int rc = mlp_fill(nets, n_nets, rows, mb, a, wgs, LG_BWD_SLOTS, true)) return rc;
    if (!workspace || workspace_bytes < lg_mlp_workspace_bytes(nets, n_nets)) return fail(-1, "workspace too small (lg_mlp_workspace_bytes)");
    lg::MlpReduceArgs r; memset(&r, 0, sizeof r);
    float *ws = workspace;
    int max_gf = 0;
    for (int n = 0; n < n_nets; n++) {
        if (!batch && !nets[n].grad_output) return fail(-1, "null grad_output");
        a.net[n].partial = ws; r.partial[n] = ws; ws += (size_t)a.net[n].part_stride * LG_TRAIN_WGS;
        r.grad_floats[n] = a.net[n].grad_floats; r.part_stride[n] = a.net[n].part_stride;
        if (a.net[n].grad_floats > max_gf) max_gf = a.net[n].grad_floats;
        memcpy(r.dims[n], nets[n].dims, sizeof r.dims[n]);
        for (int l = 0; l < 4; l++) {
            if (!nets[n].grad_weights[l] || !nets[n].grad_biases[l]) return fail(-1, "null gradient pointer");
            r.gw[n][l] = nets[n].grad_weights[l]; r.gb[n][l] = nets[n].grad_biases[l];
        }
    }
    r.n_partials = wgs;                        // the groups of a workgroup fold their sums before writing
    if (batch) {
        if (n_nets != 2 || nets[1].dims[4] != 1) return fail(-1, "lg_ppo_minibatch needs nets = {actor, critic (one output)}");
        if (!rows || !batch->actions || !batch->old_log_prob || !batch->old_mu || !batch->old_sigma || !batch->advantages || !batch->old_values ||
            !batch->returns || !batch->std || !batch->d_std || !batch->stats) return fail(-1, "null PPO batch pointer");
        a.ppo = lg::PpoArgs{batch->actions, batch->old_log_prob, batch->old_mu, batch->old_sigma, batch->advantages, batch->old_values, batch->returns,
                            batch->std, batch->clip, batch->value_coef, 1.0f / (float)mb, batch->use_clipped_value};
        r.loss = 1; r.num_actions = nets[0].dims[4]; r.std = batch->std; r.ecoef = batch->entropy_coef; r.d_std = batch->d_std; r.stats = batch->stats; r.loss_acc = batch->loss_acc;
    }
    hipStream_t st = (hipStream_t)stream;
    constexpr size_t lds_bytes = lg::TrainLds<3, 8, 4, 2, true, LG_BWD_SLOTS>::floats * sizeof(float);
    static_assert(lds_bytes <= 160 * 1024, "k_mlp_train backward exceeds the CU's LDS");
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void *)lg::k_mlp_train<3, 8, 4, 2, true, LG_BWD_SLOTS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        HIP_TRY(hipFuncSetAttribute((const void *)lg::k_mlp_train<3, 8, 4, 2, true, LG_BWD_SLOTS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_set = true;
    }
    if (batch) hipLaunchKernelGGL((lg::k_mlp_train<3, 8, 4, 2, true, LG_BWD_SLOTS, true>), dim3(wgs, n_nets), dim3(64 * LG_TRAIN_WAVES * LG_BWD_SLOTS), lds_bytes, st, a);
    else hipLaunchKernelGGL((lg::k_mlp_train<3, 8, 4, 2, true, LG_BWD_SLOTS, false>), dim3(wgs, n_nets), dim3(64 * LG_TRAIN_WAVES * LG_BWD_SLOTS), lds_bytes, st, a);
    hipLaunchKernelGGL(lg::k_mlp_reduce, dim3((max_gf + (batch ? LG_PPO_EXTRA : 0) + 31) / 32, n_nets), dim3(256), 0, st, r);
    HIP_TRY(hipGetLastError());
    return 0;
}

int lg_mlp_backward(const lg_mlp_net *nets, int32_t n_nets, const int64_t *rows, int32_t mb, float *workspace, size_t workspace_bytes,
                    void *stream) {
    return mlp_backward_launch(nets, n_nets, rows, mb, nullptr, workspace, workspace_bytes, stream);
}

int lg_ppo_minibatch(const lg_mlp_net *nets, const int64_t *rows, int32_t mb, const lg_ppo_batch *batch, float *workspace,
                     size_t workspace_bytes, void *stream) {
    if (!batch) return fail(-1, "null batch");
    return mlp_backward_launch(nets, 2, rows, mb, batch, workspace, workspace_bytes, stream);
}

// ---- learner kernels for the wide MLPs (lg_gemm.h): per layer a tiled f32-MFMA GEMM with the element-wise work in its epilogue
struct WideLayout { size_t x[4], g[4], part[4], x0p, w0p, wpk[4], bpk[4], total; int splits[4]; int kchunk[4]; int k0p; int ks[4], ot[4]; bool chain; int out_chunks; bool out_narrow; };      // float offsets into one net's workspace slice
static void wide_layout(const lg_mlp_net &n, int mb, WideLayout &L) {
    size_t o = 0;
    for (int l = 1; l <= 3; l++) { L.x[l] = o; o += (size_t)mb * n.dims[l]; }
    for (int l = 1; l <= 3; l++) { L.g[l] = o; o += (size_t)mb * n.dims[l]; }
    L.out_narrow = n.dims[3] == 128 && n.dims[4] <= LG_OUT_MAXN;      // k_wide_out_bwd: dX + dW of the output layer in one pass
    L.out_chunks = mb >= 2048 ? 256 : (mb + 7) / 8;
    o = (o + 3) & ~(size_t)3;
    for (int l = 0; l < 4; l++) {                                     // one partial buffer per layer: all four are summed by ONE k_wide_reduce launch
        const int tiles = ((n.dims[l + 1] + LG_GT - 1) / LG_GT) * ((n.dims[l] + LG_GT - 1) / LG_GT);
        int sp = (384 + tiles - 1) / tiles;                                   // enough workgroups for the chip: tiles x splits >= ~1.5 x CUs
        if (sp > LG_WIDE_MAX_SPLITS) sp = LG_WIDE_MAX_SPLITS;
        int chunk = (mb + sp - 1) / sp;
        chunk = ((chunk + LG_BK - 1) / LG_BK) * LG_BK;       /* multiple of both kernels' stage depths */
        if (chunk < LG_BK) chunk = LG_BK;
        sp = (mb + chunk - 1) / chunk;
        L.splits[l] = sp; L.kchunk[l] = chunk;
        size_t p = (size_t)sp * n.dims[l + 1] * ((n.dims[l] + 1 + 3) & ~3);
        if (l == 3 && L.out_narrow) { const size_t q = (size_t)L.out_chunks * n.dims[4] * ((n.dims[3] + 1 + 3) & ~3); if (q > p) p = q; }
        L.part[l] = o; o += (p + 3) & ~(size_t)3;
    }
    L.k0p = (n.dims[0] + 3) & ~3;                     // aligned, gather-free copies of the layer-0 operands (k_wide_prep)
    L.x0p = o; o += (size_t)mb * L.k0p;
    L.w0p = o; o += (size_t)n.dims[1] * L.k0p;
    // chain forward (k_mlp_chain_fwd64): split-bf16 operand streams of the four layers, re-packed every call
    const int k0s = (n.dims[0] + 15) / 16;
    L.chain = n.dims[1] == 512 && n.dims[2] == 256 && n.dims[3] == 128 && n.dims[4] <= 16 && (k0s == 15 || k0s == 11);
    for (int l = 0; l < 4; l++) {
        L.ks[l] = l == 0 ? k0s : n.dims[l] / 16; L.ot[l] = (n.dims[l + 1] + 31) / 32;
        o = (o + 3) & ~(size_t)3;
        L.wpk[l] = o; if (L.chain) o += (size_t)L.ot[l] * L.ks[l] * 512;          // 1024 bf16 per (tile, k-step)
        L.bpk[l] = o; if (L.chain) o += (size_t)L.ot[l] * 32;
    }
    L.total = (o + 3) & ~(size_t)3;
}
// The gathered, padded copy of net n's input rows (k_wide_prep).  Actor and critic of the registered tasks read the SAME observation
// tensor (no privileged observations): one copy then serves both nets -- half the gather traffic, and layer 0's dW / the chain forward
// of the second net find the rows in cache.
static bool wide_shared_input(const lg_mlp_net *nets, int32_t n_nets) {
    return n_nets == 2 && nets[0].input == nets[1].input && nets[0].dims[0] == nets[1].dims[0];
}
static float *wide_x0p(const lg_mlp_net *nets, int32_t n_nets, int32_t mb, float *workspace, int n) {
    float *ws = workspace;
    const int owner = wide_shared_input(nets, n_nets) ? 0 : n;
    for (int i = 0; i < owner; i++) { WideLayout L; wide_layout(nets[i], mb, L); ws += L.total; }
    WideLayout L; wide_layout(nets[owner], mb, L);
    return ws + L.x0p;
}
static int wide_check(const lg_mlp_net *nets, int32_t n_nets, int32_t mb) {
    if (!nets || n_nets < 1 || n_nets > 2 || mb <= 0) return fail(-1, "bad argument");
    for (int n = 0; n < n_nets; n++) {
        for (int l = 0; l <= 4; l++) if (nets[n].dims[l] <= 0 || nets[n].dims[l] > 4096) return fail(-4, "lg_mlp_wide_*: layer widths must be in 1..4096");
        for (int l = 0; l < 4; l++) if (!nets[n].weights[l] || !nets[n].biases[l]) return fail(-1, "null layer pointer");
        if (!nets[n].input) return fail(-1, "null input");
    }
    return 0;
}

/* g_wide_precision (defined with the fused-actor host code): 0: exact f32 MFMA (k_gemm_wide), 1: split-bf16 (k_gemm_wide_bf16x3) */
#define LAUNCH_WIDE(MODE, GRID, ARGS)                                                                              \
    { if (g_wide_precision == 0) hipLaunchKernelGGL((lg::k_gemm_wide<MODE>), GRID, dim3(256), 0, st, ARGS);       \
      else hipLaunchKernelGGL((lg::k_gemm_wide_bf16x3<MODE>), GRID, dim3(256), 0, st, ARGS); }

extern "C" {

/* 0: exact f32 MFMA; 1 (default): split-bf16 products hi*hi + hi*lo + lo*hi with f32 accumulation (~2^-15 relative, ~5 x faster).  Returns the previous setting. */
int lg_mlp_wide_set_precision(int mode) { const int old = g_wide_precision; if (mode == 0 || mode == 1) g_wide_precision = mode; return old; }

size_t lg_mlp_wide_workspace_bytes(const lg_mlp_net *nets, int32_t n_nets, int32_t mb) {
    if (!nets || mb <= 0) return 0;
    size_t total = 0;
    for (int n = 0; n < n_nets; n++) { WideLayout L; wide_layout(nets[n], mb, L); total += L.total; }
    return total * sizeof(float);
}

int lg_mlp_wide_forward(const lg_mlp_net *nets, int32_t n_nets, const int64_t *rows, int32_t mb, float *workspace, size_t workspace_bytes, void *stream) {
    if (int rc = wide_check(nets, n_nets, mb)) return rc;
    if (!workspace || workspace_bytes < lg_mlp_wide_workspace_bytes(nets, n_nets, mb)) return fail(-1, "workspace too small (lg_mlp_wide_workspace_bytes)");
    for (int n = 0; n < n_nets; n++) if (!nets[n].output) return fail(-1, "null output");
    hipStream_t st = (hipStream_t)stream;
    {   // layer-0 operands: gathered, padded, aligned
        lg::WidePrepArgs pr; memset(&pr, 0, sizeof pr);
        float *ws = workspace;
        size_t work = 0;
        for (int n = 0; n < n_nets; n++) {
            WideLayout L; wide_layout(nets[n], mb, L);
            pr.x[n] = nets[n].input; pr.w[n] = nets[n].weights[0]; pr.xp[n] = ws + L.x0p; pr.wp[n] = ws + L.w0p;
            pr.skip_x[n] = n > 0 && wide_shared_input(nets, n_nets);
            pr.d0[n] = nets[n].dims[0]; pr.d1[n] = nets[n].dims[1]; pr.k0p[n] = L.k0p;
            const size_t w_ = ((size_t)mb + nets[n].dims[1]) * (L.k0p / 4);
            if (w_ > work) work = w_;
            ws += L.total;
        }
        pr.rows = rows; pr.mb = mb;
        int blocks = (int)((work + 255) / 256); if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(lg::k_wide_prep, dim3(blocks, n_nets), dim3(256), 0, st, pr);
    }
    bool chain = g_wide_precision == 1;
    int k0s = 0;
    for (int n = 0; n < n_nets; n++) {
        WideLayout L; wide_layout(nets[n], mb, L);
        chain = chain && L.chain && (n == 0 || L.ks[0] == k0s);
        k0s = L.ks[0];
    }
    if (chain) {                                       // all four layers in one launch, activations on chip (lg_policy.h: k_mlp_chain_fwd64)
        lg::ChainPackArgs pk; memset(&pk, 0, sizeof pk);
        lg::ChainArgs c; memset(&c, 0, sizeof c);
        c.mb = mb;
        float *ws = workspace;
        size_t work = 0;
        for (int n = 0; n < 2; n++) {
            const int m = n < n_nets ? n : 0;          // a single net: the second descriptor mirrors the first (never launched: grid z / y = n_nets)
            if (n == n_nets) ws = workspace;
            WideLayout L; wide_layout(nets[m], mb, L);
            lg::ChainNet &cn = c.net[n];
            cn.x = wide_x0p(nets, n_nets, mb, workspace, m); cn.ldx = L.k0p; cn.num_in = nets[m].dims[0];
            for (int l = 0; l < 4; l++) {
                pk.W[n][l] = nets[m].weights[l]; pk.b[n][l] = nets[m].biases[l];
                pk.wp[n][l] = reinterpret_cast<__bf16 *>(ws + L.wpk[l]); pk.bp[n][l] = ws + L.bpk[l];
                pk.in_dim[n][l] = nets[m].dims[l]; pk.out_dim[n][l] = nets[m].dims[l + 1]; pk.KS[n][l] = L.ks[l]; pk.OT[n][l] = L.ot[l];
                cn.wb[l] = reinterpret_cast<const lg::bf16x8g *>(ws + L.wpk[l]); cn.bb[l] = ws + L.bpk[l];
                const size_t w_ = (size_t)L.ot[l] * L.ks[l] * 512 + (size_t)L.ot[l] * 32;
                if (w_ > work) work = w_;
            }
            for (int l = 0; l < 3; l++) { cn.act[l] = ws + L.x[l + 1]; cn.lda[l] = nets[m].dims[l + 1]; }
            cn.out = nets[m].output; cn.out_dim = nets[m].dims[4];
            ws += L.total;
        }
        int blocks = (int)((work + 255) / 256); if (blocks > 512) blocks = 512;
        hipLaunchKernelGGL(lg::k_chain_pack, dim3(blocks, 4, n_nets), dim3(256), 0, st, pk);
        const dim3 grid((mb + 63) / 64, n_nets), block(64 * LG_PW_WAVES);                  // 64 rows per workgroup (k_mlp_chain_fwd64)
        if (k0s == 15) hipLaunchKernelGGL((lg::k_mlp_chain_fwd64<15>), grid, block, 0, st, c);
        else hipLaunchKernelGGL((lg::k_mlp_chain_fwd64<11>), grid, block, 0, st, c);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    for (int l = 0; l < 4; l++) {
        lg::GemmArgs a; memset(&a, 0, sizeof a);
        a.rows = nullptr; a.gather_a_rows = 0; a.mb = mb;
        int gx = 0, gy = 0;
        float *ws = workspace;
        for (int n = 0; n < n_nets; n++) {
            WideLayout L; wide_layout(nets[n], mb, L);
            lg::GemmNet &g = a.net[n];
            const int32_t *d = nets[n].dims;
            g.A = l == 0 ? wide_x0p(nets, n_nets, mb, workspace, n) : ws + L.x[l]; g.lda = l == 0 ? L.k0p : d[l];
            g.B = l == 0 ? ws + L.w0p : nets[n].weights[l]; g.ldb = l == 0 ? L.k0p : d[l]; g.bias = nets[n].biases[l];
            g.C = l == 3 ? nets[n].output : ws + L.x[l + 1]; g.ldc = d[l + 1];
            g.M = mb; g.N = d[l + 1]; g.K = d[l]; g.elu = l < 3; g.splits = 1; g.k_chunk = g.K;
            g.tiles_m = (g.M + LG_GT - 1) / LG_GT; g.tiles_n = (g.N + LG_GT - 1) / LG_GT;
            if (g.tiles_m > gx) gx = g.tiles_m;
            if (g.tiles_n > gy) gy = g.tiles_n;
            ws += L.total;
        }
        LAUNCH_WIDE(lg::GEMM_FWD, dim3(gx, gy, n_nets), a)
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

/* Gradients of all weights / biases given dL/d output (nets[n].grad_output); uses the activations the preceding
 * lg_mlp_wide_forward left in the same workspace. */
int lg_mlp_wide_backward(const lg_mlp_net *nets, int32_t n_nets, const int64_t *rows, int32_t mb, float *workspace, size_t workspace_bytes, void *stream) {
    if (int rc = wide_check(nets, n_nets, mb)) return rc;
    if (!workspace || workspace_bytes < lg_mlp_wide_workspace_bytes(nets, n_nets, mb)) return fail(-1, "workspace too small (lg_mlp_wide_workspace_bytes)");
    for (int n = 0; n < n_nets; n++) {
        if (!nets[n].grad_output) return fail(-1, "null grad_output");
        for (int l = 0; l < 4; l++) if (!nets[n].grad_weights[l] || !nets[n].grad_biases[l]) return fail(-1, "null gradient pointer");
    }
    hipStream_t st = (hipStream_t)stream;
    bool narrow = true;
    for (int n = 0; n < n_nets; n++) { WideLayout L; wide_layout(nets[n], mb, L); narrow = narrow && L.out_narrow; }
    lg::WideReduceArgs red; memset(&red, 0, sizeof red);            // the partials of all layers, summed by one launch behind the last GEMM
    int red_max = 0;
    auto add_job = [&](int l, int n, const float *part, int N_, int K_, int ld, int splits, int group) {
        lg::WideReduceJob &j = red.job[2 * l + n];
        j.part = part; j.gw = nets[n].grad_weights[l]; j.gb = nets[n].grad_biases[l]; j.N = N_; j.K = K_; j.ld = ld; j.splits = splits; j.group = group;
        if (N_ * ld * group > red_max) red_max = N_ * ld * group;
    };
    for (int l = 3; l >= 0; l--) {
        if (l == 3 && narrow) {                        // the <= 16-wide output layer: G_3, dW_3, db_3 from one read of A_3 (lg_gemm.h: k_wide_out_bwd)
            lg::OutBwdArgs o; memset(&o, 0, sizeof o);
            o.mb = mb;
            int chunks = 0;
            float *ws = workspace;
            for (int n = 0; n < n_nets; n++) {
                WideLayout L; wide_layout(nets[n], mb, L);
                const int32_t *d = nets[n].dims;
                lg::OutBwdNet &q = o.net[n];
                q.dz = nets[n].grad_output; q.w = nets[n].weights[3]; q.act = ws + L.x[3]; q.g = ws + L.g[3]; q.part = ws + L.part[3];
                q.N = d[4]; q.K = d[3]; q.ld = (d[3] + 1 + 3) & ~3; q.chunks = L.out_chunks; q.rows_per_chunk = (mb + L.out_chunks - 1) / L.out_chunks;
                if (q.chunks > chunks) chunks = q.chunks;
                add_job(3, n, q.part, d[4], d[3], q.ld, q.chunks, 8);
                ws += L.total;
            }
            hipLaunchKernelGGL((lg::k_wide_out_bwd<LG_OUT_MAXN>), dim3(chunks, n_nets), dim3(256), 0, st, o);
            continue;
        }
        // dW_l, db_l (split over the mini-batch rows) ...
        lg::GemmArgs a; memset(&a, 0, sizeof a);
        a.rows = nullptr; a.gather_b_k = 0; a.mb = mb;          // layer 0 reads the gathered copy the forward pass left in the workspace
        int gx = 0, gy = 0;
        float *ws = workspace;
        for (int n = 0; n < n_nets; n++) {
            WideLayout L; wide_layout(nets[n], mb, L);
            lg::GemmNet &g = a.net[n];
            const int32_t *d = nets[n].dims;
            g.A = l == 3 ? nets[n].grad_output : ws + L.g[l + 1]; g.lda = d[l + 1];
            g.B = l == 0 ? wide_x0p(nets, n_nets, mb, workspace, n) : ws + L.x[l]; g.ldb = l == 0 ? L.k0p : d[l];
            g.C = ws + L.part[l]; g.ldc = (d[l] + 1 + 3) & ~3;
            g.M = d[l + 1]; g.N = d[l]; g.K = mb; g.splits = L.splits[l]; g.k_chunk = L.kchunk[l];
            g.tiles_m = (g.M + LG_GT - 1) / LG_GT; g.tiles_n = (d[l] + LG_GT - 1) / LG_GT;
            if (g.tiles_m > gx) gx = g.tiles_m;
            if (g.tiles_n * g.splits > gy) gy = g.tiles_n * g.splits;
            add_job(l, n, ws + L.part[l], d[l + 1], d[l], (d[l] + 1 + 3) & ~3, L.splits[l], 1);
            ws += L.total;
        }
        LAUNCH_WIDE(lg::GEMM_DW, dim3(gx, gy, n_nets), a)
        if (l == 0) break;
        // ... and G_l = (G_{l+1} W_l) * elu'(X_l)
        lg::GemmArgs b; memset(&b, 0, sizeof b);
        b.mb = mb;
        gx = gy = 0; ws = workspace;
        for (int n = 0; n < n_nets; n++) {
            WideLayout L; wide_layout(nets[n], mb, L);
            lg::GemmNet &g = b.net[n];
            const int32_t *d = nets[n].dims;
            g.A = l == 3 ? nets[n].grad_output : ws + L.g[l + 1]; g.lda = d[l + 1];
            g.B = nets[n].weights[l]; g.ldb = d[l];
            g.C = ws + L.g[l]; g.ldc = d[l]; g.act = ws + L.x[l];
            g.M = mb; g.N = d[l]; g.K = d[l + 1]; g.splits = 1; g.k_chunk = g.K;
            g.tiles_m = (g.M + LG_GT - 1) / LG_GT; g.tiles_n = (g.N + LG_GT - 1) / LG_GT;
            if (g.tiles_m > gx) gx = g.tiles_m;
            if (g.tiles_n > gy) gy = g.tiles_n;
            ws += L.total;
        }
        LAUNCH_WIDE(lg::GEMM_DX, dim3(gx, gy, n_nets), b)
    }
    hipLaunchKernelGGL(lg::k_wide_reduce, dim3((red_max + 255) / 256, LG_REDUCE_JOBS), dim3(256), 0, st, red);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C" (wide MLP entry points; still inside the enclosing block)

int lg_rollout_record(const lg_rollout_step *s, void *stream) {
    if (!s || !s->obs || !s->actions || !s->mean || !s->rewards || !s->dones || !s->storage_obs || !s->storage_actions || !s->storage_mu ||
        !s->storage_rewards || !s->storage_dones) return fail(-1, "null argument");
    if (s->num_envs <= 0 || s->num_obs <= 0 || s->num_actions <= 0 || s->num_actions > s->num_obs) return fail(-1, "bad sizes");
    if ((s->cur_return != nullptr) != (s->cur_length != nullptr) || (s->cur_return && !s->sums)) return fail(-1, "incomplete episode statistics");
    lg::RecordArgs a{s->obs, s->actions, s->mean, s->rewards, s->dones, s->time_outs, s->storage_obs, s->storage_actions, s->storage_mu,
                     s->storage_rewards, s->storage_dones, s->storage_time_outs, s->cur_return, s->cur_length, s->sums,
                     s->std, s->storage_sigma, s->storage_log_prob, s->num_envs, s->num_obs, s->num_actions};
    if (s->std && (!s->storage_sigma || !s->storage_log_prob)) return fail(-1, "std given without storage_sigma / storage_log_prob");
    const int64_t n = (int64_t)s->num_envs * s->num_obs;
    if (s->num_actions > 16) return fail(-1, "lg_rollout_record: at most 16 actions");
    hipLaunchKernelGGL(lg::k_rollout_record, dim3((unsigned)((n + 255) / 256 + ((int64_t)s->num_envs * 16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int lg_rollout_finish(const lg_rollout_post *s, void *stream) {
    if (!s || !s->actions || !s->mean || !s->rewards || !s->dones || !s->std || !s->sigma || !s->log_prob) return fail(-1, "null argument");
    if (s->steps <= 0 || s->num_envs <= 0 || s->num_actions <= 0 || s->num_actions > 16) return fail(-1, "bad sizes (at most 16 actions)");
    if ((s->cur_return != nullptr) != (s->cur_length != nullptr) || (s->cur_return && !s->sums)) return fail(-1, "incomplete episode statistics");
    lg::RollPostArgs a{s->actions, s->mean, s->rewards, s->dones, s->time_outs, s->std, s->sigma, s->log_prob, s->time_outs_f,
                       s->cur_return, s->cur_length, s->sums, s->steps, s->num_envs, s->num_actions};
    const int64_t n_tr = (int64_t)s->steps * s->num_envs;
    hipLaunchKernelGGL(lg::k_rollout_post, dim3((unsigned)((n_tr * 16 + 255) / 256 + (s->num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int lg_adam_step(const lg_adam_tensor *tensors, int32_t n_tensors, float *lr, float beta1, float beta2, float eps, float max_grad_norm,
                 const float *kl, float desired_kl, float *scratch, void *stream) {
    if (!tensors || !lr || !scratch || n_tensors < 1 || n_tensors > LG_ADAM_MAX_TENSORS) return fail(-1, "bad argument");
    lg::AdamArgs a; memset(&a, 0, sizeof a);
    int64_t max_n = 0;
    for (int k = 0; k < n_tensors; k++) {
        const lg_adam_tensor &t = tensors[k];
        if (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq || !t.step || t.numel <= 0) return fail(-1, "incomplete optimiser tensor");
        a.t[k] = lg::AdamTensor{t.param, t.grad, t.exp_avg, t.exp_avg_sq, t.step, t.numel};
        if (t.numel > max_n) max_n = t.numel;
    }
    a.n_tensors = n_tensors; a.lr = lr; a.kl = kl; a.scratch = scratch;
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.max_norm = max_grad_norm; a.desired_kl = desired_kl;
    hipStream_t st = (hipStream_t)stream;
    static_assert(LG_ADAM_SCRATCH_FLOATS >= 2 + LG_ADAM_MAX_TENSORS * LG_ADAM_CHUNKS, "scratch contract");
    const int64_t chunk_len = (max_n + LG_ADAM_CHUNKS - 1) / LG_ADAM_CHUNKS;
    hipLaunchKernelGGL(lg::k_adam_sumsq, dim3(LG_ADAM_CHUNKS, n_tensors), dim3(256), 0, st, a, chunk_len);
    hipLaunchKernelGGL(lg::k_adam_prepare, dim3(1), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(lg::k_adam_update, dim3((unsigned)((max_n + 255) / 256), n_tensors), dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"

extern "C" {

static int policy_pack_wide(lg_policy *p, const float *const weights[4], const float *const biases[4], hipStream_t st) {
    for (int i = 0; i < 4; i++) {
        const size_t n = (size_t)p->wide_ot[i] * p->wide_ks[i] * 512 + (size_t)p->wide_ot[i] * 32;
        hipLaunchKernelGGL(lg::k_policy_pack_wide, dim3((unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256)), dim3(256), 0, st, weights[i], biases[i],
                           p->dims[i], p->dims[i + 1], p->wide_ks[i], p->wide_ot[i], i == 0 ? 1 : 0, p->d_wb[i], p->d_bb[i]);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int lg_policy_load_device(lg_policy *p, const float *const weights[4], const float *const biases[4], const float *std, void *stream) {
    if (!p || !weights || !biases || !std) return fail(-1, "null argument");
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < 4; i++) {
        if (!weights[i] || !biases[i]) return fail(-1, "null layer pointer");
        const int in_t = p->tiles[i], out_t = (i < 3) ? p->tiles[i + 1] : 1;
        const size_t n = (size_t)out_t * in_t * 4 * 64 + (size_t)out_t * 4 * 64;
        hipLaunchKernelGGL(k_policy_pack, dim3((unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, st, weights[i], biases[i],
                           p->dims[i], p->dims[i + 1], in_t, out_t, p->d_w[i], p->d_b[i]);
    }
    if (p->wide) { int rc = policy_pack_wide(p, weights, biases, st); if (rc) return rc; }
    HIP_TRY(hipMemcpyAsync(p->d_std, std, p->dims[4] * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipGetLastError());
    return 0;
}

int lg_policy_create(const int32_t dims[5], const float *const weights[4], const float *const biases[4], const float *std,
                     int device_id, lg_policy **out) {
    if (!dims || !weights || !biases || !std || !out) return fail(-1, "null argument");
    for (int i = 1; i <= 3; i++) if (dims[i] % 16 || dims[i] <= 0 || dims[i] > 512) return fail(-4, "hidden widths must be multiples of 16, <= 512");
    if (dims[0] <= 0 || dims[0] > 256 || dims[4] <= 0 || dims[4] > 16) return fail(-4, "unsupported obs / action width");
    HIP_TRY(hipSetDevice(device_id));
    lg_policy *p = new (std::nothrow) lg_policy();
    if (!p) return fail(-5, "out of host memory");
    memcpy(p->dims, dims, sizeof p->dims); p->device = device_id;
    for (int i = 0; i < 4; i++) { p->d_w[i] = nullptr; p->d_b[i] = nullptr; p->d_wb[i] = nullptr; p->d_bb[i] = nullptr; }
    p->d_std = nullptr;
    p->tiles[0] = (dims[0] + 15) / 16; p->tiles[1] = dims[1] / 16; p->tiles[2] = dims[2] / 16; p->tiles[3] = dims[3] / 16;
    p->wide = dims[1] == 512 && dims[2] == 256 && dims[3] == 128;
    for (int i = 0; i < 4; i++) {
        int in_t = p->tiles[i], out_t = (i < 3) ? p->tiles[i + 1] : 1;
        size_t nw = (size_t)out_t * in_t * 4 * 64, nb = (size_t)out_t * 4 * 64;
        float *hw = (float *)malloc(nw * 4), *hb = (float *)malloc(nb * 4);
        policy_pack_layer(weights[i], biases[i], dims[i], dims[i + 1], in_t, out_t, hw, hb);
        bool ok = hipMalloc(&p->d_w[i], nw * 4) == hipSuccess && hipMalloc(&p->d_b[i], nb * 4) == hipSuccess &&
                  hipMemcpy(p->d_w[i], hw, nw * 4, hipMemcpyHostToDevice) == hipSuccess &&
                  hipMemcpy(p->d_b[i], hb, nb * 4, hipMemcpyHostToDevice) == hipSuccess;
        free(hw); free(hb);
        if (!ok) { lg_policy_destroy(p); return fail(-10, "policy weight upload failed"); }
    }
    if (hipMalloc(&p->d_std, 16 * 4) != hipSuccess || hipMemcpy(p->d_std, std, dims[4] * 4, hipMemcpyHostToDevice) != hipSuccess) {
        lg_policy_destroy(p); return fail(-10, "policy std upload failed");
    }
    if (p->wide) {                                                 // split-bf16 operand stream: raw parameters up, packed on the device
        float *raw_w[4] = {nullptr, nullptr, nullptr, nullptr}, *raw_b[4] = {nullptr, nullptr, nullptr, nullptr};
        bool ok = true;
        for (int i = 0; i < 4 && ok; i++) {
            p->wide_ks[i] = i == 0 ? (dims[0] + 15) / 16 : dims[i] / 16;
            p->wide_ot[i] = (dims[i + 1] + 31) / 32;
            const size_t nw = (size_t)dims[i] * dims[i + 1], nb = (size_t)dims[i + 1];
            ok = hipMalloc(&p->d_wb[i], (size_t)p->wide_ot[i] * p->wide_ks[i] * 1024 * sizeof(__bf16)) == hipSuccess &&
                 hipMalloc(&p->d_bb[i], (size_t)p->wide_ot[i] * 32 * 4) == hipSuccess &&
                 hipMalloc(&raw_w[i], nw * 4) == hipSuccess && hipMalloc(&raw_b[i], nb * 4) == hipSuccess &&
                 hipMemcpy(raw_w[i], weights[i], nw * 4, hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(raw_b[i], biases[i], nb * 4, hipMemcpyHostToDevice) == hipSuccess;
        }
        if (ok) ok = policy_pack_wide(p, raw_w, raw_b, nullptr) == 0 && hipStreamSynchronize(nullptr) == hipSuccess;
        for (int i = 0; i < 4; i++) { if (raw_w[i]) (void)hipFree(raw_w[i]); if (raw_b[i]) (void)hipFree(raw_b[i]); }
        if (!ok) { lg_policy_destroy(p); return fail(-10, "wide policy weight upload failed"); }
    }
    *out = p;
    return 0;
}

void lg_policy_destroy(lg_policy *p) {
    if (!p) return;
    for (int i = 0; i < 4; i++) {
        if (p->d_w[i]) (void)hipFree(p->d_w[i]);
        if (p->d_b[i]) (void)hipFree(p->d_b[i]);
        if (p->d_wb[i]) (void)hipFree(p->d_wb[i]);
        if (p->d_bb[i]) (void)hipFree(p->d_bb[i]);
    }
    if (p->d_std) (void)hipFree(p->d_std);
    delete p;
}

static void fill_policy_args(const lg_policy *p, PolicyArgs &a, const float *obs, float *actions, float *mean, int32_t num_envs, uint64_t seed,
                             int64_t step, const int64_t *step_counter, int32_t deterministic) {
    a.obs = obs; a.actions = actions; a.mean = mean; a.std = p->d_std; a.step_counter = step_counter; a.step = step; a.seed = seed;
    a.num_envs = num_envs; a.num_obs = p->dims[0]; a.num_actions = p->dims[4]; a.deterministic = deterministic;
    for (int i = 0; i < 4; i++) { a.w[i] = p->d_w[i]; a.b[i] = p->d_b[i]; }
}

int lg_policy_act(lg_policy *p, const float *obs, float *actions, float *mean, int32_t num_envs, uint64_t seed, int64_t step,
                  const int64_t *step_counter, int32_t deterministic, void *stream) {
    if (!p || !obs || !actions) return fail(-1, "null argument");
    if (num_envs <= 0) return 0;
    PolicyArgs a;
    fill_policy_args(p, a, obs, actions, mean, num_envs, seed, step, step_counter, deterministic);
    dim3 g((num_envs + 15) / 16), b(64 * LG_POLICY_WAVES);
    hipStream_t st = (hipStream_t)stream;
    const int t0 = p->tiles[0], t1 = p->tiles[1], t2 = p->tiles[2], t3 = p->tiles[3];
    if (p->wide && g_wide_precision == 1 && (t0 == 15 || t0 == 11)) {          // 32 envs per workgroup on the bf16 matrix cores
        lg::PolicyWideArgs w; w.base = a;
        for (int i = 0; i < 4; i++) { w.wb[i] = reinterpret_cast<const lg::bf16x8g *>(p->d_wb[i]); w.bb[i] = p->d_bb[i]; }
        dim3 gw((num_envs + LG_PW_ENVS - 1) / LG_PW_ENVS), bw(64 * LG_PW_WAVES);
        if (t0 == 15) hipLaunchKernelGGL((lg::k_policy_act_wide<15, 16, 8, 4>), gw, bw, 0, st, w);     // rough: 235-512-256-128
        else hipLaunchKernelGGL((lg::k_policy_act_wide<11, 16, 8, 4>), gw, bw, 0, st, w);              // cassie: 169-512-256-128
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (t0 == 3 && t1 == 8 && t2 == 4 && t3 == 2) hipLaunchKernelGGL((k_policy_act<3, 8, 4, 2>), g, b, 0, st, a);            // flat: 48-128-64-32
    else if (t0 == 15 && t1 == 32 && t2 == 16 && t3 == 8) hipLaunchKernelGGL((k_policy_act<15, 32, 16, 8>), g, b, 0, st, a);  // rough: 235-512-256-128
    else if (t0 == 11 && t1 == 32 && t2 == 16 && t3 == 8) hipLaunchKernelGGL((k_policy_act<11, 32, 16, 8>), g, b, 0, st, a);  // cassie: 169-512-256-128
    else return fail(-4, "actor widths are not one of the compiled-in shapes (48-128-64-32, 235/169-512-256-128)");
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ GAE scan (PPO rollout post-processing)
__global__ void __launch_bounds__(256) k_gae(const float *__restrict__ rewards, const float *__restrict__ values, const uint8_t *__restrict__ dones,
                                             const float *__restrict__ last_values, float gamma, float lam, float *__restrict__ returns,
                                             float *__restrict__ advantages, int T, int N) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    float adv = 0.0f, nxt = last_values[e];
    for (int t = T - 1; t >= 0; t--) {                 // coalesced across envs at every t
        const size_t i = (size_t)t * N + e;
        const float v = values[i], nd = 1.0f - (float)dones[i];
        const float delta = rewards[i] + nd * gamma * nxt - v;
        adv = delta + nd * gamma * lam * adv;
        returns[i] = adv + v;
        advantages[i] = (adv + v) - v;                 // = returns - values, as the reference computes it
        nxt = v;
    }
}

int lg_gae_returns(const float *rewards, const float *values, const uint8_t *dones, const float *last_values, float gamma, float lam,
                   float *returns, float *advantages, int32_t num_steps, int32_t num_envs, void *stream) {
    if (!rewards || !values || !dones || !last_values || !returns || !advantages) return fail(-1, "null argument");
    if (num_steps <= 0 || num_envs <= 0) return 0;
    hipLaunchKernelGGL(k_gae, dim3((num_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, rewards, values, dones, last_values, gamma, lam,
                       returns, advantages, num_steps, num_envs);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ fused PPO loss + gradient w.r.t. the network outputs
#define LG_PPO_MAX_ACTIONS 16
// 16 lanes per row, one action per lane: the row's actions / old means / old sigmas are 48 contiguous bytes read by one instruction per
// array (one row per lane meant 36 load instructions of 64 scattered lines each: 17.6 us for 24 576 rows), the log-probability and the KL
// are 16-lane butterflies, d_mu leaves coalesced.  A workgroup walks LG_LOSS_ROWS_PER_WG rows (passes unrolled: their gathers overlap) and
// keeps its sums in registers.  Measured 15.9 us at 64 rows per workgroup (32: 17.1, 128: 21.3): what is left is the ~770 same-line
// float atomics (2 wave instructions per workgroup) at ~20 ns each; fewer workgroups trade them for longer serial chains.
#define LG_LOSS_ROWS_PER_WG 64
__global__ void __launch_bounds__(256) k_ppo_loss(const float *__restrict__ mu, const float *__restrict__ stdp, const float *__restrict__ value,
                                                  const int64_t *__restrict__ rows, const float *__restrict__ actions, const float *__restrict__ old_lp,
                                                  const float *__restrict__ old_mu, const float *__restrict__ old_sigma, const float *__restrict__ adv,
                                                  const float *__restrict__ old_values, const float *__restrict__ returns, float clip, float vcoef,
                                                  float ecoef, int clipped_value, float *__restrict__ d_mu, float *__restrict__ d_std,
                                                  float *__restrict__ d_value, float *__restrict__ stats, int mb, int A) {
    __shared__ float red[4 + LG_PPO_MAX_ACTIONS];
    if (threadIdx.x < 4 + LG_PPO_MAX_ACTIONS) red[threadIdx.x] = 0.0f;
    __syncthreads();
    const int a = threadIdx.x & 15, rl = threadIdx.x >> 4;             // action of this lane; row slot 0..15 of the pass
    const float inv_n = 1.0f / (float)mb;
    const bool act_lane = a < A;
    const float sg = act_lane ? stdp[a] : 1.0f, isg = 1.0f / sg, lsg = __logf(sg);
    float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f, gstd = 0.0f;
    const int r_begin = blockIdx.x * LG_LOSS_ROWS_PER_WG, r_end = min(mb, r_begin + LG_LOSS_ROWS_PER_WG);
#pragma unroll 4                                                       // the four passes' gathers in flight together (each pass alone is a ~2.5 us chain)
    for (int i0 = r_begin; i0 < r_end; i0 += 16) {
        const int i = i0 + rl;
        const bool live = i < r_end;
        const int ii = live ? i : r_end - 1;
        const size_t r = (size_t)rows[ii];
        float z = 0.0f, lp = 0.0f, kl = 0.0f;
        if (act_lane) {
            const float m = mu[(size_t)ii * A + a], om = old_mu[r * A + a], os = old_sigma[r * A + a];
            z = (actions[r * A + a] - m) * isg;
            lp = -0.5f * z * z - lsg - 0.918938533f;                                   // log N(a; mu, sigma)
            kl = __logf(sg / os + 1.0e-5f) + (os * os + (om - m) * (om - m)) * (0.5f * isg * isg) - 0.5f;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { lp += __shfl_xor(lp, o); kl += __shfl_xor(kl, o); }     // every lane of the row holds the sums
        const float ad = adv[r], ratio = __expf(lp - old_lp[r]);
        const float s1 = -ad * ratio, s2 = -ad * fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
        const bool inside = ratio >= 1.0f - clip && ratio <= 1.0f + clip;
        // d max(s1, s2) / d lp: s1 wins (or ties, inside the clamp range) -> -A r; the clamped branch has no gradient outside the range
        const float dlp = (s1 > s2 || inside) ? -ad * ratio : (s1 == s2 ? -0.5f * ad * ratio : 0.0f);
        if (live && act_lane) {
            d_mu[(size_t)i * A + a] = inv_n * dlp * z * isg;                           // d lp / d mu = (a - mu) / sigma^2
            gstd += inv_n * dlp * (z * z - 1.0f) * isg;                                // d lp / d sigma = ((a - mu)^2 / sigma^2 - 1) / sigma
        }
        if (live && a == 0) {
            const float v = value[i], tv = old_values[r], R = returns[r];
            float dv, vl;
            if (clipped_value) {
                const float dvt = v - tv, vc = tv + fminf(fmaxf(dvt, -clip), clip);
                const float v1 = (v - R) * (v - R), v2 = (vc - R) * (vc - R);
                const bool in_v = dvt >= -clip && dvt <= clip;
                vl = fmaxf(v1, v2);
                dv = (v1 > v2 || in_v) ? 2.0f * (v - R) : (v1 == v2 ? (v - R) : 0.0f);
            } else {
                vl = (R - v) * (R - v);
                dv = 2.0f * (v - R);
            }
            d_value[i] = vcoef * inv_n * dv;
            acc0 += fmaxf(s1, s2); acc1 += vl; acc2 += kl;
        }
    }
    // this thread's sums: acc* on the a == 0 lanes (rows rl, rl + 16, ...), gstd for action a.  Fold the 4 row slots of the wave (lanes 16
    // apart), then the waves through LDS, then one atomic per value and workgroup.
#pragma unroll
    for (int o = 32; o >= 16; o >>= 1) { acc0 += __shfl_xor(acc0, o); acc1 += __shfl_xor(acc1, o); acc2 += __shfl_xor(acc2, o); gstd += __shfl_xor(gstd, o); }
    if ((threadIdx.x & 63) < 16) {
        if (a == 0) { atomicAdd(&red[0], acc0); atomicAdd(&red[1], acc1); atomicAdd(&red[2], acc2); }
        if (act_lane) atomicAdd(&red[4 + a], gstd);
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicAdd(stats + threadIdx.x, red[threadIdx.x] * inv_n);
    if (threadIdx.x >= 4 && threadIdx.x < 4 + A) atomicAdd(d_std + (threadIdx.x - 4), red[threadIdx.x]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {                                          // entropy is row-independent: sum_a (0.5 + 0.5 log 2 pi + log sigma_a)
        float H = 0.0f;
        for (int k = 0; k < A; k++) { H += 1.418938533f + __logf(stdp[k]); atomicAdd(d_std + k, -ecoef / stdp[k]); }
        stats[3] = H;
    }
}

__global__ void k_zero2(float *a, int na, float *b, int nb) {
    if ((int)threadIdx.x < na) a[threadIdx.x] = 0.0f;
    if ((int)threadIdx.x < nb) b[threadIdx.x] = 0.0f;
}

int lg_ppo_loss(const float *mu, const float *std, const float *value, const int64_t *rows, const float *actions, const float *old_log_prob,
                const float *old_mu, const float *old_sigma, const float *advantages, const float *old_values, const float *returns, float clip,
                float value_coef, float entropy_coef, int32_t use_clipped_value, float *d_mu, float *d_std, float *d_value, float *stats,
                int32_t mb, int32_t num_actions, void *stream) {
    if (!mu || !std || !value || !rows || !actions || !old_log_prob || !old_mu || !old_sigma || !advantages || !old_values || !returns || !d_mu ||
        !d_std || !d_value || !stats) return fail(-1, "null argument");
    if (num_actions < 1 || num_actions > LG_PPO_MAX_ACTIONS) return fail(-4, "lg_ppo_loss supports 1..16 actions");
    if (mb <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_zero2, dim3(1), dim3(64), 0, st, stats, 4, d_std, (int)num_actions);     // (a kernel, not hipMemsetAsync: replayed inside HIP graphs)
    hipLaunchKernelGGL(k_ppo_loss, dim3((mb + LG_LOSS_ROWS_PER_WG - 1) / LG_LOSS_ROWS_PER_WG), dim3(256), 0, st, mu, std, value, rows, actions, old_log_prob, old_mu, old_sigma, advantages,
                       old_values, returns, clip, value_coef, entropy_coef, (int)use_clipped_value, d_mu, d_std, d_value, stats, (int)mb, (int)num_actions);
    HIP_TRY(hipGetLastError());
    return 0;
}

const char *lg_last_error(void) { return g_err; }
int lg_abi_version(void) { return LG_ABI_VERSION; }
#ifdef LG_PROFILE
// debug builds only (not part of legged_hip.h): copy out and optionally clear the k_step section accumulators
int lg_debug_profile(lg_sim *s, unsigned long long *out16 /* [LG_NPROF] */, int reset) {
    if (!s || !s->d_prof) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    std::vector<unsigned long long> h(LG_NPROF_TOTAL);
    if (hipMemcpy(h.data(), s->d_prof, sizeof(unsigned long long) * LG_NPROF_TOTAL, hipMemcpyDeviceToHost) != hipSuccess) return -2;
    for (int i = 0; i < LG_NPROF; i++) out16[i] = 0;
    for (int b = 0; b < LG_NPROF_BLOCKS; b++) {
        const unsigned long long *a = h.data() + LG_NPROF + (size_t)b * 40;
        for (int i = 0; i < 16; i++) out16[i] += a[i];
        if (a[16] > out16[16]) out16[16] = a[16];
    }
    if (reset && hipMemset(s->d_prof, 0, sizeof(unsigned long long) * LG_NPROF_TOTAL) != hipSuccess) return -2;
    return 0;
}
// the last launch, per workgroup: out[blocks][20] = 16 section slots, start and end on the 100 MHz wall clock, two user slots
int lg_debug_profile_blocks(lg_sim *s, unsigned long long *out, int blocks) {
    if (!s || !s->d_prof || blocks < 0 || blocks > LG_NPROF_BLOCKS) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    for (int b = 0; b < blocks; b++)
        if (hipMemcpy(out + (size_t)b * 20, s->d_prof + LG_NPROF + (size_t)b * 40 + 20, sizeof(unsigned long long) * 20, hipMemcpyDeviceToHost) != hipSuccess) return -2;
    return 0;
}
// rollout kernel, last launch: out[LG_NPROF_ROLL][LG_NPROF_BLOCKS] wall-clock stamps (100 MHz) of every workgroup at every step boundary
int lg_debug_profile_roll(lg_sim *s, unsigned long long *out) {
    if (!s || !s->d_prof) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (hipMemcpy(out, s->d_prof + LG_NPROF + (size_t)LG_NPROF_BLOCKS * 40, sizeof(unsigned long long) * LG_NPROF_ROLL * LG_NPROF_BLOCKS, hipMemcpyDeviceToHost) != hipSuccess) return -2;
    return 0;
}
#endif
int lg_sizeof(int which) {
    switch (which) { case 0: return (int)sizeof(lg_params); case 1: return (int)sizeof(lg_robot_model);
                     case 2: return (int)sizeof(lg_buffers); case 3: return (int)sizeof(lg_point);
                     case 4: return (int)sizeof(lg_mlp_net); case 5: return (int)sizeof(lg_adam_tensor);
                     case 6: return (int)sizeof(lg_rollout_step); case 7: return (int)sizeof(lg_ppo_batch); default: return -1; }
}

int lg_create(const lg_params *params, const lg_robot_model *model, const float *actuator_weights, int device_id, lg_sim **out) {
    if (!params || !model || !out) return fail(-1, "null argument");
    if (params->abi_version != LG_ABI_VERSION) return fail(-3, "ABI version mismatch");
    static_assert(LG_NUM_REWARD_TERMS + 1 <= 64, "k_extras uses one wave");
    static_assert(sizeof(KArgs) <= 4096, "kernel arguments must fit the 4 KiB kernarg segment");
    RobotKind kind;
    if (check_topology<AnymalTraits>(model)) kind = ROBOT_ANYMAL;
    else if (check_topology<CassieTraits>(model)) kind = ROBOT_CASSIE;
    else return fail(-4, "robot topology is not one of the compiled-in layouts (ANYmal-C 4x3, Cassie 2x6)");
    if (params->control_type == LG_CTRL_ACTUATOR_NET && !actuator_weights) return fail(-2, "actuator-net control without weights");
    if (params->control_type == LG_CTRL_ACTUATOR_NET && kind != ROBOT_ANYMAL) return fail(-2, "actuator net is compiled for the ANYmal layout only");
    if (params->self_collision && kind != ROBOT_ANYMAL) return fail(-4, "self-collision (asset.self_collisions = 0) is compiled for the quadruped layouts only");
    if (params->self_collision && !check_capsules<AnymalTraits>(model)) return fail(-4, "collision points do not form the compiled-in capsule layout (self-collision)");
    HIP_TRY(hipSetDevice(device_id));
    lg_sim *s = new (std::nothrow) lg_sim();
    if (!s) return fail(-5, "out of host memory");
    s->P = *params; s->M = *model; s->kind = kind; s->device = device_id; s->bound = false;
    s->has_net = actuator_weights != nullptr; s->d_weights = nullptr; s->d_limb_table = nullptr; s->d_done = nullptr; s->d_prof = nullptr; s->d_accum_alt = nullptr; s->defer = 0;
    s->h_status = nullptr; s->spin_limit = 1 << 22; s->debug_skip = 0; s->d_roll_extras = nullptr;
    if (hipHostMalloc(reinterpret_cast<void **>(&s->h_status), sizeof(unsigned int), hipHostMallocMapped) != hipSuccess) { delete s; return fail(-10, "hipHostMalloc failed"); }
    *s->h_status = 0u;
    if (hipMalloc(&s->d_accum_alt, (LG_NUM_REWARD_TERMS + 2) * sizeof(float)) != hipSuccess || hipMemset(s->d_accum_alt, 0, (LG_NUM_REWARD_TERMS + 2) * sizeof(float)) != hipSuccess) { delete s; return fail(-10, "hipMalloc failed"); }
    {   // [0] the workgroup ticket, [1 ..] one terrain-level partial sum per workgroup of k_step (at most 4 lanes per env)
        const size_t n_done = 1 + 2 * (((size_t)params->num_envs * 4 + LG_BLOCK - 1) / LG_BLOCK);      // (two slots of partials, by step parity)
        if (hipMalloc(&s->d_done, n_done * sizeof(unsigned int)) != hipSuccess || hipMemset(s->d_done, 0, n_done * sizeof(unsigned int)) != hipSuccess) { delete s; return fail(-10, "hipMalloc failed"); }
    }
    memset(&s->B, 0, sizeof s->B);
    { hipDeviceProp_t prop; s->num_cus = (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256; }
#ifdef LG_PROFILE
    if (hipMalloc(&s->d_prof, sizeof(unsigned long long) * LG_NPROF_TOTAL) != hipSuccess || hipMemset(s->d_prof, 0, sizeof(unsigned long long) * LG_NPROF_TOTAL) != hipSuccess) { delete s; return fail(-10, "hipMalloc failed"); }
#endif
    if (hipMalloc(&s->d_limb_table, sizeof(float) * (LG_MAX_LIMBS * (LG_MAX_CHAIN * LG_JS + 4 * LG_MAX_LIMB_POINTS + 1) + 2 * LG_MAX_HEIGHT_POINTS)) != hipSuccess) { delete s; return fail(-10, "hipMalloc failed"); }
    if (s->has_net) {
        float table[LW_ROWS * 64];
        build_lstm_table(actuator_weights, table);
        if (hipMalloc(&s->d_weights, sizeof table) != hipSuccess) { (void)hipFree(s->d_limb_table); delete s; return fail(-10, "hipMalloc failed"); }
        if (hipMemcpy(s->d_weights, table, sizeof table, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(s->d_weights); (void)hipFree(s->d_limb_table); delete s; return fail(-10, "hipMemcpy failed"); }
    }
    int rc = upload_tables(s);
    if (rc) { lg_destroy(s); return rc; }
    *out = s;
    return 0;
}

void lg_destroy(lg_sim *s) {
    if (!s) return;
    if (s->d_limb_table) (void)hipFree(s->d_limb_table);
    if (s->d_weights) (void)hipFree(s->d_weights);
    if (s->d_done) (void)hipFree(s->d_done);
    if (s->d_accum_alt) (void)hipFree(s->d_accum_alt);
    if (s->d_prof) (void)hipFree(s->d_prof);
    if (s->h_status) (void)hipHostFree(s->h_status);
    if (s->d_roll_extras) (void)hipFree(s->d_roll_extras);
    delete s;
}

int lg_bind(lg_sim *s, const lg_buffers *b) {
    if (!s || !b) return fail(-1, "null argument");
    const void *need[] = {b->root_states, b->dof_state, b->contact_forces, b->obs_buf, b->rew_buf, b->reset_buf, b->time_out_buf,
                          b->episode_length_buf, b->torques, b->actions, b->last_actions, b->last_dof_vel, b->last_root_vel, b->commands,
                          b->feet_air_time, b->last_contacts, b->base_lin_vel, b->base_ang_vel, b->projected_gravity, b->episode_sums,
                          b->episode_means, b->extras_accum, b->env_origins};
    for (size_t i = 0; i < sizeof need / sizeof *need; i++) if (!need[i]) return fail(-6, "a required buffer pointer is null");
    if (s->P.control_type == LG_CTRL_ACTUATOR_NET && (!b->sea_hidden_state || !b->sea_cell_state)) return fail(-6, "actuator state buffers missing");
    if (s->P.measure_heights && !b->measured_heights) return fail(-6, "measured_heights buffer missing");
    if (s->P.terrain_type == LG_TERRAIN_HEIGHTFIELD && !b->height_samples) return fail(-6, "height_samples missing");
    if (s->P.terrain_type == LG_TERRAIN_HEIGHTFIELD && (s->P.hf_rows < 2 || s->P.hf_cols < 2 || (int64_t)s->P.hf_rows * s->P.hf_cols > 0x7fffffff))
        return fail(-6, "height field must have 2 <= rows, cols and rows*cols < 2^31");
    if (s->P.terrain_curriculum && (!b->terrain_levels || !b->terrain_types || !b->terrain_origins)) return fail(-6, "terrain curriculum buffers missing");
    s->B = *b; s->bound = true;
    return 0;
}

int lg_set_obs_buffer(lg_sim *s, float *obs_buf) {
    if (!s || !obs_buf) return fail(-1, "null argument");
    s->B.obs_buf = obs_buf;
    return 0;
}

int lg_set_params(lg_sim *s, const lg_params *p) {
    if (!s || !p) return fail(-1, "null argument");
    if (p->num_envs != s->P.num_envs || p->control_type != s->P.control_type) return fail(-7, "num_envs / control_type cannot change after create");
    s->P = *p;
    return upload_tables(s);
}

// helper waves only pay off while every workgroup still gets a CU of its own in one round (all waves carry 512 registers)
static int waves_for(unsigned workgroups, int num_cus) { return workgroups <= (unsigned)num_cus ? 4 : (workgroups <= 2u * num_cus ? 2 : 1); }
#define LAUNCH_NW(TRAITS, HFV, SCV)                                                                                              \
    { if (nw == 4) hipLaunchKernelGGL((k_step<TRAITS, false, HFV, false, 4, SCV>), g, dim3(4 * LG_BLOCK), 0, st, a);             \
      else if (nw == 2) hipLaunchKernelGGL((k_step<TRAITS, false, HFV, false, 2, SCV>), g, dim3(2 * LG_BLOCK), 0, st, a);        \
      else hipLaunchKernelGGL((k_step<TRAITS, false, HFV, false, 1, SCV>), g, dim3(LG_BLOCK), 0, st, a); }

int lg_step(lg_sim *s, const float *actions, int64_t common_step_counter, void *stream) {
    if (!s || !s->bound) return fail(-8, "lg_bind has not been called");
    if (int rc = status_error(s)) return rc;
    if (!actions) return fail(-1, "null actions");
    if (common_step_counter < 0 && !s->B.step_counter) return fail(-9, "common_step_counter = -1 needs a step_counter buffer");
    KArgs a; fill_args(s, a, common_step_counter); a.actions_in = actions;
    hipStream_t st = (hipStream_t)stream;
    const bool hf = s->P.terrain_type == LG_TERRAIN_HEIGHTFIELD;
    const bool net = s->P.control_type == LG_CTRL_ACTUATOR_NET;
    if (s->kind == ROBOT_ANYMAL) {
        dim3 g(grid_for<AnymalTraits>(s->P.num_envs)), b(LG_STEP_WAVES * LG_BLOCK);
        const int nw = net ? LG_STEP_WAVES : waves_for(g.x, s->num_cus);
        const bool sc = s->P.self_collision != 0;
        if (net && !hf) { if (sc) hipLaunchKernelGGL((k_step<AnymalTraits, true, false, false, 4, true>), g, b, 0, st, a);
                          else hipLaunchKernelGGL((k_step<AnymalTraits, true, false>), g, b, 0, st, a); }
        else if (net && hf) { if (sc) hipLaunchKernelGGL((k_step<AnymalTraits, true, true, false, 4, true>), g, b, 0, st, a);
                              else hipLaunchKernelGGL((k_step<AnymalTraits, true, true>), g, b, 0, st, a); }
        else if (!hf) { if (sc) LAUNCH_NW(AnymalTraits, false, true) else LAUNCH_NW(AnymalTraits, false, false) }
        else { if (sc) LAUNCH_NW(AnymalTraits, true, true) else LAUNCH_NW(AnymalTraits, true, false) }
    } else {
        dim3 g(grid_for<CassieTraits>(s->P.num_envs));
        const int nw = waves_for(g.x, s->num_cus);
        if (!hf) LAUNCH_NW(CassieTraits, false, false)
        else LAUNCH_NW(CassieTraits, true, false)
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int lg_step_policy(lg_sim *s, lg_policy *p, const float *obs, float *actions, float *mean, uint64_t seed, int32_t deterministic,
                   int64_t common_step_counter, void *stream) {
    if (!s || !s->bound) return fail(-8, "lg_bind has not been called");
    if (int rc = status_error(s)) return rc;
    if (!p || !obs || !actions) return fail(-1, "null argument");
    if (common_step_counter < 0 && !s->B.step_counter) return fail(-9, "common_step_counter = -1 needs a step_counter buffer");
    const bool flat_actor = p->tiles[0] == 3 && p->tiles[1] == 8 && p->tiles[2] == 4 && p->tiles[3] == 2 && p->dims[4] == s->M.num_limbs * s->M.chain_len;
    if (!(s->kind == ROBOT_ANYMAL && s->P.control_type == LG_CTRL_ACTUATOR_NET && s->P.terrain_type != LG_TERRAIN_HEIGHTFIELD && flat_actor
          && p->dims[0] == s->P.num_obs))
        return fail(-4, "the fused policy step is compiled for the 48-128-64-32 actor on the quadruped actuator-net plane kernel; use lg_policy_act + lg_step");
    KArgs a; fill_args(s, a, common_step_counter); a.actions_in = nullptr;
    fill_policy_args(p, a.pol, obs, actions, mean, s->P.num_envs, seed, common_step_counter, s->B.step_counter, deterministic);
    if (s->P.self_collision)
        hipLaunchKernelGGL((k_step<AnymalTraits, true, false, true, 4, true>), dim3(grid_for<AnymalTraits>(s->P.num_envs)), dim3(LG_STEP_WAVES * LG_BLOCK),
                           0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((k_step<AnymalTraits, true, false, true>), dim3(grid_for<AnymalTraits>(s->P.num_envs)), dim3(LG_STEP_WAVES * LG_BLOCK),
                           0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int lg_rollout_policy(lg_sim *s, lg_policy *p, const lg_rollout_buffers *r, uint64_t seed, int32_t deterministic, int64_t common_step_counter, void *stream) {
    if (!s || !s->bound) return fail(-8, "lg_bind has not been called");
    if (int rc = status_error(s)) return rc;
    if (!p || !r || !r->obs || !r->actions || !r->rew || !r->dones || !r->time_outs) return fail(-1, "null argument");
    if (r->steps < 1 || r->steps > LG_MAX_ROLL_STEPS) return fail(-1, "steps must be in [1, LG_MAX_ROLL_STEPS]");
    if (common_step_counter < 0 && !s->B.step_counter) return fail(-9, "common_step_counter = -1 needs a step_counter buffer");
    const bool flat_actor = p->tiles[0] == 3 && p->tiles[1] == 8 && p->tiles[2] == 4 && p->tiles[3] == 2 && p->dims[4] == s->M.num_limbs * s->M.chain_len;
    if (!(s->kind == ROBOT_ANYMAL && s->P.control_type == LG_CTRL_ACTUATOR_NET && s->P.terrain_type != LG_TERRAIN_HEIGHTFIELD && flat_actor
          && p->dims[0] == s->P.num_obs && !s->P.measure_heights))
        return fail(-4, "the multi-step rollout kernel is compiled for the 48-128-64-32 actor on the quadruped actuator-net plane kernel; use lg_step_policy / lg_policy_act + lg_step");
    hipStream_t st = (hipStream_t)stream;
    const int stride = LG_NUM_REWARD_TERMS + 2;
    if (!s->d_roll_extras) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cap);
        if (cap != hipStreamCaptureStatusNone) return fail(-9, "first lg_rollout_policy call on a handle allocates its workspace: make one call outside stream capture");
        HIP_TRY(hipSetDevice(s->device));
        if (hipMalloc(&s->d_roll_extras, sizeof(float) * LG_MAX_ROLL_STEPS * stride) != hipSuccess) return fail(-10, "hipMalloc failed");
        HIP_TRY(hipMemset(s->d_roll_extras, 0, sizeof(float) * LG_MAX_ROLL_STEPS * stride));      // roll_finish leaves it zeroed after every segment
    }
    KArgs a; fill_args(s, a, common_step_counter); a.actions_in = nullptr; a.defer = 0;
    fill_policy_args(p, a.pol, r->obs, r->actions, r->mean, s->P.num_envs, seed, common_step_counter, s->B.step_counter, deterministic);
    a.roll.steps = r->steps; a.roll.obs = r->obs; a.roll.actions = r->actions; a.roll.mean = r->mean; a.roll.rew = r->rew;
    a.roll.done = r->dones; a.roll.time_outs = r->time_outs; a.roll.extras = s->d_roll_extras;
    a.roll.obs0 = r->obs0;
    if (s->P.self_collision)
        hipLaunchKernelGGL((k_step<AnymalTraits, true, false, true, 4, true, true>), dim3(grid_for<AnymalTraits>(s->P.num_envs)), dim3(LG_STEP_WAVES * LG_BLOCK), 0, st, a);
    else
        hipLaunchKernelGGL((k_step<AnymalTraits, true, false, true, 4, false, true>), dim3(grid_for<AnymalTraits>(s->P.num_envs)), dim3(LG_STEP_WAVES * LG_BLOCK), 0, st, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int lg_reset_idx(lg_sim *s, const int32_t *env_ids, int32_t count, int64_t common_step_counter, void *stream) {
    if (!s || !s->bound) return fail(-8, "lg_bind has not been called");
    if (int rc = status_error(s)) return rc;
    if (count <= 0) return 0;
    if (!env_ids) return fail(-1, "null env_ids");
    KArgs a; fill_args(s, a, common_step_counter); a.env_ids = env_ids; a.count = count;
    hipStream_t st = (hipStream_t)stream;
    const bool net = s->P.control_type == LG_CTRL_ACTUATOR_NET;
    if (s->kind == ROBOT_ANYMAL) {
        dim3 g(grid_for<AnymalTraits>(count)), b(LG_BLOCK);
        if (net) hipLaunchKernelGGL((k_reset<AnymalTraits, true>), g, b, 0, st, a);
        else hipLaunchKernelGGL((k_reset<AnymalTraits, false>), g, b, 0, st, a);
    } else {
        dim3 g(grid_for<CassieTraits>(count)), b(LG_BLOCK);
        hipLaunchKernelGGL((k_reset<CassieTraits, false>), g, b, 0, st, a);
    }
    hipLaunchKernelGGL(k_extras, dim3(1), dim3(64), 0, st, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int lg_device_status(lg_sim *s, int32_t synchronize) {
    if (!s) return fail(-1, "null argument");
    if (synchronize) { HIP_TRY(hipSetDevice(s->device)); HIP_TRY(hipDeviceSynchronize()); }
    const unsigned int st = __atomic_load_n(s->h_status, __ATOMIC_RELAXED);
    if (st) (void)status_error(s);                           /* leaves the text in lg_last_error() */
    return (int)(st & 0x7fffffffu);
}
int lg_clear_device_status(lg_sim *s) {
    if (!s) return fail(-1, "null argument");
    HIP_TRY(hipSetDevice(s->device)); HIP_TRY(hipDeviceSynchronize());
    __atomic_store_n(s->h_status, 0u, __ATOMIC_RELAXED);
    return 0;
}
int lg_debug_handover(lg_sim *s, int32_t skip, int32_t spin_limit) {
    if (!s) return fail(-1, "null argument");
    if (skip < 0 || skip > 2) return fail(-1, "skip must be 0 (off), 1 (frame flag) or 2 (self-collision flags)");
    s->debug_skip = skip; s->spin_limit = spin_limit > 0 ? spin_limit : (1 << 22);
    return 0;
}

int lg_set_deferred_extras(lg_sim *s, int32_t on) {
    if (!s) return fail(-1, "null argument");
    s->defer = on ? 1 : 0;
    return 0;
}

int lg_extras_flush(lg_sim *s, int64_t common_step_counter, void *stream) {
    if (!s || !s->bound) return fail(-8, "lg_bind has not been called");
    if (int rc = status_error(s)) return rc;
    KArgs a; fill_args(s, a, common_step_counter);
    // the level partial sums exist when the step kernel runs with helper waves (lg_step's choice of waves per workgroup)
    const int nwg = s->kind == ROBOT_ANYMAL ? grid_for<AnymalTraits>(s->P.num_envs) : grid_for<CassieTraits>(s->P.num_envs);
    const bool net = s->P.control_type == LG_CTRL_ACTUATOR_NET;
    const int nw = (s->kind == ROBOT_ANYMAL && net) ? LG_STEP_WAVES : waves_for((unsigned)nwg, s->num_cus);
    a.flush_parts = (nw > 1 && s->P.terrain_curriculum && s->B.terrain_levels) ? nwg : 0;
    hipLaunchKernelGGL(k_extras, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int lg_actuator_forward(lg_sim *s, const float *pos_err, const float *vel, float *torques, float *hidden, float *cell, int32_t rows, void *stream) {
    if (!s || !s->has_net) return fail(-2, "no actuator weights");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(k_actuator, dim3((rows + 63) / 64), dim3(64), 0, (hipStream_t)stream, s->d_weights, pos_err, vel, torques, hidden, cell, rows);
    HIP_TRY(hipGetLastError());
    return 0;
}

int lg_physics_substep(lg_sim *s, const float *torques, int32_t write_contacts, void *stream) {
    if (!s || !s->bound) return fail(-8, "lg_bind has not been called");
    if (int rc = status_error(s)) return rc;
    KArgs a; fill_args(s, a, 0);
    hipStream_t st = (hipStream_t)stream;
    const bool hf = s->P.terrain_type == LG_TERRAIN_HEIGHTFIELD;
    if (s->kind == ROBOT_ANYMAL) {
        dim3 g(grid_for<AnymalTraits>(s->P.num_envs)), b(LG_BLOCK);
        if (s->P.self_collision) {
            if (hf) hipLaunchKernelGGL((k_physics<AnymalTraits, true, true>), g, b, 0, st, a, torques, write_contacts);
            else hipLaunchKernelGGL((k_physics<AnymalTraits, false, true>), g, b, 0, st, a, torques, write_contacts);
        } else if (hf) hipLaunchKernelGGL((k_physics<AnymalTraits, true>), g, b, 0, st, a, torques, write_contacts);
        else hipLaunchKernelGGL((k_physics<AnymalTraits, false>), g, b, 0, st, a, torques, write_contacts);
    } else {
        dim3 g(grid_for<CassieTraits>(s->P.num_envs)), b(LG_BLOCK);
        if (hf) hipLaunchKernelGGL((k_physics<CassieTraits, true>), g, b, 0, st, a, torques, write_contacts);
        else hipLaunchKernelGGL((k_physics<CassieTraits, false>), g, b, 0, st, a, torques, write_contacts);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int lg_resample_reset_commands(lg_sim *s, int64_t common_step_counter, void *stream) {
    if (!s || !s->bound) return fail(-8, "lg_bind has not been called");
    if (int rc = status_error(s)) return rc;
    if (common_step_counter < 0) return fail(-9, "lg_resample_reset_commands needs the step's counter (the Philox key of its reset draws)");
    KArgs a; fill_args(s, a, common_step_counter);
    hipLaunchKernelGGL(k_resample_reset, dim3((s->P.num_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int lg_compute_observations_only(lg_sim *s, int64_t common_step_counter, void *stream) {
    if (!s || !s->bound) return fail(-8, "lg_bind has not been called");
    if (int rc = status_error(s)) return rc;
    KArgs a; fill_args(s, a, common_step_counter);
    hipStream_t st = (hipStream_t)stream;
    if (s->kind == ROBOT_ANYMAL) hipLaunchKernelGGL((k_obs<AnymalTraits>), dim3(grid_for<AnymalTraits>(s->P.num_envs)), dim3(LG_BLOCK), 0, st, a);
    else hipLaunchKernelGGL((k_obs<CassieTraits>), dim3(grid_for<CassieTraits>(s->P.num_envs)), dim3(LG_BLOCK), 0, st, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
