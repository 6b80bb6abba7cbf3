// qr_traj.h — part of the gfx950 quadrotor step library (included by quadrotor_kernels.hip, in this order).
// Goal generation (utils/trajectory_generator.py modes 0..6) and the SoA buffer accessor.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "quadrotor_hip.h"
#include "qr_dynamics.h"

namespace qr {

// ------------------------------------------------------------------------------------
// Goal generation: utils/trajectory_generator.py modes 0 and 1, per env.
// Generator state per env, stored as traj[8][N] = {calls, theta_init, b1d_x | w_b1d, b1d_y | smooth_term,
// x_init[3], -}.  Named scalars, not an array: the episode-start branch writes it under divergent
// control flow, and as an array hipcc kept two of its words in scratch memory.
// ------------------------------------------------------------------------------------
struct Traj {
  float calls = 0.0f, theta_init = 0.0f, p2 = 0.0f, p3 = 0.0f, x0 = 0.0f, x1 = 0.0f, x2 = 0.0f, p7 = 0.0f;
  __device__ __forceinline__ float get(int f) const { return f == 0 ? calls : f == 1 ? theta_init : f == 2 ? p2 : f == 3 ? p3 : f == 4 ? x0 : f == 5 ? x1 : f == 6 ? x2 : p7; }
  __device__ __forceinline__ void set(int f, float v) {
    if (f == 0) calls = v; else if (f == 1) theta_init = v; else if (f == 2) p2 = v; else if (f == 3) p3 = v;
    else if (f == 4) x0 = v; else if (f == 5) x1 = v; else if (f == 6) x2 = v; else p7 = v;
  }
};

// Modes 2-5 (take-off, landing, stay, circle: trajectory_generator.py:279-416) are STATEFUL: the reference object carries xd, vd, b1d,
// b1d_dot, Wd and five flags from call to call (a mode writes only the components it moves; manual mode returns before Wd is
// recomputed).  Here xd, vd, b1d, Wd persist in the env's goal buffer (QrEnv.goal: required for these modes), b1d_dot in traj words
// 2 and 7, the flags in word 3, x_init (= circle centre) in words 4..6.  Constants of the reference's __init__ (:81-94):
constexpr float kTakeoffEndHeight = -0.5f, kTakeoffVelocity = -0.05f, kLandingVelocity = 1.0f, kLandingCutoffHeight = -0.25f;
constexpr float kCircleRadius = 0.7f, kCircleLinearV = 0.4f, kCircleW = 0.4f, kNumCircles = 2.0f;
enum : int { kTrajStarted = 1, kTrajComplete = 2, kTrajManual = 4, kTrajManualInit = 8, kTrajLanded = 16 };

// mark_traj_start(state) (:176-204) + the episode-start branch of calculate_desired:
//   mode 0 (:141-148): b1d = Rz(theta) b1_proj, theta ~ U(+-25 deg)
//   mode 1 (:253-266): x_init = x, t_traj ~ U(2,5), smooth = -ln(0.001)/t_traj, w_b1d ~ U(+-0.15 pi)
// STATEFUL selects which family the caller compiled in: the stateless modes 0 / 1 / 6, or the stateful 2-5 (a separate set of step-kernel
// instantiations: with both in one kernel the stateless fused launches paid 8-22 % for code they never run, profiles/r04/ab_fused_goal_modes.txt).
template <bool STATEFUL, typename T, typename X>
__device__ __forceinline__ void traj_start(Work<T, X>& w, Traj& tr, int goal_mode, float theta_b1d, float t_traj, float w_b1d) {
  const T qw = w.q[0], qx = w.q[1], qy = w.q[2], qz = w.q[3];
  const float b1x = (float)(T(1) - T(2) * (qy * qy + qz * qz)), b1y = (float)(T(2) * (qx * qy + qw * qz));
  const float theta_init = atan2_fast(b1y, b1x);  // update_initial_state (:199-204)
  tr.calls = 0.0f;
  tr.theta_init = theta_init;
  tr.p7 = 0.0f;
  if constexpr (STATEFUL) {  // modes 2-5: no draws; flags cleared, x_init = x (:176-204), and the persistent fields of a FRESH generator
    tr.p2 = 0.0f; tr.p3 = 0.0f;       // (xd = vd = Wd = 0, b1d = e1, b1d_dot = 0: __init__ :54-55, 67-68 — the reference object would carry the
    tr.x0 = (float)w.x[0]; tr.x1 = (float)w.x[1]; tr.x2 = (float)w.x[2];   //  previous episode's b1d_dot into this one)
#pragma unroll
    for (int f = 0; f < 12; ++f) w.goal[f] = f == 6 ? 1.0f : 0.0f;
  } else if (goal_mode == QR_GOAL_MODE0) {
    float sn, cs;
    sincos_small(theta_init + theta_b1d, sn, cs);  // Rz(theta) (cos th_i, sin th_i, 0)
    tr.p2 = cs; tr.p3 = sn;
    tr.x0 = tr.x1 = tr.x2 = 0.0f;
  } else {  // mode 1: x_init + draws; mode 6: eight_shaped_center = x (:430), no draws
    tr.p2 = w_b1d;
    tr.p3 = 6.907755278982137f / t_traj;  // -ln(0.001) / t_traj
    tr.x0 = (float)w.x[0]; tr.x1 = (float)w.x[1]; tr.x2 = (float)w.x[2];
  }
}

// Draws of an episode start that the reset sampler leaves unused (word 19 of the env's Philox
// stream): mode 0 takes 24 bits for theta; mode 1 splits it 16/16 into t_traj and w_b1d.
__device__ __forceinline__ void traj_draws(uint32_t r19, float& theta_b1d, float& t_traj, float& w_b1d) {
  theta_b1d = (float)(25.0 * kPi / 180.0) * fmaf((float)(r19 >> 8), 0x1p-23f, 0x1p-24f - 1.0f);
  t_traj = 2.0f + 3.0f * fmaf((float)(r19 >> 16), 0x1p-16f, 0x1p-17f);
  w_b1d = (float)(0.15 * kPi) * fmaf((float)(r19 & 0xFFFFu), 0x1p-15f, 0x1p-16f - 1.0f);
}

// get_desired(state, mode) (:113-173) for the state in w: advances the call counter, fills
// w.goal = (xd, vd, b1d, Wd) and returns b1d_dot.
template <bool STATEFUL, typename T, typename X>
__device__ __forceinline__ void traj_goal(Work<T, X>& w, Traj& tr, int goal_mode, const Coeffs& c, float (&b1d_dot)[3]) {
  tr.calls += 1.0f;  // update_current_time (:224-229): t = t + dt on every call
  float b1d[3];
  bool keep_wd = false;  // (modes 2-5 in manual mode: calculate_desired returns before the Wd computation, :137-139)
  if constexpr (STATEFUL) {  // take-off / landing / stay / circle on the persistent fields (see the constants above)
    const float dtf = 2.0f * c.hdt, t = tr.calls * dtf;
    int fl = (int)tr.p3;
    const float xc[3] = {(float)w.x[0], (float)w.x[1], (float)w.x[2]}, vc[3] = {(float)w.v[0], (float)w.v[1], (float)w.v[2]};
    // get_current_b1 (:215-218): (cos theta, sin theta, 0), theta = atan2(b1[1], b1[0]) — the first column of R(q), flattened and normalised
    const T qw = w.q[0], qx = w.q[1], qy = w.q[2], qz = w.q[3];
    const float b1x = (float)(T(1) - T(2) * (qy * qy + qz * qz)), b1y = (float)(T(2) * (qx * qy + qw * qz));
    const float hinv = __builtin_amdgcn_rsqf(fmaxf(fmaf(b1x, b1x, b1y * b1y), 1e-30f));
    const float hx = b1x * hinv, hy = b1y * hinv;
    float xd[3] = {w.goal[0], w.goal[1], w.goal[2]}, vd[3] = {w.goal[3], w.goal[4], w.goal[5]};
    float bx = w.goal[6], by = w.goal[7], bdx = tr.p2, bdy = tr.p7;
    auto to_current = [&]() {  // set_desired_states_to_current (:209-212)
#pragma unroll
      for (int j = 0; j < 3; ++j) { xd[j] = xc[j]; vd[j] = vc[j]; }
      bx = hx; by = hy;
    };
    if (fl & kTrajManual) {  // manual() (:232-250): hold the position taken over at the switch, zero velocity, the heading of the switch
      if (!(fl & kTrajManualInit)) { to_current(); fl |= kTrajManualInit; }
      vd[0] = vd[1] = vd[2] = 0.0f;
      keep_wd = true;
    } else if (goal_mode == QR_GOAL_MODE2) {  // takeoff (:279-309)
      if (!(fl & kTrajStarted)) {
        xd[0] = xc[0]; xd[1] = xc[1]; xd[2] = 0.0f; vd[0] = vd[1] = vd[2] = 0.0f;
        bx = hx; by = hy;
        fl |= kTrajStarted;
      }
      const float t_traj = (kTakeoffEndHeight - tr.x2) / kTakeoffVelocity;
      if (t < t_traj) {
        xd[2] = fmaf(kTakeoffVelocity, t, tr.x2);
      } else {
        const float d0 = xd[0] - xc[0], d1 = xd[1] - xc[1], d2 = xd[2] - xc[2];
        if (fmaf(d0, d0, fmaf(d1, d1, d2 * d2)) < 0.04f * 0.04f) {  // waypoint_reached (:312-318)
          xd[2] = kTakeoffEndHeight; vd[2] = 0.0f;
          fl |= kTrajComplete | kTrajManual;  // mark_traj_end(True)
        }
      }
    } else if (goal_mode == QR_GOAL_MODE3) {  // land (:321-349)
      if (!(fl & kTrajStarted)) { to_current(); fl |= kTrajStarted; }
      const float t_traj = (kLandingCutoffHeight - tr.x2) / kLandingVelocity;
      if (t < t_traj) {
        xd[2] = fmaf(kLandingVelocity, t, tr.x2);
      } else if (xc[2] > kLandingCutoffHeight) {
        xd[2] = kLandingCutoffHeight; vd[2] = 0.0f;
        fl |= kTrajComplete | kTrajLanded;  // mark_traj_end(False)
      } else {
        xd[2] = kLandingCutoffHeight; vd[2] = kLandingVelocity;
      }
    } else if (goal_mode == QR_GOAL_MODE4) {  // stay (:352-357)
      if (!(fl & kTrajStarted)) { to_current(); fl |= kTrajStarted; }
      fl |= kTrajComplete | kTrajManual;
    } else {  // circle (:360-416): run-up along +x, num_circles circles about the start position, then manual
      if (!(fl & kTrajStarted)) { to_current(); fl |= kTrajStarted; }
      constexpr float run_up = kCircleRadius / kCircleLinearV;
      constexpr float t_traj = run_up + kNumCircles * 2.0f * (float)kPi / kCircleW;
      // (run_up = 350 dt EXACTLY; the reference's accumulated float64 t is 1.7499999999999847 at that call — still the run-up)
      if (t < run_up * 1.000001f) {
        xd[0] = fmaf(kCircleLinearV, t, tr.x0); vd[0] = kCircleLinearV;
      } else if (t < t_traj) {
        float sn, cs;
        sincos_small(kCircleW * (t - run_up), sn, cs);
        xd[0] = fmaf(kCircleRadius, cs, tr.x0); vd[0] = -kCircleRadius * kCircleW * sn;
        xd[1] = fmaf(kCircleRadius, sn, tr.x1); vd[1] = kCircleRadius * kCircleW * cs;
        bx = -cs; by = -sn;                                   // heading angle th + pi
        bdx = kCircleW * sn; bdy = -kCircleW * cs;
      } else {
        fl |= kTrajComplete | kTrajManual;
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) { w.goal[j] = xd[j]; w.goal[3 + j] = vd[j]; }
    b1d[0] = bx; b1d[1] = by; b1d[2] = 0.0f;
    b1d_dot[0] = bdx; b1d_dot[1] = bdy; b1d_dot[2] = 0.0f;
    tr.p2 = bdx; tr.p7 = bdy; tr.p3 = (float)fl;
  } else if (goal_mode == QR_GOAL_MODE0) {  // set_desired_states_to_zero + the b1d drawn at episode start
#pragma unroll
    for (int j = 0; j < 6; ++j) w.goal[j] = 0.0f;
    b1d[0] = tr.p2; b1d[1] = tr.p3; b1d[2] = 0.0f;
    b1d_dot[0] = b1d_dot[1] = b1d_dot[2] = 0.0f;
  } else if (goal_mode == QR_GOAL_MODE6) {  // eight_shaped_curve (:418-505)
    const float t = fminf(tr.calls * (2.0f * c.hdt), c.e8_tmax);
    const float ek = expf(-c.e8_k * t);
    const float e = 1.0f - ek, de = c.e8_k * ek;  // exp_term, d/dt exp_term
    float s1, c1, s2, c2;
    sincos_small(c.e8_w1 * t, s1, c1);
    sincos_small(c.e8_w2 * t, s2, c2);
    const float za = 0.5f * (tr.x2 - c.e8_alt);  // synchronised altitude command (:487-492)
    w.goal[0] = fmaf(c.e8_A2 * s2, e, tr.x0);
    w.goal[1] = fmaf(c.e8_A1 * (c1 - 1.0f), e, tr.x1);
    w.goal[2] = fmaf(za, 1.0f - c1, tr.x2);
    w.goal[3] = c.e8_A2 * (c.e8_w2 * c2 * e + s2 * de);
    w.goal[4] = c.e8_A1 * (-c.e8_w1 * s1 * e + (c1 - 1.0f) * de);
    w.goal[5] = za * c.e8_w1 * s1;
    const float term = fmaf(c.e8_wb * t, e, tr.theta_init), dterm = c.e8_wb * (e + t * de);  // yaw (:494-498)
    float sn, cs;
    sincos_small(term, sn, cs);
    b1d[0] = cs; b1d[1] = sn; b1d[2] = 0.0f;
    b1d_dot[0] = -sn * dterm; b1d_dot[1] = cs * dterm; b1d_dot[2] = 0.0f;
  } else {  // hovering (:268-277), x_goal = 0
    const float t = tr.calls * (2.0f * c.hdt);
    const float wb = tr.p2, sm = tr.p3;
    const float e = expf(-sm * t);
    const float xi[3] = {tr.x0, tr.x1, tr.x2};
#pragma unroll
    for (int j = 0; j < 3; ++j) { w.goal[j] = xi[j] * e; w.goal[3 + j] = -xi[j] * sm * e; }
    float sn, cs;
    sincos_small(fmaf(wb, t, tr.theta_init), sn, cs);
    b1d[0] = cs; b1d[1] = sn; b1d[2] = 0.0f;
    b1d_dot[0] = -wb * sn; b1d_dot[1] = wb * cs; b1d_dot[2] = 0.0f;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) w.goal[6 + j] = b1d[j];
  // Wd = (0, 0, b3 . (b1c x b1c_dot)) with b3' = R hat(W) e3 = W2 b1 - W1 b2 (:165-172)
  T R[9];
  quat_to_R(w.q, R);
  const T W1 = w.W[0], W2 = w.W[1];
  T b3d[3], b1c[3], b1cd[3];
  const T d0 = T(b1d[0]), d1 = T(b1d[1]), d2 = T(b1d[2]);
  const T dd0 = T(b1d_dot[0]), dd1 = T(b1d_dot[1]), dd2 = T(b1d_dot[2]);
#pragma unroll
  for (int j = 0; j < 3; ++j) b3d[j] = W2 * R[j] - W1 * R[3 + j];
  const T b1d_b3 = d0 * R[6] + d1 * R[7] + d2 * R[8];
  const T b1dd_b3 = dd0 * R[6] + dd1 * R[7] + dd2 * R[8];
  const T b1d_b3d = d0 * b3d[0] + d1 * b3d[1] + d2 * b3d[2];
  const T dv[3] = {d0, d1, d2}, ddv[3] = {dd0, dd1, dd2};
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    b1c[j] = dv[j] - b1d_b3 * R[6 + j];
    b1cd[j] = ddv[j] - (b1dd_b3 * R[6 + j] + b1d_b3d * R[6 + j] + b1d_b3 * b3d[j]);
  }
  const T oc0 = b1c[1] * b1cd[2] - b1c[2] * b1cd[1];
  const T oc1 = b1c[2] * b1cd[0] - b1c[0] * b1cd[2];
  const T oc2 = b1c[0] * b1cd[1] - b1c[1] * b1cd[0];
  if (!keep_wd) {
    w.goal[9] = 0.0f; w.goal[10] = 0.0f;
    w.goal[11] = (float)(R[6] * oc0 + R[7] * oc1 + R[8] * oc2);
  }
}

__device__ __forceinline__ float sq3(const float* v) { return v[0] * v[0] + v[1] * v[1] + v[2] * v[2]; }
__device__ __forceinline__ bool out3(const float* v) { return !(fabsf(v[0]) < 1.0f) | !(fabsf(v[1]) < 1.0f) | !(fabsf(v[2]) < 1.0f); }
__device__ __forceinline__ float interp01(float r, float rmin, float inv_nrmin) { return clampT((r - rmin) * inv_nrmin, 0.0f, 1.0f); }

// ---- SoA access through buffer resources ------------------------------------------------
// Field f of env (first + lane) of a [F][L] buffer lives at byte (f*L + first + lane)*sizeof(E).
// `first` and L are wave-uniform, so the access is issued as
//     buffer_load/store  vdata, voffset = lane*sizeof(E), s[rsrc], soffset = (f*L + first)*sizeof(E)
// with the 128-bit descriptor and soffset in SGPRs (built once per wave by the SALU).  The
// equivalent global_load through a pointer makes hipcc chain 64-bit VALU address arithmetic
// per access (v_mad_u64_u32 / v_lshl_add_u64: ~110 of the ~1100 instructions of the step).
// soffset is 32-bit: every SoA buffer must be < 4 GiB (checked on the host, QR_E_SIZE).
// A null base gives a descriptor without records: its loads return 0 and touch no memory (the range check of raw
// buffer accesses), which spares optional buffers a branch.  Any other buffer gets the maximum record count: every
// access is in bounds by construction (lanes past a ragged tail are clamped onto the last env, stores are guarded), and
// an exact byte count would cost each of the step's seven descriptors half a dozen scalar instructions.
typedef int v2i_t __attribute__((ext_vector_type(2)));
// (Cache-policy bits on the LOADS were measured too, profiles/r03/ab_load_policy.txt: sc0 / sc1 equal, nt +0.5 us per launch at
// 65 536 envs — non-temporal data does not stay in the Infinity Cache for the next launch.  Plain loads.)

template <typename E>
struct SoA {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned L;  // elements between fields
  __device__ __forceinline__ SoA(const void* base, int fields, int64_t ld)
      : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, base == nullptr ? 0 : 0x7fffffff, 0x00020000)),
        L((unsigned)ld) {}
  __device__ __forceinline__ unsigned soff(int f, unsigned first) const { return ((unsigned)f * L + first) * (unsigned)sizeof(E); }
  __device__ __forceinline__ E load(int f, unsigned first, unsigned lane) const {
    if constexpr (sizeof(E) == 4) {
      return __builtin_bit_cast(E, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4u, soff(f, first), 0));
    } else {
      return __builtin_bit_cast(E, __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane * 8u, soff(f, first), 0));
    }
  }
  template <int AUX = 0>  // cache policy (qr_args.h: QR_HELP_AUX)
  __device__ __forceinline__ void store(int f, unsigned first, unsigned lane, E v) const {
    if constexpr (sizeof(E) == 4) {
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rsrc, lane * 4u, soff(f, first), AUX);
    } else {
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i_t, v), rsrc, lane * 8u, soff(f, first), AUX);
    }
  }
};

// ---- attitude in memory: "smallest three" ---------------------------------------------------
// The step is bound by bytes, so the unit quaternion is stored as THREE components: the one of
// largest magnitude is dropped (and made positive: q and -q are the same rotation) and rebuilt on
// load as sqrt(1 - k0^2 - k1^2 - k2^2) >= 1/2, which amplifies rounding by at most sqrt(3).  Which
// component was dropped (2 bits) rides in the two lowest mantissa bits of k0 — a 4e-16 (float64)
// perturbation.  -8 B read and -8 B written per env-step (-16 of 205 in the default layout).
// A K-step rollout keeps the state in registers, so every env-step ends with pack -> unpack: what the
// registers then hold is exactly what a single-step launch would have stored and re-loaded.
template <typename T>
struct QuatPack { T k[3]; };

template <typename T>
__device__ __forceinline__ void pack_quat(const T (&q)[4], QuatPack<T>& p) {
  using U = typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type;
  const T a0 = fabs(q[0]), a1 = fabs(q[1]), a2 = fabs(q[2]), a3 = fabs(q[3]);
  const bool b01 = a1 > a0, b23 = a3 > a2;
  const T m01 = b01 ? a1 : a0, m23 = b23 ? a3 : a2;
  const bool hi = m23 > m01;
  const int idx = hi ? (b23 ? 3 : 2) : (b01 ? 1 : 0);  // ties: the lowest index
  const T d = hi ? (b23 ? q[3] : q[2]) : (b01 ? q[1] : q[0]);
  const T k0 = idx == 0 ? q[1] : q[0], k1 = idx <= 1 ? q[2] : q[1], k2 = idx <= 2 ? q[3] : q[2];
  const bool flip = d < T(0);
  p.k[0] = flip ? -k0 : k0; p.k[1] = flip ? -k1 : k1; p.k[2] = flip ? -k2 : k2;
  p.k[0] = __builtin_bit_cast(T, (__builtin_bit_cast(U, p.k[0]) & ~U(3)) | U(idx));
}

template <typename T>
__device__ __forceinline__ void unpack_quat(const QuatPack<T>& p, T (&q)[4]) {
  using U = typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type;
  const U bits = __builtin_bit_cast(U, p.k[0]);
  const int idx = (int)(bits & U(3));
  const T k0 = __builtin_bit_cast(T, bits & ~U(3)), k1 = p.k[1], k2 = p.k[2];
  const T s = fmaT(-k2, k2, fmaT(-k1, k1, fmaT(-k0, k0, T(1))));  // in [1/4, 1] for a unit quaternion
  T w;
  if constexpr (sizeof(T) == 8) {  // sqrt(s) = s rsqrt(s): hardware seed, two Newton steps, one correction of the product
    double r = __builtin_amdgcn_rsq(s);
    r = r * fma(-0.5 * s, r * r, 1.5);
    r = r * fma(-0.5 * s, r * r, 1.5);
    w = s * r;
    w = fma(0.5 * r, fma(-w, w, s), w);
  } else {
    float r = __builtin_amdgcn_rsqf(s);
    r = r * fmaf(-0.5f * s, r * r, 1.5f);
    w = s * r;
    w = fmaf(0.5f * r, fmaf(-w, w, s), w);
  }
  q[0] = idx == 0 ? w : k0;
  q[1] = idx == 1 ? w : (idx == 0 ? k0 : k1);
  q[2] = idx == 2 ? w : (idx == 3 ? k2 : k1);
  q[3] = idx == 3 ? w : k2;
}

// Issue order = arrival order (a wave's vector loads complete in order, and at 65 536 envs the load phase of a launch is
// bound by what a CU can pull in, ~11 B per cycle: the LAST word of a wave's 112 B lands ~0.7 us after the first): W and
// the attitude first — the integration starts from them — x and v last: they enter the step only in its final update.
template <typename XV, typename QW>
__device__ __forceinline__ void load_state_raw(const Args& a, int64_t first64, unsigned lane, Work<QW, XV>& w, QuatPack<QW>& p) {
  const SoA<XV> pv(a.pos_vel, 6, a.ld);
  const SoA<QW> ar(a.att_rate, 6, a.ld);
  const unsigned first = (unsigned)first64;
#pragma unroll
  for (int f = 0; f < 3; ++f) p.k[f] = ar.load(f, first, lane);
#pragma unroll
  for (int f = 0; f < 3; ++f) w.W[f] = ar.load(3 + f, first, lane);
#pragma unroll
  for (int f = 0; f < 3; ++f) { w.x[f] = pv.load(f, first, lane); w.v[f] = pv.load(3 + f, first, lane); }
}

template <typename XV, typename QW>
__device__ __forceinline__ void load_state(const Args& a, int64_t first64, unsigned lane, Work<QW, XV>& w) {
  QuatPack<QW> p;
  load_state_raw<XV, QW>(a, first64, lane, w, p);
  unpack_quat(p, w.q);
}

// (the packed attitude is passed in: the step kernel forms it once per env-step, see QuatPack)
template <typename XV, typename QW, int AUX = 0>
__device__ __forceinline__ void store_state(const Args& a, int64_t first64, unsigned lane, const Work<QW, XV>& w, const QuatPack<QW>& p) {
  const SoA<XV> pv(a.pos_vel, 6, a.ld);
  const SoA<QW> ar(a.att_rate, 6, a.ld);
  const unsigned first = (unsigned)first64;
#pragma unroll
  for (int f = 0; f < 3; ++f) ar.template store<AUX>(f, first, lane, p.k[f]);
#pragma unroll
  for (int f = 0; f < 3; ++f) ar.template store<AUX>(3 + f, first, lane, w.W[f]);
#pragma unroll
  for (int f = 0; f < 3; ++f) { pv.template store<AUX>(f, first, lane, w.x[f]); pv.template store<AUX>(3 + f, first, lane, w.v[f]); }
}

template <typename XV, typename QW>
__device__ __forceinline__ void store_state(const Args& a, int64_t first64, unsigned lane, const Work<QW, XV>& w) {
  QuatPack<QW> p;
  pack_quat(w.q, p);
  store_state<XV, QW>(a, first64, lane, w, p);
}

template <typename T, typename X>
__device__ __forceinline__ void idle_work(Work<T, X>& w, const Coeffs& c) {  // lanes past the ragged tail
#pragma unroll
  for (int f = 0; f < 3; ++f) { w.x[f] = X(0); w.v[f] = X(0); w.W[f] = T(0); }
#pragma unroll
  for (int f = 0; f < 4; ++f) w.q[f] = T(f == 0 ? 1 : 0);
#pragma unroll
  for (int f = 0; f < 12; ++f) w.goal[f] = f == 6 ? 1.0f : 0.0f;
#pragma unroll
  for (int f = 0; f < 8; ++f) w.integ[f] = 0.0f;
#pragma unroll
  for (int f = 0; f < 6; ++f) w.prm[f] = c.nom_f[f];
  w.nominal = true;
}

}  // namespace qr
