// quadrotor_kernels.hip — fused env.step() kernels for gfx950 (MI355X / CDNA4).
//
// One lane = one quadrotor.  A launch does, per env and per env-step, everything the
// reference's QuadEnv.step template does (gym_rotor/envs/quad.py:142-168):
//   action map / motor mixing -> S fixed RK4 substeps of the rigid-body ODE on
//   R^3 x R^3 x SO(3) x R^3 (quad.py:321-335) with zero-order-hold (f, M) -> error
//   observation + trapezoid integrators (quad.py:421-466) -> reward -> np.interp
//   normalisation -> done -> crash override [-> auto-reset].
//
// What bounds it: measured on MI355X the step moves its working set through the fabric at
// the HBM rate even at N = 65 536 (dirty L2 lines are written back at every kernel
// boundary), and all arithmetic fits under that, so the design minimises BYTES per env:
//   * attitude is a unit quaternion (4 words) integrated directly — q' = q (0,W)/2 is the
//     same flow as R' = R hat(W) — so the state is 13 words, not 18; R(q) is rebuilt in
//     registers only where the observation / reward needs it;
//   * x, v are stored as float32 and q, W as float64 in the default (mixed) layout: fp32
//     rounding of W and R is what breaks the 1e-5 / 1000-step parity bar, x and v do not
//     feed back into the rotation;
//   * the whole working set stays in VGPRs across substeps and, in qr_rollout, across
//     env-steps: HBM is touched once in and once out.
// Per-env SoA buffers are read/written with lane-contiguous accesses; caller-facing AoS
// rows (actions, observations) go through LDS so global traffic is linear 16-byte-per-lane
// stores.  There is no contraction larger than 3x3 anywhere, so no MFMA.
//
// Written directly for CDNA4: 64-lane wavefronts, one wavefront per workgroup (N = 65 536 ->
// 1024 workgroups = one per SIMD; no cross-wave barriers), <= 256 VGPRs so that two waves fit
// per SIMD at large N.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "quadrotor_hip.h"

#ifndef QR_ABLATE
#define QR_ABLATE 0  // 0 = product build; 1..4 = measurement-only builds (tools/microbench.py)
#endif
#ifndef QR_WAVES_PER_SIMD
#define QR_WAVES_PER_SIMD 2  // 2nd __launch_bounds__ argument of the step kernel (<= 256 VGPRs)
#endif

namespace qr {

// ------------------------------------------------------------------------------------
// Kernel argument block (passed by value in kernarg memory)
// ------------------------------------------------------------------------------------
struct Coeffs {  // double-precision copy of QrCoeffs + derived reward floors
  double Cx, CIx, Cv, Cb1, CIb1, CW, Cw12, CW3, alpha, beta, dt;
  double x_lim, v_lim, W_lim, eIx_lim, eIb1_lim;
  double sin_euler_lim, tan_euler_lim, udm;
  double rmin_mono, rmin_1, rmin_2;
  // reciprocals formed once on the host (an f64 division costs ~35 VALU slots on the device)
  double inv_x_lim, inv_v_lim, inv_W_lim, inv_eIx_lim, inv_eIb1_lim, inv_nrmin_mono, inv_nrmin_1, inv_nrmin_2;
  // eight-shaped curve (trajectory_generator.py:98-110, 418-505)
  float e8_w1, e8_w2, e8_k, e8_A1, e8_A2, e8_wb, e8_alt, e8_tmax;
  double inv_w_adapt;  // 1 / w_adapt, 0 = fixed substep count
};

struct ActorW {  // QrActor's tensors (torch.nn.Linear layout: weight [out][in])
  const float *fc1_w, *fc1_b, *fc2_w, *fc2_b, *mean_w, *mean_b, *log_std;
};

struct Args {
  // per-env buffers
  void* pos_vel;
  void* att_rate;
  float* integ;
  float* params;
  float* goal;
  float* traj;
  int32_t* episode;
  int32_t* steps;
  // per-call
  const float* action;
  float* obs0;
  float* obs1;
  float* reward;
  float* reward_raw;
  uint8_t* done;
  uint8_t* truncated;
  const uint8_t* mask;
  double* rows_out;       // qr_get_state
  const double* rows_in;  // qr_set_state
  const float* draws;     // qr_traj_start: injected [3][N] theta_b1d, t_traj, w_b1d
  float* goal_rows;       // qr_get_desired: [N][15]
  int32_t goal_mode;
  int32_t store_goal;
  int64_t n;
  int64_t ld;             // elements between consecutive fields of every SoA buffer (>= n)
  int64_t env_offset;
  uint64_t seed;
  int32_t n_steps;
  int32_t substeps;
  int32_t max_episode_steps;
  uint32_t flags;
  // qr_rollout_actor: the policy in the loop
  ActorW actor[2];
  const float* obs0_in;
  const float* obs1_in;
  const float* noise;
  float* act_out;
  float* logp_out;
  uint64_t noise_seed;
  uint64_t step_base;
  float max_action;
  int32_t deterministic;
  Coeffs c;
};

// Nominal parameters (quad.py:28-33)
constexpr double kMnom = 2.15, kDnom = 0.23, kJ1nom = 0.022, kJ3nom = 0.035, kCtfNom = 0.0135,
                 kCtwNom = 2.2, kG = 9.81, kMinForce = 0.5;
constexpr double kPi = 3.14159265358979323846;

template <typename T> __device__ __forceinline__ T clampT(T v, T lo, T hi) { return v < lo ? lo : (v > hi ? hi : v); }

// 1/a for well-scaled positive a (masses, inertias): hardware seed + Newton steps instead of
// the ~35-instruction IEEE f64 division expansion.  Relative error <= 2 ulp.
__device__ __forceinline__ double recip(double a) {
  double x = __builtin_amdgcn_rcp(a);
  x = fma(fma(-a, x, 1.0), x, x);
  x = fma(fma(-a, x, 1.0), x, x);
  return x;
}
__device__ __forceinline__ float recip(float a) {
  float x = __builtin_amdgcn_rcpf(a);
  return fmaf(fmaf(-a, x, 1.0f), x, x);
}

// Per-env working set held in VGPRs.  y = (v[0..2], q[3..6] = w,x,y,z, W[7..9]) is the RK4
// vector; x' = v is integrated from the stage velocities.  Everything that is STORED as
// float32 is also HELD as float32 (converted at use): the step kernel is register-bound —
// two waves per SIMD need <= 256 VGPRs — and a float64 copy of 26 words costs 26 registers.
template <typename T>
struct Work {
  T x[3];
  T y[10];
  float prm[6];    // m, d, J1(=J2), J3, c_tf, c_tw (quad.py:359-387); kNominal[] when not randomised
  float goal[12];  // xd, vd, b1d, Wd
  float integ[8];  // eIx, g_x prev, eIb1, g_b prev
  bool nominal;    // parameters are the exact float64 nominal values, not prm[]
};

template <typename T>
struct Phys {  // what set_random_parameters derives (quad.py:389-404), formed when needed
  T m, d, J1, J3, ctf, ctw;
  T max_force, avrg_act, scale_act;
  template <typename W>
  __device__ __forceinline__ explicit Phys(const W& w) {
    if (w.nominal) {
      m = T(kMnom); d = T(kDnom); J1 = T(kJ1nom); J3 = T(kJ3nom); ctf = T(kCtfNom); ctw = T(kCtwNom);
    } else {
      m = T(w.prm[0]); d = T(w.prm[1]); J1 = T(w.prm[2]); J3 = T(w.prm[3]); ctf = T(w.prm[4]); ctw = T(w.prm[5]);
    }
    const T hover = m * T(kG * 0.25);
    max_force = ctw * hover;
    avrg_act = (T(kMinForce) + max_force) * T(0.5);
    scale_act = max_force - avrg_act;
  }
};

// ------------------------------------------------------------------------------------
// Philox4x32-10 counter-based RNG (Salmon et al., SC'11): stateless, keyed by
// (seed, global env id, episode) so that resets do not depend on launch geometry.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t (&ctr)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // separate v_mul_hi_u32 / v_mul_lo_u32: the 64-bit product form compiles to v_mad_u64_u32,
    // which measures ~2x slower than the pair on gfx950
    const uint32_t hi0 = __umulhi(0xD2511F53u, ctr[0]), lo0 = 0xD2511F53u * ctr[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, ctr[2]), lo1 = 0xCD9E8D57u * ctr[2];
    const uint32_t n0 = hi1 ^ ctr[1] ^ k0;
    const uint32_t n1 = lo1;
    const uint32_t n2 = hi0 ^ ctr[3] ^ k1;
    const uint32_t n3 = lo0;
    ctr[0] = n0; ctr[1] = n1; ctr[2] = n2; ctr[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

struct Draws {  // 20 x 32 random bits -> uniforms (24-bit mantissa: these are random draws, float is plenty)
  uint32_t r[20];
  __device__ __forceinline__ float u01(int i) const { return fmaf((float)(r[i] >> 8), 0x1p-24f, 0x1p-25f); }
  __device__ __forceinline__ float sym(int i) const { return fmaf((float)(r[i] >> 8), 0x1p-23f, 0x1p-24f - 1.0f); }
};

__device__ __forceinline__ void draw20(Draws& d, uint64_t seed, uint64_t gid, uint32_t episode) {
#pragma unroll
  for (int b = 0; b < 5; ++b) {
    uint32_t ctr[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), episode, (uint32_t)b};
    philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
    for (int j = 0; j < 4; ++j) d.r[4 * b + j] = ctr[j];
  }
}

// Wave-cooperative form for the in-step auto-reset.  Only ~1 % of the envs reset in a given
// step, but a wave runs the reset path if ANY of its 64 lanes needs it and the kernel ends with
// its slowest wave, so what counts is the instruction count of the path — and Philox is the
// bulk of it (v_mul_hi/lo_u32 are quarter-rate).  Instead of each resetting lane grinding
// through 5 Philox blocks with the rest of the wave idle, ONE Philox pass serves up to 12
// resetting envs: lane 5k+b computes block b of the k-th resetting env, then each owner pulls
// its 20 words with ds_bpermute.  Same draws as draw20.
__device__ __forceinline__ void coop_draw20(Draws& d, bool need, uint64_t seed, uint64_t gid, uint32_t episode) {
  const int lane = (int)__lane_id();
  const int glo = (int)(uint32_t)gid, ghi = (int)(uint32_t)(gid >> 32), ep = (int)episode;
#pragma unroll
  for (int j = 0; j < 20; ++j) d.r[j] = 0u;
  unsigned long long m = __ballot(need);
  const int my_rank = __popcll(m & ((1ull << lane) - 1ull));  // rank among the resetting lanes
  const int k = lane / 5, b = lane - 5 * k;                   // slot / block of this lane (k = 12: idle)
  int base = 0;
  while (m) {  // wave-uniform; one pass unless > 12 lanes of this wave reset
    int src = lane, cnt = 0;
    for (int s = 0; s < 12 && m; ++s) {  // lane index of the s-th resetting env -> lanes of slot s
      const int l = __builtin_ctzll(m);
      m &= m - 1;
      if (k == s) src = l;
      ++cnt;
    }
    uint32_t ctr[4] = {(uint32_t)__shfl(glo, src), (uint32_t)__shfl(ghi, src), (uint32_t)__shfl(ep, src), (uint32_t)b};
    philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
    const int r = my_rank - base;
    const bool mine = need && r >= 0 && r < cnt;
    const int from4 = (mine ? 5 * r : 0) << 2;  // ds_bpermute takes a byte address (lane * 4)
    // All 20 cross-lane reads are issued back to back and waited for once: written as
    // "read, select, read, select, ..." hipcc puts an s_waitcnt lgkmcnt(0) behind every
    // ds_bpermute and the ~100-cycle LDS-crossbar latency is paid 20 times in series.
    int got[20];
#pragma unroll
    for (int bb = 0; bb < 5; ++bb) {
#pragma unroll
      for (int j = 0; j < 4; ++j) got[4 * bb + j] = __builtin_amdgcn_ds_bpermute(from4 + 4 * bb, (int)ctr[j]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int w = 0; w < 20; ++w) d.r[w] = mine ? (uint32_t)got[w] : d.r[w];
    base += cnt;
  }
}

// sin and cos of a float angle of moderate size (|x| < ~1e3): Cody-Waite reduction by pi/2 and
// the cephes minimax polynomials on [-pi/4, pi/4]; ~1e-7 absolute.  Branch-free and small: the
// OCML sincosf drags its Payne-Hanek slow path (and its registers) into every kernel using it.
__device__ __forceinline__ void sincos_small(float x, float& sn, float& cs) {
  const float k = rintf(x * 0.63661977236758134f);
  float r = fmaf(-k, 1.5707962512969971f, x);
  r = fmaf(-k, 7.5497894158615964e-08f, r);
  const float r2 = r * r;
  const float ps = fmaf(r * r2, fmaf(r2, fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
  const float pc = fmaf(r2 * r2, fmaf(r2, fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                        fmaf(-0.5f, r2, 1.0f));
  const int q = (int)k;
  const float s0 = (q & 1) ? pc : ps, c0 = (q & 1) ? ps : pc;
  sn = (q & 2) ? -s0 : s0;
  cs = ((q + 1) & 2) ? -c0 : c0;
}

// sin/cos of a random angle re-normalised in f64, so every factor, hence q, has unit norm to
// f64 round-off.
__device__ __forceinline__ void unit_sincos(float ang, double& s, double& c) {
  float sf, cf;
  sincos_small(ang, sf, cf);
  s = (double)sf; c = (double)cf;
#pragma unroll
  for (int it = 0; it < 2; ++it) {  // r = 1/sqrt(n2) to first order around 1: 1e-6 -> 1e-12 -> 1e-24
    const double r = 1.5 - 0.5 * (s * s + c * c);
    s *= r; c *= r;
  }
}

// QuadEnv.reset + sample_init_error + set_random_parameters (quad.py:171-222, 338-404).
// Draw order: 0..5 m,d,J1,J3,c_tf,c_tw; 6 yaw; 7 zero-error branch; 8..10 x; 11..13 v;
// 14..16 W; 17,18 roll,pitch.  R = Rz(yaw) Ry(pitch) Rx(roll) (scipy 'xyz' extrinsic,
// quad.py:199)  <=>  q = qz(yaw) qy(pitch) qx(roll).
template <typename T>
__device__ void sample_reset(Work<T>& w, const Draws& d, bool randomise, bool eval, const Coeffs& c) {
  if (randomise) {  // float32 values: that is how the params buffer stores them
    const float p = (float)c.udm;
    w.prm[0] = (float)kMnom * fmaf(p, d.sym(0), 1.0f);
    w.prm[1] = (float)kDnom * fmaf(p, d.sym(1), 1.0f);
    w.prm[2] = (float)kJ1nom * fmaf(p, d.sym(2), 1.0f);
    w.prm[3] = (float)kJ3nom * fmaf(p, d.sym(3), 1.0f);
    w.prm[4] = (float)kCtfNom * fmaf(p, d.sym(4), 1.0f);
    w.prm[5] = (float)kCtwNom * fmaf(0.5f * p, d.sym(5), 1.0f);
    w.nominal = false;
  } else {
    w.prm[0] = (float)kMnom; w.prm[1] = (float)kDnom; w.prm[2] = (float)kJ1nom;
    w.prm[3] = (float)kJ3nom; w.prm[4] = (float)kCtfNom; w.prm[5] = (float)kCtwNom;
    w.nominal = true;
  }
  const float yaw = (float)kPi * d.sym(6);
  float ix, iv, iR, iW;
  if (eval) {  // quad.py:352-356
    ix = 0.4f; iv = 0.0f; iR = 0.0f; iW = 0.0f;
  } else if (d.u01(7) < 0.2f) {  // quad.py:342-346
    ix = 0.0f; iv = 0.0f; iR = 0.0f; iW = 0.0f;
  } else {  // quad.py:348-351
    ix = 0.6f; iv = (float)(c.v_lim * 0.5); iR = (float)(50.0 * kPi / 180.0); iW = (float)(c.W_lim * 0.5);
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    w.x[j] = T(ix * d.sym(8 + j));
    w.y[j] = T(iv * d.sym(11 + j));
    w.y[7 + j] = T(iW * d.sym(14 + j));
  }
  double sr, cr, sp, cp, sy, cy;
  unit_sincos(0.5f * iR * d.sym(17), sr, cr);
  unit_sincos(0.5f * iR * d.sym(18), sp, cp);
  unit_sincos(0.5f * yaw, sy, cy);
  w.y[3] = T(cr * cp * cy + sr * sp * sy);
  w.y[4] = T(sr * cp * cy - cr * sp * sy);
  w.y[5] = T(cr * sp * cy + sr * cp * sy);
  w.y[6] = T(cr * cp * sy - sr * sp * cy);
}

// ------------------------------------------------------------------------------------
// Attitude helpers
// ------------------------------------------------------------------------------------
// R(q), column-major like the reference's vec_F(R): R[3c + r].
template <typename T>
__device__ __forceinline__ void quat_to_R(const T* q, T (&R)[9]) {
  const T w = q[0], x = q[1], y = q[2], z = q[3];
  const T xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
  R[0] = T(1) - T(2) * (yy + zz); R[1] = T(2) * (xy + wz);        R[2] = T(2) * (xz - wy);
  R[3] = T(2) * (xy - wz);        R[4] = T(1) - T(2) * (xx + zz); R[5] = T(2) * (yz + wx);
  R[6] = T(2) * (xz + wy);        R[7] = T(2) * (yz - wx);        R[8] = T(1) - T(2) * (xx + yy);
}

// ensure_SO3 (quad_utils.py:123-142) + attitude import.  The reference replaces R by the
// nearest rotation U V^T (SVD) when R^T R or det R is off by more than 1e-5; since the
// internal attitude is a unit quaternion, the nearest rotation is taken always (for an R
// that is orthonormal to round-off this changes nothing).  The polar factor is computed by
// the Newton iteration X <- (X + X^-T)/2, which converges quadratically to U V^T (det R > 0).
__device__ void R_to_quat(const double* Rin, double (&q)[4]) {
  double X[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) X[i] = Rin[i];
  for (int it = 0; it < 40; ++it) {
    double C[9];  // C = cof(X), column-major like X; X^-T = C / det X
    C[0] = X[4] * X[8] - X[5] * X[7]; C[1] = X[5] * X[6] - X[3] * X[8]; C[2] = X[3] * X[7] - X[4] * X[6];
    C[3] = X[2] * X[7] - X[1] * X[8]; C[4] = X[0] * X[8] - X[2] * X[6]; C[5] = X[1] * X[6] - X[0] * X[7];
    C[6] = X[1] * X[5] - X[2] * X[4]; C[7] = X[2] * X[3] - X[0] * X[5]; C[8] = X[0] * X[4] - X[1] * X[3];
    const double dd = X[0] * C[0] + X[1] * C[1] + X[2] * C[2];
    if (!(fabs(dd) > 1e-300)) break;
    const double inv = 1.0 / dd;
    double delta = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const double xn = 0.5 * (X[i] + C[i] * inv);
      delta = fmax(delta, fabs(xn - X[i]));
      X[i] = xn;
    }
    if (delta < 4e-16) break;
  }
  // Shepperd's method on the (now orthonormal) X; X[3c + r] = R(r, c)
  const double r00 = X[0], r11 = X[4], r22 = X[8];
  const double tr = r00 + r11 + r22;
  double w, x, y, z;
  if (tr >= r00 && tr >= r11 && tr >= r22) {
    w = 1.0 + tr; x = X[5] - X[7]; y = X[6] - X[2]; z = X[1] - X[3];
  } else if (r00 >= r11 && r00 >= r22) {
    w = X[5] - X[7]; x = 1.0 + r00 - r11 - r22; y = X[3] + X[1]; z = X[6] + X[2];
  } else if (r11 >= r22) {
    w = X[6] - X[2]; x = X[3] + X[1]; y = 1.0 - r00 + r11 - r22; z = X[7] + X[5];
  } else {
    w = X[1] - X[3]; x = X[6] + X[2]; y = X[7] + X[5]; z = 1.0 - r00 - r11 + r22;
  }
  const double inv = 1.0 / sqrt(w * w + x * x + y * y + z * z);
  q[0] = w * inv; q[1] = x * inv; q[2] = y * inv; q[3] = z * inv;
}

// ------------------------------------------------------------------------------------
// Dynamics (quad.py:321-335) in quaternion form.
// ------------------------------------------------------------------------------------
template <typename T>
struct Dyn {
  T c;           // f/m
  T A1;          // (J2-J3)/J1 with J2 = J1; the W2' coefficient (J3-J1)/J2 is -A1
  T U1, U2, U3;  // M_i / J_i
};

template <typename T>
__device__ __forceinline__ void rhs(const T* __restrict__ y, T* __restrict__ k, const Dyn<T>& p) {
  const T qw = y[3], qx = y[4], qy = y[5], qz = y[6];
  const T W1 = y[7], W2 = y[8], W3 = y[9];
  // v' = g e3 - (f/m) R e3,  R e3 = (2(xz + wy), 2(yz - wx), 1 - 2(xx + yy))
  const T c2 = T(2) * p.c;
  k[0] = -c2 * (qx * qz + qw * qy);
  k[1] = -c2 * (qy * qz - qw * qx);
  k[2] = (T(kG) - p.c) + c2 * (qx * qx + qy * qy);
  // q' = q (0, W) / 2   (<=> R' = R hat(W))
  const T h = T(0.5);
  k[3] = -h * (qx * W1 + qy * W2 + qz * W3);
  k[4] = h * (qw * W1 + qy * W3 - qz * W2);
  k[5] = h * (qw * W2 + qz * W1 - qx * W3);
  k[6] = h * (qw * W3 + qx * W2 - qy * W1);
  // W' = J^-1 (-W x JW + M), J = diag(J1, J1, J3): the (J1 - J2) W1 W2 term of W3' vanishes
  k[7] = p.A1 * W2 * W3 + p.U1;
  k[8] = p.U2 - p.A1 * W3 * W1;
  k[9] = p.U3;
}

template <typename T>
__device__ __forceinline__ void rk4_step(T (&x)[3], T (&y)[10], T h, const Dyn<T>& p) {
  T k[10], acc[10], yt[10], xs[3];
  const T h2 = T(0.5) * h, h6 = h * T(1.0 / 6.0);
  rhs(y, k, p);
#pragma unroll
  for (int i = 0; i < 10; ++i) { acc[i] = k[i]; yt[i] = y[i] + h2 * k[i]; }
#pragma unroll
  for (int i = 0; i < 3; ++i) xs[i] = y[i];
  rhs(yt, k, p);
#pragma unroll
  for (int i = 0; i < 3; ++i) xs[i] += T(2) * yt[i];
#pragma unroll
  for (int i = 0; i < 10; ++i) { acc[i] += T(2) * k[i]; yt[i] = y[i] + h2 * k[i]; }
  rhs(yt, k, p);
#pragma unroll
  for (int i = 0; i < 3; ++i) xs[i] += T(2) * yt[i];
#pragma unroll
  for (int i = 0; i < 10; ++i) { acc[i] += T(2) * k[i]; yt[i] = y[i] + h * k[i]; }
  rhs(yt, k, p);
#pragma unroll
  for (int i = 0; i < 3; ++i) x[i] += h6 * (xs[i] + yt[i]);
#pragma unroll
  for (int i = 0; i < 10; ++i) y[i] += h6 * (acc[i] + k[i]);
}

// The flow keeps |q| = 1; RK4 only to truncation order.  Restore it to first order.
template <typename T>
__device__ __forceinline__ void renorm_quat(T* q) {
  const T r = T(1.5) - T(0.5) * (q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] *= r;
}

// ------------------------------------------------------------------------------------
// LDS transposes between lane-per-env registers and AoS rows in global memory.
// The workgroup's rows [first, first+rows) x D floats are contiguous in global memory.
// ------------------------------------------------------------------------------------
template <int B, int D>
__device__ __forceinline__ void store_rows(float* __restrict__ gbase, const float (&vals)[D], float* smem, int tid, int rows) {
#pragma unroll
  for (int j = 0; j < D; ++j) smem[tid * D + j] = vals[j];
  __syncthreads();
  if (rows == B && (reinterpret_cast<uintptr_t>(gbase) & 15u) == 0) {
    constexpr int nvec = B * D / 4;  // B is a multiple of 4
    const float4* s4 = reinterpret_cast<const float4*>(smem);
    float4* g4 = reinterpret_cast<float4*>(gbase);
#pragma unroll
    for (int idx = tid; idx < nvec; idx += B) g4[idx] = s4[idx];
  } else {
    const int total = rows * D;
    for (int idx = tid; idx < total; idx += B) gbase[idx] = smem[idx];
  }
  __syncthreads();
}

template <int B, int D>
__device__ __forceinline__ void load_rows(const float* __restrict__ gbase, float (&vals)[D], float* smem, int tid, int rows) {
  if (rows == B && (reinterpret_cast<uintptr_t>(gbase) & 15u) == 0) {
    constexpr int nvec = B * D / 4;
    float4* s4 = reinterpret_cast<float4*>(smem);
    const float4* g4 = reinterpret_cast<const float4*>(gbase);
#pragma unroll
    for (int idx = tid; idx < nvec; idx += B) s4[idx] = g4[idx];
  } else {
    const int total = rows * D;
    for (int idx = tid; idx < total; idx += B) smem[idx] = gbase[idx];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < D; ++j) vals[j] = tid < rows ? smem[tid * D + j] : 0.f;
  __syncthreads();
}

template <int KIND> struct KindTraits;
template <> struct KindTraits<QR_KIND_QUAD>      { static constexpr int A = 4, D0 = 18, D1 = 0, NAG = 1; };
template <> struct KindTraits<QR_KIND_COUPLED>   { static constexpr int A = 4, D0 = 23, D1 = 0, NAG = 1; };
template <> struct KindTraits<QR_KIND_DECOUPLED> { static constexpr int A = 5, D0 = 15, D1 = 3, NAG = 2; };

// action_wrapper of the three kinds (quad.py:225-242, coupled:44-53, decoupled:49-59 + 68-73)
template <int KIND, typename T>
__device__ __forceinline__ void action_map(const float* a, const Work<T>& w, Dyn<T>& p) {
  const Phys<T> ph(w);
  T f, M1, M2, M3;
  if constexpr (KIND == QR_KIND_QUAD) {
    T t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = clampT(ph.scale_act * T(a[j]) + ph.avrg_act, T(kMinForce), ph.max_force);
    f = ((t[0] + t[1]) + t[2]) + t[3];
    M1 = ph.d * (t[3] - t[1]);
    M2 = ph.d * (t[0] - t[2]);
    M3 = ph.ctf * ((t[1] - t[0]) + (t[3] - t[2]));
  } else {
    f = clampT(T(4) * (ph.scale_act * T(a[0]) + ph.avrg_act), T(4) * T(kMinForce), T(4) * ph.max_force);
    if constexpr (KIND == QR_KIND_COUPLED) {
      M1 = T(a[1]); M2 = T(a[2]); M3 = T(a[3]);
    } else {  // M1 = b1.tau + J3 W3 W2, M2 = b2.tau - J3 W3 W1 from (R, W) at step start
      const T t1 = T(a[1]), t2 = T(a[2]), t3 = T(a[3]);
      const T qw = w.y[3], qx = w.y[4], qy = w.y[5], qz = w.y[6];  // b1, b2 = first two columns of R(q)
      const T b1t = (T(1) - T(2) * (qy * qy + qz * qz)) * t1 + T(2) * (qx * qy + qw * qz) * t2 + T(2) * (qx * qz - qw * qy) * t3;
      const T b2t = T(2) * (qx * qy - qw * qz) * t1 + (T(1) - T(2) * (qx * qx + qz * qz)) * t2 + T(2) * (qy * qz + qw * qx) * t3;
      M1 = b1t + ph.J3 * w.y[9] * w.y[8];
      M2 = b2t - ph.J3 * w.y[9] * w.y[7];
      M3 = T(a[4]);
    }
  }
  const T iJ1 = recip(ph.J1), iJ3 = recip(ph.J3);
  p.c = f * recip(ph.m);
  p.A1 = (ph.J1 - ph.J3) * iJ1;
  p.U1 = M1 * iJ1; p.U2 = M2 * iJ1; p.U3 = M3 * iJ3;
}

// get_norm_error_state (quad.py:421-466): fills the float32 observation rows and advances
// the trapezoid integrators (quad_utils.py:38-63).
template <int KIND, typename T>
__device__ __forceinline__ void error_obs(Work<T>& w, const T (&R)[9], const Coeffs& c, float (&o0)[KindTraits<KIND>::D0],
                                          float (&o1)[KindTraits<KIND>::D1 ? KindTraits<KIND>::D1 : 1]) {
  const T xl = T(c.x_lim), ixl = T(c.inv_x_lim), ivl = T(c.inv_v_lim), iWl = T(c.inv_W_lim);
  T ex[3], ev[3], eW[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {  // x/x_lim - xd/x_lim etc. (quad.py:423-434)
    ex[j] = w.x[j] * ixl - T(w.goal[j]) * ixl;
    ev[j] = w.y[j] * ivl - T(w.goal[3 + j]) * ivl;
    eW[j] = w.y[7 + j] * iWl - T(w.goal[9 + j]) * iWl;
  }
  const T* b1 = &R[0]; const T* b2 = &R[3]; const T* b3 = &R[6];
  const T b1d[3] = {T(w.goal[6]), T(w.goal[7]), T(w.goal[8])};
  const T db3 = b1d[0] * b3[0] + b1d[1] * b3[1] + b1d[2] * b3[2];
  T b1c[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) b1c[j] = b1d[j] - db3 * b3[j];
  const T sn = -(b1c[0] * b2[0] + b1c[1] * b2[1] + b1c[2] * b2[2]);
  const T cs = b1c[0] * b1[0] + b1c[1] * b1[1] + b1c[2] * b1[2];
  const float eb1 = atan2f((float)sn, (float)cs);  // [rad]
  const float eb1n = eb1 * (float)(1.0 / kPi);
  // integrators: I += (g_prev + g) dt/2 ; g uses I before the update.  They are float32 words
  // (stored and held), advanced in float32.
  const float hdt = (float)(c.dt * 0.5);
  float eIxn[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float g = fmaf(-(float)c.alpha, w.integ[j], (float)(ex[j] * xl));
    w.integ[j] = fmaf(w.integ[3 + j] + g, hdt, w.integ[j]);
    w.integ[3 + j] = g;
    eIxn[j] = clampT(w.integ[j] * (float)c.inv_eIx_lim, -1.0f, 1.0f);
  }
  const float gb = fmaf(-(float)c.beta, w.integ[6], eb1);
  w.integ[6] = fmaf(w.integ[7] + gb, hdt, w.integ[6]);
  w.integ[7] = gb;
  const float eIb1n = clampT(w.integ[6] * (float)c.inv_eIb1_lim, -1.0f, 1.0f);
  if constexpr (KIND == QR_KIND_COUPLED) {
#pragma unroll
    for (int j = 0; j < 3; ++j) { o0[j] = (float)ex[j]; o0[3 + j] = eIxn[j]; o0[6 + j] = (float)ev[j]; o0[20 + j] = (float)eW[j]; }
#pragma unroll
    for (int j = 0; j < 9; ++j) o0[9 + j] = (float)R[j];
    o0[18] = eb1n; o0[19] = eIb1n;
  } else {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      o0[j] = (float)ex[j]; o0[3 + j] = eIxn[j]; o0[6 + j] = (float)ev[j]; o0[9 + j] = (float)b3[j];
      o0[12 + j] = (float)(eW[0] * b1[j] + eW[1] * b2[j]);
    }
    o1[0] = eb1n; o1[1] = eIb1n; o1[2] = (float)eW[2];
  }
}

// ------------------------------------------------------------------------------------
// Goal generation: utils/trajectory_generator.py modes 0 and 1, per env.
// tr[8] = {calls, theta_init, b1d_x | w_b1d, b1d_y | smooth_term, x_init[3], -}
// ------------------------------------------------------------------------------------
// mark_traj_start(state) (:176-204) + the episode-start branch of calculate_desired:
//   mode 0 (:141-148): b1d = Rz(theta) b1_proj, theta ~ U(+-25 deg)
//   mode 1 (:253-266): x_init = x, t_traj ~ U(2,5), smooth = -ln(0.001)/t_traj, w_b1d ~ U(+-0.15 pi)
template <typename T>
__device__ __forceinline__ void traj_start(const Work<T>& w, float (&tr)[8], int goal_mode, float theta_b1d, float t_traj, float w_b1d) {
  const T qw = w.y[3], qx = w.y[4], qy = w.y[5], qz = w.y[6];
  const float b1x = (float)(T(1) - T(2) * (qy * qy + qz * qz)), b1y = (float)(T(2) * (qx * qy + qw * qz));
  const float theta_init = atan2f(b1y, b1x);  // update_initial_state (:199-204)
  tr[0] = 0.0f;
  tr[1] = theta_init;
  if (goal_mode == QR_GOAL_MODE0) {
    float sn, cs;
    sincos_small(theta_init + theta_b1d, sn, cs);  // Rz(theta) (cos th_i, sin th_i, 0)
    tr[2] = cs; tr[3] = sn;
    tr[4] = tr[5] = tr[6] = 0.0f;
  } else {  // mode 1: x_init + draws; mode 6: eight_shaped_center = x (:430), no draws
    tr[2] = w_b1d;
    tr[3] = 6.907755278982137f / t_traj;  // -ln(0.001) / t_traj
#pragma unroll
    for (int j = 0; j < 3; ++j) tr[4 + j] = (float)w.x[j];
  }
  tr[7] = 0.0f;
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
template <typename T>
__device__ __forceinline__ void traj_goal(Work<T>& w, float (&tr)[8], int goal_mode, const Coeffs& c, float (&b1d_dot)[3]) {
  tr[0] += 1.0f;  // update_current_time (:224-229): t = t + dt on every call
  float b1d[3];
  if (goal_mode == QR_GOAL_MODE0) {  // set_desired_states_to_zero + the b1d drawn at episode start
#pragma unroll
    for (int j = 0; j < 6; ++j) w.goal[j] = 0.0f;
    b1d[0] = tr[2]; b1d[1] = tr[3]; b1d[2] = 0.0f;
    b1d_dot[0] = b1d_dot[1] = b1d_dot[2] = 0.0f;
  } else if (goal_mode == QR_GOAL_MODE6) {  // eight_shaped_curve (:418-505)
    const float t = fminf(tr[0] * (float)c.dt, c.e8_tmax);
    const float ek = expf(-c.e8_k * t);
    const float e = 1.0f - ek, de = c.e8_k * ek;  // exp_term, d/dt exp_term
    float s1, c1, s2, c2;
    sincos_small(c.e8_w1 * t, s1, c1);
    sincos_small(c.e8_w2 * t, s2, c2);
    const float za = 0.5f * (tr[6] - c.e8_alt);  // synchronised altitude command (:487-492)
    w.goal[0] = fmaf(c.e8_A2 * s2, e, tr[4]);
    w.goal[1] = fmaf(c.e8_A1 * (c1 - 1.0f), e, tr[5]);
    w.goal[2] = fmaf(za, 1.0f - c1, tr[6]);
    w.goal[3] = c.e8_A2 * (c.e8_w2 * c2 * e + s2 * de);
    w.goal[4] = c.e8_A1 * (-c.e8_w1 * s1 * e + (c1 - 1.0f) * de);
    w.goal[5] = za * c.e8_w1 * s1;
    const float term = fmaf(c.e8_wb * t, e, tr[1]), dterm = c.e8_wb * (e + t * de);  // yaw (:494-498)
    float sn, cs;
    sincos_small(term, sn, cs);
    b1d[0] = cs; b1d[1] = sn; b1d[2] = 0.0f;
    b1d_dot[0] = -sn * dterm; b1d_dot[1] = cs * dterm; b1d_dot[2] = 0.0f;
  } else {  // hovering (:268-277), x_goal = 0
    const float t = tr[0] * (float)c.dt;
    const float wb = tr[2], sm = tr[3];
    const float e = expf(-sm * t);
#pragma unroll
    for (int j = 0; j < 3; ++j) { w.goal[j] = tr[4 + j] * e; w.goal[3 + j] = -tr[4 + j] * sm * e; }
    float sn, cs;
    sincos_small(fmaf(wb, t, tr[1]), sn, cs);
    b1d[0] = cs; b1d[1] = sn; b1d[2] = 0.0f;
    b1d_dot[0] = -wb * sn; b1d_dot[1] = wb * cs; b1d_dot[2] = 0.0f;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) w.goal[6 + j] = b1d[j];
  // Wd = (0, 0, b3 . (b1c x b1c_dot)) with b3' = R hat(W) e3 = W2 b1 - W1 b2 (:165-172)
  T R[9];
  quat_to_R(&w.y[3], R);
  const T W1 = w.y[7], W2 = w.y[8];
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
  w.goal[9] = 0.0f; w.goal[10] = 0.0f;
  w.goal[11] = (float)(R[6] * oc0 + R[7] * oc1 + R[8] * oc2);
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
typedef int v2i_t __attribute__((ext_vector_type(2)));

template <typename E>
struct SoA {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned L;  // elements between fields
  __device__ __forceinline__ SoA(const void* base, int fields, int64_t ld)
      : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)((int64_t)fields * ld * (int64_t)sizeof(E) > 0x7fffffffLL ? 0x7fffffffLL : (int64_t)fields * ld * (int64_t)sizeof(E)), 0x00020000)),
        L((unsigned)ld) {}
  __device__ __forceinline__ unsigned soff(int f, unsigned first) const { return ((unsigned)f * L + first) * (unsigned)sizeof(E); }
  __device__ __forceinline__ E load(int f, unsigned first, unsigned lane) const {
    if constexpr (sizeof(E) == 4) {
      return __builtin_bit_cast(E, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4u, soff(f, first), 0));
    } else {
      return __builtin_bit_cast(E, __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane * 8u, soff(f, first), 0));
    }
  }
  __device__ __forceinline__ void store(int f, unsigned first, unsigned lane, E v) const {
    if constexpr (sizeof(E) == 4) {
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rsrc, lane * 4u, soff(f, first), 0);
    } else {
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i_t, v), rsrc, lane * 8u, soff(f, first), 0);
    }
  }
};

template <typename XV, typename QW, typename T>
__device__ __forceinline__ void load_state(const Args& a, int64_t first64, unsigned lane, Work<T>& w) {
  const SoA<XV> pv(a.pos_vel, 6, a.ld);
  const SoA<QW> ar(a.att_rate, 7, a.ld);
  const unsigned first = (unsigned)first64;
#pragma unroll
  for (int f = 0; f < 3; ++f) { w.x[f] = T(pv.load(f, first, lane)); w.y[f] = T(pv.load(3 + f, first, lane)); }
#pragma unroll
  for (int f = 0; f < 7; ++f) w.y[3 + f] = T(ar.load(f, first, lane));
}

template <typename XV, typename QW, typename T>
__device__ __forceinline__ void store_state(const Args& a, int64_t first64, unsigned lane, const Work<T>& w) {
  const SoA<XV> pv(a.pos_vel, 6, a.ld);
  const SoA<QW> ar(a.att_rate, 7, a.ld);
  const unsigned first = (unsigned)first64;
#pragma unroll
  for (int f = 0; f < 3; ++f) { pv.store(f, first, lane, (XV)w.x[f]); pv.store(3 + f, first, lane, (XV)w.y[f]); }
#pragma unroll
  for (int f = 0; f < 7; ++f) ar.store(f, first, lane, (QW)w.y[3 + f]);
}

template <typename T>
__device__ __forceinline__ void idle_work(Work<T>& w) {  // lanes past the ragged tail
#pragma unroll
  for (int f = 0; f < 3; ++f) w.x[f] = T(0);
#pragma unroll
  for (int f = 0; f < 10; ++f) w.y[f] = T(f == 3 ? 1 : 0);
#pragma unroll
  for (int f = 0; f < 12; ++f) w.goal[f] = f == 6 ? 1.0f : 0.0f;
#pragma unroll
  for (int f = 0; f < 8; ++f) w.integ[f] = 0.0f;
#pragma unroll
  for (int f = 0; f < 6; ++f) w.prm[f] = 0.0f;
  w.nominal = true;
}

// ------------------------------------------------------------------------------------
// PPO actor in the loop (qr_rollout_actor)
// ------------------------------------------------------------------------------------
// tanh for the action mean: (1 - e) / (1 + e), e = exp(-2|x|), sign restored.  Absolute error
// <= 2e-7 (v_exp_f32 + v_rcp_f32); branch-free, unlike the OCML tanhf (three regimes).
__device__ __forceinline__ float tanh_fast(float x) {
  const float e = __expf(-2.0f * fabsf(x));
  const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
  return copysignf(t, x);
}

// MLP_Actor_PPO.forward (ppo_mlp.py:30-43): tanh(mean_linear(relu(fc2(relu(fc1(x)))))).  One lane
// = one env = one row of the batch.  The weights are wave-uniform: they are copied once per launch
// into LDS, TRANSPOSED to [in][out] (out padded to a multiple of 4), and every lane reads the
// same address — broadcast ds_read_b128, no bank conflicts, 4 weights per LDS instruction.
// A layer is evaluated input-major: for each input k, all `out` accumulators take one FMA, so
// consecutive FMAs are independent (16 chains in flight).  Output-major — each neuron's 23-term
// dot product as one dependent FMA chain — measures 3x slower here: with one wave per SIMD
// nothing hides the ~10-cycle dependent-FMA latency.
template <int D, int H, int A>
struct ActorLds {
  static constexpr int HP = (H + 3) & ~3, AP = (A + 3) & ~3;
  static constexpr int O_FC1W = 0, O_FC1B = O_FC1W + D * HP, O_FC2W = O_FC1B + HP, O_FC2B = O_FC2W + H * HP,
                       O_MW = O_FC2B + HP, O_MB = O_MW + H * AP, O_LS = O_MB + AP, SIZE = O_LS + AP;

  __device__ static void fill(float* sm, const ActorW& p, int tid) {  // sm[k][j] = W[j][k]
    for (int i = tid; i < D * HP; i += 64) { const int k = i / HP, j = i - k * HP; sm[O_FC1W + i] = j < H ? p.fc1_w[j * D + k] : 0.0f; }
    for (int i = tid; i < H * HP; i += 64) { const int k = i / HP, j = i - k * HP; sm[O_FC2W + i] = j < H ? p.fc2_w[j * H + k] : 0.0f; }
    for (int i = tid; i < H * AP; i += 64) { const int k = i / AP, j = i - k * AP; sm[O_MW + i] = j < A ? p.mean_w[j * H + k] : 0.0f; }
    if (tid < HP) { sm[O_FC1B + tid] = tid < H ? p.fc1_b[tid] : 0.0f; sm[O_FC2B + tid] = tid < H ? p.fc2_b[tid] : 0.0f; }
    if (tid < AP) { sm[O_MB + tid] = tid < A ? p.mean_b[tid] : 0.0f; sm[O_LS + tid] = tid < A ? p.log_std[tid] : 0.0f; }
  }

  template <int NI, int NO, int NOP>
  __device__ __forceinline__ static void layer(const float* w, const float* bias, const float (&x)[NI], float (&y)[NO]) {
#pragma unroll
    for (int j = 0; j < NO; ++j) y[j] = bias[j];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
#pragma unroll
      for (int j = 0; j < NO; ++j) y[j] = fmaf(w[k * NOP + j], x[k], y[j]);
    }
  }

  __device__ __forceinline__ static void mean(const float* sm, const float (&x)[D], float (&out)[A]) {
    float h1[H], h2[H];
    layer<D, H, HP>(sm + O_FC1W, sm + O_FC1B, x, h1);
#pragma unroll
    for (int j = 0; j < H; ++j) h1[j] = fmaxf(h1[j], 0.0f);
    layer<H, H, HP>(sm + O_FC2W, sm + O_FC2B, h1, h2);
#pragma unroll
    for (int j = 0; j < H; ++j) h2[j] = fmaxf(h2[j], 0.0f);
    layer<H, A, AP>(sm + O_MW, sm + O_MB, h2, out);
#pragma unroll
    for (int j = 0; j < A; ++j) out[j] = tanh_fast(out[j]);
  }
};

// The 16-wide actor on the matrix cores.  Evaluated per-lane on the VALU the three layers are
// 688 FMAs per env-step whose 744 wave-uniform weights have to be re-delivered every step
// (LDS broadcast reads or scalar loads): with the step kernel at its VGPR limit only two
// ds_read_b128 fit in flight and the evaluation measures 3.8 us, LDS-latency-bound.  As a
// transposed GEMM  H^T[16 x 64 envs] = W[16 x K] . X^T[K x 64]  on v_mfma_f32_16x16x4_f32 (exact f32)
// the weights are the A operand and stay RESIDENT in 14 registers per lane for the whole
// rollout; only the observations move (one LDS read per MFMA for the first layer).
//   lane l: c = l & 15, g = l >> 4.   A: lane supplies A[c][k = g].  B: B[k = g][c].
//   D: lane holds D[4 g + r][c], r = 0..3.  The 64 envs are 4 column blocks b of 16.
//   layer 1: A = W1[c][4 s + g] (k-step s), B = X[env 16 b + c][4 s + g] from the LDS obs tile,
//            D_b = h1[b][r] = H1[4 g + r][env 16 b + c].
//   layer 2: the lane's h1[b][s] IS a B operand if k-step s is given the hidden units
//            k(s, g) = 4 g + s, so A = W2[c][4 g + s]: no data movement between layers.
//   layer 3: block b uses A_b = W3 placed in rows 4 b .. 4 b + 3 (zero elsewhere) and all blocks
//            accumulate into ONE D: lane (g, c) then holds mean[r] of env 16 g + c — its own env.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int D>  // obs_dim 23 (COUPLED) or 15 (DECOUPLED agent 1); hidden 16, 4 actions
struct ActorMfma {
  static constexpr int KS = (D + 3) / 4;
  float a1[KS], a2[4], w3[4], bias1[4], bias2[4], bias3[4], log_std[4];

  __device__ __forceinline__ void load(const ActorW& p, int lane) {
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int s = 0; s < KS; ++s) a1[s] = (4 * s + g < D) ? p.fc1_w[c * D + 4 * s + g] : 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) { a2[s] = p.fc2_w[c * 16 + 4 * g + s]; w3[s] = p.mean_w[(c & 3) * 16 + 4 * g + s]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bias1[r] = p.fc1_b[4 * g + r]; bias2[r] = p.fc2_b[4 * g + r]; bias3[r] = p.mean_b[r]; log_std[r] = p.log_std[r];
    }
  }

  // xs: LDS tile [64 envs][D] of the wave's observations (row = lane)
  __device__ __forceinline__ void mean(const float* xs, int lane, float (&out)[4]) const {
    const int c = lane & 15, g = lane >> 4;
    f32x4 h1[4], h2[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      h1[b] = f32x4{bias1[0], bias1[1], bias1[2], bias1[3]};
      h2[b] = f32x4{bias2[0], bias2[1], bias2[2], bias2[3]};
    }
    // all B operands of the first layer are requested before the first MFMA (the reads are
    // unconditional: past the last feature the address is clamped and the weight a1 is 0)
    float x[KS][4];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k = (4 * s + g < D) ? 4 * s + g : D - 1;
#pragma unroll
      for (int b = 0; b < 4; ++b) x[s][b] = xs[(16 * b + c) * D + k];
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int b = 0; b < 4; ++b) h1[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], x[s][b], h1[b], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int b = 0; b < 4; ++b) h2[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[s], fmaxf(h1[b][s], 0.0f), h2[b], 0, 0, 0);
    }
    f32x4 m0 = f32x4{bias3[0], bias3[1], bias3[2], bias3[3]}, m1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};  // two chains
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float w = ((c >> 2) == b) ? w3[s] : 0.0f;
        f32x4& m = (b & 1) ? m1 : m0;
        m = __builtin_amdgcn_mfma_f32_16x16x4f32(w, fmaxf(h2[b][s], 0.0f), m, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = tanh_fast(m0[r] + m1[r]);
  }
};

// PPO.choose_action (ppo.py:93-101): a = clamp(mean + exp(log_std) eps, +-max_action) and the
// per-component Normal(mean, std).log_prob of the clamped action (ppo.py:97-98).
template <int A>
__device__ __forceinline__ void actor_sample(const float* log_std, const float (&mean)[A], const float* eps, bool deterministic,
                                             float max_action, float* act, float* logp) {
#pragma unroll
  for (int j = 0; j < A; ++j) {
    const float ls = log_std[j];
    const float sd = __expf(ls);
    const float raw = deterministic ? mean[j] : fmaf(sd, eps[j], mean[j]);
    const float aj = fminf(fmaxf(raw, -max_action), max_action);
    const float z = (aj - mean[j]) * __expf(-ls);
    act[j] = aj;
    logp[j] = fmaf(-0.5f * z, z, -ls - 0.91893853320467274f);
  }
}

// 4 standard normals per Philox block (Box-Muller).  Stream: (noise_seed, global env id, global
// step, 0x80000000 | block) — the top bit keeps it apart from the reset stream (seed, id, episode, b).
__device__ __forceinline__ void normal4(float (&z)[4], uint64_t seed, uint64_t gid, uint64_t step, uint32_t block) {
  uint32_t ctr[4] = {(uint32_t)gid, (uint32_t)(gid >> 32) ^ (uint32_t)(step >> 32), (uint32_t)step, 0x80000000u | block};
  philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u1 = fmaf((float)(ctr[2 * h] >> 8), 0x1p-24f, 0x1p-25f);   // (0, 1)
    const float u2 = fmaf((float)(ctr[2 * h + 1] >> 8), 0x1p-24f, 0x1p-25f);
    const float r = sqrtf(-2.0f * __logf(u1));
    float sn, cs;
    sincos_small(6.283185307179586f * u2, sn, cs);
    z[2 * h] = r * cs; z[2 * h + 1] = r * sn;
  }
}

// ------------------------------------------------------------------------------------
// The fused step / rollout kernel
// ------------------------------------------------------------------------------------
// TRAJ = the goal generator (trajectory_generator.py modes 0/1) is fused into the step; a
// separate instantiation so that the default path carries none of its registers.  ADAPT = the
// rate-adaptive substep count (QrCoeffs::w_adapt); launch_kind() picks the plain instantiation
// whenever adaptivity provably cannot trigger.
// POLICY = qr_rollout_actor: the action of every step comes from the PPO actor(s) evaluated on the
// env's current observation, which stays in registers from one step to the next.
template <int KIND, typename XV, typename QW, int B, bool TRAJ, bool ADAPT, bool POLICY = false>
__global__ __launch_bounds__(B, ((TRAJ || POLICY) ? 1 : QR_WAVES_PER_SIMD))
void step_kernel(void* pos_vel, void* att_rate, const float* action, float* params, float* integ, int64_t n_envs,
                 int64_t ld_envs, const Args a_in) {
  // The leading scalar arguments duplicate the fields of Args the first loads depend on: as
  // plain kernel arguments they are preloaded into SGPRs by the dispatcher (gfx950 kernarg
  // preload, -mllvm -amdgpu-kernarg-preload-count), so the state loads are issued without first
  // waiting for a scalar-load round trip to the kernarg segment.
  Args a = a_in;
  a.pos_vel = pos_vel; a.att_rate = att_rate; a.action = action; a.params = params; a.integ = integ;
  a.n = n_envs; a.ld = ld_envs;
  using T = QW;  // arithmetic type
  using KT = KindTraits<KIND>;
  constexpr int A = KT::A, D0 = KT::D0, D1 = KT::D1 ? KT::D1 : 1, NAG = KT::NAG;
  __shared__ __attribute__((aligned(16))) float smem[B * (D0 > A ? D0 : A)];
  const int tid = threadIdx.x;
  const unsigned lane = threadIdx.x;
  const unsigned ufirst = blockIdx.x * (unsigned)B;
  const int64_t first = (int64_t)blockIdx.x * B;
  const int64_t i = first + tid;
  const int64_t N = a.n, L = a.ld;
  const int rows = (int)((N - first) < B ? (N - first) : B);
  const bool active = tid < rows;
  const Coeffs& c = a.c;
#if QR_ABLATE == 1  // measurement build: launch floor only
  return;
#endif

  Work<T> w;
  // ---- load the env's working set (SoA, lane-contiguous) ----
  idle_work(w);
  if (active) {
    load_state<XV, QW, T>(a, first, lane, w);
    if (a.params) {
      const SoA<float> prm(a.params, 6, L);
#pragma unroll
      for (int f = 0; f < 6; ++f) w.prm[f] = prm.load(f, ufirst, lane);
      w.nominal = false;
    }
    if (a.goal) {
      const SoA<float> goal(a.goal, 12, L);
#pragma unroll
      for (int f = 0; f < 12; ++f) w.goal[f] = goal.load(f, ufirst, lane);
    }
    if (KIND != QR_KIND_QUAD) {
      const SoA<float> integ(a.integ, 8, L);
#pragma unroll
      for (int f = 0; f < 8; ++f) w.integ[f] = integ.load(f, ufirst, lane);
    }
  }
  float tr[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) tr[f] = 0.0f;
  const int goal_mode = TRAJ ? a.goal_mode : QR_GOAL_EXTERNAL;  // wave-uniform
  if (TRAJ && active) {
    const SoA<float> traj(a.traj, 8, L);
#pragma unroll
    for (int f = 0; f < 7; ++f) tr[f] = traj.load(f, ufirst, lane);
  }
  int32_t steps = (a.steps && active) ? (a.steps + first)[lane] : 0;
  // The episode counter (RNG stream id) is fetched with the rest of the working set: read
  // lazily inside the reset path it would put a full memory round-trip (~1.5 us) on the
  // critical path of every wave that has a resetting lane.
  int32_t episode = ((a.flags & QR_FLAG_AUTO_RESET) && active) ? (a.episode + first)[lane] : 0;
  bool params_dirty = false;

  // Action rows [N][A] -> lane registers.  A = 4: one 16-byte load per lane.  A = 5: five dword
  // loads per lane (a wave covers 1280 contiguous bytes; L1 merges the sectors).  In a rollout
  // the row of step t+1 is requested before the arithmetic of step t, so its latency is hidden.
  float act_next[A];
  auto load_action = [&](int t, float (&dst)[A]) {
    const float* abase = a.action + ((int64_t)t * N + first) * A;
    if constexpr (A == 4) {
      const float4 v = reinterpret_cast<const float4*>(abase)[lane];
      dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    } else {
#pragma unroll
      for (int j = 0; j < A; ++j) dst[j] = abase[lane * A + j];
    }
  };
#pragma unroll
  for (int j = 0; j < A; ++j) act_next[j] = 0.f;
  if constexpr (!POLICY) {
    if (active) load_action(0, act_next);
  }
  // POLICY: the observation the next action is computed from (rows -> lane registers once, then
  // carried from step to step)
  float po0[D0], po1[D1];
  // agent 0 (23 / 15 -> 16 -> 16 -> 4, args_parse.py:40, main.py:68-73) on the matrix cores, weights
  // resident in registers; agent 1 of DECOUPLED (3 -> 4 -> 4 -> 1: 32 FMAs) per lane from LDS
  ActorMfma<D0> actor0;
  using Actor1 = ActorLds<3, 4, 1>;
  __shared__ __attribute__((aligned(16))) float wsm[POLICY ? Actor1::SIZE : 4];
  if constexpr (POLICY) {
    load_rows<B, D0>(a.obs0_in + first * D0, po0, smem, tid, rows);
    if constexpr (KT::D1 > 0) load_rows<B, D1>(a.obs1_in + first * D1, po1, smem, tid, rows);
    actor0.load(a.actor[0], tid);
    if constexpr (KT::D1 > 0) Actor1::fill(wsm, a.actor[1], tid);
    __syncthreads();
  }

  for (int t = 0; t < a.n_steps; ++t) {
    float act[A];
    if constexpr (POLICY) {
      float mean[A], eps[A], logp[A];
      // the wave's observation rows -> LDS tile [lane][D0] (B operands of the first layer)
#pragma unroll
      for (int j = 0; j < D0; ++j) smem[tid * D0 + j] = po0[j];
      __syncthreads();
      if constexpr (KIND == QR_KIND_COUPLED) {
        actor0.mean(smem, tid, mean);
      } else {
        float m0[4], m1[1];
        actor0.mean(smem, tid, m0);
        Actor1::mean(wsm, po1, m1);
#pragma unroll
        for (int j = 0; j < 4; ++j) mean[j] = m0[j];
        mean[A - 1] = m1[0];
      }
#pragma unroll
      for (int j = 0; j < A; ++j) eps[j] = 0.0f;
      if (!a.deterministic) {
        if (a.noise != nullptr) {  // injected draws [T][N][A]
          if (active) {
            const float* nbase = a.noise + ((int64_t)t * N + first) * A;
#pragma unroll
            for (int j = 0; j < A; ++j) eps[j] = nbase[lane * A + j];
          }
        } else {
          float z[4];
          normal4(z, a.noise_seed, (uint64_t)(a.env_offset + i), a.step_base + (uint64_t)t, 0u);
#pragma unroll
          for (int j = 0; j < 4; ++j) eps[j] = z[j];
          if constexpr (A > 4) {
            normal4(z, a.noise_seed, (uint64_t)(a.env_offset + i), a.step_base + (uint64_t)t, 1u);
            eps[A - 1] = z[0];
          }
        }
      }
      __syncthreads();  // the tile is reused by the row stores below
      actor_sample<4>(actor0.log_std, *reinterpret_cast<const float(*)[4]>(&mean[0]), &eps[0], a.deterministic != 0, a.max_action, &act[0], &logp[0]);
      if constexpr (A > 4)
        actor_sample<1>(wsm + Actor1::O_LS, *reinterpret_cast<const float(*)[1]>(&mean[A - 1]), &eps[A - 1], a.deterministic != 0, a.max_action,
                        &act[A - 1], &logp[A - 1]);
      if (active) {
        const int64_t arow = ((int64_t)t * N + first) * A;
        if constexpr (A == 4) {
          reinterpret_cast<float4*>(a.act_out + arow)[lane] = make_float4(act[0], act[1], act[2], act[3]);
          if (a.logp_out) reinterpret_cast<float4*>(a.logp_out + arow)[lane] = make_float4(logp[0], logp[1], logp[2], logp[3]);
        } else {
#pragma unroll
          for (int j = 0; j < A; ++j) {
            (a.act_out + arow)[lane * A + j] = act[j];
            if (a.logp_out) (a.logp_out + arow)[lane * A + j] = logp[j];
          }
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < A; ++j) act[j] = act_next[j];
      if (active && t + 1 < a.n_steps) load_action(t + 1, act_next);
    }

#if QR_ABLATE == 2  // measurement build: memory traffic only (no integration)
    w.x[0] += T(act[0]);
#else
    // ---- goal for this step from the pre-step state (main.py:145-147) ----
    if constexpr (TRAJ) {
      float b1d_dot[3];
      traj_goal(w, tr, goal_mode, c, b1d_dot);
    }
    // ---- action_wrapper ----
    Dyn<T> dyn;
    action_map<KIND, T>(act, w, dyn);
    // ---- observation_wrapper: integrate over dt with zero-order-hold (f, M) ----
    // The reference's DOP853 is adaptive (6 % of its steps subdivide); the fixed-step stand-in
    // is made rate-adaptive: RK4's local error grows like (|W| h)^5, so a wave that contains an
    // env spinning faster than w_adapt takes ceil(max|W_i| / w_adapt) times the substeps.  The
    // multiplier is the wave's maximum (found with ballots, so the substep loop stays wave-uniform
    // and in regime — |W| < 2 pi < w_adapt — this costs one ballot): every lane takes at least
    // the count its own rate asks for.
    int nsub = a.substeps;
    if constexpr (ADAPT) {
      const T wmax = fmax(fmax(fabs(w.y[7]), fabs(w.y[8])), fabs(w.y[9]));
      const T need = wmax * T(c.inv_w_adapt);
      int mul = 1;
      while (mul < 16 && __ballot(need > T(mul))) ++mul;
      nsub *= mul;
    }
    const T h = T(c.dt) * recip(T(nsub));
    for (int s = 0; s < nsub; ++s) rk4_step(w.x, w.y, h, dyn);
    renorm_quat(&w.y[3]);
    // x, v take their storage precision at every env-step boundary, so that a K-step rollout
    // (state kept in registers) is bit-identical to K single-step launches
    if constexpr (!std::is_same<XV, T>::value) {
#pragma unroll
      for (int j = 0; j < 3; ++j) { w.x[j] = T((XV)w.x[j]); w.y[j] = T((XV)w.y[j]); }
    }
#endif

    // ---- obs / reward / done ----
    T R[9];
    quat_to_R(&w.y[3], R);
    float o0[D0];
    float o1[D1];
    float rraw[NAG], rwd[NAG];
    bool dn[NAG];
    if constexpr (KIND == QR_KIND_QUAD) {
      // reward_wrapper (quad.py:274-298)
      T eX2 = 0, eV2 = 0, W2 = 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const T dx = w.x[j] - T(w.goal[j]), dv = w.y[j] - T(w.goal[3 + j]);
        eX2 += dx * dx; eV2 += dv * dv; W2 += w.y[7 + j] * w.y[7 + j];
      }
      // eb1 = signed angle from b1d to b1_proj ~ (R00, R10, 0) (quad_utils.py:97-101,157-177).
      // acos(du.cu) with the sign of (du x cu)_z == atan2(|du x cu|, du.cu), which is invariant
      // to the lengths of both vectors, so neither is normalised.
      const T g6 = T(w.goal[6]), g7 = T(w.goal[7]), g8 = T(w.goal[8]);
      const T dot = g6 * R[0] + g7 * R[1];
      const T cz = g6 * R[1] - g7 * R[0];
      const T hy2 = R[0] * R[0] + R[1] * R[1];
      const float sabs = sqrtf((float)(g8 * g8 * hy2 + cz * cz));
      float ang = atan2f(sabs, (float)dot);
      if (cz < T(0)) ang = -ang;
      const T eb1 = T(ang) * T(1.0 / kPi);
      const T r = -T(c.Cx) * eX2 - T(c.Cb1) * fabs(eb1) - T(c.Cv) * eV2 - T(c.CW) * W2;
      rraw[0] = (float)r;
      rwd[0] = (float)clampT((r - T(c.rmin_mono)) * T(c.inv_nrmin_mono), T(0), T(1));
      // done_wrapper (quad.py:301-318): roll = atan2(R21,R22), pitch = -asin(R20)
      // (bitwise | on purpose: straight-line compares, no short-circuit branches)
      bool d = false;
#pragma unroll
      for (int j = 0; j < 3; ++j)
        d = d | !(fabs(w.x[j]) < T(c.x_lim)) | !(fabs(w.y[j]) < T(c.v_lim)) | !(fabs(w.y[7 + j]) < T(c.W_lim));
      d = d | !(fabs(R[2]) < T(c.sin_euler_lim));           // |pitch| >= lim
      d = d | !(fabs(R[5]) < T(c.tan_euler_lim) * R[8]);    // |atan2(R21,R22)| >= lim
      dn[0] = d;
    } else {
      error_obs<KIND, T>(w, R, c, o0, o1);
      if constexpr (KIND == QR_KIND_COUPLED) {  // coupled:78-110, float32 arithmetic on the float32 obs
        const float r = -(float)c.Cx * sq3(&o0[0]) + -(float)c.CIx * sq3(&o0[3]) + -(float)c.Cv * sq3(&o0[6]) +
                        -(float)c.Cb1 * fabsf(o0[18]) + -(float)c.CIb1 * (o0[19] * o0[19]) + -(float)c.CW * sq3(&o0[20]);
        rraw[0] = r;
        rwd[0] = interp01(r, (float)c.rmin_mono, (float)c.inv_nrmin_mono);
        dn[0] = out3(&o0[0]) | out3(&o0[6]) | out3(&o0[20]);
      } else {  // decoupled:92-140
        const float r1 = -(float)c.Cx * sq3(&o0[0]) + -(float)c.CIx * sq3(&o0[3]) + -(float)c.Cv * sq3(&o0[6]) +
                         -(float)c.Cw12 * sq3(&o0[12]);
        const float r2 = -(float)c.Cb1 * fabsf(o1[0]) + -(float)c.CIb1 * (o1[1] * o1[1]) + -(float)c.CW3 * (o1[2] * o1[2]);
        rraw[0] = r1; rraw[NAG - 1] = r2;
        rwd[0] = interp01(r1, (float)c.rmin_1, (float)c.inv_nrmin_1); rwd[NAG - 1] = interp01(r2, (float)c.rmin_2, (float)c.inv_nrmin_2);
        dn[0] = out3(&o0[0]) | out3(&o0[6]) | out3(&o0[12]);
        dn[NAG - 1] = !(fabsf(o1[2]) < 1.0f);
      }
    }
    // crash override (quad.py:162-166)
#pragma unroll
    for (int g = 0; g < NAG; ++g)
      if (dn[g]) rwd[g] = -1.0f;

    // ---- time limit + auto-reset ----
    steps += 1;
    const bool trunc = a.max_episode_steps > 0 && steps >= a.max_episode_steps;
    bool any_done = trunc;
#pragma unroll
    for (int g = 0; g < NAG; ++g) any_done = any_done | dn[g];
    const bool need_reset = (a.flags & QR_FLAG_AUTO_RESET) && any_done && active;
    if (__ballot(need_reset)) {  // wave-uniform: skip unless some lane of this wave resets
      if (need_reset) episode += 1;
      Draws d;
#if QR_ABLATE == 3  // measurement build: reset path without the RNG
#pragma unroll
      for (int j = 0; j < 20; ++j) d.r[j] = 0x9E3779B9u * (uint32_t)(j + 1) + (uint32_t)episode;
#else
      coop_draw20(d, need_reset, a.seed, (uint64_t)(a.env_offset + i), (uint32_t)episode);
#endif
#if QR_ABLATE == 4  // measurement build: RNG only, trivial consumption
      if (need_reset) {
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < 20; ++j) acc ^= d.r[j];
        w.x[0] = T((float)(acc & 0xFFFF) * 1e-5f);
        a.episode[i] = episode;
      }
#else
      if (need_reset) {
        const bool eval = (a.flags & QR_FLAG_EVAL_RESET) != 0;
        const bool randomise = !eval && !(a.flags & QR_FLAG_NO_UDM) && a.params != nullptr;
        sample_reset(w, d, randomise, eval, c);
        if (a.params != nullptr) params_dirty = true;
        (a.episode + first)[lane] = episode;
        steps = 0;
        if constexpr (TRAJ) {  // mark_traj_start + first get_desired of the episode (main.py:227-229)
          float th, tt, wb, b1d_dot[3];
          traj_draws(d.r[19], th, tt, wb);
          traj_start(w, tr, goal_mode, th, tt, wb);
          traj_goal(w, tr, goal_mode, c, b1d_dot);
        }
        quat_to_R(&w.y[3], R);
        if constexpr (KIND != QR_KIND_QUAD) {
#pragma unroll
          for (int f = 0; f < 8; ++f) w.integ[f] = 0.0f;
          error_obs<KIND, T>(w, R, c, o0, o1);  // first observation of the new episode (main.py:226-230)
        }
      }
#endif
    }

    // ---- outputs of step t ----
    const int64_t row0 = (int64_t)t * N + first;
    if constexpr (KIND == QR_KIND_QUAD) {
      if (a.obs0 != nullptr) {  // next state in the reference's order (x, v, vec_F(R), W)
#pragma unroll
        for (int j = 0; j < 3; ++j) { o0[j] = (float)w.x[j]; o0[3 + j] = (float)w.y[j]; o0[15 + j] = (float)w.y[7 + j]; }
#pragma unroll
        for (int j = 0; j < 9; ++j) o0[6 + j] = (float)R[j];
        store_rows<B, D0>(a.obs0 + row0 * D0, o0, smem, tid, rows);
      }
    } else {
      store_rows<B, D0>(a.obs0 + row0 * D0, o0, smem, tid, rows);
    }
    if constexpr (KT::D1 > 0) store_rows<B, D1>(a.obs1 + row0 * D1, o1, smem, tid, rows);
    if constexpr (POLICY) {
#pragma unroll
      for (int j = 0; j < D0; ++j) po0[j] = o0[j];
#pragma unroll
      for (int j = 0; j < D1; ++j) po1[j] = o1[j];
    }
    if (active) {
      if constexpr (NAG == 1) {
        (a.reward + row0)[lane] = rwd[0];
        if (a.reward_raw) (a.reward_raw + row0)[lane] = rraw[0];
        (a.done + row0)[lane] = dn[0] ? 1 : 0;
      } else {
        (reinterpret_cast<float2*>(a.reward) + row0)[lane] = make_float2(rwd[0], rwd[NAG - 1]);
        if (a.reward_raw) (reinterpret_cast<float2*>(a.reward_raw) + row0)[lane] = make_float2(rraw[0], rraw[NAG - 1]);
        (reinterpret_cast<uchar2*>(a.done) + row0)[lane] = make_uchar2(dn[0] ? 1 : 0, dn[NAG - 1] ? 1 : 0);
      }
      if (a.truncated) (a.truncated + row0)[lane] = trunc ? 1 : 0;
    }
  }

  // ---- write the working set back ----
  if (active) {
    store_state<XV, QW, T>(a, first, lane, w);
    if (KIND != QR_KIND_QUAD) {
      const SoA<float> integ(a.integ, 8, L);
#pragma unroll
      for (int f = 0; f < 8; ++f) integ.store(f, ufirst, lane, w.integ[f]);
    }
    if (a.steps) (a.steps + first)[lane] = steps;
    if constexpr (TRAJ) {
      const SoA<float> traj(a.traj, 8, L);
      traj.store(0, ufirst, lane, tr[0]);
      if (params_dirty || (a.flags & QR_FLAG_AUTO_RESET)) {  // the rest changes only at a reset
#pragma unroll
        for (int f = 1; f < 7; ++f) traj.store(f, ufirst, lane, tr[f]);
      }
    }
    if (params_dirty) {
      const SoA<float> prm(a.params, 6, L);
#pragma unroll
      for (int f = 0; f < 6; ++f) prm.store(f, ufirst, lane, w.prm[f]);
    }
  }
}

// get_norm_error_state on the current state (quad.py:421-466)
template <int KIND, typename XV, typename QW>
__global__ __launch_bounds__(64) void error_obs_kernel(const Args a) {
  using T = QW;
  using KT = KindTraits<KIND>;
  constexpr int B = 64, D0 = KT::D0, D1 = KT::D1 ? KT::D1 : 1;
  __shared__ __attribute__((aligned(16))) float smem[B * D0];
  const int tid = threadIdx.x;
  const int64_t first = (int64_t)blockIdx.x * B;
  const int64_t i = first + tid;
  const int64_t N = a.n, L = a.ld;
  const int rows = (int)((N - first) < B ? (N - first) : B);
  const bool active = tid < rows;
  Work<T> w;
  idle_work(w);
  if (active) {
    load_state<XV, QW, T>(a, first, (unsigned)tid, w);
    if (a.goal) {
#pragma unroll
      for (int f = 0; f < 12; ++f) w.goal[f] = a.goal[(int64_t)f * L + i];
    }
#pragma unroll
    for (int f = 0; f < 8; ++f) w.integ[f] = a.integ[(int64_t)f * L + i];
  }
  T R[9];
  quat_to_R(&w.y[3], R);
  float o0[D0];
  float o1[D1];
  error_obs<KIND, T>(w, R, a.c, o0, o1);
  store_rows<B, D0>(a.obs0 + first * D0, o0, smem, tid, rows);
  if constexpr (KT::D1 > 0) store_rows<B, D1>(a.obs1 + first * D1, o1, smem, tid, rows);
  if (active) {
#pragma unroll
    for (int f = 0; f < 8; ++f) a.integ[(int64_t)f * L + i] = w.integ[f];
  }
}

// QuadEnv.reset for masked envs
template <typename XV, typename QW>
__global__ __launch_bounds__(64) void reset_kernel(const Args a) {
  using T = QW;
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int64_t N = a.n, L = a.ld;
  if (i >= N) return;
  if (a.mask && !a.mask[i]) return;
  const int32_t episode = a.episode[i] + 1;
  const bool eval = (a.flags & QR_FLAG_EVAL_RESET) != 0;
  const bool randomise = !eval && !(a.flags & QR_FLAG_NO_UDM);
  Work<T> w;
  Draws d;
  draw20(d, a.seed, (uint64_t)(a.env_offset + i), (uint32_t)episode);
  sample_reset(w, d, randomise, eval, a.c);
  store_state<XV, QW, T>(a, (int64_t)blockIdx.x * 64, threadIdx.x, w);
  if (a.params) {
#pragma unroll
    for (int f = 0; f < 6; ++f) a.params[(int64_t)f * L + i] = w.prm[f];
  }
  if (a.integ) {
#pragma unroll
    for (int f = 0; f < 8; ++f) a.integ[(int64_t)f * L + i] = 0.f;
  }
  if (a.steps) a.steps[i] = 0;
  a.episode[i] = episode;
}

// get_current_state: 13-word internal state -> the reference's float64 18-vector rows
template <typename XV, typename QW>
__global__ __launch_bounds__(64) void get_state_kernel(const Args a) {
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= a.n) return;
  Work<double> w;
  load_state<XV, QW, double>(a, (int64_t)blockIdx.x * 64, threadIdx.x, w);
  double R[9];
  quat_to_R(&w.y[3], R);
  double* o = a.rows_out + i * 18;
#pragma unroll
  for (int j = 0; j < 3; ++j) { o[j] = w.x[j]; o[3 + j] = w.y[j]; o[15 + j] = w.y[7 + j]; }
#pragma unroll
  for (int j = 0; j < 9; ++j) o[6 + j] = R[j];
}

// state injection: float64 18-vector rows -> 13-word internal state (R -> nearest rotation -> q)
template <typename XV, typename QW>
__global__ __launch_bounds__(64) void set_state_kernel(const Args a) {
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= a.n) return;
  if (a.mask && !a.mask[i]) return;
  const double* r = a.rows_in + i * 18;
  Work<double> w;
#pragma unroll
  for (int j = 0; j < 3; ++j) { w.x[j] = r[j]; w.y[j] = r[3 + j]; w.y[7 + j] = r[15 + j]; }
  double q[4];
  R_to_quat(r + 6, q);
#pragma unroll
  for (int j = 0; j < 4; ++j) w.y[3 + j] = q[j];
  store_state<XV, QW, double>(a, (int64_t)blockIdx.x * 64, threadIdx.x, w);
}

// mark_traj_start for masked envs, from the current state
template <typename XV, typename QW>
__global__ __launch_bounds__(64) void traj_start_kernel(const Args a) {
  using T = QW;
  const int64_t first = (int64_t)blockIdx.x * 64;
  const unsigned lane = threadIdx.x;
  const int64_t i = first + lane;
  if (i >= a.n) return;
  if (a.mask && !a.mask[i]) return;
  Work<T> w;
  load_state<XV, QW, T>(a, first, lane, w);
  float th, tt, wb;
  if (a.draws) {
    th = a.draws[i]; tt = a.draws[a.n + i]; wb = a.draws[2 * a.n + i];
  } else {
    Draws d;
    draw20(d, a.seed, (uint64_t)(a.env_offset + i), (uint32_t)a.episode[i]);
    traj_draws(d.r[19], th, tt, wb);
  }
  float tr[8];
  traj_start(w, tr, a.goal_mode, th, tt, wb);
  const SoA<float> traj(a.traj, 8, a.ld);
#pragma unroll
  for (int f = 0; f < 8; ++f) traj.store(f, (unsigned)first, lane, tr[f]);
}

// get_desired for the current state: rows [N][15] = xd, vd, b1d, b1d_dot, Wd
template <typename XV, typename QW>
__global__ __launch_bounds__(64) void get_desired_kernel(const Args a) {
  using T = QW;
  const int64_t first = (int64_t)blockIdx.x * 64;
  const unsigned lane = threadIdx.x;
  const int64_t i = first + lane;
  if (i >= a.n) return;
  if (a.mask && !a.mask[i]) return;
  Work<T> w;
  idle_work(w);
  load_state<XV, QW, T>(a, first, lane, w);
  const SoA<float> traj(a.traj, 8, a.ld);
  float tr[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) tr[f] = traj.load(f, (unsigned)first, lane);
  float b1d_dot[3];
  traj_goal(w, tr, a.goal_mode, a.c, b1d_dot);
  traj.store(0, (unsigned)first, lane, tr[0]);
  if (a.goal_rows) {
    float* o = a.goal_rows + i * 15;
#pragma unroll
    for (int j = 0; j < 9; ++j) o[j] = w.goal[j];
#pragma unroll
    for (int j = 0; j < 3; ++j) { o[9 + j] = b1d_dot[j]; o[12 + j] = w.goal[9 + j]; }
  }
  if (a.store_goal && a.goal) {
    const SoA<float> goal(a.goal, 12, a.ld);
#pragma unroll
    for (int f = 0; f < 12; ++f) goal.store(f, (unsigned)first, lane, w.goal[f]);
  }
}

// ------------------------------------------------------------------------------------
// GAE reverse scan (algos/ppo/ppo.py:134-146): one lane per (env, agent) column, T steps.
// The recurrence is serial in t but the loads are not: they are issued kU steps ahead so that
// a wave keeps kU rows in flight instead of paying one memory round-trip per step.
// ------------------------------------------------------------------------------------
struct GaeArgs {
  const float* reward; const uint8_t* done; const float* value; const float* next_value;
  float* advantage; float* td_target; double* partials;
  int64_t m; int32_t T; float gamma; float lam;
};

__global__ __launch_bounds__(64) void gae_kernel(const GaeArgs g) {
  constexpr int kU = 8;
  const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const bool active = j < g.m;
  const int64_t M = g.m;
  float adv = 0.0f;
  double s1 = 0.0, s2 = 0.0;
  if (active) {
    float vnext = g.next_value ? 0.0f : g.value[(int64_t)g.T * M + j];  // bootstrap row
    for (int t0 = g.T; t0 > 0; t0 -= kU) {
      const int nb = t0 < kU ? t0 : kU;
      float r[kU], v[kU], vn[kU];
      uint8_t d[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (u < nb) {
          const int64_t idx = (int64_t)(t0 - 1 - u) * M + j;
          r[u] = g.reward[idx]; d[u] = g.done[idx]; v[u] = g.value[idx];
          vn[u] = g.next_value ? g.next_value[idx] : 0.0f;
        }
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (u < nb) {
          const int64_t idx = (int64_t)(t0 - 1 - u) * M + j;
          const float nd = d[u] ? 0.0f : 1.0f;
          const float vnx = g.next_value ? vn[u] : vnext;
          const float delta = r[u] + g.gamma * vnx * nd - v[u];
          adv = delta + g.gamma * nd * g.lam * adv;
          g.advantage[idx] = adv;
          g.td_target[idx] = adv + v[u];
          s1 += (double)adv; s2 += (double)adv * (double)adv;
          vnext = v[u];
        }
      }
    }
  }
  if (g.partials) {  // wave reduction (DPP/bpermute shuffles), one pair of doubles per workgroup
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off); s2 += __shfl_down(s2, off); }
    if (threadIdx.x == 0) { g.partials[2 * (int64_t)blockIdx.x] = s1; g.partials[2 * (int64_t)blockIdx.x + 1] = s2; }
  }
}

// ------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------
static void fill_coeffs(Coeffs& o, const QrCoeffs& q) {
  o.Cx = q.Cx; o.CIx = q.CIx; o.Cv = q.Cv; o.Cb1 = q.Cb1; o.CIb1 = q.CIb1; o.CW = q.CW; o.Cw12 = q.Cw12; o.CW3 = q.CW3;
  o.alpha = q.alpha; o.beta = q.beta; o.dt = q.dt;
  o.x_lim = q.x_lim; o.v_lim = q.v_lim; o.W_lim = q.W_lim; o.eIx_lim = q.eIx_lim; o.eIb1_lim = q.eIb1_lim;
  const double lim = q.euler_lim_deg * kPi / 180.0;
  o.sin_euler_lim = sin(lim); o.tan_euler_lim = tan(lim); o.udm = q.udm_fraction;
  o.rmin_mono = -ceil(q.Cx + q.CIx + q.Cv + q.Cb1 + q.CIb1 + q.CW);  // quad.py:81
  o.rmin_1 = -ceil(q.Cx + q.CIx + q.Cv + q.Cw12);                    // quad.py:85
  o.rmin_2 = -ceil(q.Cb1 + q.CW3 + q.CIb1);                          // quad.py:88
  o.inv_x_lim = 1.0 / q.x_lim; o.inv_v_lim = 1.0 / q.v_lim; o.inv_W_lim = 1.0 / q.W_lim;
  o.inv_eIx_lim = 1.0 / q.eIx_lim; o.inv_eIb1_lim = 1.0 / q.eIb1_lim;
  o.inv_nrmin_mono = -1.0 / o.rmin_mono; o.inv_nrmin_1 = -1.0 / o.rmin_1; o.inv_nrmin_2 = -1.0 / o.rmin_2;
  const double T8 = q.eight_T > 0 ? q.eight_T : 9.0;
  o.e8_w1 = (float)(2.0 * kPi / T8); o.e8_w2 = (float)(4.0 * kPi / T8);                // :102-103
  o.e8_k = (float)(-log(q.eight_eps > 0 ? q.eight_eps : 0.01) / T8);                   // :107-108
  o.e8_A1 = (float)q.eight_A1; o.e8_A2 = (float)q.eight_A2; o.e8_wb = (float)q.eight_w_b1d; o.e8_alt = (float)q.eight_alt_d;
  o.e8_tmax = (float)(q.eight_count * T8);                                             // :436
  o.inv_w_adapt = q.w_adapt > 0 ? 1.0 / q.w_adapt : 0.0;
}

static int fill_env(Args& a, const QrEnv* e) {
  if (!e) return QR_E_NULL;
  if (e->kind < 0 || e->kind > 2 || e->layout < 0 || e->layout > 2) return QR_E_KIND;
  if (e->num_envs < 0 || (e->field_stride != 0 && (e->field_stride < e->num_envs || (e->field_stride & 3)))) return QR_E_SIZE;
  if ((e->field_stride > 0 ? e->field_stride : e->num_envs) > (int64_t)0x7fffffff / (12 * 8)) return QR_E_SIZE;  // SoA buffers < 2 GiB (32-bit buffer offsets)
  if (e->goal_mode < 0 || e->goal_mode > 3) return QR_E_KIND;
  if (e->goal_mode != QR_GOAL_EXTERNAL && !e->traj) return QR_E_NULL;
  if (!e->pos_vel || !e->att_rate) return QR_E_NULL;
  if ((reinterpret_cast<uintptr_t>(e->pos_vel) | reinterpret_cast<uintptr_t>(e->att_rate)) & 15u) return QR_E_ALIGN;
  a.pos_vel = e->pos_vel; a.att_rate = e->att_rate; a.integ = e->integ; a.params = e->params; a.goal = e->goal;
  a.traj = e->traj; a.goal_mode = e->goal_mode;
  a.episode = e->episode; a.steps = e->steps;
  a.n = e->num_envs; a.ld = e->field_stride > 0 ? e->field_stride : e->num_envs;
  a.env_offset = e->env_offset; a.seed = e->seed;
  a.max_episode_steps = e->max_episode_steps; a.flags = e->flags;
  fill_coeffs(a.c, e->coeffs);
  return 0;
}

// Workgroup size: one wavefront per workgroup at every batch size.  Small batches: every SIMD
// gets a wave (N = 65 536 -> 1024 workgroups) and the LDS transposes need no cross-wave
// barrier.  Large batches: measured faster than 256-thread workgroups too (1 M envs: 38.7 vs
// 42.5 us Quad-v0, 82 vs 114 us Decoupled) — the dispatcher's ~3.6 workgroups/ns is far above
// what a bandwidth-bound launch needs, and barriers of 4-wave groups at 1-2 waves/SIMD stall.
static inline int pick_block(int64_t) { return 64; }

template <int KIND, typename XV, typename QW>
static void launch_kind(const Args& a, hipStream_t s) {
  const dim3 grid((unsigned)((a.n + 63) / 64));
  // Rate adaptivity can only trigger when an env starts a step with max|W_i| > w_adapt.  With
  // AUTO_RESET every env whose rate error left its bound was re-sampled at the end of the step
  // that took it there (done): Quad-v0 |W_i| < W_lim, Coupled |W_i - Wd_i| < W_lim, Decoupled
  // |W - Wd| < 2 W_lim (|ew12_i| < W_lim and |eW3| < W_lim).  For goal rates |Wd| <= W_lim / 2
  // and w_adapt >= 2.5 W_lim (the default 16 rad/s is) the plain kernel computes the same bits.
  const bool adapt = a.c.inv_w_adapt > 0 &&
                     (!(a.flags & QR_FLAG_AUTO_RESET) || a.c.inv_w_adapt * a.c.W_lim * 2.5 > 1.0);
#define QR_STEP_ARGS a.pos_vel, a.att_rate, a.action, a.params, a.integ, a.n, a.ld, a
  if constexpr (KIND != QR_KIND_QUAD) {
    if (a.act_out != nullptr) {  // qr_rollout_actor
      if (a.goal_mode != QR_GOAL_EXTERNAL) hipLaunchKernelGGL((step_kernel<KIND, XV, QW, 64, true, true, true>), grid, dim3(64), 0, s, QR_STEP_ARGS);
      else hipLaunchKernelGGL((step_kernel<KIND, XV, QW, 64, false, true, true>), grid, dim3(64), 0, s, QR_STEP_ARGS);
      return;
    }
  }
  if (a.goal_mode != QR_GOAL_EXTERNAL) hipLaunchKernelGGL((step_kernel<KIND, XV, QW, 64, true, true>), grid, dim3(64), 0, s, QR_STEP_ARGS);
  else if (adapt) hipLaunchKernelGGL((step_kernel<KIND, XV, QW, 64, false, true>), grid, dim3(64), 0, s, QR_STEP_ARGS);
  else hipLaunchKernelGGL((step_kernel<KIND, XV, QW, 64, false, false>), grid, dim3(64), 0, s, QR_STEP_ARGS);
#undef QR_STEP_ARGS
}

template <typename XV, typename QW>
static int launch_step(const Args& a, int kind, hipStream_t s) {
  if (a.n == 0) return 0;
  switch (kind) {
    case QR_KIND_QUAD: launch_kind<QR_KIND_QUAD, XV, QW>(a, s); break;
    case QR_KIND_COUPLED: launch_kind<QR_KIND_COUPLED, XV, QW>(a, s); break;
    default: launch_kind<QR_KIND_DECOUPLED, XV, QW>(a, s); break;
  }
  return (int)hipGetLastError();
}

#define QR_DISPATCH_LAYOUT(layout, CALL)                                        \
  switch (layout) {                                                             \
    case QR_LAYOUT_MIXED: { using XV = float; using QW = double; CALL; } break; \
    case QR_LAYOUT_F64:   { using XV = double; using QW = double; CALL; } break; \
    default:              { using XV = float; using QW = float; CALL; } break;  \
  }

template <typename XV, typename QW>
static void launch_error_obs(const Args& a, int kind, unsigned grid, hipStream_t s) {
  if (kind == QR_KIND_COUPLED) hipLaunchKernelGGL((error_obs_kernel<QR_KIND_COUPLED, XV, QW>), dim3(grid), dim3(64), 0, s, a);
  else hipLaunchKernelGGL((error_obs_kernel<QR_KIND_DECOUPLED, XV, QW>), dim3(grid), dim3(64), 0, s, a);
}
template <typename XV, typename QW>
static void launch_reset(const Args& a, unsigned grid, hipStream_t s) {
  hipLaunchKernelGGL((reset_kernel<XV, QW>), dim3(grid), dim3(64), 0, s, a);
}
template <typename XV, typename QW>
static void launch_get_state(const Args& a, unsigned grid, hipStream_t s) {
  hipLaunchKernelGGL((get_state_kernel<XV, QW>), dim3(grid), dim3(64), 0, s, a);
}
template <typename XV, typename QW>
static void launch_set_state(const Args& a, unsigned grid, hipStream_t s) {
  hipLaunchKernelGGL((set_state_kernel<XV, QW>), dim3(grid), dim3(64), 0, s, a);
}
template <typename XV, typename QW>
static void launch_traj_start(const Args& a, unsigned grid, hipStream_t s) {
  hipLaunchKernelGGL((traj_start_kernel<XV, QW>), dim3(grid), dim3(64), 0, s, a);
}
template <typename XV, typename QW>
static void launch_get_desired(const Args& a, unsigned grid, hipStream_t s) {
  hipLaunchKernelGGL((get_desired_kernel<XV, QW>), dim3(grid), dim3(64), 0, s, a);
}

static int fill_actor(ActorW& w, const QrActor& q, int obs_dim, int hidden, int action_dim) {
  if (q.obs_dim != obs_dim || q.hidden_dim != hidden || q.action_dim != action_dim) return QR_E_SIZE;
  if (!q.fc1_w || !q.fc1_b || !q.fc2_w || !q.fc2_b || !q.mean_w || !q.mean_b || !q.log_std) return QR_E_NULL;
  w.fc1_w = q.fc1_w; w.fc1_b = q.fc1_b; w.fc2_w = q.fc2_w; w.fc2_b = q.fc2_b;
  w.mean_w = q.mean_w; w.mean_b = q.mean_b; w.log_std = q.log_std;
  return 0;
}

static int do_rollout(const QrEnv* env, const float* action, const QrPolicyRollout* pol, int32_t n_steps, int32_t substeps,
                      const QrStepOut* out, void* stream) {
  Args a{};
  if (int rc = fill_env(a, env)) return rc;
  if ((!action && !pol) || !out || !out->reward || !out->done) return QR_E_NULL;
  if (substeps < 1 || n_steps < 1) return QR_E_SIZE;
  if (env->kind != QR_KIND_QUAD && (!env->integ || !out->obs0)) return QR_E_NULL;
  if (env->kind == QR_KIND_DECOUPLED && !out->obs1) return QR_E_NULL;
  if ((env->flags & QR_FLAG_AUTO_RESET) && !env->episode) return QR_E_NULL;
  if (pol) {
    if (env->kind == QR_KIND_QUAD) return QR_E_KIND;
    if (!pol->actors || !pol->obs0_in || !pol->action_out) return QR_E_NULL;
    if (env->kind == QR_KIND_COUPLED) {
      if (int rc = fill_actor(a.actor[0], pol->actors[0], 23, 16, 4)) return rc;
    } else {
      if (!pol->obs1_in) return QR_E_NULL;
      if (int rc = fill_actor(a.actor[0], pol->actors[0], 15, 16, 4)) return rc;
      if (int rc = fill_actor(a.actor[1], pol->actors[1], 3, 4, 1)) return rc;
    }
    const uintptr_t amask = env->kind == QR_KIND_DECOUPLED ? 3u : 15u;  // A = 4: one 16-byte store per lane
    if ((reinterpret_cast<uintptr_t>(pol->action_out) | reinterpret_cast<uintptr_t>(pol->logprob_out)) & amask) return QR_E_ALIGN;
    if (!(pol->max_action > 0.0f)) return QR_E_SIZE;
    a.obs0_in = pol->obs0_in; a.obs1_in = pol->obs1_in; a.noise = pol->noise;
    a.act_out = pol->action_out; a.logp_out = pol->logprob_out;
    a.noise_seed = pol->noise_seed; a.step_base = pol->step_base;
    a.max_action = pol->max_action; a.deterministic = pol->deterministic;
  } else {
    // action rows: A = 4 is read with one 16-byte load per lane; A = 5 (DECOUPLED) with dword loads
    if (reinterpret_cast<uintptr_t>(action) & (env->kind == QR_KIND_DECOUPLED ? 3u : 15u)) return QR_E_ALIGN;
  }
  a.action = action; a.obs0 = out->obs0; a.obs1 = out->obs1;
  a.reward = out->reward; a.reward_raw = out->reward_raw; a.done = out->done; a.truncated = out->truncated;
  a.n_steps = n_steps; a.substeps = substeps;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rc = 0;
  QR_DISPATCH_LAYOUT(env->layout, (rc = launch_step<XV, QW>(a, env->kind, s)));
  return rc;
}

}  // namespace qr

extern "C" {

int qr_abi_version(void) { return QR_ABI_VERSION; }

void qr_default_coeffs(QrCoeffs* c) {
  if (!c) return;
  c->Cx = 6.0; c->CIx = 0.1; c->Cv = 0.4; c->Cb1 = 6.0; c->CIb1 = 0.1; c->CW = 0.6;  // args_parse.py:23-31, quad.py:80
  c->Cw12 = 0.6; c->CW3 = 0.1; c->alpha = 0.01; c->beta = 0.05;
  c->dt = 1.0 / 200.0;
  c->x_lim = 1.0; c->v_lim = 4.0; c->W_lim = 2.0 * qr::kPi;
  c->eIx_lim = 3.0; c->eIb1_lim = 3.0; c->euler_lim_deg = 85.0; c->udm_fraction = 0.1;
  c->eight_T = 9.0; c->eight_A1 = 1.5; c->eight_A2 = 1.0; c->eight_w_b1d = 0.349066; c->eight_alt_d = -0.6;  // trajectory_generator.py:98-110
  c->eight_eps = 0.01; c->eight_count = 3.0;
  c->w_adapt = 16.0;
}

int qr_step(const QrEnv* env, const float* action, int32_t substeps, const QrStepOut* out, void* stream) {
  return qr::do_rollout(env, action, nullptr, 1, substeps, out, stream);
}

int qr_rollout(const QrEnv* env, const float* action, int32_t n_steps, int32_t substeps, const QrStepOut* out, void* stream) {
  return qr::do_rollout(env, action, nullptr, n_steps, substeps, out, stream);
}

int qr_rollout_actor(const QrEnv* env, const QrPolicyRollout* policy, int32_t n_steps, int32_t substeps, const QrStepOut* out,
                     void* stream) {
  if (!policy) return QR_E_NULL;
  return qr::do_rollout(env, nullptr, policy, n_steps, substeps, out, stream);
}

int qr_error_obs(const QrEnv* env, float* obs0, float* obs1, void* stream) {
  qr::Args a{};
  if (int rc = qr::fill_env(a, env)) return rc;
  if (env->kind == QR_KIND_QUAD) return QR_E_KIND;
  if (!env->integ || !obs0 || (env->kind == QR_KIND_DECOUPLED && !obs1)) return QR_E_NULL;
  a.obs0 = obs0; a.obs1 = obs1;
  const unsigned grid = (unsigned)((a.n + 63) / 64);
  if (grid == 0) return 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  QR_DISPATCH_LAYOUT(env->layout, (qr::launch_error_obs<XV, QW>(a, env->kind, grid, s)));
  return (int)hipGetLastError();
}

int qr_reset(const QrEnv* env, const uint8_t* mask, void* stream) {
  qr::Args a{};
  if (int rc = qr::fill_env(a, env)) return rc;
  if (!env->episode) return QR_E_NULL;
  a.mask = mask;
  const unsigned grid = (unsigned)((a.n + 63) / 64);
  if (grid == 0) return 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  QR_DISPATCH_LAYOUT(env->layout, (qr::launch_reset<XV, QW>(a, grid, s)));
  return (int)hipGetLastError();
}

int qr_get_state(const QrEnv* env, double* rows, void* stream) {
  qr::Args a{};
  if (int rc = qr::fill_env(a, env)) return rc;
  if (!rows) return QR_E_NULL;
  a.rows_out = rows;
  const unsigned grid = (unsigned)((a.n + 63) / 64);
  if (grid == 0) return 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  QR_DISPATCH_LAYOUT(env->layout, (qr::launch_get_state<XV, QW>(a, grid, s)));
  return (int)hipGetLastError();
}

int qr_set_state(const QrEnv* env, const double* rows, const uint8_t* mask, void* stream) {
  qr::Args a{};
  if (int rc = qr::fill_env(a, env)) return rc;
  if (!rows) return QR_E_NULL;
  a.rows_in = rows; a.mask = mask;
  const unsigned grid = (unsigned)((a.n + 63) / 64);
  if (grid == 0) return 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  QR_DISPATCH_LAYOUT(env->layout, (qr::launch_set_state<XV, QW>(a, grid, s)));
  return (int)hipGetLastError();
}

int qr_traj_start(const QrEnv* env, const uint8_t* mask, const float* draws, void* stream) {
  qr::Args a{};
  if (int rc = qr::fill_env(a, env)) return rc;
  if (env->goal_mode == QR_GOAL_EXTERNAL) return QR_E_KIND;
  if (!draws && !env->episode) return QR_E_NULL;
  a.mask = mask; a.draws = draws;
  const unsigned grid = (unsigned)((a.n + 63) / 64);
  if (grid == 0) return 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  QR_DISPATCH_LAYOUT(env->layout, (qr::launch_traj_start<XV, QW>(a, grid, s)));
  return (int)hipGetLastError();
}

int qr_get_desired(const QrEnv* env, const uint8_t* mask, float* rows, int32_t store_goal, void* stream) {
  qr::Args a{};
  if (int rc = qr::fill_env(a, env)) return rc;
  if (env->goal_mode == QR_GOAL_EXTERNAL) return QR_E_KIND;
  if (!rows && !store_goal) return QR_E_NULL;
  if (store_goal && !env->goal) return QR_E_NULL;
  a.goal_rows = rows; a.store_goal = store_goal; a.mask = mask;
  const unsigned grid = (unsigned)((a.n + 63) / 64);
  if (grid == 0) return 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  QR_DISPATCH_LAYOUT(env->layout, (qr::launch_get_desired<XV, QW>(a, grid, s)));
  return (int)hipGetLastError();
}

int qr_gae(const float* reward, const uint8_t* done, const float* value, const float* next_value, int32_t n_steps,
           int64_t n_cols, float gamma, float lam, float* advantage, float* td_target, double* partials, void* stream) {
  if (!reward || !done || !value || !advantage || !td_target) return QR_E_NULL;
  if (n_steps < 1 || n_cols < 0) return QR_E_SIZE;
  if (n_cols == 0) return 0;
  qr::GaeArgs g{reward, done, value, next_value, advantage, td_target, partials, n_cols, n_steps, gamma, lam};
  hipLaunchKernelGGL(qr::gae_kernel, dim3((unsigned)((n_cols + 63) / 64)), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), g);
  return (int)hipGetLastError();
}

const char* qr_step_kernel_info(int32_t kind, int32_t layout, int64_t num_envs, int32_t* grid, int32_t* block) {
  const int b = qr::pick_block(num_envs);
  if (grid) *grid = (int32_t)((num_envs + b - 1) / b);
  if (block) *block = b;
  (void)layout;
  switch (kind) {
    case QR_KIND_QUAD: return "qr::step_kernel<0,...>";
    case QR_KIND_COUPLED: return "qr::step_kernel<1,...>";
    case QR_KIND_DECOUPLED: return "qr::step_kernel<2,...>";
    default: return "";
  }
}

}  // extern "C"
