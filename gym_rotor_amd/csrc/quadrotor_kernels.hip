// quadrotor_kernels.hip — fused env.step() kernels for gfx950 (MI355X / CDNA4).
//
// One lane = one quadrotor.  A launch does, per env and per env-step, everything the
// reference's QuadEnv.step template does (gym_rotor/envs/quad.py:142-168):
//   action map / motor mixing -> S fixed RK4 substeps of the 18-dim rigid-body ODE on
//   R^3 x R^3 x SO(3) x R^3 (quad.py:321-335) with zero-order-hold (f, M) -> SO(3)
//   re-orthonormalisation -> error observation + trapezoid integrators (quad.py:421-466)
//   -> reward -> np.interp normalisation -> done -> crash override [-> auto-reset].
// The whole working set (18 state words, 8 integrator words, goal, parameters) lives in
// VGPRs across all substeps and, in qr_rollout, across env-steps: HBM is touched once in
// and once out.  Per-env SoA buffers are read/written with lane-contiguous accesses;
// caller-facing AoS rows (actions, observations) go through LDS so that global traffic
// is issued as linear 16-byte-per-lane stores.  There is no contraction larger than
// 3x3 * 3x3 anywhere, so no MFMA; the kernel is bound by HBM / launch latency.
//
// Written directly for CDNA4: 64-lane wavefronts, one wavefront per workgroup so that
// N = 65 536 envs still gives 1024 workgroups (4 per CU, one per SIMD) and the LDS
// transposes need no cross-wave barrier traffic.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "quadrotor_hip.h"

namespace qr {

constexpr int kBlock = 64;  // one wavefront per workgroup

// ------------------------------------------------------------------------------------
// Kernel argument block (passed by value in kernarg memory)
// ------------------------------------------------------------------------------------
struct Coeffs {  // double-precision copy of QrCoeffs + derived reward floors
  double Cx, CIx, Cv, Cb1, CIb1, CW, Cw12, CW3, alpha, beta, dt;
  double x_lim, v_lim, W_lim, eIx_lim, eIb1_lim;
  double sin_euler_lim, tan_euler_lim, udm;
  double rmin_mono, rmin_1, rmin_2;
};

struct Args {
  // per-env buffers
  void* state;
  float* integ;
  float* params;
  float* goal;
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
  int64_t n;
  int64_t env_offset;
  uint64_t seed;
  int32_t n_steps;
  int32_t substeps;
  int32_t max_episode_steps;
  uint32_t flags;
  Coeffs c;
};

// Nominal parameters (quad.py:28-33)
constexpr double kMnom = 2.15, kDnom = 0.23, kJ1nom = 0.022, kJ3nom = 0.035, kCtfNom = 0.0135,
                 kCtwNom = 2.2, kG = 9.81, kMinForce = 0.5;
constexpr double kPi = 3.14159265358979323846;

template <typename T>
struct Phys {  // per-env physical parameters + what set_random_parameters derives (quad.py:389-404)
  T m, d, J1, J3, ctf, ctw;
  T max_force, avrg_act, scale_act;
  __device__ __forceinline__ void derive() {
    const T hover = m * T(kG) / T(4);
    max_force = ctw * hover;
    avrg_act = (T(kMinForce) + max_force) / T(2);
    scale_act = max_force - avrg_act;
  }
};

template <typename T> __device__ __forceinline__ T clampT(T v, T lo, T hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ------------------------------------------------------------------------------------
// Philox4x32-10 counter-based RNG (Salmon et al., SC'11): stateless, keyed by
// (seed, global env id, episode) so that resets do not depend on launch geometry.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t (&ctr)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * ctr[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * ctr[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ ctr[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ ctr[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    ctr[0] = n0; ctr[1] = n1; ctr[2] = n2; ctr[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

struct Draws {  // 20 uniforms in (-1, 1) / (0, 1)
  uint32_t r[20];
  __device__ __forceinline__ double u01(int i) const { return ((double)r[i] + 0.5) * (1.0 / 4294967296.0); }
  __device__ __forceinline__ double sym(int i) const { return 2.0 * u01(i) - 1.0; }
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

// QuadEnv.reset + sample_init_error + set_random_parameters (quad.py:171-222, 338-404).
// Draw order: 0..5 m,d,J1,J3,c_tf,c_tw; 6 yaw; 7 zero-error branch; 8..10 x; 11..13 v;
// 14..16 W; 17,18 roll,pitch.
template <typename T>
__device__ void sample_reset(T (&y)[18], Phys<T>& ph, bool randomise, bool eval, const Coeffs& c,
                             uint64_t seed, uint64_t gid, uint32_t episode) {
  Draws d;
  draw20(d, seed, gid, episode);
  if (randomise) {
    const double p = c.udm;
    ph.m = T((float)(kMnom * (1.0 + p * d.sym(0))));
    ph.d = T((float)(kDnom * (1.0 + p * d.sym(1))));
    ph.J1 = T((float)(kJ1nom * (1.0 + p * d.sym(2))));
    ph.J3 = T((float)(kJ3nom * (1.0 + p * d.sym(3))));
    ph.ctf = T((float)(kCtfNom * (1.0 + p * d.sym(4))));
    ph.ctw = T((float)(kCtwNom * (1.0 + 0.5 * p * d.sym(5))));
  } else {
    ph.m = T(kMnom); ph.d = T(kDnom); ph.J1 = T(kJ1nom); ph.J3 = T(kJ3nom); ph.ctf = T(kCtfNom); ph.ctw = T(kCtwNom);
  }
  ph.derive();
  const double yaw = kPi * d.sym(6);
  double ix, iv, iR, iW;
  if (eval) {  // quad.py:352-356
    ix = 0.4; iv = 0.0; iR = 0.0; iW = 0.0;
  } else if (d.u01(7) < 0.2) {  // quad.py:342-346
    ix = 0.0; iv = 0.0; iR = 0.0; iW = 0.0;
  } else {  // quad.py:348-351
    ix = 0.6; iv = c.v_lim * 0.5; iR = 50.0 * kPi / 180.0; iW = c.W_lim * 0.5;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    y[j] = T(ix * d.sym(8 + j));
    y[3 + j] = T(iv * d.sym(11 + j));
    y[15 + j] = T(iW * d.sym(14 + j));
  }
  const double roll = iR * d.sym(17), pitch = iR * d.sym(18);
  double sr, cr, sp, cp, sy, cy;
  sincos(roll, &sr, &cr); sincos(pitch, &sp, &cp); sincos(yaw, &sy, &cy);
  // R = Rz(yaw) Ry(pitch) Rx(roll)  (scipy 'xyz' extrinsic, quad.py:199), column-major
  y[6] = T(cy * cp);                y[7] = T(sy * cp);                y[8] = T(-sp);
  y[9] = T(cy * sp * sr - sy * cr); y[10] = T(sy * sp * sr + cy * cr); y[11] = T(cp * sr);
  y[12] = T(cy * sp * cr + sy * sr); y[13] = T(sy * sp * cr - cy * sr); y[14] = T(cp * cr);
}

// ------------------------------------------------------------------------------------
// SO(3) helpers
// ------------------------------------------------------------------------------------
// One Newton-Schulz step R <- R (3I - R^T R)/2: removes first-order orthogonality drift.
template <typename T>
__device__ __forceinline__ void newton_schulz(T* R /* column-major 9 */) {
  const T g00 = R[0] * R[0] + R[1] * R[1] + R[2] * R[2];
  const T g11 = R[3] * R[3] + R[4] * R[4] + R[5] * R[5];
  const T g22 = R[6] * R[6] + R[7] * R[7] + R[8] * R[8];
  const T g01 = R[0] * R[3] + R[1] * R[4] + R[2] * R[5];
  const T g02 = R[0] * R[6] + R[1] * R[7] + R[2] * R[8];
  const T g12 = R[3] * R[6] + R[4] * R[7] + R[5] * R[8];
  const T h = T(0.5);
  const T s00 = h * (T(3) - g00), s11 = h * (T(3) - g11), s22 = h * (T(3) - g22);
  const T s01 = -h * g01, s02 = -h * g02, s12 = -h * g12;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const T a = R[i], b = R[3 + i], cc = R[6 + i];
    R[i] = a * s00 + b * s01 + cc * s02;
    R[3 + i] = a * s01 + b * s11 + cc * s12;
    R[6 + i] = a * s02 + b * s12 + cc * s22;
  }
}

// ensure_SO3 (quad_utils.py:123-142): if R^T R or det R is off by more than 1e-5, replace R
// by the nearest rotation (the polar factor U V^T the reference gets from an SVD).  The
// polar factor is computed with the scaled Newton iteration X <- (X + X^-T)/2, which
// converges quadratically to the same matrix for det R > 0.
template <typename T>
__device__ void so3_guard(T* R) {
  const T tol = T(1e-5);
  const T g00 = R[0] * R[0] + R[1] * R[1] + R[2] * R[2];
  const T g11 = R[3] * R[3] + R[4] * R[4] + R[5] * R[5];
  const T g22 = R[6] * R[6] + R[7] * R[7] + R[8] * R[8];
  const T g01 = R[0] * R[3] + R[1] * R[4] + R[2] * R[5];
  const T g02 = R[0] * R[6] + R[1] * R[7] + R[2] * R[8];
  const T g12 = R[3] * R[6] + R[4] * R[7] + R[5] * R[8];
  const T det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[3] * (R[1] * R[8] - R[2] * R[7]) + R[6] * (R[1] * R[5] - R[2] * R[4]);
  // np.allclose(R^T R, I, rtol=atol=1e-5): |a-b| <= atol + rtol*|b|
  const bool ok = fabs(g00 - T(1)) <= T(2) * tol && fabs(g11 - T(1)) <= T(2) * tol && fabs(g22 - T(1)) <= T(2) * tol &&
                  fabs(g01) <= tol && fabs(g02) <= tol && fabs(g12) <= tol && fabs(det - T(1)) <= T(1e-8) + tol;
  if (ok) return;
  double X[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) X[i] = (double)R[i];
  for (int it = 0; it < 30; ++it) {
    double C[9];
    C[0] = X[4] * X[8] - X[5] * X[7]; C[1] = X[5] * X[6] - X[3] * X[8]; C[2] = X[3] * X[7] - X[4] * X[6];
    C[3] = X[2] * X[7] - X[1] * X[8]; C[4] = X[0] * X[8] - X[2] * X[6]; C[5] = X[1] * X[6] - X[0] * X[7];
    C[6] = X[1] * X[5] - X[2] * X[4]; C[7] = X[2] * X[3] - X[0] * X[5]; C[8] = X[0] * X[4] - X[1] * X[3];
    // C = cof(X), column-major like X; X^-T = C / det X
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
    if (delta < 1e-15) break;
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = (T)X[i];
}

// ------------------------------------------------------------------------------------
// Dynamics (quad.py:321-335): z = (v[3], R[9] column-major, W[3]); x' = v is integrated
// from the stage velocities.  dz depends on (R, W) only.
// ------------------------------------------------------------------------------------
template <typename T>
struct Dyn {
  T c;           // f/m
  T A1, A2, A3;  // (J2-J3)/J1, (J3-J1)/J2, (J1-J2)/J3
  T U1, U2, U3;  // M_i / J_i
};

template <typename T>
__device__ __forceinline__ void rhs(const T* __restrict__ z, T* __restrict__ k, const Dyn<T>& p) {
  const T W1 = z[12], W2 = z[13], W3 = z[14];
  // v' = g e3 - (f/m) b3
  k[0] = -p.c * z[9];
  k[1] = -p.c * z[10];
  k[2] = T(kG) - p.c * z[11];
  // R' = R hat(W): b1' = W3 b2 - W2 b3 ; b2' = -W3 b1 + W1 b3 ; b3' = W2 b1 - W1 b2
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const T b1 = z[3 + i], b2 = z[6 + i], b3 = z[9 + i];
    k[3 + i] = W3 * b2 - W2 * b3;
    k[6 + i] = W1 * b3 - W3 * b1;
    k[9 + i] = W2 * b1 - W1 * b2;
  }
  // W' = J^-1 (-W x JW + M), J diagonal
  k[12] = p.A1 * W2 * W3 + p.U1;
  k[13] = p.A2 * W3 * W1 + p.U2;
  k[14] = p.A3 * W1 * W2 + p.U3;
}

template <typename T>
__device__ __forceinline__ void rk4_step(T (&y)[18], T h, const Dyn<T>& p) {
  T* z = &y[3];
  T k[15], acc[15], zt[15], xs[3];
  const T h2 = T(0.5) * h, h6 = h / T(6);
  rhs(z, k, p);
#pragma unroll
  for (int i = 0; i < 15; ++i) { acc[i] = k[i]; zt[i] = z[i] + h2 * k[i]; }
#pragma unroll
  for (int i = 0; i < 3; ++i) xs[i] = z[i];
  rhs(zt, k, p);
#pragma unroll
  for (int i = 0; i < 3; ++i) xs[i] += T(2) * zt[i];
#pragma unroll
  for (int i = 0; i < 15; ++i) { acc[i] += T(2) * k[i]; zt[i] = z[i] + h2 * k[i]; }
  rhs(zt, k, p);
#pragma unroll
  for (int i = 0; i < 3; ++i) xs[i] += T(2) * zt[i];
#pragma unroll
  for (int i = 0; i < 15; ++i) { acc[i] += T(2) * k[i]; zt[i] = z[i] + h * k[i]; }
  rhs(zt, k, p);
#pragma unroll
  for (int i = 0; i < 3; ++i) y[i] += h6 * (xs[i] + zt[i]);
#pragma unroll
  for (int i = 0; i < 15; ++i) z[i] += h6 * (acc[i] + k[i]);
}

// ------------------------------------------------------------------------------------
// LDS transposes between lane-per-env registers and AoS rows in global memory.
// The workgroup's rows [first, first+rows) x D floats are contiguous in global memory.
// ------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void store_rows(float* __restrict__ gbase, const float (&vals)[D], float* smem, int tid, int rows) {
#pragma unroll
  for (int j = 0; j < D; ++j) smem[tid * D + j] = vals[j];
  __syncthreads();
  const int total = rows * D;
  if (rows == kBlock && (reinterpret_cast<uintptr_t>(gbase) & 15u) == 0) {
    constexpr int nvec = kBlock * D / 4;  // kBlock is a multiple of 4
    const float4* s4 = reinterpret_cast<const float4*>(smem);
    float4* g4 = reinterpret_cast<float4*>(gbase);
#pragma unroll
    for (int idx = tid; idx < nvec; idx += kBlock) g4[idx] = s4[idx];
  } else {
    for (int idx = tid; idx < total; idx += kBlock) gbase[idx] = smem[idx];
  }
  __syncthreads();
}

template <int D>
__device__ __forceinline__ void load_rows(const float* __restrict__ gbase, float (&vals)[D], float* smem, int tid, int rows) {
  const int total = rows * D;
  if (rows == kBlock && (reinterpret_cast<uintptr_t>(gbase) & 15u) == 0) {
    constexpr int nvec = kBlock * D / 4;
    float4* s4 = reinterpret_cast<float4*>(smem);
    const float4* g4 = reinterpret_cast<const float4*>(gbase);
#pragma unroll
    for (int idx = tid; idx < nvec; idx += kBlock) s4[idx] = g4[idx];
  } else {
    for (int idx = tid; idx < total; idx += kBlock) smem[idx] = gbase[idx];
  }
  __syncthreads();
  if (tid < rows) {
#pragma unroll
    for (int j = 0; j < D; ++j) vals[j] = smem[tid * D + j];
  } else {
#pragma unroll
    for (int j = 0; j < D; ++j) vals[j] = 0.f;
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------
// Per-env working set
// ------------------------------------------------------------------------------------
template <typename T>
struct Work {
  T y[18];
  Phys<T> ph;
  T goal[12];   // xd, vd, b1d, Wd
  T integ[8];   // eIx, g_x prev, eIb1, g_b prev
};

template <int KIND> struct KindTraits;
template <> struct KindTraits<QR_KIND_QUAD>      { static constexpr int A = 4, D0 = 18, D1 = 0, NAG = 1; };
template <> struct KindTraits<QR_KIND_COUPLED>   { static constexpr int A = 4, D0 = 23, D1 = 0, NAG = 1; };
template <> struct KindTraits<QR_KIND_DECOUPLED> { static constexpr int A = 5, D0 = 15, D1 = 3, NAG = 2; };

// action_wrapper of the three kinds (quad.py:225-242, coupled:44-53, decoupled:49-59 + 68-73)
template <int KIND, typename T>
__device__ __forceinline__ void action_map(const float* a, const Work<T>& w, Dyn<T>& p) {
  const Phys<T>& ph = w.ph;
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
    } else {
      const T t1 = T(a[1]), t2 = T(a[2]), t3 = T(a[3]);
      const T* y = w.y;
      M1 = (y[6] * t1 + y[7] * t2 + y[8] * t3) + ph.J3 * y[17] * y[16];
      M2 = (y[9] * t1 + y[10] * t2 + y[11] * t3) - ph.J3 * y[17] * y[15];
      M3 = T(a[4]);
    }
  }
  const T J1 = ph.J1, J2 = ph.J1, J3 = ph.J3;  // J2 = J1 (quad.py:383)
  p.c = f / ph.m;
  p.A1 = (J2 - J3) / J1; p.A2 = (J3 - J1) / J2; p.A3 = (J1 - J2) / J3;
  p.U1 = M1 / J1; p.U2 = M2 / J2; p.U3 = M3 / J3;
}

// get_norm_error_state (quad.py:421-466): fills the float32 observation rows and advances
// the trapezoid integrators (quad_utils.py:38-63).
template <int KIND, typename T>
__device__ __forceinline__ void error_obs(Work<T>& w, const Coeffs& c, float (&o0)[KindTraits<KIND>::D0],
                                          float (&o1)[KindTraits<KIND>::D1 ? KindTraits<KIND>::D1 : 1]) {
  const T* y = w.y;
  const T xl = T(c.x_lim), vl = T(c.v_lim), Wl = T(c.W_lim);
  T ex[3], ev[3], eW[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    ex[j] = y[j] / xl - w.goal[j] / xl;
    ev[j] = y[3 + j] / vl - w.goal[3 + j] / vl;
    eW[j] = y[15 + j] / Wl - w.goal[9 + j] / Wl;
  }
  const T* b1 = &y[6]; const T* b2 = &y[9]; const T* b3 = &y[12];
  const T* b1d = &w.goal[6];
  const T db3 = b1d[0] * b3[0] + b1d[1] * b3[1] + b1d[2] * b3[2];
  T b1c[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) b1c[j] = b1d[j] - db3 * b3[j];
  const T sn = -(b1c[0] * b2[0] + b1c[1] * b2[1] + b1c[2] * b2[2]);
  const T cs = b1c[0] * b1[0] + b1c[1] * b1[1] + b1c[2] * b1[2];
  const T eb1 = T(atan2f((float)sn, (float)cs));  // [rad]
  const T eb1n = eb1 / T(kPi);
  // integrators: I += (g_prev + g) dt/2 ; g uses I before the update
  const T hdt = T(c.dt) / T(2);
  T eIxn[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const T g = -T(c.alpha) * w.integ[j] + ex[j] * xl;
    w.integ[j] += (w.integ[3 + j] + g) * hdt;
    w.integ[3 + j] = g;
    eIxn[j] = clampT(w.integ[j] / T(c.eIx_lim), T(-1), T(1));
  }
  const T gb = -T(c.beta) * w.integ[6] + eb1n * T(kPi);
  w.integ[6] += (w.integ[7] + gb) * hdt;
  w.integ[7] = gb;
  const T eIb1n = clampT(w.integ[6] / T(c.eIb1_lim), T(-1), T(1));
  if constexpr (KIND == QR_KIND_COUPLED) {
#pragma unroll
    for (int j = 0; j < 3; ++j) { o0[j] = (float)ex[j]; o0[3 + j] = (float)eIxn[j]; o0[6 + j] = (float)ev[j]; o0[20 + j] = (float)eW[j]; }
#pragma unroll
    for (int j = 0; j < 9; ++j) o0[9 + j] = (float)y[6 + j];
    o0[18] = (float)eb1n; o0[19] = (float)eIb1n;
  } else {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      o0[j] = (float)ex[j]; o0[3 + j] = (float)eIxn[j]; o0[6 + j] = (float)ev[j]; o0[9 + j] = (float)b3[j];
      o0[12 + j] = (float)(eW[0] * b1[j] + eW[1] * b2[j]);
    }
    o1[0] = (float)eb1n; o1[1] = (float)eIb1n; o1[2] = (float)eW[2];
  }
}

__device__ __forceinline__ float sq3(const float* v) { return v[0] * v[0] + v[1] * v[1] + v[2] * v[2]; }
__device__ __forceinline__ bool out3(const float* v) { return !(fabsf(v[0]) < 1.0f) || !(fabsf(v[1]) < 1.0f) || !(fabsf(v[2]) < 1.0f); }
__device__ __forceinline__ float interp01(float r, float rmin) { return clampT((r - rmin) / (-rmin), 0.0f, 1.0f); }

// ------------------------------------------------------------------------------------
// The fused step / rollout kernel
// ------------------------------------------------------------------------------------
template <int KIND, typename T>
__global__ __launch_bounds__(kBlock) void step_kernel(const Args a) {
  using KT = KindTraits<KIND>;
  constexpr int A = KT::A, D0 = KT::D0, D1 = KT::D1 ? KT::D1 : 1, NAG = KT::NAG;
  __shared__ __attribute__((aligned(16))) float smem[kBlock * (D0 > A ? D0 : A)];
  const int tid = threadIdx.x;
  const int64_t first = (int64_t)blockIdx.x * kBlock;
  const int64_t i = first + tid;
  const int64_t N = a.n;
  const int rows = (int)((N - first) < kBlock ? (N - first) : kBlock);
  const bool active = tid < rows;
  const Coeffs& c = a.c;

  Work<T> w;
  // ---- load the env's working set (SoA, lane-contiguous) ----
  const T* st = reinterpret_cast<const T*>(a.state);
  if (active) {
#pragma unroll
    for (int f = 0; f < 18; ++f) w.y[f] = st[(int64_t)f * N + i];
    if (a.params) {
      w.ph.m = T(a.params[i]); w.ph.d = T(a.params[N + i]); w.ph.J1 = T(a.params[2 * N + i]);
      w.ph.J3 = T(a.params[3 * N + i]); w.ph.ctf = T(a.params[4 * N + i]); w.ph.ctw = T(a.params[5 * N + i]);
    } else {
      w.ph.m = T(kMnom); w.ph.d = T(kDnom); w.ph.J1 = T(kJ1nom); w.ph.J3 = T(kJ3nom); w.ph.ctf = T(kCtfNom); w.ph.ctw = T(kCtwNom);
    }
    if (a.goal) {
#pragma unroll
      for (int f = 0; f < 12; ++f) w.goal[f] = T(a.goal[(int64_t)f * N + i]);
    } else {
#pragma unroll
      for (int f = 0; f < 12; ++f) w.goal[f] = T(f == 6 ? 1 : 0);
    }
    if (KIND != QR_KIND_QUAD) {
#pragma unroll
      for (int f = 0; f < 8; ++f) w.integ[f] = T(a.integ[(int64_t)f * N + i]);
    }
  } else {
#pragma unroll
    for (int f = 0; f < 18; ++f) w.y[f] = T((f == 6 || f == 10 || f == 14) ? 1 : 0);
    w.ph.m = T(kMnom); w.ph.d = T(kDnom); w.ph.J1 = T(kJ1nom); w.ph.J3 = T(kJ3nom); w.ph.ctf = T(kCtfNom); w.ph.ctw = T(kCtwNom);
#pragma unroll
    for (int f = 0; f < 12; ++f) w.goal[f] = T(f == 6 ? 1 : 0);
#pragma unroll
    for (int f = 0; f < 8; ++f) w.integ[f] = T(0);
  }
  w.ph.derive();
  int32_t steps = (a.steps && active) ? a.steps[i] : 0;
  int32_t episode = (a.episode && active) ? a.episode[i] : 0;
  bool params_dirty = false;

  for (int t = 0; t < a.n_steps; ++t) {
    // ---- action rows [N][A] -> lane registers ----
    float act[A];
    const float* abase = a.action + ((int64_t)t * N + first) * A;
    if constexpr (A == 4) {
      if (active) {
        const float4 v = reinterpret_cast<const float4*>(abase)[tid];
        act[0] = v.x; act[1] = v.y; act[2] = v.z; act[3] = v.w;
      } else {
        act[0] = act[1] = act[2] = act[3] = 0.f;
      }
    } else {
      load_rows<A>(abase, act, smem, tid, rows);
    }

    // ---- state_decomposition at step start: ensure_SO3 (quad_utils.py:12-16) ----
    so3_guard(&w.y[6]);

    // ---- action_wrapper ----
    Dyn<T> dyn;
    action_map<KIND, T>(act, w, dyn);

    // ---- observation_wrapper: integrate over dt with zero-order-hold (f, M) ----
    const T h = T(c.dt) / T(a.substeps);
    for (int s = 0; s < a.substeps; ++s) rk4_step(w.y, h, dyn);
    newton_schulz(&w.y[6]);

    // ---- obs / reward / done ----
    float o0[D0];
    float o1[D1];
    float rraw[NAG], rwd[NAG];
    bool dn[NAG];
    if constexpr (KIND == QR_KIND_QUAD) {
      const T* y = w.y;
      // reward_wrapper (quad.py:274-298)
      T eX2 = 0, eV2 = 0, W2 = 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const T dx = y[j] - w.goal[j], dv = y[3 + j] - w.goal[3 + j];
        eX2 += dx * dx; eV2 += dv * dv; W2 += y[15 + j] * y[15 + j];
      }
      // eb1 = signed angle from b1d to b1_proj = (R00, R10, 0)/hypot  (quad_utils.py:97-101,157-177)
      const T hy = sqrt(y[6] * y[6] + y[7] * y[7]);
      const T cux = hy > T(0) ? y[6] / hy : T(1), cuy = hy > T(0) ? y[7] / hy : T(0);
      const T dn_ = sqrt(w.goal[6] * w.goal[6] + w.goal[7] * w.goal[7] + w.goal[8] * w.goal[8]);
      const T dux = w.goal[6] / dn_, duy = w.goal[7] / dn_, duz = w.goal[8] / dn_;
      const T dot = dux * cux + duy * cuy;
      const T cz = dux * cuy - duy * cux;
      const T sabs = sqrt(duz * duz + cz * cz);  // |du x cu|
      float ang = atan2f((float)sabs, (float)dot);  // = acos(clip(dot)) for unit vectors
      if (cz < T(0)) ang = -ang;
      const T eb1 = T(ang) / T(kPi);
      const T r = -T(c.Cx) * eX2 - T(c.Cb1) * fabs(eb1) - T(c.Cv) * eV2 - T(c.CW) * W2;
      rraw[0] = (float)r;
      rwd[0] = (float)clampT((r - T(c.rmin_mono)) / (-T(c.rmin_mono)), T(0), T(1));
      // done_wrapper (quad.py:301-318): roll = atan2(R21,R22), pitch = -asin(R20)
      bool d = false;
#pragma unroll
      for (int j = 0; j < 3; ++j)
        d = d || !(fabs(y[j]) < T(c.x_lim)) || !(fabs(y[3 + j]) < T(c.v_lim)) || !(fabs(y[15 + j]) < T(c.W_lim));
      d = d || !(fabs(y[8]) < T(c.sin_euler_lim));               // |pitch| >= lim
      d = d || !(fabs(y[11]) < T(c.tan_euler_lim) * y[14]);      // |atan2(R21,R22)| >= lim
      dn[0] = d;
#pragma unroll
      for (int j = 0; j < 18; ++j) o0[j] = (float)y[j];
    } else {
      error_obs<KIND, T>(w, c, o0, o1);
      if constexpr (KIND == QR_KIND_COUPLED) {  // coupled:78-110, float32 arithmetic on the float32 obs
        const float r = -(float)c.Cx * sq3(&o0[0]) + -(float)c.CIx * sq3(&o0[3]) + -(float)c.Cv * sq3(&o0[6]) +
                        -(float)c.Cb1 * fabsf(o0[18]) + -(float)c.CIb1 * (o0[19] * o0[19]) + -(float)c.CW * sq3(&o0[20]);
        rraw[0] = r;
        rwd[0] = interp01(r, (float)c.rmin_mono);
        dn[0] = out3(&o0[0]) || out3(&o0[6]) || out3(&o0[20]);
      } else {  // decoupled:92-140
        const float r1 = -(float)c.Cx * sq3(&o0[0]) + -(float)c.CIx * sq3(&o0[3]) + -(float)c.Cv * sq3(&o0[6]) +
                         -(float)c.Cw12 * sq3(&o0[12]);
        const float r2 = -(float)c.Cb1 * fabsf(o1[0]) + -(float)c.CIb1 * (o1[1] * o1[1]) + -(float)c.CW3 * (o1[2] * o1[2]);
        rraw[0] = r1; rraw[1] = r2;
        rwd[0] = interp01(r1, (float)c.rmin_1); rwd[1] = interp01(r2, (float)c.rmin_2);
        dn[0] = out3(&o0[0]) || out3(&o0[6]) || out3(&o0[12]);
        dn[NAG - 1] = !(fabsf(o1[2]) < 1.0f);
      }
    }
    // crash override (quad.py:162-166)
#pragma unroll
    for (int g = 0; g < NAG; ++g)
      if (dn[g]) rwd[g] = -1.0f;

    // ---- time limit + auto-reset ----
    steps += 1;
    bool trunc = a.max_episode_steps > 0 && steps >= a.max_episode_steps;
    bool any_done = trunc;
#pragma unroll
    for (int g = 0; g < NAG; ++g) any_done = any_done || dn[g];
    if ((a.flags & QR_FLAG_AUTO_RESET) && any_done && active) {
      episode += 1;
      const bool eval = (a.flags & QR_FLAG_EVAL_RESET) != 0;
      const bool randomise = !eval && !(a.flags & QR_FLAG_NO_UDM) && a.params != nullptr;
      if (a.params != nullptr) {
        sample_reset(w.y, w.ph, randomise, eval, c, a.seed, (uint64_t)(a.env_offset + i), (uint32_t)episode);
        params_dirty = true;
      } else {
        Phys<T> keep = w.ph;
        sample_reset(w.y, w.ph, false, eval, c, a.seed, (uint64_t)(a.env_offset + i), (uint32_t)episode);
        w.ph = keep;
      }
      steps = 0;
      if constexpr (KIND == QR_KIND_QUAD) {
#pragma unroll
        for (int j = 0; j < 18; ++j) o0[j] = (float)w.y[j];
      } else {
#pragma unroll
        for (int f = 0; f < 8; ++f) w.integ[f] = T(0);
        error_obs<KIND, T>(w, c, o0, o1);  // first observation of the new episode (main.py:226-230)
      }
    }

    // ---- outputs of step t ----
    const int64_t row0 = (int64_t)t * N + first;
    if (KIND != QR_KIND_QUAD || a.obs0 != nullptr) store_rows<D0>(a.obs0 + row0 * D0, o0, smem, tid, rows);
    if constexpr (KT::D1 > 0) store_rows<D1>(a.obs1 + row0 * D1, o1, smem, tid, rows);
    if (active) {
      if constexpr (NAG == 1) {
        a.reward[row0 + tid] = rwd[0];
        if (a.reward_raw) a.reward_raw[row0 + tid] = rraw[0];
        a.done[row0 + tid] = dn[0] ? 1 : 0;
      } else {
        reinterpret_cast<float2*>(a.reward)[row0 + tid] = make_float2(rwd[0], rwd[1]);
        if (a.reward_raw) reinterpret_cast<float2*>(a.reward_raw)[row0 + tid] = make_float2(rraw[0], rraw[1]);
        reinterpret_cast<uchar2*>(a.done)[row0 + tid] = make_uchar2(dn[0] ? 1 : 0, dn[NAG - 1] ? 1 : 0);
      }
      if (a.truncated) a.truncated[row0 + tid] = trunc ? 1 : 0;
    }
  }

  // ---- write the working set back ----
  if (active) {
    T* sto = reinterpret_cast<T*>(a.state);
#pragma unroll
    for (int f = 0; f < 18; ++f) sto[(int64_t)f * N + i] = w.y[f];
    if (KIND != QR_KIND_QUAD) {
#pragma unroll
      for (int f = 0; f < 8; ++f) a.integ[(int64_t)f * N + i] = (float)w.integ[f];
    }
    if (a.steps) a.steps[i] = steps;
    if (a.flags & QR_FLAG_AUTO_RESET) {
      if (a.episode) a.episode[i] = episode;
      if (params_dirty) {
        a.params[i] = (float)w.ph.m; a.params[N + i] = (float)w.ph.d; a.params[2 * N + i] = (float)w.ph.J1;
        a.params[3 * N + i] = (float)w.ph.J3; a.params[4 * N + i] = (float)w.ph.ctf; a.params[5 * N + i] = (float)w.ph.ctw;
      }
    }
  }
}

// get_norm_error_state on the current state (quad.py:421-466)
template <int KIND, typename T>
__global__ __launch_bounds__(kBlock) void error_obs_kernel(const Args a) {
  using KT = KindTraits<KIND>;
  constexpr int D0 = KT::D0, D1 = KT::D1 ? KT::D1 : 1;
  __shared__ __attribute__((aligned(16))) float smem[kBlock * D0];
  const int tid = threadIdx.x;
  const int64_t first = (int64_t)blockIdx.x * kBlock;
  const int64_t i = first + tid;
  const int64_t N = a.n;
  const int rows = (int)((N - first) < kBlock ? (N - first) : kBlock);
  const bool active = tid < rows;
  Work<T> w;
  const T* st = reinterpret_cast<const T*>(a.state);
#pragma unroll
  for (int f = 0; f < 18; ++f) w.y[f] = active ? st[(int64_t)f * N + i] : T((f == 6 || f == 10 || f == 14) ? 1 : 0);
#pragma unroll
  for (int f = 0; f < 12; ++f) w.goal[f] = (active && a.goal) ? T(a.goal[(int64_t)f * N + i]) : T(f == 6 ? 1 : 0);
#pragma unroll
  for (int f = 0; f < 8; ++f) w.integ[f] = active ? T(a.integ[(int64_t)f * N + i]) : T(0);
  so3_guard(&w.y[6]);  // state_normalization -> ensure_SO3 (quad_utils.py:20-26)
  float o0[D0];
  float o1[D1];
  error_obs<KIND, T>(w, a.c, o0, o1);
  store_rows<D0>(a.obs0 + first * D0, o0, smem, tid, rows);
  if constexpr (KT::D1 > 0) store_rows<D1>(a.obs1 + first * D1, o1, smem, tid, rows);
  if (active) {
#pragma unroll
    for (int f = 0; f < 8; ++f) a.integ[(int64_t)f * N + i] = (float)w.integ[f];
  }
}

// QuadEnv.reset for masked envs
template <typename T>
__global__ __launch_bounds__(kBlock) void reset_kernel(const Args a) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t N = a.n;
  if (i >= N) return;
  if (a.mask && !a.mask[i]) return;
  const int32_t episode = a.episode[i] + 1;
  const bool eval = (a.flags & QR_FLAG_EVAL_RESET) != 0;
  const bool randomise = !eval && !(a.flags & QR_FLAG_NO_UDM);
  T y[18];
  Phys<T> ph;
  sample_reset(y, ph, randomise, eval, a.c, a.seed, (uint64_t)(a.env_offset + i), (uint32_t)episode);
  T* sto = reinterpret_cast<T*>(a.state);
#pragma unroll
  for (int f = 0; f < 18; ++f) sto[(int64_t)f * N + i] = y[f];
  if (a.params) {
    a.params[i] = (float)ph.m; a.params[N + i] = (float)ph.d; a.params[2 * N + i] = (float)ph.J1;
    a.params[3 * N + i] = (float)ph.J3; a.params[4 * N + i] = (float)ph.ctf; a.params[5 * N + i] = (float)ph.ctw;
  }
  if (a.integ) {
#pragma unroll
    for (int f = 0; f < 8; ++f) a.integ[(int64_t)f * N + i] = 0.f;
  }
  if (a.steps) a.steps[i] = 0;
  a.episode[i] = episode;
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
}

static int fill_env(Args& a, const QrEnv* e) {
  if (!e) return QR_E_NULL;
  if (e->kind < 0 || e->kind > 2) return QR_E_KIND;
  if (e->num_envs < 0) return QR_E_SIZE;
  if (!e->state) return QR_E_NULL;
  if (reinterpret_cast<uintptr_t>(e->state) & 15u) return QR_E_ALIGN;
  a.state = e->state; a.integ = e->integ; a.params = e->params; a.goal = e->goal;
  a.episode = e->episode; a.steps = e->steps;
  a.n = e->num_envs; a.env_offset = e->env_offset; a.seed = e->seed;
  a.max_episode_steps = e->max_episode_steps; a.flags = e->flags;
  fill_coeffs(a.c, e->coeffs);
  return 0;
}

template <typename T>
static int launch_step(const Args& a, int kind, hipStream_t s) {
  const unsigned grid = (unsigned)((a.n + kBlock - 1) / kBlock);
  if (grid == 0) return 0;
  switch (kind) {
    case QR_KIND_QUAD: hipLaunchKernelGGL((step_kernel<QR_KIND_QUAD, T>), dim3(grid), dim3(kBlock), 0, s, a); break;
    case QR_KIND_COUPLED: hipLaunchKernelGGL((step_kernel<QR_KIND_COUPLED, T>), dim3(grid), dim3(kBlock), 0, s, a); break;
    default: hipLaunchKernelGGL((step_kernel<QR_KIND_DECOUPLED, T>), dim3(grid), dim3(kBlock), 0, s, a); break;
  }
  return (int)hipGetLastError();
}

static int do_rollout(const QrEnv* env, const float* action, int32_t n_steps, int32_t substeps, const QrStepOut* out, void* stream) {
  Args a{};
  if (int rc = fill_env(a, env)) return rc;
  if (!action || !out || !out->reward || !out->done) return QR_E_NULL;
  if (substeps < 1 || n_steps < 1) return QR_E_SIZE;
  if (env->kind != QR_KIND_QUAD && (!env->integ || !out->obs0)) return QR_E_NULL;
  if (env->kind == QR_KIND_DECOUPLED && !out->obs1) return QR_E_NULL;
  if ((env->flags & QR_FLAG_AUTO_RESET) && !env->episode) return QR_E_NULL;
  if (reinterpret_cast<uintptr_t>(action) & 15u) return QR_E_ALIGN;
  a.action = action; a.obs0 = out->obs0; a.obs1 = out->obs1;
  a.reward = out->reward; a.reward_raw = out->reward_raw; a.done = out->done; a.truncated = out->truncated;
  a.n_steps = n_steps; a.substeps = substeps;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return env->state_f64 ? launch_step<double>(a, env->kind, s) : launch_step<float>(a, env->kind, s);
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
}

int qr_step(const QrEnv* env, const float* action, int32_t substeps, const QrStepOut* out, void* stream) {
  return qr::do_rollout(env, action, 1, substeps, out, stream);
}

int qr_rollout(const QrEnv* env, const float* action, int32_t n_steps, int32_t substeps, const QrStepOut* out, void* stream) {
  return qr::do_rollout(env, action, n_steps, substeps, out, stream);
}

int qr_error_obs(const QrEnv* env, float* obs0, float* obs1, void* stream) {
  qr::Args a{};
  if (int rc = qr::fill_env(a, env)) return rc;
  if (env->kind == QR_KIND_QUAD) return QR_E_KIND;
  if (!env->integ || !obs0 || (env->kind == QR_KIND_DECOUPLED && !obs1)) return QR_E_NULL;
  a.obs0 = obs0; a.obs1 = obs1;
  const unsigned grid = (unsigned)((a.n + qr::kBlock - 1) / qr::kBlock);
  if (grid == 0) return 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (env->state_f64) {
    if (env->kind == QR_KIND_COUPLED) hipLaunchKernelGGL((qr::error_obs_kernel<QR_KIND_COUPLED, double>), dim3(grid), dim3(qr::kBlock), 0, s, a);
    else hipLaunchKernelGGL((qr::error_obs_kernel<QR_KIND_DECOUPLED, double>), dim3(grid), dim3(qr::kBlock), 0, s, a);
  } else {
    if (env->kind == QR_KIND_COUPLED) hipLaunchKernelGGL((qr::error_obs_kernel<QR_KIND_COUPLED, float>), dim3(grid), dim3(qr::kBlock), 0, s, a);
    else hipLaunchKernelGGL((qr::error_obs_kernel<QR_KIND_DECOUPLED, float>), dim3(grid), dim3(qr::kBlock), 0, s, a);
  }
  return (int)hipGetLastError();
}

int qr_reset(const QrEnv* env, const uint8_t* mask, void* stream) {
  qr::Args a{};
  if (int rc = qr::fill_env(a, env)) return rc;
  if (!env->episode) return QR_E_NULL;
  a.mask = mask;
  const unsigned grid = (unsigned)((a.n + qr::kBlock - 1) / qr::kBlock);
  if (grid == 0) return 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (env->state_f64) hipLaunchKernelGGL((qr::reset_kernel<double>), dim3(grid), dim3(qr::kBlock), 0, s, a);
  else hipLaunchKernelGGL((qr::reset_kernel<float>), dim3(grid), dim3(qr::kBlock), 0, s, a);
  return (int)hipGetLastError();
}

const char* qr_step_kernel_info(int32_t kind, int32_t state_f64, int64_t num_envs, int32_t* grid, int32_t* block) {
  if (grid) *grid = (int32_t)((num_envs + qr::kBlock - 1) / qr::kBlock);
  if (block) *block = qr::kBlock;
  (void)state_f64;
  switch (kind) {
    case QR_KIND_QUAD: return "qr::step_kernel<0>";
    case QR_KIND_COUPLED: return "qr::step_kernel<1>";
    case QR_KIND_DECOUPLED: return "qr::step_kernel<2>";
    default: return "";
  }
}

}  // extern "C"
