// qr_dynamics.h — part of the gfx950 quadrotor step library (included by quadrotor_kernels.hip, in this order).
// Attitude helpers, quaternion-form dynamics + RK4, LDS row transposes, action maps, error observations.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "quadrotor_hip.h"
#include "qr_rng.h"

namespace qr {

// ------------------------------------------------------------------------------------
// Attitude helpers
// ------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T fmaT(T a, T b, T c) {
  if constexpr (std::is_same<T, double>::value) return fma(a, b, c); else return fmaf(a, b, c);
}
template <typename T> __device__ __forceinline__ T fma_ss(T a, T b, T c, T d) {  // a b + c d
  if constexpr (std::is_same<T, double>::value) return fma(a, b, c * d); else return fmaf(a, b, c * d);
}
template <typename T> __device__ __forceinline__ T fma_sd(T a, T b, T c, T d) {  // a b - c d
  if constexpr (std::is_same<T, double>::value) return fma(a, b, -(c * d)); else return fmaf(a, b, -(c * d));
}
template <typename T> __device__ __forceinline__ T fma_1m2(T s) {  // 1 - 2 s
  if constexpr (std::is_same<T, double>::value) return fma(-2.0, s, 1.0); else return fmaf(-2.0f, s, 1.0f);
}
// R(q), column-major like the reference's vec_F(R): R[3c + r].
template <typename T>
__device__ __forceinline__ void quat_to_R(const T* q, T (&R)[9]) {
  const T w = q[0], x = q[1], y = q[2], z = q[3];
  const T two = T(2);
  // 1 - 2 (a^2 + b^2) and 2 (a b +- c d) as explicit fma chains (see renorm_quat)
  R[0] = fma_1m2(fma_ss(y, y, z, z)); R[1] = two * fma_ss(x, y, w, z);  R[2] = two * fma_sd(x, z, w, y);
  R[3] = two * fma_sd(x, y, w, z);    R[4] = fma_1m2(fma_ss(x, x, z, z)); R[5] = two * fma_ss(y, z, w, x);
  R[6] = two * fma_ss(x, z, w, y);    R[7] = two * fma_sd(y, z, w, x);  R[8] = fma_1m2(fma_ss(x, x, y, y));
}

// ensure_SO3 (quad_utils.py:123-142) + attitude import.  The reference replaces R by the
// nearest rotation U V^T (SVD) when R^T R or det R is off by more than 1e-5; since the
// internal attitude is a unit quaternion, the nearest rotation is taken always (for an R
// that is orthonormal to round-off this changes nothing).  The polar factor is computed by
// the Newton iteration X <- (X + X^-T)/2, which converges quadratically to U V^T (det R > 0).
// Returns false — q untouched — when R has no nearest rotation: det R <= 0 (the closest orthogonal
// matrix is a reflection) or a non-finite entry.
__device__ bool R_to_quat(const double* Rin, double (&q)[4]) {
  double X[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) X[i] = Rin[i];
  {
    const double det = X[0] * (X[4] * X[8] - X[5] * X[7]) + X[1] * (X[5] * X[6] - X[3] * X[8]) + X[2] * (X[3] * X[7] - X[4] * X[6]);
    double amax = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) amax = fmax(amax, fabs(X[i]));
    if (!(det > 1e-300) || !(amax < 1e150)) return false;  // also catches NaN / Inf
  }
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
  return true;
}

// ------------------------------------------------------------------------------------
// Dynamics (quad.py:321-335) in quaternion form.
// ------------------------------------------------------------------------------------
template <typename T>
struct Dyn {
  T c;           // f/m
  T A1;          // (J2-J3)/J1 with J2 = J1; the W2' coefficient (J3-J1)/J2 is -A1
  T U1, U2, U3;  // M_i / J_i
  T g;
};

// y = (v[0..2], q[3..6] = w,x,y,z, W[7..9])
template <typename T>
__device__ __forceinline__ void rhs(const T* __restrict__ y, T* __restrict__ k, const Dyn<T>& p) {
  const T qw = y[3], qx = y[4], qy = y[5], qz = y[6];
  const T W1 = y[7], W2 = y[8], W3 = y[9];
  // v' = g e3 - (f/m) R e3,  R e3 = (2(xz + wy), 2(yz - wx), 1 - 2(xx + yy))     (explicit fma chains: see renorm_quat)
  const T c2 = T(2) * p.c;
  k[0] = -c2 * fma_ss(qx, qz, qw, qy);
  k[1] = -c2 * fma_sd(qy, qz, qw, qx);
  k[2] = fmaT(c2, fma_ss(qx, qx, qy, qy), p.g - p.c);
  // q' = q (0, W) / 2   (<=> R' = R hat(W))
  const T h = T(0.5);
  k[3] = -h * fmaT(qx, W1, fmaT(qy, W2, qz * W3));
  k[4] = h * fmaT(qw, W1, fmaT(qy, W3, -(qz * W2)));
  k[5] = h * fmaT(qw, W2, fmaT(qz, W1, -(qx * W3)));
  k[6] = h * fmaT(qw, W3, fmaT(qx, W2, -(qy * W1)));
  // W' = J^-1 (-W x JW + M), J = diag(J1, J1, J3): the (J1 - J2) W1 W2 term of W3' vanishes
  const T a3 = p.A1 * W3;
  k[7] = fmaT(a3, W2, p.U1);
  k[8] = fmaT(-a3, W1, p.U2);
  k[9] = p.U3;
}

template <typename T>
__device__ __forceinline__ void rk4_step(T (&x)[3], T (&y)[10], T h, const Dyn<T>& p) {
  T k[10], acc[10], yt[10], xs[3];
  const T h2 = T(0.5) * h, h6 = h * T(1.0 / 6.0);
  rhs(y, k, p);
#pragma unroll
  for (int i = 0; i < 10; ++i) { acc[i] = k[i]; yt[i] = fmaT(h2, k[i], y[i]); }
#pragma unroll
  for (int i = 0; i < 3; ++i) xs[i] = y[i];
  rhs(yt, k, p);
#pragma unroll
  for (int i = 0; i < 3; ++i) xs[i] = fmaT(T(2), yt[i], xs[i]);
#pragma unroll
  for (int i = 0; i < 10; ++i) { acc[i] = fmaT(T(2), k[i], acc[i]); yt[i] = fmaT(h2, k[i], y[i]); }
  rhs(yt, k, p);
#pragma unroll
  for (int i = 0; i < 3; ++i) xs[i] = fmaT(T(2), yt[i], xs[i]);
#pragma unroll
  for (int i = 0; i < 10; ++i) { acc[i] = fmaT(T(2), k[i], acc[i]); yt[i] = fmaT(h, k[i], y[i]); }
  rhs(yt, k, p);
#pragma unroll
  for (int i = 0; i < 3; ++i) x[i] = fmaT(h6, xs[i] + yt[i], x[i]);
#pragma unroll
  for (int i = 0; i < 10; ++i) y[i] = fmaT(h6, acc[i] + k[i], y[i]);
}

// (explicit fma chains: with -ffp-contract=fast the compiler is otherwise free to contract a sum of products
// differently in every instantiation of the kernel, and instantiations must agree to the bit)
template <typename T>
__device__ __forceinline__ void renorm_quat(T* q) {
  const T n2 = fmaT(q[0], q[0], fmaT(q[1], q[1], fmaT(q[2], q[2], q[3] * q[3])));
  const T r = fmaT(T(-0.5), n2, T(1.5));
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] *= r;
}

// Uniform-precision layouts (f64: the reference-grade mode, 4th-order convergence to the
// reference's DOP853 down to 1e-10; f32: the approximate mode): plain RK4 in T.
template <typename T>
__device__ __forceinline__ void integrate(T (&x)[3], T (&v)[3], T (&q)[4], T (&W)[3], const Dyn<T>& p, int nsub, T h) {
  T y[10];
#pragma unroll
  for (int j = 0; j < 3; ++j) { y[j] = v[j]; y[7 + j] = W[j]; }
#pragma unroll
  for (int j = 0; j < 4; ++j) y[3 + j] = q[j];
  for (int s = 0; s < nsub; ++s) rk4_step(x, y, h, p);
#pragma unroll
  for (int j = 0; j < 3; ++j) { v[j] = y[j]; W[j] = y[7 + j]; }
#pragma unroll
  for (int j = 0; j < 4; ++j) q[j] = y[3 + j];
}

// ------------------------------------------------------------------------------------
// The default (`mixed`) layout: x, v float32; q, W float64.  Which arithmetic needs float64?
//   * W: its increments are large (|W'| up to ~120 rad/s^2 from the torques) and every error in W is integrated once more into q.
//     W3' = M3/J3 is constant over the step (zero-order-hold torque, J1 = J2), so W3(t) is linear in time, drops out of the RK4
//     vector and turns the W1-W2 system into a linear one with a known coefficient a(t) = A1 W3(t):
//         W1' = a(t) W2 + U1,   W2' = -a(t) W1 + U2.
//     The TORQUE part of the increment (dt U, the large one) is taken in float64, exactly.  The gyroscopic COUPLING part — RK4 of
//     a(t) (W2, -W1), at most ~22 rad/s^2 in regime — comes out of the float32 stage chain the quaternion needs anyway and is
//     accumulated like q (round 6; rounds at ~7e-7 rad/s^2 per stage, i.e. ~3e-9 rad/s per step).  Rounds 2-5 ran the whole
//     W1-W2 RK4 in float64: 24 float64 instructions per substep instead of none (profiles/r06/ab_w_coupling.txt: 1 M envs x 10
//     substeps 60.3 -> 53.2 us, 131 072 x 10 10.1 -> 8.7, one substep unchanged; production-mode error 1.42-1.56e-6 either way).
//     The launches WITHOUT in-launch resets, whose envs can leave the regime (|W| to 30 rad/s), keep float64 for W: integrate_delta.
//   * q: only its ACCUMULATION.  The increment dq = h/6 (k1 + 2 k2 + 2 k3 + k4) is ~|W| h / 2 <= 0.1,
//     so forming the stage quaternions and derivatives in float32 (from the float32-rounded
//     step-start q and float32 copies of the stage rates) perturbs q by ~1e-9 per step,
//     pseudo-randomly; the float64 state absorbs the step's increment exactly.  Within ONE env-step the substeps' increments
//     are summed in float32 (<= 16 terms of <= 0.016: ~4e-9 per step) and the float64 state takes the sum once.
//   * x, v are float32 in memory already; their increments are quadratures of the thrust
//     direction R(q) e3 over the stages, linear in it, so the stage sums are simply accumulated
//     over all substeps (float32) and applied once.
// Measured against the float64 DOP853 oracle (tools/numerics_f32stage.py: 1000 free-run steps,
// |W| up to 26 rad/s): R 1.3e-6 (all-float64 RK4: 1.0e-6, truncation), x, v 4e-7; with 10
// substeps R 9e-7 — the bar is 1e-5.  On gfx950 an f32 VALU instruction issues at up to twice
// the f64 rate once two waves share a SIMD, and the float32 stage code needs half the registers.
// ------------------------------------------------------------------------------------
// (Measured and NOT adopted, profiles/r02/ab_quad_builds.txt column q_pk: the quaternion stages on v_pk_fma_f32 / v_pk_mul_f32 with
// op_sel / neg swizzles, six packed instructions per derivative — 4.45-4.56 against 4.40-4.47 us per launch at 65 536 envs,
// 33.7-33.8 against 33.0-33.2 at 1 M: the code is gone, the record stays.)
__device__ __forceinline__ void integrate(float (&x)[3], float (&v)[3], double (&q)[4], double (&W)[3], const Dyn<double>& p, int nsub,
                                          double h) {
  const float hf = (float)h, h2f = 0.5f * hf, h6f = hf * (1.0f / 6.0f);
  const double h2 = 0.5 * h;
  // half body rates (q' = q (0, W/2)); a(t) = A1 W3(t) and W3(t)/2 advance by a constant per half substep
  float a3f = (float)(p.A1 * W[2]), w3 = 0.5f * (float)W[2];
  const float daf = (float)(p.A1 * p.U3 * h2), dw3 = (float)(0.5 * p.U3 * h2);
  const float u1 = (float)(0.5 * p.U1), u2 = (float)(0.5 * p.U2);
  double W1 = W[0], W2 = W[1];
  // running float32 sums of the substeps' increments (see the loop's end): dq; the coupling part of d(W/2); and the torque part of
  // d(W/2) per substep, for the float32 track of the stage rates
  float dqs[4] = {0.f, 0.f, 0.f, 0.f}, csa = 0.f, csb = 0.f;
  const float hu1h = (float)(0.5 * h * p.U1), hu2h = (float)(0.5 * h * p.U2);
  float g1[3] = {0.f, 0.f, 0.f}, g23[3] = {0.f, 0.f, 0.f}, g4[3] = {0.f, 0.f, 0.f}, xx[3] = {0.f, 0.f, 0.f};
  // thrust direction, un-normalised: R e3 = (2 u0, 2 u1, 1 - 2 u2), u = (xz + wy, yz - wx, xx + yy)
#define QR_THRUST(G, Qw, Qx, Qy, Qz)                            \
  G[0] = fmaf(Qx, Qz, fmaf(Qw, Qy, G[0]));                      \
  G[1] = fmaf(Qy, Qz, fmaf(-Qw, Qx, G[1]));                     \
  G[2] = fmaf(Qx, Qx, fmaf(Qy, Qy, G[2]));
  float qs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) qs[j] = (float)q[j];
  float w1 = 0.5f * (float)W[0], w2 = 0.5f * (float)W[1];
#define QR_QDOT(K, Q, A, B, C)                                  \
  K[0] = -fmaf(Q[1], A, fmaf(Q[2], B, Q[3] * C));               \
  K[1] = fmaf(Q[0], A, fmaf(Q[2], C, -Q[3] * B));               \
  K[2] = fmaf(Q[0], B, fmaf(Q[3], A, -Q[1] * C));               \
  K[3] = fmaf(Q[0], C, fmaf(Q[1], B, -Q[2] * A));
  for (int s = 0; s < nsub; ++s) {
    if (s > 0) {  // prefix sums of the substeps' thrust sums: the double integral for x
#pragma unroll
      for (int j = 0; j < 3; ++j) xx[j] += fmaf(2.0f, g23[j], g1[j] + g4[j]);
    }
    // ---- stage rates in float32 (half units); c = the gyroscopic coupling part of W' (k = c + u), summed with RK4's weights ----
    const float b0 = a3f, bm = a3f + daf, b1 = bm + daf;
    const float z0 = w3, zm = w3 + dw3, z1 = zm + dw3;
    float kq[4], acc[4], qt[4];
    // stage 1
    QR_QDOT(kq, qs, w1, w2, z0)
    QR_THRUST(g1, qs[0], qs[1], qs[2], qs[3])
    float ca = b0 * w2, cb = -b0 * w1;            // coupling parts (half units), k = c + u
    float sca = ca, scb = cb;
    float t1 = fmaf(h2f, ca + u1, w1), t2 = fmaf(h2f, cb + u2, w2);
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[j] = kq[j]; qt[j] = fmaf(h2f, kq[j], qs[j]); }
    // stage 2
    QR_QDOT(kq, qt, t1, t2, zm)
    QR_THRUST(g23, qt[0], qt[1], qt[2], qt[3])
    ca = bm * t2; cb = -bm * t1;
    sca = fmaf(2.0f, ca, sca); scb = fmaf(2.0f, cb, scb);
    t1 = fmaf(h2f, ca + u1, w1); t2 = fmaf(h2f, cb + u2, w2);
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[j] = fmaf(2.0f, kq[j], acc[j]); qt[j] = fmaf(h2f, kq[j], qs[j]); }
    // stage 3
    QR_QDOT(kq, qt, t1, t2, zm)
    QR_THRUST(g23, qt[0], qt[1], qt[2], qt[3])
    ca = bm * t2; cb = -bm * t1;
    sca = fmaf(2.0f, ca, sca); scb = fmaf(2.0f, cb, scb);
    t1 = fmaf(hf, ca + u1, w1); t2 = fmaf(hf, cb + u2, w2);
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[j] = fmaf(2.0f, kq[j], acc[j]); qt[j] = fmaf(hf, kq[j], qs[j]); }
    // stage 4
    QR_QDOT(kq, qt, t1, t2, z1)
    QR_THRUST(g4, qt[0], qt[1], qt[2], qt[3])
    sca = fmaf(b1, t2, sca); scb = fmaf(-b1, t1, scb);
    // Within ONE env-step every stage quantity is float32 already; so are, here, the running sums of the substeps' increments —
    // the float64 state takes them once, at the end of the step (below).  A sum of <= 16 increments of <= 0.016 (q) / 0.06 (W/2)
    // rounds at ~4e-9 per step, below the float32 stage noise that is there anyway; with ONE substep the result is the same.
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float dq = h6f * (acc[j] + kq[j]);
      dqs[j] += dq;
      qs[j] += dq;          // float32 track for the next substep's stages (re-synchronised every env-step)
    }
    const float ia = h6f * sca, ib = h6f * scb;   // coupling increments (half units)
    csa += ia; csb += ib;
    w1 += hu1h + ia; w2 += hu2h + ib;             // float32 track of W/2 (re-synchronised every env-step)
    a3f = b1; w3 = z1;
  }
#undef QR_THRUST
#undef QR_QDOT
  // v' = g e3 - c R e3 integrated over all substeps: sum_n h/6 (k1 + 2 k2 + 2 k3 + k4)_n, and x' = v:
  //   v_end = v0 + dt (0, 0, g - c) + (h c / 3) s (G1 + 2 G23 + G4),        s = (-1, -1, +1)
  //   x_end = x0 + dt v0 + dt^2/2 (0, 0, g - c) + (h^2 c / 3) s (XX + G1 + G23)
  // (g - c formed in float64: near hover the two cancel, and rounding each to float32 first would bias the
  // vertical acceleration by ~5e-7 m/s^2 for a whole flight)
  const float cf = (float)p.c, dtf = hf * (float)nsub, gc = (float)(p.g - p.c);
  const float hc3 = hf * cf * (1.0f / 3.0f), hhc3 = hf * hc3;
  float G[3], X2[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) { G[j] = fmaf(2.0f, g23[j], g1[j] + g4[j]); X2[j] = xx[j] + (g1[j] + g23[j]); }
  x[0] = fmaf(-hhc3, X2[0], fmaf(dtf, v[0], x[0]));
  x[1] = fmaf(-hhc3, X2[1], fmaf(dtf, v[1], x[1]));
  x[2] = fmaf(hhc3, X2[2], fmaf(0.5f * dtf * dtf, gc, fmaf(dtf, v[2], x[2])));
  v[0] = fmaf(-hc3, G[0], v[0]);
  v[1] = fmaf(-hc3, G[1], v[1]);
  v[2] = fmaf(hc3, G[2], fmaf(dtf, gc, v[2]));
  {  // the float64 state takes the step's increments once
    const double dt = h * (double)nsub;
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] += (double)dqs[j];
    W1 = fma(dt, p.U1, W1) + 2.0 * (double)csa;   // torque part exact; coupling part from the float32 chain (half units -> x 2)
    W2 = fma(dt, p.U2, W2) + 2.0 * (double)csb;
  }
  W[0] = W1; W[1] = W2;
  W[2] = fma(p.U3, h * (double)nsub, W[2]);
}

// ------------------------------------------------------------------------------------
// TWO OR MORE substeps per env-step in the default layout (instantiations MAG = true, picked by the host from `substeps`; round 6,
// second pass): a 4th-order Lie-group (Magnus) substep instead of RK4's four stages — 74 float32 VALU instructions per substep
// instead of 149 (tools/numerics_magnus.py is the NumPy emulation against the DOP853 oracle).  What makes it cheap is what the
// zero-order hold already gave RK4: W3(t) = W3 + U3 t is exact, and w = W1 + i W2 obeys the LINEAR equation w' = -i a(t) w + u with
// a(t) = A1 W3(t), so
//   * w(t) over the whole env-step is a Taylor polynomial about the step's start, (k + 1) c_{k+1} = -i (a0 c_k + a' c_{k-1})
//     (+ u for k = 0), formed ONCE per env-step.  |a| dt <= 0.02 in regime (0.09 at |W3| = 30 rad/s), so degree 4 leaves
//     (|a| dt)^5 / 120 |w| <= 2e-10 rad/s (emulated with degree 5: no difference; degree 3 is visibly short);
//   * a substep [t, t + h] needs no stages: with Wm, Wm', Wm'' at its midpoint (the polynomial, then the equation itself)
//         Theta = h Wm + h^3/24 Wm'' + h^3/12 (Wm x Wm')        (Magnus terms 1 and 2 about the midpoint: local error O(h^5))
//         q <- q (x) exp(Theta / 2),   taken as the INCREMENT q (x) (cos - 1, sin ...) so that float32 only rounds terms <= 0.016;
//   * the thrust direction u(q) at the substep BOUNDARIES only: the integrals for v and x by the trapezoid rule plus the
//     Euler-Maclaurin end correction h^2/12 (g'(0) - g'(dt)) — the interior derivative terms telescope, so ONE u per substep and
//     u' = (R (W x e3))-terms at the two ends of the env-step (error O(dt h^4), any substep count).
// Precision bookkeeping as in RK4 above: every per-substep quantity float32, the substeps' increments of q summed in float32, the
// float64 state updated once per env-step, the torque part of dW (dt U) exact in float64 and only the coupling part
// C(dt) = w(dt) - w0 - u dt from the float32 polynomial.  Emulated against DOP853, x and v kept float64 between steps (the
// integrator's own error, in regime): 1.3e-7 / 7.9e-9 at 1 / 2 substeps against RK4's 2.2e-7 / 1.4e-8; with the float32 storage both
// sit at 1.4-1.5e-6.
// Why ONE substep keeps RK4 although this is 100 instructions shorter there too: its instructions depend on one another.  A lone
// wave pays 3.6-3.9 ns for a vector instruction that waits for the one before it and 2.0-2.3 ns for an independent one
// (tools/valu_ilp_microbench.hip, profiles/r06/valu_ilp_microbench.json); RK4's four quaternion components are four chains side by
// side, the Magnus substep is one.  Measured with this code on every substep count: the 65 536-env step +2.1 %, the fused rollout
// +9.3 % (one stepping wave per SIMD: latency binds), 1 M x 10 substeps -22 %, 131 072 x 10 -17 % (issue binds): profiles/r06/ab_magnus.txt.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void integrate_magnus(float (&x)[3], float (&v)[3], double (&q)[4], double (&W)[3], const Dyn<double>& p,
                                                 int nsub, double h) {
  const float hf = (float)h, hh = 0.5f * hf;     // Theta / 2 directly: half units
  const float h3 = hf * hf * hf, kdd = h3 * (1.0f / 48.0f), kcr = h3 * (1.0f / 24.0f);
  const float a0 = (float)(p.A1 * W[2]), ad = (float)(p.A1 * p.U3);
  const float U1 = (float)p.U1, U2 = (float)p.U2, U3 = (float)p.U3, w3_0 = (float)W[2];
  // Taylor coefficients of w(t); (-i)(zr + i zi) = zi - i zr.  k1 = the coupling part of c1 (c1 = k1 + u).
  const float c0r = (float)W[0], c0i = (float)W[1];
  const float k1r = a0 * c0i, k1i = -(a0 * c0r);
  const float c1r = fmaf(a0, c0i, U1), c1i = fmaf(-a0, c0r, U2);
  float zr = fmaf(a0, c1r, ad * c0r), zi = fmaf(a0, c1i, ad * c0i);
  const float c2r = 0.5f * zi, c2i = -0.5f * zr;
  zr = fmaf(a0, c2r, ad * c1r); zi = fmaf(a0, c2i, ad * c1i);
  const float c3r = (1.0f / 3.0f) * zi, c3i = (-1.0f / 3.0f) * zr;
  zr = fmaf(a0, c3r, ad * c2r); zi = fmaf(a0, c3i, ad * c2i);
  const float c4r = 0.25f * zi, c4i = -0.25f * zr;
  float qs[4], dqs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) qs[j] = (float)q[j];
  // Thrust direction, un-normalised: R e3 = (2 u0, 2 u1, 1 - 2 u2), u = (xz + wy, yz - wx, xx + yy); its rate along the flow,
  // R (W x e3) = W2 b1 - W1 b2 (b1, b2: columns of R), in the same convention: ud = (b3'_0 / 2, b3'_1 / 2, -b3'_2 / 2).
#define QR_UVEC_ACC(Gv)                                          \
  Gv[0] = fmaf(qs[1], qs[3], fmaf(qs[0], qs[2], Gv[0]));         \
  Gv[1] = fmaf(qs[2], qs[3], fmaf(-qs[0], qs[1], Gv[1]));        \
  Gv[2] = fmaf(qs[1], qs[1], fmaf(qs[2], qs[2], Gv[2]));
#define QR_UDOT(UD, HW1, HW2) /* HW = W / 2 */                                                  \
  {                                                                                             \
    const float xy = qs[1] * qs[2];                                                             \
    const float n1 = fmaf(-2.0f, fmaf(qs[2], qs[2], qs[3] * qs[3]), 1.0f);   /* b1_0 */         \
    const float n2_ = fmaf(-2.0f, fmaf(qs[1], qs[1], qs[3] * qs[3]), 1.0f);  /* b2_1 */         \
    UD[0] = fmaf(HW2, n1, -2.0f * (HW1 * fmaf(-qs[0], qs[3], xy)));                             \
    UD[1] = fmaf(-HW1, n2_, 2.0f * (HW2 * fmaf(qs[0], qs[3], xy)));                             \
    UD[2] = 2.0f * fmaf(HW1, fmaf(qs[2], qs[3], qs[0] * qs[1]), -(HW2 * fmaf(qs[1], qs[3], -(qs[0] * qs[2])))); \
  }
  float u0[3] = {0.f, 0.f, 0.f}, ud0[3], G[3], XX[3] = {0.f, 0.f, 0.f};
  QR_UVEC_ACC(u0)
  QR_UDOT(ud0, 0.5f * c0r, 0.5f * c0i)
#pragma unroll
  for (int j = 0; j < 3; ++j) G[j] = u0[j];
  float tm = hh;                                  // the substep's midpoint
  for (int s = 0; s < nsub; ++s) {
    const float W1m = fmaf(tm, fmaf(tm, fmaf(tm, fmaf(tm, c4r, c3r), c2r), c1r), c0r);
    const float W2m = fmaf(tm, fmaf(tm, fmaf(tm, fmaf(tm, c4i, c3i), c2i), c1i), c0i);
    const float W3m = fmaf(U3, tm, w3_0), am = fmaf(ad, tm, a0);
    const float d1 = fmaf(am, W2m, U1), d2 = fmaf(-am, W1m, U2);             // Wm'
    const float e1 = fmaf(ad, W2m, am * d2), e2 = fmaf(ad, W1m, am * d1);    // Wm'' = (e1, -e2, 0)
    const float cx = fmaf(W2m, U3, -(W3m * d2)), cy = fmaf(W3m, d1, -(W1m * U3)), cz = fmaf(W1m, d2, -(W2m * d1));
    const float t1 = fmaf(kcr, cx, fmaf(kdd, e1, hh * W1m));
    const float t2 = fmaf(kcr, cy, fmaf(-kdd, e2, hh * W2m));
    const float t3 = fmaf(kcr, cz, hh * W3m);
    const float n2 = fmaf(t1, t1, fmaf(t2, t2, t3 * t3));
    const float sn = fmaf(n2, fmaf(n2, 1.0f / 120.0f, -1.0f / 6.0f), 1.0f);  // sin|t| / |t|
    const float cm = n2 * fmaf(n2, 1.0f / 24.0f, -0.5f);                     // cos|t| - 1
    const float A = sn * t1, B = sn * t2, C = sn * t3;
    float dq[4];
    dq[0] = fmaf(qs[0], cm, -fmaf(qs[1], A, fmaf(qs[2], B, qs[3] * C)));
    dq[1] = fmaf(qs[1], cm, fmaf(qs[0], A, fmaf(qs[2], C, -(qs[3] * B))));
    dq[2] = fmaf(qs[2], cm, fmaf(qs[0], B, fmaf(qs[3], A, -(qs[1] * C))));
    dq[3] = fmaf(qs[3], cm, fmaf(qs[0], C, fmaf(qs[1], B, -(qs[2] * A))));
#pragma unroll
    for (int j = 0; j < 4; ++j) { dqs[j] += dq[j]; qs[j] += dq[j]; }
#pragma unroll
    for (int j = 0; j < 3; ++j) XX[j] += G[j];    // prefix sums: sum_{k < n} G_k = sum_k (n - k) u_k, the double integral for x
    QR_UVEC_ACC(G)
    tm += hf;
  }
  // W1, W2 at the step's end (float32: for u' there), and the coupling part alone (for the float64 state)
  const float dtf = hf * (float)nsub;
  const float Pr = fmaf(dtf, fmaf(dtf, c4r, c3r), c2r), Pi = fmaf(dtf, fmaf(dtf, c4i, c3i), c2i);
  const float Cr = dtf * fmaf(dtf, Pr, k1r), Ci = dtf * fmaf(dtf, Pi, k1i);
  const float W1e = fmaf(dtf, fmaf(dtf, Pr, c1r), c0r), W2e = fmaf(dtf, fmaf(dtf, Pi, c1i), c0i);
  float un[3], udn[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) un[j] = 0.f;
  QR_UVEC_ACC(un)
  QR_UDOT(udn, 0.5f * W1e, 0.5f * W2e)
#undef QR_UVEC_ACC
#undef QR_UDOT
  // With Iv = int u dt = h (G - (u0 + un) / 2) + h^2/12 (u0' - un') and Ix = int (dt - t) u dt = h^2 (XX - n u0 / 2) + h^2/12 (un - u0 + dt u0'):
  //   v_end = v0 + dt (0, 0, g - c) + 2 c s Iv,   x_end = x0 + dt v0 + dt^2/2 (0, 0, g - c) + 2 c s Ix,   s = (-1, -1, +1)
  // (g - c formed in float64: near hover the two cancel, and rounding each to float32 first would bias the vertical
  // acceleration by ~5e-7 m/s^2 for a whole flight)
  const float cf = (float)p.c, gc = (float)(p.g - p.c);
  const float c2h = 2.0f * cf * hf, c2hh = c2h * hf, e12 = c2h * hf * (1.0f / 12.0f), nh = -0.5f * (float)nsub;
  float Iv[3], Ix[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    Iv[j] = fmaf(e12, ud0[j] - udn[j], c2h * fmaf(-0.5f, u0[j] + un[j], G[j]));
    Ix[j] = fmaf(e12, fmaf(dtf, ud0[j], un[j] - u0[j]), c2hh * fmaf(nh, u0[j], XX[j]));
  }
  x[0] = fmaf(dtf, v[0], x[0]) - Ix[0];
  x[1] = fmaf(dtf, v[1], x[1]) - Ix[1];
  x[2] = fmaf(0.5f * dtf * dtf, gc, fmaf(dtf, v[2], x[2])) + Ix[2];
  v[0] -= Iv[0];
  v[1] -= Iv[1];
  v[2] = fmaf(dtf, gc, v[2]) + Iv[2];
  {  // the float64 state takes the step's increments once: dq; W = W0 + dt U (exact) + the coupling part of the polynomial at dt
    const double dt = h * (double)nsub;
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] += (double)dqs[j];
    W[0] = fma(dt, p.U1, W[0]) + (double)Cr;
    W[1] = fma(dt, p.U2, W[1]) + (double)Ci;
    W[2] = fma(p.U3, dt, W[2]);
  }
}

// The integrator of an instantiation: MAG = true exists for the default layout only (the uniform layouts are plain RK4 in T).
template <bool MAG, typename XV, typename QW>
__device__ __forceinline__ void integrate_sel(XV (&x)[3], XV (&v)[3], QW (&q)[4], QW (&W)[3], const Dyn<QW>& p, int nsub, QW h) {
  if constexpr (MAG && std::is_same<XV, float>::value && std::is_same<QW, double>::value) integrate_magnus(x, v, q, W, p, nsub, h);
  else integrate(x, v, q, W, p, nsub, h);
}

// ------------------------------------------------------------------------------------
// The same step with the quaternion stages in DELTA FORM — the arithmetic of the launches that step envs on WITHOUT
// in-launch resets (the rate-adaptive instantiations, ADAPT), where a free run far beyond termination amplifies the
// float32 stage noise of `integrate` (~1e-9 per step) through the Decoupled action map's feedback ~100x over 900 steps
// (6.8e-6 after 1000 steps, against 1.0e-6 for all-float64 stages):
//     dq = h k1 + h/6 (2 (k2 - k1) + 2 (k3 - k1) + (k4 - k1)),      k1 = q (0, W/2) in FLOAT64,
//     k_i - k1 = qdot(q, w_i - w_1) + qdot(qt_i - q, w_i)           in float32 (qdot is bilinear),
// i.e. what is large (k1 ~ |W|/2) is exact, and float32 only ever rounds quantities that are O(h |W|^2): the stage
// noise drops ~10x and the free run lands on the truncation floor of RK4 itself (tools/numerics_delta.py: Decoupled,
// 256 envs x 1000 steps: 7.5e-6 -> 1.3e-6 = all-float64 stages; applied only beyond termination it changes nothing —
// the perturbations that the feedback amplifies are planted IN regime).  +12 f64 and ~+40 f32 instructions per
// substep: not in the launches with in-launch resets, whose envs never leave the regime (1.0-1.5e-6 there).
// Same W chain, same thrust sums, same final x / v update as `integrate`.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void integrate_delta(float (&x)[3], float (&v)[3], double (&q)[4], double (&W)[3], const Dyn<double>& p, int nsub,
                                                double h) {
  const float hf = (float)h, h2f = 0.5f * hf, h6f = hf * (1.0f / 6.0f);
  const double h2 = 0.5 * h, h6 = h * (1.0 / 6.0);
  double a3 = p.A1 * W[2];
  const double da = p.A1 * p.U3 * h2;
  double W3h = 0.5 * W[2];                       // W3 / 2 at the substep's start, float64 (k1)
  const double dW3h = 0.5 * p.U3 * h;           // its advance per substep
  const float dw3 = (float)(0.5 * p.U3 * h2);
  const float u1 = (float)(0.5 * p.U1), u2 = (float)(0.5 * p.U2);
  double W1 = W[0], W2 = W[1];
  float g1[3] = {0.f, 0.f, 0.f}, g23[3] = {0.f, 0.f, 0.f}, g4[3] = {0.f, 0.f, 0.f}, xx[3] = {0.f, 0.f, 0.f};
#define QR_THRUST(G, Qw, Qx, Qy, Qz)                            \
  G[0] = fmaf(Qx, Qz, fmaf(Qw, Qy, G[0]));                      \
  G[1] = fmaf(Qy, Qz, fmaf(-Qw, Qx, G[1]));                     \
  G[2] = fmaf(Qx, Qx, fmaf(Qy, Qy, G[2]));
#define QR_QDOT(K, Q, A, B, C)                                  \
  K[0] = -fmaf(Q[1], A, fmaf(Q[2], B, Q[3] * C));               \
  K[1] = fmaf(Q[0], A, fmaf(Q[2], C, -Q[3] * B));               \
  K[2] = fmaf(Q[0], B, fmaf(Q[3], A, -Q[1] * C));               \
  K[3] = fmaf(Q[0], C, fmaf(Q[1], B, -Q[2] * A));
#define QR_QDOT_ACC(K, Q, A, B, C)                              \
  K[0] = fmaf(-Q[1], A, fmaf(-Q[2], B, fmaf(-Q[3], C, K[0])));  \
  K[1] = fmaf(Q[0], A, fmaf(Q[2], C, fmaf(-Q[3], B, K[1])));    \
  K[2] = fmaf(Q[0], B, fmaf(Q[3], A, fmaf(-Q[1], C, K[2])));    \
  K[3] = fmaf(Q[0], C, fmaf(Q[1], B, fmaf(-Q[2], A, K[3])));
  for (int s = 0; s < nsub; ++s) {
    if (s > 0) {
#pragma unroll
      for (int j = 0; j < 3; ++j) xx[j] += fmaf(2.0f, g23[j], g1[j] + g4[j]);
    }
    // ---- k1 = q (0, W/2) in float64, and the float32 copies the stages start from ----
    float qs[4], k1f[4];
    {
      const double hw1 = 0.5 * W1, hw2 = 0.5 * W2;
      const double k0 = -fma(q[1], hw1, fma(q[2], hw2, q[3] * W3h));
      const double k1 = fma(q[0], hw1, fma(q[2], W3h, -(q[3] * hw2)));
      const double k2 = fma(q[0], hw2, fma(q[3], hw1, -(q[1] * W3h)));
      const double k3 = fma(q[0], W3h, fma(q[1], hw2, -(q[2] * hw1)));
#pragma unroll
      for (int j = 0; j < 4; ++j) qs[j] = (float)q[j];
      k1f[0] = (float)k0; k1f[1] = (float)k1; k1f[2] = (float)k2; k1f[3] = (float)k3;
      q[0] = fma(h, k0, q[0]); q[1] = fma(h, k1, q[1]); q[2] = fma(h, k2, q[2]); q[3] = fma(h, k3, q[3]);  // h k1: exact part of the increment
    }
    const float w1 = (float)(0.5 * W1), w2 = (float)(0.5 * W2), w3 = (float)W3h;
    const float a3f = (float)a3, daf = (float)da;
    // ---- W1, W2 in float64 (running sums: nothing of the chain stays live) ----
    {
      const double am = a3 + da, a1 = am + da;
      double ka = fma(a3, W2, p.U1), kb = fma(-a3, W1, p.U2);
      double sa = ka, sb = kb;
      double s1 = fma(h2, ka, W1), s2 = fma(h2, kb, W2);
      ka = fma(am, s2, p.U1); kb = fma(-am, s1, p.U2);
      sa = fma(2.0, ka, sa); sb = fma(2.0, kb, sb);
      s1 = fma(h2, ka, W1); s2 = fma(h2, kb, W2);
      ka = fma(am, s2, p.U1); kb = fma(-am, s1, p.U2);
      sa = fma(2.0, ka, sa); sb = fma(2.0, kb, sb);
      s1 = fma(h, ka, W1); s2 = fma(h, kb, W2);
      ka = fma(a1, s2, p.U1); kb = fma(-a1, s1, p.U2);
      W1 = fma(h6, sa + ka, W1); W2 = fma(h6, sb + kb, W2);
      a3 = a1; W3h += dW3h;
    }
    // ---- float32 stage rates (half units) and their DIFFERENCES from the start rate ----
    const float bm = a3f + daf, b1 = bm + daf;
    const float zm = w3 + dw3, z1 = zm + dw3;
    float dq_[4], d[4], acc[4], qt[4];
    // stage 1: thrust direction at the substep's start
    QR_THRUST(g1, qs[0], qs[1], qs[2], qs[3])
    float ka = fmaf(a3f, w2, u1), kb = fmaf(-a3f, w1, u2);
    float dA = h2f * ka, dB = h2f * kb;                 // (w_2 - w_1): exactly the increment the float32 rate track takes
    float t1 = w1 + dA, t2 = w2 + dB;
#pragma unroll
    for (int j = 0; j < 4; ++j) { dq_[j] = h2f * k1f[j]; qt[j] = qs[j] + dq_[j]; }
    // stage 2
    QR_THRUST(g23, qt[0], qt[1], qt[2], qt[3])
    QR_QDOT(d, qs, dA, dB, dw3)
    QR_QDOT_ACC(d, dq_, t1, t2, zm)
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[j] = 2.0f * d[j]; dq_[j] = h2f * (k1f[j] + d[j]); qt[j] = qs[j] + dq_[j]; }
    ka = fmaf(bm, t2, u1); kb = fmaf(-bm, t1, u2);
    dA = h2f * ka; dB = h2f * kb;
    t1 = w1 + dA; t2 = w2 + dB;
    // stage 3
    QR_THRUST(g23, qt[0], qt[1], qt[2], qt[3])
    QR_QDOT(d, qs, dA, dB, dw3)
    QR_QDOT_ACC(d, dq_, t1, t2, zm)
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[j] = fmaf(2.0f, d[j], acc[j]); dq_[j] = hf * (k1f[j] + d[j]); qt[j] = qs[j] + dq_[j]; }
    ka = fmaf(bm, t2, u1); kb = fmaf(-bm, t1, u2);
    dA = hf * ka; dB = hf * kb;
    t1 = w1 + dA; t2 = w2 + dB;
    // stage 4
    QR_THRUST(g4, qt[0], qt[1], qt[2], qt[3])
    QR_QDOT(d, qs, dA, dB, 2.0f * dw3)
    QR_QDOT_ACC(d, dq_, t1, t2, z1)
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] += (double)(h6f * (acc[j] + d[j]));   // + h/6 (2 d2 + 2 d3 + d4): the small part, exactly
    (void)b1;
  }
#undef QR_THRUST
#undef QR_QDOT
#undef QR_QDOT_ACC
  const float cf = (float)p.c, dtf = hf * (float)nsub, gc = (float)(p.g - p.c);
  const float hc3 = hf * cf * (1.0f / 3.0f), hhc3 = hf * hc3;
  float G[3], X2[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) { G[j] = fmaf(2.0f, g23[j], g1[j] + g4[j]); X2[j] = xx[j] + (g1[j] + g23[j]); }
  x[0] = fmaf(-hhc3, X2[0], fmaf(dtf, v[0], x[0]));
  x[1] = fmaf(-hhc3, X2[1], fmaf(dtf, v[1], x[1]));
  x[2] = fmaf(hhc3, X2[2], fmaf(0.5f * dtf * dtf, gc, fmaf(dtf, v[2], x[2])));
  v[0] = fmaf(-hc3, G[0], v[0]);
  v[1] = fmaf(-hc3, G[1], v[1]);
  v[2] = fmaf(hc3, G[2], fmaf(dtf, gc, v[2]));
  W[0] = W1; W[1] = W2;
  W[2] = fma(p.U3, h * (double)nsub, W[2]);
}
// (uniform layouts: plain RK4 in T already — nothing to refine)
template <typename T>
__device__ __forceinline__ void integrate_delta(T (&x)[3], T (&v)[3], T (&q)[4], T (&W)[3], const Dyn<T>& p, int nsub, T h) {
  integrate(x, v, q, W, p, nsub, h);
}

// ------------------------------------------------------------------------------------
// LDS transposes between lane-per-env registers and AoS rows in global memory.
// The workgroup's rows [first, first+rows) x D floats are contiguous in global memory.
// ------------------------------------------------------------------------------------
// The tile is one wavefront (B == 64): its LDS accesses execute in program order, so the exchange needs a
// compiler-level ordering only — and must not be a workgroup barrier in the instantiation whose workgroup
// carries a second (helper) wave that takes no part in it.
template <int B>
__device__ __forceinline__ void tile_sync() {
  if constexpr (B == 64) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

// Half one: this thread's row into the LDS tile.  Half two: the tile, as it lies, to global memory with 16-byte stores.
// (The two halves may run in different wavefronts of a workgroup, with a barrier between them.)
template <int D>
__device__ __forceinline__ void rows_to_lds(const float (&vals)[D], float* smem, int tid) {
#pragma unroll
  for (int j = 0; j < D; ++j) smem[tid * D + j] = vals[j];
}

template <int B, int D, int AUX = 0>
__device__ __forceinline__ void lds_to_rows(float* __restrict__ gbase, const float* smem, int tid, int rows) {
  if (rows == B && (reinterpret_cast<uintptr_t>(gbase) & 15u) == 0) {
    constexpr int nvec = B * D / 4;  // B is a multiple of 4
    const float4* s4 = reinterpret_cast<const float4*>(smem);
    float4* g4 = reinterpret_cast<float4*>(gbase);
#pragma unroll
    for (int idx = tid; idx < nvec; idx += B) gstore<AUX>(g4 + idx, s4[idx]);
  } else {
    const int total = rows * D;
    for (int idx = tid; idx < total; idx += B) gstore<AUX>(gbase + idx, smem[idx]);
  }
}

template <int B, int D, int AUX = 0>
__device__ __forceinline__ void store_rows(float* __restrict__ gbase, const float (&vals)[D], float* smem, int tid, int rows) {
  rows_to_lds<D>(vals, smem, tid);
  tile_sync<B>();
  lds_to_rows<B, D, AUX>(gbase, smem, tid, rows);
  tile_sync<B>();
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
  tile_sync<B>();
#pragma unroll
  for (int j = 0; j < D; ++j) vals[j] = tid < rows ? smem[tid * D + j] : 0.f;
  tile_sync<B>();
}

template <int KIND> struct KindTraits;
template <> struct KindTraits<QR_KIND_QUAD>      { static constexpr int A = 4, D0 = 18, D1 = 0, NAG = 1; };
template <> struct KindTraits<QR_KIND_COUPLED>   { static constexpr int A = 4, D0 = 23, D1 = 0, NAG = 1; };
template <> struct KindTraits<QR_KIND_DECOUPLED> { static constexpr int A = 5, D0 = 15, D1 = 3, NAG = 2; };

// What the action map needs of the env's PARAMETERS only (quad.py:389-404 and the reciprocals): constant between two resets of an
// env, so a rollout forms it once per episode instead of once per env-step (~35 instructions, a v_rcp_f64 among them, off the
// stepping wave's path per step).  The same expressions on the same float32 parameter words as a one-step launch: the same bits.
template <typename T>
struct ActConsts {
  T min_force, max_force, avrg_act, scale_act, d, ctf, J3;
  T cm;        // 1 / m
  T iJ1, iJ3;  // 1 / J1, 1 / J3
  T A1;        // (J1 - J3) / J1
};

template <typename T, typename X>
__device__ __forceinline__ void act_consts(const Work<T, X>& w, const Coeffs& c, ActConsts<T>& k) {
  const Phys<T> ph(w, c);
  k.min_force = ph.min_force; k.max_force = ph.max_force; k.avrg_act = ph.avrg_act; k.scale_act = ph.scale_act;
  k.d = ph.d; k.ctf = ph.ctf; k.J3 = ph.J3;
  // 1/m, 1/J1, 1/J3 from ONE reciprocal (of their product): a v_rcp_f64 + Newton steps is ~9 VALU slots
  const T mJ1 = ph.m * ph.J1, r = recip(mJ1 * ph.J3);
  const T rJ3 = r * ph.J3;
  k.iJ3 = r * mJ1; k.iJ1 = rJ3 * ph.m;
  k.cm = rJ3 * ph.J1;
  k.A1 = (ph.J1 - ph.J3) * k.iJ1;
}

// action_wrapper of the three kinds (quad.py:225-242, coupled:44-53, decoupled:49-59 + 68-73)
template <int KIND, typename T, typename X>
__device__ __forceinline__ void action_map(const float* a, const Work<T, X>& w, const ActConsts<T>& k, const Coeffs& c, Dyn<T>& p) {
  T f, M1, M2, M3;
  if constexpr (KIND == QR_KIND_QUAD) {
    T t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = clampT(k.scale_act * T(a[j]) + k.avrg_act, k.min_force, k.max_force);
    f = ((t[0] + t[1]) + t[2]) + t[3];
    M1 = k.d * (t[3] - t[1]);
    M2 = k.d * (t[0] - t[2]);
    M3 = k.ctf * ((t[1] - t[0]) + (t[3] - t[2]));
  } else {
    f = clampT(T(4) * (k.scale_act * T(a[0]) + k.avrg_act), T(4) * k.min_force, T(4) * k.max_force);
    if constexpr (KIND == QR_KIND_COUPLED) {
      M1 = T(a[1]); M2 = T(a[2]); M3 = T(a[3]);
    } else {  // M1 = b1.tau + J3 W3 W2, M2 = b2.tau - J3 W3 W1 from (R, W) at step start
      const T t1 = T(a[1]), t2 = T(a[2]), t3 = T(a[3]);
      const T qw = w.q[0], qx = w.q[1], qy = w.q[2], qz = w.q[3];  // b1, b2 = first two columns of R(q)
      const T two = T(2);
      const T b1x = fma_1m2(fma_ss(qy, qy, qz, qz)), b1y = two * fma_ss(qx, qy, qw, qz), b1z = two * fma_sd(qx, qz, qw, qy);
      const T b2x = two * fma_sd(qx, qy, qw, qz), b2y = fma_1m2(fma_ss(qx, qx, qz, qz)), b2z = two * fma_ss(qy, qz, qw, qx);
      const T b1t = fmaT(b1x, t1, fmaT(b1y, t2, b1z * t3));
      const T b2t = fmaT(b2x, t1, fmaT(b2y, t2, b2z * t3));
      const T j3w3 = k.J3 * w.W[2];
      M1 = fmaT(j3w3, w.W[1], b1t);
      M2 = fmaT(-j3w3, w.W[0], b2t);
      M3 = T(a[4]);
    }
  }
  p.c = f * k.cm;
  p.A1 = k.A1;
  p.U1 = M1 * k.iJ1; p.U2 = M2 * k.iJ1; p.U3 = M3 * k.iJ3;
  p.g = T(c.g);
}

// get_norm_error_state (quad.py:421-466): fills the float32 observation rows and advances
// the trapezoid integrators (quad_utils.py:38-63).  The normalised errors are formed in T
// (float64 in the default layout) and rounded once, like the reference's `.astype(float32)`: the
// wrappers' terminations compare these float32 numbers with 1.
template <int KIND, typename T, typename X>
__device__ __forceinline__ void error_obs(Work<T, X>& w, const T (&R)[9], const Coeffs& c, float (&o0)[KindTraits<KIND>::D0],
                                          float (&o1)[KindTraits<KIND>::D1 ? KindTraits<KIND>::D1 : 1]) {
  const T ixl = T(c.inv_x_lim), ivl = T(c.inv_v_lim), iWl = T(c.inv_W_lim);
  T ex[3], ev[3], eW[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {  // x/x_lim - xd/x_lim etc. (quad.py:423-434)
    ex[j] = fmaT(T(w.x[j]), ixl, -(T(w.goal[j]) * ixl));   // (explicit fma chains: see renorm_quat)
    ev[j] = fmaT(T(w.v[j]), ivl, -(T(w.goal[3 + j]) * ivl));
    eW[j] = fmaT(w.W[j], iWl, -(T(w.goal[9 + j]) * iWl));
  }
  const T* b1 = &R[0]; const T* b2 = &R[3]; const T* b3 = &R[6];
  const T b1d[3] = {T(w.goal[6]), T(w.goal[7]), T(w.goal[8])};
  const T db3 = fmaT(b1d[0], b3[0], fmaT(b1d[1], b3[1], b1d[2] * b3[2]));
  T b1c[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) b1c[j] = fmaT(-db3, b3[j], b1d[j]);
  const T sn = -fmaT(b1c[0], b2[0], fmaT(b1c[1], b2[1], b1c[2] * b2[2]));
  const T cs = fmaT(b1c[0], b1[0], fmaT(b1c[1], b1[1], b1c[2] * b1[2]));
  const float eb1 = atan2_fast((float)sn, (float)cs);  // [rad]
  const float eb1n = eb1 * (float)(1.0 / kPi);
  // integrators: I += (g_prev + g) dt/2 ; g uses I before the update.  They are float32 words
  // (stored and held), advanced in float32.
  const float hdt = c.hdt;
  float eIxn[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float exf = (float)ex[j];
    o0[j] = exf;
    const float g = fmaf(-c.alpha, w.integ[j], exf * c.x_lim_f);
    w.integ[j] = fmaf(w.integ[3 + j] + g, hdt, w.integ[j]);
    w.integ[3 + j] = g;
    eIxn[j] = clampT(w.integ[j] * c.inv_eIx_lim, -1.0f, 1.0f);
  }
  const float gb = fmaf(-c.beta, w.integ[6], eb1);
  w.integ[6] = fmaf(w.integ[7] + gb, hdt, w.integ[6]);
  w.integ[7] = gb;
  const float eIb1n = clampT(w.integ[6] * c.inv_eIb1_lim, -1.0f, 1.0f);
  if constexpr (KIND == QR_KIND_COUPLED) {
#pragma unroll
    for (int j = 0; j < 3; ++j) { o0[3 + j] = eIxn[j]; o0[6 + j] = (float)ev[j]; o0[20 + j] = (float)eW[j]; }
#pragma unroll
    for (int j = 0; j < 9; ++j) o0[9 + j] = (float)R[j];
    o0[18] = eb1n; o0[19] = eIb1n;
  } else {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      o0[3 + j] = eIxn[j]; o0[6 + j] = (float)ev[j]; o0[9 + j] = (float)b3[j];
      o0[12 + j] = (float)fma_ss(eW[0], b1[j], eW[1], b2[j]);
    }
    o1[0] = eb1n; o1[1] = eIb1n; o1[2] = (float)eW[2];
  }
}

}  // namespace qr
