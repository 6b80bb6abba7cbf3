// qr_rng.h — part of the gfx950 quadrotor step library (included by quadrotor_kernels.hip, in this order).
// Philox4x32-10, the wave-cooperative draw for in-step resets, small sincos, reset sampling.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "quadrotor_hip.h"
#include "qr_args.h"

namespace qr {

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
// atan2 for float arguments, ~1.5e-7 rad: octant reduction (cephes atanf: [0, tan(pi/8)] by
// (a - 1)/(a + 1)), quotients by v_rcp_f32 + one Newton step, degree-4 polynomial in a^2.
// Branch-free and ~30 instructions (the OCML atan2f carries an IEEE float division and special-case
// branches; with one wave per SIMD every instruction of the slowest wave is exposed).
__device__ __forceinline__ float rcp_nr(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  return r * (2.0f - x * r);
}
__device__ __forceinline__ float atan2_fast(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  float a = (mx > 0.0f) ? mn * rcp_nr(mx) : 0.0f;           // in [0, 1]
  const bool hi = a > 0.41421356237f;                        // tan(pi/8)
  const float z = hi ? (a - 1.0f) * rcp_nr(a + 1.0f) : a;    // |z| <= tan(pi/8)
  const float z2 = z * z;
  float p = fmaf(fmaf(fmaf(fmaf(8.05374449538e-2f, z2, -1.38776856032e-1f), z2, 1.99777106478e-1f), z2, -3.33329491539e-1f) * z2, z, z);
  p = hi ? p + 0.78539816339744831f : p;                     // atan(mn / mx)
  p = (ay > ax) ? 1.57079632679489662f - p : p;              // first quadrant
  p = (x < 0.0f) ? 3.14159265358979324f - p : p;
  return copysignf(p, y);
}

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

}  // namespace qr
