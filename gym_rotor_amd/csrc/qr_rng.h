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
// quad.py:199)  <=>  q = qz(yaw) qy(pitch) qx(roll).  Every sampled value is a float32 number
// (W is returned as such); q has unit norm to float64 round-off.
template <typename T, typename X>
__device__ __forceinline__ void sample_start(const Draws& d, bool randomise, bool eval, const Coeffs& c, X (&x)[3], X (&v)[3],
                                             T (&q)[4], float (&Wf)[3], float (&prm)[6]) {
  if (randomise) {  // float32 values: that is how the params buffer stores them
    const float p = c.udm;
#pragma unroll
    for (int j = 0; j < 5; ++j) prm[j] = c.nom_f[j] * fmaf(p, d.sym(j), 1.0f);
    prm[5] = c.nom_f[5] * fmaf(0.5f * p, d.sym(5), 1.0f);
  } else {
#pragma unroll
    for (int j = 0; j < 6; ++j) prm[j] = c.nom_f[j];
  }
  const float yaw = (float)kPi * d.sym(6);
  float ix, iv, iR, iW;
  if (eval) {  // quad.py:352-356
    ix = 0.4f; iv = 0.0f; iR = 0.0f; iW = 0.0f;
  } else if (d.u01(7) < 0.2f) {  // quad.py:342-346
    ix = 0.0f; iv = 0.0f; iR = 0.0f; iW = 0.0f;
  } else {  // quad.py:348-351
    ix = 0.6f; iv = c.reset_v; iR = (float)(50.0 * kPi / 180.0); iW = c.reset_W;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    x[j] = X(ix * d.sym(8 + j));
    v[j] = X(iv * d.sym(11 + j));
    Wf[j] = iW * d.sym(14 + j);
  }
  double sr, cr, sp, cp, sy, cy;
  unit_sincos(0.5f * iR * d.sym(17), sr, cr);
  unit_sincos(0.5f * iR * d.sym(18), sp, cp);
  unit_sincos(0.5f * yaw, sy, cy);
  q[0] = T(cr * cp * cy + sr * sp * sy);
  q[1] = T(sr * cp * cy - cr * sp * sy);
  q[2] = T(cr * sp * cy + sr * cp * sy);
  q[3] = T(cr * cp * sy - sr * sp * cy);
}

template <typename T, typename X>
__device__ __forceinline__ void sample_reset(Work<T, X>& w, const Draws& d, bool randomise, bool eval, const Coeffs& c) {
  float Wf[3];
  sample_start<T, X>(d, randomise, eval, c, w.x, w.v, w.q, Wf, w.prm);
#pragma unroll
  for (int j = 0; j < 3; ++j) w.W[j] = T(Wf[j]);
  w.nominal = !randomise;
}

// ------------------------------------------------------------------------------------
// In-launch auto-reset: a POOL of freshly sampled episode starts per wavefront.
//
// Only ~1 % of the envs reset in a given step, but ~50 % of the waves contain one and a launch
// ends with its slowest wave, so the reset's instructions sit on the critical path of the whole
// launch.  Nothing about WHICH lane resets is known before the step has been integrated — but
// what a resetting lane needs (Philox draws + the sampling arithmetic) does not depend on the
// lane at all if the stream is keyed by the wave instead of by the env:
//     draws(slot s) = Philox4x32-10(key = seed; ctr = (global id of the wave's first env [64 bit],
//                                   reset counter of this wave's tile, 0x40000000 | s << 8 | block))
// One cooperative Philox pass (lane 5k+b computes block b of slot k; 12 slots) plus the sampling
// of the 12 starts (lane 5k) is therefore issued right after the wave's state loads and runs
// while they are in flight, when the SIMD has nothing else to do.  A lane that resets takes the
// slot given by its rank among the wave's resetting lanes (ds_bpermute from lane 5 * rank);
// ranks >= 12 (e.g. a time limit ending all 64 episodes at once) draw further pools on demand
// (slots 12 p + k).  The tile's counter advances by one per env-step, so no (tile, counter, slot)
// is ever used twice — also under hipGraph replay, because the counter lives in device memory.
// Results are independent of how the batch is sharded as long as shards start at multiples of 64.
// ------------------------------------------------------------------------------------
template <typename T, typename X>
struct ResetPool {  // meaningful in lanes 5k, k = 0..11
  X x[3], v[3];
  T q[4];
  float W[3];
  float prm[6];
  uint32_t r19;  // the word the goal generator's episode-start draws are taken from
};

template <typename T, typename X>
__device__ __forceinline__ void make_pool(ResetPool<T, X>& p, uint64_t seed, uint64_t gfirst, uint32_t count, int pass, bool randomise,
                                          bool eval, const Coeffs& c) {
  const int lane = (int)__lane_id();
  const int k = lane / 5, b = lane - 5 * k;  // slot / block of this lane (k = 12: lanes 60..63 idle)
  uint32_t ctr[4] = {(uint32_t)gfirst, (uint32_t)(gfirst >> 32), count, 0x40000000u | ((uint32_t)(12 * pass + k) << 8) | (uint32_t)b};
  philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
  // lane 5k gathers blocks 1..4 of its slot from lanes 5k+1 .. 5k+4 (all reads issued, one wait)
  Draws d;
  int got[16];
#pragma unroll
  for (int bb = 1; bb < 5; ++bb) {
#pragma unroll
    for (int j = 0; j < 4; ++j) got[4 * (bb - 1) + j] = __builtin_amdgcn_ds_bpermute((lane + bb) << 2, (int)ctr[j]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) d.r[j] = ctr[j];
#pragma unroll
  for (int j = 0; j < 16; ++j) d.r[4 + j] = (uint32_t)got[j];
  sample_start<T, X>(d, randomise, eval, c, p.x, p.v, p.q, p.W, p.prm);
  p.r19 = d.r[19];
}

__device__ __forceinline__ float bperm(int addr4, float v) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr4, __builtin_bit_cast(int, v))); }
__device__ __forceinline__ double bperm(int addr4, double v) {
  const uint64_t u = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(addr4, (int)(uint32_t)u);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(addr4, (int)(uint32_t)(u >> 32));
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// Lanes with take == true copy slot `slot` of the pool into their working set.  Executed by the whole wave.
template <typename T, typename X, bool TRAJ>
__device__ __forceinline__ void take_from_pool(const ResetPool<T, X>& p, bool take, int slot, Work<T, X>& w, uint32_t& r19) {
  const int addr = (take ? 5 * slot : 0) << 2;
  X x[3], v[3]; T q[4]; float W[3], prm[6];
#pragma unroll
  for (int j = 0; j < 3; ++j) { x[j] = bperm(addr, p.x[j]); v[j] = bperm(addr, p.v[j]); W[j] = bperm(addr, p.W[j]); }
#pragma unroll
  for (int j = 0; j < 4; ++j) q[j] = bperm(addr, p.q[j]);
#pragma unroll
  for (int j = 0; j < 6; ++j) prm[j] = bperm(addr, p.prm[j]);
  uint32_t r = 0;
  if constexpr (TRAJ) r = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)p.r19);
  __builtin_amdgcn_sched_barrier(0);  // all cross-lane reads are in flight before the first select waits for one
  if (take) {
#pragma unroll
    for (int j = 0; j < 3; ++j) { w.x[j] = x[j]; w.v[j] = v[j]; w.W[j] = T(W[j]); }
#pragma unroll
    for (int j = 0; j < 4; ++j) w.q[j] = q[j];
#pragma unroll
    for (int j = 0; j < 6; ++j) w.prm[j] = prm[j];
    if constexpr (TRAJ) r19 = r;
  }
}

}  // namespace qr
