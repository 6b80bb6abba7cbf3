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

// Unit quaternion of R = Rz(yaw) Ry(pitch) Rx(roll) (scipy 'xyz' extrinsic, quad.py:199)  <=>
// q = qz(yaw) qy(pitch) qx(roll), for |roll|, |pitch| <= 50 deg + margin and |yaw| <= pi.  The six
// half-angle sines / cosines and the products are float32 (the angles are random draws); the result
// is normalised in float64 (two first-order steps: 1e-7 -> 1e-14 -> 1e-28), so q has unit norm to
// float64 round-off.  Half roll / pitch are below 0.45 rad: plain Taylor polynomials, no reduction.
template <typename T>
__device__ __forceinline__ void sample_attitude(float yaw, float roll, float pitch, T (&q)[4]) {
  float sr, cr, sp, cp, sy, cy;
  {
    const float x = 0.5f * roll, x2 = x * x;
    sr = x * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 2.7557319e-6f, -1.9841270e-4f), 8.3333333e-3f), -1.6666667e-1f), 1.0f);
    cr = fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 2.4801587e-5f, -1.3888889e-3f), 4.1666667e-2f), -0.5f), 1.0f);
  }
  {
    const float x = 0.5f * pitch, x2 = x * x;
    sp = x * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 2.7557319e-6f, -1.9841270e-4f), 8.3333333e-3f), -1.6666667e-1f), 1.0f);
    cp = fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 2.4801587e-5f, -1.3888889e-3f), 4.1666667e-2f), -0.5f), 1.0f);
  }
  sincos_small(0.5f * yaw, sy, cy);
  const float crcp = cr * cp, srsp = sr * sp, srcp = sr * cp, crsp = cr * sp;
  double qd[4] = {(double)fmaf(crcp, cy, srsp * sy), (double)fmaf(srcp, cy, -crsp * sy), (double)fmaf(crsp, cy, srcp * sy),
                  (double)fmaf(crcp, sy, -srsp * cy)};
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const double r = fma(-0.5, fma(qd[0], qd[0], fma(qd[1], qd[1], fma(qd[2], qd[2], qd[3] * qd[3]))), 1.5);
#pragma unroll
    for (int j = 0; j < 4; ++j) qd[j] *= r;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) q[j] = T(qd[j]);
}

// QuadEnv.reset + sample_init_error + set_random_parameters (quad.py:171-222, 338-404) from the 20
// draws of one env.  Draw order: 0..5 m,d,J1,J3,c_tf,c_tw; 6,7 x0,x1; 8 x2; 9..11 v; 12..14 W;
// 15 zero-error branch; 16 yaw; 17 roll; 18 pitch; 19 goal-generator draws.
// Every sampled value is a float32 number (W is returned as such).
template <typename T, typename X>
__device__ __forceinline__ void sample_start(const Draws& d, bool randomise, bool eval, const Coeffs& c, X (&x)[3], X (&v)[3],
                                             T (&q)[4], float (&Wf)[3], float (&prm)[6]) {
  if (randomise) {  // float32 values: that is how the params buffer stores them
    const float p = c.udm;
#pragma unroll
    for (int j = 0; j < 5; ++j) prm[j] = fmaf(c.nom_f[j] * p, d.sym(j), c.nom_f[j]);
    prm[5] = fmaf(c.nom_f[5] * (0.5f * p), d.sym(5), c.nom_f[5]);
  } else {
#pragma unroll
    for (int j = 0; j < 6; ++j) prm[j] = c.nom_f[j];
  }
  float ix, iv, iR, iW;
  if (eval) {  // quad.py:352-356
    ix = 0.4f; iv = 0.0f; iR = 0.0f; iW = 0.0f;
  } else if (d.u01(15) < 0.2f) {  // quad.py:342-346
    ix = 0.0f; iv = 0.0f; iR = 0.0f; iW = 0.0f;
  } else {  // quad.py:348-351
    ix = 0.6f; iv = c.reset_v; iR = (float)(50.0 * kPi / 180.0); iW = c.reset_W;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    x[j] = X(ix * d.sym(6 + j));
    v[j] = X(iv * d.sym(9 + j));
    Wf[j] = iW * d.sym(12 + j);
  }
  sample_attitude<T>((float)kPi * d.sym(16), iR * d.sym(17), iR * d.sym(18), q);
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
// Only ~1 % of the envs reset in a given step, but more than half of the waves contain one and a
// launch ends with its slowest wave, so the reset's instructions sit on the critical path of the
// whole launch (measured with in-kernel clock stamps, tools/stamp_timeline.py).  Nothing about WHICH
// lane resets is known before the step has been integrated — but what a resetting lane needs
// (Philox draws + the sampling arithmetic) does not depend on the lane at all if the stream is
// keyed by the wave instead of by the env:
//     block b of slot s = Philox4x32-10(key = seed; ctr = (global id of the wave's first env [64 bit],
//                                       reset counter of this wave's tile, 0x40000000 | s << 8 | b))
// The wave samples 12 slots cooperatively — the five lanes 5s .. 5s+4 of slot s each take one
// Philox block (4 words) AND turn it into its share of the episode start:
//     b = 0: m, d, J1, J3      b = 1: c_tf, c_tw, x0, x1      b = 2: x2, v0, v1, v2
//     b = 3: W0, W1, W2, zero-error branch      b = 4: yaw, roll, pitch -> unit quaternion; word 3 = goal-generator draws
// so one pass is ~95 instructions of Philox + ~110 of sampling for up to 12 resets, run by a wave
// when one of its lanes resets (the per-lane scale / offset constants of the roles are formed at
// the wave's start, under the latency of its loads).  A lane that resets takes the slot given by
// its rank among the wave's resetting lanes
// (ds_bpermute from lanes 5 * rank + b); ranks >= 12 (e.g. a time limit ending all 64 episodes at
// once) draw further pools on demand (slots 12 p + s).  The tile's counter advances by one per
// env-step, so no (tile, counter, slot) is ever used twice — also under hipGraph replay, because
// the counter lives in device memory.  Results are independent of how the batch is sharded as
// long as shards start at multiples of 64 envs.
// ------------------------------------------------------------------------------------
struct PoolRole {  // per-lane constants of the lane's role: value_j = off_j + (zero-error ? scl_z_j : scl_j) * sym(word_j)
  float off[4], scl[4], scl_z[4];
};

__device__ __forceinline__ float sel5(int b, float v0, float v1, float v2, float v3, float v4) {  // branch-free 5-way select
  float v = v4;
  v = b == 3 ? v3 : v;
  v = b == 2 ? v2 : v;
  v = b == 1 ? v1 : v;
  v = b == 0 ? v0 : v;
  return v;
}

// The lane's role constants, from scalar coefficients only (~60 selects and multiplies): formed at the wave's start,
// between the arrival of the kernarg scalars (~0.45 us) and that of the working set (~0.85 us), when the SIMD has
// nothing else to do.  (Measured alternatives: the table built on the host and read per lane from the kernarg segment
// — host-visible memory, 1024 waves x 3 loads of it cost 0.4-0.9 us per launch wherever they sat in the load queue.)
__device__ __forceinline__ void pool_role(PoolRole& r, bool randomise, bool eval, const Coeffs& c) {
  const int lane = (int)__lane_id();
  const int b = lane - 5 * (lane / 5);
  // (all coefficient reads are unconditional scalar loads; the per-lane part is selects only)
  const float n0 = c.nom_f[0], n1 = c.nom_f[1], n2 = c.nom_f[2], n3 = c.nom_f[3], n4 = c.nom_f[4], n5 = c.nom_f[5];
  const float rv = c.reset_v, rW = c.reset_W;
  const float p = randomise ? c.udm : 0.0f;
  const float ix = eval ? 0.4f : 0.6f, iv = eval ? 0.0f : rv, iW = eval ? 0.0f : rW;
  const float iR = eval ? 0.0f : (float)(50.0 * kPi / 180.0);
  const float zx = eval ? ix : 0.0f;  // (the zero-error branch exists in 'train' only; with 'eval' it changes nothing)
  const float pi = (float)kPi;
  // role table: value_j = off_j + scl_j * sym_j, scl_z_j in the zero-error branch
  //   b = 0: m, d, J1, J3   1: c_tf, c_tw, x0, x1   2: x2, v0, v1, v2   3: W0, W1, W2, branch   4: yaw, roll, pitch, raw
  r.off[0] = sel5(b, n0, n4, 0.f, 0.f, 0.f); r.off[1] = sel5(b, n1, n5, 0.f, 0.f, 0.f);
  r.off[2] = sel5(b, n2, 0.f, 0.f, 0.f, 0.f); r.off[3] = sel5(b, n3, 0.f, 0.f, 0.f, 0.f);
  r.scl[0] = sel5(b, n0 * p, n4 * p, ix, iW, pi);            r.scl_z[0] = sel5(b, n0 * p, n4 * p, zx, 0.f, pi);
  r.scl[1] = sel5(b, n1 * p, n5 * (0.5f * p), iv, iW, iR);   r.scl_z[1] = sel5(b, n1 * p, n5 * (0.5f * p), 0.f, 0.f, 0.f);
  r.scl[2] = sel5(b, n2 * p, ix, iv, iW, iR);                r.scl_z[2] = sel5(b, n2 * p, zx, 0.f, 0.f, 0.f);
  r.scl[3] = sel5(b, n3 * p, ix, iv, 0.f, 0.f);              r.scl_z[3] = sel5(b, n3 * p, zx, 0.f, 0.f, 0.f);
}

template <typename T>
struct ResetPool {  // this lane's share of its slot
  float v[4];       // the four sampled values of the lane's role (role 4, word 3: the raw draw, as bits)
  T q[4];           // role 4: the unit quaternion
};

template <typename T>
__device__ __forceinline__ void make_pool(ResetPool<T>& p, const PoolRole& role, uint64_t seed, uint64_t gfirst, uint32_t count, int pass) {
  const int lane = (int)__lane_id();
  const int k = lane / 5, b = lane - 5 * k;  // slot / block of this lane (k = 12: lanes 60..63 idle)
  uint32_t ctr[4] = {(uint32_t)gfirst, (uint32_t)(gfirst >> 32), count, 0x40000000u | ((uint32_t)(12 * pass + k) << 8) | (uint32_t)b};
  philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
  // zero-error branch of the slot (quad.py:342-346): word 3 of role 3, broadcast to the slot's lanes
  const float u = fmaf((float)(ctr[3] >> 8), 0x1p-24f, 0x1p-25f);
  const bool zero = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((5 * k + 3) << 2, __builtin_bit_cast(int, u))) < 0.2f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float sym = fmaf((float)(ctr[j] >> 8), 0x1p-23f, 0x1p-24f - 1.0f);
    p.v[j] = fmaf(zero ? role.scl_z[j] : role.scl[j], sym, role.off[j]);
  }
  sample_attitude<T>(p.v[0], p.v[1], p.v[2], p.q);  // meaningful in role 4 (yaw, roll, pitch)
  if (b == 4) p.v[3] = __builtin_bit_cast(float, ctr[3]);
}

// The pool as the helper wave of a workgroup leaves it in LDS for the stepping wave (HELP instantiation of the step
// kernel): 16 bytes of role values per lane, the quaternion of a slot from its role-4 lane.
template <typename T>
struct PoolLds {
  float4 v[64];
  T q[12][4];
};

template <typename T>
__device__ __forceinline__ void pool_to_lds(PoolLds<T>& s, const ResetPool<T>& p) {
  const int lane = (int)__lane_id();
  const int k = lane / 5, b = lane - 5 * k;
  s.v[lane] = make_float4(p.v[0], p.v[1], p.v[2], p.v[3]);
  if (b == 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) s.q[k][j] = p.q[j];
  }
}

// Lanes with take == true copy slot `slot` of the pool (in LDS) into their working set.  Executed by the whole wave.  (Through
// LDS — six 16-byte reads per taking lane — rather than 23 ds_bpermute with all their results in flight at once: 128 instead of
// 142 VGPRs for the plain Quad-v0 kernel, four waves per SIMD instead of three)
template <typename T, typename X, bool TRAJ>
__device__ __forceinline__ void take_from_lds(const PoolLds<T>& s, bool take, int slot, Work<T, X>& w, uint32_t& r19) {
  const int k = take ? slot : 0;
  const float4 r0 = s.v[5 * k], r1 = s.v[5 * k + 1], r2 = s.v[5 * k + 2], r3 = s.v[5 * k + 3];
  T q[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) q[j] = s.q[k][j];
  float rb = 0.0f;
  if constexpr (TRAJ) rb = s.v[5 * k + 4].w;
  if (take) {
    w.prm[0] = r0.x; w.prm[1] = r0.y; w.prm[2] = r0.z; w.prm[3] = r0.w; w.prm[4] = r1.x; w.prm[5] = r1.y;
    w.x[0] = X(r1.z); w.x[1] = X(r1.w); w.x[2] = X(r2.x);
    w.v[0] = X(r2.y); w.v[1] = X(r2.z); w.v[2] = X(r2.w);
    w.W[0] = T(r3.x); w.W[1] = T(r3.y); w.W[2] = T(r3.z);
#pragma unroll
    for (int j = 0; j < 4; ++j) w.q[j] = q[j];
    if constexpr (TRAJ) r19 = __builtin_bit_cast(uint32_t, rb);
  }
}

}  // namespace qr
