// qr_args.h — part of the gfx950 quadrotor step library (included by quadrotor_kernels.hip, in this order).
// Kernel argument block, constants and the per-env working set.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "quadrotor_hip.h"

namespace qr {

// ------------------------------------------------------------------------------------
// Kernel argument block (passed by value in kernarg memory)
// ------------------------------------------------------------------------------------
// Coefficients as the kernels consume them.  Only what enters float64 arithmetic is a double:
// every field that is used sits in SGPRs for the whole kernel, and with ~50 doubles the round-1
// kernel spilled SGPRs into VGPR lanes (42 v_writelane + 12 v_readlane on the step's critical
// path).  Reciprocals are formed once on the host (an f64 division is ~35 VALU slots).
struct Coeffs {
  // float64: step length, the limits observations / terminations are formed with, nominal parameters
  double dt, inv_x_lim, inv_v_lim, inv_W_lim, W_lim, sin_euler_lim, tan_euler_lim, inv_w_adapt;
  double x_lim, v_lim;                       // (float64 layout only: x, v held as doubles)
  double nom[6], g, min_force;               // quad.py:28-36 (a QuadConstants may override them)
  // float32: rewards (formed in float32 on float32 observations, like NumPy >= 2), integrators, reset ranges
  float Cx, CIx, Cv, Cb1, CIb1, CW, Cw12, CW3, alpha, beta, hdt;
  float x_lim_f, x_lim_up, v_lim_up;         // *_up = smallest float >= the limit: |x_f32| < up  <=>  |x_f32| < limit
  float inv_eIx_lim, inv_eIb1_lim;
  float rmin_mono, rmin_1, rmin_2, inv_nrmin_mono, inv_nrmin_1, inv_nrmin_2;
  float udm, reset_v, reset_W;               // reset sampling: UDM fraction, v_lim / 2, W_lim / 2 (quad.py:348-351)
  float nom_f[6], g_f;
  // eight-shaped curve (trajectory_generator.py:98-110, 418-505)
  float e8_w1, e8_w2, e8_k, e8_A1, e8_A2, e8_wb, e8_alt, e8_tmax;
};

struct ActorW {  // QrActor's tensors (torch.nn.Linear layout: weight [out][in])
  const float *fc1_w, *fc1_b, *fc2_w, *fc2_b, *mean_w, *mean_b, *log_std;
  const float *ls_w, *ls_b;  // optional state-dependent log_std head (SAC); log_std is then unused
  int32_t squash;            // QR_ACTOR_TANH_MEAN | QR_ACTOR_TANH_SAMPLE
};

struct Args {
  // Field ORDER matters for speed only: the kernarg segment lives in host-visible memory and a scalar-cache miss on it
  // costs the stepping wave ~0.3 us wherever it waits for one.  The step kernel receives the buffers its loads need as
  // leading (preloaded) arguments; of the rest, the pointers it reads late — in the epilogue and the reset block — share
  // ONE 64-byte line (kernarg bytes 0x40-0x7f = Args bytes 0x08-0x47) with those its prologue reads (goal, steps), so
  // that line is warm when the epilogue wants it.
  void* pos_vel;
  float* goal;
  int32_t* steps;
  int32_t* episode;
  uint8_t* done;
  uint8_t* truncated;
  float* reward;
  float* reward_raw;
  float* final_obs0;      // optional: pre-reset observation rows of envs that are re-sampled in the launch
  // per-env buffers
  void* att_rate;
  float* integ;
  float* params;
  float* traj;
  int32_t* reset_count;   // [ceil(N/64)] per-tile counter of the in-launch reset stream
  // per-call
  const float* action;
  float* obs0;
  float* obs1;
  float* final_obs1;
  const uint8_t* mask;
  double* rows_out;       // qr_get_state
  const double* rows_in;  // qr_set_state
  int32_t* status;        // qr_set_state: count of rejected rows (det R <= 0 / non-finite)
  const float* draws;     // qr_traj_start: injected [3][N] theta_b1d, t_traj, w_b1d
  float* goal_rows;       // qr_get_desired: [N][15]
  int32_t goal_mode;
  int32_t store_goal;     // qr_get_desired: also write the goal buffer
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
  int32_t dry_run;        // qr_check_state: set_state_kernel validates and counts, writes nothing
  int32_t tile_base;      // multi-step helper launches split into chunks of resident tiles: the first tile of this launch (else 0)
  Coeffs c;
#ifdef QR_SPAN            // diagnostic build (tools/span_timeline.py); BEHIND the coefficient block: the product's kernarg layout is untouched
  unsigned long long* span_buf;  // [slots][2 * tiles][2] clock stamps (NULL: none)
  int32_t span_slot;      // which row of the span buffer this launch writes (-1: none)
#endif
};

// Cache policy of the step kernel's stores (gfx940+ aux bits of the buffer / global stores: 1 = sc0, 2 = nt, 16 = sc1; ONE value
// per launch, applied to the SoA buffer stores as aux bits and to the row / reward / done stores by gstore<AUX> below).
// Measured on MI355X (profiles/r03/ab_store_policy_all_kinds.txt, ab_load_order_store_policy.txt): with every store of a
// launch written THROUGH (sc1) the launch-to-launch time of the helper-wave launches drops by 8-15 % (Quad-v0 65 536 envs
// 4.41 -> 4.00 us, Coupled 5.97 -> 5.10, Decoupled 32 768 4.98 -> 4.31): a kernel that ends with clean L2s has nothing to
// write back between its last wave and the next dispatch.  nt and sc0 alone change nothing.  The plain launches of large
// grids are bound by bytes and gain nothing (1 M envs: equal; Coupled 131 072 envs 9.2 -> 9.6-9.9 us), and a rollout has one
// boundary per horizon and pays for the write-through instead (Quad-v0 65 536 envs, T = 100: 1.73 -> 1.91 us per env-step,
// profiles/r03/ab_rollout_store_policy.txt): both keep plain stores.
#ifndef QR_HELP_AUX
#define QR_HELP_AUX 16   // one-step helper-wave instantiations (grids in the launch-latency regime)
#endif
#ifndef QR_PLAIN_AUX
#define QR_PLAIN_AUX 0   // everything else
#endif

// A store of a caller-facing output (any address, per lane) with the cache policy AUX: 0 = a plain store; otherwise the SAME aux
// bits the SoA buffer stores of the launch carry (SoA::store<AUX>), expressed the way the compiler offers them for flat global
// stores — as the scope of a relaxed atomic store (gfx950 memory model: workgroup = sc0, agent = sc1, system = sc0 sc1):
//   AUX 16 (sc1)      -> agent scope      (the product's write-through policy, QR_HELP_AUX)
//   AUX 17 (sc0 sc1)  -> system scope
//   AUX  1 (sc0)      -> workgroup scope
// 16-byte stores have no atomic form: inline asm with the same bits.  The aux encoding and the sc0 / sc1 modifiers are gfx940+.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "qr_args.h: the store cache policies (sc0 / sc1 aux bits) are written for gfx950 (Makefile: ARCH)"
#endif
typedef float f4_t __attribute__((ext_vector_type(4)));
#ifndef QR_GSTORE_EXTRA_BITS
#define QR_GSTORE_EXTRA_BITS 0  // measurement builds only: 1 = sc0 on top of the launch's bits for these stores (round 3 wrote them at system
#endif                          // scope, sc0 sc1, whatever AUX said; profiles/r04/ab_gstore_scope.txt)
template <int AUX_IN = 0, typename V>
__device__ __forceinline__ void gstore(V* p, V v) {
  static_assert(AUX_IN == 0 || AUX_IN == 16 || AUX_IN == 17 || AUX_IN == 1, "store policy: 0 (plain), 16 (sc1), 17 (sc0 sc1) or 1 (sc0)");
  constexpr int AUX = AUX_IN == 0 ? 0 : (AUX_IN | QR_GSTORE_EXTRA_BITS);
  constexpr int kScope = AUX == 16 ? __HIP_MEMORY_SCOPE_AGENT : AUX == 17 ? __HIP_MEMORY_SCOPE_SYSTEM : __HIP_MEMORY_SCOPE_WORKGROUP;
  if constexpr (AUX == 0) {
    *p = v;
  } else if constexpr (sizeof(V) == 1) {
    __hip_atomic_store(reinterpret_cast<uint8_t*>(p), __builtin_bit_cast(uint8_t, v), __ATOMIC_RELAXED, kScope);
  } else if constexpr (sizeof(V) == 2) {
    __hip_atomic_store(reinterpret_cast<uint16_t*>(p), __builtin_bit_cast(uint16_t, v), __ATOMIC_RELAXED, kScope);
  } else if constexpr (sizeof(V) == 4) {
    __hip_atomic_store(reinterpret_cast<uint32_t*>(p), __builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED, kScope);
  } else if constexpr (sizeof(V) == 8) {
    __hip_atomic_store(reinterpret_cast<uint64_t*>(p), __builtin_bit_cast(uint64_t, v), __ATOMIC_RELAXED, kScope);
  } else {
    static_assert(sizeof(V) == 16, "1, 2, 4, 8 or 16 bytes");
    const f4_t x = __builtin_bit_cast(f4_t, v);
    if constexpr (AUX == 16) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
    else if constexpr (AUX == 17) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(x) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(x) : "memory");
  }
}

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

// Per-env working set held in VGPRs: the 13-word state (x, v, unit quaternion q = (w,x,y,z), W) plus
// parameters, goal and integrator words.  T is the type q and W are held and accumulated in, X the
// type of x and v (mixed layout: T = double, X = float — what is STORED as float32 is also HELD as
// float32: the kernel's occupancy is set by its VGPR count).
template <typename T, typename X>
struct Work {
  X x[3];
  X v[3];
  T q[4];
  T W[3];
  float prm[6];    // m, d, J1(=J2), J3, c_tf, c_tw (quad.py:359-387); the nominal values when not randomised
  float goal[12];  // xd, vd, b1d, Wd
  float integ[8];  // eIx, g_x prev, eIb1, g_b prev
  bool nominal;    // parameters are the exact float64 nominal values, not prm[]
};

template <typename T>
struct Phys {  // what set_random_parameters derives (quad.py:389-404), formed when needed
  T m, d, J1, J3, ctf, ctw;
  T min_force, max_force, avrg_act, scale_act;
  template <typename W>
  __device__ __forceinline__ Phys(const W& w, const Coeffs& c) {
    if (w.nominal) {
      m = T(c.nom[0]); d = T(c.nom[1]); J1 = T(c.nom[2]); J3 = T(c.nom[3]); ctf = T(c.nom[4]); ctw = T(c.nom[5]);
    } else {
      m = T(w.prm[0]); d = T(w.prm[1]); J1 = T(w.prm[2]); J3 = T(w.prm[3]); ctf = T(w.prm[4]); ctw = T(w.prm[5]);
    }
    const T hover = m * T(c.g * 0.25);
    min_force = T(c.min_force);
    max_force = ctw * hover;
    avrg_act = (min_force + max_force) * T(0.5);
    scale_act = max_force - avrg_act;
  }
};

}  // namespace qr
