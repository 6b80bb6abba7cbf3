// quadrotor_kernels.hip — fused env.step() kernels for gfx950 (MI355X / CDNA4).
//
// One lane = one quadrotor.  A launch does, per env and per env-step, everything the
// reference's QuadEnv.step template does (gym_rotor/envs/quad.py:142-168):
//   action map / motor mixing -> S fixed RK4 substeps of the rigid-body ODE on
//   R^3 x R^3 x SO(3) x R^3 (quad.py:321-335) with zero-order-hold (f, M) -> error
//   observation + trapezoid integrators (quad.py:421-466) -> reward -> np.interp
//   normalisation -> done -> crash override [-> auto-reset].
//
// What bounds it: at N >= ~250 000 envs the bytes — the step streams its working set through the
// fabric at 67-73 % of the HBM roofline; at the metric's N = 65 536 (1024 tiles = one stepping wave per SIMD) the
// launch boundary (1.8 us between dependent kernels) plus the length of ONE wave's instruction stream, which a lone
// wave issues at one VALU instruction per ~5.6 cycles whatever its type.  So the design minimises BYTES per env and
// INSTRUCTIONS on the stepping wave's path:
//   * attitude is a unit quaternion integrated directly — q' = q (0,W)/2 is the same flow as R' = R hat(W) — and
//     stored as its smallest three components: the state is 12 words, not 18; R(q) is rebuilt in registers only
//     where the observation / reward needs it;
//   * x, v are stored as float32 and q, W as float64 in the default (mixed) layout; W is integrated and q accumulated
//     in float64, the RK4 stage quaternions are float32 (qr_dynamics.h);
//   * the whole working set stays in VGPRs across substeps and, in qr_rollout, across env-steps: HBM is touched once
//     in and once out;
//   * for grids small enough that every wave is resident at once, each tile gets a second, HELPER wavefront in the
//     same workgroup (HELP): it samples the tile's reset pool, forms Quad-v0's reward, samples the policy's noise
//     and carries observation rows out, in issue slots the lone stepping wave leaves empty;
//   * arguments are read so that scalar-cache misses on the kernarg segment (~0.4 us each) stay off the stepping
//     wave's critical path.
// Per-env SoA buffers are read/written with lane-contiguous accesses through buffer descriptors; caller-facing AoS
// rows (actions, observations) go through LDS so global traffic is linear 16-byte-per-lane stores.  The physics has
// no contraction larger than 3x3, so no MFMA there; the one real contraction on the path — the 16-wide PPO actor of
// qr_rollout_actor — does run on the matrix cores (qr_actor.h).
//
// Written directly for CDNA4: 64-lane wavefronts, one stepping wavefront per 64-env tile (N = 65 536 -> 1024
// workgroups = one per SIMD-32), workgroups of one or two wavefronts, 120-128 VGPRs for the Quad-v0 one-step kernels
// (four waves per SIMD at large N).
//
// Files (included in this order):
//   qr_args.h      kernel argument block, constants, per-env working set
//   qr_rng.h       Philox4x32-10, wave-cooperative reset draws, reset sampling
//   qr_dynamics.h  attitude helpers, quaternion-form RHS + RK4, row transposes, action maps, error obs
//   qr_traj.h      goal generator (trajectory_generator.py modes 0/1/6), SoA buffer accessor
//   qr_actor.h     PPO actor (MFMA / LDS forms), action sampling
//   this file      step / rollout kernel, auxiliary kernels, host launchers and the C-ABI
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <type_traits>

#include "quadrotor_hip.h"

#ifndef QR_ABLATE
#define QR_ABLATE 0  // 0 = product build; measurement-only builds (Makefile: evidence-libs): 1 = the kernel returns at once
#endif               // (launch floor), 2 = loads and stores only (no integration)

#include "qr_args.h"
#include "qr_rng.h"
#include "qr_dynamics.h"
#include "qr_traj.h"
#include "qr_actor.h"

namespace qr {

// byte offset of the Args block in the step kernel's kernarg segment: 6 pointers + 2 x int32 precede it
[[maybe_unused]] constexpr int kArgsOffset = 6 * 8 + 2 * 4;
static_assert(alignof(Args) == 8, "Args follows the leading scalar arguments without padding");
#ifndef QR_SPAN
static_assert(sizeof(Coeffs) <= 5 * 64 && offsetof(Args, c) + sizeof(Coeffs) == sizeof(Args), "the step kernel touches the five kernarg lines of the coefficient block (the last field of Args)");
#endif

// QR_STAMPS: diagnostic build (tools/stamp_timeline.py).  Every wave records the 100 MHz real-time clock at
// seven points of the step; the values go to a buffer of their own that nothing else reads.
#ifdef QR_STAMPS
__device__ unsigned long long* g_stamps = nullptr;
#define QR_STAMP(k, dep)                                                                          \
  do {                                                                                            \
    unsigned long long t_;                                                                        \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(dep) : "memory");    \
    if (g_stamps != nullptr && lane == 0) g_stamps[(size_t)blockIdx.x * 8 + (k)] = t_;            \
  } while (0)
// the helper wave's stamps: rows gridDim.x .. 2 gridDim.x - 1 of the same buffer
#define QR_HSTAMP(k, dep)                                                                         \
  do {                                                                                            \
    unsigned long long t_;                                                                        \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(dep) : "memory");    \
    if (g_stamps != nullptr && threadIdx.x == 64) g_stamps[((size_t)gridDim.x + blockIdx.x) * 8 + (k)] = t_; \
  } while (0)
// the multi-step launches (tools/phase_timeline.py): row (tile * n_steps + t) of the buffer, stamp k of step t
#define QR_PSTAMP(k, dep)                                                                         \
  do {                                                                                            \
    unsigned long long t_;                                                                        \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(dep) : "memory");    \
    if (g_stamps != nullptr && lane == 0) g_stamps[((size_t)blockIdx.x * n_steps + t) * 8 + (k)] = t_; \
  } while (0)
// where the hardware put each wave (tools/wave_placement.py): HW_ID | XCC_ID << 32, row blockIdx.x, column = wave of the workgroup
__device__ unsigned long long* g_hwid = nullptr;
#define QR_HWID()                                                                                 \
  do {                                                                                            \
    unsigned h_, x_;                                                                              \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h_));                              \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x_));                             \
    if (g_hwid != nullptr && (threadIdx.x & 63u) == 0)                                            \
      g_hwid[(size_t)blockIdx.x * 2 + (threadIdx.x >> 6)] = (unsigned long long)h_ | ((unsigned long long)x_ << 32); \
  } while (0)
#else
#define QR_STAMP(k, dep) do { } while (0)
#define QR_HSTAMP(k, dep) do { } while (0)
#define QR_PSTAMP(k, dep) do { } while (0)
#define QR_HWID() do { } while (0)
#endif

// QR_SPAN: the light diagnostic build (tools/span_timeline.py).  Every wave records the 100 MHz real-time clock twice — with its first
// instruction and behind its last — into row `span_slot` of a buffer of its own; a chain of launches with slots 0, 1, 2, ... then
// shows, on the device's own clock, each launch's SPAN (first wave in to last wave out) and the GAP to the next launch.  Two scalar
// memory-time reads per wave and one 16-byte store at the very end: the build runs within a few per cent of the product's period
// (the seven-stamp QR_STAMPS build: +50 %), which is what makes span + gap a usable clock for kernels rocprofv3 inflates.
#ifdef QR_SPAN
static unsigned long long* g_span_buf = nullptr;  // (host) the stamp buffer and the row the next launches write: qr_debug_set_span[_slot]
static int g_span_slot = -1;
// Buffer pointer and row come with the launch's own kernarg.  A clock read is a scalar
// memory operation whose result lands asynchronously: it is WAITED FOR on the spot (the compiler knows nothing of the pending
// write and would otherwise reuse the register pair), which costs its wave ~0.3 us.  So only a sample of the waves pays:
//   ENTRY stamps: the first eight workgroups (one per XCD; the dispatcher starts with them) — a launch's "first wave in";
//   EXIT stamps: the workgroups of every fourth tile, spread over the XCDs — a launch's "last wave out" is then a stamped one in
//   a quarter of the launches, and the chain's MEDIAN period stays within ~1 % of the product's.
// (buffer pointer and row are read from the kernarg segment by the stamped waves only, at their end and BEHIND the clock read: one more
// kernarg line requested at the kernel's start would sit in every wave's first scalar wait — 0.3 us per launch)
#define QR_SPAN_BEGIN()                                                  \
  unsigned long long span_t0_ = 0;                                       \
  if (blockIdx.x < 8u) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(span_t0_) : : "memory")
#define QR_SPAN_END()                                                                                          \
  do {                                                                                                         \
    const bool exit_ = (((blockIdx.x >> 3) + blockIdx.x) & 3u) == 0u;                                          \
    if (exit_ || blockIdx.x < 8u) {                                                                            \
      unsigned long long t1_ = 0;                                                                              \
      if (exit_) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_) : : "memory");            \
      unsigned long long* const span_buf_ = ka.span_buf;                                                       \
      const int span_slot_ = ka.span_slot;                                                                     \
      if (span_buf_ != nullptr && span_slot_ >= 0 && (threadIdx.x & 63u) == 0) {                               \
        const size_t w_ = ((size_t)span_slot_ * (((unsigned)n_envs + 63u) >> 6) + blockIdx.x) * 2 + (threadIdx.x >> 6); \
        span_buf_[2 * w_] = span_t0_; span_buf_[2 * w_ + 1] = t1_;                                             \
      }                                                                                                        \
    }                                                                                                          \
  } while (0)
#else
#define QR_SPAN_BEGIN() do { } while (0)
#define QR_SPAN_END() do { } while (0)
#endif

#ifndef QR_STEP_PRIO
#define QR_STEP_PRIO 3  // s_setprio of the stepping wave in the helper-wave launches (0: the A/B arm without it)
#endif
#ifndef QR_HELP_REWARD_TILES
#define QR_HELP_REWARD_TILES 1408  // one-step Quad-v0 helper launches beyond this many tiles form the reward on the stepping wave
#endif
#ifndef QR_HELP_ROWS_TILES
#define QR_HELP_ROWS_TILES 1600  // one-step wrapper helper launches beyond this many tiles store their rows from the stepping wave
#endif
#ifndef QR_PRIO_SUBSTEPS
#define QR_PRIO_SUBSTEPS 2
#endif
#ifndef QR_HELPER_GRID_WRAP_SUBSTEPS
#define QR_HELPER_GRID_WRAP_SUBSTEPS 1664  // the wrappers' one-step helper-wave launches with two or more substeps: up to this many tiles (wants_helper)
#endif
#ifndef QR_PRIO_SINGLE_TILES
#define QR_PRIO_SINGLE_TILES 768
#endif
#ifndef QR_DELTA_STAGES
#define QR_DELTA_STAGES 1  // 0: the rate-adaptive instantiations use the plain stage arithmetic (numerics: tools/numerics_delta.py and the free-run rows of
                           // profiles/r03/parity_summary.txt, 6.8e-6 -> 2.4e-6; cost: the "free run in regime" rows of profiles/r03/runtime_ab.json, 4.39 against 4.03 us)
#endif
#ifndef QR_EARLY_STORE_GRID
#define QR_EARLY_STORE_GRID 4096  // grids up to this many waves store a resetting wave's settled lanes before it samples (65 536 envs 5.43 / 5.49 us, 1 M 39.4 / 37.4)
#endif
#ifndef QR_HELP_POLICY
#define QR_HELP_POLICY 1  // 0: no helper wave in qr_rollout_actor (Coupled 65 536 envs, T = 32: 5.37 -> 4.51 us per env-step)
#endif
#ifndef QR_HELP_REWARD
#define QR_HELP_REWARD 1        // the helper wave forms Quad-v0's reward (§3.3 item 2; evidence build q_norew: profiles/r03/ab_quad_builds.txt)
#endif
// (Settled A/Bs whose losing arms are gone from the tree — the winning arm is the code, the measurement is cited where it applies:
//  kernarg lines requested with the wave's first instructions (Decoupled 5.16 -> 5.02 us), output pointers read with the first scalar
//  batch in the plain launches (1 M envs 34.7 -> 33.9 us), rows to the LDS tile before the reset block + late goal / integrator loads
//  in the plain wrapper launches (262 144 envs 19.1 -> 16.5 us), role constants formed in the reset block of one-step launches
//  (142 -> 128 VGPRs), observation rows carried out by the helper wave, per-episode action-map constants in the rollouts
//  (profiles/r04/ab_hoist_act.txt).  DESIGN.md / docs/EXPERIMENTS.md name the files.)
#ifndef QR_XCD_GRID
#define QR_XCD_GRID 1536   // one-step helper-wave launches up to this many tiles give every XCD a contiguous range of tiles (see tile_id); 0 = never
#endif
#ifndef QR_HELPER_GRID
// Grids up to this many tiles run the one-step kernel with a helper wave per tile (HELP).  The limit is an EMPIRICAL crossover,
// not a residency rule: 2560 tiles are 5120 waves, more than the 4096 wave slots the 120-VGPR kernel has at four waves per SIMD —
// the helper waves are short-lived and the launch still wins there (profiles/r03/ab_helper_thresholds.txt, with the write-through
// stores of DESIGN.md 3.5: Quad-v0 163 840 envs 7.3 against 8.3 us plain, 196 608 equal, 262 144 10.3 against 9.9).  Round 5, with
// the reward on the stepping wave beyond QR_HELP_REWARD_TILES (one substep): 196 608 envs 8.0-8.2 against 8.8-8.9 plain, 229 376
// 8.6-9.2 against 9.7-9.8, 245 760 9.1-9.9 against 9.9-10.1, 262 144 9.8-10.6 against 10.1-10.3 (profiles/r05/ab_step_prio.txt):
// 3328 tiles for one substep; launches with more substeps keep 2560.  The environment variable QR_HELPER_GRID and the
// QR_FLAG_*_HELPER bits override it (see `tuning`).
#define QR_HELPER_GRID 3328
#endif
#ifndef QR_HELPER_GRID_SUBSTEPS
#define QR_HELPER_GRID_SUBSTEPS 2560  // Quad-v0 with >= 2 substeps or the fused goal generator (its own guard: a build may set QR_HELPER_GRID alone)
#endif
#ifndef QR_HELPER_GRID_ROLLOUT
#define QR_HELPER_GRID_ROLLOUT (QR_HELPER_GRID < 1024 ? QR_HELPER_GRID : 1024)  // qr_rollout / qr_rollout_actor (two waves per SIMD)
#endif
#ifndef QR_HELPER_GRID_WRAP
#define QR_HELPER_GRID_WRAP (QR_HELPER_GRID < 2560 ? QR_HELPER_GRID : 2560)  // the wrappers: ahead of the plain launch up to 262 144 envs while the action rows come
// from cache (r03/ab_helper_thresholds.txt, 8 slabs: 14.9 against 15.7 us), behind it beyond 131 072 envs when they stream from HBM (r03/ab_helper_wave.txt, 64 slabs:
// 131 072 envs 9.4 against 9.1 us, 262 144 envs 18.4 against 16.5 — three stepping waves per SIMD hide less latency than four).  Round 5, with the rows on
// the stepping wave beyond QR_HELP_ROWS_TILES: ahead up to 163 840 envs with either action source (10.2-10.4 against 10.5-10.7), mixed at 196 608: 2560 tiles
// (one substep; 1664 with more: QR_HELPER_GRID_WRAP_SUBSTEPS; 2048 with the fused goal generator)
#endif
// ------------------------------------------------------------------------------------
// Quad-v0 reward and termination (quad.py:274-318) from the post-step state
// ------------------------------------------------------------------------------------
// reward_wrapper (quad.py:274-298), formed in float32 (its result is a float32 word)
template <typename T, typename X>
__device__ __forceinline__ float quad_reward_raw(const X (&x)[3], const X (&v)[3], const T (&q)[4], const T (&W)[3],
                                                 const float (&goal)[12], const Coeffs& c) {
  const T qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  const T R00 = fma_1m2(fma_ss(qy, qy, qz, qz)), R10 = T(2) * fma_ss(qx, qy, qw, qz);  // b1 = first column of R(q)
  float eX2 = 0.f, eV2 = 0.f, W2 = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float dx = (float)x[j] - goal[j], dv = (float)v[j] - goal[3 + j], wj = (float)W[j];
    eX2 = fmaf(dx, dx, eX2); eV2 = fmaf(dv, dv, eV2); W2 = fmaf(wj, wj, W2);
  }
  // eb1 = signed angle from b1d to b1_proj ~ (R00, R10, 0) (quad_utils.py:97-101,157-177).
  // acos(du.cu) with the sign of (du x cu)_z == atan2(|du x cu|, du.cu), which is invariant
  // to the lengths of both vectors, so neither is normalised.
  const float r00 = (float)R00, r10 = (float)R10;
  const float g6 = goal[6], g7 = goal[7], g8 = goal[8];
  const float dot = g6 * r00 + g7 * r10;
  const float cz = g6 * r10 - g7 * r00;
  const float hy2 = r00 * r00 + r10 * r10;
  const float sabs = sqrtf(g8 * g8 * hy2 + cz * cz);
  float ang = atan2_fast(sabs, dot);
  if (cz < 0.0f) ang = -ang;
  const float eb1 = ang * (float)(1.0 / kPi);
  return -c.Cx * eX2 - c.Cb1 * fabsf(eb1) - c.Cv * eV2 - c.CW * W2;
}

// done_wrapper (quad.py:301-318): roll = atan2(R21,R22), pitch = -asin(R20); |angle| >= 85 deg
// without inverse trig.  x, v: float32 numbers compared with the limit rounded UP to float32,
// which decides exactly as the float64 comparison does.  (bitwise | on purpose: no branches)
template <typename T, typename X>
__device__ __forceinline__ bool quad_done(const X (&x)[3], const X (&v)[3], const T (&q)[4], const T (&W)[3], const Coeffs& c) {
  const T qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  const T R20 = T(2) * fma_sd(qx, qz, qw, qy), R21 = T(2) * fma_ss(qy, qz, qw, qx), R22 = fma_1m2(fma_ss(qx, qx, qy, qy));
  X xl, vl;
  if constexpr (std::is_same<X, float>::value) { xl = c.x_lim_up; vl = c.v_lim_up; } else { xl = X(c.x_lim); vl = X(c.v_lim); }
  bool d = false;
#pragma unroll
  for (int j = 0; j < 3; ++j)
    d = d | !(fabs(x[j]) < xl) | !(fabs(v[j]) < vl) | !(fabs(W[j]) < T(c.W_lim));
  d = d | !(fabs(R20) < T(c.sin_euler_lim));          // |pitch| >= lim
  d = d | !(fabs(R21) < T(c.tan_euler_lim) * R22);    // |atan2(R21,R22)| >= lim
  return d;
}

// The post-step state of a tile as its stepping wave leaves it in LDS for the helper wave (HELP, Quad-v0), which
// forms and stores the reward from it.
template <typename T, typename X>
struct PostLds {
  X x[3][64], v[3][64];
  T q[4][64], W[3][64];
  uint32_t done[64];  // the stepping wave's termination flag (the helper's crash override needs it: not formed twice)
};

// ------------------------------------------------------------------------------------
// The fused step / rollout kernel
// ------------------------------------------------------------------------------------
// TRAJ != 0 = the goal generator (utils/trajectory_generator.py) is fused into the step; separate instantiations
// so that the default path carries none of its registers: 1 = the stateless modes 0 / 1 / 6, 2 = the stateful modes 2-5
// (take-off, landing, stay, circle: persistent goal fields loaded / stored with the working set).  ADAPT = the
// rate-adaptive substep count (QrCoeffs::w_adapt); launch_kind() picks the plain instantiation
// whenever adaptivity provably cannot trigger.
// POLICY != 0 = qr_rollout_actor: the action of every step comes from the actor(s) evaluated on the
// env's current observation, which stays in registers from one step to the next.  1: PPO / TD3 actors
// (parameter log_std, tanh-of-mean rule); 2: any reference MLP actor (adds SAC's log_std head and rule).
// SINGLE = exactly one env-step per launch (qr_step): no loop over steps, so nothing is hoisted out of it and kept
// live across the whole kernel (fewer SGPRs to spill, a shorter prologue).
// HELP (with SINGLE, for grids in the launch-latency regime) = the workgroup carries a second wavefront that does
// nothing but sample the tile's reset pool into LDS while the stepping wave waits for its loads and integrates:
// a lone wave issues one VALU instruction per ~5.6 cycles, two waves on a SIMD one per ~2.9 (tools/valu_microbench.hip),
// so the helper runs in issue slots that are otherwise empty, and the stepping wave's reset block shrinks from
// ~230 instructions (Philox, role scaling, attitude, 24 cross-lane reads) to six LDS reads.
// MAG = the substeps are Magnus substeps (qr_dynamics.h: integrate_magnus; default layout, `substeps` >= 2 — the host's choice from
// the substep count alone, pick_instance); MAG = false kernels hold RK4 only and run the one-substep launches.
template <int KIND, typename XV, typename QW, int B, int TRAJ, bool ADAPT, int POLICY = 0, bool SINGLE = false, bool HELP = false, bool HREW = true,
          bool MAG = false>
__global__ __launch_bounds__(B + (HELP ? 64 : 0), ((HELP && POLICY) ? 2 : (TRAJ || POLICY) ? 1 : 2))  // (HELP: both waves of every tile resident)
void step_kernel(void* pos_vel, void* att_rate, const float* action, float* params, float* integ, int32_t* reset_count,
                 int32_t n_envs, int32_t ld_envs, const Args a_in) {
  // The leading scalar arguments duplicate the fields of Args that the wave's loads depend on: as
  // plain kernel arguments they are preloaded into SGPRs by the dispatcher (gfx950 kernarg
  // preload, -mllvm -amdgpu-kernarg-preload-count=16: 14 dwords is what the hardware hands over), so
  // every load of the working set — and the scalar load of the tile's reset counter — is issued in
  // the wave's first instructions, without waiting for a scalar-load round trip to the kernarg
  // segment (host-visible memory: ~0.7 us, measured with in-kernel clock stamps).
  // Everything else is read from the kernarg segment WHERE IT IS USED: referenced as a by-value
  // struct, every used field of Args would be loaded in the kernel's entry block (that is how the
  // AMDGPU backend lowers kernel arguments) and stay live in SGPRs from there on — far more than the
  // 102 a wave has, so the round-1 kernel spilled them into VGPR lanes (v_writelane / v_readlane, a
  // VALU slot each, ~220 on the step's path).  Through the segment pointer they are ordinary
  // scalar loads from constant memory with short live ranges.
#if defined(__HIP_DEVICE_COMPILE__)
  const Args& ka = *reinterpret_cast<const Args*>(reinterpret_cast<const char*>(__builtin_amdgcn_kernarg_segment_ptr()) + kArgsOffset);
  (void)a_in;
#else
  const Args& ka = a_in;  // (host pass of the single-source compile: never executed)
#endif
  QR_SPAN_BEGIN();
  Args a;  // the fields the helpers touch, assembled from the preloaded scalars
  a.pos_vel = pos_vel; a.att_rate = att_rate; a.action = action; a.params = params; a.integ = integ;
  a.reset_count = reset_count;
  a.n = n_envs; a.ld = ld_envs;
  using T = QW;  // q, W are held and accumulated in their storage type
  using X = XV;  // and so are x, v
  using KT = KindTraits<KIND>;
  constexpr int A = KT::A, D0 = KT::D0, D1 = KT::D1 ? KT::D1 : 1, NAG = KT::NAG;
  constexpr int AUX = (HELP && SINGLE) ? QR_HELP_AUX : QR_PLAIN_AUX;  // cache policy of every store of this launch (qr_args.h)
  __shared__ __attribute__((aligned(16))) float smem[B * (D0 > A ? D0 : A)];
  const int tid = threadIdx.x;
  const unsigned lane = threadIdx.x;
  // XCD-aware tile map.  Workgroups are dealt round-robin over the 8 XCDs (workgroup b runs on XCD b % 8).  With tile = blockIdx.x
  // an XCD therefore touches every EIGHTH 256- / 512-byte segment of each SoA field — its requests alias onto a few of its L2's
  // channels.  For the grids whose working set is cache-resident (the one-step helper-wave launches up to QR_XCD_GRID tiles) every XCD
  // gets a CONTIGUOUS range of tiles instead (a bijection for any tile count: XCD x owns q + (x < r) tiles, q = tiles / 8, r = tiles % 8):
  // 16 384 ... 81 920 envs 0.5-4 % faster for all three kinds (Quad-v0 65 536: 4.15 -> 4.07 us, Coupled 5.34 -> 5.18, Decoupled 5.17 ->
  // 4.98; profiles/r05/ab_xcd_map.txt), nothing at <= 8192 envs.  Larger grids stream from HBM, where the default deal keeps the eight
  // XCDs inside the same DRAM pages: kept there (1 M envs: +1.7 % with contiguous ranges).  Which workgroup steps which tile changes no
  // result bit (tools/ab_equal.py: identical).
  unsigned tile_id = blockIdx.x;
  if constexpr (HELP && SINGLE) {
    const unsigned n_tiles = ((unsigned)n_envs + 63u) >> 6;
    if (n_tiles <= (unsigned)QR_XCD_GRID) {
      const unsigned xcd = blockIdx.x & 7u, q8 = n_tiles >> 3, r8 = n_tiles & 7u;
      tile_id = xcd * q8 + (xcd < r8 ? xcd : r8) + (blockIdx.x >> 3);
    }
  }
  if constexpr (HELP && !SINGLE) tile_id += (unsigned)ka.tile_base;  // (a chunk of a larger grid: launch_kind)
  const unsigned ufirst = tile_id * (unsigned)B;
  const int64_t first = (int64_t)ufirst;
  const int64_t i = first + tid;
  QR_HWID();
  const int64_t N = a.n, L = a.ld;
  const int rows = min(n_envs - (int)ufirst, B);   // (n_envs < 2^31: checked on the host)
  const bool active = tid < rows;
  // lanes past a ragged tail read the tail's last env (valid memory, finite numbers) and store nothing
  const unsigned ll = min(lane, (unsigned)(rows - 1));
  // (Measured and NOT adopted, profiles/r03/ab_dev_coeffs.txt: the coefficient block in a device-resident global instead of
  // the kernarg segment — 4.63 against 4.16 us per launch at 65 536 envs.)
  const Coeffs& c = ka.c;
#if QR_ABLATE == 1  // measurement build: launch floor only
  return;
#endif
  static_assert(!HELP || B == 64, "the helper wave belongs to the one-wave-per-tile kernels");
  // (a rollout alternates between two pools: the helper samples step t+1's while the stepping wave takes from step t's)
  __shared__ typename std::conditional<HELP, PoolLds<T>, char>::type pool_lds[(SINGLE || POLICY) ? 1 : 2];  // (unused without HELP: dropped)
  // (POLICY with a helper wave) the step's exploration noise, sampled a step ahead by the helper: [t & 1][lane][8]
  __shared__ __attribute__((aligned(16))) float eps_lds[HELP && POLICY ? 2 * 64 * 8 : 4];
  // Quad-v0's reward (an atan2, a sqrt: ~90 instructions) is formed by the helper wave as well
  // (HREW = false: the one-step Quad-v0 launch on grids where some SIMDs hold a second stepping wave — launch_kind)
  constexpr bool kHelpReward = HELP && !POLICY && !TRAJ && KIND == QR_KIND_QUAD && QR_HELP_REWARD && HREW;  // (TRAJ: the goal lives in the stepping wave's registers)
  __shared__ typename std::conditional<kHelpReward, PostLds<T, X>, char>::type post_lds[SINGLE ? 1 : 2];  // (a rollout alternates)
  __shared__ PoolLds<T> own_pool;  // pools this wave samples itself (no helper; or a tile's 13th.. resetting lane)
  // (HREW = false for a wrapper: the rows stay with the stepping wave — one-step grids beyond QR_HELP_ROWS_TILES tiles, launch_kind)
  constexpr bool kHelpRows = HELP && SINGLE && (KIND == QR_KIND_QUAD || HREW);
  // (plain one-step wrapper kernels: large grids) the observation rows go to their LDS tile as soon as they are formed,
  // BEFORE the reset block, and a re-sampled env overwrites its row there: the 18-23 row registers need not survive the
  // reset block.  With the late loads below: Coupled 150 -> 114 VGPRs, Decoupled 148 -> 115, i.e. four waves per SIMD
  // (262 144 envs 19.1 -> 16.5 us).  Not in the helper-wave launches, where the second write of a re-sampled env's row
  // is on the stepping wave's path (65 536 envs: 6.03 -> 6.15 us with it).
  constexpr bool kEarlyTile = SINGLE && !HELP && !POLICY && KIND != QR_KIND_QUAD;
  __shared__ __attribute__((aligned(16))) float smem1[(kHelpRows || kEarlyTile || (HELP && POLICY)) && KT::D1 > 0 ? B * D1 : 4];  // (Decoupled: both tiles at once)
  // (rollouts of the wrappers with a helper wave) the observation rows of step t go to tile t & 1 at the end of the step and the
  // helper carries them out behind the next step's pool barrier: 91 vector instructions and 24 stores per env-step off the
  // stepping wave (65 536 envs, T = 100: Coupled and Decoupled 2.14 -> 1.89 us per env-step; identical bits; profiles/r05/ab_roll_rows.txt)
  constexpr bool kRollRows = HELP && !SINGLE && !POLICY && KIND != QR_KIND_QUAD;
  __shared__ __attribute__((aligned(16))) float rtile0[kRollRows ? 2 * B * D0 : 4];
  __shared__ __attribute__((aligned(16))) float rtile1[kRollRows && KT::D1 > 0 ? 2 * B * D1 : 4];
  if constexpr (HELP) {
    // (the wave's first lane decides: a wave-uniform branch in the compiler's eyes too — on threadIdx.x itself everything
    // after it counts as divergent control flow, and scalar offsets of the loads below were re-derived per lane)
    if (__builtin_amdgcn_readfirstlane((int)threadIdx.x) >= B) {  // ---- the helper wavefront: pass 0 of the tile's reset pool -> LDS ----  //@sec helper-wave
      QR_HSTAMP(0, threadIdx.x);
      float hgoal[12];
#pragma unroll
      for (int f = 0; f < 12; ++f) hgoal[f] = f == 6 ? 1.0f : 0.0f;  // hover default (quad.py:98-101)
      if constexpr (kHelpReward) {
        if (float* const gp = ka.goal) {
          const SoA<float> goal(gp, 12, ld_envs);
          const unsigned hll = min(threadIdx.x - B, (unsigned)(rows - 1));
#pragma unroll
          for (int f = 0; f < 9; ++f) hgoal[f] = goal.load(f, ufirst, hll);
        }
      }
      // the helper's own output pointers, read with its first scalar loads: behind the barriers they would be a kernarg
      // cache miss at the very end of the launch
      float* const hob0 = ka.obs0;
      float* const hob1 = KT::D1 > 0 ? ka.obs1 : nullptr;
      float* const hrew = ka.reward;
      float* const hraw = ka.reward_raw;
      asm volatile("" ::"s"(hob0), "s"(hob1), "s"(hrew), "s"(hraw));
      const uint32_t rc = (uint32_t)reset_count[tile_id];
      const uint32_t hflags = ka.flags;
      const uint64_t hseed = ka.seed;
      const uint64_t hgfirst = (uint64_t)(ka.env_offset + first);
      QR_HSTAMP(5, (float)hflags + (float)hseed + (float)hgfirst);   // (diagnostic builds) kernarg scalars back
      QR_HSTAMP(6, (float)rc);                                       // the tile counter (global memory) back
      // (Measured and NOT adopted, profiles/r03/ab_helper_touch.txt: requesting this wave's kernarg lines with dummy loads in its
      // first instructions, like the stepping wave does — 4.148 against 4.151 us per launch; the pool is in LDS ~0.7 us before
      // the stepping wave asks for it either way.)
      const bool heval = (hflags & QR_FLAG_EVAL_RESET) != 0;
      PoolRole hrole;
      pool_role(hrole, !heval && !(hflags & QR_FLAG_NO_UDM) && params != nullptr, heval, c);
      // the reward of env-step t from the post-step state the stepping wave left in LDS (kHelpReward)
      auto help_reward = [&](int t) {
        if constexpr (kHelpReward) {
          const unsigned hl = threadIdx.x - B;
          const auto& ps = post_lds[SINGLE ? 0 : (t & 1)];
          X hx[3], hv[3];
          T hq[4], hW[3];
#pragma unroll
          for (int j = 0; j < 3; ++j) { hx[j] = ps.x[j][hl]; hv[j] = ps.v[j][hl]; hW[j] = ps.W[j][hl]; }
#pragma unroll
          for (int j = 0; j < 4; ++j) hq[j] = ps.q[j][hl];
          const float r = quad_reward_raw<T, X>(hx, hv, hq, hW, hgoal, c);
          // (rollouts: the stepping wave's own flag — the helper there is about as long as the stepping wave, 1.297 -> 1.283 us per
          //  env-step without the second quad_done; one-step launches: formed here, 4.13 against 4.16 us with the LDS word)
          const bool d = SINGLE ? quad_done<T, X>(hx, hv, hq, hW, c) : (ps.done[hl] != 0u);
          if ((int)hl < rows) {
            const int64_t hrow = (int64_t)t * n_envs + first;
            gstore<AUX>(hrew + hrow + hl, d ? -1.0f : interp01(r, c.rmin_mono, c.inv_nrmin_mono));  // crash override (quad.py:162-166)
            if (hraw) gstore<AUX>(hraw + hrow + hl, r);
          }
        }
      };
      ResetPool<T> hp;
      if constexpr (POLICY != 0) {
        // qr_rollout_actor.  Per env-step t the helper meets the stepping wave twice: B1(t), at the top of the step — the
        // step's noise is in LDS and the tile holds the observation rows of step t-1 (which this wave then carries out while
        // the stepping wave evaluates the actor) — and B2(t), when the step's reset pool and the NEXT step's noise are in LDS.
        const int hsteps = ka.n_steps;
        const int hl = (int)threadIdx.x - B;
        const bool own_noise = !ka.deterministic && ka.noise == nullptr;
        const uint64_t nseed = ka.noise_seed, sbase = ka.step_base, hgid = (uint64_t)(ka.env_offset + first + hl);
        float* const ob0 = hob0;
        float* const ob1 = hob1;
        auto make_eps = [&](int t) {
          if (!own_noise) return;
          float z[4];
          normal4(z, nseed, hgid, sbase + (uint64_t)t, 0u);
          float* e = eps_lds + ((t & 1) * 64 + hl) * 8;
          *reinterpret_cast<float4*>(e) = make_float4(z[0], z[1], z[2], z[3]);
          if constexpr (A > 4) {
            normal4(z, nseed, hgid, sbase + (uint64_t)t, 1u);
            e[4] = z[0];
          }
        };
        make_eps(0);
        for (int t = 0; t < hsteps; ++t) {
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // B1(t)
          if (t > 0) {
            lds_to_rows<B, D0, AUX>(ob0 + ((int64_t)(t - 1) * n_envs + first) * D0, smem, hl, rows);
            if constexpr (KT::D1 > 0) lds_to_rows<B, D1, AUX>(ob1 + ((int64_t)(t - 1) * n_envs + first) * D1, smem1, hl, rows);
          }
          make_pool<T>(hp, hrole, hseed, hgfirst, rc + (uint32_t)t, 0);
          pool_to_lds(pool_lds[0], hp);
          if (t + 1 < hsteps) make_eps(t + 1);
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // B2(t)
        }
        asm volatile("s_barrier" ::: "memory");  // the tile of the last step
        lds_to_rows<B, D0, AUX>(ob0 + ((int64_t)(hsteps - 1) * n_envs + first) * D0, smem, hl, rows);
        if constexpr (KT::D1 > 0) lds_to_rows<B, D1, AUX>(ob1 + ((int64_t)(hsteps - 1) * n_envs + first) * D1, smem1, hl, rows);
        QR_SPAN_END();
        return;
      }
      if constexpr (!SINGLE) {  // a rollout: one pool per env-step, each handed over at that step's barrier
        const int hsteps = ka.n_steps;
        for (int t = 0; t < hsteps; ++t) {
          make_pool<T>(hp, hrole, hseed, hgfirst, rc + (uint32_t)t, 0);
          pool_to_lds(pool_lds[t & 1], hp);
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          help_reward(t);  // (Quad-v0) this step's reward, before the next step's pool
          if constexpr (kRollRows) {  // (wrappers) the rows of step t - 1: complete in their tile since the end of that step
            if (t > 0) {
              const int hl = (int)threadIdx.x - B;
              lds_to_rows<B, D0, AUX>(hob0 + ((int64_t)(t - 1) * n_envs + first) * D0, rtile0 + ((t - 1) & 1) * (B * D0), hl, rows);
              if constexpr (KT::D1 > 0) lds_to_rows<B, D1, AUX>(hob1 + ((int64_t)(t - 1) * n_envs + first) * D1, rtile1 + ((t - 1) & 1) * (B * D1), hl, rows);
            }
          }
        }
        if constexpr (kRollRows) {  // the last step's rows
          asm volatile("s_barrier" ::: "memory");
          const int hl = (int)threadIdx.x - B;
          lds_to_rows<B, D0, AUX>(hob0 + ((int64_t)(hsteps - 1) * n_envs + first) * D0, rtile0 + ((hsteps - 1) & 1) * (B * D0), hl, rows);
          if constexpr (KT::D1 > 0) lds_to_rows<B, D1, AUX>(hob1 + ((int64_t)(hsteps - 1) * n_envs + first) * D1, rtile1 + ((hsteps - 1) & 1) * (B * D1), hl, rows);
        }
        QR_SPAN_END();
        return;
      }
      QR_HSTAMP(1, hrole.off[0] + (float)rc);
      make_pool<T>(hp, hrole, hseed, hgfirst, rc, 0);
      pool_to_lds(pool_lds[0], hp);
      QR_HSTAMP(2, hp.v[0] + (float)hp.q[0]);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      QR_HSTAMP(3, threadIdx.x);
      if constexpr (kHelpReward) help_reward(0);
      QR_HSTAMP(4, threadIdx.x);  // ---- then the reward of the step, from the post-step state the stepping wave left in LDS ----
      if constexpr (kHelpRows) {  // ---- and the observation rows: the stepping wave leaves the tile in LDS, this wave carries it out ----
        if (KIND != QR_KIND_QUAD || hob0 != nullptr) {
          asm volatile("s_barrier" ::: "memory");
          const int hl = (int)threadIdx.x - B;
          lds_to_rows<B, D0, AUX>(hob0 + first * D0, smem, hl, rows);
          if constexpr (KT::D1 > 0) lds_to_rows<B, D1, AUX>(hob1 + first * D1, smem1, hl, rows);
        }
      }
      QR_SPAN_END();
      return;
    }
  }
  // Issue priority of the stepping wave over the helper wave it shares a SIMD with (and over its own helper in the CU's shared
  // front end): the helper then runs in the slots the stepping wave leaves empty instead of taking every other one.  Multi-step
  // launches: for the whole launch — the helper's work per step is a third of the stepping wave's and is asked for a step later
  // (65 536 envs, per env-step: Quad-v0 rollout 1.44 -> 1.29 us, Coupled 2.25 -> 2.12, PPO collection step 3.55 -> 3.17).
  // One-step launches: the helper's pool is wanted within the same microsecond and its reward / rows trail the launch, so only
  // up to the pool barrier and only on grids of at most 768 tiles (32 768 envs: wrappers -3.7 %, Quad-v0 -1.5 %; 65 536 envs
  // +0.3...3 %, 98 304 +9 % with it).  Changes no result.  profiles/r05/ab_step_prio.txt
  if constexpr (HELP && !SINGLE) __builtin_amdgcn_s_setprio(QR_STEP_PRIO);
  if constexpr (HELP && SINGLE) {
    if ((((unsigned)n_envs + 63u) >> 6) <= (unsigned)QR_PRIO_SINGLE_TILES) __builtin_amdgcn_s_setprio(QR_STEP_PRIO);
  }
  QR_STAMP(0, tid);  //@sec prologue-loads
#if defined(__HIP_DEVICE_COMPILE__)
  // The coefficient block spans five 64-byte lines of the kernarg segment (host-visible memory: ~0.5 us per miss).  The
  // compiler reads coefficients where they are used, i.e. it requests those lines only AFTER the first batch of scalar
  // loads is back, and the first arithmetic then waits for them.  One dummy word per line, requested with the wave's
  // first instructions, has them in the scalar cache by then.  (The words are never used; their registers stay
  // reserved until a point behind the first scalar wait, see below.)
  uint32_t ctouch[6];  // ([5]: the line of the per-call integers — substeps is wanted at the first RK4 stage)
  {
    constexpr int kC = kArgsOffset + (int)offsetof(Args, c);
    asm volatile("s_load_dword %0, %6, %7\n\ts_load_dword %1, %6, %8\n\ts_load_dword %2, %6, %9\n\t"
                 "s_load_dword %3, %6, %10\n\ts_load_dword %4, %6, %11\n\ts_load_dword %5, %6, %12"
                 : "=&s"(ctouch[0]), "=&s"(ctouch[1]), "=&s"(ctouch[2]), "=&s"(ctouch[3]), "=&s"(ctouch[4]), "=&s"(ctouch[5])
                 : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "i"(kC), "i"(kC + 64), "i"(kC + 128), "i"(kC + 192), "i"(kC + 256),
                   "i"(kArgsOffset + (int)offsetof(Args, substeps)));
  }
#endif

  // ---- issue the loads of the env's working set (SoA, lane-contiguous) and of its action row ----
  Work<T, X> w;
  float act_next[A];
#pragma unroll
  for (int j = 0; j < A; ++j) act_next[j] = 0.f;
  // Action rows [N][A] -> lane registers.  A = 4: one 16-byte load per lane.  A = 5: five dword
  // loads per lane (a wave covers 1280 contiguous bytes; L1 merges the sectors).  In a rollout
  // the row of step t+1 is requested before the arithmetic of step t, so its latency is hidden.
  auto load_action = [&](int t, float (&dst)[A]) {
    const float* abase = a.action + ((int64_t)t * N + first) * A;
    if constexpr (A == 4) {
      const float4 v = reinterpret_cast<const float4*>(abase)[ll];
      dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    } else {
#pragma unroll
      for (int j = 0; j < A; ++j) dst[j] = abase[ll * A + j];
    }
  };
  const bool auto_reset = HELP || reset_count != nullptr;  // passed only with QR_FLAG_AUTO_RESET: its presence IS the flag, known without a load
  PoolRole role;
  uint32_t rcount_s = 0;  // the tile's position in the in-launch reset stream
  // (Measured and NOT adopted, profiles/r03/ab_load_order_store_policy.txt: requesting parameters and action row first, x and v
  // last, and rebuilding the quaternion behind the action map — 4.47 against 4.46 us per launch at 65 536 envs.)
  load_state<XV, QW>(a, first, ll, w);
  w.nominal = a.params == nullptr;
  {  // (without a params buffer: a descriptor without records, the loads return 0 — no branch between the load batches)
    const SoA<float> prm(a.params, 6, L);
#pragma unroll
    for (int f = 0; f < 6; ++f) w.prm[f] = prm.load(f, ufirst, ll);
  }
  if constexpr (!POLICY) load_action(0, act_next);
  // The tile's position in the in-launch reset stream: a scalar load.  (Measured alternatives, bench.py at 65 536 envs:
  // a vector load at the end of the wave's load queue 5.71 us, at its front 5.70 us, a load deferred until the working set
  // has been consumed 6.12 us — against 5.42 us, although scalar loads return out of order and the first use of a kernarg
  // coefficient therefore also waits for this one: the in-kernel timelines of those variants are shorter, their launches not.)
  if (!HELP && auto_reset) rcount_s = (uint32_t)reset_count[tile_id];
  // (kLateLoads) The plain one-step wrapper kernel serves grids of several waves per SIMD, where a load's latency is
  // other waves' time: it requests the 20 words only the error observation wants (goal, integrators) AFTER the
  // integration instead of holding them across it — registers for occupancy.
  constexpr bool kLateLoads = SINGLE && !HELP && !ADAPT && !TRAJ && !POLICY && KIND != QR_KIND_QUAD;
  if constexpr (KIND != QR_KIND_QUAD && !kLateLoads) {
    const SoA<float> integ(a.integ, 8, L);
#pragma unroll
    for (int f = 0; f < 8; ++f) w.integ[f] = integ.load(f, ufirst, ll);
  } else {
#pragma unroll
    for (int f = 0; f < 8; ++f) w.integ[f] = 0.0f;
  }
  // ---- the rest of the arguments: one batch of scalar loads from the kernarg segment ----  //@sec prologue-args
  const uint32_t flags = ka.flags;
  const uint64_t seed = ka.seed;
  const uint64_t gfirst = (uint64_t)(ka.env_offset + first);
  float* const goal_ptr = ka.goal;
  int32_t* const steps_ptr = ka.steps;
  const int n_steps = SINGLE ? 1 : ka.n_steps;
  const bool eval_reset = (flags & QR_FLAG_EVAL_RESET) != 0;
  const bool randomise = !eval_reset && !(flags & QR_FLAG_NO_UDM) && a.params != nullptr;
  // One-step launches form the role constants in the reset block (below); a rollout forms them once, here.
  constexpr bool kLazyRole = SINGLE;
  if (!kLazyRole && !HELP && auto_reset) pool_role(role, randomise, eval_reset, c);  // (scalars only: runs while the loads are in flight)
#pragma unroll
  for (int f = 0; f < 12; ++f) w.goal[f] = f == 6 ? 1.0f : 0.0f;  // hover default (quad.py:98-101)
  if (!TRAJ && !kLateLoads && goal_ptr) {  // (with the fused generator the goal is formed in registers every step)
    const SoA<float> goal(goal_ptr, 12, L);
#pragma unroll
    for (int f = 0; f < 12; ++f) w.goal[f] = goal.load(f, ufirst, ll);
  }

  // ---- in-launch reset: this wave's pool of episode starts (qr_rng.h), sampled while the loads are in flight ----
  // (Measured and NOT adopted, profiles/r02/ab_quad_builds.txt, columns q_spec / q_nohelp: sampling the pool speculatively in the
  // same wave right after issuing its loads — 5.86-5.92 against 5.28 us per launch at 65 536 envs: the pool's inputs arrive only
  // ~0.5 us after the wave's first instruction, so most of its ~0.6 us does not hide under the loads and EVERY wave pays it.)
  ResetPool<T> pool;
  QR_STAMP(1, 0.0f);

  Traj tr;
  const int goal_mode = TRAJ ? ka.goal_mode : QR_GOAL_EXTERNAL;  // wave-uniform
  constexpr bool kStateful = TRAJ == 2;
  if constexpr (TRAJ) {
    const SoA<float> traj(ka.traj, 8, L);
#pragma unroll
    for (int f = 0; f < (kStateful ? 8 : 7); ++f) tr.set(f, traj.load(f, ufirst, ll));
    if constexpr (kStateful) {  // xd, vd, b1d, Wd persist in the goal buffer (required for these modes)
      const SoA<float> goal(goal_ptr, 12, L);
#pragma unroll
      for (int f = 0; f < 12; ++f) w.goal[f] = goal.load(f, ufirst, ll);
    }
  }
#if defined(__HIP_DEVICE_COMPILE__)
  // (steps_ptr is back => s_waitcnt lgkmcnt(0) has been passed => the dummy words have landed: their registers are free)
  asm volatile("" ::"s"(ctouch[0]), "s"(ctouch[1]), "s"(ctouch[2]), "s"(ctouch[3]), "s"(ctouch[4]), "s"(ctouch[5]), "s"(steps_ptr));
#endif
  int32_t steps = (steps_ptr && active) ? (steps_ptr + first)[lane] : 0;
  bool params_dirty = false;
  bool traj_dirty = false;  // this lane started a new episode: its generator state changed
  bool stored_early = false;  // (SINGLE) this lane's state went out before its wave sampled a reset pool
  // (n_envs: a preloaded SGPR — gridDim.x would be a scalar load.  With a helper wave the reset block is six LDS reads:
  // nothing to overlap.)
  const bool early_store = SINGLE && !HELP && n_envs <= QR_EARLY_STORE_GRID * 64;
  QuatPack<T> qp;             // attitude in its storage form, formed once per env-step
  qp.k[0] = qp.k[1] = qp.k[2] = T(0);

  //@sec prologue-policy
  // POLICY: the observation the next action is computed from (rows -> lane registers once, then
  // carried from step to step)
  float po0[D0], po1[D1];
  // agent 0 (23 / 15 -> 16 -> 16 -> 4, args_parse.py:40, main.py:68-73) on the matrix cores, weights
  // resident in registers; agent 1 of DECOUPLED (3 -> 4 -> 4 -> 1: 32 FMAs) per lane from LDS
  constexpr bool GENERAL = POLICY == 2;
  ActorMfma<D0, GENERAL> actor0;
  using Actor1 = ActorLds<3, 4, 1>;
  __shared__ __attribute__((aligned(16))) float wsm[POLICY ? Actor1::SIZE : 4];
  if constexpr (POLICY) {
    load_rows<B, D0>(ka.obs0_in + first * D0, po0, smem, tid, rows);
    if constexpr (KT::D1 > 0) load_rows<B, D1>(ka.obs1_in + first * D1, po1, smem, tid, rows);
    actor0.load(ka.actor[0], tid);
    if constexpr (KT::D1 > 0) Actor1::fill(wsm, ka.actor[1], tid);
    // (HELP) the tile always holds the current observation rows: written here and at the end of every step, read by the
    // first layer's MFMAs and — as the rows of the step that produced them — carried out by the helper wave
    if constexpr (HELP) rows_to_lds<D0>(po0, smem, tid);
    tile_sync<B>();
  }

  // What the action map needs of the PARAMETERS only (masses, inertias, their reciprocals: qr_dynamics.h, ActConsts) changes only
  // when an env is re-sampled: a rollout forms it here and again behind a reset, not in every env-step.  (Not with the policy in
  // the loop: those kernels are at their register limit; not for Decoupled, whose rollout kernel is 1 % SLOWER with the 22 more
  // registers held across the loop.  Measured, profiles/r04/ab_hoist_act.txt: Quad-v0 65 536 envs 1.492 -> 1.469 us per env-step,
  // 262 144 envs 4.515 -> 4.354 = 60.2 G env-steps/s; Coupled 2.321 -> 2.289.)
  constexpr bool kHoistAct = !SINGLE && !POLICY && KIND != QR_KIND_DECOUPLED;
  // (Measured and NOT adopted, profiles/r05/ab_rollout_diet.txt: the same constants parked in LDS by the kernels that cannot afford the
  // registers — bit-identical, qr_rollout_actor 0.3-1.5 % and the Decoupled rollout 5 % SLOWER: eight ds_read_b64 on a lone wave's
  // critical path cost more than the ~30 VALU instructions they replace.)
  ActConsts<T> ac;
  if constexpr (kHoistAct) act_consts(w, c, ac);

  for (int t = 0; t < n_steps; ++t) {  //@sec action-source
    float act[A];
    if constexpr (!SINGLE) QR_PSTAMP(0, tid);
    if constexpr (POLICY) {
      float pre[A], ls[A], eps[A], logp[A];
      // the wave's observation rows -> LDS tile [lane][D0] (B operands of the first layer)
      if constexpr (!HELP) {
#pragma unroll
        for (int j = 0; j < D0; ++j) smem[tid * D0 + j] = po0[j];
        tile_sync<B>();
      }
      // B1(t), BEFORE the heads: the tile already holds the rows of step t - 1 (written at the end of that step) and the step's
      // noise was sampled before B2(t - 1), so the helper carries the rows out and samples the step's pool and the next step's
      // noise while this wave waits on the matrix pipe — in slots that are empty.  (Behind the heads, as before round 5, the
      // helper's work was the critical section between B1 and B2: Decoupled PPO collection 4.16 -> 3.43 us per env-step, SAC forms
      // -5...7 %, Coupled PPO -1.4 %; identical bits.  profiles/r05/ab_b1_early.txt)
      if constexpr (HELP) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      {
        float p0[4], l0[4];
        actor0.heads(smem, tid, p0, l0);
#pragma unroll
        for (int j = 0; j < 4; ++j) { pre[j] = p0[j]; ls[j] = l0[j]; }
      }
      if constexpr (KIND == QR_KIND_DECOUPLED) {
        float p1[1], l1[1];
        Actor1::heads(wsm, GENERAL && ka.actor[1].ls_w != nullptr, po1, p1, l1);
        pre[A - 1] = p1[0]; ls[A - 1] = l1[0];
      }
#pragma unroll
      for (int j = 0; j < A; ++j) eps[j] = 0.0f;
      if constexpr (!SINGLE) QR_PSTAMP(1, pre[0] + pre[A - 1]);   // actor heads done
      if constexpr (!SINGLE) QR_PSTAMP(2, tid);                   // past the noise barrier
      if (!ka.deterministic) {
        if (ka.noise != nullptr) {  // injected draws [T][N][A]
          if (active) {
            const float* nbase = ka.noise + ((int64_t)t * N + first) * A;
#pragma unroll
            for (int j = 0; j < A; ++j) eps[j] = nbase[lane * A + j];
          }
        } else if constexpr (HELP) {  // sampled by the helper wave, one step ahead
          const float* e = eps_lds + ((t & 1) * 64 + tid) * 8;
          const float4 z4 = *reinterpret_cast<const float4*>(e);
          eps[0] = z4.x; eps[1] = z4.y; eps[2] = z4.z; eps[3] = z4.w;
          if constexpr (A > 4) eps[A - 1] = e[4];
        } else {
          float z[4];
          normal4(z, ka.noise_seed, (uint64_t)(ka.env_offset + i), ka.step_base + (uint64_t)t, 0u);
#pragma unroll
          for (int j = 0; j < 4; ++j) eps[j] = z[j];
          if constexpr (A > 4) {
            normal4(z, ka.noise_seed, (uint64_t)(ka.env_offset + i), ka.step_base + (uint64_t)t, 1u);
            eps[A - 1] = z[0];
          }
        }
      }
      if constexpr (!HELP) tile_sync<B>();  // the tile is reused by the row stores below
      actor_sample<4, GENERAL>(ka.actor[0].squash, actor0.ls_head, &pre[0], &ls[0], &eps[0], ka.deterministic != 0, ka.max_action, &act[0], &logp[0]);
      if constexpr (A > 4)
        actor_sample<1, GENERAL>(ka.actor[1].squash, ka.actor[1].ls_w != nullptr, &pre[A - 1], &ls[A - 1], &eps[A - 1], ka.deterministic != 0,
                        ka.max_action, &act[A - 1], &logp[A - 1]);
      if (active) {
        const int64_t arow = ((int64_t)t * N + first) * A;
        if constexpr (A == 4) {
          gstore<AUX>(reinterpret_cast<float4*>(ka.act_out + arow) + lane, make_float4(act[0], act[1], act[2], act[3]));
          if (ka.logp_out) gstore<AUX>(reinterpret_cast<float4*>(ka.logp_out + arow) + lane, make_float4(logp[0], logp[1], logp[2], logp[3]));
        } else {
#pragma unroll
          for (int j = 0; j < A; ++j) {
            gstore<AUX>(ka.act_out + arow + lane * A + j, act[j]);
            if (ka.logp_out) gstore<AUX>(ka.logp_out + arow + lane * A + j, logp[j]);
          }
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < A; ++j) act[j] = act_next[j];
      if (t + 1 < n_steps) load_action(t + 1, act_next);
    }

#if QR_ABLATE == 2  // measurement build: memory traffic only (no integration)
    w.x[0] += X(act[0]);
    pack_quat(w.q, qp);
    uint8_t* const done_ptr = ka.done;
    uint8_t* const trunc_ptr = ka.truncated;
#else
    // ---- goal for this step from the pre-step state (main.py:145-147) ----  //@sec traj-goal
    if constexpr (TRAJ) {
      float b1d_dot[3];
      traj_goal<kStateful>(w, tr, goal_mode, c, b1d_dot);
    }
    QR_STAMP(2, (float)w.q[0] + (float)w.x[0] + act[0] + w.prm[0] + (float)w.W[2]);
    if constexpr (!SINGLE) QR_PSTAMP(3, act[0] + act[A - 1]);     // action sampled (policy) / loaded
    // ---- action_wrapper ----  //@sec action-map
    Dyn<T> dyn;
    if constexpr (!kHoistAct) act_consts(w, c, ac);
    action_map<KIND, T, X>(act, w, ac, c, dyn);
    if constexpr (SINGLE && !HELP) {  // (plain one-step launch: the loaded parameters are dead from here on — a re-sampled
#pragma unroll                        //  env stores the ones it takes from the pool — so their registers need not survive)
      for (int f = 0; f < 6; ++f) w.prm[f] = 0.0f;
    }
    // The output pointers sit in a kernarg cache line that nothing before the epilogue touches.  In the plain
    // instantiation (large grids) they are read with the first batch of scalar loads, so that the scalar-cache miss does
    // not sit in front of the first output store with the wave's registers held meanwhile (1 M envs: Quad-v0 36.6 -> 34.0 us,
    // Coupled 62.8 -> 59.5).  In the helper-wave launches — one stepping wave per SIMD, every wait on its critical path —
    // that batch is the wave's first wait and the extra line lengthens it: there they are read where they are used
    // (65 536 envs: 4.41 us against 4.49 with the early read, 4.67 with a read pinned behind the action map).
    uint8_t* const done_ptr = ka.done;
    uint8_t* const trunc_ptr = ka.truncated;
    if constexpr (!HELP) asm volatile("" ::"s"(done_ptr), "s"(trunc_ptr));
    // ---- observation_wrapper: integrate over dt with zero-order-hold (f, M) ----  //@sec integrate
    // The reference's DOP853 is adaptive (6 % of its steps subdivide); the fixed-step stand-in
    // is made rate-adaptive: RK4's local error grows like (|W| h)^5, so a wave that contains an
    // env spinning faster than w_adapt takes ceil(max|W_i| / w_adapt) times the substeps.  The
    // multiplier is the wave's maximum (found with ballots, so the substep loop stays wave-uniform
    // and in regime — |W| < 2 pi < w_adapt — this costs one ballot): every lane takes at least
    // the count its own rate asks for.
    if constexpr (ADAPT) {
      // (kDelta) The launches that step envs on without in-launch resets — the only ones in which an env can leave the
      // regime — form the quaternion stages in delta form (qr_dynamics.h: integrate_delta): their free run lands on RK4's
      // truncation floor instead of 7x above it.  qr_rollout_actor keeps the plain stages (its kernel is at its register limit).
      constexpr bool kDelta = QR_DELTA_STAGES && !POLICY;
      const T wmax = fmax(fmax(fabs(w.W[0]), fabs(w.W[1])), fabs(w.W[2]));
      const T need = wmax * T(c.inv_w_adapt);
      if constexpr (kDelta) {
        int mul = 1;
        if (__ballot(need > T(1)) != 0) {  // (in regime: one ballot)
          mul = 2;
          while (mul < 16 && __ballot(need > T(mul))) ++mul;
        }
        const int nsub = ka.substeps * mul;
        integrate_delta(w.x, w.v, w.q, w.W, dyn, nsub, T(c.dt) * recip(T(nsub)));
      } else if (__ballot(need > T(1)) == 0) {  // in regime: exactly the plain kernel's code path (one ballot)
        const int nsub = ka.substeps;
        integrate_sel<MAG>(w.x, w.v, w.q, w.W, dyn, nsub, T(c.dt) * recip(T(nsub)));
      } else {
        int mul = 2;
        while (mul < 16 && __ballot(need > T(mul))) ++mul;
        const int nsub = ka.substeps * mul;
        integrate_sel<MAG>(w.x, w.v, w.q, w.W, dyn, nsub, T(c.dt) * recip(T(nsub)));
      }
    } else {
      const int nsub = ka.substeps;
      // (one-step helper launch with two or more substeps: priority for the chain from here to the pool barrier, on any grid —
      //  Quad-v0 131 072 envs x 10 substeps 9.66 -> 9.21 us, x 4: 7.04 -> 6.67, x 2: 5.97 -> 5.78; with ONE substep it loses
      //  at 65 536 envs (+1.2 %) and is left out: profiles/r05/ab_step_prio.txt)
      if constexpr (HELP && SINGLE) {
        if (nsub >= QR_PRIO_SUBSTEPS) __builtin_amdgcn_s_setprio(QR_STEP_PRIO);
      }
      integrate_sel<MAG>(w.x, w.v, w.q, w.W, dyn, nsub, T(c.dt) * recip(T(nsub)));
    }
    renorm_quat(w.q);  //@sec renorm-late-loads-pack
    if constexpr (kLateLoads) {
      const SoA<float> integ(a.integ, 8, L);
#pragma unroll
      for (int f = 0; f < 8; ++f) w.integ[f] = integ.load(f, ufirst, ll);
      if (goal_ptr) {
        const SoA<float> goal(goal_ptr, 12, L);
#pragma unroll
        for (int f = 0; f < 12; ++f) w.goal[f] = goal.load(f, ufirst, ll);
      }
    }
    // The attitude as it is stored (qr_traj.h: QuatPack) is formed once per env-step: here when this wave may store
    // its settled lanes early (below), otherwise after the reset block, when every lane holds what it will store.
    if (early_store) pack_quat(w.q, qp);
    QR_STAMP(3, (float)w.q[0] + (float)w.x[0] + (float)w.v[2] + (float)w.W[0]);
    if constexpr (!SINGLE) QR_PSTAMP(4, (float)w.q[0] + (float)w.x[0] + (float)w.v[2] + (float)w.W[0]);   // integrated
#endif

    // ---- obs / reward / done ----  //@sec obs-reward-done
    T R[9];
    float o0[D0];
    float o1[D1];
    float rraw[NAG], rwd[NAG];
    bool dn[NAG];
    if constexpr (KIND == QR_KIND_QUAD) {
      if constexpr (kHelpReward) {
        // the helper wave forms and stores the reward (below, after the barrier): hand it the post-step state
        auto& ps = post_lds[SINGLE ? 0 : (t & 1)];
#pragma unroll
        for (int j = 0; j < 3; ++j) { ps.x[j][lane] = w.x[j]; ps.v[j][lane] = w.v[j]; ps.W[j][lane] = w.W[j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) ps.q[j][lane] = w.q[j];
        dn[0] = quad_done<T, X>(w.x, w.v, w.q, w.W, c);  // (formed while the LDS writes land)
        if constexpr (!SINGLE) ps.done[lane] = dn[0] ? 1u : 0u;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if constexpr (SINGLE) __builtin_amdgcn_s_setprio(0);
        rraw[0] = rwd[0] = 0.0f;
      } else {
        const float r = quad_reward_raw<T, X>(w.x, w.v, w.q, w.W, w.goal, c);
        rraw[0] = r;
        rwd[0] = interp01(r, c.rmin_mono, c.inv_nrmin_mono);
        dn[0] = quad_done<T, X>(w.x, w.v, w.q, w.W, c);
      }
    } else {
      quat_to_R(w.q, R);
      error_obs<KIND, T, X>(w, R, c, o0, o1);
      if constexpr (KIND == QR_KIND_COUPLED) {  // coupled:78-110, float32 arithmetic on the float32 obs
        const float r = -c.Cx * sq3(&o0[0]) + -c.CIx * sq3(&o0[3]) + -c.Cv * sq3(&o0[6]) +
                        -c.Cb1 * fabsf(o0[18]) + -c.CIb1 * (o0[19] * o0[19]) + -c.CW * sq3(&o0[20]);
        rraw[0] = r;
        rwd[0] = interp01(r, c.rmin_mono, c.inv_nrmin_mono);
        dn[0] = out3(&o0[0]) | out3(&o0[6]) | out3(&o0[20]);
      } else {  // decoupled:92-140
        const float r1 = -c.Cx * sq3(&o0[0]) + -c.CIx * sq3(&o0[3]) + -c.Cv * sq3(&o0[6]) + -c.Cw12 * sq3(&o0[12]);
        const float r2 = -c.Cb1 * fabsf(o1[0]) + -c.CIb1 * (o1[1] * o1[1]) + -c.CW3 * (o1[2] * o1[2]);
        rraw[0] = r1; rraw[NAG - 1] = r2;
        rwd[0] = interp01(r1, c.rmin_1, c.inv_nrmin_1); rwd[NAG - 1] = interp01(r2, c.rmin_2, c.inv_nrmin_2);
        dn[0] = out3(&o0[0]) | out3(&o0[6]) | out3(&o0[12]);
        dn[NAG - 1] = !(fabsf(o1[2]) < 1.0f);
      }
    }
    if constexpr (kEarlyTile) {
      rows_to_lds<D0>(o0, smem, tid);
      if constexpr (KT::D1 > 0) rows_to_lds<D1>(o1, smem1, tid);
    }
    // crash override (quad.py:162-166)
#pragma unroll
    for (int g = 0; g < NAG; ++g)
      if (dn[g]) rwd[g] = -1.0f;

    QR_STAMP(4, rwd[0] + (dn[0] ? 1.0f : 0.0f));
    if constexpr (!SINGLE) QR_PSTAMP(5, rwd[0] + (dn[0] ? 1.0f : 0.0f));   // observation, reward, done formed
    // ---- time limit + auto-reset ----  //@sec reward-done-stores
    steps += 1;
    const bool trunc = ka.max_episode_steps > 0 && steps >= ka.max_episode_steps;
    bool any_done = trunc;
#pragma unroll
    for (int g = 0; g < NAG; ++g) any_done = any_done | dn[g];
    const bool need_reset = auto_reset && any_done && active;
    const int64_t row0 = (int64_t)t * N + first;
    // ---- reward / done of step t (they belong to the step that just ended, whatever the reset does next) ----
    if (active) {
      if constexpr (NAG == 1) {
        if constexpr (!kHelpReward) {
          gstore<AUX>(ka.reward + row0 + lane, rwd[0]);
          if (ka.reward_raw) gstore<AUX>(ka.reward_raw + row0 + lane, rraw[0]);
        }
        gstore<AUX>(done_ptr + row0 + lane, (uint8_t)(dn[0] ? 1 : 0));
      } else {
        gstore<AUX>(reinterpret_cast<float2*>(ka.reward) + row0 + lane, make_float2(rwd[0], rwd[NAG - 1]));
        if (ka.reward_raw) gstore<AUX>(reinterpret_cast<float2*>(ka.reward_raw) + row0 + lane, make_float2(rraw[0], rraw[NAG - 1]));
        gstore<AUX>(reinterpret_cast<uchar2*>(done_ptr) + row0 + lane, make_uchar2(dn[0] ? 1 : 0, dn[NAG - 1] ? 1 : 0));
      }
      if (trunc_ptr) gstore<AUX>(trunc_ptr + row0 + lane, (uint8_t)(trunc ? 1 : 0));
    }
    // (HELP) the helper wave's pool is in LDS: it got there while this wave waited for its loads.  A bare s_barrier:
    // nothing of this wave's own (its reward / done stores in flight) has to be waited for.
    if constexpr (kRollRows) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (+ the previous step's tile has landed)
    else if constexpr (HELP && !kHelpReward) asm volatile("s_barrier" ::: "memory");
    if constexpr (HELP && !kHelpReward && SINGLE) __builtin_amdgcn_s_setprio(0);
    const unsigned long long rmask = __ballot(need_reset);
    if constexpr (!SINGLE) QR_PSTAMP(6, tid);                     // stores issued, past the pool barrier
    if (rmask) {  // wave-uniform: skipped unless some lane of this wave starts a new episode  //@sec reset-block
      if (early_store) {
        // This wave is about to spend ~0.5 us sampling episode starts.  The state of its lanes that do NOT
        // reset is final: hand it to the memory system first, so that those stores drain meanwhile.  Only for
        // grids in the launch-latency regime: the resetting lanes' own stores then are partial-line writes, which
        // cost more than the overlap gains once the launch is bound by bytes (measured, bench.py: 65 536 envs 5.43
        // with / 5.49 us without; 1 M envs 39.4 with / 37.4 us without).
        if (active && !need_reset) {
          store_state<XV, QW, AUX>(a, first, lane, w, qp);
          if (KIND != QR_KIND_QUAD) {
            const SoA<float> integ(a.integ, 8, L);
#pragma unroll
            for (int f = 0; f < 8; ++f) integ.store<AUX>(f, ufirst, lane, w.integ[f]);
          }
        }
        stored_early = active && !need_reset;
      }
      // the terminal observation of the episode that ends here (what a learner bootstraps from):
      // written for the resetting lanes only
      if (need_reset && ka.final_obs0 != nullptr) {
        if constexpr (KIND == QR_KIND_QUAD) {
          T Rf[9];
          quat_to_R(w.q, Rf);
          float* fo = ka.final_obs0 + (row0 + lane) * D0;
#pragma unroll
          for (int j = 0; j < 3; ++j) { gstore<AUX>(fo + j, (float)w.x[j]); gstore<AUX>(fo + 3 + j, (float)w.v[j]); gstore<AUX>(fo + 15 + j, (float)w.W[j]); }
#pragma unroll
          for (int j = 0; j < 9; ++j) gstore<AUX>(fo + 6 + j, (float)Rf[j]);
        } else {
          float* fo = ka.final_obs0 + (row0 + lane) * D0;
#pragma unroll
          for (int j = 0; j < D0; ++j) gstore<AUX>(fo + j, kEarlyTile ? smem[tid * D0 + j] : o0[j]);
          if constexpr (KT::D1 > 0) {
            float* f1 = ka.final_obs1 + (row0 + lane) * D1;
#pragma unroll
            for (int j = 0; j < D1; ++j) gstore<AUX>(f1 + j, kEarlyTile ? smem1[tid * D1 + j] : o1[j]);
          }
        }
      }
      const int rank = __popcll(rmask & ((1ull << lane) - 1ull));  // rank among the wave's resetting lanes
      const int total = __popcll(rmask);
      uint32_t r19 = 0;
      int pass0 = 0;
      if constexpr (HELP) {  // pass 0 comes from the helper wave
        take_from_lds<T, X, TRAJ != 0>(pool_lds[(SINGLE || POLICY) ? 0 : (t & 1)], need_reset && rank < 12, rank, w, r19);
        pass0 = 1;
        if (total > 12) {  // more than 12 lanes reset at once (rare): this wave samples the further passes itself
          pool_role(role, randomise, eval_reset, c);
          rcount_s = (uint32_t)reset_count[tile_id];  // (still this launch's base: advanced only at the end)
        }
      }
      // (one-step launches) the lane's role constants are formed HERE, not while the loads are in flight: twelve values
      // held across the whole step cost the plain kernel its fourth wave per SIMD (142 -> 128 VGPRs), and the grids that
      // run it are either large (other waves cover this) or take the helper-wave instantiation.
      if (kLazyRole && !HELP) pool_role(role, randomise, eval_reset, c);
      for (int pass = pass0; 12 * pass < total; ++pass) {  // one pass unless more than 12 lanes reset at once
        const int slot = rank - 12 * pass;
#if defined(__HIP_DEVICE_COMPILE__)
        // (multi-step kernels) keep the Philox key schedule — 20 seed-derived words — out of the step loop's preamble: made opaque
        // HERE, the seed's derived values are formed where this rare path uses them instead of being hoisted out of the loop and
        // spilled into VGPR lanes that the loop then reads back (in-loop v_readlane: Quad-v0 rollout 26 -> 13, Coupled actor rollout
        // 14 -> 3; bit-identical, rollouts 1-2 % faster: profiles/r05/ab_local_keys.txt)
        uint64_t seed_here = seed, gfirst_here = gfirst;
        if constexpr (!SINGLE) asm volatile("" : "+s"(seed_here), "+s"(gfirst_here));
        make_pool<T>(pool, role, seed_here, gfirst_here, rcount_s + (uint32_t)t, pass);
#else
        make_pool<T>(pool, role, seed, gfirst, rcount_s + (uint32_t)t, pass);
#endif
        // through LDS (six 16-byte reads per taking lane) rather than 23 ds_bpermute with all their results in flight at
        // once: 128 instead of 142 VGPRs for the plain Quad-v0 kernel, i.e. four waves per SIMD instead of three
        pool_to_lds(own_pool, pool);
        tile_sync<B>();
        take_from_lds<T, X, TRAJ != 0>(own_pool, need_reset && slot >= 0 && slot < 12, slot, w, r19);
        tile_sync<B>();
      }
      if (need_reset) {
        // (with a params buffer the float32 words about to be stored are also what the following steps of a rollout use,
        // randomised or not: exactly what a one-step launch re-loads)
        w.nominal = a.params == nullptr;
        if (a.params != nullptr) params_dirty = true;
        // episode counter (stream id of qr_reset / qr_traj_start): fire-and-forget, nothing here waits for it
        __hip_atomic_fetch_add(ka.episode + i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        steps = 0;
        if constexpr (TRAJ) {  // mark_traj_start + first get_desired of the episode (main.py:227-229)
          float th, tt, wb, b1d_dot[3];
          traj_draws(r19, th, tt, wb);
          traj_start<kStateful>(w, tr, goal_mode, th, tt, wb);
          traj_goal<kStateful>(w, tr, goal_mode, c, b1d_dot);
          traj_dirty = true;
        }
        if constexpr (KIND != QR_KIND_QUAD) {
          quat_to_R(w.q, R);
#pragma unroll
          for (int f = 0; f < 8; ++f) w.integ[f] = 0.0f;
          error_obs<KIND, T, X>(w, R, c, o0, o1);  // first observation of the new episode (main.py:226-230)
          if constexpr (kEarlyTile) {
            rows_to_lds<D0>(o0, smem, tid);
            if constexpr (KT::D1 > 0) rows_to_lds<D1>(o1, smem1, tid);
          }
        }
        if (early_store) pack_quat(w.q, qp);
      }
      if constexpr (kHoistAct) act_consts(w, c, ac);  // (some lane of the wave holds new parameters: every lane re-forms — the same values for the others)
    }
    if (!early_store) pack_quat(w.q, qp);  //@sec pack-quat
    QR_STAMP(5, (float)w.q[0] + (float)w.x[0] + w.prm[0]);
#ifdef QR_STAMPS
    if (g_stamps != nullptr && lane == 0) g_stamps[(size_t)blockIdx.x * 8 + 7] = rmask;
#endif

    // ---- outputs of step t ----  //@sec obs-rows-out
    if constexpr (kHelpRows) {  // rows -> LDS tile(s); the helper wave stores them
      if (KIND != QR_KIND_QUAD || ka.obs0 != nullptr) {
        if constexpr (KIND == QR_KIND_QUAD) {  // next state in the reference's order (x, v, vec_F(R), W)
          quat_to_R(w.q, R);
#pragma unroll
          for (int j = 0; j < 3; ++j) { o0[j] = (float)w.x[j]; o0[3 + j] = (float)w.v[j]; o0[15 + j] = (float)w.W[j]; }
#pragma unroll
          for (int j = 0; j < 9; ++j) o0[6 + j] = (float)R[j];
        }
        if constexpr (!kEarlyTile) {
          rows_to_lds<D0>(o0, smem, tid);
          if constexpr (KT::D1 > 0) rows_to_lds<D1>(o1, smem1, tid);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
    } else if constexpr (kEarlyTile) {  // (plain one-step wrapper kernel) the tile(s) are complete: carry them out
      tile_sync<B>();
      lds_to_rows<B, D0, AUX>(ka.obs0 + row0 * D0, smem, tid, rows);
      if constexpr (KT::D1 > 0) lds_to_rows<B, D1, AUX>(ka.obs1 + row0 * D1, smem1, tid, rows);
    } else if constexpr (HELP && POLICY != 0) {  // the tile is next step's MFMA operand AND this step's rows (helper wave)
      rows_to_lds<D0>(o0, smem, tid);
      if constexpr (KT::D1 > 0) rows_to_lds<D1>(o1, smem1, tid);
      tile_sync<B>();
    } else if constexpr (kRollRows) {  // tile t & 1; the helper carries it out behind the next pool barrier
      rows_to_lds<D0>(o0, rtile0 + (t & 1) * (B * D0), tid);
      if constexpr (KT::D1 > 0) rows_to_lds<D1>(o1, rtile1 + (t & 1) * (B * D1), tid);
    } else {
    if constexpr (KIND == QR_KIND_QUAD) {
      if (ka.obs0 != nullptr) {  // next state in the reference's order (x, v, vec_F(R), W)
        quat_to_R(w.q, R);
#pragma unroll
        for (int j = 0; j < 3; ++j) { o0[j] = (float)w.x[j]; o0[3 + j] = (float)w.v[j]; o0[15 + j] = (float)w.W[j]; }
#pragma unroll
        for (int j = 0; j < 9; ++j) o0[6 + j] = (float)R[j];
        store_rows<B, D0, AUX>(ka.obs0 + row0 * D0, o0, smem, tid, rows);
      }
    } else {
      store_rows<B, D0, AUX>(ka.obs0 + row0 * D0, o0, smem, tid, rows);
    }
    if constexpr (KT::D1 > 0) store_rows<B, D1, AUX>(ka.obs1 + row0 * D1, o1, smem, tid, rows);
    }
    if constexpr (POLICY) {
#pragma unroll
      for (int j = 0; j < D0; ++j) po0[j] = o0[j];
#pragma unroll
      for (int j = 0; j < D1; ++j) po1[j] = o1[j];
    }
    if constexpr (!SINGLE) unpack_quat(qp, w.q);  //@sec unpack-quat  // the next env-step starts from what a single-step launch would have re-loaded
    if constexpr (!SINGLE) QR_PSTAMP(7, (float)w.q[0] + (rmask ? 1.0f : 0.0f));   // reset block, pack, rows handed over, unpack
  }

  if constexpr ((HELP && POLICY != 0) || kRollRows) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the last step's tile: see the helper wave
  // ---- write the working set back ----  //@sec epilogue-stores
  if (active) {
    if (!(SINGLE && stored_early)) {
      store_state<XV, QW, AUX>(a, first, lane, w, qp);
      if (KIND != QR_KIND_QUAD) {
        const SoA<float> integ(a.integ, 8, L);
#pragma unroll
        for (int f = 0; f < 8; ++f) integ.store<AUX>(f, ufirst, lane, w.integ[f]);
      }
    }
    if (steps_ptr) gstore<AUX>(steps_ptr + first + lane, steps);
    if constexpr (TRAJ) {
      const SoA<float> traj(ka.traj, 8, L);
      traj.store<AUX>(0, ufirst, lane, tr.calls);
      if (traj_dirty || kStateful) {  // the rest changes only at a reset — or, in the stateful modes, with any call
#pragma unroll
        for (int f = 1; f < (kStateful ? 8 : 7); ++f) traj.store<AUX>(f, ufirst, lane, tr.get(f));
      }
      if constexpr (kStateful) {
        const SoA<float> goal(goal_ptr, 12, L);
#pragma unroll
        for (int f = 0; f < 12; ++f) goal.store<AUX>(f, ufirst, lane, w.goal[f]);
      }
    }
    if (params_dirty) {
      const SoA<float> prm(a.params, 6, L);
#pragma unroll
      for (int f = 0; f < 6; ++f) prm.store<AUX>(f, ufirst, lane, w.prm[f]);
    }
  }
  if constexpr (HELP) {  // (the helper wave read the counter; this wave only advances it)
    if (lane == 0) __hip_atomic_fetch_add(a.reset_count + tile_id, n_steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (auto_reset && lane == 0) {
    a.reset_count[tile_id] = (int32_t)(rcount_s + (uint32_t)n_steps);  // never reuse a (tile, counter)
  }
  QR_STAMP(6, tid);
  QR_SPAN_END();
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
  Work<T, XV> w;
  idle_work(w, a.c);
  if (active) {
    load_state<XV, QW>(a, first, (unsigned)tid, w);
    if (a.goal) {
#pragma unroll
      for (int f = 0; f < 12; ++f) w.goal[f] = a.goal[(int64_t)f * L + i];
    }
#pragma unroll
    for (int f = 0; f < 8; ++f) w.integ[f] = a.integ[(int64_t)f * L + i];
  }
  T R[9];
  quat_to_R(w.q, R);
  float o0[D0];
  float o1[D1];
  error_obs<KIND, T, XV>(w, R, a.c, o0, o1);
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
  Work<T, XV> w;
  Draws d;
  draw20(d, a.seed, (uint64_t)(a.env_offset + i), (uint32_t)episode);
  sample_reset(w, d, randomise, eval, a.c);
  store_state<XV, QW>(a, (int64_t)blockIdx.x * 64, threadIdx.x, w);
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
  Work<QW, XV> w;
  load_state<XV, QW>(a, (int64_t)blockIdx.x * 64, threadIdx.x, w);
  double q[4], R[9];
#pragma unroll
  for (int j = 0; j < 4; ++j) q[j] = (double)w.q[j];
  quat_to_R(q, R);
  double* o = a.rows_out + i * 18;
#pragma unroll
  for (int j = 0; j < 3; ++j) { o[j] = (double)w.x[j]; o[3 + j] = (double)w.v[j]; o[15 + j] = (double)w.W[j]; }
#pragma unroll
  for (int j = 0; j < 9; ++j) o[6 + j] = R[j];
}

// state injection: float64 18-vector rows -> 13-word internal state (R -> nearest rotation -> q).
// A row whose attitude block has no nearest rotation in SO(3) (det R <= 0, or non-finite entries:
// quad_utils.py:123-142 would hand such an R to the SVD and return a reflection-corrected matrix
// that has nothing to do with the input) is REJECTED: the env keeps its state and the row is
// counted in *status, which the host side turns into an error.
template <typename XV, typename QW>
__global__ __launch_bounds__(64) void set_state_kernel(const Args a) {
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= a.n) return;
  if (a.mask && !a.mask[i]) return;
  const double* r = a.rows_in + i * 18;
  Work<QW, XV> w;
#pragma unroll
  for (int j = 0; j < 3; ++j) { w.x[j] = (XV)r[j]; w.v[j] = (XV)r[3 + j]; w.W[j] = (QW)r[15 + j]; }
  double q[4];
  const bool ok = R_to_quat(r + 6, q);
  if (!ok) {
    if (a.status) atomicAdd(a.status, 1);
    return;
  }
  if (a.dry_run) return;  // (qr_check_state: the same decision, nothing written)
#pragma unroll
  for (int j = 0; j < 4; ++j) w.q[j] = (QW)q[j];
  store_state<XV, QW>(a, (int64_t)blockIdx.x * 64, threadIdx.x, w);
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
  Work<T, XV> w;
  load_state<XV, QW>(a, first, lane, w);
  float th, tt, wb;
  if (a.draws) {
    th = a.draws[i]; tt = a.draws[a.n + i]; wb = a.draws[2 * a.n + i];
  } else {
    Draws d;
    draw20(d, a.seed, (uint64_t)(a.env_offset + i), (uint32_t)a.episode[i]);
    traj_draws(d.r[19], th, tt, wb);
  }
  Traj tr;
  if (a.goal_mode >= QR_GOAL_MODE2) traj_start<true>(w, tr, a.goal_mode, th, tt, wb);
  else traj_start<false>(w, tr, a.goal_mode, th, tt, wb);
  const SoA<float> traj(a.traj, 8, a.ld);
#pragma unroll
  for (int f = 0; f < 8; ++f) traj.store(f, (unsigned)first, lane, tr.get(f));
  if (a.goal_mode >= QR_GOAL_MODE2) {  // the stateful modes: the persistent fields of a fresh generator
    const SoA<float> goal(a.goal, 12, a.ld);
#pragma unroll
    for (int f = 0; f < 12; ++f) goal.store(f, (unsigned)first, lane, w.goal[f]);
  }
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
  Work<T, XV> w;
  idle_work(w, a.c);
  load_state<XV, QW>(a, first, lane, w);
  const SoA<float> traj(a.traj, 8, a.ld);
  Traj tr;
#pragma unroll
  for (int f = 0; f < 8; ++f) tr.set(f, traj.load(f, (unsigned)first, lane));
  const bool stateful = a.goal_mode >= QR_GOAL_MODE2;  // modes 2-5: xd, vd, b1d, Wd persist in the goal buffer
  if (stateful) {
    const SoA<float> goal(a.goal, 12, a.ld);
#pragma unroll
    for (int f = 0; f < 12; ++f) w.goal[f] = goal.load(f, (unsigned)first, lane);
  }
  float b1d_dot[3];
  if (stateful) traj_goal<true>(w, tr, a.goal_mode, a.c, b1d_dot);
  else traj_goal<false>(w, tr, a.goal_mode, a.c, b1d_dot);
  traj.store(0, (unsigned)first, lane, tr.calls);
  if (stateful) {
#pragma unroll
    for (int f = 1; f < 8; ++f) traj.store(f, (unsigned)first, lane, tr.get(f));
  }
  if (a.goal_rows) {
    float* o = a.goal_rows + i * 15;
#pragma unroll
    for (int j = 0; j < 9; ++j) o[j] = w.goal[j];
#pragma unroll
    for (int j = 0; j < 3; ++j) { o[9 + j] = b1d_dot[j]; o[12 + j] = w.goal[9 + j]; }
  }
  if ((a.store_goal || stateful) && a.goal) {
    const SoA<float> goal(a.goal, 12, a.ld);
#pragma unroll
    for (int f = 0; f < 12; ++f) goal.store(f, (unsigned)first, lane, w.goal[f]);
  }
}

// ------------------------------------------------------------------------------------
// qr_touch: the step's memory traffic and nothing else — the yardstick bench.py prices a step against (roofline.noop_kernel_us).
// Per env exactly what qr_step moves: state in and out (same SoA accesses, same widths), parameters, action row, [goal],
// [integrator words in and out], [observation rows out], reward and done rows out; no arithmetic beyond one sum that keeps
// the loads alive.  One wavefront per 64-env tile, plain stores.  The state is written back as read (bit for bit); the
// output rows hold zeros afterwards.
// ------------------------------------------------------------------------------------
template <int KIND, typename XV, typename QW>
__global__ __launch_bounds__(64) void touch_kernel(void* pos_vel, void* att_rate, const float* action, float* params, float* integ_ptr, float* reward,
                                                   int32_t n_envs, int32_t ld_envs, const Args a_in) {
  // (like step_kernel: what the first loads need arrives in preloaded SGPRs, the rest is read from the kernarg segment where it is used)
#if defined(__HIP_DEVICE_COMPILE__)
  const Args& a = *reinterpret_cast<const Args*>(reinterpret_cast<const char*>(__builtin_amdgcn_kernarg_segment_ptr()) + kArgsOffset);
  (void)a_in;
#else
  const Args& a = a_in;
#endif
  using KT = KindTraits<KIND>;
  constexpr int A = KT::A, D0 = KT::D0, D1 = KT::D1, NAG = KT::NAG;
  // the output pointers, read with the wave's first scalar loads (not at the very end, where a scalar-cache miss on the kernarg
  // segment would hold the wave's registers: the plain step kernel does the same, quadrotor_kernels.hip "done_ptr")
  uint8_t* const done_ptr = a.done;
  float* const obs0_ptr = a.obs0;
  float* const obs1_ptr = KT::D1 > 0 ? a.obs1 : nullptr;
  float* const goal_ptr = KIND != QR_KIND_QUAD ? a.goal : nullptr;
  const unsigned first = blockIdx.x * 64u, lane = threadIdx.x;
  const int rows = min(n_envs - (int)first, 64);
  const unsigned ll = min(lane, (unsigned)(rows - 1));
  const bool active = (int)lane < rows;
  const SoA<XV> pv(pos_vel, 6, ld_envs);
  const SoA<QW> ar(att_rate, 6, ld_envs);
  const SoA<float> prm(params, 6, ld_envs), integ(integ_ptr, 8, ld_envs), goal(goal_ptr, 12, ld_envs);
  XV x[6]; QW q[6]; float ig[8], pr[6], ac[A > 4 ? A : 4], gl[12];
  float s = 0.0f;
  // every load of the wave is issued before anything waits (one batch, like the step kernel's prologue)
#pragma unroll
  for (int f = 0; f < 6; ++f) q[f] = ar.load(f, first, ll);
#pragma unroll
  for (int f = 0; f < 6; ++f) x[f] = pv.load(f, first, ll);
#pragma unroll
  for (int f = 0; f < 6; ++f) pr[f] = prm.load(f, first, ll);
  const float* abase = action + (int64_t)first * A;
  if constexpr (A == 4) {
    const float4 v = reinterpret_cast<const float4*>(abase)[ll];
    ac[0] = v.x; ac[1] = v.y; ac[2] = v.z; ac[3] = v.w;
  } else {
#pragma unroll
    for (int j = 0; j < A; ++j) ac[j] = abase[ll * A + j];
  }
  const bool has_goal = KIND != QR_KIND_QUAD && goal_ptr != nullptr;
  if constexpr (KIND != QR_KIND_QUAD) {
#pragma unroll
    for (int f = 0; f < 8; ++f) ig[f] = integ.load(f, first, ll);
    if (has_goal) {
#pragma unroll
      for (int f = 0; f < 12; ++f) gl[f] = goal.load(f, first, ll);
    } else {
#pragma unroll
      for (int f = 0; f < 12; ++f) gl[f] = 0.0f;
    }
  }
  asm volatile("" ::"s"(done_ptr), "s"(obs0_ptr), "s"(obs1_ptr));   // (the scalar batch is waited for HERE: behind the vector loads' issue)
  // (pinned: left alone, the compiler sinks the state loads into the `active` block below, BEHIND the wait for the parameter and
  // action loads — two dependent round trips per wave instead of one batch: 28.7 instead of 24.7 us at 1 M envs)
#pragma unroll
  for (int f = 0; f < 6; ++f) asm volatile("" : "+v"(q[f]), "+v"(x[f]), "+v"(pr[f]));
#pragma unroll
  for (int j = 0; j < A; ++j) asm volatile("" : "+v"(ac[j]));
  if constexpr (KIND != QR_KIND_QUAD) {
#pragma unroll
    for (int f = 0; f < 8; ++f) asm volatile("" : "+v"(ig[f]));
#pragma unroll
    for (int f = 0; f < 12; ++f) { asm volatile("" : "+v"(gl[f])); s += gl[f]; }
  }
#pragma unroll
  for (int f = 0; f < 6; ++f) s += pr[f];
#pragma unroll
  for (int j = 0; j < A; ++j) s += ac[j];
  s = s * 0.0f;  // (0 for finite inputs; not foldable without fast-math, so the loads stay)
  if (active) {
#pragma unroll
    for (int f = 0; f < 6; ++f) ar.store(f, first, lane, q[f]);
#pragma unroll
    for (int f = 0; f < 6; ++f) pv.store(f, first, lane, x[f]);
    if constexpr (KIND != QR_KIND_QUAD) {
#pragma unroll
      for (int f = 0; f < 8; ++f) integ.store(f, first, lane, ig[f]);
    }
    if constexpr (NAG == 1) reward[first + lane] = s;
    else reinterpret_cast<float2*>(reward)[first + lane] = make_float2(s, s);
    if constexpr (NAG == 1) done_ptr[first + lane] = 0;
    else reinterpret_cast<uchar2*>(done_ptr)[first + lane] = make_uchar2(0, 0);
  }
  auto rows_out = [&](float* base, int D) {  // the tile's rows as they lie in memory: 16-byte stores, like lds_to_rows
    if (base == nullptr) return;
    float* g = base + (int64_t)first * D;
    if (rows == 64 && (reinterpret_cast<uintptr_t>(g) & 15u) == 0) {
      for (int idx = (int)lane; idx < 16 * D; idx += 64) reinterpret_cast<float4*>(g)[idx] = make_float4(s, s, s, s);
    } else {
      for (int idx = (int)lane; idx < rows * D; idx += 64) g[idx] = s;
    }
  };
  rows_out(obs0_ptr, D0);
  if constexpr (D1 > 0) rows_out(obs1_ptr, D1);
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
static float round_up_to_float(double v) {  // smallest float >= v
  float f = (float)v;
  if ((double)f < v) f = nextafterf(f, INFINITY);
  return f;
}

static void fill_coeffs(Coeffs& o, const QrCoeffs& q) {
  o.Cx = (float)q.Cx; o.CIx = (float)q.CIx; o.Cv = (float)q.Cv; o.Cb1 = (float)q.Cb1; o.CIb1 = (float)q.CIb1; o.CW = (float)q.CW;
  o.Cw12 = (float)q.Cw12; o.CW3 = (float)q.CW3;
  o.alpha = (float)q.alpha; o.beta = (float)q.beta; o.dt = q.dt; o.hdt = (float)(q.dt * 0.5);
  o.x_lim = q.x_lim; o.v_lim = q.v_lim; o.W_lim = q.W_lim;
  o.x_lim_f = (float)q.x_lim; o.x_lim_up = round_up_to_float(q.x_lim); o.v_lim_up = round_up_to_float(q.v_lim);
  const double lim = q.euler_lim_deg * kPi / 180.0;
  o.sin_euler_lim = sin(lim); o.tan_euler_lim = tan(lim); o.udm = (float)q.udm_fraction;
  o.reset_v = (float)(q.v_lim * 0.5); o.reset_W = (float)(q.W_lim * 0.5);
  const double rmin_mono = -ceil(q.Cx + q.CIx + q.Cv + q.Cb1 + q.CIb1 + q.CW);  // quad.py:81
  const double rmin_1 = -ceil(q.Cx + q.CIx + q.Cv + q.Cw12);                    // quad.py:85
  const double rmin_2 = -ceil(q.Cb1 + q.CW3 + q.CIb1);                          // quad.py:88
  o.rmin_mono = (float)rmin_mono; o.rmin_1 = (float)rmin_1; o.rmin_2 = (float)rmin_2;
  o.inv_x_lim = 1.0 / q.x_lim; o.inv_v_lim = 1.0 / q.v_lim; o.inv_W_lim = 1.0 / q.W_lim;
  o.inv_eIx_lim = (float)(1.0 / q.eIx_lim); o.inv_eIb1_lim = (float)(1.0 / q.eIb1_lim);
  o.inv_nrmin_mono = (float)(-1.0 / rmin_mono); o.inv_nrmin_1 = (float)(-1.0 / rmin_1); o.inv_nrmin_2 = (float)(-1.0 / rmin_2);
  const double nom[6] = {q.m_nominal, q.d_nominal, q.J1_nominal, q.J3_nominal, q.c_tf_nominal, q.c_tw_nominal};
  for (int j = 0; j < 6; ++j) { o.nom[j] = nom[j]; o.nom_f[j] = (float)nom[j]; }
  o.g = q.g; o.g_f = (float)q.g; o.min_force = q.min_force;
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
  if (e->goal_mode < 0 || e->goal_mode > QR_GOAL_MODE5) return QR_E_KIND;
  if (e->goal_mode != QR_GOAL_EXTERNAL && !e->traj) return QR_E_NULL;
  if (e->goal_mode >= QR_GOAL_MODE2 && !e->goal) return QR_E_NULL;  // the stateful modes keep xd, vd, b1d, Wd there
  if (!e->pos_vel || !e->att_rate) return QR_E_NULL;
  const QrCoeffs& q = e->coeffs;
  if (!(q.m_nominal > 0 && q.d_nominal > 0 && q.J1_nominal > 0 && q.J3_nominal > 0 && q.c_tf_nominal > 0 && q.c_tw_nominal > 0 &&
        q.g > 0 && q.min_force >= 0 && q.dt > 0)) return QR_E_SIZE;  // a zero-initialised QrCoeffs: call qr_default_coeffs first
  if ((reinterpret_cast<uintptr_t>(e->pos_vel) | reinterpret_cast<uintptr_t>(e->att_rate)) & 15u) return QR_E_ALIGN;
  a.pos_vel = e->pos_vel; a.att_rate = e->att_rate; a.integ = e->integ; a.params = e->params; a.goal = e->goal;
  a.traj = e->traj; a.goal_mode = e->goal_mode;
  a.episode = e->episode; a.steps = e->steps; a.reset_count = e->reset_count;
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

// The launch rule's thresholds.  The compiled-in defaults are crossovers measured on the pool's MI355X boxes (the comments at
// QR_HELPER_GRID*); boxes differ by 7-10 % in what they stream, and the wrappers' crossover moves with where the action rows come
// from, so every threshold can be overridden per process — environment variables of the same names, read once — and per env
// through QrEnv.flags (QR_FLAG_FORCE_HELPER / QR_FLAG_NO_HELPER: what QuadVecEnv(autotune=True) sets after timing both
// instantiations for ITS kind, size, box and action source).  No choice changes a result bit
// (tests/test_gpu_parity.py: test_helper_wave_launch_equals_the_plain_one, test_launch_rule_overrides_change_no_bit).
struct Tuning {
  unsigned helper_grid, helper_grid_wrap, helper_grid_rollout;
};
static unsigned env_uint(const char* name, unsigned dflt) {
  const char* v = getenv(name);
  if (!v || !*v) return dflt;
  char* end = nullptr;
  const unsigned long x = strtoul(v, &end, 10);
  return (end && *end == 0) ? (unsigned)x : dflt;
}
static const Tuning& tuning() {
  static const Tuning t = [] {
    Tuning x;
    x.helper_grid = env_uint("QR_HELPER_GRID", QR_HELPER_GRID);
    x.helper_grid_wrap = env_uint("QR_HELPER_GRID_WRAP", x.helper_grid < (unsigned)QR_HELPER_GRID_WRAP ? x.helper_grid : (unsigned)QR_HELPER_GRID_WRAP);
    x.helper_grid_rollout = env_uint("QR_HELPER_GRID_ROLLOUT", x.helper_grid < (unsigned)QR_HELPER_GRID_ROLLOUT ? x.helper_grid : (unsigned)QR_HELPER_GRID_ROLLOUT);
    return x;
  }();
  return t;
}

// Which instantiation a launch gets (shared by launch_kind and qr_step_kernel_info).
// (in regime for sure: done envs are re-sampled — in the launch, or, between two ONE-STEP launches, by the caller
// (QR_FLAG_CALLER_RESETS: a promise nobody can keep between the steps of a multi-step launch, which therefore ignores it))
static inline bool wants_adapt(const Args& a) {
  const bool resampled = (a.flags & QR_FLAG_AUTO_RESET) || ((a.flags & QR_FLAG_CALLER_RESETS) && a.n_steps == 1);
  return a.c.inv_w_adapt > 0 && (!resampled || a.c.inv_w_adapt * a.c.W_lim * 2.5 > 1.0);
}
static inline bool helper_choice(const Args& a, unsigned tiles, unsigned limit) {  // (the instantiation exists: rule, or the env's override for this launch family)
  const bool multi = a.n_steps > 1 || a.act_out != nullptr;
  if (a.flags & (multi ? QR_FLAG_NO_HELPER_ROLLOUT : QR_FLAG_NO_HELPER)) return false;
  if (a.flags & (multi ? QR_FLAG_FORCE_HELPER_ROLLOUT : QR_FLAG_FORCE_HELPER)) return true;
  return tiles <= limit;
}
static inline bool wants_helper(const Args& a, int kind, int layout, unsigned tiles_of_launch = 0) {  // a helper wave per tile (HELP)
  const unsigned tiles = tiles_of_launch ? tiles_of_launch : (unsigned)((a.n + 63) / 64);
  // (the multi-step instantiations hold the loop's state across steps: 181-216 VGPRs = two waves per SIMD, so a stepping
  // and a helper wave per tile are all resident only up to 1024 tiles; beyond, measured: Quad-v0 98 304 envs 3.52 against
  // 2.97 us per env-step plain, Coupled 5.06 against 3.74)
  const Tuning& tn = tuning();
  const unsigned quad_limit = a.substeps <= 1 || tn.helper_grid < (unsigned)QR_HELPER_GRID_SUBSTEPS ? tn.helper_grid : (unsigned)QR_HELPER_GRID_SUBSTEPS;
  // (2560 measured with one substep only.  Several substeps — since round 6 the Magnus substep — re-measured, profiles/r06/
  //  ab_magnus_helper_sweep.txt: the wrappers' helper launch is ahead up to 1664 tiles (x 2 / x 4: -6...7 %), level at 1792, behind
  //  from 1920 on (2048 tiles: +5...15 %); Quad-v0 keeps QR_HELPER_GRID_SUBSTEPS = 2560.)
  const unsigned wrap_limit = a.substeps <= 1 || tn.helper_grid_wrap < (unsigned)QR_HELPER_GRID_WRAP_SUBSTEPS ? tn.helper_grid_wrap : (unsigned)QR_HELPER_GRID_WRAP_SUBSTEPS;
  const unsigned limit = a.n_steps > 1 ? tn.helper_grid_rollout : (kind == QR_KIND_QUAD ? quad_limit : wrap_limit);
  return layout == QR_LAYOUT_MIXED && a.act_out == nullptr && a.goal_mode == QR_GOAL_EXTERNAL && !wants_adapt(a) &&
         (a.flags & QR_FLAG_AUTO_RESET) && helper_choice(a, tiles, limit);
}

static inline bool wants_helper_traj(const Args& a, int kind) {  // the same with the fused goal generator (one-step launches)
  const unsigned tiles = (unsigned)((a.n + 63) / 64);
  const Tuning& tn = tuning();
  const unsigned wrap_traj = a.substeps <= 1 ? 2048u : (unsigned)QR_HELPER_GRID_WRAP_SUBSTEPS;
  return a.act_out == nullptr && a.goal_mode != QR_GOAL_EXTERNAL && a.goal_mode < QR_GOAL_MODE2 && !wants_adapt(a) && (a.flags & QR_FLAG_AUTO_RESET) &&
         helper_choice(a, tiles, kind == QR_KIND_QUAD ? (tn.helper_grid < (unsigned)QR_HELPER_GRID_SUBSTEPS ? tn.helper_grid : (unsigned)QR_HELPER_GRID_SUBSTEPS)
                                                      : (tn.helper_grid_wrap < wrap_traj ? tn.helper_grid_wrap : wrap_traj));
}

// qr_rollout_actor beyond the grid on which a stepping AND a helper wave per tile are all resident: instead of the plain
// instantiation over the whole grid, the helper-wave instantiation over chunks of that many tiles, one launch after the other (each
// runs all n_steps of its envs; results do not depend on the split).  Measured, profiles/r05/ab_chunked_rollouts.txt: Coupled PPO
// collection 98 304 / 131 072 / 262 144 envs 8.00 / 8.13 / 16.3 -> 6.28 / 6.40 / 13.3 us per env-step, Decoupled 262 144 18.1 -> 14.1.
// Not for the plain rollouts, whose two stepping waves per SIMD use the vector unit better than chunks do (Coupled 262 144: 6.77
// against 7.60 us chunked).
static inline unsigned rollout_chunk(const Args& a, int kind, int layout) {
  const unsigned tiles = (unsigned)((a.n + 63) / 64), limit = tuning().helper_grid_rollout;
  if (a.act_out == nullptr || !QR_HELP_POLICY || kind == QR_KIND_QUAD || layout != QR_LAYOUT_MIXED || tiles <= limit || limit == 0) return 0;
  if ((a.flags & QR_FLAG_NO_HELPER_ROLLOUT) || !(a.flags & QR_FLAG_AUTO_RESET) || a.goal_mode != QR_GOAL_EXTERNAL) return 0;
  return limit;
}

// ------------------------------------------------------------------------------------
// Which instantiation of step_kernel a launch gets: ONE function decides (launch_kind dispatches on its result, qr_launch_plan
// reports it), and ONE table (QR_INSTANCES) lists every instantiation that exists.
// ------------------------------------------------------------------------------------
struct Pick {
  int traj; bool adapt; int policy; bool single, help, hrew, mag = false;
  // the bits qr_launch_stats counts under (with layout << 16 | kind << 8; MAG is bit 12, above the kind's two bits)
  unsigned bits() const {
    return (unsigned)traj | (adapt ? 4u : 0u) | ((unsigned)policy << 3) | (single ? 32u : 0u) | (help ? 64u : 0u) | (hrew ? 128u : 0u) | (mag ? 0x1000u : 0u);
  }
  unsigned slot() const { return (bits() & 0xFFu) | (mag ? 0x100u : 0u); }   // index into the counters
};

// `tiles_of_launch` != 0: one chunk of a chunked qr_rollout_actor (rollout_chunk).
static inline Pick pick_shape(const Args& a, int kind, int layout, unsigned tiles_of_launch) {
  const bool mixed = layout == QR_LAYOUT_MIXED;  // the only layout with one-step (SINGLE) and helper-wave (HELP) instantiations
  const unsigned tiles = tiles_of_launch ? tiles_of_launch : (unsigned)((a.n + 63) / 64);
  // Rate adaptivity can only trigger when an env starts a step with max|W_i| > w_adapt.  With
  // AUTO_RESET every env whose rate error left its bound was re-sampled at the end of the step
  // that took it there (done): Quad-v0 |W_i| < W_lim, Coupled |W_i - Wd_i| < W_lim, Decoupled
  // |W - Wd| < 2 W_lim (|ew12_i| < W_lim and |eW3| < W_lim).  For goal rates |Wd| <= W_lim / 2
  // and w_adapt >= 2.5 W_lim (the default 16 rad/s is) the plain kernel computes the same bits.
  const bool adapt = wants_adapt(a);
  const bool traj = a.goal_mode != QR_GOAL_EXTERNAL, stateful = a.goal_mode >= QR_GOAL_MODE2;
  if (kind != QR_KIND_QUAD && a.act_out != nullptr) {  // ---- qr_rollout_actor ----
    const bool general = a.actor[0].ls_w || a.actor[0].squash != QR_ACTOR_TANH_MEAN ||
                         (kind == QR_KIND_DECOUPLED && (a.actor[1].ls_w || a.actor[1].squash != QR_ACTOR_TANH_MEAN));
    if (mixed) {
      // Actors with in-launch resets and external goals, on grids where every wave is resident: a helper wave
      // per tile (noise, reset pool, observation rows).  Measured, Coupled 65 536 envs, T = 32: 5.37 -> 4.51 us per step;
      // with the fused goal generator the same split measured SLOWER (5.65 -> 6.25 us per step, tools/ppo_rollout_bench.py;
      // both waves of a tile must be resident, which caps the kernel at 256 registers) and is not instantiated.
      // Stage arithmetic: like every other launch, the plain (non-adaptive) instantiation whenever adaptivity provably cannot
      // trigger (in-launch resets, w_adapt >= 2.5 W_lim) — the actor rollout then computes the same bits as qr_step on the
      // actions it sampled, and the delta-form stages are off its path (65 536 envs: 3.70 -> 3.56 us per env-step,
      // profiles/r05/ab_actor_plain.txt).  External goals only: with the fused generator the actor launches stay rate-adaptive.
      // (the general form — SAC's log_std head and rule — with the same split; measured, profiles/r05/ab_sac_helper.txt,
      //  65 536 envs, T = 32: Coupled 4.95 -> 4.38 us per env-step, Decoupled 5.54 -> 4.87, bit-identical)
      if (QR_HELP_POLICY && !traj && (a.flags & QR_FLAG_AUTO_RESET) && helper_choice(a, tiles, tuning().helper_grid_rollout))
        return {0, adapt, general ? 2 : 1, false, true, true};
      if (!traj && !adapt) return {0, false, general ? 2 : 1, false, false, true};
    }
    if (stateful) return {2, true, 2, false, false, true};  // stateful goal modes: the general actor form
    return {traj ? 1 : 0, true, general ? 2 : 1, false, false, true};
  }
  const bool help = mixed && wants_helper(a, kind, QR_LAYOUT_MIXED, tiles_of_launch);
  if (mixed && a.n_steps == 1) {  // ---- qr_step in the default layout: the instantiations without the loop over env-steps ----
    if (stateful) return {2, adapt, 0, true, false, true};  // take-off, landing, stay, circle: their own instantiations
    if (traj) {
      if (adapt) return {1, true, 0, true, false, true};
      return {1, false, 0, true, wants_helper_traj(a, kind), true};  // (fused goal generator + helper wave: one-step launches only)
    }
    if (adapt) return {0, true, 0, true, false, true};
    if (help) {
      // (Quad-v0, one substep, more than QR_HELP_REWARD_TILES tiles: the reward stays on the stepping wave — measured with the
      //  product's other choices in place, profiles/r05/ab_step_prio.txt: 98 304 envs 5.12 -> 4.92 us, 163 840 envs 7.31 -> 6.57;
      //  identical bits.  The wrappers, one substep, more than QR_HELP_ROWS_TILES tiles: the helper only samples the pool, the
      //  rows go out with the stepping wave — 114 688 envs Coupled 8.14 -> 6.98 us, Decoupled 8.21 -> 6.97; 131 072: 9.07 -> 8.61 /
      //  9.11 -> 8.69; 98 304 envs and below are better with the helper's rows; profiles/r05/ab_step_prio.txt)
      const unsigned lim = kind == QR_KIND_QUAD ? (unsigned)QR_HELP_REWARD_TILES : (unsigned)QR_HELP_ROWS_TILES;
      return {0, false, 0, true, true, !(a.substeps == 1 && tiles > lim)};
    }
    return {0, false, 0, true, false, true};
  }
  // ---- qr_rollout (any layout) and qr_step of the uniform layouts ----
  if (stateful) return {2, adapt, 0, false, false, true};
  if (traj) return {1, adapt, 0, false, false, true};
  if (adapt) return {0, true, 0, false, false, true};
  return {0, false, 0, false, help, true};  // (rollouts in the default layout: a helper wave per tile for grids it pays on)
}

// The integrator rides on the env's `substeps` alone — never on the grid, so that a shard computes the bits of the global batch:
// two or more substeps in the default layout take the Magnus substep (MAG; qr_dynamics.h: 74 instead of 149 instructions per
// substep), one substep keeps RK4 in kernels that hold nothing else (byte-identical to the build without MAG).  The rate-adaptive
// delta-form instantiations (ADAPT without an actor: the launches whose envs may leave the regime) have their own arithmetic.
static constexpr bool uses_plain_integrate(int adapt, int policy) { return !(adapt && QR_DELTA_STAGES && !policy); }
static inline Pick pick_instance(const Args& a, int kind, int layout, unsigned tiles_of_launch = 0) {
  Pick p = pick_shape(a, kind, layout, tiles_of_launch);
  p.mag = layout == QR_LAYOUT_MIXED && a.substeps >= 2 && uses_plain_integrate(p.adapt, p.policy);
  return p;
}

// Every instantiation: (TRAJ, ADAPT, POLICY, SINGLE, HELP, HREW) x MAG.  POLICY != 0 exists for the wrappers only; SINGLE, HELP and the
// non-adaptive actor rollouts for the default layout only (inst_exists) — 16 Quad-v0 + 2 x 27 wrapper kernels in the default
// layout, 6 + 2 x 11 in each uniform one; MAG = 1 twins of the default layout's rows that call `integrate` (all but the delta-form
// ones and the one-substep-only HREW = 0 rows): 9 Quad-v0 + 2 x 20.  175 in all.  tests/test_gpu_instances.py walks this table and checks that the suite launches all of it.
#define QR_INSTANCES(X)                                                                                                  \
  X(0, 0, 0, 0, 0, 1) X(0, 1, 0, 0, 0, 1) X(1, 0, 0, 0, 0, 1) X(1, 1, 0, 0, 0, 1) X(2, 0, 0, 0, 0, 1) X(2, 1, 0, 0, 0, 1) \
  X(0, 1, 1, 0, 0, 1) X(0, 1, 2, 0, 0, 1) X(1, 1, 1, 0, 0, 1) X(1, 1, 2, 0, 0, 1) X(2, 1, 2, 0, 0, 1)                     \
  X(0, 0, 1, 0, 0, 1) X(0, 0, 2, 0, 0, 1) X(0, 0, 1, 0, 1, 1) X(0, 1, 1, 0, 1, 1) X(0, 0, 2, 0, 1, 1) X(0, 1, 2, 0, 1, 1) \
  X(0, 0, 0, 0, 1, 1)                                                                                                    \
  X(0, 0, 0, 1, 0, 1) X(0, 1, 0, 1, 0, 1) X(1, 0, 0, 1, 0, 1) X(1, 1, 0, 1, 0, 1) X(2, 0, 0, 1, 0, 1) X(2, 1, 0, 1, 0, 1) \
  X(1, 0, 0, 1, 1, 1) X(0, 0, 0, 1, 1, 1) X(0, 0, 0, 1, 1, 0)
static constexpr bool inst_exists(int kind, bool mixed, int tr, int ad, int po, int si, int he, int hr, int mg = 0) {
  (void)tr;
  // (MAG: the default layout's rows that call `integrate`; HREW = 0 is a one-substep choice, pick_shape)
  return !(po != 0 && kind == QR_KIND_QUAD) && (mixed || !(si || he || (po != 0 && !ad))) && (!mg || (mixed && uses_plain_integrate(ad, po) && hr));
}
// (a consumer of the table defines QR_X1 with the seventh column, MAG)
#define QR_X(TR, AD, PO, SI, HE, HR) QR_X1(TR, AD, PO, SI, HE, HR, 0) QR_X1(TR, AD, PO, SI, HE, HR, 1)

// Host-side launch counters, one per (layout, kind, instantiation): which kernels a process really ran (qr_launch_stats).
static std::atomic<uint32_t> g_launches[3][3][512];

template <int KIND, typename XV, typename QW>
static int launch_kind(const Args& a, hipStream_t s, unsigned tiles_of_launch = 0) {
  constexpr bool kMixed = std::is_same<XV, float>::value && std::is_same<QW, double>::value;
  constexpr int kLayout = kMixed ? QR_LAYOUT_MIXED : (std::is_same<XV, double>::value ? QR_LAYOUT_F64 : QR_LAYOUT_F32);
  if constexpr (kMixed) {
    if (tiles_of_launch == 0) {
      if (const unsigned chunk = rollout_chunk(a, KIND, QR_LAYOUT_MIXED)) {
        const unsigned tiles = (unsigned)((a.n + 63) / 64);
        for (unsigned base = 0; base < tiles; base += chunk) {
          Args b = a;
          b.tile_base = (int32_t)base;
          if (int rc = launch_kind<KIND, XV, QW>(b, s, tiles - base < chunk ? tiles - base : chunk)) return rc;
        }
        return 0;
      }
    }
  }
  const dim3 grid(tiles_of_launch ? tiles_of_launch : (unsigned)((a.n + 63) / 64));
  const Pick p = pick_instance(a, KIND, kLayout, tiles_of_launch);
#define QR_STEP_ARGS a.pos_vel, a.att_rate, a.action, a.params, a.integ, ((a.flags & QR_FLAG_AUTO_RESET) ? a.reset_count : nullptr), (int32_t)a.n, (int32_t)a.ld, a
#define QR_X1(TR, AD, PO, SI, HE, HR, MG)                                                                                         \
  if constexpr (inst_exists(KIND, kMixed, TR, AD, PO, SI, HE, HR, MG)) {                                                          \
    if (p.traj == TR && p.adapt == (bool)AD && p.policy == PO && p.single == (bool)SI && p.help == (bool)HE && p.hrew == (bool)HR && \
        p.mag == (bool)MG) {                                                                                                      \
      g_launches[kLayout][KIND][p.slot()].fetch_add(1u, std::memory_order_relaxed);                                              \
      hipLaunchKernelGGL((step_kernel<KIND, XV, QW, 64, TR, (bool)AD, PO, (bool)SI, (bool)HE, (bool)HR, (bool)MG>), grid, dim3(HE ? 128 : 64), 0, s, QR_STEP_ARGS); \
      return 0;                                                                                                                   \
    }                                                                                                                             \
  }
  QR_INSTANCES(QR_X)
#undef QR_X1
#undef QR_STEP_ARGS
  return QR_E_KIND;  // (unreachable: pick_instance only returns rows of the table)
}

// QR_ONLY_KIND / QR_ONLY_LAYOUT: experiment builds that instantiate one env kind / one layout only
// (seconds instead of a minute to compile; tools/ab_libs.py); the product build has neither.
template <typename XV, typename QW>
static int launch_step(const Args& a, int kind, hipStream_t s) {
  if (a.n == 0) return 0;
#ifdef QR_ONLY_KIND
  if (kind != QR_ONLY_KIND) return QR_E_KIND;
  int rc = launch_kind<QR_ONLY_KIND, XV, QW>(a, s);
#else
  int rc = 0;
  switch (kind) {
    case QR_KIND_QUAD: rc = launch_kind<QR_KIND_QUAD, XV, QW>(a, s); break;
    case QR_KIND_COUPLED: rc = launch_kind<QR_KIND_COUPLED, XV, QW>(a, s); break;
    default: rc = launch_kind<QR_KIND_DECOUPLED, XV, QW>(a, s); break;
  }
#endif
  return rc ? rc : (int)hipGetLastError();
}

#ifdef QR_ONLY_LAYOUT
#define QR_DISPATCH_LAYOUT(layout, CALL) { using XV = float; using QW = double; CALL; }
#else
#define QR_DISPATCH_LAYOUT(layout, CALL)                                        \
  switch (layout) {                                                             \
    case QR_LAYOUT_MIXED: { using XV = float; using QW = double; CALL; } break; \
    case QR_LAYOUT_F64:   { using XV = double; using QW = double; CALL; } break; \
    default:              { using XV = float; using QW = float; CALL; } break;  \
  }
#endif

template <typename XV, typename QW>
static void launch_error_obs(const Args& a, int kind, unsigned grid, hipStream_t s) {
  if (kind == QR_KIND_COUPLED) hipLaunchKernelGGL((error_obs_kernel<QR_KIND_COUPLED, XV, QW>), dim3(grid), dim3(64), 0, s, a);
  else hipLaunchKernelGGL((error_obs_kernel<QR_KIND_DECOUPLED, XV, QW>), dim3(grid), dim3(64), 0, s, a);
}
template <typename XV, typename QW>
static void launch_touch(const Args& a, int kind, unsigned grid, hipStream_t s) {
#define QR_TOUCH_ARGS a.pos_vel, a.att_rate, a.action, a.params, a.integ, a.reward, (int32_t)a.n, (int32_t)a.ld, a
#ifdef QR_ONLY_KIND
  hipLaunchKernelGGL((touch_kernel<QR_ONLY_KIND, XV, QW>), dim3(grid), dim3(64), 0, s, QR_TOUCH_ARGS);
  return;
#endif
  if (kind == QR_KIND_QUAD) hipLaunchKernelGGL((touch_kernel<QR_KIND_QUAD, XV, QW>), dim3(grid), dim3(64), 0, s, QR_TOUCH_ARGS);
  else if (kind == QR_KIND_COUPLED) hipLaunchKernelGGL((touch_kernel<QR_KIND_COUPLED, XV, QW>), dim3(grid), dim3(64), 0, s, QR_TOUCH_ARGS);
  else hipLaunchKernelGGL((touch_kernel<QR_KIND_DECOUPLED, XV, QW>), dim3(grid), dim3(64), 0, s, QR_TOUCH_ARGS);
#undef QR_TOUCH_ARGS
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
  if (!q.fc1_w || !q.fc1_b || !q.fc2_w || !q.fc2_b || !q.mean_w || !q.mean_b) return QR_E_NULL;
  if (!q.log_std && !(q.log_std_w && q.log_std_b)) return QR_E_NULL;  // one of the two log_std sources
  if ((q.log_std_w == nullptr) != (q.log_std_b == nullptr)) return QR_E_NULL;
  if (q.squash != QR_ACTOR_TANH_MEAN && q.squash != QR_ACTOR_TANH_SAMPLE) return QR_E_KIND;
  w.fc1_w = q.fc1_w; w.fc1_b = q.fc1_b; w.fc2_w = q.fc2_w; w.fc2_b = q.fc2_b;
  w.mean_w = q.mean_w; w.mean_b = q.mean_b; w.log_std = q.log_std;
  w.ls_w = q.log_std_w; w.ls_b = q.log_std_b; w.squash = q.squash;
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
  if ((env->flags & QR_FLAG_AUTO_RESET) && (!env->episode || !env->reset_count)) return QR_E_NULL;
  if (env->kind == QR_KIND_DECOUPLED && out->final_obs0 && !out->final_obs1) return QR_E_NULL;
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
  a.action = action; a.obs0 = out->obs0; a.obs1 = out->obs1; a.final_obs0 = out->final_obs0; a.final_obs1 = out->final_obs1;
  a.reward = out->reward; a.reward_raw = out->reward_raw; a.done = out->done; a.truncated = out->truncated;
  a.n_steps = n_steps; a.substeps = substeps;
#ifdef QR_SPAN
  a.span_slot = g_span_slot; a.span_buf = g_span_buf;
#endif
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rc = 0;
  QR_DISPATCH_LAYOUT(env->layout, (rc = launch_step<XV, QW>(a, env->kind, s)));
  return rc;
}

}  // namespace qr

extern "C" {

#ifdef QR_SPAN
int qr_debug_set_span(void* buf) {  // diagnostic build only: device buffer [slots][2 * tiles][2] of uint64 (NULL = off) for the NEXT launches
  qr::g_span_buf = reinterpret_cast<unsigned long long*>(buf);
  return 0;
}
void qr_debug_set_span_slot(int slot) { qr::g_span_slot = slot; }  // the row the NEXT launches write (baked into a captured launch)
#endif

#ifdef QR_STAMPS
int qr_debug_set_stamps(void* buf) {  // diagnostic builds only: device buffer of 8 x uint64 per wave (NULL = off)
  unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(qr::g_stamps), &p, sizeof(p));
}
int qr_debug_set_hwid(void* buf) {  // 2 x uint64 per workgroup: HW_ID | XCC_ID << 32 of its stepping and helper wave (NULL = off)
  unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(qr::g_hwid), &p, sizeof(p));
}
#endif

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
  c->m_nominal = 2.15; c->d_nominal = 0.23; c->J1_nominal = 0.022; c->J3_nominal = 0.035;  // quad.py:28-33
  c->c_tf_nominal = 0.0135; c->c_tw_nominal = 2.2; c->g = 9.81; c->min_force = 0.5;         // quad.py:31-36
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

int qr_error_obs_format(const QrEnv* env, int32_t format, float* obs0, float* obs1, void* stream) {
  qr::Args a{};
  if (int rc = qr::fill_env(a, env)) return rc;
  if (env->kind == QR_KIND_QUAD) return QR_E_KIND;
  if (format != QR_KIND_COUPLED && format != QR_KIND_DECOUPLED) return QR_E_KIND;
  if (!env->integ || !obs0 || (format == QR_KIND_DECOUPLED && !obs1)) return QR_E_NULL;
  a.obs0 = obs0; a.obs1 = obs1;
  const unsigned grid = (unsigned)((a.n + 63) / 64);
  if (grid == 0) return 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  QR_DISPATCH_LAYOUT(env->layout, (qr::launch_error_obs<XV, QW>(a, format, grid, s)));
  return (int)hipGetLastError();
}

int qr_error_obs(const QrEnv* env, float* obs0, float* obs1, void* stream) {
  if (!env) return QR_E_NULL;
  return qr_error_obs_format(env, env->kind, obs0, obs1, stream);
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

int qr_set_state(const QrEnv* env, const double* rows, const uint8_t* mask, int32_t* rejected, void* stream) {
  qr::Args a{};
  if (int rc = qr::fill_env(a, env)) return rc;
  if (!rows) return QR_E_NULL;
  a.rows_in = rows; a.mask = mask; a.status = rejected;
  const unsigned grid = (unsigned)((a.n + 63) / 64);
  if (grid == 0) return 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  QR_DISPATCH_LAYOUT(env->layout, (qr::launch_set_state<XV, QW>(a, grid, s)));
  return (int)hipGetLastError();
}

int qr_check_state(const QrEnv* env, const double* rows, const uint8_t* mask, int32_t* rejected, void* stream) {
  qr::Args a{};
  if (int rc = qr::fill_env(a, env)) return rc;
  if (!rows || !rejected) return QR_E_NULL;
  a.rows_in = rows; a.mask = mask; a.status = rejected;
  a.dry_run = 1;  // count, write nothing
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

void qr_launch_thresholds(int32_t* step_quad, int32_t* step_wrappers, int32_t* rollout) {
  const qr::Tuning& tn = qr::tuning();
  if (step_quad) *step_quad = (int32_t)tn.helper_grid;
  if (step_wrappers) *step_wrappers = (int32_t)tn.helper_grid_wrap;
  if (rollout) *rollout = (int32_t)tn.helper_grid_rollout;
}

int qr_launch_plan(const QrEnv* env, int32_t n_steps, int32_t substeps, int32_t actor, QrLaunchPlan* plan) {
  if (!plan) return QR_E_NULL;
  memset(plan, 0, sizeof(*plan));
  qr::Args a{};
  if (int rc = qr::fill_env(a, env)) return rc;
  if (n_steps < 1 || substeps < 1 || actor < 0 || actor > 2) return QR_E_SIZE;
  if (actor && env->kind == QR_KIND_QUAD) return QR_E_KIND;
  a.n_steps = n_steps; a.substeps = substeps;
  if (actor) {  // qr_rollout_actor: what the decision reads of the policy block
    static float dummy;
    a.act_out = &dummy;
    a.actor[0].squash = a.actor[1].squash = actor == 2 ? QR_ACTOR_TANH_SAMPLE : QR_ACTOR_TANH_MEAN;
  }
  const unsigned tiles = (unsigned)((a.n + 63) / 64);
  const unsigned chunk = env->layout == QR_LAYOUT_MIXED ? qr::rollout_chunk(a, env->kind, env->layout) : 0u;
  const qr::Pick p = qr::pick_instance(a, env->kind, env->layout, chunk);
  plan->grid = (int32_t)(chunk ? chunk : tiles);
  plan->block = p.help ? 128 : 64;
  plan->launches = chunk ? (int32_t)((tiles + chunk - 1) / chunk) : 1;
  plan->traj = p.traj; plan->adapt = p.adapt; plan->policy = p.policy; plan->single = p.single; plan->help = p.help; plan->hrew = p.hrew;
  plan->mag = p.mag;
  plan->key = ((uint32_t)env->layout << 16) | ((uint32_t)env->kind << 8) | p.bits();
  static const char* const kXV[3] = {"float", "double", "float"};
  static const char* const kQW[3] = {"double", "double", "float"};
  snprintf(plan->name, sizeof(plan->name), "qr::step_kernel<%d,%s,%s,64,%d,%d,%d,%d,%d,%d,%d>", (int)env->kind, kXV[env->layout], kQW[env->layout],
           p.traj, (int)p.adapt, p.policy, (int)p.single, (int)p.help, (int)p.hrew, (int)p.mag);
  return 0;
}

const char* qr_step_kernel_info(const QrEnv* env, int32_t n_steps, int32_t* grid, int32_t* block) {
  QrLaunchPlan plan;
  if (qr_launch_plan(env, n_steps < 1 ? 1 : n_steps, 1, 0, &plan) != 0) return "";
  if (grid) *grid = plan.grid;
  if (block) *block = plan.block;
  switch (env->kind) {
    case QR_KIND_QUAD: return "qr::step_kernel<0,...>";
    case QR_KIND_COUPLED: return "qr::step_kernel<1,...>";
    case QR_KIND_DECOUPLED: return "qr::step_kernel<2,...>";
    default: return "";
  }
}

int32_t qr_launch_stats(uint32_t* keys, uint32_t* counts, int32_t capacity, int32_t reset) {
  int32_t n = 0;
  for (int l = 0; l < 3; ++l)
    for (int k = 0; k < 3; ++k)
      for (int b = 0; b < 512; ++b) {
        const uint32_t c = reset ? qr::g_launches[l][k][b].exchange(0u, std::memory_order_relaxed) : qr::g_launches[l][k][b].load(std::memory_order_relaxed);
        if (c == 0) continue;
        if (n < capacity && keys && counts) { keys[n] = ((uint32_t)l << 16) | ((uint32_t)k << 8) | (uint32_t)(b & 0xFF) | (b & 0x100 ? 0x1000u : 0u); counts[n] = c; }
        ++n;
      }
  return n;
}

int32_t qr_instance_table(uint32_t* keys, int32_t capacity) {
  int32_t n = 0;
  for (int l = 0; l < 3; ++l)
    for (int k = 0; k < 3; ++k) {
#define QR_X1(TR, AD, PO, SI, HE, HR, MG)                                                              \
  if (qr::inst_exists(k, l == QR_LAYOUT_MIXED, TR, AD, PO, SI, HE, HR, MG)) {                          \
    if (n < capacity && keys) keys[n] = ((uint32_t)l << 16) | ((uint32_t)k << 8) | qr::Pick{TR, (bool)AD, PO, (bool)SI, (bool)HE, (bool)HR, (bool)MG}.bits(); \
    ++n;                                                                                               \
  }
      QR_INSTANCES(QR_X)
#undef QR_X1
    }
  return n;
}

int qr_touch(const QrEnv* env, const float* action, const QrStepOut* out, void* stream) {
  qr::Args a{};
  if (int rc = qr::fill_env(a, env)) return rc;
  if (!action || !out || !out->reward || !out->done) return QR_E_NULL;
  if (env->kind != QR_KIND_QUAD && (!env->integ || !out->obs0)) return QR_E_NULL;
  if (env->kind == QR_KIND_DECOUPLED && !out->obs1) return QR_E_NULL;
  if (reinterpret_cast<uintptr_t>(action) & (env->kind == QR_KIND_DECOUPLED ? 3u : 15u)) return QR_E_ALIGN;
  a.action = action; a.obs0 = out->obs0; a.obs1 = out->obs1; a.reward = out->reward; a.done = out->done;
  const unsigned grid = (unsigned)((a.n + 63) / 64);
  if (grid == 0) return 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#ifdef QR_ONLY_KIND
  if (env->kind != QR_ONLY_KIND) return QR_E_KIND;
#endif
  QR_DISPATCH_LAYOUT(env->layout, (qr::launch_touch<XV, QW>(a, env->kind, grid, s)));
  return (int)hipGetLastError();
}

}  // extern "C"
