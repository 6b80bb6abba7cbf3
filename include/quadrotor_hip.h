/*
 * quadrotor_hip.h — C-ABI of the MI355X-native batched quadrotor dynamics engine.
 *
 * Drop-in boundary for ONE hot path of fdcl-gwu/gym-rotor: env.step() of Quad-v0 /
 * CoupledWrapper / DecoupledWrapper (the template method QuadEnv.step,
 * gym_rotor/envs/quad.py:142-168, and the hooks it calls), batched over N independent
 * quadrotors that live in device memory.  The reference is pure Python and has no
 * native layer, so each entry point below names the reference *method* it replaces.
 *
 * Conventions
 *   - plain C: pointers, sizes, PODs; no torch / C++ types.  All pointers are DEVICE
 *     pointers (HBM) unless stated otherwise.  `stream` is a hipStream_t passed as void*
 *     (NULL = the default stream).  Every call is asynchronous on `stream`.
 *   - return value: 0 on success; <0 = argument error (QR_E_*); >0 = hipError_t of the
 *     launch.  Nothing is ever computed on the host: there is no CPU fallback.
 *   - layout: per-env quantities are SoA, `field-major [F][N]` (lane i touches
 *     base[f*N + i], so a 64-lane wavefront reads 256/512 contiguous bytes per field).
 *     With field_stride = L the address is base[f*L + i].
 *     Caller-facing rows (actions in, observations/reward/done out) are AoS `[N][D]`
 *     row-major, i.e. what a policy network produces/consumes; the kernels transpose
 *     through LDS so that those rows are written with coalesced 16-byte stores.
 *   - INTERNAL STATE (12 words per env instead of the reference's 18): the step is
 *     HBM/fabric-bound, so the rotation is kept as a unit quaternion, which the kernel
 *     integrates directly (q' = q (0,W)/2 is the same flow as R' = R hat(W),
 *     quad.py:328) and stores as its THREE components of smaller magnitude ("smallest
 *     three": the dropped one is made positive and rebuilt as sqrt(1 - sum of squares) >= 1/2;
 *     its index rides in the two lowest mantissa bits of the first stored component):
 *         pos_vel  [6][N]  x(3), v(3)
 *         att_rate [6][N]  k0, k1, k2 of q = (w,x,y,z) with R = R(q); W(3)
 *     An all-zero att_rate is the identity attitude at rest.
 *     The reference's 18-vector (x, v, vec_F(R) column-major, W; quad.py:146,
 *     quad_utils.py:12-16) is produced / consumed by qr_get_state / qr_set_state.
 *   - precision (`layout`): QR_LAYOUT_MIXED (default) stores x,v as float32 and q,W as
 *     float64; W is integrated and q ACCUMULATED in float64, the RK4 stage quaternions and
 *     the thrust direction are formed in float32 (same 1000-step error as
 *     all-float64 arithmetic).  QR_LAYOUT_F64 stores and computes everything in float64 (the
 *     reference-grade mode: 4th-order convergence to the reference down to 1e-10);
 *     QR_LAYOUT_F32 stores and computes in float32 (fast, NOT inside the 1e-5/1000-step
 *     parity bar).  See DESIGN.md §4 for the measurements behind this.
 */
#ifndef QUADROTOR_HIP_H
#define QUADROTOR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QR_ABI_VERSION 15

/* env kinds */
#define QR_KIND_QUAD      0 /* QuadEnv            gym_rotor/envs/quad.py:19            */
#define QR_KIND_COUPLED   1 /* CoupledWrapper     wrappers/coupled_yaw_wrapper.py:11   */
#define QR_KIND_DECOUPLED 2 /* DecoupledWrapper   wrappers/decoupled_yaw_wrapper.py:12 */

/* state layouts */
#define QR_LAYOUT_MIXED 0 /* pos_vel float32, att_rate float64; float64 W + q accumulation, float32 RK4 stages */
#define QR_LAYOUT_F64   1 /* all float64                                           */
#define QR_LAYOUT_F32   2 /* all float32                                           */

/* argument errors */
#define QR_E_NULL   (-1) /* a required pointer is NULL             */
#define QR_E_KIND   (-2) /* kind / layout / actor.squash out of range, or an entry point undefined for the kind */
#define QR_E_SIZE   (-3) /* num_envs < 0 or > 22 369 621 (32-bit SoA offsets), bad field_stride, substeps < 1, n_steps < 1,
                            or a QrCoeffs with non-positive nominal parameters / dt (not filled by qr_default_coeffs) */
#define QR_E_ALIGN  (-4) /* a buffer is not 16-byte aligned        */

/* goal sources */
#define QR_GOAL_EXTERNAL 0 /* goal buffer set by the caller (set_goal_state), or the hover default    */
#define QR_GOAL_MODE0    1 /* TrajectoryGenerator mode 0: xd = vd = 0, b1d drawn at episode start     */
#define QR_GOAL_MODE1    2 /* TrajectoryGenerator mode 1: exponential approach of the origin + yaw rate */
#define QR_GOAL_MODE6    3 /* TrajectoryGenerator mode 6: eight-shaped (Lissajous) curve (:418-505); after
                              eight_count periods the last goal is held (the reference switches to manual) */
/* The STATEFUL modes (:279-416): the generator object carries xd, vd, b1d, b1d_dot, Wd and its flags from call to call (a mode
 * writes only the components it moves; after completion it switches to manual mode, which holds position and heading and no
 * longer recomputes Wd).  They need QrEnv.goal: xd, vd, b1d, Wd persist THERE; b1d_dot, the flags and x_init in QrEnv.traj. */
#define QR_GOAL_MODE2    4 /* mode 2 take-off: climb at 0.05 m/s from the start position to the height -0.5 m, then manual    */
#define QR_GOAL_MODE3    5 /* mode 3 landing: descend at 1 m/s to the motor cut-off height -0.25 m                            */
#define QR_GOAL_MODE4    6 /* mode 4 stay: hold the current position and heading (manual mode from the second call on)        */
#define QR_GOAL_MODE5    7 /* mode 5 circle: 1.75 s run-up along +x, two circles of radius 0.7 m at 0.4 rad/s, then manual    */
/* Two deviations of the stateful modes (2-5) from the reference's long-lived TrajectoryGenerator object, both bounded:
 *  (1) Episode start.  mark_traj_start (trajectory_generator.py:176-191) keeps xd, vd, b1d, b1d_dot, Wd of the PREVIOUS episode on
 *      the object; here every episode starts from a fresh generator (xd = vd = Wd = 0, b1d = e1, b1d_dot = 0).  Modes 2-4
 *      overwrite all of them in their first call, so only mode 5 can tell: an episode that ended mid-circle would carry
 *      b1d_dot != 0 into the next run-up, i.e. a non-zero Wd for the first call (|Wd_3| <= 0.4 rad/s), until the run-up branch
 *      rewrites b1d_dot at that same call's end.  (A per-env generator has no "previous object"; the oracle models the same.)
 *  (2) Phase boundaries.  The generator's clock is t = calls * dt formed in float32; the reference accumulates t += dt in
 *      float64 (:224-229).  A phase test `t < t_switch` therefore flips at the same call unless t_switch lies within ~1e-6 s
 *      of a multiple of dt — then one call early or late: a one-step goal error of at most |v| dt (take-off 0.25 mm, landing
 *      5 mm, circle run-up 2 mm).  The one boundary that IS an exact multiple of the default dt (the circle's run-up, 350 dt)
 *      is matched to the reference's accumulated value (1.7499999999999847 s: still run-up).  Pinned by the closed-loop
 *      goldens for the default dt (tests/test_closedloop_td3.py); other dt: the bound above. */

/* flags */
#define QR_FLAG_AUTO_RESET   1u /* re-sample a done env inside the same launch (train distribution) */
#define QR_FLAG_EVAL_RESET   2u /* resets use env_type='eval' (quad.py:352-356) instead of 'train' */
#define QR_FLAG_NO_UDM       4u /* resets keep nominal parameters (use_UDM False / eval)            */
#define QR_FLAG_CALLER_RESETS 32u /* WITHOUT QR_FLAG_AUTO_RESET: the caller promises the reference's own loop (main.py:183-186,
                                    212-230): every env a step reports done is reset (qr_reset / qr_set_state) before it is stepped
                                    again.  Then no env ever starts a step outside the termination bounds, rate adaptivity
                                    (QrCoeffs.w_adapt) cannot trigger, and the launcher takes the kernel compiled without it — the
                                    same arithmetic, bit for bit, as with QR_FLAG_AUTO_RESET (4.03 instead of 4.39 us per 65 536-env
                                    launch).  A caller that steps done envs on anyway only loses the adaptivity: fixed substeps.
                                    Honoured by ONE-STEP launches only (qr_step): inside a qr_rollout / qr_rollout_actor launch of
                                    several steps nobody can reset an env between two of them, so those keep the rate-adaptive kernel. */
/* Launch-rule overrides (speed only: no choice changes a result bit).  By default qr_step / qr_rollout(_actor) decide from the
 * grid size whether every 64-env tile gets a second, helper wavefront, with thresholds measured on MI355X (environment variables
 * QR_HELPER_GRID, QR_HELPER_GRID_WRAP, QR_HELPER_GRID_ROLLOUT override them per process, read once); these bits pin the choice
 * per env — what an autotuner that timed both on ITS box, kind, size and action source sets.  Ignored where the instantiation
 * does not exist (a helper wavefront needs QR_FLAG_AUTO_RESET, the default layout and no rate adaptivity in reach). */
#define QR_FLAG_FORCE_HELPER   8u /* qr_step: a helper wavefront per 64-env tile whatever the grid size */
#define QR_FLAG_NO_HELPER     16u /* qr_step: never                                                     */
#define QR_FLAG_FORCE_HELPER_ROLLOUT  64u /* the same two for the multi-step launches (qr_rollout, qr_rollout_actor), whose crossover  */
#define QR_FLAG_NO_HELPER_ROLLOUT    128u /* is a different one (two waves per SIMD): a choice timed on one family says nothing about the other */
/* (qr_rollout_actor on grids beyond that crossover runs the helper-wave instantiation over chunks of resident tiles, one launch after
 *  the other, unless QR_FLAG_NO_HELPER_ROLLOUT is set: the same results, 18-22 % less time at 98 304 ... 262 144 envs.) */

/* Coefficients a caller may override (args_parse.py:23-35); qr_default_coeffs() fills the
 * reference defaults.  reward_min* are derived inside (quad.py:81-88). */
typedef struct QrCoeffs {
  double Cx, CIx, Cv, Cb1, CIb1, CW;  /* MONO / Quad-v0 reward (CW = Cw12, quad.py:80) */
  double Cw12, CW3;                   /* MODUL                                         */
  double alpha, beta;                 /* integral leak terms (quad.py:448-450)         */
  double dt;                          /* 1/200 s (quad.py:60-61)                       */
  double x_lim, v_lim, W_lim;         /* 1.0, 4.0, 2*pi (quad.py:104-106)              */
  double eIx_lim, eIb1_lim;           /* 3.0, 3.0 (coupled_yaw_wrapper.py:23-24)       */
  double euler_lim_deg;               /* 85 (quad.py:107)                              */
  double udm_fraction;                /* UDM_percentage/100 = 0.1 (quad.py:370)        */
  /* eight-shaped curve of the goal generator (utils/trajectory_generator.py:98-110) */
  double eight_T, eight_A1, eight_A2; /* period 9 s, amplitudes 1.5 / 1.0                  */
  double eight_w_b1d, eight_alt_d;    /* yaw rate 0.349066 rad/s, desired altitude -0.6 m  */
  double eight_eps, eight_count;      /* smoothing epsilon 0.01, number of eights 3        */
  /* Rate-adaptive substepping (the fixed-step stand-in for DOP853's error control, quad.py:265):
   * a wavefront (64 consecutive envs) containing an env whose body rate max|W_i| exceeds w_adapt
   * takes ceil(max|W_i| / w_adapt) times the requested substeps (capped at 16x), so envs that
   * are stepped on far beyond termination (free run, no AUTO_RESET) keep the 1e-5 trajectory
   * bar.  Default 16 rad/s ~ 2.5x the termination bound W_lim = 2 pi: in-regime envs always take
   * exactly `substeps`, and with QR_FLAG_AUTO_RESET (every env is re-sampled when its rate
   * error leaves its bound; goal rates |Wd| <= W_lim / 2) it can never trigger, so the launcher
   * then uses the kernel compiled without it.  0 disables it. */
  double w_adapt;
  /* Nominal vehicle and environment constants (QuadEnv.__init__, quad.py:28-36): what a reset without
   * domain randomisation restores, what UDM scatters around, and what an env without a params
   * buffer flies with. */
  double m_nominal, d_nominal, J1_nominal, J3_nominal; /* 2.15 kg, 0.23 m, 0.022 (= J2), 0.035 kg m^2 */
  double c_tf_nominal, c_tw_nominal;                   /* 0.0135, 2.2                                 */
  double g, min_force;                                 /* 9.81 m/s^2, 0.5 N                           */
} QrCoeffs;

/* Per-env device buffers owned by the caller (the Python env object). */
typedef struct QrEnv {
  int32_t kind;        /* QR_KIND_*                                                       */
  int32_t layout;      /* QR_LAYOUT_*                                                     */
  int64_t num_envs;    /* N                                                               */
  int64_t field_stride;/* elements between consecutive fields of EVERY SoA buffer below; 0 = N.
                          A multiple of 4, >= N.  Padding it off a power of two avoids all
                          fields of an env landing on one HBM channel.        */
  int64_t env_offset;  /* global id of local env 0 (multi-GPU shard offset; RNG key)      */
  uint64_t seed;       /* RNG seed.  qr_reset draws depend only on (seed, global env id, episode); in-launch
                          resets (QR_FLAG_AUTO_RESET) on (seed, global id of the 64-env tile's first env,
                          reset_count of that tile, rank of the env among the tile's envs that reset in
                          that step) — see reset_count                                                */
  void*    pos_vel;    /* [6][N]  x, v                                         in/out      */
  void*    att_rate;   /* [6][N]  q (smallest three), W                        in/out      */
  float*   integ;      /* [8][N]  eIx(3), g_x prev(3), eIb1, g_b prev (quad_utils.py:38-63); NULL for QUAD */
  float*   params;     /* [6][N]  m,d,J1(=J2),J3,c_tf,c_tw (quad.py:359-387); NULL = nominal */
  float*   goal;       /* [12][N] xd,vd,b1d,Wd (quad.py:413-418); NULL = hover default     */
  float*   traj;       /* [8][N]  goal-generator state, required when goal_mode != QR_GOAL_EXTERNAL:
                          0 #get_desired calls since mark_traj_start (t = calls*dt), 1 theta_init,
                          2,3 b1d x,y (mode 0) | w_b1d, smooth_term (mode 1) | b1d_dot x, flags (modes 2-5),
                          4..6 x_init, 7 b1d_dot y (modes 2-5) */
  int32_t  goal_mode;  /* QR_GOAL_EXTERNAL (0), or a utils/trajectory_generator.py mode fused into
                          the step: QR_GOAL_MODE0 idle/warm-up (:141-148), QR_GOAL_MODE1 hovering (:252-277),
                          QR_GOAL_MODE6 eight-shaped curve (:418-505), QR_GOAL_MODE2..5 take-off / landing / stay /
                          circle (:279-416; these need `goal`) */
  int32_t  reserved0;
  int32_t* episode;    /* [N]     episode counter (RNG stream id); required for resets     */
  int32_t* steps;      /* [N]     steps since reset; NULL = no time-limit bookkeeping      */
  int32_t* reset_count;/* [ceil(N/64)] stream position of the in-launch reset of each 64-env tile (one
                          wavefront): advanced by 1 per env-step by qr_step / qr_rollout(_actor) with
                          QR_FLAG_AUTO_RESET (required then), so that a (tile, count) pair is never
                          used twice — also when a captured hipGraph is replayed.  The draws of tile
                          j depend only on (seed, env_offset + 64 j, count, slot): results do not
                          depend on how the batch is sharded as long as every shard starts at a
                          multiple of 64 envs.  Zero-initialise.                                     */
  int32_t  max_episode_steps; /* >0: truncated[i]=1 when steps reaches it (gym_rotor/__init__.py:3-7) */
  uint32_t flags;      /* QR_FLAG_*                                                       */
  QrCoeffs coeffs;
} QrEnv;

/* Outputs of one step, all AoS rows.  obs0 is float32 [N][18] (QUAD: the next state in the
 * reference's order, quad.py:269-271; optional), [N][23] (COUPLED) or [N][15] (DECOUPLED
 * agent 1); obs1 is [N][3] (DECOUPLED agent 2) else NULL.  reward/done are [N][n_agents],
 * n_agents = 2 for DECOUPLED, else 1. */
typedef struct QrStepOut {
  float*   obs0;        /* QUAD: may be NULL (no observation rows written)                */
  float*   obs1;
  float*   reward;      /* normalised to [0,1], -1 on crash (quad.py:154-166)             */
  float*   reward_raw;  /* optional: hook output before np.interp / crash override        */
  uint8_t* done;        /* terminated flags (done_wrapper)                                */
  uint8_t* truncated;   /* optional [N]                                                   */
  /* Optional, with QR_FLAG_AUTO_RESET: rows [N][D0] / [N][D1] (same shapes as obs0 / obs1; QUAD: [N][18])
   * that receive the TERMINAL observation of every env that is re-sampled in this step — the
   * obs_next the reference stores for that transition (main.py:163-175; obs0/obs1 then already hold
   * the first observation of the new episode).  Rows of envs that did not reset are left untouched. */
  float*   final_obs0;
  float*   final_obs1;
} QrStepOut;

/* Replaces QuadEnv.step(action) (quad.py:142-168) = action_wrapper -> observation_wrapper
 * (ODE solve over dt + get_norm_error_state) -> reward_wrapper -> interp -> done_wrapper ->
 * crash override, for all N envs in one fused launch.
 *   action   [N][A] float32, A = 4 (QUAD, COUPLED) or 5 (DECOUPLED: agents' actions
 *            concatenated, main.py:161).  16-byte aligned (A = 4) / 4-byte aligned (A = 5).
 *   substeps number of fixed 4th-order substeps of h = dt/substeps replacing
 *            scipy.integrate.solve_ivp(DOP853) (quad.py:265); >= 1.  ONE substep is an RK4 step; with two or more the default
 *            layout takes a Lie-group (Magnus) substep of the same order at half the instructions (qr_dynamics.h; the uniform
 *            layouts stay RK4).  The choice rides on `substeps` alone, never on the batch size or the launch family: a shard, a
 *            rollout and an actor rollout compute the bits of the global, per-step launches of the same `substeps`. */
int qr_step(const QrEnv* env, const float* action, int32_t substeps, const QrStepOut* out, void* stream);

/* K env-steps in ONE launch with the state held in registers between steps:
 *   action [K][N][A]; outputs [K][N][...] (pointers in `out` are the t=0 slices, the
 *   per-step strides are the full [N][..] extents).  Same maths as K calls of qr_step. */
int qr_rollout(const QrEnv* env, const float* action, int32_t n_steps, int32_t substeps,
               const QrStepOut* out, void* stream);

/* One MLP actor of the reference: fc1 -> relu -> fc2 -> relu -> heads.  Device pointers to float32
 * tensors in torch.nn.Linear layout (weight [out][in] row-major), i.e. `actor.fc1.weight.data_ptr()`
 * etc. as they are.
 *   QR_ACTOR_TANH_MEAN   PPO  MLP_Actor_PPO (algos/ppo/ppo_mlp.py:6-58) and TD3 MLP_Actor_TD3
 *                        (algos/td3/td3_mlp.py:5-34, mean head = fc3): mean = tanh(head(h));
 *                        action = clamp(mean + exp(log_std) eps, +-max_action)  (ppo.py:93-99, td3.py:93-96)
 *   QR_ACTOR_TANH_SAMPLE SAC  MLP_Actor_SAC (algos/sac/sac_mlp.py:16-82): mean = mean_linear(h),
 *                        log_std = clamp(log_std_linear(h), -20, 2); action = tanh(mean + exp(log_std) eps)
 * log_std: the state-independent parameter [action_dim] (PPO; TD3: log of the exploration std) —
 * or, when log_std_w is set, the second head log_std_w [action_dim][hidden], log_std_b [action_dim]. */
#define QR_ACTOR_TANH_MEAN   0
#define QR_ACTOR_TANH_SAMPLE 1
typedef struct QrActor {
  const float* fc1_w;   const float* fc1_b;   /* [hidden][obs_dim], [hidden]        */
  const float* fc2_w;   const float* fc2_b;   /* [hidden][hidden],  [hidden]        */
  const float* mean_w;  const float* mean_b;  /* [action_dim][hidden], [action_dim] */
  const float* log_std;                        /* [action_dim], or NULL with a log_std head */
  const float* log_std_w; const float* log_std_b; /* optional state-dependent log_std head  */
  int32_t obs_dim, hidden_dim, action_dim;
  int32_t squash;                              /* QR_ACTOR_TANH_MEAN | QR_ACTOR_TANH_SAMPLE */
} QrActor;

/* Caller side of a PPO collection loop (main.py:141-166 with PPO.choose_action, ppo.py:82-101)
 * for qr_rollout_actor. */
typedef struct QrPolicyRollout {
  const QrActor* actors;  /* host array, one per agent with the reference's default sizes
                             (args_parse.py:40): COUPLED 1 actor 23->16->16->4; DECOUPLED 2 actors
                             15->16->16->4 and 3->4->4->1.  Other sizes: QR_E_SIZE.             */
  const float* obs0_in;   /* [N][D0] observation the FIRST action is computed from (what the
                             last step / reset + get_norm_error_state returned)                */
  const float* obs1_in;   /* [N][3], DECOUPLED only                                            */
  const float* noise;     /* optional [K][N][A] standard-normal draws eps (action = mean + std
                             eps: reproduces a given torch sample; tests).  NULL = drawn in the
                             kernel: Philox4x32-10 keyed by (noise_seed, global env id,
                             step_base + t) + Box-Muller, independent of the sharding.         */
  uint64_t noise_seed;
  uint64_t step_base;     /* global step index of t = 0 (advance by K per call)               */
  float max_action;       /* clamp (args_parse.py:45: 1.0)                                    */
  int32_t deterministic;  /* != 0: action = clamp(mean) resp. tanh(mean) (is_eval: ppo.py:100-101, sac.py:104-105) */
  float* action_out;      /* [K][N][A] actions taken (agents concatenated, main.py:161)       */
  float* logprob_out;     /* [K][N][A] per component: TANH_MEAN Normal(mean, std).log_prob(action) of the
                             CLAMPED action (ppo.py:97-98); TANH_SAMPLE Normal.log_prob(u) - log(1 -
                             action^2 + 1e-6), u the pre-tanh sample (sac_mlp.py:74-77).  NULL allowed */
} QrPolicyRollout;

/* K env-steps in ONE launch with the policy inside the loop: per step, each env's actor(s) are
 * evaluated on its current observation (registers), the action is sampled, clamped and stepped,
 * and obs / action / logprob / reward / done rows are written as [K][N][..] (what
 * algos/replay_buffer.py:20-39 stores per step).  `out` as in qr_rollout (obs rows required):
 * out->obs0[t] is the observation AFTER step t, i.e. the input of step t+1.  COUPLED and
 * DECOUPLED only (the reference trains on the wrappers). */
int qr_rollout_actor(const QrEnv* env, const QrPolicyRollout* policy, int32_t n_steps, int32_t substeps,
                     const QrStepOut* out, void* stream);

/* Replaces QuadEnv.get_norm_error_state(framework) (quad.py:421-466): normalised error
 * observation of the CURRENT state; advances both trapezoid integrators (same side
 * effect as the reference).  kind must be COUPLED or DECOUPLED. */
int qr_error_obs(const QrEnv* env, float* obs0, float* obs1, void* stream);

/* The same with the FORMAT chosen by the caller, as the reference's `framework` argument does (quad.py:452-466: "MONO" ->
 * one [N][23] row, "MODUL" -> [N][15] + [N][3], on whichever wrapper class the method is called): format = QR_KIND_COUPLED
 * (MONO; obs1 unused) or QR_KIND_DECOUPLED (MODUL; obs1 required), independent of env->kind.  env->kind must still be COUPLED
 * or DECOUPLED (the integrator terms live there; a bare QuadEnv raises AttributeError in the reference): QR_E_KIND otherwise.
 * Same side effect on the integrators. */
int qr_error_obs_format(const QrEnv* env, int32_t format, float* obs0, float* obs1, void* stream);

/* Replaces QuadEnv.reset(env_type) (quad.py:171-222, 338-404) for every env with
 * mask[i] != 0 (mask NULL = all): draws UDM parameters (unless QR_FLAG_NO_UDM / no params
 * buffer), the initial error state, R = Rz(yaw)Ry(pitch)Rx(roll), zeroes the integrators
 * and step counters and bumps the episode counter.  Counter-based Philox4x32-10 keyed by
 * (seed, env_offset+i, episode): results do not depend on sharding or launch geometry. */
int qr_reset(const QrEnv* env, const uint8_t* mask, void* stream);

/* Replaces QuadEnv.get_current_state() (quad.py:409-410): writes the reference's float64
 * 18-vector rows [N][18] = (x, v, vec_F(R(q)), W). */
int qr_get_state(const QrEnv* env, double* rows, void* stream);

/* Replaces assignment to QuadEnv.state: reads float64 rows [N][18] for envs with
 * mask[i] != 0 (NULL = all).  R is first passed through the reference's ensure_SO3 rule
 * (quad_utils.py:123-142) — and, because the internal attitude is a unit quaternion, always
 * projected onto SO(3) (nearest rotation, the same U V^T the reference's SVD yields).
 * A row whose attitude block has det R <= 0 or a non-finite entry has no nearest rotation: it is
 * REJECTED (that env keeps its previous state) and counted in *rejected (device int32, optional,
 * zero it before the call; the host wrapper raises when it is non-zero). */
int qr_set_state(const QrEnv* env, const double* rows, const uint8_t* mask, int32_t* rejected, void* stream);

/* Dry run of qr_set_state: counts in *rejected (device int32, required, zero it before the call) the rows qr_set_state would
 * reject — the same test on the same data — and writes NOTHING.  Lets a caller that injects states together with other
 * per-env data (integrator terms, parameters) refuse the whole injection before any of it is applied. */
int qr_check_state(const QrEnv* env, const double* rows, const uint8_t* mask, int32_t* rejected, void* stream);

/* Replaces TrajectoryGenerator.mark_traj_start(state) (utils/trajectory_generator.py:176-204)
 * plus the episode-start branch of calculate_desired (mode 0 :141-148: b1d = Rz(theta) b1_proj,
 * theta ~ U(+-25 deg); mode 1 :253-266: x_init, t_traj ~ U(2,5) s, w_b1d ~ U(+-0.15 pi) rad/s)
 * for envs with mask[i] != 0 (NULL = all), from the CURRENT state.  `draws` (optional, device,
 * [3][N] = theta_b1d [rad], t_traj [s], w_b1d [rad/s]) injects the random draws; NULL draws them
 * from the env's Philox stream.  Requires env->traj and goal_mode != QR_GOAL_EXTERNAL. */
int qr_traj_start(const QrEnv* env, const uint8_t* mask, const float* draws, void* stream);

/* Replaces TrajectoryGenerator.get_desired(state, mode) (trajectory_generator.py:113-173) for
 * the CURRENT state of envs with mask[i] != 0 (NULL = all): advances the per-env call counter
 * (t += dt) and writes rows [N][15] = (xd, vd, b1d, b1d_dot, Wd), Wd = (0, 0, b3 . (b1c x b1c_dot)) (:165-172).  With
 * store_goal != 0 the result also becomes the env's goal buffer (set_goal_state).  The fused
 * path (goal_mode != QR_GOAL_EXTERNAL in qr_step / qr_rollout) does the same at the start of every env-step
 * and after an in-launch reset, exactly as main.py:145-147,226-229 call it. */
int qr_get_desired(const QrEnv* env, const uint8_t* mask, float* rows, int32_t store_goal, void* stream);

/* Replaces the GAE loop of the reference's PPO (algos/ppo/ppo.py:134-146) for a [T][M] rollout
 * (M = num_envs * n_agents columns, each an independent time series):
 *     delta_t = r_t + gamma * Vnext_t * (1 - done_t) - V_t
 *     A_t     = delta_t + gamma * lambda * (1 - done_t) * A_{t+1},   A_T = 0   (reverse scan)
 *     target_t = A_t + V_t
 *   reward [T][M] f32, done [T][M] u8, value [T][M] f32 (or [T+1][M] when next_value is NULL:
 *   then Vnext_t = value[t+1], row T being the bootstrap value), next_value [T][M] f32 or NULL.
 *   Outputs advantage, td_target [T][M] f32 and, if `partials` != NULL, per-workgroup
 *   (sum, sum of squares) of the advantages as double [grid][2] (grid = ceil(M/64)) for the
 *   normalisation (adv - mean)/(std + 1e-4) (ppo.py:147); summing them on the host side of the
 *   launch is deterministic (no atomics). */
int qr_gae(const float* reward, const uint8_t* done, const float* value, const float* next_value,
           int32_t n_steps, int64_t n_cols, float gamma, float lam,
           float* advantage, float* td_target, double* partials, void* stream);

/* Host-side helpers (no device work). */
void qr_default_coeffs(QrCoeffs* c);
int  qr_abi_version(void);
/* What the launcher would run for this env — host-side, no device work; nothing in the reference (profilers, autotuners, tests).
 * qr_step / qr_rollout (actor = 0) or qr_rollout_actor (actor = 1: PPO / TD3-form actors, 2: the general form with SAC's
 * log_std head) over n_steps env-steps of `substeps` substeps: `launches` launches of `grid` workgroups (64-env tiles)
 * of `block` threads — 64 = one wavefront per tile, 128 = plus a helper wavefront (QR_FLAG_AUTO_RESET, default layout, grids
 * in the launch-latency regime; launches > 1: qr_rollout_actor in chunks of `grid` resident tiles) — of the instantiation
 * `name` = qr::step_kernel<KIND, XV, QW, 64, TRAJ, ADAPT, POLICY, SINGLE, HELP, HREW, MAG> with the values filled in (the prefix of
 * the kernel's name in a rocprofv3 trace; MAG = 1: the Magnus substep, substeps >= 2 in the default layout).  `key` = layout << 16 |
 * MAG << 12 | kind << 8 | TRAJ | ADAPT << 2 | POLICY << 3 | SINGLE << 5 | HELP << 6 | HREW << 7: what qr_launch_stats counts under. */
typedef struct QrLaunchPlan {
  int32_t grid, block, launches;
  int32_t traj, adapt, policy, single, help, hrew, mag;
  uint32_t key;
  char name[96];
} QrLaunchPlan;
int qr_launch_plan(const QrEnv* env, int32_t n_steps, int32_t substeps, int32_t actor, QrLaunchPlan* plan);
/* The older, shorter form: kernel family name + launch geometry of qr_step (n_steps = 1) / qr_rollout with ONE substep.
 * "" if the env descriptor is invalid. */
const char* qr_step_kernel_info(const QrEnv* env, int32_t n_steps, int32_t* grid, int32_t* block);
/* Host-side launch counters of this process: how many step-kernel launches each (layout, kind, instantiation) `key` has had
 * since load (or since the last call with reset != 0).  Writes up to `capacity` (key, count) pairs with count > 0, returns
 * how many there are.  qr_instance_table lists the keys of EVERY instantiation the library holds (175), same convention:
 * together they tell a test suite which kernels it really ran. */
int32_t qr_launch_stats(uint32_t* keys, uint32_t* counts, int32_t capacity, int32_t reset);
int32_t qr_instance_table(uint32_t* keys, int32_t capacity);

/* The step's memory traffic and nothing else: one launch that moves, per env, exactly what qr_step moves — state in and out,
 * parameters, the action row, [goal], [integrator words in and out], [observation rows], reward and done rows — with no
 * arithmetic.  The yardstick a step's time is priced against (bench.py: roofline.noop_kernel_us, measured live on the box
 * and the buffers of the run).  The env's state is written back as read, bit for bit; the output rows hold zeros afterwards. */
int qr_touch(const QrEnv* env, const float* action, const QrStepOut* out, void* stream);

/* The launch rule's thresholds in force in this process (compiled-in defaults or the QR_HELPER_GRID* environment variables), in
 * 64-env tiles: grids up to *step_quad (Quad-v0) / *step_wrappers (Coupled, Decoupled) tiles run qr_step with a helper wavefront
 * per tile, grids up to *rollout tiles run qr_rollout / qr_rollout_actor with one.  What a host-side autotuner needs to know
 * whether a grid is close enough to a crossover to be worth timing (QuadVecEnv: within +-25 %).  NULL pointers are skipped. */
void qr_launch_thresholds(int32_t* step_quad, int32_t* step_wrappers, int32_t* rollout);

#ifdef __cplusplus
}
#endif
#endif /* QUADROTOR_HIP_H */
