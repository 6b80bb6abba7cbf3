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
  const float *ls_w, *ls_b;  // optional state-dependent log_std head (SAC); log_std is then unused
  int32_t squash;            // QR_ACTOR_TANH_MEAN | QR_ACTOR_TANH_SAMPLE
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

}  // namespace qr
