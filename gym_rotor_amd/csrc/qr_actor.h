// qr_actor.h — part of the gfx950 quadrotor step library (included by quadrotor_kernels.hip, in this order).
// The reference's MLP actors (PPO, TD3, SAC) in the loop (qr_rollout_actor): MFMA actor, VALU/LDS actor, sampling, normals.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "quadrotor_hip.h"
#include "qr_traj.h"

namespace qr {

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
                       O_MW = O_FC2B + HP, O_MB = O_MW + H * AP, O_LS = O_MB + AP, O_LW = O_LS + AP, O_LB = O_LW + H * AP,
                       SIZE = O_LB + AP;

  __device__ static void fill(float* sm, const ActorW& p, int tid) {  // sm[k][j] = W[j][k]
    for (int i = tid; i < D * HP; i += 64) { const int k = i / HP, j = i - k * HP; sm[O_FC1W + i] = j < H ? p.fc1_w[j * D + k] : 0.0f; }
    for (int i = tid; i < H * HP; i += 64) { const int k = i / HP, j = i - k * HP; sm[O_FC2W + i] = j < H ? p.fc2_w[j * H + k] : 0.0f; }
    for (int i = tid; i < H * AP; i += 64) { const int k = i / AP, j = i - k * AP; sm[O_MW + i] = j < A ? p.mean_w[j * H + k] : 0.0f; }
    if (tid < HP) { sm[O_FC1B + tid] = tid < H ? p.fc1_b[tid] : 0.0f; sm[O_FC2B + tid] = tid < H ? p.fc2_b[tid] : 0.0f; }
    if (tid < AP) { sm[O_MB + tid] = tid < A ? p.mean_b[tid] : 0.0f; sm[O_LS + tid] = (tid < A && p.log_std) ? p.log_std[tid] : 0.0f; }
    if (p.ls_w) {
      for (int i = tid; i < H * AP; i += 64) { const int k = i / AP, j = i - k * AP; sm[O_LW + i] = j < A ? p.ls_w[j * H + k] : 0.0f; }
      if (tid < AP) sm[O_LB + tid] = tid < A ? p.ls_b[tid] : 0.0f;
    }
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

  // pre[] = mean head before any squashing; ls[] = log_std (parameter, or the head's output)
  __device__ __forceinline__ static void heads(const float* sm, bool ls_head, const float (&x)[D], float (&pre)[A], float (&ls)[A]) {
    float h1[H], h2[H];
    layer<D, H, HP>(sm + O_FC1W, sm + O_FC1B, x, h1);
#pragma unroll
    for (int j = 0; j < H; ++j) h1[j] = fmaxf(h1[j], 0.0f);
    layer<H, H, HP>(sm + O_FC2W, sm + O_FC2B, h1, h2);
#pragma unroll
    for (int j = 0; j < H; ++j) h2[j] = fmaxf(h2[j], 0.0f);
    layer<H, A, AP>(sm + O_MW, sm + O_MB, h2, pre);
    if (ls_head) {
      layer<H, A, AP>(sm + O_LW, sm + O_LB, h2, ls);
    } else {
#pragma unroll
      for (int j = 0; j < A; ++j) ls[j] = sm[O_LS + j];
    }
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
//   layer 3 (4 outputs: a 16 x 16 tile would be three quarters zeros — 512 clocks of the matrix pipe per head, measured
//            3.84 against 3.69 us per env-step of the PPO collection step, profiles/r05/ab_l3_blocks.txt): the 16-block
//            instruction v_mfma_f32_4x4x1 on the lane's own h2 values, then a transpose-reduce over the four 16-lane
//            rows with v_permlane16_swap / v_permlane32_swap; lane (g, c) ends with mean[f] of env 16 g + c — its own env.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// GENERAL = false: the PPO / TD3 form only (no log_std head, tanh-of-mean rule) — what the launcher
// picks when no actor needs more, so that the common path carries neither the second head's registers
// nor the rule's branches.
template <int D, bool GENERAL>  // obs_dim 23 (COUPLED) or 15 (DECOUPLED agent 1); hidden 16, 4 actions
struct ActorMfma {
  static constexpr int KS = (D + 3) / 4;
  float a1[KS], a2[4], w3[4], w3s[4], bias1[4], bias2[4], bias3[4];
  float ls_const[4];  // the log_std head's bias (ls_head) or the state-independent log_std parameter: one of the two exists per actor
  bool ls_head;

  __device__ __forceinline__ void load(const ActorW& p, int lane) {
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int s = 0; s < KS; ++s) a1[s] = (4 * s + g < D) ? p.fc1_w[c * D + 4 * s + g] : 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) { a2[s] = p.fc2_w[c * 16 + 4 * g + s]; w3[s] = p.mean_w[(c & 3) * 16 + 4 * g + s]; }
    ls_head = GENERAL && p.ls_w != nullptr;  // wave-uniform
#pragma unroll
    for (int s = 0; s < 4; ++s) w3s[s] = ls_head ? p.ls_w[(c & 3) * 16 + 4 * g + s] : 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bias1[r] = p.fc1_b[4 * g + r]; bias2[r] = p.fc2_b[4 * g + r]; bias3[r] = p.mean_b[r];
      ls_const[r] = ls_head ? p.ls_b[r] : (p.log_std ? p.log_std[r] : 0.0f);
    }
  }

  // xs: LDS tile [64 envs][D] of the wave's observations (row = lane)
  // pre[] = mean head before any squashing; ls[] = log_std (parameter, or the second head's output)
  __device__ __forceinline__ void heads(const float* xs, int lane, float (&pre)[4], float (&ls)[4]) const {
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
    // Layer 3 on the 16-block instruction v_mfma_f32_4x4x1 (8 clocks of the matrix pipe against 32): a block is four neighbouring
    // lanes (one hidden group g, four envs), A = W3[f = lane & 3][4 g + s] (the same w3[s]), B = the lane's own relu(h2[b][s]):
    // D[f] of lane (g, c) = the partial sum over hidden group g for env 16 b + c.  The sum over g and the move of block b's
    // result to row g' = b is a 4 x 4 transpose-reduce over the wave's four 16-lane rows: v_permlane16_swap exchanges the odd
    // rows of one register with the even rows of another, v_permlane32_swap the upper half with the lower half.
    auto head = [&](const float (&w)[4], const float (&bias)[4], float (&out)[4]) {
      f32x4 P[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) P[b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int s = 0; s < 4; ++s) {  // (four independent chains, interleaved)
#pragma unroll
        for (int b = 0; b < 4; ++b) P[b] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[s], fmaxf(h2[b][s], 0.0f), P[b], 0, 0, 0);
      }
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const auto s01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(P[0][f]), __float_as_uint(P[1][f]), false, false);
        const auto s23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(P[2][f]), __float_as_uint(P[3][f]), false, false);
        const float q01 = __uint_as_float(s01[0]) + __uint_as_float(s01[1]);  // rows: P0 (g 0+1), P1 (g 0+1), P0 (g 2+3), P1 (g 2+3)
        const float q23 = __uint_as_float(s23[0]) + __uint_as_float(s23[1]);  //       P2 (g 0+1), P3 (g 0+1), P2 (g 2+3), P3 (g 2+3)
        const auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(q01), __float_as_uint(q23), false, false);
        out[f] = bias[f] + (__uint_as_float(t[0]) + __uint_as_float(t[1]));   // row b: block b summed over the four g
      }
    };
    head(w3, bias3, pre);
    if (ls_head) {
      head(w3s, ls_const, ls);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) ls[r] = ls_const[r];
    }
  }
};

// Action selection from the heads' outputs.
//   QR_ACTOR_TANH_MEAN   PPO.choose_action (ppo.py:93-101), TD3.choose_action (td3.py:93-96):
//     mean = tanh(pre); a = clamp(mean + exp(log_std) eps, +-max_action); logp = per-component
//     Normal(mean, std).log_prob of the CLAMPED action (ppo.py:97-98).
//   QR_ACTOR_TANH_SAMPLE MLP_Actor_SAC.sample (sac_mlp.py:60-82): log_std clamped to [-20, 2];
//     u = pre + exp(log_std) eps; a = tanh(u); logp = Normal(pre, std).log_prob(u) - log(1 - a^2 + 1e-6).
template <int A, bool GENERAL>
__device__ __forceinline__ void actor_sample(int squash, bool ls_head, const float* pre, const float* log_std, const float* eps,
                                             bool deterministic, float max_action, float* act, float* logp) {
#pragma unroll
  for (int j = 0; j < A; ++j) {
    if (GENERAL && squash == QR_ACTOR_TANH_SAMPLE) {
      const float ls = fminf(fmaxf(log_std[j], -20.0f), 2.0f);
      const float sd = __expf(ls);
      const float z = deterministic ? 0.0f : eps[j];
      const float aj = tanh_fast(fmaf(sd, z, pre[j]));
      act[j] = aj;
      logp[j] = fmaf(-0.5f * z, z, -ls - 0.91893853320467274f) - __logf(fmaf(-aj, aj, 1.0f) + 1e-6f);
    } else {
      const float ls = (GENERAL && ls_head) ? fminf(fmaxf(log_std[j], -20.0f), 2.0f) : log_std[j];
      const float sd = __expf(ls);
      const float mean = tanh_fast(pre[j]);
      const float raw = deterministic ? mean : fmaf(sd, eps[j], mean);
      const float aj = fminf(fmaxf(raw, -max_action), max_action);
      const float z = (aj - mean) * __expf(-ls);
      act[j] = aj;
      logp[j] = fmaf(-0.5f * z, z, -ls - 0.91893853320467274f);
    }
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

}  // namespace qr
