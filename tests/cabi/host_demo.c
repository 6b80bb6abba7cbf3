/* A host that is NOT Python: plain C against include/quadrotor_hip.h and libquadrotor_hip.so (the drop-in boundary), device memory from
 * the HIP runtime's C API.  Resets N Quad-v0 envs (train distribution, seed from argv), steps them K times with a fixed action pattern
 * and in-launch resets, and prints the state rows of a few envs plus sums over all of them — tests/test_gpu_edge_cases.py runs the same
 * through QuadVecEnv and expects the same bits.
 *     gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/cabi/host_demo.c -Lgym_rotor_amd -lquadrotor_hip -L/opt/rocm/lib -lamdhip64 \
 *         -Wl,-rpath,$PWD/gym_rotor_amd -Wl,-rpath,/opt/rocm/lib -o host_demo        (gcc: the boundary needs no HIP compiler on the host side)
 *     ./host_demo <num_envs> <steps> <seed>
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "quadrotor_hip.h"

#define HCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define QCK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s -> %d\n", #x, r_); return 3; } } while (0)

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 4096;
  const int steps = argc > 2 ? atoi(argv[2]) : 50;
  const uint64_t seed = argc > 3 ? strtoull(argv[3], NULL, 10) : 7;
  const int64_t ld = (n + 3) / 4 * 4, tiles = (n + 63) / 64;
  if (qr_abi_version() != QR_ABI_VERSION) { fprintf(stderr, "ABI %d != header %d\n", qr_abi_version(), QR_ABI_VERSION); return 1; }

  QrEnv env;
  memset(&env, 0, sizeof env);
  qr_default_coeffs(&env.coeffs);
  env.kind = QR_KIND_QUAD; env.layout = QR_LAYOUT_MIXED;
  env.num_envs = n; env.field_stride = ld; env.seed = seed; env.flags = QR_FLAG_AUTO_RESET;
  float *params, *action, *reward;
  uint8_t* done;
  double* rows;
  HCK(hipMalloc(&env.pos_vel, 6 * ld * sizeof(float)));   HCK(hipMemset(env.pos_vel, 0, 6 * ld * sizeof(float)));
  HCK(hipMalloc(&env.att_rate, 6 * ld * sizeof(double))); HCK(hipMemset(env.att_rate, 0, 6 * ld * sizeof(double)));
  HCK(hipMalloc((void**)&params, 6 * ld * sizeof(float)));
  HCK(hipMalloc((void**)&env.episode, n * sizeof(int32_t)));         HCK(hipMemset(env.episode, 0, n * sizeof(int32_t)));
  HCK(hipMalloc((void**)&env.reset_count, tiles * sizeof(int32_t))); HCK(hipMemset(env.reset_count, 0, tiles * sizeof(int32_t)));
  HCK(hipMalloc((void**)&action, n * 4 * sizeof(float)));
  HCK(hipMalloc((void**)&reward, n * sizeof(float)));
  HCK(hipMalloc((void**)&done, n));
  HCK(hipMalloc((void**)&rows, n * 18 * sizeof(double)));
  env.params = params;
  {  /* nominal parameters until the first reset randomises them (quad.py:28-33) */
    float* h = (float*)malloc(6 * ld * sizeof(float));
    const float nom[6] = {2.15f, 0.23f, 0.022f, 0.035f, 0.0135f, 2.2f};
    for (int f = 0; f < 6; ++f) for (int64_t i = 0; i < ld; ++i) h[f * ld + i] = nom[f];
    HCK(hipMemcpy(params, h, 6 * ld * sizeof(float), hipMemcpyHostToDevice));
    free(h);
  }
  float* hact = (float*)malloc(n * 4 * sizeof(float));
  QrStepOut out;
  memset(&out, 0, sizeof out);
  out.reward = reward; out.done = done;

  {  /* QuadEnv.reset('train') for every env: the reset call itself runs without the in-launch-reset flag */
    QrEnv r = env;
    r.flags = 0;
    QCK(qr_reset(&r, NULL, NULL));
  }
  long total_done = 0;
  uint8_t* hdone = (uint8_t*)malloc(n);
  for (int t = 0; t < steps; ++t) {
    for (int64_t i = 0; i < n; ++i)
      for (int j = 0; j < 4; ++j) hact[i * 4 + j] = (float)((int)((i * 7 + j * 3 + t * 5) % 21) - 10) * 0.1f;   /* in [-1, 1], float32-exact pattern */
    HCK(hipMemcpy(action, hact, n * 4 * sizeof(float), hipMemcpyHostToDevice));
    QCK(qr_step(&env, action, 1, &out, NULL));
    HCK(hipMemcpy(hdone, done, n, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; ++i) total_done += hdone[i];
  }
  QCK(qr_get_state(&env, rows, NULL));
  double* h = (double*)malloc(n * 18 * sizeof(double));
  HCK(hipMemcpy(h, rows, n * 18 * sizeof(double), hipMemcpyDeviceToHost));
  double sums[18] = {0};
  for (int64_t i = 0; i < n; ++i) for (int j = 0; j < 18; ++j) sums[j] += h[i * 18 + j];
  printf("done %ld\n", total_done);
  for (int j = 0; j < 18; ++j) printf("sum %d %.17g\n", j, sums[j]);
  for (int64_t i = 0; i < n; i += n / 4 > 0 ? n / 4 : 1) for (int j = 0; j < 18; ++j) printf("row %lld %d %.17g\n", (long long)i, j, h[i * 18 + j]);
  return 0;
}
