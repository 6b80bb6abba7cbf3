// touch_cabi_microbench.cpp — qr_touch and qr_step through the C-ABI on hipMalloc'ed buffers (no torch, no Python): does the
// allocator explain why qr_touch on torch tensors (28.9 us at 1 M envs) trails the stand-alone microbenchmark (24.7 us)?
//   hipcc -O3 --offload-arch=gfx950 -Iinclude -o /tmp/touch_cabi tools/touch_cabi_microbench.cpp -Lgym_rotor_amd -lquadrotor_hip -Wl,-rpath,$PWD/gym_rotor_amd
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "quadrotor_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename F>
static double time_chain(hipStream_t s, F launch, int K = 100, int R = 15) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int k = 0; k < K; ++k) launch(k);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  std::vector<double> ts;
  for (int r = 0; r < R; ++r) {
    CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e0, s));
    CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ts.push_back(ms * 1e3 / K);
  }
  std::sort(ts.begin(), ts.end());
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return ts[ts.size() / 2];
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  printf("{\n \"what\": \"us per launch through the C-ABI on hipMalloc'ed buffers, chain of 100 launches in one hipGraph, median of 15; 8 action slabs\"");
  for (int n : {65536, 131072, 1048576}) {
    const int NA = 8;
    float *pv, *prm, *rew, *act; double* ar; uint8_t* done; int32_t *episode, *rc;
    CK(hipMalloc(&pv, (size_t)6 * n * 4)); CK(hipMalloc(&ar, (size_t)6 * n * 8)); CK(hipMalloc(&prm, (size_t)6 * n * 4));
    CK(hipMalloc(&act, (size_t)NA * n * 16)); CK(hipMalloc(&done, n)); CK(hipMalloc(&rew, (size_t)n * 4));
    CK(hipMalloc(&episode, (size_t)n * 4)); CK(hipMalloc(&rc, (size_t)(n / 64) * 4));
    CK(hipMemset(pv, 0, (size_t)6 * n * 4)); CK(hipMemset(ar, 0, (size_t)6 * n * 8)); CK(hipMemset(act, 0, (size_t)NA * n * 16));
    CK(hipMemset(episode, 0, (size_t)n * 4)); CK(hipMemset(rc, 0, (size_t)(n / 64) * 4));
    QrEnv e; memset(&e, 0, sizeof(e));
    qr_default_coeffs(&e.coeffs);
    e.kind = QR_KIND_QUAD; e.layout = QR_LAYOUT_MIXED; e.num_envs = n; e.seed = 1;
    e.pos_vel = pv; e.att_rate = ar; e.params = prm; e.episode = episode; e.reset_count = rc; e.flags = QR_FLAG_AUTO_RESET;
    QrStepOut o; memset(&o, 0, sizeof(o));
    o.reward = rew; o.done = done;
    if (qr_reset(&e, nullptr, s) != 0) { fprintf(stderr, "qr_reset failed\n"); return 1; }
    CK(hipStreamSynchronize(s));
    double us = time_chain(s, [&](int k) { if (qr_touch(&e, act + (size_t)(k % NA) * n * 4, &o, s)) exit(2); });
    printf(",\n \"qr_touch %d\": %.3f", n, us);
    us = time_chain(s, [&](int k) { if (qr_step(&e, act + (size_t)(k % NA) * n * 4, 1, &o, s)) exit(3); });
    printf(",\n \"qr_step %d\": %.3f", n, us);
    us = time_chain(s, [&](int k) { if (qr_touch(&e, act + (size_t)(k % NA) * n * 4, &o, s)) exit(2); });
    printf(",\n \"qr_touch %d again\": %.3f", n, us);
    CK(hipFree(pv)); CK(hipFree(ar)); CK(hipFree(prm)); CK(hipFree(act)); CK(hipFree(done)); CK(hipFree(rew)); CK(hipFree(episode)); CK(hipFree(rc));
  }
  printf("\n}\n");
  return 0;
}
