// touch_variants_microbench.hip — why does qr_touch (28.7 us at 1 M envs) trail the stand-alone soa_kernel of
// tools/vmem_width_microbench.hip (25.1 us) on the same box and the same bytes?  Same chain-of-launches harness, 1 M and 131 072
// envs, 8 action slabs; variants of ONE kernel that differ in one thing each:
//   flat        global loads / stores through raw pointers, every store depends on every load           (= soa_kernel)
//   flat_indep  the same, the state written back as read (stores independent of the other loads)       (what qr_touch does)
//   buffer      flat_indep through buffer descriptors (raw_buffer_load / store: the library's SoA accessor)
//   bigarg      buffer + a 760-byte by-value struct behind the pointers (the library's Args block in the kernarg segment)
//   +predicate  ragged-tail clamp and an early exit for lanes past the batch; +state written back as read: the stores carry the loaded values
//   round #1 runs on non-zero data (hipMemset 0x3c), round #0 on zeros
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16 -o /tmp/touch_mb tools/touch_variants_microbench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define LAUNCH(...) do { hipLaunchKernelGGL(__VA_ARGS__); CK(hipGetLastError()); } while (0)

struct Big { char pad[760]; };
typedef int v2i_t __attribute__((ext_vector_type(2)));

template <int MODE>  // 0 flat, 1 flat_indep
__global__ __launch_bounds__(64) void flat_kernel(float* __restrict__ pv, double* __restrict__ ar, const float* __restrict__ prm,
                                                   const float4* __restrict__ act, uint8_t* __restrict__ done, float* __restrict__ rew, int n) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  float x[6]; double q[6]; float p[6];
#pragma unroll
  for (int f = 0; f < 6; ++f) q[f] = ar[(size_t)f * n + i];
#pragma unroll
  for (int f = 0; f < 6; ++f) x[f] = pv[(size_t)f * n + i];
#pragma unroll
  for (int f = 0; f < 6; ++f) p[f] = prm[(size_t)f * n + i];
  const float4 a = act[i];
  float s = (a.x + a.y + a.z + a.w) * 1e-6f + (p[0] + p[1] + p[2] + p[3] + p[4] + p[5]) * 1e-9f;
  if (MODE == 0) {
#pragma unroll
    for (int f = 0; f < 6; ++f) { x[f] += s; q[f] += (double)s; }
  }
#pragma unroll
  for (int f = 0; f < 6; ++f) ar[(size_t)f * n + i] = q[f];
#pragma unroll
  for (int f = 0; f < 6; ++f) pv[(size_t)f * n + i] = x[f];
  rew[i] = s;
  done[i] = s > 1.0f;
}

template <bool BIG, bool PRED = false, bool SAME = false>
__global__ __launch_bounds__(64) void buffer_kernel(float* pv, double* ar, const float* prm, const float4* act, uint8_t* done, float* rew, int n,
                                                     int ld, const Big big) {
  const unsigned first = blockIdx.x * 64u, lane0 = threadIdx.x;
  const int rows = min(n - (int)first, 64);
  const unsigned lane = PRED ? min(lane0, (unsigned)(rows - 1)) : lane0;
  auto rs = [](const void* b) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(b), 0, 0x7fffffff, 0x00020000); };
  const auto rpv = rs(pv), rar = rs(ar), rprm = rs(prm);
  float x[6]; double q[6]; float s = 0.f;
#pragma unroll
  for (int f = 0; f < 6; ++f) q[f] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rar, lane * 8u, ((unsigned)f * ld + first) * 8u, 0));
#pragma unroll
  for (int f = 0; f < 6; ++f) x[f] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rpv, lane * 4u, ((unsigned)f * ld + first) * 4u, 0));
#pragma unroll
  for (int f = 0; f < 6; ++f) s += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rprm, lane * 4u, ((unsigned)f * ld + first) * 4u, 0));
  const float4 a = act[first + lane];
  s += (a.x + a.y) + (a.z + a.w);
  s *= 0.0f;
  if (BIG) s += (float)big.pad[blockIdx.x % 760] * 0.0f;
  if (SAME) {
#pragma unroll
    for (int f = 0; f < 6; ++f) asm volatile("" : "+v"(q[f]), "+v"(x[f]));
  } else {
#pragma unroll
    for (int f = 0; f < 6; ++f) { x[f] += s + 1.0f; q[f] += (double)s + 1.0; }
  }
  if (PRED && (int)lane0 >= rows) return;
#pragma unroll
  for (int f = 0; f < 6; ++f) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i_t, q[f]), rar, lane * 8u, ((unsigned)f * ld + first) * 8u, 0);
#pragma unroll
  for (int f = 0; f < 6; ++f) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, x[f]), rpv, lane * 4u, ((unsigned)f * ld + first) * 4u, 0);
  rew[first + lane] = s;
  done[first + lane] = 0;
}

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
  printf("{\n \"what\": \"us per launch, chain of 100 dependent launches in one hipGraph, median of 15 (HIP events); 8 action slabs\"");
  for (int n : {131072, 1048576}) {
    const int tiles = n / 64, NA = 8;
    float *pv, *prm, *rew; double* ar; float4* act; uint8_t* done;
    CK(hipMalloc(&pv, (size_t)6 * n * 4)); CK(hipMalloc(&ar, (size_t)6 * n * 8)); CK(hipMalloc(&prm, (size_t)6 * n * 4));
    CK(hipMalloc(&act, (size_t)NA * n * 16)); CK(hipMalloc(&done, n)); CK(hipMalloc(&rew, (size_t)n * 4));
    CK(hipMemset(pv, 0, (size_t)6 * n * 4)); CK(hipMemset(ar, 0, (size_t)6 * n * 8)); CK(hipMemset(prm, 0, (size_t)6 * n * 4));
    CK(hipMemset(act, 0, (size_t)NA * n * 16));
    Big big{};
    for (int rep = 0; rep < 2; ++rep) {
      double us = time_chain(s, [&](int k) { LAUNCH(flat_kernel<0>, dim3(tiles), dim3(64), 0, s, pv, ar, prm, act + (size_t)(k % NA) * n, done, rew, n); });
      printf(",\n \"flat %d #%d\": %.3f", n, rep, us);
      us = time_chain(s, [&](int k) { LAUNCH(flat_kernel<1>, dim3(tiles), dim3(64), 0, s, pv, ar, prm, act + (size_t)(k % NA) * n, done, rew, n); });
      printf(",\n \"flat_indep %d #%d\": %.3f", n, rep, us);
      us = time_chain(s, [&](int k) { LAUNCH(buffer_kernel<false>, dim3(tiles), dim3(64), 0, s, pv, ar, prm, act + (size_t)(k % NA) * n, done, rew, n, n, big); });
      printf(",\n \"buffer %d #%d\": %.3f", n, rep, us);
      us = time_chain(s, [&](int k) { LAUNCH(buffer_kernel<true>, dim3(tiles), dim3(64), 0, s, pv, ar, prm, act + (size_t)(k % NA) * n, done, rew, n, n, big); });
      printf(",\n \"bigarg %d #%d\": %.3f", n, rep, us);
      us = time_chain(s, [&](int k) { LAUNCH((buffer_kernel<true, true>), dim3(tiles), dim3(64), 0, s, pv, ar, prm, act + (size_t)(k % NA) * n, done, rew, n, n, big); });
      printf(",\n \"bigarg+predicate %d #%d\": %.3f", n, rep, us);
      us = time_chain(s, [&](int k) { LAUNCH((buffer_kernel<true, true, true>), dim3(tiles), dim3(64), 0, s, pv, ar, prm, act + (size_t)(k % NA) * n, done, rew, n, n, big); });
      printf(",\n \"bigarg+predicate+state written back as read %d #%d\": %.3f", n, rep, us);
      if (rep == 0) {  // the second round runs on non-zero data
        CK(hipMemset(pv, 0x3c, (size_t)6 * n * 4)); CK(hipMemset(ar, 0x3c, (size_t)6 * n * 8)); CK(hipMemset(prm, 0x3c, (size_t)6 * n * 4));
        CK(hipMemset(act, 0x3c, (size_t)NA * n * 16));
      }
    }
    CK(hipFree(pv)); CK(hipFree(ar)); CK(hipFree(prm)); CK(hipFree(act)); CK(hipFree(done)); CK(hipFree(rew));
  }
  printf("\n}\n");
  return 0;
}
