// first_load_microbench.hip — how long the FIRST vector load of a wave takes in a chain of dependent launches, on gfx950.
//
// Each launch: 1024 workgroups x 64 lanes; every wave (optionally after idling `delay` x 64 clocks with s_sleep) issues
// `nloads` coalesced 8-byte loads of a buffer the PREVIOUS launch wrote, stamps s_memrealtime (100 MHz) at entry, at the
// issue of the loads and at their arrival, then writes the buffer back.  Printed: median over waves (of the last launch
// of a 200-launch graph) of entry -> issue, issue -> arrival, relative to the earliest wave's entry.
// Question answered: is the ~1 us that the step kernel's working set takes to arrive a property of the bytes
// (bandwidth), or of the launch boundary (loads issued in a kernel's first instructions wait for the cache
// invalidation that the dispatch started)?
//
//   hipcc -O3 --offload-arch=gfx950 -o build/first_load_mb tools/first_load_microbench.hip && build/first_load_mb
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NL, bool SC1>
__global__ __launch_bounds__(64) void probe(double* __restrict__ buf, unsigned long long* __restrict__ stamps, int n, int delay) {
  unsigned long long t0, t1, t2;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(1);
  const int i = blockIdx.x * 64 + threadIdx.x;
  double v[NL];
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
#pragma unroll
  for (int f = 0; f < NL; ++f) v[f] = buf[(size_t)f * n + i];
  double s = 0;
#pragma unroll
  for (int f = 0; f < NL; ++f) s += v[f];
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2) : "v"(s) : "memory");
#pragma unroll
  for (int f = 0; f < NL; ++f) {
    if constexpr (SC1) __hip_atomic_store(reinterpret_cast<uint64_t*>(buf + (size_t)f * n + i), __builtin_bit_cast(uint64_t, v[f] + 1.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else buf[(size_t)f * n + i] = v[f] + 1.0;
  }
  if (threadIdx.x == 0) { stamps[blockIdx.x * 4 + 0] = t0; stamps[blockIdx.x * 4 + 1] = t1; stamps[blockIdx.x * 4 + 2] = t2; }
}

template <int NL, bool SC1>
static void run(hipStream_t s, int delay, int grid) {
  const int n = grid * 64;
  double* buf; unsigned long long* st;
  CK(hipMalloc(&buf, (size_t)NL * n * 8)); CK(hipMemset(buf, 0, (size_t)NL * n * 8));
  CK(hipMalloc(&st, (size_t)grid * 4 * 8));
  hipGraph_t g; hipGraphExec_t ge;
  const int K = 200;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int k = 0; k < K; ++k) { hipLaunchKernelGGL((probe<NL, SC1>), dim3(grid), dim3(64), 0, s, buf, st, n, delay); CK(hipGetLastError()); }
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
  CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h((size_t)grid * 4);
  CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long first = ~0ull;
  for (int b = 0; b < grid; ++b) first = std::min(first, h[b * 4]);
  std::vector<double> entry, issue, lat, arrive;
  for (int b = 0; b < grid; ++b) {
    entry.push_back((h[b * 4] - first) * 0.01); issue.push_back((h[b * 4 + 1] - first) * 0.01);
    lat.push_back((h[b * 4 + 2] - h[b * 4 + 1]) * 0.01); arrive.push_back((h[b * 4 + 2] - first) * 0.01);
  }
  auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  auto mx = [](std::vector<double>& v) { return *std::max_element(v.begin(), v.end()); };
  printf("  {\"grid\": %d, \"loads_per_wave\": %d, \"bytes_per_launch_MB\": %.2f, \"stores\": \"%s\", \"idle_before_loads_us_nominal\": %.2f, \"us_per_launch\": %.2f, "
         "\"entry_median\": %.2f, \"issue_median\": %.2f, \"issue_to_arrival_median\": %.2f, \"arrival_median\": %.2f, \"arrival_max\": %.2f},\n",
         grid, NL, NL * n * 8 / 1e6, SC1 ? "sc1" : "plain", delay * 64 / 2400.0, ms * 1e3 / K, med(entry), med(issue), med(lat), med(arrive), mx(arrive));
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipFree(buf)); CK(hipFree(st));
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  printf("[\n");
  for (int delay : {0, 20, 40, 80}) { run<1, false>(s, delay, 1024); run<1, true>(s, delay, 1024); }
  for (int delay : {0, 40}) { run<4, false>(s, delay, 1024); run<12, false>(s, delay, 1024); run<12, true>(s, delay, 1024); }
  run<1, false>(s, 0, 64); run<12, false>(s, 0, 64); run<12, false>(s, 0, 256);
  printf("  {}\n]\n");
  return 0;
}
