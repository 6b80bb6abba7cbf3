// resident_pacing_microbench.hip — the "boundary-free" design point for env.step() with an EXTERNAL per-step action producer.
//
// Question: qr_step pays a launch boundary per env-step (4.2 us per launch at 65 536 envs, of
// which ~1.6 us is the bare boundary between two dependent kernels).  qr_rollout has no boundaries (1.5 us per env-step) but needs
// all actions up front.  In between: a horizon-long RESIDENT step kernel that, per env-step, waits on a device flag which the
// kernel that produced this step's actions sets in its epilogue.  What would that cost per step?
//
// This program measures the mechanism in isolation, without the product's arithmetic in the way:
//   resident kernel   1024 workgroups x 128 threads (the 65 536-env launch shape), T steps; per step: lane 0 of every wave polls
//                     flag[t] (agent-scope relaxed load + s_sleep), acquires, the wave loads its 16 B/lane "action" row (written by
//                     the producer), runs `work` dependent FMAs per lane (440 ~ the stepping wave's chain), stores 8 B/lane.
//   producer kernel   one launch per step on ANOTHER stream (a hipGraph of T launches): 256 workgroups write the step's 1 MiB
//                     action slab; the last workgroup to finish (device counter) publishes flag[t] = 1 with a release.
// Reported: us per step seen by the resident kernel (s_memrealtime stamps of workgroup 0, median step-to-step interval), the
// producer chain's own period without a consumer, and the resident kernel's period when every flag is already set (= a rollout).
// Every spin is bounded (a wave gives up after ~50 ms and marks the run as failed).
//
//   hipcc -O3 --offload-arch=gfx950 -o build/evidence/resident_mb tools/resident_pacing_microbench.hip && build/evidence/resident_mb
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4_t __attribute__((ext_vector_type(4)));

template <bool FENCE>
__global__ __launch_bounds__(128) void resident(const uint32_t* flags, const float4* actions, float2* out, unsigned long long* stamps,
                                                 uint32_t* failed, int T, int n, int work) {
  const int i = blockIdx.x * 128 + threadIdx.x;
  float acc = (float)i * 1e-9f;
  for (int t = 0; t < T; ++t) {
    // wait for the step's actions: one lane polls, the wave follows; a wave that waited ~50 ms in vain marks the run as failed,
    // and every wave leaves as soon as it sees that mark
    int give_up = 0;
    if ((threadIdx.x & 63) == 0) {
      int spins = 0;
      while (__hip_atomic_load(flags + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 1023) == 0 && __hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { give_up = 1; break; }
        if (spins > (1 << 17)) { __hip_atomic_store(failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); give_up = 1; break; }
      }
    }
    if (__builtin_amdgcn_readfirstlane(give_up)) return;
    __builtin_amdgcn_wave_barrier();
    f4_t a;
    if (FENCE) {  // variant A: an agent-scope acquire per step (invalidates this CU's L1), then a plain load
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      a = *reinterpret_cast<const f4_t*>(actions + (size_t)t * n + i);
    } else {      // variant B: no fence — the handed-off rows are READ with sc1 loads (and were written with sc1 stores)
      asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(a) : "v"(actions + (size_t)t * n + i) : "memory");
    }
    float x = a.x + a.y + a.z + a.w + acc;
    for (int k = 0; k < work; ++k) x = fmaf(x, 0.999999f, 1e-7f);   // the step's dependent chain
    acc = x;
    out[(size_t)(t & 7) * n + i] = make_float2(x, (float)t);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      unsigned long long ts;
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts) : "v"(x) : "memory");
      stamps[t] = ts;
    }
  }
}

template <bool FENCE>
__global__ __launch_bounds__(256) void producer(uint32_t* flags, uint32_t* counters, float4* actions, int t, int n, float seed) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const f4_t v = {seed, seed * 0.5f, -seed, 0.25f};
  if (FENCE) {  // variant A: plain stores + __threadfence() (an L2 write-back per thread)
    if (i < n) *reinterpret_cast<f4_t*>(actions + (size_t)t * n + i) = v;
    __threadfence();
  } else {      // variant B: written through (sc1) and drained; no fence
    if (i < n) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" ::"v"(actions + (size_t)t * n + i), "v"(v) : "memory");
  }
  __syncthreads();                       // this workgroup's rows are visible device-wide before ...
  if (threadIdx.x == 0) {
    const uint32_t done = __hip_atomic_fetch_add(counters + t, 1u, FENCE ? __ATOMIC_ACQ_REL : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (done == gridDim.x) __hip_atomic_store(flags + t, 1u, FENCE ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... the last one publishes
  }
}

static double median_interval_us(const std::vector<unsigned long long>& st, int from) {
  std::vector<double> d;
  for (size_t t = from + 1; t < st.size(); ++t) d.push_back((double)(st[t] - st[t - 1]) * 0.01);
  std::sort(d.begin(), d.end());
  return d[d.size() / 2];
}

template <bool FENCE>
static int run(const char* name) {
  const int n = 65536, T = 400, work = 440;
  hipStream_t s_env, s_prod;
  CK(hipStreamCreateWithFlags(&s_env, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s_prod, hipStreamNonBlocking));
  uint32_t *flags, *counters, *failed; float4* actions; float2* out; unsigned long long* stamps;
  CK(hipMalloc(&flags, T * 4)); CK(hipMalloc(&counters, T * 4)); CK(hipMalloc(&failed, 4));
  CK(hipMalloc(&actions, (size_t)T * n * 16)); CK(hipMalloc(&out, (size_t)8 * n * 8)); CK(hipMalloc(&stamps, T * 8));
  CK(hipMemset(actions, 0, (size_t)T * n * 16));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s_prod, hipStreamCaptureModeGlobal));
  for (int t = 0; t < T; ++t) { hipLaunchKernelGGL(producer<FENCE>, dim3(n / 256), dim3(256), 0, s_prod, flags, counters, actions, t, n, 1.0f + t); CK(hipGetLastError()); }
  CK(hipStreamEndCapture(s_prod, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<unsigned long long> st(T);
  uint32_t hfailed = 0;
  printf(" \"%s\": {\n", name);
  for (int rep = 0; rep < 2; ++rep) {   // (a) producer chain alone: its launch-to-launch period
    CK(hipMemsetAsync(flags, 0, T * 4, s_prod)); CK(hipMemsetAsync(counters, 0, T * 4, s_prod));
    CK(hipEventRecord(e0, s_prod)); CK(hipGraphLaunch(ge, s_prod)); CK(hipEventRecord(e1, s_prod)); CK(hipStreamSynchronize(s_prod));
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("  \"producer chain alone (one launch per step: 1 MiB of actions + flag)\": %.3f,\n", ms * 1e3 / T);
  CK(hipMemset(failed, 0, 4));
  for (int rep = 0; rep < 2; ++rep) {   // (b) every flag already set: the resident kernel free-runs (a rollout)
    CK(hipEventRecord(e0, s_env));
    hipLaunchKernelGGL(resident<FENCE>, dim3(n / 128), dim3(128), 0, s_env, flags, actions, out, stamps, failed, T, n, work); CK(hipGetLastError());
    CK(hipEventRecord(e1, s_env)); CK(hipStreamSynchronize(s_env));
  }
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipMemcpy(st.data(), stamps, T * 8, hipMemcpyDeviceToHost));
  printf("  \"resident kernel, all flags set in advance (event time / T, stamp median)\": [%.3f, %.3f],\n", ms * 1e3 / T, median_interval_us(st, 10));
  for (int rep = 0; rep < 3; ++rep) {   // (c) paced: the resident kernel starts first and waits; the producer graph follows on the other stream
    CK(hipMemset(flags, 0, T * 4)); CK(hipMemset(counters, 0, T * 4)); CK(hipMemset(failed, 0, 4));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, s_env));
    hipLaunchKernelGGL(resident<FENCE>, dim3(n / 128), dim3(128), 0, s_env, flags, actions, out, stamps, failed, T, n, work); CK(hipGetLastError());
    CK(hipEventRecord(e1, s_env));
    CK(hipGraphLaunch(ge, s_prod));
    CK(hipStreamSynchronize(s_prod)); CK(hipStreamSynchronize(s_env));
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(st.data(), stamps, T * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hfailed, failed, 4, hipMemcpyDeviceToHost));
    printf("  \"paced by the producer, run %d (event time / T, stamp median of steps 50.., gave up waiting: %u)\": [%.3f, %.3f]%s\n", rep, hfailed,
           ms * 1e3 / T, median_interval_us(st, 50), rep < 2 ? "," : "");
  }
  printf(" },\n");
  CK(hipFree(flags)); CK(hipFree(counters)); CK(hipFree(failed)); CK(hipFree(actions)); CK(hipFree(out)); CK(hipFree(stamps));
  return hfailed ? 2 : 0;
}

int main() {
  printf("{\n \"what\": \"resident step kernel (1024 x 128 threads, 440 dependent FMAs per step) paced per step by device flags; us per step\",\n");
  int rc = run<true>("hand-over by fences (__threadfence in the producer, agent-scope acquire per step in the consumer)");
  rc |= run<false>("hand-over by write-through: sc1 stores drained before the flag, sc1 loads behind it, no fence");
  printf(" \"T\": 400\n}\n");
  return rc;
}
