// vmem_width_microbench.hip — what a launch of the 65 536-env step costs for its MEMORY INSTRUCTIONS alone, on gfx950.
//
// The Quad-v0 step moves 189 B per env.  In the SoA layout ([field][N]) a wave does that with 19 narrow loads
// (6 x b32 + 6 x b64 state, 6 x b32 params, 1 x b128 action) and 13 narrow stores; in a tile-blocked layout
// ([tile][16-byte group][lane]) the same bytes are 8 wide loads and 6 wide stores.  This program times both —
// and the empty kernel at several grid shapes — as chains of DEPENDENT launches inside one hipGraph, the way
// bench.py times qr_step.  Same bytes, same (trivial) arithmetic: what differs is the instruction count on the
// memory path of a wave that is alone on its SIMD.
//
//   hipcc -O3 --offload-arch=gfx950 -o build/vmem_mb tools/vmem_width_microbench.hip && build/vmem_mb > out.json
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
// every launch is checked — also inside a stream capture, where a launch that violates the kernel's launch bounds fails
// at once and would otherwise leave an EMPTY graph that "runs" in 0.02 us (round 3 reported two such rows as measurements)
#define LAUNCH(...) do { hipLaunchKernelGGL(__VA_ARGS__); CK(hipGetLastError()); } while (0)

__global__ __launch_bounds__(512) void empty_kernel(float* p) { if (p == nullptr) p[0] = 0; }

// ---- SoA, narrow accesses: the round-2 layout ----
__global__ __launch_bounds__(64) void soa_kernel(float* __restrict__ pv, double* __restrict__ ar, const float* __restrict__ prm,
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
  const float s = (a.x + a.y + a.z + a.w) * 1e-6f + (p[0] + p[1] + p[2] + p[3] + p[4] + p[5]) * 1e-9f;
#pragma unroll
  for (int f = 0; f < 6; ++f) { x[f] += s; q[f] += (double)s; }
#pragma unroll
  for (int f = 0; f < 6; ++f) ar[(size_t)f * n + i] = q[f];
#pragma unroll
  for (int f = 0; f < 6; ++f) pv[(size_t)f * n + i] = x[f];
  rew[i] = s;
  done[i] = s > 1.0f;
}

// ---- tile-blocked, 16-byte accesses: tile t = 64 envs; state block = 4 groups of 16 B + 1 of 8 B per lane ----
__global__ __launch_bounds__(64) void blk_kernel(float4* __restrict__ st, double* __restrict__ st8, const float4* __restrict__ prm4,
                                                  const float2* __restrict__ prm2, const float4* __restrict__ act, uint8_t* __restrict__ done,
                                                  float* __restrict__ rew, int n) {
  const int t = blockIdx.x, l = threadIdx.x, i = t * 64 + l;
  float4 g[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) g[k] = st[((size_t)t * 4 + k) * 64 + l];
  double w2 = st8[i];
  const float4 p4 = prm4[i];
  const float2 p2 = prm2[i];
  const float4 a = act[i];
  const float s = (a.x + a.y + a.z + a.w) * 1e-6f + (p4.x + p4.y + p4.z + p4.w + p2.x + p2.y) * 1e-9f;
#pragma unroll
  for (int k = 0; k < 4; ++k) { g[k].x += s; g[k].y += s; g[k].z += s; g[k].w += s; }
  w2 += (double)s;
#pragma unroll
  for (int k = 0; k < 4; ++k) st[((size_t)t * 4 + k) * 64 + l] = g[k];
  st8[i] = w2;
  rew[i] = s;
  done[i] = s > 1.0f;
}

// ---- tile-blocked, one 96-byte record per env incl. params (6 x 16 B), reward+done packed into one 8-byte store ----
__global__ __launch_bounds__(64) void rec_kernel(float4* __restrict__ st, const float4* __restrict__ act, float2* __restrict__ out, int n) {
  const int t = blockIdx.x, l = threadIdx.x, i = t * 64 + l;
  float4 g[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) g[k] = st[((size_t)t * 6 + k) * 64 + l];
  const float4 a = act[i];
  const float s = (a.x + a.y + a.z + a.w) * 1e-6f + (g[5].x + g[5].y + g[5].z + g[5].w + g[4].z + g[4].w) * 1e-9f;
#pragma unroll
  for (int k = 0; k < 5; ++k) { g[k].x += s; g[k].y += s; g[k].z += (k < 4 ? s : 0.f); g[k].w += (k < 4 ? s : 0.f); }
#pragma unroll
  for (int k = 0; k < 5; ++k) st[((size_t)t * 6 + k) * 64 + l] = g[k];
  out[i] = make_float2(s, s > 1.0f ? 1.f : 0.f);
}

template <typename F>
static double time_chain(hipStream_t s, F launch, int K = 200, int R = 9) {
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
    CK(hipGraphLaunch(ge, s));  // lead-in
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
  printf("{\n \"what\": \"us per launch, chain of 200 dependent launches in one hipGraph, median of 9 (HIP events)\"");
  float* dummy; CK(hipMalloc(&dummy, 1024));
  const int shapes[][2] = {{1, 64}, {256, 64}, {1024, 64}, {1024, 128}, {512, 256}, {256, 256}, {256, 512}, {2048, 64}, {3072, 64}, {4096, 64}, {16384, 64}};
  for (auto& sh : shapes) {
    const double us = time_chain(s, [&](int) { LAUNCH(empty_kernel, dim3(sh[0]), dim3(sh[1]), 0, s, dummy); });
    printf(",\n \"empty %dx%d\": %.3f", sh[0], sh[1], us);
  }
  for (int n : {65536, 131072, 262144, 1048576}) {
    const int tiles = n / 64;
    const int NA = 8;  // action slabs cycled through
    float *pv, *prm, *rew; double* ar; float4* act; uint8_t* done;
    CK(hipMalloc(&pv, (size_t)6 * n * 4)); CK(hipMalloc(&ar, (size_t)6 * n * 8)); CK(hipMalloc(&prm, (size_t)6 * n * 4));
    CK(hipMalloc(&act, (size_t)NA * n * 16)); CK(hipMalloc(&done, n)); CK(hipMalloc(&rew, (size_t)n * 4));
    CK(hipMemset(pv, 0, (size_t)6 * n * 4)); CK(hipMemset(ar, 0, (size_t)6 * n * 8)); CK(hipMemset(prm, 0, (size_t)6 * n * 4));
    CK(hipMemset(act, 0, (size_t)NA * n * 16));
    double us = time_chain(s, [&](int k) { LAUNCH(soa_kernel, dim3(tiles), dim3(64), 0, s, pv, ar, prm, act + (size_t)(k % NA) * n, done, rew, n); });
    printf(",\n \"soa narrow (19 loads, 14 stores) %d\": %.3f", n, us);
    // blocked: state 72 B/env = [tile][4][64] float4 + [n] double; params [n] float4 + [n] float2
    float4* st; double* st8; float4* p4; float2* p2;
    CK(hipMalloc(&st, (size_t)n * 64)); CK(hipMalloc(&st8, (size_t)n * 8)); CK(hipMalloc(&p4, (size_t)n * 16)); CK(hipMalloc(&p2, (size_t)n * 8));
    CK(hipMemset(st, 0, (size_t)n * 64)); CK(hipMemset(st8, 0, (size_t)n * 8)); CK(hipMemset(p4, 0, (size_t)n * 16)); CK(hipMemset(p2, 0, (size_t)n * 8));
    us = time_chain(s, [&](int k) { LAUNCH(blk_kernel, dim3(tiles), dim3(64), 0, s, st, st8, p4, p2, act + (size_t)(k % NA) * n, done, rew, n); });
    printf(",\n \"blocked wide (8 loads, 7 stores) %d\": %.3f", n, us);
    float4* rec; float2* out2;
    CK(hipMalloc(&rec, (size_t)n * 96)); CK(hipMalloc(&out2, (size_t)n * 8)); CK(hipMemset(rec, 0, (size_t)n * 96));
    us = time_chain(s, [&](int k) { LAUNCH(rec_kernel, dim3(tiles), dim3(64), 0, s, rec, act + (size_t)(k % NA) * n, out2, n); });
    printf(",\n \"record wide (7 loads, 6 stores) %d\": %.3f", n, us);
    CK(hipFree(pv)); CK(hipFree(ar)); CK(hipFree(prm)); CK(hipFree(act)); CK(hipFree(done)); CK(hipFree(rew));
    CK(hipFree(st)); CK(hipFree(st8)); CK(hipFree(p4)); CK(hipFree(p2)); CK(hipFree(rec)); CK(hipFree(out2));
  }
  printf("\n}\n");
  return 0;
}
