// valu_ilp_microbench.hip — what a DEPENDENT vector instruction costs a wave on gfx950, against an independent one.
// tools/valu_microbench.hip prices instructions in 16 independent chains (a lone wave: one every ~5.6 clocks); the step kernel's
// critical path is mostly dependent.  Here each kernel runs ITERS x CH instructions in CH independent accumulator chains, CH = 1
// (every instruction waits for the one before) ... 16, one wave per workgroup, grid 1024 / 2048 (one / two waves per SIMD).
//
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/ilp_mb tools/valu_ilp_microbench.hip && /tmp/ilp_mb > out.json
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int TOTAL = 32768;   // instructions per wave, whatever CH

struct FmaF32 { using T = float; static __device__ T init(float s, int i) { return s * i; }
  static __device__ void op(T& a, float s) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "v"(s)); } };
struct FmacF32 { using T = float; static __device__ T init(float s, int i) { return s * i; }
  static __device__ void op(T& a, float s) { asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a) : "v"(s)); } };
struct MulF32 { using T = float; static __device__ T init(float s, int i) { return 1.0f + s * i; }
  static __device__ void op(T& a, float s) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(s)); } };
struct FmaF64 { using T = double; static __device__ T init(float s, int i) { return 1.0 + (double)s * i; }
  static __device__ void op(T& a, float s) { const double d = (double)s; asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a) : "v"(d)); } };

template <typename Op, int CH>
__global__ __launch_bounds__(64) void bench_kernel(float* out, uint64_t* cyc, float s) {
  typename Op::T a[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) a[c] = Op::init(s, c + threadIdx.x);
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < TOTAL / CH / 4; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int c = 0; c < CH; ++c) Op::op(a[c], s);
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) r += (float)a[c];
  out[blockIdx.x * 64 + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef void (*kern_t)(float*, uint64_t*, float);
struct Case { const char* name; int ch; kern_t k; };
#define CASES(NAME, OP) {NAME, 1, bench_kernel<OP, 1>}, {NAME, 2, bench_kernel<OP, 2>}, {NAME, 3, bench_kernel<OP, 3>}, {NAME, 4, bench_kernel<OP, 4>}, \
                        {NAME, 8, bench_kernel<OP, 8>}, {NAME, 16, bench_kernel<OP, 16>}

int main() {
  const Case cases[] = {CASES("v_fma_f32", FmaF32), CASES("v_fmac_f32", FmacF32), CASES("v_mul_f32", MulF32), CASES("v_fma_f64", FmaF64)};
  const int grids[] = {1024, 2048, 4096};
  float* out; uint64_t* cyc;
  CK(hipMalloc(&out, 4096 * 64 * sizeof(float)));
  CK(hipMalloc(&cyc, 4096 * sizeof(uint64_t)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("{\"what\": \"wall ns per instruction of ONE wave (kernel duration / instructions per wave), %d instructions per wave in CH independent chains; "
         "one wave per workgroup, grid / 1024 waves per SIMD\", \"results\": [\n", TOTAL);
  bool first = true;
  for (const Case& c : cases) {
    for (int g : grids) {
      for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(c.k, dim3(g), dim3(64), 0, 0, out, cyc, 1e-3f); CK(hipGetLastError()); }
      float best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(c.k, dim3(g), dim3(64), 0, 0, out, cyc, 1e-3f);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
      }
      const int per_wave = (TOTAL / c.ch / 4) * 4 * c.ch;
      printf("%s{\"instr\": \"%s\", \"chains\": %d, \"waves_per_simd\": %d, \"kernel_us\": %.1f, \"ns_per_instr_of_a_wave\": %.3f, \"ns_per_instr_per_simd\": %.3f}",
             first ? "" : ",\n", c.name, c.ch, g / 1024, best * 1e3, best * 1e6 / per_wave, best * 1e6 / per_wave / (g / 1024));
      first = false;
    }
  }
  printf("]}\n");
  return 0;
}
