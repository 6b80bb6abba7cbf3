// valu_microbench.hip — issue cost of the vector instructions the step kernel is built from, on gfx950.
// One wave per workgroup; grid = 1024 (one wave per SIMD) or 2048 / 3072 / 4096 (two / three / four waves per SIMD).
// Each kernel runs ITERS iterations of 16 independent accumulator chains of one instruction and
// reports shader cycles (s_memtime) per wave-instruction, median over waves.
//
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_mb tools/valu_microbench.hip && /tmp/valu_mb > out.json
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define LAUNCH(...) do { hipLaunchKernelGGL(__VA_ARGS__); CK(hipGetLastError()); } while (0)
constexpr int ITERS = 2000;
constexpr int CH = 16;


template <typename Op>
__global__ __launch_bounds__(64) void bench_kernel(float* out, uint64_t* cyc, float s) {
  typename Op::T a[CH];
  const typename Op::S sv = Op::operand(s);
#pragma unroll
  for (int c = 0; c < CH; ++c) a[c] = Op::init(s, c + threadIdx.x);
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int c = 0; c < CH; ++c) Op::op(a[c], sv);
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) r += Op::sink(a[c]);
  out[blockIdx.x * 64 + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

struct F32 { using T = float; using S = float;
  static __device__ S operand(float s) { return s; }
  static __device__ T init(float s, int i) { return s * i; }
  static __device__ float sink(T a) { return a; } };
struct PK { using T = float2v; using S = float2v;
  static __device__ S operand(float s) { S v; v.x = s; v.y = s * 0.5f; return v; }
  static __device__ T init(float s, int i) { T v; v.x = s * i; v.y = s + i; return v; }
  static __device__ float sink(T a) { return a.x + a.y; } };
struct F64 { using T = double; using S = double;
  static __device__ S operand(float s) { return (double)s; }
  static __device__ T init(float s, int i) { return 1.0 + (double)s * i; }
  static __device__ float sink(T a) { return (float)a; } };
struct U32 { using T = uint32_t; using S = uint32_t;
  static __device__ S operand(float s) { return 0xD2511F53u + (uint32_t)s; }
  static __device__ T init(float s, int i) { return (uint32_t)i * 2654435761u; }
  static __device__ float sink(T a) { return (float)a; } };

struct FmaF32 : F32 { static __device__ void op(T& a, S s) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "v"(s)); } };
struct PkFma : PK { static __device__ void op(T& a, S s) { asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a) : "v"(s)); } };
struct PkFmaSwz : PK { static __device__ void op(T& a, S s) { asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]" : "+v"(a) : "v"(s)); } };
struct PkMul : PK { static __device__ void op(T& a, S s) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(s)); } };
struct PkAdd : PK { static __device__ void op(T& a, S s) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(s)); } };
struct FmaF64 : F64 { static __device__ void op(T& a, S s) { asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a) : "v"(s)); } };
struct AddF64 : F64 { static __device__ void op(T& a, S s) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(s)); } };
struct MulF64 : F64 { static __device__ void op(T& a, S s) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(s)); } };
struct RcpF64 : F64 { static __device__ void op(T& a, S s) { asm volatile("v_rcp_f64 %0, %0" : "+v"(a)); } };
struct CvtDown : F64 { static __device__ void op(T& a, S s) { float lo; asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(lo) : "v"(a)); asm volatile("" :: "v"(lo)); } };
struct CvtUp : F64 { static __device__ void op(T& a, S s) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a) : "v"((float)s)); } };
struct MulHi : U32 { static __device__ void op(T& a, S s) { asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(s)); } };
struct MulLo : U32 { static __device__ void op(T& a, S s) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(s)); } };
struct RcpF32 : F32 { static __device__ void op(T& a, S s) { asm volatile("v_rcp_f32 %0, %0" : "+v"(a)); } };

__global__ __launch_bounds__(64) void k_empty(float* out, uint64_t* cyc, float s) {
  if (s < -1e30f) out[0] = 0;
}

typedef void (*kern_t)(float*, uint64_t*, float);
struct Case { const char* name; kern_t k; };

int main() {
  const Case cases[] = {{"v_fma_f32", bench_kernel<FmaF32>}, {"v_pk_fma_f32", bench_kernel<PkFma>},
                        {"v_pk_fma_f32_opsel_neg", bench_kernel<PkFmaSwz>}, {"v_pk_mul_f32", bench_kernel<PkMul>}, {"v_pk_add_f32", bench_kernel<PkAdd>},
                        {"v_fma_f64", bench_kernel<FmaF64>}, {"v_add_f64", bench_kernel<AddF64>}, {"v_mul_f64", bench_kernel<MulF64>},
                        {"v_rcp_f64", bench_kernel<RcpF64>}, {"v_rcp_f32", bench_kernel<RcpF32>}, {"v_cvt_f32_f64", bench_kernel<CvtDown>},
                        {"v_cvt_f64_f32", bench_kernel<CvtUp>}, {"v_mul_hi_u32", bench_kernel<MulHi>}, {"v_mul_lo_u32", bench_kernel<MulLo>}};
  const int grids[] = {1024, 2048, 3072, 4096};
  float* out; uint64_t* cyc;
  CK(hipMalloc(&out, 4096 * 64 * sizeof(float)));
  CK(hipMalloc(&cyc, 4096 * sizeof(uint64_t)));
  std::vector<uint64_t> h(4096);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("{\"what\": \"cycles per wave-instruction (s_memtime, median over waves) and wall ns per instruction per SIMD; %d iters x %d independent chains, one wave per workgroup\", \"results\": [\n", ITERS, CH);
  bool first = true;
  for (const Case& c : cases) {
    for (int g : grids) {
      for (int rep = 0; rep < 3; ++rep) LAUNCH(c.k, dim3(g), dim3(64), 0, 0, out, cyc, 1e-3f);
      CK(hipEventRecord(e0));
      LAUNCH(c.k, dim3(g), dim3(64), 0, 0, out, cyc, 1e-3f);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(h.data(), cyc, g * sizeof(uint64_t), hipMemcpyDeviceToHost));
      std::sort(h.begin(), h.begin() + g);
      const double per = (double)h[g / 2] / ((double)ITERS * CH);
      const double waves_per_simd = g / 1024.0;
      // wall clock: the kernel's duration over the instructions one SIMD issued (its waves' ITERS x CH each) — the figure to price a
      // VALU-bound launch with; the s_memtime "cycles" above tick at their own rate (NOT the 2.4 GHz shader clock)
      printf("%s{\"instr\": \"%s\", \"waves_per_simd\": %.0f, \"cycles_per_wave_instr\": %.2f, \"cycles_per_instr_per_simd\": %.2f, \"kernel_us\": %.1f, "
             "\"wall_ns_per_instr_per_simd\": %.3f}",
             first ? "" : ",\n", c.name, waves_per_simd, per, per / waves_per_simd, ms * 1e3, ms * 1e6 / (waves_per_simd * ITERS * CH));
      first = false;
    }
  }
  // launch floor: back-to-back empty kernels, 1024 workgroups
  for (int rep = 0; rep < 20; ++rep) LAUNCH(k_empty, dim3(1024), dim3(64), 0, 0, out, cyc, 1.0f);
  CK(hipEventRecord(e0));
  for (int rep = 0; rep < 1000; ++rep) LAUNCH(k_empty, dim3(1024), dim3(64), 0, 0, out, cyc, 1.0f);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("],\n\"empty_kernel_1024wg_eager_us_per_launch\": %.3f}\n", ms);
  return 0;
}
