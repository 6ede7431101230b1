// Microbenchmark: issue rate of the vector instructions the augmentation staging leans on
// (float64 arithmetic and conversions next to float32 fma), one wave per SIMD, 8 independent
// chains per lane.  Prints cycles per wave-instruction relative to the wall clock.
#include <hip/hip_runtime.h>
#include <cstdio>

#define KERNEL(NAME, DECL, BODY)                                                   \
  __global__ __launch_bounds__(64) void NAME(double* out, int iters) {             \
    DECL;                                                                          \
    for (int it = 0; it < iters; ++it) {                                           \
      BODY BODY BODY BODY                                                          \
    }                                                                              \
    out[blockIdx.x * 64 + threadIdx.x] = (double)(sink());                         \
  }

struct R {
  double d[8];
  float f[8];
  int i[8];
};
#define DECL8                                                                           \
  double d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7; \
  float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7; \
  int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7; \
  auto sink = [&]() { return d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7; }

#define OP8(ASM, C0, C1)                                                                                      \
  asm volatile(ASM "\n" : "+" C0(x0) : C1(y0)); asm volatile(ASM "\n" : "+" C0(x1) : C1(y1));                 \
  asm volatile(ASM "\n" : "+" C0(x2) : C1(y2)); asm volatile(ASM "\n" : "+" C0(x3) : C1(y3));                 \
  asm volatile(ASM "\n" : "+" C0(x4) : C1(y4)); asm volatile(ASM "\n" : "+" C0(x5) : C1(y5));                 \
  asm volatile(ASM "\n" : "+" C0(x6) : C1(y6)); asm volatile(ASM "\n" : "+" C0(x7) : C1(y7));

#define MK(NAME, ASM, XT, YT)                                                                         \
  __global__ __launch_bounds__(64) void NAME(double* out, int iters) {                                \
    DECL8;                                                                                            \
    for (int it = 0; it < iters; ++it) {                                                              \
      for (int r = 0; r < 4; ++r) {                                                                   \
        asm volatile(ASM : "+v"(XT##0) : "v"(YT##0)); asm volatile(ASM : "+v"(XT##1) : "v"(YT##1));   \
        asm volatile(ASM : "+v"(XT##2) : "v"(YT##2)); asm volatile(ASM : "+v"(XT##3) : "v"(YT##3));   \
        asm volatile(ASM : "+v"(XT##4) : "v"(YT##4)); asm volatile(ASM : "+v"(XT##5) : "v"(YT##5));   \
        asm volatile(ASM : "+v"(XT##6) : "v"(YT##6)); asm volatile(ASM : "+v"(XT##7) : "v"(YT##7));   \
      }                                                                                               \
    }                                                                                                 \
    out[blockIdx.x * 64 + threadIdx.x] = (double)(sink());                                            \
  }

MK(k_fma_f32, "v_fma_f32 %0, %1, %1, %0", f, f)
MK(k_add_u32, "v_add_u32 %0, %1, %0", i, i)
MK(k_fma_f64, "v_fma_f64 %0, %1, %1, %0", d, d)
MK(k_mul_f64, "v_mul_f64 %0, %1, %0", d, d)
MK(k_add_f64, "v_add_f64 %0, %1, %0", d, d)
MK(k_trunc_f64, "v_trunc_f64 %0, %1", d, d)
MK(k_cvt_f64_i32, "v_cvt_f64_i32 %0, %1", d, i)
MK(k_cvt_i32_f64, "v_cvt_i32_f64 %0, %1", i, d)
MK(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %1", d, f)
MK(k_cvt_f32_f64, "v_cvt_f32_f64 %0, %1", f, d)
MK(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %1", f, i)
MK(k_mul_lo_u32, "v_mul_lo_u32 %0, %1, %0", i, i)
MK(k_mul_hi_u32, "v_mul_hi_u32 %0, %1, %0", i, i)
MK(k_rcp_f32, "v_rcp_f32 %0, %1", f, f)
MK(k_cndmask, "v_cndmask_b32 %0, %1, %0, vcc", i, i)

template <class K> void run(const char* name, K k) {
  const int blocks = 1024, iters = 20000;
  double* out;
  hipMalloc(&out, (size_t)blocks * 64 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_wave = (double)iters * 32;            // wave-instructions issued by each wave
  printf("%-16s %8.3f ms  %6.2f ns per wave-instruction (1 wave per SIMD)\n", name, ms, ms * 1e6 / per_wave);
  hipFree(out);
}

int main() {
  run("v_fma_f32", k_fma_f32);
  run("v_add_u32", k_add_u32);
  run("v_cndmask_b32", k_cndmask);
  run("v_mul_lo_u32", k_mul_lo_u32);
  run("v_mul_hi_u32", k_mul_hi_u32);
  run("v_rcp_f32", k_rcp_f32);
  run("v_cvt_f32_i32", k_cvt_f32_i32);
  run("v_fma_f64", k_fma_f64);
  run("v_mul_f64", k_mul_f64);
  run("v_add_f64", k_add_f64);
  run("v_trunc_f64", k_trunc_f64);
  run("v_cvt_f64_i32", k_cvt_f64_i32);
  run("v_cvt_i32_f64", k_cvt_i32_f64);
  run("v_cvt_f64_f32", k_cvt_f64_f32);
  run("v_cvt_f32_f64", k_cvt_f32_f64);
  return 0;
}
