// Microbenchmark: what ONE vector instruction of each kind costs next to a stream of
// f32 MFMAs, per MFMA shape, at one and two waves per SIMD -- measured in shader cycles (s_memtime) inside
// the kernel, so the clock does not matter.
//   shape   v_mfma_f32_16x16x4_f32 (32 cycles, 2048 FLOP)  |  v_mfma_f32_32x32x2_f32 (64 cycles, 4096 FLOP)
//   filler  per 2048 FLOP of MFMA (so the 32x32x2 stream carries two per instruction):
//           none | v_add_f32 x1 x2 x4 | v_pk_add_f32 x1 | v_pk_fma_f32 x1 | v_fma_f32 x2 | v_log_f32 x1 |
//           v_max3_f32 x1 | ds_read_b32 x1
// Output: cycles per 2048 FLOP per SIMD (32.0 = the matrix pipe's rate).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

enum Filler { NONE, ADD1, ADD2, ADD4, PKADD1, PKFMA1, FMA2, LOG1, MAX3, DSREAD, NFILL };
static const char* kName[NFILL] = {"none", "v_add_f32 x1", "v_add_f32 x2", "v_add_f32 x4", "v_pk_add_f32 x1",
                                   "v_pk_fma_f32 x1", "v_fma_f32 x2", "v_log_f32 x1", "v_max3_f32 x1", "ds_read_b32 x1"};

template <int F>
__device__ __forceinline__ void filler(float (&x)[8], f32x2 (&p)[4], const float* lds, int i) {
  // independent of the MFMA results; registers rotate so that no filler waits for the previous one
  const int a = i & 3, b = (i + 1) & 3;
  if (F == ADD1) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[a]) : "v"(x[4 + a]));
  if (F == ADD2) {
    asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[a]) : "v"(x[4 + a]));
    asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[b]) : "v"(x[4 + b]));
  }
  if (F == ADD4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[j]) : "v"(x[4 + j]));
  }
  if (F == PKADD1) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[a]) : "v"(p[b]));
  if (F == PKFMA1) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[a]) : "v"(p[b]));
  if (F == FMA2) {
    asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(x[a]) : "v"(x[4 + a]));
    asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(x[b]) : "v"(x[4 + b]));
  }
  if (F == LOG1) asm volatile("v_log_f32 %0, %1" : "=v"(x[a]) : "v"(x[4 + a]));
  if (F == MAX3) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(x[a]) : "v"(x[4 + a]), "v"(x[4 + b]));
  if (F == DSREAD) x[a] = lds[(threadIdx.x + 64 * i) & 2047];
}

template <int F, bool BIG>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ tab, float* out, unsigned long long* cyc, int iters) {
  __shared__ float lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = tab[i & 1023];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float x[8];
  f32x2 p[4];
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = tab[lane + j] + 1.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j) p[j] = f32x2{x[j], x[4 + j]};
  const float a0 = tab[lane] + 0.5f, b0 = tab[lane + 64] + 0.25f;
  f32x4 acc[8];
  f32x16 big[2];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) big[j][e] = 0.f;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
    if (BIG) {
#pragma unroll
      for (int m = 0; m < 14; ++m) {            // 14 x 4096 FLOP = 28 x 2048
        big[m & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, big[m & 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        filler<F>(x, p, lds, 2 * m);
        filler<F>(x, p, lds, 2 * m + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int m = 0; m < 28; ++m) {
        acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[m & 7], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        filler<F>(x, p, lds, m);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][3] + x[j];
#pragma unroll
  for (int j = 0; j < 4; ++j) s += p[j][0] + p[j][1];
  s += big[0][0] + big[1][15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int F, bool BIG>
void run(const float* tab, int blocks, int iters) {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipMalloc(&cyc, (size_t)blocks * 4 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<F, BIG>), dim3(blocks), dim3(256), 0, 0, tab, out, cyc, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<F, BIG>), dim3(blocks), dim3(256), 0, 0, tab, out, cyc, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)blocks * 4);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double waves_per_simd = blocks * 4 / 1024.0;
  const double units = (double)iters * 28;                     // 2048-FLOP units per wave
  const double med = (double)h[h.size() / 2];
  printf("%-12s %-16s %d wave/SIMD  %6.2f cycles per 2048 FLOP per SIMD   (%.3f ms, %.1f TFLOP/s)\n",
         BIG ? "32x32x2" : "16x16x4", kName[F], (int)waves_per_simd, med / (units * waves_per_simd),
         ms, (double)blocks * 4 * units * 2048.0 / ms / 1e9);
  hipFree(out);
  hipFree(cyc);
}

template <bool BIG>
void all(const float* tab, int blocks) {
  const int it = 1500;
  run<NONE, BIG>(tab, blocks, it);
  run<ADD1, BIG>(tab, blocks, it);
  run<ADD2, BIG>(tab, blocks, it);
  run<ADD4, BIG>(tab, blocks, it);
  run<PKADD1, BIG>(tab, blocks, it);
  run<PKFMA1, BIG>(tab, blocks, it);
  run<FMA2, BIG>(tab, blocks, it);
  run<LOG1, BIG>(tab, blocks, it);
  run<MAX3, BIG>(tab, blocks, it);
  run<DSREAD, BIG>(tab, blocks, it);
}

int main() {
  float* tab;
  hipMalloc(&tab, 8192);
  std::vector<float> h(2048);
  for (int i = 0; i < 2048; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f;   // random-ish operands
  hipMemcpy(tab, h.data(), 8192, hipMemcpyHostToDevice);
  for (int blocks : {256, 512}) {
    all<false>(tab, blocks);
    all<true>(tab, blocks);
  }
  return 0;
}
