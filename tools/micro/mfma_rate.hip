// Microbenchmark: issue rate of f32-input MFMAs from one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC, bool SAME_B>
__global__ __launch_bounds__(256) void k16(float* out, unsigned long long* cyc, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a[NACC], b[NACC];
  for (int i = 0; i < NACC; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = threadIdx.x * 0.002f - i; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], SAME_B ? b[0] : b[i], acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  float a[NACC], b[NACC];
  for (int i = 0; i < NACC; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = threadIdx.x * 0.002f - i; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class K> void run(const char* name, K kern, int nacc, int blocks) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
  const int iters = 2000;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long h[4]; hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
  printf("%-28s blocks=%4d cycles/MFMA = %.2f\n", name, blocks, (double)h[0] / iters / nacc);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int blocks : {1, 256, 512}) {
    run("16x16x4 28 acc", k16<28, false>, 28, blocks);
    run("16x16x4 28 acc same B", k16<28, true>, 28, blocks);
    run("16x16x4 4 acc", k16<4, false>, 4, blocks);
    run("16x16x4 2 acc", k16<2, false>, 2, blocks);
    run("32x32x2 7 acc", k32<7>, 7, blocks);
    run("32x32x2 2 acc", k32<2>, 2, blocks);
  }
  return 0;
}
