// Microbenchmark: sustained chip-wide rate of v_mfma_f32_16x16x4_f32 measured on the wall clock
// (HIP events), 1 / 2 / 4 waves per SIMD, nothing but MFMAs in the loop.  This is the ceiling
// kernel 1's matrix phase can reach on this part at the clock it sustains under that load.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a[NACC], b[NACC];
  for (int i = 0; i < NACC; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = threadIdx.x * 0.002f - i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC> void run(int blocks, int iters) {
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k16<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k16<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * 4 * iters * NACC * 2048.0;
  printf("acc=%2d blocks=%5d (%.1f waves/SIMD) %8.3f ms  %.1f TFLOP/s\n", NACC, blocks, blocks * 4 / 1024.0, ms, flop / ms / 1e9);
  hipFree(out);
}

int main() {
  for (int blocks : {256, 512, 1024}) {
    run<28>(blocks, 20000);
    run<14>(blocks, 40000);
  }
  run<28>(512 * 8, 5000);
  return 0;
}
