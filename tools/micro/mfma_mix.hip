// Microbenchmark: what the non-MFMA instructions of the front end's k-step cost a
// stream of 28 independent v_mfma_f32_16x16x4_f32 (one wave per SIMD, 4-wave block).
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x3 = __attribute__((ext_vector_type(3))) float;

// MODE bits: 1 = 8 fragment reads (b128+b96 x4), 2 = 4 ds_read_b32, 4 = 2 ds_write_b128, 8 = barrier,
//            16 = 2 global loads, 32 = fragments feed the MFMAs (ping-pong)
template <int MODE>
__global__ __launch_bounds__(256) void kmix(float* out, const f32x4* gsrc, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16384; i += 256) lds[i] = i * 1e-4f;
  __syncthreads();
  f32x4 acc[28];
  for (int i = 0; i < 28; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a[28];
  for (int i = 0; i < 28; ++i) a[i] = tid * 0.001f + i;
  float b0 = tid * 0.002f, b1 = b0 + 1, b2 = b0 + 2, b3 = b0 + 3;
  f32x4 fl[4]; f32x3 fh[4]; float xs[4]; f32x4 g0 = {0,0,0,0}, g1 = {0,0,0,0};
  for (int g = 0; g < 4; ++g) { fl[g] = f32x4{1,2,3,4}; fh[g] = f32x3{5,6,7}; xs[g] = g; }
  const f32x4* slot = reinterpret_cast<const f32x4*>(lds) + lane;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 28; ++m) {
      float av = a[m];
      if (MODE & 32) { const int g = m / 7, bt = m % 7; av = bt < 4 ? fl[g][bt & 3] : fh[g][bt & 3]; }
      const float bv = (m / 7 == 0) ? b0 : (m / 7 == 1) ? b1 : (m / 7 == 2) ? b2 : b3;
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv + ((MODE & 2) ? xs[m / 7] : 0.f), acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (m < 2 && (MODE & 16)) { if (m == 0) g0 = gsrc[(it & 15) * 512 + tid]; else g1 = gsrc[(it & 15) * 512 + 256 + tid]; }
      if (m >= 2 && m < 4 && (MODE & 4)) reinterpret_cast<f32x4*>(lds)[2048 + (m - 2) * 256 + tid] = (m == 2) ? g0 : g1;
      if (m >= 4 && m < 12 && (MODE & 1)) {
        const int g = (m - 4) >> 1;
        if ((m - 4) & 1) fh[g] = *reinterpret_cast<const f32x3*>(slot + ((it & 1) * 512 + g * 128 + 64));
        else fl[g] = slot[(it & 1) * 512 + g * 128];
      }
      if (m >= 12 && m < 16 && (MODE & 2)) xs[m - 12] = lds[4096 + (tid * 161 + m + it) % 8192];
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE & 8) __syncthreads();
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 28; ++i) s += acc[i][0] + acc[i][3];
  for (int g = 0; g < 4; ++g) s += fl[g][0] + fh[g][2] + xs[g];
  s += g0[0] + g1[3];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class K> void run(const char* name, K kern, int blocks, float* out, f32x4* gsrc, unsigned long long* cyc) {
  const int iters = 1000;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, gsrc, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long h[1]; hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s blocks=%4d cycles/step = %7.1f  (28 MFMA = 896)\n", name, blocks, (double)h[0] / iters);
}

int main() {
  float* out; unsigned long long* cyc; f32x4* gsrc;
  hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 512 * 8); hipMalloc(&gsrc, 16 * 512 * 16 * 2); hipMemset(gsrc, 0, 16 * 512 * 16 * 2);
  for (int blocks : {256, 512}) {
    run("pure MFMA", kmix<0>, blocks, out, gsrc, cyc);
    run("+8 fragment reads", kmix<1>, blocks, out, gsrc, cyc);
    run("+8 fragment reads feeding MFMAs", kmix<1 | 32>, blocks, out, gsrc, cyc);
    run("+4 ds_read_b32", kmix<2>, blocks, out, gsrc, cyc);
    run("+2 global loads", kmix<16>, blocks, out, gsrc, cyc);
    run("+2 global loads +2 ds_write_b128", kmix<16 | 4>, blocks, out, gsrc, cyc);
    run("+barrier", kmix<8>, blocks, out, gsrc, cyc);
    run("everything", kmix<63>, blocks, out, gsrc, cyc);
    run("everything but barrier", kmix<63 - 8>, blocks, out, gsrc, cyc);
  }
  return 0;
}
