// Microbenchmark: what keeps ONE or TWO waves per SIMD from issuing v_mfma_f32_16x16x4_f32 every
// 32 cycles?  Variants add, one at a time, what kernel 1R's GEMM phase has around its MFMAs.
//   V0  4 accumulator chains, fixed A / B registers
//   V1  100 distinct A registers (the resident twiddles), fixed B
//   V2  V1 + B built by a VALU add / sub per MFMA
//   V3  V2 + the four ds_read_b32 per k-step, three steps ahead
//   V4  V0 with 8 chains
//   V5  V3 with the adds of step s + 1 dealt out BEHIND the MFMAs of step s (no MFMA waits on a
//       VALU result issued just before it)
//   V6  V5 without the LDS reads (adds on registers)
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int V>
__global__ __launch_bounds__(256) void k2(const float* __restrict__ tab, float* out, int iters) {
  __shared__ float lds[8192];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = tab[i & 1023];
  __syncthreads();
  f32x4 A[25];
#pragma unroll
  for (int s = 0; s < 25; ++s) A[s] = *reinterpret_cast<const f32x4*>(tab + (s * 64 + lane) * 4);
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
  const float* p = lds + lane;
  float r0 = lane * 0.001f, r1 = lane * 0.002f;
  for (int it = 0; it < iters; ++it) {
    float xf[4][2], xb[4][2];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (V == 5) xf[d][c] = p[64 * (2 * d + c)], xb[d][c] = p[64 * (2 * d + c) + 3000];
        else xf[d][c] = r0 + d, xb[d][c] = r1 - c;
      }
    float e0 = xf[0][0] + xb[0][0], o0 = xf[0][0] - xb[0][0], e1 = xf[0][1] + xb[0][1], o1 = xf[0][1] - xb[0][1];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 25; ++s) {
      const int d = s % 4, dn = (s + 1) % 4;
      float ne0, no0, ne1, no1;
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s][0], e0, acc[0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ne0 = xf[dn][0] + xb[dn][0];
      no0 = xf[dn][0] - xb[dn][0];
      __builtin_amdgcn_sched_barrier(0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s][1], o0, acc[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ne1 = xf[dn][1] + xb[dn][1];
      no1 = xf[dn][1] - xb[dn][1];
      __builtin_amdgcn_sched_barrier(0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s][2], e1, acc[2], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (V == 5 && s + 4 < 25) {
#pragma unroll
        for (int c = 0; c < 2; ++c) xf[d][c] = p[64 * (2 * (s + 4) + c)], xb[d][c] = p[64 * (2 * (s + 4) + c) + 3000];
      }
      if (V == 6) xf[d][0] = no1, xb[d][1] = ne0;
      __builtin_amdgcn_sched_barrier(0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s][3], o1, acc[3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      e0 = ne0, o0 = no0, e1 = ne1, o1 = no1;
    }
    r0 = e0;
  }
  float sacc = 0;
  for (int i = 0; i < 4; ++i) sacc += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = sacc;
}

template <int V>
__global__ __launch_bounds__(256) void k(const float* __restrict__ tab, float* out, int iters) {
  __shared__ float lds[8192];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = tab[i & 1023];
  __syncthreads();
  f32x4 A[25];
#pragma unroll
  for (int s = 0; s < 25; ++s) A[s] = *reinterpret_cast<const f32x4*>(tab + (s * 64 + lane) * 4);
  constexpr int NACC = (V == 4) ? 8 : 4;
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float b0 = lane * 0.001f, b1 = lane * 0.002f;
  const float* p = lds + lane;
  for (int it = 0; it < iters; ++it) {
    float xf[3][2], xb[3][2];
    if (V == 3) {
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int c = 0; c < 2; ++c) xf[d][c] = p[64 * (2 * d + c)], xb[d][c] = p[64 * (2 * d + c) + 3000];
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s = 0; s < 25; ++s) {
      float e0 = b0, o0 = b1, e1 = b0, o1 = b1;
      if (V == 2) {
        e0 = b0 + b1, o0 = b0 - b1, e1 = b1 + e0, o1 = b1 - o0;
        b0 = o1;
      }
      if (V == 3) {
        const int d = s % 3;
        e0 = xf[d][0] + xb[d][0], o0 = xf[d][0] - xb[d][0], e1 = xf[d][1] + xb[d][1], o1 = xf[d][1] - xb[d][1];
        __builtin_amdgcn_sched_barrier(0);
        if (s + 3 < 25) {
#pragma unroll
          for (int c = 0; c < 2; ++c) xf[d][c] = p[64 * (2 * (s + 3) + c)], xb[d][c] = p[64 * (2 * (s + 3) + c) + 3000];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (V == 0 || V == 4) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[0][0], b0, acc[j], 0, 0, 0);
      } else {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s][0], e0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s][1], o0, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s][2], e1, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s][3], o1, acc[3], 0, 0, 0);
      }
      if (V == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  float sacc = 0;
  for (int i = 0; i < NACC; ++i) sacc += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = sacc;
}

template <int V>
void run(const float* tab, int blocks, int iters) {
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  if (V >= 5) hipLaunchKernelGGL(k2<V>, dim3(blocks), dim3(256), 0, 0, tab, out, iters);
  else hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, tab, out, iters);
  hipEventRecord(e0, 0);
  if (V >= 5) hipLaunchKernelGGL(k2<V>, dim3(blocks), dim3(256), 0, 0, tab, out, iters);
  else hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, tab, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const int per_step = (V == 4) ? 8 : 4;
  const double n_mfma = (double)iters * 25 * per_step;                 // per wave
  const double waves_per_simd = blocks * 4 / 1024.0;
  // cycles per MFMA per SIMD at 2.4 GHz
  printf("V%d blocks=%4d (%.0f waves/SIMD) %8.3f ms  %.1f TFLOP/s  %.1f cycles/MFMA/SIMD @2.4GHz\n", V, blocks, waves_per_simd, ms,
         (double)blocks * 4 * n_mfma * 2048.0 / ms / 1e9, ms * 1e-3 * 2.4e9 / (n_mfma * waves_per_simd));
  hipFree(out);
}

int main() {
  float* tab;
  hipMalloc(&tab, 25 * 64 * 4 * 4 + 4096);
  hipMemset(tab, 0, 25 * 64 * 4 * 4 + 4096);
  for (int blocks : {256, 512}) {
    run<0>(tab, blocks, 2000);
    run<4>(tab, blocks, 1000);
    run<1>(tab, blocks, 2000);
    run<2>(tab, blocks, 2000);
    run<3>(tab, blocks, 2000);
    run<5>(tab, blocks, 2000);
    run<6>(tab, blocks, 2000);
  }
  return 0;
}
