// Microbenchmark: what does a wave that STAGES (global loads + VALU) see while the other wave on its SIMD runs
// an MFMA stream?  One workgroup of 8 waves per CU: waves 0..3 stage, waves 4..7 run v_mfma_f32_16x16x4_f32
// back to back (or nothing).  The staging waves time, with s_memtime,
//   issue: cycles to ISSUE n loads (no wait)      wait: cycles until they have all returned
//   valu:  cycles for a chain-free run of 256 VALU instructions
// for dword and dwordx4 loads of the same bytes, streaming from HBM (every address touched once).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

template <int W, int N>     // W floats per lane and load, N loads per lane and repetition
__global__ __launch_bounds__(512) void k(const float* __restrict__ src, float* out, unsigned long long* times, int reps,
                                         int mfma_iters) {
  __shared__ float lds[24 * 1024];            // 96 KB: one workgroup per CU
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid < 64) lds[tid] = 0.f;
  if (wave >= 4) {                             // the MFMA stream
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
    const float a = lane * 0.001f, b = lane * 0.002f;
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j & 3], 0, 0, 0);
    }
    out[blockIdx.x * 512 + tid] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    return;
  }
  const int t = tid;                           // 0..255
  const size_t per_rep = (size_t)256 * W * N;
  const float* base = src + ((size_t)blockIdx.x * reps) * per_rep;
  unsigned long long issue = 0, wait = 0, valu = 0;
  float sum = 0.f;
  for (int r = 0; r < reps; ++r) {
    const float* p = base + (size_t)r * per_rep + (size_t)t * W;
    float v[N][W];
    const unsigned long long t0 = now();
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (W == 1) v[i][0] = p[256 * i];
      else {
        const f32x4 q = *reinterpret_cast<const f32x4*>(p + 1024 * i);
        v[i][0] = q[0], v[i][1 % W] = q[1], v[i][2 % W] = q[2], v[i][3 % W] = q[3];
      }
    }
    const unsigned long long t1 = now();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = now();
    float x = sum, y = 1.0f;
#pragma unroll
    for (int i = 0; i < 128; ++i) {           // 256 independent-ish VALU instructions (two chains)
      x = __builtin_fmaf(x, 1.0001f, 0.5f);
      y = __builtin_fmaf(y, 0.9999f, 0.25f);
    }
    const unsigned long long t3 = now();
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int w = 0; w < W; ++w) sum += v[i][w];
    sum += x + y;
    issue += t1 - t0, wait += t2 - t1, valu += t3 - t2;
  }
  out[blockIdx.x * 512 + tid] = sum;
  if (lane == 0) {
    unsigned long long* row = times + (blockIdx.x * 4 + wave) * 3;
    row[0] = issue, row[1] = wait, row[2] = valu;
  }
}

template <int W, int N>
void run(const float* src, int reps, int mfma_iters, const char* label) {
  const int blocks = 256;
  float* out;
  unsigned long long* times;
  hipMalloc(&out, blocks * 512 * 4);
  hipMalloc(&times, blocks * 4 * 3 * 8);
  hipMemset(times, 0, blocks * 4 * 3 * 8);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<W, N>), dim3(blocks), dim3(512), 0, 0, src, out, times, reps, mfma_iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 4 * 3);
  hipMemcpy(h.data(), times, h.size() * 8, hipMemcpyDeviceToHost);
  double s[3] = {0, 0, 0};
  for (int i = 0; i < blocks * 4; ++i)
    for (int j = 0; j < 3; ++j) s[j] += (double)h[i * 3 + j];
  for (int j = 0; j < 3; ++j) s[j] /= (double)blocks * 4 * reps;
  const double bytes = 256.0 * W * N * 4;    // per workgroup and repetition
  printf("%-28s W=%d N=%2d: issue %7.0f (%5.1f / load)  wait %7.0f  -> %5.2f B/cycle/CU   256 VALU %6.0f (%4.1f each)\n", label, W, N,
         s[0], s[0] / N, s[1], bytes / (s[0] + s[1]), s[2], s[2] / 256);
  hipFree(out);
  hipFree(times);
}

int main() {
  const int reps = 40;
  const size_t floats = (size_t)256 * reps * 256 * 48 + 4096;     // every repetition of every workgroup reads its own span
  float* src;
  hipMalloc(&src, floats * 4);
  hipMemset(src, 0, floats * 4);
  // s_memtime ticks: compare with the MFMA count to calibrate (16 MFMAs of 32 cycles per iteration)
  run<1, 42>(src, reps, 0, "alone, dword");
  run<4, 11>(src, reps, 0, "alone, dwordx4");
  run<1, 42>(src, reps, 40000, "next to MFMA, dword");
  run<4, 11>(src, reps, 40000, "next to MFMA, dwordx4");
  run<1, 21>(src, reps, 40000, "next to MFMA, dword");
  run<4, 5>(src, reps, 40000, "next to MFMA, dwordx4");
  return 0;
}
