// Microbenchmark: how fast does a wave issue VALU / SALU / LDS / VMEM instructions while the OTHER wave on its
// SIMD streams v_mfma_f32_16x16x4_f32?  One workgroup of 8 waves per CU (96 KB LDS): waves 0..3 run a
// block of 256 instructions of one kind between two s_memtime, waves 4..7 run MFMAs back to back
// (mode 1), MFMAs with an LDS read behind each (mode 2, as kernel 1's k-loop has), or nothing (mode 0).
// prio: s_setprio of the instruction waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define R4(X) X X X X
#define R16(X) R4(R4(X))
#define R256(X) R16(R16(X))

template <int KIND, int MODE, int PRIO>
__global__ __launch_bounds__(512) void k(const float* __restrict__ src, float* out, unsigned long long* times, int reps,
                                         int mfma_iters) {
  __shared__ float lds[24 * 1024];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 24 * 1024; i += 512) lds[i] = 0.f;
  __syncthreads();
  if (wave >= 4) {
    if (MODE == 0) return;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = lane * 0.001f, b = lane * 0.002f;
    const float* p = lds + lane;
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j & 3], 0, 0, 0);
        if (MODE == 2) {
          __builtin_amdgcn_sched_barrier(0);
          b += p[64 * j];                      // one ds_read (and a dependent add) behind each MFMA
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    out[blockIdx.x * 512 + tid] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    return;
  }
  if (PRIO) __builtin_amdgcn_s_setprio(3);
  unsigned long long total = 0;
  float f0 = lane, f1 = lane + 1, f2 = lane + 2, f3 = lane + 3;
  int s0 = __builtin_amdgcn_readfirstlane(wave);
  const float* p = lds + lane;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 20, 0x00020000);
  for (int r = 0; r < reps; ++r) {
    const unsigned long long t0 = now();
    if (KIND == 0) {          // VALU, four independent chains
      R16(R4(asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f0)); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f1));
             asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f2)); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f3));))
    } else if (KIND == 1) {   // SALU
      R256(asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0));)
    } else if (KIND == 2) {   // LDS reads
      R16(R4(asm volatile("ds_read_b32 %0, %1" : "=v"(f0) : "v"(lane * 4)); asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(f1) : "v"(lane * 4));
             asm volatile("ds_read_b32 %0, %1 offset:512" : "=v"(f2) : "v"(lane * 4)); asm volatile("ds_read_b32 %0, %1 offset:768" : "=v"(f3) : "v"(lane * 4));))
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {                  // VMEM: 64 cached 16-byte loads (L1 / L2 hits), no address arithmetic
      f32x4 v0, v1, v2, v3;
      R16(asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v0) : "v"(lane * 16), "s"(rs));
          asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:1024" : "=v"(v1) : "v"(lane * 16), "s"(rs));
          asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:2048" : "=v"(v2) : "v"(lane * 16), "s"(rs));
          asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:3072" : "=v"(v3) : "v"(lane * 16), "s"(rs));)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      f0 += v0[0] + v1[1] + v2[2] + v3[3];
    }
    total += now() - t0;
  }
  out[blockIdx.x * 512 + tid] = f0 + f1 + f2 + f3 + s0;
  if (lane == 0) times[blockIdx.x * 4 + wave] = total;
}

template <int KIND, int MODE, int PRIO>
void run(const float* src, const char* label) {
  const int blocks = 256, reps = 200;
  const int n = KIND == 3 ? 64 : 256;
  float* out;
  unsigned long long* times;
  hipMalloc(&out, blocks * 512 * 4);
  hipMalloc(&times, blocks * 4 * 8);
  hipMemset(times, 0, blocks * 4 * 8);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND, MODE, PRIO>), dim3(blocks), dim3(512), 0, 0, src, out, times, reps, 100000);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), times, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  s /= (double)blocks * 4 * reps;
  printf("%-6s partner=%s prio=%d: %8.0f cycles per block of %d -> %6.1f per instruction\n", label,
         MODE == 0 ? "idle      " : MODE == 1 ? "mfma      " : "mfma+lds  ", PRIO, s, n, s / n);
  hipFree(out);
  hipFree(times);
}

int main() {
  float* src;
  hipMalloc(&src, 1 << 20);
  hipMemset(src, 0, 1 << 20);
#define ALL(KIND, LABEL) \
  run<KIND, 0, 0>(src, LABEL); run<KIND, 1, 0>(src, LABEL); run<KIND, 1, 1>(src, LABEL); run<KIND, 2, 0>(src, LABEL); run<KIND, 2, 1>(src, LABEL);
  ALL(0, "VALU")
  ALL(1, "SALU")
  ALL(2, "LDS")
  ALL(3, "VMEM")
  return 0;
}
