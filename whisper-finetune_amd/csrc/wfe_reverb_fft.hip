// wfe_reverb_fft.hip -- reverberation with LONG impulse responses as a partitioned overlap-save FFT convolution
// (round 3).  CustomDataset.add_reverb (utils/reader.py:342-348) calls scipy.signal.fftconvolve(sample, rir,
// mode='same'): an FFT product.  wfe_kernels.hip restates it as the direct float32 sum (reverb_window_kernel),
// which costs 2 N Lr FLOP -- 41 ms per 256 x 30 s with a 1 s response at 95 TFLOP/s.  Here, for responses of
// kFftMinTaps taps and more:
//   * the response is cut into partitions of B = 2048 taps, each zero-padded to N = 4096 and transformed once per
//     bank (reverb_spectra_kernel; cached in the handle);
//   * an utterance is cut into blocks of B outputs of the FULL convolution; block k needs the transforms Z_j of the
//     input windows x[(j - 1) B, (j + 1) B) for j = k, k - 1, ..., k - P + 1:  Y_k = sum_p Z_{k-p} H_p, and the last B
//     samples of IFFT(Y_k) are full[k B, (k + 1) B)  (uniformly partitioned overlap-save);
//   * convolution with a REAL response keeps real and imaginary parts apart, so one complex transform carries TWO
//     blocks: a workgroup owns a range of blocks [k0, k1) of one utterance and pairs block k0 + i with block
//     k0 + i + Kh (Kh = half the range); the pairs i - p a block pair needs are the pairs it computed before
//     (a ring of P spectra in global memory, L2 resident), started P - 1 pairs early;
//   * the 4096-point transform is a radix-16 Stockham FFT in LDS, three passes, 16 points per thread in registers,
//     twiddles exp(-2 pi i m / 4096) as the product of two 64-entry tables built in float64; the inverse transform is
//     the forward one on swapped components, its 1 / N folded into the response's spectra.
// 'same' mode keeps full[c + n], c = (Lr - 1) / 2, n < len; the peak normalisation is reverb_scale_kernel's.
// Not bit-identical to the direct sum (neither is scipy's float32 FFT): within the same bound,
// |err| <= 3e-6 max|y| sqrt(taps / 64) of the reference's output (tests/test_augment_gpu.py).
#include <hip/hip_runtime.h>

#include "wfe_kernels.h"

namespace wfe {

namespace {

using c32 = float2;
constexpr int kN = kRvFftN, kB = kRvFftN / 2;

__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return c32{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return c32{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ c32 cmul(c32 a, c32 b) { return c32{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ c32 mul_mi(c32 a) { return c32{a.y, -a.x}; }        // a * (-i)

// forward 4-point DFT (exp(-2 pi i n k / 4)), in place
__device__ __forceinline__ void dft4(c32& a0, c32& a1, c32& a2, c32& a3) {
  const c32 s02 = cadd(a0, a2), d02 = csub(a0, a2), s13 = cadd(a1, a3), d13 = mul_mi(csub(a1, a3));
  a0 = cadd(s02, s13);
  a2 = csub(s02, s13);
  a1 = cadd(d02, d13);
  a3 = csub(d02, d13);
}

// forward 16-point DFT of v[n], n = 4 n1 + n2; afterwards v[4 k1 + k2] = X[k1 + 4 k2]
__device__ __forceinline__ void dft16(c32 (&v)[16]) {
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) dft4(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
  // v[4 k1 + n2] *= exp(-2 pi i n2 k1 / 16)
  constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
  const c32 w1{c1, -s1}, w2{h, -h}, w3{s1, -c1}, w6{-h, -h}, w9{-c1, s1};
  v[5] = cmul(v[5], w1);
  v[6] = cmul(v[6], w2);
  v[7] = cmul(v[7], w3);
  v[9] = cmul(v[9], w2);
  v[10] = mul_mi(v[10]);                        // w4 = -i
  v[11] = cmul(v[11], w6);
  v[13] = cmul(v[13], w3);
  v[14] = cmul(v[14], w6);
  v[15] = cmul(v[15], w9);
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) dft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
}

// Thread-constant twiddles: thread j multiplies input r of pass 1 by exp(-2 pi i (j mod 16) r / 256) and of pass 2 by
// exp(-2 pi i j r / 4096), the same for every transform it takes part in.  exp(-2 pi i m / 4096) is the float32
// product of two entries of the 64-entry tables (tw[a] = exp(-2 pi i 64 a / N), tw[64 + b] = exp(-2 pi i b / N),
// float64 on the host).
struct Twiddles {
  c32 p1[15], p2[15];
};
__device__ __forceinline__ void make_twiddles(Twiddles& t, const c32* __restrict__ tw, int j) {
#pragma unroll
  for (int r = 1; r < 16; ++r) {
    const int m1 = (j & 15) * r * 16, m2 = j * r;
    t.p1[r - 1] = cmul(tw[m1 >> 6], tw[64 + (m1 & 63)]);
    t.p2[r - 1] = cmul(tw[m2 >> 6], tw[64 + (m2 & 63)]);
  }
}

// LDS index of point i: one pad element per 16, so that pass 0's stores (lane stride 16 points) and pass 1's
// (16 consecutive lanes, then a jump of 256 points) spread over the banks
__device__ __forceinline__ int lp(int i) { return i + (i >> 4); }
constexpr int kBufLen = kN + kN / 16;

// Forward 4096-point FFT by all 256 threads.  In: v[r] = x[j + 256 r].  Out: v[k] = X[j + 256 k] -- the SAME thread
// owns point j + 256 k on both sides, so transforms chain through registers (window -> spectrum -> product ->
// inverse -> samples) and buf is scratch.  Stockham autosort, radix 16, passes with Ns = 1, 16, 256: thread j
// takes the points j + 256 r, multiplies by exp(-2 pi i (j mod Ns) r / (16 Ns)), transforms, and puts X[k] at
// (j div Ns) 16 Ns + (j mod Ns) + k Ns.  The barrier before pass 0's stores also ends the previous call's reads.
__device__ __forceinline__ void fft4096(c32 (&v)[16], c32* buf, const Twiddles& t, int j) {
  dft16(v);
  __syncthreads();
#pragma unroll
  for (int pos = 0; pos < 16; ++pos) buf[17 * j + ((pos >> 2) + 4 * (pos & 3))] = v[pos];   // lp(16 j + k), k < 16
  __syncthreads();
  const int rd = j + (j >> 4);                  // lp(j + 256 r) = rd + 272 r
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = buf[rd + 272 * r];
#pragma unroll
  for (int r = 1; r < 16; ++r) v[r] = cmul(v[r], t.p1[r - 1]);
  dft16(v);
  __syncthreads();
  {
    const int jm = j & 15, base = (j - jm) * 16 + jm;
#pragma unroll
    for (int pos = 0; pos < 16; ++pos) buf[lp(base + ((pos >> 2) + 4 * (pos & 3)) * 16)] = v[pos];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = buf[rd + 272 * r];
#pragma unroll
  for (int r = 1; r < 16; ++r) v[r] = cmul(v[r], t.p2[r - 1]);
  dft16(v);
  c32 w[16];
#pragma unroll
  for (int pos = 0; pos < 16; ++pos) w[(pos >> 2) + 4 * (pos & 3)] = v[pos];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = w[k];
}

// what every kernel below needs to know of utterance b
struct Utt {
  const float* x;
  float* o;
  const c32* Hr;
  int len, ri, lr, c, P, k_first, k_last;
};
__device__ __forceinline__ Utt utterance(const ReverbArgs& a, int b, const c32* H, int p_max) {
  Utt u;
  u.len = a.lengths[b];
  if (u.len < 0) u.len = 0;
  if (u.len > a.out_stride) u.len = (int)a.out_stride;
  u.ri = a.rir_index[b];
  u.x = a.wav + (a.offsets ? a.offsets[b] : (long long)b * a.row_stride);
  u.o = a.out + (long long)b * a.out_stride;
  u.lr = 0;
  if (u.ri >= 0) u.lr = (int)(a.rir_off[u.ri + 1] - a.rir_off[u.ri]);
  u.c = u.lr > 0 ? (u.lr - 1) / 2 : 0;
  u.P = u.lr > 0 ? (u.lr + kB - 1) / kB : 1;        // (an empty response: y = 0, as the direct sum)
  u.k_first = u.c / kB;
  u.k_last = u.len > 0 ? (u.c + u.len - 1) / kB : u.k_first - 1;
  u.Hr = H + (long long)(u.ri >= 0 ? u.ri : 0) * p_max * kN;
  return u;
}

// the windows of blocks ka (real part) and kb (imaginary part; zero unless b_valid), and the peak of what was read
// (R0 = 8: the first halves are the second halves of the pair before, which the caller still holds)
template <int R0 = 0>
__device__ __forceinline__ void load_windows(c32 (&v)[16], const Utt& u, int ka, int kb, bool b_valid, int tid, float& pk_x) {
  const long long sa = (long long)(ka - 1) * kB + tid, sb = (long long)(kb - 1) * kB + tid;
#pragma unroll
  for (int r = R0; r < 16; ++r) {
    const long long ia = sa + 256 * r, ib = sb + 256 * r;
    const float xa = (ia >= 0 && ia < u.len) ? u.x[ia] : 0.f;
    const float xb = (b_valid && ib >= 0 && ib < u.len) ? u.x[ib] : 0.f;
    pk_x = fmaxf(pk_x, fmaxf(fabsf(xa), fabsf(xb)));
    v[r] = c32{xa, xb};
  }
}

// v = the transform of the swapped product: .y is the real part.  Its last B samples are full[k B, (k + 1) B) of
// blocks ka and kb; 'same' mode keeps full[c + n].
__device__ __forceinline__ void store_blocks(const c32 (&v)[16], const Utt& u, int ka, int kb, bool b_valid, int tid, float& pk_y) {
#pragma unroll
  for (int k = 8; k < 16; ++k) {
    const int n = tid + 256 * k - kB;
    const long long oa = (long long)ka * kB + n - u.c, ob = (long long)kb * kB + n - u.c;
    if (oa >= 0 && oa < u.len) {
      u.o[oa] = v[k].y;
      pk_y = fmaxf(pk_y, fabsf(v[k].y));
    }
    if (b_valid && ob >= 0 && ob < u.len) {
      u.o[ob] = v[k].x;
      pk_y = fmaxf(pk_y, fabsf(v[k].x));
    }
  }
}

// per-utterance peaks, as reverb_window_kernel (NaN peaks are not propagated: max ignores them)
__device__ __forceinline__ void put_peaks(const ReverbArgs& a, int b, int tid, float pk_x, float pk_y, bool x, bool y) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    pk_x = fmaxf(pk_x, __shfl_xor(pk_x, d, 64));
    pk_y = fmaxf(pk_y, __shfl_xor(pk_y, d, 64));
  }
  if ((tid & 63) == 0) {
    if (x) atomicMax(a.peaks + 2 * b, __float_as_uint(pk_x));
    if (y) atomicMax(a.peaks + 2 * b + 1, __float_as_uint(pk_y));
  }
}

}  // namespace

// H[rir][p][4096] = FFT(h[p B, (p + 1) B) zero-padded) / N, p < parts of the bank's longest response
__global__ __launch_bounds__(256) void reverb_spectra_kernel(const float* __restrict__ rir, const long long* __restrict__ rir_off,
                                                            const float2* __restrict__ tw, float2* __restrict__ H, int p_max) {
  __shared__ c32 buf[kBufLen];
  const int tid = threadIdx.x, p = blockIdx.x, j = blockIdx.y;
  Twiddles t;
  make_twiddles(t, tw, tid);
  const long long h0 = rir_off[j];
  const int lr = (int)(rir_off[j + 1] - h0);
  c32 v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = tid + 256 * r, q = p * kB + n;
    v[r] = c32{(n < kB && q < lr) ? rir[h0 + q] : 0.f, 0.f};
  }
  fft4096(v, buf, t, tid);
  float2* __restrict__ o = H + ((long long)j * p_max + p) * kN;
  constexpr float inv = 1.0f / (float)kN;
#pragma unroll
  for (int k = 0; k < 16; ++k) o[tid + 256 * k] = c32{v[k].x * inv, v[k].y * inv};
}

// ---- short responses (one or two partitions): everything in one kernel.
// One workgroup = the blocks [k0, k0 + seg_blocks) of the full convolution of utterance blockIdx.y; the spectra of
// its earlier pairs wait in a ring in global memory (every thread reads back the bins it wrote itself).
__global__ __launch_bounds__(256) void reverb_fft_kernel(const ReverbArgs a, const float2* __restrict__ tw,
                                                         const float2* __restrict__ H, int p_max, float2* __restrict__ ring,
                                                         int seg_blocks) {
  __shared__ c32 buf[kBufLen];
  const int tid = threadIdx.x, b = blockIdx.y;
  const Utt u = utterance(a, b, H, p_max);
  if (u.len == 0) return;                               // (the row was zeroed by the caller)
  if (u.ri < 0) {                                       // no reverb for this utterance: copied through
    const long long lo = (long long)blockIdx.x * seg_blocks * kB, hi = lo + (long long)seg_blocks * kB;
    for (long long n = lo + tid; n < hi && n < u.len; n += 256) u.o[n] = u.x[n];
    return;
  }
  const int P = u.P;
  const int k0 = u.k_first + (int)blockIdx.x * seg_blocks;
  if (k0 > u.k_last) return;
  const int k1 = (k0 + seg_blocks < u.k_last + 1) ? k0 + seg_blocks : u.k_last + 1;
  const int kh = (k1 - k0 + 1) / 2;
  Twiddles t;
  make_twiddles(t, tw, tid);
  c32* __restrict__ my_ring = ring + ((long long)(blockIdx.y * gridDim.x + blockIdx.x) * p_max) * kN + tid;
  const c32* __restrict__ Hr = u.Hr + tid;
  float pk_x = 0.f, pk_y = 0.f;
  c32 nx[16];                                           // the next pair's windows, loaded a pair ahead
  load_windows(nx, u, k0 - (P - 1), k0 - (P - 1) + kh, k0 - (P - 1) + kh < k1, tid, pk_x);
  for (int i = -(P - 1); i < kh; ++i) {
    const int ka = k0 + i, kb = ka + kh;
    const bool b_valid = kb < k1;
    c32 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = nx[r];
    if (i + 1 < kh) {
#pragma unroll
      for (int r = 0; r < 8; ++r) nx[r] = c32{v[r + 8].x, kb + 1 < k1 ? v[r + 8].y : 0.f};
      load_windows<8>(nx, u, ka + 1, kb + 1, kb + 1 < k1, tid, pk_x);
    }
    fft4096(v, buf, t, tid);
    // Z_i: into the ring (for the pairs to come) and, times H_0, into the sum; then the earlier pairs
    const int slot = (i + 64 * P) % P;
    c32 acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (P > 1) my_ring[(long long)slot * kN + 256 * k] = v[k];
      acc[k] = u.lr > 0 ? cmul(v[k], Hr[256 * k]) : c32{0.f, 0.f};
    }
    for (int p = 1; p < P && i - p >= -(P - 1); ++p) {
      const c32* __restrict__ zp = my_ring + (long long)((i - p + 64 * P) % P) * kN;
      const c32* __restrict__ hp = Hr + (long long)p * kN;
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k] = cadd(acc[k], cmul(zp[256 * k], hp[256 * k]));
    }
    if (i < 0) continue;
    // inverse transform = the forward one on swapped components (1 / N is in H)
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = c32{acc[k].y, acc[k].x};
    fft4096(v, buf, t, tid);
    store_blocks(v, u, ka, kb, b_valid, tid, pk_y);
  }
  put_peaks(a, b, tid, pk_x, pk_y, true, true);
}

// ---- longer responses: three kernels, so that the products are formed by the OWNER OF A BIN (reverb_mac_kernel),
// who reads every spectrum once however many partitions the response has.
// Pairs of utterance b: nblk = k_last - k_first + 1 blocks, Kh = ceil(nblk / 2); pair i = blocks k_first + i and
// k_first + i + Kh, i in [-(P - 1), Kh): the first P - 1 exist for their spectra only.  Pair i is row q = i + P - 1 of
// Z (b, pairs_cap, N); the products of pair i >= 0 are row i of Y (b, kh_cap, N), components swapped.
// (A workgroup takes kPairsPerWg consecutive pairs and loads the next pair's operands before it transforms the
// current one: a workgroup that lives for one transform waits for its loads with nothing else to do.)
constexpr int kPairsPerWg = 8;
__global__ __launch_bounds__(256) void reverb_fwd_kernel(const ReverbArgs a, const float2* __restrict__ tw, int p_max,
                                                         float2* __restrict__ Z, int pairs_cap) {
  __shared__ c32 buf[kBufLen];
  const int tid = threadIdx.x, b = blockIdx.y, q0 = blockIdx.x * kPairsPerWg;
  const Utt u = utterance(a, b, nullptr, p_max);
  if (u.len == 0) return;
  if (u.ri < 0) {                                       // copied through, 4096 samples per pair of the grid
    const long long lo = (long long)q0 * kN, hi = lo + (long long)kPairsPerWg * kN;
    for (long long n = lo + tid; n < hi && n < u.len; n += 256) u.o[n] = u.x[n];
    return;
  }
  const int kh = (u.k_last - u.k_first + 2) / 2;
  const int q1 = (q0 + kPairsPerWg < kh + u.P - 1) ? q0 + kPairsPerWg : kh + u.P - 1;
  if (q0 >= q1) return;
  Twiddles t;
  make_twiddles(t, tw, tid);
  float pk_x = 0.f;
  c32 nx[16];
  {
    const int ka = u.k_first + q0 - (u.P - 1);
    load_windows(nx, u, ka, ka + kh, ka + kh <= u.k_last, tid, pk_x);
  }
  for (int q = q0; q < q1; ++q) {
    c32 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = nx[r];
    if (q + 1 < q1) {
      const int ka = u.k_first + q + 1 - (u.P - 1);
#pragma unroll
      for (int r = 0; r < 8; ++r) nx[r] = c32{v[r + 8].x, ka + kh <= u.k_last ? v[r + 8].y : 0.f};
      load_windows<8>(nx, u, ka, ka + kh, ka + kh <= u.k_last, tid, pk_x);
    }
    fft4096(v, buf, t, tid);
    c32* __restrict__ z = Z + ((long long)b * pairs_cap + q) * kN + tid;
#pragma unroll
    for (int k = 0; k < 16; ++k) z[256 * k] = v[k];
  }
  put_peaks(a, b, tid, pk_x, 0.f, true, false);
}

// Y_i[n] = sum_p Z_{i-p}[n] H_p[n]: a convolution along the PAIR index, one per bin.  A thread owns one bin n of one
// utterance: H_p[n], p < P8, sit in its column of an LDS array, the last P8 + 8 spectra's bin in its column of a ring;
// eight pairs at a time, their eight operands in registers and shifted by one pair per partition (the sliding
// window of reverb_window_kernel), so a partition costs two LDS reads for eight complex products.  Columns are
// private to their thread: no barrier anywhere.  (Tried: H_p read from global memory / L2 instead of LDS, which halves
// the LDS and doubles the resident workgroups: 48 000 taps 2.29 -> 2.11 ms, 8 000 - 32 000 taps 1 - 3 % slower; not kept.)
constexpr int kMacThreads = 128, kMacT = 8;
__global__ __launch_bounds__(kMacThreads) void reverb_mac_kernel(const ReverbArgs a, const float2* __restrict__ H, int p_max,
                                                                 const float2* __restrict__ Z, int pairs_cap,
                                                                 float2* __restrict__ Y, int kh_cap, int p8_cap) {
  extern __shared__ c32 lds[];                          // Hs[p8_cap][128], then ring[p8_cap + 8][128]
  const int tid = threadIdx.x, b = blockIdx.y, n = blockIdx.x * kMacThreads + tid;
  const Utt u = utterance(a, b, H, p_max);
  if (u.len == 0 || u.ri < 0) return;
  const int P = u.P, P8 = (P + 7) & ~7, R = P8 + kMacT;
  const int kh = (u.k_last - u.k_first + 2) / 2, Q = kh + P - 1;
  c32* __restrict__ Hs = lds + tid;
  c32* __restrict__ rg = lds + (long long)p8_cap * kMacThreads + tid;
  for (int p = 0; p < P8; ++p) Hs[p * kMacThreads] = (p < P && u.lr > 0) ? u.Hr[(long long)p * kN + n] : c32{0.f, 0.f};
  for (int r = 0; r < R; ++r) rg[r * kMacThreads] = c32{0.f, 0.f};
  const c32* __restrict__ z = Z + (long long)b * pairs_cap * kN + n;
  c32* __restrict__ y = Y + (long long)b * kh_cap * kN + n;
  c32 nx[kMacT];
#pragma unroll
  for (int s = 0; s < kMacT; ++s) nx[s] = s < Q ? z[(long long)s * kN] : c32{0.f, 0.f};
  int slot0 = 0;                                        // ring row of pair q0
  for (int q0 = 0; q0 < Q; q0 += kMacT) {
    c32 w[kMacT], acc[kMacT];
#pragma unroll
    for (int s = 0; s < kMacT; ++s) {
      w[s] = nx[s];
      acc[s] = c32{0.f, 0.f};
      int row = slot0 + s;
      if (row >= R) row -= R;
      rg[row * kMacThreads] = nx[s];
    }
#pragma unroll
    for (int s = 0; s < kMacT; ++s) {                   // the next eight, in flight during the products
      const int q = q0 + kMacT + s;
      nx[s] = q < Q ? z[(long long)q * kN] : c32{0.f, 0.f};
    }
    int row = slot0;                                    // row of pair q0 - p - 1 is (slot0 - p - 1) mod R
    for (int p0 = 0; p0 < P8; p0 += kMacT) {
#pragma unroll
      for (int r = 0; r < kMacT; ++r) {
        const c32 h = Hs[(p0 + r) * kMacThreads];
        // before partition p0 + r, w[(s - r) & 7] holds Z_{q0 + s - p0 - r}
#pragma unroll
        for (int s = 0; s < kMacT; ++s) acc[s] = cadd(acc[s], cmul(w[(s - r) & 7], h));
        row = row == 0 ? R - 1 : row - 1;
        w[(7 - r) & 7] = rg[row * kMacThreads];         // Z_{q0 - p0 - r - 1}, for s = 0 of the next partition
      }
    }
#pragma unroll
    for (int s = 0; s < kMacT; ++s) {
      const int i = q0 + s - (P - 1);
      if (i >= 0 && i < kh) y[(long long)i * kN] = c32{acc[s].y, acc[s].x};
    }
    slot0 += kMacT;
    if (slot0 >= R) slot0 -= R;
  }
}

__global__ __launch_bounds__(256) void reverb_inv_kernel(const ReverbArgs a, const float2* __restrict__ tw, int p_max,
                                                         const float2* __restrict__ Y, int kh_cap) {
  __shared__ c32 buf[kBufLen];
  const int tid = threadIdx.x, b = blockIdx.y, i0 = blockIdx.x * kPairsPerWg;
  const Utt u = utterance(a, b, nullptr, p_max);
  if (u.len == 0 || u.ri < 0) return;
  const int kh = (u.k_last - u.k_first + 2) / 2;
  const int i1 = i0 + kPairsPerWg < kh ? i0 + kPairsPerWg : kh;
  if (i0 >= i1) return;
  Twiddles t;
  make_twiddles(t, tw, tid);
  const c32* __restrict__ y = Y + ((long long)b * kh_cap) * kN + tid;
  c32 nx[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) nx[k] = y[(long long)i0 * kN + 256 * k];
  float pk_y = 0.f;
  for (int i = i0; i < i1; ++i) {
    c32 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = nx[k];
    if (i + 1 < i1) {
#pragma unroll
      for (int k = 0; k < 16; ++k) nx[k] = y[(long long)(i + 1) * kN + 256 * k];
    }
    fft4096(v, buf, t, tid);
    const int ka = u.k_first + i, kb = ka + kh;
    store_blocks(v, u, ka, kb, kb <= u.k_last, tid, pk_y);
  }
  put_peaks(a, b, tid, 0.f, pk_y, false, true);
}

hipError_t launch_reverb_spectra(const float* rir, const long long* rir_off, int n_rir, int p_max, const float2* tw, float2* H,
                                 hipStream_t stream) {
  if (n_rir <= 0 || p_max <= 0) return hipSuccess;
  hipLaunchKernelGGL(reverb_spectra_kernel, dim3(p_max, n_rir), dim3(256), 0, stream, rir, rir_off, tw, H, p_max);
  return hipGetLastError();
}

hipError_t launch_reverb_fft(const ReverbArgs& a, int batch, const float2* tw, const float2* H, int p_max, float2* ring,
                             int segments, int seg_blocks, hipStream_t stream) {
  hipLaunchKernelGGL(reverb_fft_kernel, dim3(segments, batch), dim3(256), 0, stream, a, tw, H, p_max, ring, seg_blocks);
  return hipGetLastError();
}

int reverb_mac_lds_bytes(int p_max) {
  const int p8 = (p_max + 7) & ~7;
  return (2 * p8 + kMacT) * kMacThreads * (int)sizeof(c32);
}

hipError_t launch_reverb_fft3(const ReverbArgs& a, int batch, const float2* tw, const float2* H, int p_max, float2* Z,
                              int pairs_cap, float2* Y, int kh_cap, hipStream_t stream) {
  const int p8 = (p_max + 7) & ~7, lds = reverb_mac_lds_bytes(p_max);
  if (lds > 64 * 1024) {                                // (per device and cheap: asked for at every launch)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(reverb_mac_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
  }
  // copied-through utterances take 4096 samples per workgroup of the first kernel: pairs_cap >= out_stride / 4096
  const int per = kPairsPerWg;
  hipLaunchKernelGGL(reverb_fwd_kernel, dim3((pairs_cap + per - 1) / per, batch), dim3(256), 0, stream, a, tw, p_max, Z, pairs_cap);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(reverb_mac_kernel, dim3(kN / kMacThreads, batch), dim3(kMacThreads), lds, stream, a, H, p_max, Z,
                     pairs_cap, Y, kh_cap, p8);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(reverb_inv_kernel, dim3((kh_cap + per - 1) / per, batch), dim3(256), 0, stream, a, tw, p_max, Y, kh_cap);
  return hipGetLastError();
}

}  // namespace wfe
