// wfe_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the Whisper audio front end.
//
// Kernel 1 (frontend_kernel): one workgroup = 128 consecutive frames of one
// utterance.  The padded-signal span those frames cover (20720 positions) is
// staged once into LDS -- truncation, zero padding, reflect padding and (AUG
// builds) the whole augmentation chain are applied while staging, so no padded,
// augmented or framed copy of the waveform ever exists in HBM.  Each of the 4
// waves owns 2 tiles of 16 frames and all 101 folded bins: 4 GEMMs (Ce, Se, Co,
// So; wfe_layout.h) x 7 bin tiles x 2 frame tiles = 56 accumulators of
// v_mfma_f32_16x16x4_f32, K = 100 in 25 k-steps.  B operands (e = x[n]+x[400-n],
// o = x[n]-x[400-n]) are built from two ds_read_b32 of the LDS tile; A operands
// (Hann-windowed twiddles, fp64-built, L2 resident, 200 KB) are fetched one
// k-step ahead straight into registers as 8 x float4 per lane.  The butterflies
// Ce+-Co / Se+-So and |.|^2 run on the accumulators, whose C/D layout (frame on
// the lane, bins in registers) is exactly the B layout the mel MFMA wants, so the
// power spectrum never leaves registers.  Epilogue: log10, per-wave max ->
// one atomicMax per wave on the utterance's max word, store.
//
// Kernel 2 (clamp_kernel): max(x, clipmax - 8), (x + 4) / 4, float4 in place.
//
// Numerics: f32 MFMA is a k-ordered fmaf chain (exact f32).  Tables are built in
// float64 and rounded once.
#include <hip/hip_runtime.h>

#include "wfe_kernels.h"
#include "wfe_layout.h"

namespace wfe {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// ------------------------------------------------------------------ utilities
__device__ __forceinline__ unsigned enc_max(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_max(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// ------------------------------------------------- augmentation chain (AUG only)
// Restates CustomDataset.change_speed / shift / volume / add_noise
// (utils/reader.py:281-340) as a pull-style evaluator: value of the augmented
// signal at one index.  A chain is  seg0 [SPEED seg1]  where seg* hold only
// index-preserving ops (VOLUME, NOISE, AFFINE) and SHIFT.
struct Chain {
  const wfe_aug_op* ops;
  const Resolved* res;
  int n_ops;
  const float* w;
  int len0;          // raw length
  int len1;          // length after the SPEED op (== len0 when there is none)
  int speed_at;      // index of the SPEED op or -1
  double step;       // np.linspace(0, len0, len1) step = len0 / (len1 - 1)
  const float* noise;
  const long long* noise_off;
};

__device__ inline Chain make_chain(const FrontArgs& a, int b, const float* w, int raw_len) {
  Chain c;
  const int lo = a.op_begin[b], hi = a.op_begin[b + 1];
  c.ops = a.ops + lo;
  c.res = a.resolved + lo;
  c.n_ops = hi - lo;
  c.w = w;
  c.len0 = raw_len;
  c.len1 = raw_len;
  c.speed_at = -1;
  c.step = 0.0;
  c.noise = a.noise;
  c.noise_off = a.noise_off;
  for (int k = 0; k < c.n_ops; ++k)
    if (c.ops[k].kind == WFE_AUG_SPEED && c.speed_at < 0) {
      c.speed_at = k;
      c.len1 = c.ops[k].i0;
      c.step = (c.len1 > 1) ? (double)raw_len / (double)(c.len1 - 1) : 0.0;
    }
  return c;
}

// value after ops [lo, hi) at output index idx, for a segment of length L whose
// input is `input(j)`, j in [0, L)
template <class In>
__device__ inline float seg_eval(const Chain& c, int lo, int hi, int L, int idx, In input) {
  int cur = idx;
  for (int k = hi - 1; k >= lo; --k)
    if (c.ops[k].kind == WFE_AUG_SHIFT) cur += c.ops[k].i0;
  float v = (cur >= 0 && cur < L) ? input(cur) : 0.f;
  for (int k = lo; k < hi; ++k) {
    const wfe_aug_op op = c.ops[k];
    const bool in = cur >= 0 && cur < L;
    if (op.kind == WFE_AUG_VOLUME) {
      v = __fmul_rn(v, op.f0);                                   // sample *= 10**(gain/20)
    } else if (op.kind == WFE_AUG_SHIFT) {
      if (!in) v = 0.f;                                          // zero-filled region
      cur -= op.i0;
    } else if (op.kind == WFE_AUG_NOISE) {
      if (in) {
        const Resolved r = c.res[k];
        const long long o0 = c.noise_off[op.i0];
        const int ln = (int)(c.noise_off[op.i0 + 1] - o0);
        const int j = (ln < L) ? (cur % ln) : (cur + op.i1);     // np.pad 'wrap' | slice at start
        // three separate float32 passes in the reference: no contraction
        v = __fadd_rn(__fmul_rn(v, r.a), __fmul_rn(c.noise[o0 + j], r.b));
      }
    } else if (op.kind == WFE_AUG_AFFINE) {
      if (in) v = __fdiv_rn(__fsub_rn(v, c.res[k].a), c.res[k].b);
    }
  }
  return v;
}

// np.interp(np.linspace(0, len0, len1), arange(len0), seg0)[i] -> float32
__device__ inline float speed_eval(const Chain& c, int i) {
  auto raw = [&](int j) { return c.w[j]; };
  auto s0 = [&](int j) { return seg_eval(c, 0, c.speed_at, c.len0, j, raw); };
  double pos = 0.0;
  if (c.len1 > 1) pos = (i == c.len1 - 1) ? (double)c.len0 : __dmul_rn((double)i, c.step);
  const int last = c.len0 - 1;
  if (pos >= (double)last) return s0(last);
  const int j = (int)pos;
  const double a0 = (double)s0(j);
  const double fr = pos - (double)j;
  if (fr == 0.0) return (float)a0;
  const double a1 = (double)s0(j + 1);
  return (float)__dadd_rn(__dmul_rn(a1 - a0, fr), a0);
}

// value of the signal after the first n_ops_used ops at index i
__device__ inline float chain_eval(const Chain& c, int n_ops_used, int i) {
  if (c.speed_at < 0 || c.speed_at >= n_ops_used) {
    auto raw = [&](int j) { return c.w[j]; };
    return seg_eval(c, 0, n_ops_used, c.len0, i, raw);
  }
  auto sp = [&](int j) { return speed_eval(c, j); };
  return seg_eval(c, c.speed_at + 1, n_ops_used, c.len1, i, sp);
}
__device__ inline int chain_len(const Chain& c, int n_ops_used) {
  return (c.speed_at >= 0 && c.speed_at < n_ops_used) ? c.len1 : c.len0;
}

// ------------------------------------------------------------- kernel 1
template <int NMT, bool POWER, bool AUG, int BANK>
__global__ __launch_bounds__(256) void frontend_kernel(const FrontArgs a) {
  __shared__ float lds[kLdsFloats];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * kBlockFrames;
  const int n = a.n_samples;
  const int T = a.n_frames;
  const float* __restrict__ w = a.wav + (a.offsets ? a.offsets[b] : (long long)b * a.row_stride);
  int raw_len = a.lengths[b];
  if (raw_len < 0) raw_len = 0;

  // ---- stage the padded-signal span [160 t0, 160 t0 + 20720) into LDS
  {
    Chain c;
    int len = raw_len;
    if (AUG) {
      c = make_chain(a, b, w, raw_len);
      len = chain_len(c, c.n_ops);
    }
    if (len > n) len = n;                       // truncation=True (HF:300-307)
    const int p0 = t0 * kHop;
    const int i_first = p0 - kHalf;
    const bool interior = !AUG && i_first >= 0 && i_first + kBlockSamples <= len &&
                          ((reinterpret_cast<uintptr_t>(w + i_first) & 15u) == 0);
    if (interior) {
      // all 21 loads in flight before the first LDS write: the tile is one HBM
      // round trip, not 21
      const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(w + i_first);
      constexpr int kIters = (kBlockSamples / 4 + 255) / 256;
      f32x4 buf[kIters];
#pragma unroll
      for (int it = 0; it < kIters; ++it) {
        const int v4 = tid + 256 * it;
        if (v4 < kBlockSamples / 4) buf[it] = src[v4];
      }
#pragma unroll
      for (int it = 0; it < kIters; ++it) {
        const int v4 = tid + 256 * it;
        if (v4 < kBlockSamples / 4) {
          const int q = 4 * v4;                 // 4 | 160, so the four share q / 160
          const int base = q + q / kHop;
          lds[base + 0] = buf[it][0];
          lds[base + 1] = buf[it][1];
          lds[base + 2] = buf[it][2];
          lds[base + 3] = buf[it][3];
        }
      }
    } else {
      constexpr int kBatch = 9;                 // 81 = 9 x 9 positions per thread
      for (int q0 = tid; q0 < kBlockSamples; q0 += 256 * kBatch) {
        float vals[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
          const int q = q0 + 256 * u;
          int i = p0 + q - kHalf;               // index into the length-n padded utterance
          if (i < 0) i = -i;                    // torch.stft center=True, pad_mode="reflect"
          if (i >= n) i = 2 * (n - 1) - i;
          float v = 0.f;                        // right zero padding (padding_value 0.0)
          if (q < kBlockSamples && i >= 0 && i < len) v = AUG ? chain_eval(c, c.n_ops, i) : w[i];
          vals[u] = v;
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
          const int q = q0 + 256 * u;
          if (q < kBlockSamples) lds[pad_q(q)] = vals[u];
        }
      }
    }
  }
  __syncthreads();

  // ---- folded DFT as 4 GEMMs on v_mfma_f32_16x16x4_f32
  const int jl = lane & 15;
  const int kk = lane >> 4;
  const int d_lo = 2 * (kk >> 1) + 16 * (kk & 1);
  const int d_hi = 2 * kk;
  int rowbase[2];
#pragma unroll
  for (int ft = 0; ft < 2; ++ft) rowbase[ft] = 161 * ((wave * 2 + ft) * 16 + jl);

  f32x4 acc[2][4][7];
#pragma unroll
  for (int ft = 0; ft < 2; ++ft)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int bt = 0; bt < 7; ++bt) acc[ft][g][bt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const f32x4* __restrict__ tab = reinterpret_cast<const f32x4*>(a.dft_tab) + lane;
  f32x4 A[4][2];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int h = 0; h < 2; ++h) A[g][h] = tab[(g * 2 + h) * 64];

#pragma unroll 1
  for (int s = 0; s < kSteps; ++s) {
    // next step's twiddles, one step ahead (the reload at s = 24 is harmless)
    const int sn = (s + 1 < kSteps) ? s + 1 : s;
    f32x4 An[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int h = 0; h < 2; ++h) An[g][h] = tab[((sn * 4 + g) * 2 + h) * 64];

    const int nb = (s < 24) ? (32 * (s >> 2) + 4 * (s & 3) + d_lo) : (192 + d_hi);
    float e[2][2], o[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int nn = nb + (c ? 1 : 2);                 // dft_n_of(c, s, kk), 1..200
      const int mm = kNfft - nn;                       // 200..399
      const int off_f = nn + (nn >= kHop ? 1 : 0);
      const int off_b = mm + 1 + (mm >= 2 * kHop ? 1 : 0);
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
        const float xf = lds[rowbase[ft] + off_f];
        const float xb = lds[rowbase[ft] + off_b];
        e[ft][c] = xf + xb;
        o[ft][c] = xf - xb;
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int bt = 0; bt < 7; ++bt)
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
          acc[ft][g][bt] = __builtin_amdgcn_mfma_f32_16x16x4f32(
              A[g][bt >> 2][bt & 3], (g & 1) ? o[ft][g >> 1] : e[ft][g >> 1], acc[ft][g][bt], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int h = 0; h < 2; ++h) A[g][h] = An[g][h];
  }

  // ---- butterflies + power: family 0 = bin k, family 1 = bin 200 - k
  f32x4 P[2][2][7];
#pragma unroll
  for (int ft = 0; ft < 2; ++ft)
#pragma unroll
    for (int bt = 0; bt < 7; ++bt) {
      const f32x4 ce = acc[ft][0][bt], se = acc[ft][1][bt], co = acc[ft][2][bt], so = acc[ft][3][bt];
      const f32x4 rp = ce + co, ip = se + so, rm = ce - co, im = se - so;
      P[ft][0][bt] = rp * rp + ip * ip;
      P[ft][1][bt] = rm * rm + im * im;
    }

  if (POWER) {
    float* __restrict__ po = a.out + (long long)b * kNfreq * T;
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
      const int frame = t0 + (wave * 2 + ft) * 16 + jl;
#pragma unroll
      for (int bt = 0; bt < 7; ++bt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = 16 * bt + 4 * kk + r;
          if (frame < T && k < kFoldBins) {
            po[(long long)k * T + frame] = P[ft][0][bt][r];
            if (k < kFoldBins - 1) po[(long long)(200 - k) * T + frame] = P[ft][1][bt][r];
          }
        }
    }
    return;
  }

  // ---- mel projection on the same MFMA; all-zero (bin tile, mel tile) blocks skipped
  const f32x4* __restrict__ mtab = reinterpret_cast<const f32x4*>(a.mel_tab) + lane;
  float* __restrict__ lo = a.out + (long long)b * a.n_mel * T;
  float wmax = -INFINITY;
#pragma unroll
  for (int ft = 0; ft < 2; ++ft) {
    f32x4 m[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) m[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int bt = 0; bt < 7; ++bt) {
        const unsigned mask = BANK == 1 ? kMelMask80[f * 7 + bt] : BANK == 2 ? kMelMask128[f * 7 + bt] : a.mel_mask[f * 7 + bt];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
          if ((mask >> mt) & 1u) {
            const f32x4 wv = mtab[((f * 7 + bt) * kMelTiles + mt) * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r)
              m[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[r], P[ft][f][bt][r], m[mt], 0, 0, 0);
          }
        }
      }
    const int frame = t0 + (wave * 2 + ft) * 16 + jl;
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mel = 16 * mt + 4 * kk + r;
        if (mel < a.n_mel && frame < T) {
          // clamp(min=1e-10).log10() (HF:159).  log10 of the float32 floor rounds to
          // exactly -10; the floor is by far the most common value (silence, padding),
          // so it is returned exactly rather than through the 1-ulp device log10f.
          const float pw = m[mt][r];
          const float v = (pw > 1e-10f) ? log10f(pw) : -10.0f;
          lo[(long long)mel * T + frame] = v;
          wmax = fmaxf(wmax, v);
        }
      }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, d, 64));
  if (lane == 0 && wmax > -INFINITY) atomicMax(a.clip_max + b, enc_max(wmax));
}

// ------------------------------------------------------------- kernel 2
// log_spec = maximum(log_spec, max - 8.0); (log_spec + 4.0) / 4.0   (HF:160-165)
__global__ __launch_bounds__(256) void clamp_kernel(float* __restrict__ out, const unsigned* __restrict__ clip_max,
                                                    long long per_clip) {
  const int b = blockIdx.y;
  const float floor_v = dec_max(clip_max[b]) - 8.0f;
  float* __restrict__ p = out + (long long)b * per_clip;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if ((per_clip & 3) == 0 && ((reinterpret_cast<uintptr_t>(p) & 15u) == 0)) {
    f32x4* __restrict__ p4 = reinterpret_cast<f32x4*>(p);
    for (long long i = i0; i < per_clip / 4; i += stride) {
      f32x4 v = p4[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (fmaxf(v[j], floor_v) + 4.0f) * 0.25f;
      p4[i] = v;
    }
  } else {
    for (long long i = i0; i < per_clip; i += stride) p[i] = (fmaxf(p[i], floor_v) + 4.0f) * 0.25f;
  }
}

// ------------------------------------------------------ augmentation helpers
template <int THREADS>
__device__ inline double block_sum(double v, double* sh) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wv] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < THREADS / 64; ++i) t += sh[i];
  return t;
}

__global__ __launch_bounds__(1024) void noise_meansq_kernel(const float* __restrict__ noise,
                                                            const long long* __restrict__ off,
                                                            double* __restrict__ meansq) {
  __shared__ double sh[16];
  const int j = blockIdx.x;
  const long long o0 = off[j], ln = off[j + 1] - o0;
  double s = 0.0;
  for (long long i = threadIdx.x; i < ln; i += 1024) {
    const float x = noise[o0 + i];
    s += (double)__fmul_rn(x, x);                    // float32 squares, as sample ** 2
  }
  s = block_sum<1024>(s, sh);
  if (threadIdx.x == 0) meansq[j] = ln > 0 ? s / (double)ln : 0.0;
}

// rms_db (utils/reader.py:350-353) under NumPy-2 float32 scalar semantics
__device__ inline float rms_db_f32(double meansq) { return __fmul_rn(10.0f, log10f((float)meansq)); }

// One workgroup per utterance: walks its chain in order and fills the
// data-dependent scalars of NOISE (gains, utils/reader.py:325-331) and AFFINE
// (mean / std, HF:172-191) ops.  An op's statistics are taken over the signal
// ENTERING it, so earlier ops -- including earlier resolved ones -- are applied.
__global__ __launch_bounds__(1024) void resolve_kernel(const FrontArgs a) {
  __shared__ double sh[16];
  const int b = blockIdx.x;
  const float* w = a.wav + (a.offsets ? a.offsets[b] : (long long)b * a.row_stride);
  int raw_len = a.lengths[b];
  if (raw_len < 0) raw_len = 0;
  const Chain c = make_chain(a, b, w, raw_len);
  Resolved* res = a.resolved + a.op_begin[b];
  for (int k = 0; k < c.n_ops; ++k) {
    const int kind = c.ops[k].kind;
    if (kind != WFE_AUG_NOISE && kind != WFE_AUG_AFFINE) continue;
    int L = chain_len(c, k);
    if (kind == WFE_AUG_AFFINE && L > a.n_samples) L = a.n_samples;   // stats over the kept prefix
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < L; i += 1024) {
      const float x = chain_eval(c, k, i);
      s1 += (double)x;
      s2 += (double)__fmul_rn(x, x);
    }
    s1 = block_sum<1024>(s1, sh);
    s2 = block_sum<1024>(s2, sh);
    if (threadIdx.x == 0) {
      Resolved r{0.f, 1.f};
      if (kind == WFE_AUG_NOISE) {
        const double ms = L > 0 ? s2 / (double)L : 0.0;
        const float gain = fminf(300.0f, __fsub_rn(-20.0f, rms_db_f32(ms)));
        const float g1 = powf(10.0f, __fdiv_rn(gain, 20.0f));
        const float s_db = rms_db_f32(ms * (double)g1 * (double)g1);
        const float n_db = rms_db_f32(a.noise_meansq[c.ops[k].i0]);
        const float ng = fminf(__fsub_rn(__fsub_rn(s_db, n_db), (float)c.ops[k].i2), 300.0f);
        r.a = g1;
        r.b = powf(10.0f, __fdiv_rn(ng, 20.0f));
      } else if (L > 0) {
        const double mean = s1 / (double)L;
        double var = s2 / (double)L - mean * mean;
        if (var < 0.0) var = 0.0;
        r.a = (float)mean;
        r.b = sqrtf(__fadd_rn((float)var, 1e-7f));
      }
      res[k] = r;
    }
    __threadfence();
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void waveform_kernel(const FrontArgs a, float* __restrict__ out,
                                                       int* __restrict__ out_len) {
  const int b = blockIdx.y;
  const float* w = a.wav + (a.offsets ? a.offsets[b] : (long long)b * a.row_stride);
  int raw_len = a.lengths[b];
  if (raw_len < 0) raw_len = 0;
  const Chain c = make_chain(a, b, w, raw_len);
  const int L = chain_len(c, c.n_ops);
  const int keep = L < a.n_samples ? L : a.n_samples;
  if (blockIdx.x == 0 && threadIdx.x == 0 && out_len) out_len[b] = L;
  float* o = out + (long long)b * a.n_samples;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n_samples; i += gridDim.x * 256)
    o[i] = i < keep ? chain_eval(c, c.n_ops, i) : 0.f;
}

// ------------------------------------------------------------- launchers
template <int NMT, bool POWER, bool AUG, int BANK>
static hipError_t launch_k1(const FrontArgs& a, int batch, hipStream_t stream) {
  const dim3 grid((a.n_frames + kBlockFrames - 1) / kBlockFrames, batch);
  hipLaunchKernelGGL((frontend_kernel<NMT, POWER, AUG, BANK>), grid, dim3(256), 0, stream, a);
  return hipGetLastError();
}

// bank: 1 = the 80-mel Whisper bank, 2 = the 128-mel bank (compile-time block
// masks), 0 = any other feature_size (run-time masks).
hipError_t launch_frontend(const FrontArgs& a, int batch, int mode, bool aug, int bank, hipStream_t stream) {
  if (mode == 1)
    return aug ? launch_k1<5, true, true, 0>(a, batch, stream) : launch_k1<5, true, false, 0>(a, batch, stream);
  if (bank == 1)
    return aug ? launch_k1<5, false, true, 1>(a, batch, stream) : launch_k1<5, false, false, 1>(a, batch, stream);
  if (bank == 2)
    return aug ? launch_k1<8, false, true, 2>(a, batch, stream) : launch_k1<8, false, false, 2>(a, batch, stream);
  if (a.n_mel <= 80)
    return aug ? launch_k1<5, false, true, 0>(a, batch, stream) : launch_k1<5, false, false, 0>(a, batch, stream);
  return aug ? launch_k1<8, false, true, 0>(a, batch, stream) : launch_k1<8, false, false, 0>(a, batch, stream);
}

hipError_t launch_clamp(float* out, const unsigned* clip_max, int batch, long long per_clip, hipStream_t stream) {
  long long work = (per_clip + 3) / 4;
  int gx = (int)((work + 256 * 4 - 1) / (256 * 4));
  if (gx < 1) gx = 1;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(clamp_kernel, dim3(gx, batch), dim3(256), 0, stream, out, clip_max, per_clip);
  return hipGetLastError();
}

hipError_t launch_noise_meansq(const float* noise, const long long* off, int n_noise, double* meansq,
                               hipStream_t stream) {
  if (n_noise <= 0) return hipSuccess;
  hipLaunchKernelGGL(noise_meansq_kernel, dim3(n_noise), dim3(1024), 0, stream, noise, off, meansq);
  return hipGetLastError();
}

hipError_t launch_resolve(const FrontArgs& a, int batch, hipStream_t stream) {
  hipLaunchKernelGGL(resolve_kernel, dim3(batch), dim3(1024), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_waveform(const FrontArgs& a, int batch, float* out, int* out_lengths, hipStream_t stream) {
  int gx = (a.n_samples + 256 * 8 - 1) / (256 * 8);
  if (gx < 1) gx = 1;
  if (gx > 128) gx = 128;
  hipLaunchKernelGGL(waveform_kernel, dim3(gx, batch), dim3(256), 0, stream, a, out, out_lengths);
  return hipGetLastError();
}

}  // namespace wfe
