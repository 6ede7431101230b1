// wfe_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the Whisper audio front end.
//
// Kernel 1 (frontend_kernel): one workgroup = 64 consecutive frames of one
// utterance, two workgroups resident per CU (2 waves per SIMD) so that one
// workgroup's staging and epilogue hide under the other's MFMA stream.  The
// padded-signal span the frames cover (10480 positions) is staged once into LDS
// -- truncation, zero padding, reflect padding and (AUG builds) the whole
// augmentation chain are applied while staging, so no padded, augmented or
// framed copy of the waveform ever exists in HBM.  Each of the 4 waves owns one
// tile of 16 frames and all 101 folded bins: 4 GEMMs (Ce, Se, Co, So;
// wfe_layout.h) x 7 bin tiles = 28 accumulators of v_mfma_f32_16x16x4_f32,
// K = 100 in 25 k-steps.  B operands (e = x[n]+x[400-n], o = x[n]-x[400-n]) are
// built from two ds_read_b32 of the LDS tile; A operands (Hann-windowed twiddles,
// fp64-built, L2 resident, 200 KB): the 8 KB of one k-step are shared by the 4
// waves through a double-buffered LDS slot (global -> registers four steps ahead,
// registers -> LDS two steps ahead, ds_read_b128 + b96 into ping-pong fragment
// registers one step ahead), one barrier per k-step.  The butterflies
// Ce+-Co / Se+-So and |.|^2 run on the accumulators, whose C/D layout (frame on
// the lane, bins in registers) is exactly the B layout the mel MFMA wants, so the
// power spectrum never leaves registers.  Epilogue: log10, per-wave max / min ->
// plain stores into the tile's statistics; the block's (n_mel x 64)
// output tile is transposed through LDS and leaves as 256-byte row segments.
//
// Kernel 2 (clamp_kernel): kernel 1 stores g(x) = (x + 4) / 4 and, per tile, its minima and
// per-wave maxima; kernel 2 reduces the utterance maximum from those and applies
// max(., g(max - 8)) only to the half tiles whose minimum is below max - 8
// (bit-identical to clamping first: g is monotone -- and so is the rounding of the optional
// float16 / bfloat16 output).
//
// Augmentation (AUG builds): see stage_shipped / seg_apply; data-dependent gains are
// resolved beforehand by resolve_partial_kernel (its last workgroup per utterance), which also leaves the
// chain compiled for kernel 1 (chain_from_head / copy_cops).
//
// Numerics: f32 MFMA is a k-ordered fmaf chain (exact f32).  Tables are built in
// float64 and rounded once.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "wfe_kernels.h"
#include "wfe_layout.h"

namespace wfe {

// Diagnostic build only (make stamps): s_memtime at the phase boundaries of kernel
// 1, written to a buffer nothing else reads.  The shipped library executes none.
#ifdef WFE_STAMPS
constexpr int kStampRow = 16;   // 0..5 phase boundaries, 6 hardware id, 7 k-loop waits, 8..11 staging passes of SPEED chains
#define WFE_STAMP(i)                                                                   \
  do {                                                                                 \
    if (a.stamps && lane == 0) {                                                       \
      unsigned long long t_;                                                           \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
      a.stamps[((long long)(blockIdx.y * ((a.n_frames + 63) / 64) + tx) * 4 + wave) * kStampRow + (i)] = t_; \
    }                                                                                  \
  } while (0)
#else
#define WFE_STAMP(i) do {} while (0)
#endif

// Diagnostic builds only (make variant DEFS=-DWFE_ABLATE=mask; wrong results): parts of kernel 1 removed to price
// them -- 1 the staging's global loads, 2 butterflies, 4 mel MFMAs, 8 log, 16 statistics reductions, 32 output stores,
// 64 the k-loop's barriers, 128 its requests (twiddle and sample traffic), 256 its e / o adds.
#ifndef WFE_ABLATE
#define WFE_ABLATE 0
#endif
// The same for resample_mfma_kernel (make variant DEFS="-DWFE_FAST_BUILD -DWFE_RS_ABLATE=mask"; wrong results):
// 1 the span's global loads, 2 the fragment loads, 4 the window reads from LDS, 8 the output stores, 16 the MFMAs.
#ifndef WFE_RS_ABLATE
#define WFE_RS_ABLATE 0
#endif
// Twiddle slots in LDS: kSlots k-steps of the table are parked at a time.  With 2 slots the workgroup
// needs a barrier behind every k-step, with 4 (written 3 steps ahead instead of 2) behind every other one.
// Measured equal (4 slots 0.5 % slower, 10 KB more LDS): the barriers are not what the k-loop waits for.
#ifndef WFE_SLOTS
#define WFE_SLOTS 2
#endif
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x3 = __attribute__((ext_vector_type(3))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
// two consecutive floats at any 4-byte boundary: one global_load_dwordx2 (the memory pipe
// spends ~16 cycles per wave on ANY per-lane-addressed load, so fewer, wider loads)
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
// four consecutive floats at any 4-byte boundary: one global_load_dwordx4 (dword alignment is all the memory pipe asks)
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// The 7 twiddles (one per bin tile) a lane needs per GEMM and k-step.  The table
// stores 8 (two float4) but the 8th is padding: it is fetched as float4 + float3
// (global_load_dwordx4 + dwordx3) so that no dead register is the destination of
// an in-flight load -- hipcc reuses such a register at once and then has to wait
// for the load (a vmcnt stall right after issue).
struct Frag7 {
  f32x4 lo;
  f32x3 hi;
  __device__ __forceinline__ float at(int bt) const { return bt < 4 ? lo[bt & 3] : hi[bt & 3]; }
};
__device__ __forceinline__ Frag7 load_frag7(const f32x4* p) {
  Frag7 f;
  f.lo = p[0];
  f.hi = *reinterpret_cast<const f32x3*>(p + 64);
  return f;
}

// ------------------------------------------------------------------ utilities
// float32 -> the 16-bit output formats, round to nearest even (what torch's .half() /
// .bfloat16() do), and back for kernel 2's comparison.  Values are finite here.
__device__ __forceinline__ unsigned to16(float v, bool bf16) {
  if (bf16) {
    const unsigned u = __float_as_uint(v);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
  }
  const _Float16 h = (_Float16)v;
  return (unsigned)__builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float from16(unsigned h, bool bf16) {
  if (bf16) return __uint_as_float(h << 16);
  return (float)__builtin_bit_cast(_Float16, (unsigned short)h);
}

// ---- single, un-contracted IEEE operations.  hipcc compiles device code with
// -ffp-contract=fast and HIP's __fmul_rn / __fadd_rn are the plain operators, so
// a * b + c written with them can still become one fma (it did: 5 % of the noise-mixed
// samples were one ulp off numpy's three separate float32 passes).  An instruction
// emitted under contract(off) carries no `contract` flag and is never fused, inlined or not.
#define WFE_EXACT_OP(NAME, T, EXPR)                 \
  __device__ __forceinline__ T NAME(T a, T b) {     \
    _Pragma("clang fp contract(off)")               \
    return EXPR;                                    \
  }
WFE_EXACT_OP(mul_rn, float, a * b)
WFE_EXACT_OP(add_rn, float, a + b)
WFE_EXACT_OP(sub_rn, float, a - b)
WFE_EXACT_OP(div_rn, float, a / b)
WFE_EXACT_OP(mul_rn, double, a * b)
WFE_EXACT_OP(add_rn, double, a + b)
WFE_EXACT_OP(sub_rn, double, a - b)
#undef WFE_EXACT_OP

// max / min over the 64 lanes without LDS traffic (ds_bpermute queues behind the
// partner workgroup's k-loop reads): four DPP row rotations leave every lane with
// its row's value, four v_readlane combine the rows.  Result uniform.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <bool MAX>
__device__ __forceinline__ float wave_reduce(float v) {
  auto op = [](float x, float y) { return MAX ? fmaxf(x, y) : fminf(x, y); };
  v = op(v, dpp_mov<0x128>(v));      // row_ror:8
  v = op(v, dpp_mov<0x124>(v));      // row_ror:4
  v = op(v, dpp_mov<0x122>(v));      // row_ror:2
  v = op(v, dpp_mov<0x121>(v));      // row_ror:1
  const int iv = __float_as_int(v);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(iv, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(iv, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(iv, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(iv, 48));
  return op(op(r0, r1), op(r2, r3));
}

// ------------------------------------------------- augmentation chain (AUG only)
// Restates CustomDataset.change_speed / shift / volume / add_noise
// (utils/reader.py:281-340) as a pull-style evaluator: value of the augmented
// signal at one index.  A chain is  seg0 [SPEED seg1]  where seg* hold only
// index-preserving ops (VOLUME, NOISE, AFFINE) and SHIFT.
// Where an utterance's samples come from: float32, or interleaved 16-bit PCM
// converted exactly as soundfile.read(dtype='float32') (x / 32768, a power of two)
// and mixed down exactly as librosa.to_mono (np.mean over channels in float32:
// the sum of <= 256 such values is exact, then one float32 division).
struct Source {
  const float* w;
  const short* pcm;
  int ch;
  __device__ __forceinline__ float at(int i) const {
    if (!pcm) return w[i];
    constexpr float kScale = 1.0f / 32768.0f;
    if (ch == 1) return (float)pcm[i] * kScale;
    int t = 0;
    for (int cc = 0; cc < ch; ++cc) t += pcm[(long long)i * ch + cc];
    return div_rn((float)t * kScale, (float)ch);
  }
  // v[u] = sample cur[u], 0 outside [0, L).  Branch-free: every lane loads (index clamped into
  // the utterance) and the value is selected afterwards -- hipcc turns `in ? w[i] : 0` into a
  // branch per element, each load in a basic block of its own with a full wait behind it, which
  // serialises the U round trips.  The format test is outside the unrolled loop for the same reason.
  // v[u] = sample idx[u]; every index is valid (the caller clamped it).  No selects at all.
  template <int U>
  __device__ __forceinline__ void load(const int (&idx)[U], float (&v)[U]) const {
    constexpr float kScale = 1.0f / 32768.0f;
    if (!pcm) {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = w[idx[u]];
    } else if (ch == 1) {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = (float)pcm[idx[u]] * kScale;
    } else if (ch == 2 && (reinterpret_cast<uintptr_t>(pcm) & 3u) == 0) {
      const int* __restrict__ pairs = reinterpret_cast<const int*>(pcm);       // a stereo frame = one aligned dword
      int raw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) raw[u] = pairs[idx[u]];
#pragma unroll
      for (int u = 0; u < U; ++u)
        v[u] = div_rn((float)((int)(short)(raw[u] & 0xffff) + (raw[u] >> 16)) * kScale, 2.0f);
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = at(idx[u]);
    }
  }
  template <int U>
  __device__ __forceinline__ void gather(const int (&cur)[U], int L, float (&v)[U]) const {
    constexpr float kScale = 1.0f / 32768.0f;
    if (L <= 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = 0.f;
      return;
    }
    int idx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) idx[u] = min(max(cur[u], 0), L - 1);
    if (!pcm) {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = w[idx[u]];
    } else if (ch == 1) {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = (float)pcm[idx[u]] * kScale;
    } else if (ch == 2 && (reinterpret_cast<uintptr_t>(pcm) & 3u) == 0) {
      const int* __restrict__ pairs = reinterpret_cast<const int*>(pcm);       // a stereo frame = one aligned dword
      int raw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) raw[u] = pairs[idx[u]];
#pragma unroll
      for (int u = 0; u < U; ++u)
        v[u] = div_rn((float)((int)(short)(raw[u] & 0xffff) + (raw[u] >> 16)) * kScale, 2.0f);
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = at(idx[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (cur[u] >= 0 && cur[u] < L) ? v[u] : 0.f;
  }
};
__device__ __forceinline__ Source make_source(const FrontArgs& a, int b) {
  const long long o = a.offsets ? a.offsets[b] : (long long)b * a.row_stride;
  Source s;
  s.w = a.pcm ? nullptr : a.wav + o;
  s.pcm = a.pcm ? a.pcm + o : nullptr;
  s.ch = a.channels;
  return s;
}

// One op of a chain with everything the per-position evaluation needs, gathered once
// per workgroup into LDS (compile_chain): reading the descriptors, the resolved gains
// and the noise offsets from global memory inside the staging loops costs a chain of
// dependent scalar-load round trips per batch of positions.
struct COp {
  int kind, i0, i1, ln;          // ln: NOISE clip length
  float f0, ra, rb, pad_;        // ra, rb: resolved scalars (NOISE gains / AFFINE mean, std)
  long long o0;                  // NOISE: offset of the clip in the bank
  double inv;                    // NOISE: 1 / ln
};
constexpr int kMaxCops = 16;     // longer chains are evaluated from global memory (slow, correct)
static_assert(sizeof(COp) == kCopBytes, "wfe_kernels.h: kCopBytes");

struct Chain {
  const COp* cops;               // LDS copy of the ops, or null
  const wfe_aug_op* ops;
  const Resolved* res;
  int n_ops;
  Source src;
  int len0;          // raw length
  int len1;          // length after the SPEED op (== len0 when there is none)
  int speed_at;      // index of the SPEED op or -1
  double step;       // np.linspace(0, len0, len1) step = len0 / (len1 - 1)
  const float* noise;
  const long long* noise_off;
};

__device__ inline Chain make_chain(const FrontArgs& a, int b, const Source& src, int raw_len) {
  Chain c;
  const int lo = a.op_begin[b], hi = a.op_begin[b + 1];
  c.cops = nullptr;
  c.ops = a.ops + lo;
  c.res = a.resolved + lo;
  c.n_ops = hi - lo;
  c.src = src;
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
      v = mul_rn(v, op.f0);                                   // sample *= 10**(gain/20)
    } else if (op.kind == WFE_AUG_SHIFT) {
      if (!in) v = 0.f;                                          // zero-filled region
      cur -= op.i0;
    } else if (op.kind == WFE_AUG_NOISE) {
      if (in) {
        const Resolved r = c.res[k];
        const long long o0 = c.noise_off[op.i0];
        const int ln = (int)(c.noise_off[op.i0 + 1] - o0);
        if (ln > 0) {                                            // (an empty clip mixes nothing in, as noise_gather)
          int j = (ln < L) ? (cur % ln) : (cur + op.i1);         // np.pad 'wrap' | slice at start
          j = min(max(j, 0), ln - 1);                            // a start beyond the clip cannot leave it
          // three separate float32 passes in the reference: no contraction
          v = add_rn(mul_rn(v, r.a), mul_rn(c.noise[o0 + j], r.b));
        }
      }
    } else if (op.kind == WFE_AUG_AFFINE) {
      if (in) v = div_rn(sub_rn(v, c.res[k].a), c.res[k].b);
    }
  }
  return v;
}

// np.interp(np.linspace(0, len0, len1), arange(len0), seg0)[i] -> float32
__device__ inline float speed_eval(const Chain& c, int i) {
  auto raw = [&](int j) { return c.src.at(j); };
  auto s0 = [&](int j) { return seg_eval(c, 0, c.speed_at, c.len0, j, raw); };
  double pos = 0.0;
  if (c.len1 > 1) pos = (i == c.len1 - 1) ? (double)c.len0 : mul_rn((double)i, c.step);
  const int last = c.len0 - 1;
  if (pos >= (double)last) return s0(last);
  const int j = (int)pos;
  const double a0 = (double)s0(j);
  const double fr = pos - (double)j;
  if (fr == 0.0) return (float)a0;
  const double a1 = (double)s0(j + 1);
  return (float)add_rn(mul_rn(sub_rn(a1, a0), fr), a0);
}

// value of the signal after the first n_ops_used ops at index i
__device__ inline float chain_eval(const Chain& c, int n_ops_used, int i) {
  if (c.speed_at < 0 || c.speed_at >= n_ops_used) {
    auto raw = [&](int j) { return c.src.at(j); };
    return seg_eval(c, 0, n_ops_used, c.len0, i, raw);
  }
  auto sp = [&](int j) { return speed_eval(c, j); };
  return seg_eval(c, c.speed_at + 1, n_ops_used, c.len1, i, sp);
}
__device__ inline int chain_len(const Chain& c, int n_ops_used) {
  return (c.speed_at >= 0 && c.speed_at < n_ops_used) ? c.len1 : c.len0;
}

// ---- the same chain, op-major over a register batch (kernel 1's staging).
// chain_eval above walks one index through the whole chain: every gather waits for
// the previous one and a sped-up position evaluates seg0 twice.  Here the ops are
// the OUTER loop and U positions the unrolled inner one, so the U gathers of a
// NOISE op (and the U raw loads) are in flight together, and a SPEED chain is
// evaluated in two passes through LDS: seg0 once per SOURCE index into a buffer,
// then interpolation + seg1 per output position.  Per-position arithmetic is the
// identical sequence of float32 / float64 roundings.
constexpr int kDeadIndex = -(1 << 29);       // "not a sample": out of every segment's range, |shift sums| < 2^28

__device__ inline COp make_cop(const Chain& c, int k) {
  const wfe_aug_op op = c.ops[k];
  COp o;
  o.kind = op.kind;
  o.i0 = op.i0;
  o.i1 = op.i1;
  o.f0 = op.f0;
  o.ln = 0;
  o.ra = 0.f;
  o.rb = 1.f;
  o.pad_ = 0.f;
  o.o0 = 0;
  o.inv = 0.0;
  if (op.kind == WFE_AUG_NOISE) {
    o.o0 = c.noise_off[op.i0];
    o.ln = (int)(c.noise_off[op.i0 + 1] - o.o0);
    o.inv = o.ln > 0 ? 1.0 / (double)o.ln : 0.0;
  }
  if (op.kind == WFE_AUG_NOISE || op.kind == WFE_AUG_AFFINE) {
    o.ra = c.res[k].a;
    o.rb = c.res[k].b;
  }
  return o;
}
// ops [0, n_use) -> LDS, one thread per op; the caller synchronises the workgroup and
// sets c.cops.  (n_use stops short of an op whose scalars are not resolved yet.)
__device__ inline bool compile_chain(const Chain& c, COp* cops, int n_use, int tid) {
  if (n_use > kMaxCops) return false;
  if (tid < n_use) cops[tid] = make_cop(c, tid);
  return true;
}
// Kernel 1's set-up: the chain as resolve_utterance left it.  The head alone (one uniform load) serves the
// contiguous staging; the gather forms also want the compiled ops in LDS (copy_cops: a second load round
// trip and a barrier, paid only by the tiles that fall back).
__device__ __forceinline__ Chain chain_from_head(const FrontArgs& a, const ChainHead& h, int b, const Source& src,
                                                 int raw_len) {
  const int lo = a.op_begin[b];
  Chain c;
  c.ops = a.ops + lo;
  c.res = a.resolved + lo;
  c.n_ops = h.n_ops;
  c.src = src;
  c.len0 = raw_len;
  c.len1 = h.speed_at >= 0 ? h.len1 : raw_len;
  c.speed_at = h.speed_at;
  c.step = h.step;
  c.noise = a.noise;
  c.noise_off = a.noise_off;
  c.cops = nullptr;
  return c;
}
// (uniform: every thread of the workgroup calls it, or none)
__device__ __forceinline__ void copy_cops(const FrontArgs& a, const ChainHead& h, int b, Chain& c, COp* cops_lds, int tid) {
  if (!h.compiled) return;                    // more than kMaxCops ops: evaluated from global memory
  if (tid < h.n_ops) cops_lds[tid] = reinterpret_cast<const COp*>(a.cops)[a.op_begin[b] + tid];
  __syncthreads();
  c.cops = cops_lds;
}
// (seg_shift_sum / seg_apply below require the compiled chain, c.cops != null)
__device__ __forceinline__ COp get_cop(const Chain& c, int k) { return c.cops[k]; }

__device__ __forceinline__ int seg_shift_sum(const Chain& c, int lo, int hi) {
  int s = 0;
  for (int k = lo; k < hi; ++k) {
    const COp op = get_cop(c, k);
    if (op.kind == WFE_AUG_SHIFT) s += op.i0;
  }
  return s;
}

// nz[u] = the sample of NOISE op `op`'s clip that meets segment index cur[u] - back: np.pad(...,
// 'wrap') of a shorter clip, the slice at op.i1 of a longer one.  Branch-free (clamped index;
// lanes outside the segment read a valid sample that the caller drops).
template <int U>
__device__ __forceinline__ void noise_gather(const Chain& c, const COp& op, int L, const int (&cur)[U], int back,
                                             float (&nz)[U]) {
  const int ln = op.ln;
  // (uniform conditions stay OUTSIDE the unrolled loops: hipcc does not unswitch them, and a
  // branch per element puts every load into a basic block of its own with a wait behind it)
  if (ln <= 0) {
#pragma unroll
    for (int u = 0; u < U; ++u) nz[u] = 0.f;
    return;
  }
  int j[U];
  if (ln < L) {                                     // cu % ln without the integer divide
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int cu = cur[u] - back;
      const int qd = (int)((double)cu * op.inv);
      int r = cu - qd * ln;
      r = (r < 0) ? r + ln : r;
      j[u] = (r >= ln) ? r - ln : r;
    }
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) j[u] = cur[u] - back + op.i1;
  }
  const float* __restrict__ clip = c.noise + op.o0;
#pragma unroll
  for (int u = 0; u < U; ++u) nz[u] = clip[min(max(j[u], 0), ln - 1)];
}

// applies ops [lo, hi) in order to v[u], the segment's input at cur[u] (0 outside [0, L)).
// k_pre >= 0: the gathers of NOISE op k_pre were issued by the caller (nz_pre).
template <int U>
__device__ __forceinline__ void seg_apply(const Chain& c, int lo, int hi, int L, int (&cur)[U], float (&v)[U],
                                          int k_pre, const float (&nz_pre)[U]) {
  for (int k = lo; k < hi; ++k) {
    const COp op = get_cop(c, k);
    if (op.kind == WFE_AUG_VOLUME) {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = mul_rn(v[u], op.f0);
    } else if (op.kind == WFE_AUG_SHIFT) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!(cur[u] >= 0 && cur[u] < L)) v[u] = 0.f;
        cur[u] -= op.i0;
      }
    } else if (op.kind == WFE_AUG_NOISE) {
      float nz[U];
      if (k == k_pre) {
#pragma unroll
        for (int u = 0; u < U; ++u) nz[u] = nz_pre[u];
      } else {
        noise_gather<U>(c, op, L, cur, 0, nz);
      }
      // three separate float32 passes in the reference (sample *= g; noise *= g; sample += noise)
      if (op.ln > 0) {
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (cur[u] >= 0 && cur[u] < L) v[u] = add_rn(mul_rn(v[u], op.ra), mul_rn(nz[u], op.rb));
      }
    } else if (op.kind == WFE_AUG_AFFINE) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (cur[u] >= 0 && cur[u] < L) v[u] = div_rn(sub_rn(v[u], op.ra), op.rb);
    }
  }
}
template <int U>
__device__ __forceinline__ void seg_apply(const Chain& c, int lo, int hi, int L, int (&cur)[U], float (&v)[U]) {
  float none[U];
#pragma unroll
  for (int u = 0; u < U; ++u) none[u] = 0.f;
  seg_apply<U>(c, lo, hi, L, cur, v, -1, none);
}

// source samples at cur[u], then ops [lo, hi): the gathers of the segment's first NOISE op go
// out BEFORE the source samples are waited for (its indices depend on positions only), so a
// batch costs one memory round trip, not two.
template <int U>
__device__ __forceinline__ void seg_gather_apply(const Chain& c, int lo, int hi, int L, int (&cur)[U], float (&v)[U]) {
  int k_pre = -1, back = 0;
  for (int k = lo; k < hi; ++k) {
    const COp op = get_cop(c, k);
    if (op.kind == WFE_AUG_NOISE) {
      k_pre = k;
      break;
    }
    if (op.kind == WFE_AUG_SHIFT) back += op.i0;
  }
  float nz[U];
  if (k_pre >= 0) {
    noise_gather<U>(c, get_cop(c, k_pre), L, cur, back, nz);
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) nz[u] = 0.f;
  }
  c.src.gather<U>(cur, L, v);
  seg_apply<U>(c, lo, hi, L, cur, v, k_pre, nz);
}

// index into the (truncated, zero-padded, reflect-padded) utterance of staging
// position q, or -1 where the padded signal is zero
__device__ __forceinline__ int padded_index(int q, int p0, int n, int len) {
  int i = p0 + q - kHalf;
  if (i < 0) i = -i;                    // torch.stft center=True, pad_mode="reflect"
  if (i >= n) i = 2 * (n - 1) - i;
  return (q < kBlockSamples && i >= 0 && i < len) ? i : -1;
}

__device__ __forceinline__ double speed_pos(const Chain& c, int i) {
  if (c.len1 <= 1) return 0.0;
  return (i == c.len1 - 1) ? (double)c.len0 : mul_rn((double)i, c.step);
}

// Diagnostic builds (-DWFE_STAMPS): cycles spent in up to four parts of the staging, summed
// over the batches of one tile.  Empty in the shipped library.
struct Ticker {
#ifdef WFE_STAMPS
  unsigned long long prev, acc[4], t_start;
  int batches;
  __device__ __forceinline__ void start() {
    acc[0] = acc[1] = acc[2] = acc[3] = 0;
    batches = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(prev)::"memory");
    t_start = prev;
  }
  __device__ __forceinline__ void tick(int k) {
    unsigned long long t_;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
    acc[k] += t_ - prev;
    prev = t_;
  }
  __device__ __forceinline__ void mark(int k) {          // as tick, without draining the loads in flight
    unsigned long long t_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
    acc[k] += t_ - prev;
    prev = t_;
  }
  __device__ __forceinline__ void batch() { ++batches; }
#else
  __device__ __forceinline__ void start() {}
  __device__ __forceinline__ void tick(int) {}
  __device__ __forceinline__ void mark(int) {}
  __device__ __forceinline__ void batch() {}
#endif
};

// ---- the shipped chain shape, straight-line.  configs/augmentation.json orders its entries
// noise -> speed -> shift -> volume, so every chain the reference's own configuration can draw
// is [NOISE]? [SPEED]? then at most one SHIFT and one VOLUME (those two commute exactly: the
// gain is positive).  For that shape the whole evaluation is unrolled without the op loop of
// seg_apply (whose switch costs ~25 register moves per op and element): absent SHIFT / VOLUME
// are the exact identities shift 0 / gain 1.0f.  Any other order takes the generic evaluator
// below; both produce the same bits (tests: chain orders).
struct Shipped {
  bool ok;
  int noise_k, shift;
  float gain;
};
__device__ inline Shipped match_shipped(const Chain& c) {
  Shipped f;
  f.ok = c.cops != nullptr;
  f.noise_k = -1;
  f.shift = 0;
  f.gain = 1.0f;
  int n_shift = 0, n_vol = 0;
  for (int k = 0; k < c.n_ops && f.ok; ++k) {
    const COp op = get_cop(c, k);
    const bool before_speed = c.speed_at >= 0 && k < c.speed_at;
    const bool after_speed = c.speed_at >= 0 && k > c.speed_at;
    if (op.kind == WFE_AUG_NOISE) {
      if (f.noise_k >= 0 || n_shift || n_vol || after_speed || op.ln <= 0) f.ok = false;
      f.noise_k = k;
    } else if (op.kind == WFE_AUG_SHIFT) {
      if (n_shift++ || before_speed) f.ok = false;
      f.shift = op.i0;
    } else if (op.kind == WFE_AUG_VOLUME) {
      if (n_vol++ || before_speed) f.ok = false;
      f.gain = op.f0;
    } else if (op.kind != WFE_AUG_SPEED) {
      f.ok = false;
    }
  }
  return f;
}

template <bool SPEED, bool NOISE, int U>
__device__ __forceinline__ void stage_shipped(const Chain& c, const Shipped f, float* __restrict__ lds, int p0, int n,
                                              int len, int tid, Ticker& tk) {
  constexpr int G = SPEED ? 2 * U : U;          // source samples per batch: both np.interp neighbours
  const int L0 = c.len0, last = L0 - 1;
  const int L1 = SPEED ? c.len1 : L0;
  COp nop = get_cop(c, NOISE ? f.noise_k : 0);
  const float* __restrict__ clip = c.noise + (NOISE ? nop.o0 : 0);
  const bool wrap = nop.ln < L0;
  // tile without reflection or padding: position q is sample p0 + q - 200
  const bool plain = p0 - kHalf >= 0 && p0 - kHalf + kBlockSamples <= len;
  // float32 source of >= 2 samples (and a clip of >= 2): neighbours fetched as pairs
  const bool pairs = SPEED && c.src.w && L0 >= 2 && (!NOISE || nop.ln >= 2);
  const float clip0 = NOISE ? clip[0] : 0.f;
  if (L0 <= 0) {                                  // empty utterance: the chain of nothing is silence
    for (int q = tid; q < kBlockSamples; q += 256) lds[pad_q(q)] = 0.f;
    return;
  }
  for (int q0 = tid; q0 < kBlockSamples; q0 += 256 * U) {
    int cur[U], idx[G];
    bool in[U];
    double fr[U];
    if (plain) {
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = (q0 + 256 * u < kBlockSamples) ? p0 + q0 + 256 * u - kHalf + f.shift : kDeadIndex;
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = padded_index(q0 + 256 * u, p0, n, len);
        cur[u] = i >= 0 ? i + f.shift : kDeadIndex;
      }
    }
    // np.interp(np.linspace(0, len0, len1), arange(len0), seg0): position cur * step, clamped to the
    // last sample (np.interp right of xp[-1] returns fp[-1]; linspace's exact last point len0 lands
    // there too, and step is 0 for a one-sample result) -- then the fraction is 0 and j + 1 unused
#pragma unroll
    for (int u = 0; u < U; ++u) {
      in[u] = cur[u] >= 0 && cur[u] < L1;      // outside: the zero fill of the shift / beyond the signal
      if (SPEED) {
        const double pos = fmin(mul_rn((double)(in[u] ? cur[u] : 0), c.step), (double)last);
        const int j = (int)pos;
        fr[u] = pos - (double)j;
        idx[2 * u] = j;
        idx[2 * u + 1] = min(j + 1, last);
      } else {
        idx[u] = in[u] ? cur[u] : 0;
      }
    }
    tk.tick(0);
    // the noise gathers go out before the source samples are waited for: one round trip per batch
    float nz[G], x[G];
    if (SPEED && pairs) {
      // both np.interp neighbours in ONE load per lane: the pair starts at min(j, last - 1); at
      // the last sample the right neighbour is not used (fraction 0).  Noise alike, with the
      // clip's first sample (uniform) as the right neighbour where the wrap falls between the two.
      f32x2u np_[U], xp[U];
      bool nhi[U];
      if (NOISE) {
        int m0[U];
        if (wrap) {                                 // (uniform: outside the unrolled loop)
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int qd = (int)((double)idx[2 * u] * nop.inv);
            int r = idx[2 * u] - qd * nop.ln;
            r = (r < 0) ? r + nop.ln : r;
            m0[u] = (r >= nop.ln) ? r - nop.ln : r;
          }
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u) m0[u] = idx[2 * u] + nop.i1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int mc = min(max(m0[u], 0), nop.ln - 1);
          const int mb = min(mc, nop.ln - 2);
          nhi[u] = mc > mb;
          np_[u] = *reinterpret_cast<const f32x2u*>(clip + mb);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) xp[u] = *reinterpret_cast<const f32x2u*>(c.src.w + min(idx[2 * u], last - 1));
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool xhi = idx[2 * u] == last;
        x[2 * u] = xhi ? xp[u].y : xp[u].x;
        x[2 * u + 1] = xp[u].y;
        if (NOISE) {
          nz[2 * u] = nhi[u] ? np_[u].y : np_[u].x;
          nz[2 * u + 1] = nhi[u] ? clip0 : np_[u].y;
        }
      }
    } else {
      if (NOISE) {
        int m[G];
        if (wrap) {                                 // np.pad(noise, 'wrap'): idx % ln without the integer divide
#pragma unroll
          for (int g = 0; g < G; ++g) {
            if (SPEED && (g & 1)) {                 // the right neighbour: one step on from the left one
              const int nx = (m[g - 1] + 1 == nop.ln) ? 0 : m[g - 1] + 1;
              m[g] = (idx[g] == idx[g - 1]) ? m[g - 1] : nx;
            } else {
              const int qd = (int)((double)idx[g] * nop.inv);
              int r = idx[g] - qd * nop.ln;
              r = (r < 0) ? r + nop.ln : r;
              m[g] = (r >= nop.ln) ? r - nop.ln : r;
            }
          }
        } else {
#pragma unroll
          for (int g = 0; g < G; ++g) m[g] = idx[g] + nop.i1;
        }
#pragma unroll
        for (int g = 0; g < G; ++g) nz[g] = clip[min(max(m[g], 0), nop.ln - 1)];
      }
      c.src.load<G>(idx, x);                        // (indices are valid by construction; L0 > 0)
    }
    tk.tick(1);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float v;
      if (SPEED) {
        float s0 = x[2 * u], s1 = x[2 * u + 1];
        if (NOISE) {                              // sample *= gain; noise *= gain; sample += noise: three roundings
          s0 = add_rn(mul_rn(s0, nop.ra), mul_rn(nz[2 * u], nop.rb));
          s1 = add_rn(mul_rn(s1, nop.ra), mul_rn(nz[2 * u + 1], nop.rb));
        }
        const double a0 = (double)s0, a1 = (double)s1;
        const float mid = (float)add_rn(mul_rn(sub_rn(a1, a0), fr[u]), a0);
        v = (fr[u] == 0.0) ? s0 : mid;
      } else {
        v = x[u];
        if (NOISE) v = add_rn(mul_rn(v, nop.ra), mul_rn(nz[u], nop.rb));
      }
      v = mul_rn(in[u] ? v : 0.f, f.gain);
      const int q = q0 + 256 * u;
      if (q < kBlockSamples) lds[pad_q(q)] = v;
    }
    tk.tick(2);
    tk.batch();
  }
}

// ---- the shipped chain over a float32 source, contiguous form.  stage_shipped above gathers per
// OUTPUT position: ~50 VALU instructions each, and two 4- or 8-byte loads per position.  What that costs
// (tools/phase_breakdown.py, tools/micro/vmem_issue.hip): a CU keeps only ~64 wave-level loads in flight, so
// a tile's ~500 narrow loads are 8 dependent memory round trips of ~1.5k cycles however they are batched
// -- the staging of a SPEED + NOISE tile ran 28k cycles on an otherwise EMPTY CU, as long as the whole
// k-loop, and twice that next to the partner workgroup.  A tile without reflection or padding reads a
// CONTIGUOUS run of the source, so here
//   * source and noise travel as 16-byte loads through buffer resources (lane offset + immediate: no
//     address arithmetic, no clamps): 11 + 11 loads per lane without SPEED, 3 x (5 + 5) with;
//   * the np.pad-wrap index of the noise advances by 1024 per row with one unsigned min; the one lane in
//     ~ln / 4 whose four samples straddle the end of the clip is redone sample by sample afterwards;
//   * a SPEED chain mixes the noise ONCE per source sample into the (still unused) twiddle / output-tile
//     area of LDS and interpolates from there with one ds_read2 per position, in three chunks of
//     15 x 256 positions (3840 = 24 hops, so the image address is lane base + immediate throughout);
//   * the image address uses five lane bases: pad_q(q + 1280) = pad_q(q) + 1288.
// Same roundings in the same order as stage_shipped / chain_eval (tests: chain goldens).  Returns false
// (nothing staged, no barrier passed) when the tile or the chain does not fit -- reflection, padding, the
// end of the utterance inside the fetched run, 16-bit source, clips shorter than 1024 samples, rates above
// 1.33 -- and the caller falls back to the gather form.
constexpr int kChunkRows = 15;                 // output rows (256 positions each) per chunk
constexpr int kSrcVecs = 5;                    // 16-byte source loads per lane and chunk: 5120 floats of LDS
constexpr int kSrcFloats = kSrcVecs * 1024;
constexpr int kStageAuxFloats = kSrcFloats + 64;
constexpr int kImageRows = (kBlockSamples + 255) / 256;          // 41, the last holds 240 positions
constexpr int kLastRowLanes = kBlockSamples - 256 * (kImageRows - 1);
constexpr int kImageVecs = (kBlockSamples + 1023) / 1024;        // 11 rows of 256 x 4 positions without SPEED

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long uni(long long v) {
  const unsigned lo = (unsigned)uni((int)(v & 0xffffffffll)), hi = (unsigned)uni((int)(v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}
// floats [first, first + count) of p as a raw buffer: loads beyond it return 0 instead of faulting
__device__ __forceinline__ __amdgpu_buffer_rsrc_t f32_rsrc(const float* p, long long first, long long count) {
  count = count < 0 ? 0 : (count > (1ll << 29) ? (1ll << 29) : count);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p + first), 0, (int)(count * 4), 0x00020000);
}
__device__ __forceinline__ f32x4 buf_f32x4(__amdgpu_buffer_rsrc_t r, int byte_offset) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_offset, 0, 0));
}
__device__ __forceinline__ int mod_clip(int s, int ln, double inv) {       // s mod ln, s >= 0, without the integer divide
  const int qd = (int)((double)s * inv);
  int r = s - qd * ln;
  r = (r < 0) ? r + ln : r;
  return (r >= ln) ? r - ln : r;
}

template <bool SPEED, bool NOISE>
__device__ __forceinline__ bool stage_contiguous(const Chain& c, const ChainHead& f, float* __restrict__ lds,
                                                 float* __restrict__ aux, int p0, int len, int tid, Ticker& tk) {
  const int L0 = c.len0, last = L0 - 1, L1 = SPEED ? c.len1 : L0;
  const int cur0 = p0 - kHalf + f.shift;          // position q is index cur0 + q of the chain's last segment
  const int ln = NOISE ? f.noise_len : 0, start = NOISE ? f.noise_start : 0;
  const bool wrap = NOISE && ln < L0;
  const double step = c.step, dlast = (double)last;
  auto src_of = [&](int cu) { return (int)fmin(mul_rn((double)cu, step), dlast); };
  bool ok = c.src.w && L0 > 0 && p0 - kHalf >= 0 && p0 - kHalf + kBlockSamples <= len;
  if (NOISE) ok = ok && (wrap ? ln >= 1024 : (start >= 0 && (long long)start + L0 <= ln));
  if (SPEED) {
    // every chunk's run of kSrcFloats source samples lies inside the utterance (the last chunk's starts furthest in)
    const int first = cur0 + 256 * kChunkRows * ((kImageRows - 1) / kChunkRows);
    ok = ok && L1 >= 1 && step >= 0.0 && step * (double)(kChunkRows * 256 - 1) + 3.0 <= (double)kSrcFloats &&
         (long long)src_of(min(max(first, 0), L1 - 1)) + kSrcFloats <= L0;
  } else {
    ok = ok && cur0 >= 0 && (long long)cur0 + kImageVecs * 1024 <= L0;
  }
  if (!uni((int)ok)) return false;

  const float* __restrict__ clip = c.noise + (NOISE ? f.noise_o0 : 0ll);
  const float ra = f.ra, rb = f.rb, gain = f.gain;
  const double inv = f.noise_inv;
  const int t16 = tid * 16;
  const unsigned ln4 = (unsigned)ln * 4u;
  // sample *= gain; noise *= gain; sample += noise: three float32 roundings (utils/reader.py:330-340)
  auto mixed = [&](float x, float nz) { return NOISE ? add_rn(mul_rn(x, ra), mul_rn(nz, rb)) : x; };
  // R rows (256 lanes x 4 samples) of the source from sample s0 on, and the noise that meets them.  A lane
  // whose four noise samples would straddle the end of the clip reads the clip's last four instead and is
  // counted in `split`.
  // `need` (uniform): how many samples from s0 on the caller will read.  Loads beyond them are out of the buffer's
  // range -- they return 0 and fetch nothing (round 3: a chunk of a slowed-down utterance needs 4 of its 5 rows,
  // the last row of a tile without SPEED 240 of its 1024 samples) -- with no branch in the load sequence.
  bool split = false;
  auto fetch = [&](auto& X, auto& NZ, int s0, int need, auto rows) {
    constexpr int R = decltype(rows)::value;
    const long long left = (long long)L0 - s0;
    const __amdgpu_buffer_rsrc_t rx = f32_rsrc(c.src.w, s0, left < need ? left : (long long)need);
#pragma unroll
    for (int i = 0; i < R; ++i) X[i] = buf_f32x4(rx, t16 + 4096 * i);
    if (NOISE) {
      if (wrap) {                                   // np.pad(noise, 'wrap'): s mod ln, advanced by 1024 per row
        const __amdgpu_buffer_rsrc_t rn = f32_rsrc(clip, 0, ln);
        unsigned mb = (unsigned)mod_clip(s0 + 4 * tid, ln, inv) * 4u;
        const unsigned top = ln4 - 16u;
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const bool wanted = 1024 * i < need;      // (uniform) whole rows beyond `need`: an offset outside the clip
          split = split || (wanted && mb > top);
          NZ[i] = buf_f32x4(rn, wanted ? (int)min(mb, top) : 0x7ffffff0);
          mb += 4096u;
          mb = min(mb, mb - ln4);                   // ln >= 1024: at most one wrap per row
        }
      } else {                                      // the slice noise[start : start + L0]
        const long long nleft = (long long)ln - s0 - start;
        const __amdgpu_buffer_rsrc_t rn = f32_rsrc(clip, (long long)s0 + start, nleft < need ? nleft : (long long)need);
#pragma unroll
        for (int i = 0; i < R; ++i) NZ[i] = buf_f32x4(rn, t16 + 4096 * i);
      }
    }
  };
  // the straddling lanes again, sample by sample (wave-uniform branch: taken by one wave in ~ln / 1024 rows);
  // put(row, element, value) stores the mixed sample of source index s0 + 4 tid + 1024 row + element
  auto redo_split = [&](int s0, int rows, auto put) {
    if (!NOISE || !__builtin_amdgcn_ballot_w64(split)) return;
    if (split) {
      for (int i = 0; i < rows; ++i) {
        const int s = s0 + 4 * tid + 1024 * i;
        const int m = mod_clip(s, ln, inv);
        if (m > ln - 4) {
          for (int e = 0; e < 4; ++e) {
            const int me = (m + e >= ln) ? m + e - ln : m + e;
            put(i, e, mixed(s + e < L0 ? c.src.w[s + e] : 0.f, clip[me]));
          }
        }
      }
    }
    split = false;
  };

  if (!SPEED) {
    int pb[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) pb[r] = pad_q(4 * tid + 1024 * r);      // pad_q(q + 5120) = pad_q(q) + 5152
    f32x4 X[kImageVecs], NZ[kImageVecs];
    fetch(X, NZ, cur0, kBlockSamples, std::integral_constant<int, kImageVecs>{});
    constexpr int kTailLanes = (kBlockSamples - 1024 * (kImageVecs - 1)) / 4;        // 60 lanes in the last row
#pragma unroll
    for (int row = 0; row < kImageVecs; ++row) {
      if (row < kImageVecs - 1 || tid < kTailLanes) {
        float* __restrict__ o = lds + pb[row % 5] + 5152 * (row / 5);     // 4 | 160: the four share q / 160
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = mul_rn(mixed(X[row][e], NZ[row][e]), gain);
      }
    }
    redo_split(cur0, kImageVecs, [&](int i, int e, float v) {
      const int q = 4 * tid + 1024 * i + e;
      if (q < kBlockSamples) lds[pad_q(q)] = mul_rn(v, gain);
    });
    return true;
  }

  // SPEED: np.interp(np.linspace(0, len0, len1), arange(len0), seg0) at cur = position cur * step, clamped to
  // the last sample (see stage_shipped).  Chunk k covers image rows [15 k, 15 k + rows).
  int pb[5];
#pragma unroll
  for (int r = 0; r < 5; ++r) pb[r] = pad_q(tid + 256 * r);
  auto image = [&](int row, float v) {            // position tid + 256 row of the image
    if (row < kImageRows - 1 || tid < kLastRowLanes) lds[pb[row % 5] + 1288 * (row / 5)] = v;
  };
  f32x4 X[kSrcVecs], NZ[kSrcVecs];
  struct Span { int lo, hi, jlo, need; };
  auto span_of = [&](int k, int out_rows) {
    const int first = cur0 + 256 * kChunkRows * k;
    Span s;
    s.lo = uni(min(max(first, 0), L1 - 1));
    s.hi = uni(min(max(first + 256 * out_rows - 1, 0), L1 - 1));
    s.jlo = uni(src_of(s.lo));                      // (src_of(hi) + 1 - jlo < kSrcFloats by the check on step above)
    s.need = uni(src_of(s.hi)) - s.jlo + 2;         // np.interp reads source samples jlo .. src_of(hi) + 1
    return s;
  };
  constexpr int kChunks = (kImageRows + kChunkRows - 1) / kChunkRows;       // 3
  Span nxt = span_of(0, kChunkRows);
  tk.mark(0);
  fetch(X, NZ, nxt.jlo, nxt.need, std::integral_constant<int, kSrcVecs>{});
#pragma unroll
  for (int k = 0; k < kChunks; ++k) {
    const Span s = nxt;
    const int out_rows = (k + 1 < kChunks) ? kChunkRows : kImageRows - kChunkRows * k;
#pragma unroll
    for (int i = 0; i < kSrcVecs; ++i) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = mixed(X[i][e], NZ[i][e]);
      *reinterpret_cast<f32x4*>(aux + 4 * tid + 1024 * i) = v;
    }
    redo_split(s.jlo, kSrcVecs, [&](int i, int e, float v) { aux[4 * tid + 1024 * i + e] = v; });
    tk.mark(1);
    if (k + 1 < kChunks) {                          // the next chunk's rows travel under this chunk's interpolation
      nxt = span_of(k + 1, (k + 2 < kChunks) ? kChunkRows : kImageRows - kChunkRows * (k + 1));
      fetch(X, NZ, nxt.jlo, nxt.need, std::integral_constant<int, kSrcVecs>{});
    }
    __syncthreads();
    tk.mark(2);
    const int first = cur0 + 256 * kChunkRows * k + tid;
    // a chunk whose positions all lie inside [0, L1) -- nearly every one -- needs neither the clamp of the
    // position nor the zero fill, and counts its positions in float64 directly
    const bool inside = uni((int)(cur0 + 256 * kChunkRows * k >= 0 && cur0 + 256 * (kChunkRows * k + out_rows) - 1 < L1));
    auto interpolate = [&](double pos, int u) {
      const int j = (int)pos;
      const double fr = __builtin_amdgcn_fract(pos);              // pos - floor(pos), exact; pos >= 0
      const float s0 = aux[j - s.jlo], s1 = aux[j - s.jlo + 1];   // (at j == last the fraction is 0, s1 unused)
      const float mid = (float)add_rn(mul_rn(sub_rn((double)s1, (double)s0), fr), (double)s0);
      return (fr == 0.0) ? s0 : mid;
    };
    if (inside) {
      const double dfirst = (double)first;
#pragma unroll
      for (int u = 0; u < out_rows; ++u)
        image(kChunkRows * k + u, mul_rn(interpolate(fmin(mul_rn(dfirst + (double)(256 * u), step), dlast), u), gain));
    } else {
#pragma unroll
      for (int u = 0; u < out_rows; ++u) {
        const int cu = first + 256 * u;
        const int cc = min(max(cu, s.lo), s.hi);    // keeps j inside the chunk; positions outside [0, L1) are 0 below
        float v = interpolate(fmin(mul_rn((double)cc, step), dlast), u);
        v = ((unsigned)cu < (unsigned)L1) ? v : 0.f;  // outside: the zero fill of the shift / beyond the signal
        image(kChunkRows * k + u, mul_rn(v, gain));
      }
    }
    __syncthreads();                                // the area is rewritten by the next chunk, then by the twiddles
    tk.mark(3);
    tk.batch();
  }
  return true;
}

// Stages the augmented padded signal of one workgroup into the LDS image: the gather forms (every tile, every
// chain); stage_from_head tries the contiguous form first.
// (Kernel 1's first twiddles are requested AFTER this: issued behind the first fetch instead, they cost
// 32 registers through the staging -- spills at the 256-register limit -- and measured 0.853 against 0.831 ms.)
__device__ __forceinline__ void stage_augmented(const FrontArgs& a, const ChainHead& h, int b, Chain& c, COp* cops_lds,
                                                float* __restrict__ lds, float* __restrict__ aux, int p0, int n, int len,
                                                int tid, unsigned long long* st = nullptr) {
  Ticker tk;
  tk.start();
  auto report = [&]() {
#ifdef WFE_STAMPS
    if (st) {
      st[8] = tk.acc[0];
      st[9] = tk.acc[1];
      st[10] = tk.acc[2];
      st[11] = tk.batches;
      st[12] = tk.acc[3];
      st[13] = tk.t_start;                      // absolute: entry of the staging, and its exit
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st[14])::"memory");
    }
#endif
  };
  if (h.shipped && !(a.tune & 1)) {             // (WFE_TUNE & 1: the gather forms everywhere, for A/B and the identity test)
    const bool sp = h.speed_at >= 0, nz = h.noise_k >= 0;
    const bool done = sp ? (nz ? stage_contiguous<true, true>(c, h, lds, aux, p0, len, tid, tk)
                               : stage_contiguous<true, false>(c, h, lds, aux, p0, len, tid, tk))
                         : (nz ? stage_contiguous<false, true>(c, h, lds, aux, p0, len, tid, tk)
                               : stage_contiguous<false, false>(c, h, lds, aux, p0, len, tid, tk));
    if (done) {
      report();
      return;
    }
  }
  copy_cops(a, h, b, c, cops_lds, tid);
  if (!c.cops) {                                // more than kMaxCops ops: per-position evaluation from global memory
    for (int q = tid; q < kBlockSamples; q += 256) {
      const int i = padded_index(q, p0, n, len);
      lds[pad_q(q)] = i >= 0 ? chain_eval(c, c.n_ops, i) : 0.f;
    }
    return;
  }
  {
    const Shipped f = match_shipped(c);
    if (f.ok) {
      if (c.speed_at >= 0) {
        if (f.noise_k >= 0) stage_shipped<true, true, 14>(c, f, lds, p0, n, len, tid, tk);
        else stage_shipped<true, false, 21>(c, f, lds, p0, n, len, tid, tk);
      } else {
        if (f.noise_k >= 0) stage_shipped<false, true, 41>(c, f, lds, p0, n, len, tid, tk);
        else stage_shipped<false, false, 41>(c, f, lds, p0, n, len, tid, tk);
      }
      report();
      return;
    }
  }
  if (c.speed_at < 0) {
    // 41 positions per thread in 2 batches of 21: each batch pays one HBM round trip for the raw
    // samples and one gather round trip per NOISE op (~3k cycles each under load), so few, large
    // batches -- possible since the gathers are branch-free
    constexpr int U = 21;
    const int S = seg_shift_sum(c, 0, c.n_ops);
    for (int q0 = tid; q0 < kBlockSamples; q0 += 256 * U) {
      int cur[U];
      float v[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = padded_index(q0 + 256 * u, p0, n, len);
        ok[u] = i >= 0;
        cur[u] = ok[u] ? i + S : kDeadIndex;
      }
      seg_gather_apply<U>(c, 0, c.n_ops, c.len0, cur, v);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int q = q0 + 256 * u;
        if (q < kBlockSamples) lds[pad_q(q)] = ok[u] ? v[u] : 0.f;
      }
    }
    return;
  }

  // SPEED chain: np.interp needs seg0 at the two source indices j, j + 1 around every output
  // position.  Both are evaluated per output position, in the same register batch (2 U
  // gathers per NOISE op, all in flight together): seg0 is computed twice per sample, but
  // there is no pass through LDS, no barrier and no limit on the speed rate, and a batch is
  // one round trip.  (A two-pass version -- seg0 per SOURCE index into LDS, then
  // interpolation -- measured 39k + 37k cycles per tile in kernel 1: 5 + 4 round trips.)
  constexpr int U = 11;
  const int S0 = seg_shift_sum(c, 0, c.speed_at);
  const int S1 = seg_shift_sum(c, c.speed_at + 1, c.n_ops);
  const int last = c.len0 - 1;
  for (int q0 = tid; q0 < kBlockSamples; q0 += 256 * U) {
    int cur[U], cur0[2 * U];
    float v[U], v0[2 * U];
    double fr[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = padded_index(q0 + 256 * u, p0, n, len);
      ok[u] = i >= 0;
      cur[u] = ok[u] ? i + S1 : kDeadIndex;
      const bool in = cur[u] >= 0 && cur[u] < c.len1;
      const double pos = speed_pos(c, in ? cur[u] : 0);
      const bool at_end = pos >= (double)last;           // np.interp right of xp[-1]: fp[-1]
      const int j = at_end ? last : (int)pos;
      fr[u] = at_end ? 0.0 : pos - (double)j;
      cur0[2 * u] = in ? j + S0 : kDeadIndex;
      cur0[2 * u + 1] = in ? (j < last ? j + 1 : last) + S0 : kDeadIndex;
    }
    seg_gather_apply<2 * U>(c, 0, c.speed_at, c.len0, cur0, v0);
    tk.tick(1);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double a0 = (double)v0[2 * u], a1 = (double)v0[2 * u + 1];
      const float mid = (float)add_rn(mul_rn(sub_rn(a1, a0), fr[u]), a0);
      v[u] = (fr[u] == 0.0) ? v0[2 * u] : mid;           // (out-of-range positions: both gathers are 0)
    }
    seg_apply<U>(c, c.speed_at + 1, c.n_ops, c.len1, cur, v);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = q0 + 256 * u;
      if (q < kBlockSamples) lds[pad_q(q)] = ok[u] ? v[u] : 0.f;
    }
    tk.tick(2);
    tk.batch();
  }
  report();
}

// Which 64-frame tile of utterance blockIdx.y this workgroup takes.  Workgroups are dealt round-robin to the 8
// XCDs by their linear index (blockIdx.x fastest).  With an odd number of tiles per utterance tile tx of
// successive utterances rotates over all XCDs; with an even number it would land on the same few, and in a batch
// of ragged utterances -- whose early tiles are live and late tiles silent -- those XCDs would carry the launch
// (48 tiles per utterance: 0.356 ms against 0.249 for 47 or 49, tools/xcd_parity.py).  An even count is therefore
// skewed by the utterance index, which makes it rotate like an odd one.
__device__ __forceinline__ int tile_of_block(int nblk) {
#ifdef WFE_NO_SKEW
  return (int)blockIdx.x;
#else
  return (nblk & 1) ? (int)blockIdx.x : (int)((blockIdx.x + blockIdx.y) % (unsigned)nblk);
#endif
}

// ------------------------------------------------------------- kernel 1
template <int NMT, bool POWER, bool AUG, int BANK>
__global__ __launch_bounds__(256, 2) void frontend_kernel(const FrontArgs a) {
  // waveform image | twiddle slots, reused as the output tile after the k-loop
  constexpr int kSlots = WFE_SLOTS;                       // 2 or 4
  constexpr int kAuxTile = (16 * NMT * kTileStride > kSlots * kDftStepFloats) ? 16 * NMT * kTileStride : kSlots * kDftStepFloats;
  constexpr int kAuxFloats = (AUG && kAuxTile < kStageAuxFloats) ? kStageAuxFloats : kAuxTile;   // (AUG: stage_contiguous)
  __shared__ __attribute__((aligned(16))) float lds[kLdsWaveFloats + kAuxFloats];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int b = blockIdx.y;
  const int n = a.n_samples;
  const int T = a.n_frames;
  const int nblk = (T + kBlockFrames - 1) / kBlockFrames;   // == gridDim.x, without the late kernarg load
  const int tx = tile_of_block(nblk);
  const int t0 = tx * kBlockFrames;
  // AUG: the staging outranks the partner workgroup's MFMA stream in the SIMD's arbiter (-1.4 % on the
  // augmented kernel; the k-loop at priority 2 or 3 instead, or both at 0, measured the same or worse;
  // without augmentation the staging is short and the switch costs 1.4 %)
  if (AUG) __builtin_amdgcn_s_setprio(3);
#ifdef WFE_ASYM
  // experiment: break the lockstep of the two resident workgroups of a CU.  HW_ID[3:0] is the wave's slot on its
  // SIMD: 0 or 1 here, different for the two resident workgroups, the same for the 4 waves of one (stamps).
  const bool slot0 = (__builtin_amdgcn_s_getreg((3 << 11) | 4) & 1u) == 0u;
#if WFE_ASYM == 1
  if (slot0) __builtin_amdgcn_s_setprio(1);
#elif WFE_ASYM == 2 || WFE_ASYM == 4
  __builtin_amdgcn_s_setprio(3);
#elif WFE_ASYM == 3
  if (!slot0 && blockIdx.y * gridDim.x + blockIdx.x < 512) {
    __builtin_amdgcn_s_sleep(127);
    __builtin_amdgcn_s_sleep(127);
  }
#endif
#endif
  const Source src = make_source(a, b);
  const float* __restrict__ w = src.w;
  int raw_len = a.lengths[b];
  if (raw_len < 0) raw_len = 0;
  WFE_STAMP(0);
  Chain c;
  int len = raw_len;
  __shared__ COp cops[AUG ? kMaxCops : 1];
  ChainHead head;
  if (AUG) {
    head = a.heads[b];
    c = chain_from_head(a, head, b, src, raw_len);
    len = chain_len(c, c.n_ops);
  }
  if (len > n) len = n;                         // truncation=True (HF:300-307)
  const int p0 = t0 * kHop;
  const int i_first = p0 - kHalf;

  // Silent workgroup: every padded-signal position of these 64 frames maps (after
  // reflection at the end of the padded utterance) to the zero padding behind the
  // utterance.  Every sample is 0, so every power and mel value is exactly 0 and the
  // log-mel is exactly the floor log10(1e-10) = -10: no DFT, and no store either (below).  For
  // a 6 s utterance padded to 30 s that is 4/5 of the workgroups.
  {
    const int i_last = i_first + kBlockSamples - 1;
    const bool silent = i_first >= len && (i_last < n || 2 * (n - 1) - i_last >= len);
    if (silent) {
      if (POWER) {
        const int frames_here = (T - t0 < kBlockFrames) ? (T - t0) : kBlockFrames;
        float* __restrict__ po = a.out + (long long)b * kNfreq * T + t0;
        for (int idx = tid; idx < kNfreq * kBlockFrames; idx += 256) {
          const int k = idx >> 6, fb = idx & 63;
          if (fb < frames_here) po[(long long)k * T + fb] = 0.f;
        }
      } else {
        // Nothing is written here: the tile is the constant g(-10) = -1.5 before the
        // clamp, so kernel 2 (which knows the utterance maximum) writes its final value
        // once.  -inf marks the tile as silent; real minima are >= -10.
        if (tid < kTileStats)
          a.tile_stats[((long long)b * nblk + tx) * kTileStats + tid] = tid < 32 ? -INFINITY : -1.5f;     // (g(-10): its maximum)
      }
      return;
    }
  }

  // (A stagger heuristic lived here in early versions: the second workgroup to arrive on a CU slept
  // one non-MFMA phase so that the two resident workgroups would not run in lockstep.  It helped
  // the first ~40 launches after idle; at the sustained clock it costs 0.8-1 % on every workload,
  // plus a memset and an atomic per workgroup, so it is gone.)
#ifdef WFE_STAMPS
  if (a.stamps && lane == 0) {                  // which SIMD this wave ran on (tools/phase_breakdown.py)
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);       // HW_REG_XCC_ID[3:0]
    unsigned long long* st = a.stamps + ((long long)(blockIdx.y * ((a.n_frames + 63) / 64) + tx) * 4 + wave) * kStampRow;
    st[6] = ((unsigned long long)xcc << 32) | hw;
    st[7] = 0;
  }
#endif

  // k-steps 0 and 1 of the twiddle table (8 KB each): in flight while the
  // waveform is staged, then parked in the two LDS slots
  // (through a buffer resource: the per-k-step offset rides in an SGPR, the lane offset is one VGPR for
  // the whole kernel -- a global_load of base + constant beyond the 12-bit immediate costs a 64-bit VALU
  // add per k-step, and VALU instructions are what the matrix pipe waits for)
  const __amdgpu_buffer_rsrc_t tw_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dft_tab), 0, (int)(kDftTableFloats * 4), 0x00020000);
  const int tw_lane = tid * 16;
  auto tw_load = [&](int f4_index) {       // float4 number f4_index + tid of the table
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(tw_rsrc, tw_lane, f4_index * 16, 0));
  };
  f32x4* const slots = reinterpret_cast<f32x4*>(lds + kLdsWaveFloats);       // kSlots x 512 float4
  // k-step s parks A(s + kAhead) in LDS (slot (s + kAhead) mod kSlots) and requests A(s + kAhead + 2) from L2:
  // three register sets in rotation, two whole steps for the L2 round trip under load.  A(0) .. A(kAhead - 1)
  // are parked before the loop.
  constexpr int kAhead = kSlots == 2 ? 2 : 3;
  f32x4 gp[2 * kAhead];
  f32x4 G[3][2];
  auto request_twiddles = [&]() {
#pragma unroll
    for (int i = 0; i < 2 * kAhead; ++i) gp[i] = tw_load(256 * i);
#pragma unroll
    for (int k = kAhead; k < kAhead + 2; ++k) {          // A(kAhead), A(kAhead + 1): written out in steps 0 and 1
      G[k % 3][0] = tw_load(512 * k);
      G[k % 3][1] = tw_load(512 * k + 256);
    }
    G[(kAhead + 2) % 3][0] = G[kAhead % 3][0];           // (defined; overwritten in step 0)
    G[(kAhead + 2) % 3][1] = G[kAhead % 3][1];
  };
  if (!AUG) request_twiddles();            // (AUG: after the staging, which wants the 32 registers itself)

  // ---- stage the padded-signal span [160 t0, 160 t0 + 10480) into LDS
  {
    const bool inside = !AUG && i_first >= 0 && i_first + kBlockSamples <= len;
    const bool interior = inside && w && ((reinterpret_cast<uintptr_t>(w + i_first) & 15u) == 0);
    const bool interior16 = inside && src.pcm && src.ch == 1 &&
                            ((reinterpret_cast<uintptr_t>(src.pcm + i_first) & 15u) == 0);
    if (AUG) {
#ifdef WFE_STAMPS
      stage_augmented(a, head, b, c, cops, lds, lds + kLdsWaveFloats, p0, n, len, tid,
                                  (a.stamps && lane == 0) ? a.stamps + ((long long)(blockIdx.y * ((a.n_frames + 63) / 64) + tx) * 4 + wave) * kStampRow : nullptr);
#else
      stage_augmented(a, head, b, c, cops, lds, lds + kLdsWaveFloats, p0, n, len, tid);
#endif
    } else if (interior16) {
      // 16-bit PCM: 8 samples per 16-byte load, converted on the way into LDS
      using i32x4 = __attribute__((ext_vector_type(4))) int;
      const i32x4* __restrict__ src8 = reinterpret_cast<const i32x4*>(src.pcm + i_first);
      constexpr int kVec = kBlockSamples / 8;               // 1310
      constexpr int kIters = (kVec + 255) / 256;
      i32x4 buf[kIters];
#pragma unroll
      for (int it = 0; it < kIters; ++it) {
        const int v8 = tid + 256 * it;
        if (v8 < kVec) buf[it] = src8[v8];
      }
      constexpr float kScale = 1.0f / 32768.0f;
#pragma unroll
      for (int it = 0; it < kIters; ++it) {
        const int v8 = tid + 256 * it;
        if (v8 < kVec) {
          const int q = 8 * v8;                 // 8 | 160, so the eight share q / 160
          const int base = q + q / kHop;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int pair = buf[it][j];
            lds[base + 2 * j + 0] = (float)(short)(pair & 0xffff) * kScale;
            lds[base + 2 * j + 1] = (float)(pair >> 16) * kScale;
          }
        }
      }
    } else if (interior) {
      // every load in flight before the first LDS write: one HBM round trip per tile
      const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(w + i_first);
      constexpr int kIters = (kBlockSamples / 4 + 255) / 256;
      f32x4 buf[kIters];
#pragma unroll
      for (int it = 0; it < kIters; ++it) {
        const int v4 = tid + 256 * it;
#if WFE_ABLATE & 1
        buf[it] = f32x4{0.f, 0.f, 0.f, (float)v4};
#else
        if (v4 < kBlockSamples / 4) buf[it] = src[v4];
#endif
      }
#pragma unroll
      for (int it = 0; it < kIters; ++it) {
        const int v4 = tid + 256 * it;
        if (v4 < kBlockSamples / 4) {
          const int q = 4 * v4;                 // 4 | 160, so the four share q / 160
          const int base = q + q / kHop;
#ifdef WFE_ABLATE_STAGE_WRITES                  // diagnostic (wrong results): lane-consecutive words instead, to read the counters
          lds[v4 + 0] = buf[it][0], lds[v4 + 2624] = buf[it][1], lds[v4 + 5248] = buf[it][2], lds[v4 + 7872] = buf[it][3];
          continue;
#endif
          lds[base + 0] = buf[it][0];
          lds[base + 1] = buf[it][1];
          lds[base + 2] = buf[it][2];
          lds[base + 3] = buf[it][3];
        }
      }
    } else {
      // boundary tiles (reflection, end of the utterance): 41 positions per thread in 2 batches,
      // gathered branch-free (Source::gather) so that a batch is one HBM round trip
      constexpr int kBatch = 21;
      for (int q0 = tid; q0 < kBlockSamples; q0 += 256 * kBatch) {
        float vals[kBatch];
        int idx[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
          const int q = q0 + 256 * u;
          int i = p0 + q - kHalf;               // index into the length-n padded utterance
          i = (i < 0) ? -i : i;                 // torch.stft center=True, pad_mode="reflect"
          i = (i >= n) ? 2 * (n - 1) - i : i;
          idx[u] = (q < kBlockSamples) ? i : -1;
        }
        src.gather<kBatch>(idx, len, vals);     // 0 outside [0, len): right zero padding (padding_value 0.0)
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
          const int q = q0 + 256 * u;
          if (q < kBlockSamples) lds[pad_q(q)] = vals[u];
        }
      }
    }
  }
  if (AUG) request_twiddles();
#pragma unroll
  for (int i = 0; i < 2 * kAhead; ++i) slots[256 * i + tid] = gp[i];
  WFE_STAMP(1);
  __syncthreads();
  WFE_STAMP(2);

  if (AUG) __builtin_amdgcn_s_setprio(0);
#if defined(WFE_ASYM) && (WFE_ASYM == 2 || WFE_ASYM == 4)
  if (slot0) __builtin_amdgcn_s_setprio(WFE_ASYM == 2 ? 2 : 1);
  else __builtin_amdgcn_s_setprio(0);
#endif
  // ---- folded DFT as 6 GEMMs on v_mfma_f32_16x16x4_f32 (wfe_layout.h: levels B and C)
  const int jl = lane & 15;
  const int kk = lane >> 4;
  const int d_lo = 2 * (kk >> 1) + 16 * (kk & 1);      // odd class, steps 0..23
  const int d_hi = 2 * kk;                             // odd class, step 24
  const int d_ev = 4 * (kk >> 1) + 16 * (kk & 1);      // even class
  const int rowbase = 161 * (wave * 16 + jl);          // pad_q(160 * frame-in-block)

  // accumulators: odd class Co / So over the 7 tiles U0..3, V0..2; even class [cos | sin][ee | eo] over U0..3
  f32x4 acc_o[2][kOddTiles], acc_e[2][2][kEvenTiles];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
#pragma unroll
    for (int ti = 0; ti < kOddTiles; ++ti) acc_o[g][ti] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cl = 0; cl < 2; ++cl)
#pragma unroll
      for (int ti = 0; ti < kEvenTiles; ++ti) acc_e[g][cl][ti] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // The k-loop, fully unrolled (25 steps) so that every step-dependent index is a
  // compile-time constant and the LDS addresses are lane base + immediate.  The
  // twiddles of one k-step (8 KB) are shared by the 4 waves through a
  // double-buffered LDS slot, so the CU's L1/TA path carries each of them once per
  // workgroup instead of once per wave.  Everything a step consumes was requested
  // a whole step earlier, and the requests are dealt out between the MFMAs of the
  // step so that the matrix pipe never waits on their issue:
  //   step s:  MFMA 1,2    | global->reg A(s+4) (2 x 16 B/thread), three register sets in rotation
  //            MFMA 3,4    | reg->LDS A(s+2) into slot s&1 (fetched during step s-2; the slot was
  //                        |   read out in step s-1)
  //            MFMA 5..10  | LDS->reg the 6 fragment pieces of A(s+1)
  //            MFMA 11..14 | LDS->reg the 4 samples of step s+1
  //            MFMA 15..22 | barrier        (step 23 also requests the two pieces and two samples of even-step 25)
  //   (one request behind each MFMA, pinned there)
  // k-step s multiplies the odd class's step s (14 MFMAs) and the even class's even-step s (8 MFMAs, class ee
  // for s < 13, eo after); even-step 25 rides in the last k-step.
  // What keeps hipcc from serialising this (checked in the .s): the two fragment
  // register sets ping-pong (a single set plus a copy gets the copy, and its wait,
  // hoisted to a few MFMAs after the reads); scheduling barriers pin each batch
  // (otherwise requests sink to their first use); the padding twiddles are never
  // read (a dead destination register is reused at once and forces a wait).
  float xf[2], xb[2], xf2 = 0.f, xb2 = 0.f;            // [0] even class, [1] odd class; xf2 / xb2: even-step 25
  // Lane bases of the sample reads, forward (x[n]) and backward (x[400 - n]) for the three k-lane patterns: six LDS
  // pointers, made opaque so that each stays in ONE register for the whole k-loop and every read is base +
  // immediate (hipcc otherwise rebuilds the backward ones from their parts with two VALU instructions per read:
  // 96 per tile, next to the MFMA stream).
  using lds_cptr = const __attribute__((address_space(3))) float*;
  lds_cptr fw_lo = (lds_cptr)(lds + rowbase + d_lo), fw_hi = (lds_cptr)(lds + rowbase + d_hi), fw_ev = (lds_cptr)(lds + rowbase + d_ev);
  lds_cptr bw_lo = (lds_cptr)(lds + rowbase - d_lo), bw_hi = (lds_cptr)(lds + rowbase - d_hi), bw_ev = (lds_cptr)(lds + rowbase - d_ev);
  // The image's pad word per hop (pad_q) makes a read's x / 160 term lane-dependent in the 8 steps whose four
  // k-lanes straddle a multiple of 160 -- all of them BACKWARD reads of n = 65..96 (m = 400 - n around 320), and in
  // every one of them it is the k-lanes with (kk & 1) == 0 that lie beyond: two more bases, one word further for
  // those lanes, replace a compare / select / add per straddling read (48 vector instructions per tile).
  lds_cptr bw_lo_x = bw_lo + ((kk & 1) ? 0 : 1), bw_ev_x = bw_ev + ((kk & 1) ? 0 : 1);
  asm volatile("" : "+v"(fw_lo), "+v"(fw_hi), "+v"(fw_ev), "+v"(bw_lo), "+v"(bw_hi), "+v"(bw_ev), "+v"(bw_lo_x), "+v"(bw_ev_x));
  // sample n = dft_n_of(c, step, kk) of this lane's frame, or its mirror 400 - n
  auto sample_at = [&](int step, int c, int backward) {
    int nb, dl, c0;
    lds_cptr fw, bw;
    if (c) {
      const bool tail = step == kSteps - 1;
      nb = tail ? 192 : (32 * (step >> 2) + 4 * (step & 3));
      dl = tail ? d_hi : d_lo;
      fw = tail ? fw_hi : fw_lo;
      bw = tail ? bw_hi : bw_lo;
      c0 = 1;
    } else {
      const int u = step < 13 ? step : step - 13;
      nb = u < 12 ? 32 * (u >> 1) + 8 * (u & 1) : 192;
      dl = d_ev;
      fw = fw_ev;
      bw = bw_ev;
      c0 = step < 13 ? 4 : 2;
    }
    const int nn = nb + dl + c0;                       // 1..216 (beyond 200: the dummy k-lanes, zero twiddles)
    const int mm = kNfft - nn;
    // pad_q(160 j + x) = 161 j + x + x / 160; the x / 160 term is a compile-time
    // constant except in the few steps whose 4 k-lanes straddle 160 or 320
    if (backward) {
      // m >= 320 for all four k-lanes, for none, or (n = nb + c0 + {0, 2 | 4, 16, 18 | 20} around 80) for kk & 1 == 0 only
      const int m_hi = kNfft - nb - c0, m_lo = m_hi - (c ? 18 : 20);          // m of k-lane 0 and of k-lane 3
      if (m_hi >= 2 * kHop && m_lo < 2 * kHop) {
        const int m_mid = m_hi - (c ? 2 : 4), m_next = m_hi - 16;              // k-lanes 2 (kk = 2) and 1
        if (m_mid >= 2 * kHop && m_next < 2 * kHop && !(c && step == kSteps - 1))
          return (c ? bw_lo_x : bw_ev_x)[kNfft - nb - c0 + 1];
        return bw[kNfft - nb - c0 + 1 + (mm >= 2 * kHop ? 1 : 0)];
      }
      return bw[kNfft - nb - c0 + 1 + (m_hi >= 2 * kHop ? 1 : 0)];
    }
    return fw[nb + c0 + (nn >= kHop ? 1 : 0)];
  };
  auto sample_request = [&](int step, int c, int backward) {
    if (backward) xb[c] = sample_at(step, c, 1);
    else xf[c] = sample_at(step, c, 0);
  };
#ifdef WFE_STAMPS
  unsigned long long barrier_cycles = 0;
#endif
  struct StepFrag {
    Frag7 oc, os;                                      // odd class: cos, sin rows of the 7 tiles
    f32x4 ec, es;                                      // even class: cos, sin rows of U0..3
  };
  auto load_frag = [&](StepFrag& f, const f32x4* slot, int piece) {      // piece 0..5 of the step's 6
    if (piece == 0) f.oc.lo = slot[0];
    else if (piece == 1) f.oc.hi = *reinterpret_cast<const f32x3*>(slot + 64);
    else if (piece == 2) f.os.lo = slot[128];
    else if (piece == 3) f.os.hi = *reinterpret_cast<const f32x3*>(slot + 192);
    else if (piece == 4) f.ec = slot[256];
    else f.es = slot[320];
  };
  StepFrag Ra, Rb;
  f32x4 ec2 = f32x4{0.f, 0.f, 0.f, 0.f}, es2 = ec2;    // even-step 25 (quads 6, 7 of the last k-step)
#pragma unroll
  for (int pc = 0; pc < 6; ++pc) load_frag(Ra, slots + lane, pc);
#pragma unroll
  for (int i = 0; i < 4; ++i) sample_request(0, i >> 1, i & 1);
  __syncthreads();      // slot 0 is overwritten in the middle of step 0: every wave must have read it out
#pragma unroll
  for (int s = 0; s < kSteps; ++s) {
    StepFrag& Rc = (s & 1) ? Rb : Ra;
    StepFrag& Rn = (s & 1) ? Ra : Rb;
    const bool next = s + 1 < kSteps, stage = s + kAhead < kSteps, last = s == kSteps - 1;
    const f32x4* nslot = slots + ((s + 1) % kSlots) * 512 + lane;
    const int cl = s < 13 ? 0 : 1;                     // class of even-step s
    float e[2], o[2], e2 = 0.f, o2 = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#if WFE_ABLATE & 256
      e[c] = xf[c];
      o[c] = xb[c];
#else
      e[c] = xf[c] + xb[c];
      o[c] = xf[c] - xb[c];
#endif
    }
    if (last) e2 = xf2 + xb2, o2 = xf2 - xb2;
    constexpr int kMain = 2 * kOddTiles + 2 * kEvenTiles;          // 22
    auto mfma = [&](int m) {
      if (m < kOddTiles)
        acc_o[0][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(Rc.oc.at(m), e[1], acc_o[0][m], 0, 0, 0);
      else if (m < 2 * kOddTiles)
        acc_o[1][m - kOddTiles] = __builtin_amdgcn_mfma_f32_16x16x4f32(Rc.os.at(m - kOddTiles), o[1], acc_o[1][m - kOddTiles], 0, 0, 0);
      else if (m < 2 * kOddTiles + kEvenTiles)
        acc_e[0][cl][m - 14] = __builtin_amdgcn_mfma_f32_16x16x4f32(Rc.ec[m - 14], e[0], acc_e[0][cl][m - 14], 0, 0, 0);
      else if (m < kMain)
        acc_e[1][cl][m - 18] = __builtin_amdgcn_mfma_f32_16x16x4f32(Rc.es[m - 18], o[0], acc_e[1][cl][m - 18], 0, 0, 0);
      else if (m < kMain + kEvenTiles)                             // even-step 25 (class eo), last k-step only
        acc_e[0][1][m - 22] = __builtin_amdgcn_mfma_f32_16x16x4f32(ec2[m - 22], e2, acc_e[0][1][m - 22], 0, 0, 0);
      else
        acc_e[1][1][m - 26] = __builtin_amdgcn_mfma_f32_16x16x4f32(es2[m - 26], o2, acc_e[1][1][m - 26], 0, 0, 0);
    };
    f32x4(&Gw)[2] = G[(s + kAhead) % 3];        // holds A(s + kAhead), requested during step s-2
    f32x4(&Gr)[2] = G[(s + kAhead + 2) % 3];    // receives A(s + kAhead + 2)
    // the requests of a step, one behind each of its first MFMAs (a clump of
    // them between two MFMAs idles the matrix pipe for the clump's issue time)
    auto request = [&](int i) {
      if (i < 2) {                                          // global -> registers, A(s + kAhead + 2)
        if (s + kAhead + 2 < kSteps) Gr[i] = tw_load((s + kAhead + 2) * 512 + 256 * i);
      } else if (i < 4) {                                   // registers -> LDS, A(s + kAhead)
        if (stage) slots[((s + kAhead) % kSlots) * 512 + 256 * (i - 2) + tid] = Gw[i - 2];
      } else if (i < 10) {                                  // LDS -> registers, fragments of A(s+1)
        if (next) load_frag(Rn, nslot, i - 4);
      } else if (i < 14) {                                  // LDS -> registers, samples of step s+1
        if (next) sample_request(s + 1, (i - 10) >> 1, (i - 10) & 1);
      } else if (s + 2 == kSteps) {                         // for the last k-step: even-step 25 as well
        if (i == 14) ec2 = nslot[384];
        else if (i == 15) es2 = nslot[448];
        else if (i == 16) xf2 = sample_at(kEvenSteps - 1, 0, 0);
        else if (i == 17) xb2 = sample_at(kEvenSteps - 1, 0, 1);
      }
    };
    const int n_mfma = last ? kMain + 2 * kEvenTiles : kMain;
#pragma unroll
    for (int m = 0; m < kMain + 2 * kEvenTiles; ++m) {
      if (m < n_mfma) {
        mfma(m);
        __builtin_amdgcn_sched_barrier(0);
        if (m < 18) {
#if !(WFE_ABLATE & 128)
          request(m);
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
#ifdef WFE_STAMPS
    if (next && (kSlots == 2 || (s & 1))) {
      unsigned long long tb0, tb1;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb0)::"memory");
      __syncthreads();
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb1)::"memory");
      barrier_cycles += tb1 - tb0;
    }
#else
    // 2 slots: behind every step.  4 slots: what step s reads and overwrites was written / read in step s - 2,
    // and a barrier behind every odd step separates any two steps that far apart.
    if (next && (kSlots == 2 || (s & 1)) && !(WFE_ABLATE & 64)) __syncthreads();
#endif
  }

  WFE_STAMP(3);
#ifdef WFE_STAMPS
  if (a.stamps && lane == 0)
    a.stamps[((long long)(blockIdx.y * ((a.n_frames + 63) / 64) + tx) * 4 + wave) * kStampRow + 7] = barrier_cycles;
#endif
#ifdef WFE_EPI_PRIO           // experiment: the epilogue's vector instructions outrank the partner's MFMA stream
  __builtin_amdgcn_s_setprio(WFE_EPI_PRIO);
#endif
#if defined(WFE_ASYM) && (WFE_ASYM == 2 || WFE_ASYM == 4)
  __builtin_amdgcn_s_setprio(3);
#endif
  // Mel weight fragments of every live (bin tile, mel tile) block: all requested
  // here, ahead of the butterflies, so that the mel phase pays one L2 round trip
  // instead of one per block (the scheduling barrier pins the batch).
  // (buffer loads for the same reason as the twiddles: block offset in an SGPR, no VALU address math)
  const __amdgpu_buffer_rsrc_t mel_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.mel_tab), 0, (int)(kMelTableFloats * 4), 0x00020000);
  f32x4 wv[kPowerTiles][NMT];
  if (!POWER && BANK != 0) {
#pragma unroll
    for (int pt = 0; pt < kPowerTiles; ++pt)
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt)
        if (((BANK == 1 ? kMelMask80[pt] : kMelMask128[pt]) >> mt) & 1u)
          wv[pt][mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
              mel_rsrc, lane * 16, (pt * kMelTiles + mt) * 64 * 16, 0));
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- butterflies + power (wfe_layout.h).  Level C on the even class: position p of U tile ti gives
  // Ce[p], Se[p] (sums) and Ce[100-p], Se[100-p] (differences); level B pairs them with the odd class's rows
  // of U (bins p, 200 - p) and of V (bins 100 - p, 100 + p).  Every butterfly combines accumulators of one
  // lane and register.  Power tiles: U lo 0..3, U hi 4..7, V a 8..10, V b 11..13.
  // (Scalar f32 instructions: beside the partner wave's MFMA stream a packed v_pk_add / v_pk_fma costs the
  // matrix pipe more than the two scalar ones it replaces -- tools/micro/mfma_fillers.hip.)
  f32x4 P[kPowerTiles];
#pragma unroll
  for (int ti = 0; ti < kUTiles; ++ti) {
#if WFE_ABLATE & 2
    P[ti] = acc_e[0][0][ti] + acc_e[1][1][ti] + acc_o[0][ti];       // (all GEMMs stay live)
    P[4 + ti] = acc_e[0][1][ti] + acc_e[1][0][ti] + acc_o[1][ti];
    if (ti < kVTiles) P[8 + ti] = acc_o[0][4 + ti], P[11 + ti] = acc_o[1][4 + ti];
#else
    const f32x4 ce = acc_e[0][0][ti] + acc_e[0][1][ti], se = acc_e[1][0][ti] + acc_e[1][1][ti];
    {
      const f32x4 co = acc_o[0][ti], so = acc_o[1][ti];
      const f32x4 rp = ce + co, ip = se + so, rm = ce - co, im = se - so;
      P[ti] = __builtin_elementwise_fma(rp, rp, ip * ip);
      P[4 + ti] = __builtin_elementwise_fma(rm, rm, im * im);
    }
    if (ti < kVTiles) {
      const f32x4 cm = acc_e[0][0][ti] - acc_e[0][1][ti], sm = acc_e[1][1][ti] - acc_e[1][0][ti];
      const f32x4 co = acc_o[0][4 + ti], so = acc_o[1][4 + ti];
      const f32x4 rp = cm + co, ip = sm + so, rm = cm - co, im = sm - so;
      P[8 + ti] = __builtin_elementwise_fma(rp, rp, ip * ip);
      P[11 + ti] = __builtin_elementwise_fma(rm, rm, im * im);
    }
#endif
  }
  const int frame = t0 + wave * 16 + jl;

  if (POWER) {
    float* __restrict__ po = a.out + (long long)b * kNfreq * T;
#pragma unroll
    for (int pt = 0; pt < kPowerTiles; ++pt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int bin = power_bin_of(pt, 4 * kk + r);
        if (frame < T && bin >= 0) po[(long long)bin * T + frame] = P[pt][r];
      }
    return;
  }

  // ---- mel projection on the same MFMA; all-zero (power tile, mel tile) blocks skipped
  f32x4 m[NMT];
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt) m[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (BANK != 0) {
#pragma unroll
    for (int pt = 0; pt < kPowerTiles; ++pt)
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt)
        if (((BANK == 1 ? kMelMask80[pt] : kMelMask128[pt]) >> mt) & 1u) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#if WFE_ABLATE & 4
            m[mt][r] += wv[pt][mt][r] + P[pt][r];
#else
            m[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[pt][mt][r], P[pt][r], m[mt], 0, 0, 0);
#endif
        }
  } else {
#pragma unroll
    for (int pt = 0; pt < kPowerTiles; ++pt) {
      const unsigned mask = a.mel_mask[pt];
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) {
        if ((mask >> mt) & 1u) {
          // uniform base + 32-bit lane offset: keeps the address in SGPRs (one VGPR
          // pair per block would spill in this run-time-mask build)
          const f32x4 w4 = reinterpret_cast<const f32x4*>(a.mel_tab)[(pt * kMelTiles + mt) * 64 + lane];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            m[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[r], P[pt][r], m[mt], 0, 0, 0);
        }
      }
    }
  }
  WFE_STAMP(4);

  // ---- log10, utterance max, and the block's (n_mel x 64 frames) tile through LDS
  // so that every mel row leaves as one 256-byte run of float4 stores.
  // clamp(min=1e-10).log10() (HF:159): log10 of the float32 floor rounds to exactly
  // -10 and the floor is by far the most common value (silence, padding), so it is
  // produced exactly; above it v_log_f32 * log10(2) is within 1.2e-6 of log10.
  float* const tile = lds + kLdsWaveFloats;            // [16 * NMT][kTileStride]
  // The value stored is already (v + 4) / 4: that map is monotone, so the max-8 clamp
  // of kernel 2 commutes with it exactly, max(g(v), g(max-8)) == g(max(v, max-8)), and
  // kernel 2 only has to touch the workgroups whose minimum lies below max-8.
  // Statistics on the POWER: log10 with its floor is monotone, so max / min of the log values are the
  // log values of the max / min powers, exactly -- three-input max / min over the accumulators, then
  // the log of two wave-uniform numbers, instead of two instructions per value.
  const bool frame_ok = frame < T;
  // g(log10(max(pw, 1e-10))) in three instructions per value (round 3; five before): v_log_f32, one fma with
  // log10(2) / 4 folded into the multiplier, and a max against g(-10) = -1.5 -- the floor of clamp(min=1e-10):
  // powers at or below it (silence, padding: by far the most common value) give exactly -1.5, and the map
  // stays monotone, which is all the statistics and kernel 2 rely on.  Within 1.5e-7 of the five-instruction
  // form (one rounding fewer).
  auto g_of = [](float pw) {
#if WFE_ABLATE & 8
    return pw;
#else
    return __builtin_fmaxf(__builtin_fmaf(__builtin_amdgcn_logf(pw), 0.25f * 0.30102999566398120f, 1.0f), -1.5f);
#endif
  };
  // Per wave: ONE maximum, and a minimum per MEL TILE (16 rows) -- round 3: kernel 2 then looks at the 16-frame x
  // 16-row blocks that hold a value below the floor (a few percent of them) instead of at whole half tiles.
  float pmax = -INFINITY, lmin[8];
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) lmin[mt] = INFINITY;
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt) {
    float hi = __builtin_fmaxf(__builtin_fmaxf(m[mt][0], m[mt][1]), m[mt][2]);      // v_max3_f32
    float lo = __builtin_fminf(__builtin_fminf(m[mt][0], m[mt][1]), m[mt][2]);      // v_min3_f32
    float hi3 = m[mt][3], lo3 = m[mt][3];
    if (BANK == 0) {                                   // another feature_size: rows at and beyond n_mel do not count
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool live = 16 * mt + 4 * kk + r < a.n_mel;
        if (r == 0) hi = live ? m[mt][0] : -INFINITY, lo = live ? m[mt][0] : INFINITY;
        else if (r < 3) hi = live ? __builtin_fmaxf(hi, m[mt][r]) : hi, lo = live ? __builtin_fminf(lo, m[mt][r]) : lo;
        else hi3 = live ? m[mt][3] : -INFINITY, lo3 = live ? m[mt][3] : INFINITY;
      }
    }
    pmax = __builtin_fmaxf(__builtin_fmaxf(pmax, hi), hi3);
    lmin[mt] = frame_ok ? __builtin_fminf(lo, lo3) : INFINITY;
  }
#if !(WFE_ABLATE & 16)
  pmax = wave_reduce<true>(frame_ok ? pmax : -INFINITY);
#endif
  // The 8 minima over the 64 lanes by a transposing butterfly: each exchange halves the values a lane carries
  // (a lane keeps the half its own bit selects and takes the partner's minimum of the same half), so the whole
  // reduction is 7 + 1 + 2 min instructions and 14 selects instead of 8 separate wave reductions.  Afterwards
  // lane l holds the wave's minimum of mel tile 4 b0 + 2 b1 + b2 (b = bits of l).
  float red;
  {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    float u[4], w2[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float keep = b0 ? lmin[i + 4] : lmin[i], send = b0 ? lmin[i] : lmin[i + 4];
      u[i] = __builtin_fminf(keep, dpp_mov<0xB1>(send));        // quad_perm [1,0,3,2]: lane ^ 1
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float keep = b1 ? u[i + 2] : u[i], send = b1 ? u[i] : u[i + 2];
      w2[i] = __builtin_fminf(keep, dpp_mov<0x4E>(send));       // quad_perm [2,3,0,1]: lane ^ 2
    }
    {
      const float keep = b2 ? w2[1] : w2[0], send = b2 ? w2[0] : w2[1];
      red = __builtin_fminf(keep, dpp_mov<0x124>(send));        // row_ror:4: the sender's bit 2 is the opposite one
    }
    red = __builtin_fminf(red, dpp_mov<0x128>(red));            // row_ror:8: the whole row of 16
    // across the four rows of 16: the swaps leave (own row, other row) in the two results
    const unsigned ur = __float_as_uint(red);
    const auto s16 = __builtin_amdgcn_permlane16_swap(ur, ur, false, false);
    red = __builtin_fminf(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
    const unsigned us = __float_as_uint(red);
    const auto s32 = __builtin_amdgcn_permlane32_swap(us, us, false, false);
    red = __builtin_fminf(__uint_as_float(s32[0]), __uint_as_float(s32[1]));
  }
  // statistics in the STORED scale (g): g of the extreme powers = the extremes of the stored values, exactly
  // (a wave without a valid frame leaves -inf / +inf, as before)
  const float wmax = pmax == -INFINITY ? -INFINITY : g_of(pmax);
  const float wmin = red == INFINITY ? INFINITY : g_of(red);
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int mel = 16 * mt + 4 * kk + r;
      tile[mel * kTileStride + wave * 16 + jl] = g_of(m[mt][r]);
    }
  {
    // [wave 4][mel tile 8] minima, then [wave 4] maxima: plain stores, kernel 2 reduces the maxima (no atomics, no zeroing)
    float* st = a.tile_stats + ((long long)b * nblk + tx) * kTileStats;
    if (lane < 8) st[8 * wave + 4 * (lane & 1) + (lane & 2) + ((lane >> 2) & 1)] = wmin;
    if (lane == 0) st[32 + wave] = wmax;
  }
  __syncthreads();
  float* __restrict__ lo = a.out + (long long)b * a.n_mel * T + t0;
  const int frames_here = (T - t0 < kBlockFrames) ? (T - t0) : kBlockFrames;
  if (a.out_kind != 0) {
    // float16 / bfloat16 output (the encoder's input dtype under fp16 / bf16 training): the same
    // rows, rounded once here -- rounding is monotone, so kernel 2's clamp commutes with it too
    const bool bf = a.out_kind == 2;
    unsigned short* __restrict__ lo16 = reinterpret_cast<unsigned short*>(a.out) + (long long)b * a.n_mel * T + t0;
    if ((T & 3) == 0 && (reinterpret_cast<uintptr_t>(lo16) & 7u) == 0) {
      constexpr int kIt = 16 * NMT / 16;
      f32x4 row[kIt];
#pragma unroll
      for (int it = 0; it < kIt; ++it)
        row[it] = *reinterpret_cast<const f32x4*>(tile + (it * 16 + (tid >> 4)) * kTileStride + 4 * (tid & 15));
#pragma unroll
      for (int it = 0; it < kIt; ++it) {
        const int mel = it * 16 + (tid >> 4), c4 = tid & 15;
        if (mel < a.n_mel && 4 * c4 < frames_here) {
          uint2 pk;
          pk.x = to16(row[it][0], bf) | (to16(row[it][1], bf) << 16);
          pk.y = to16(row[it][2], bf) | (to16(row[it][3], bf) << 16);
          *reinterpret_cast<uint2*>(lo16 + (long long)mel * T + 4 * c4) = pk;
        }
      }
    } else {
      for (int idx = tid; idx < a.n_mel * kBlockFrames; idx += 256) {
        const int mel = idx >> 6, fb = idx & 63;
        if (fb < frames_here) lo16[(long long)mel * T + fb] = (unsigned short)to16(tile[mel * kTileStride + fb], bf);
      }
    }
  } else if ((T & 3) == 0 && (reinterpret_cast<uintptr_t>(lo) & 15u) == 0) {
    constexpr int kIt = 16 * NMT / 16;                  // (16 NMT rows x 16 float4) / 256 threads
    f32x4 row[kIt];
#pragma unroll
    for (int it = 0; it < kIt; ++it)
      row[it] = *reinterpret_cast<const f32x4*>(tile + (it * 16 + (tid >> 4)) * kTileStride + 4 * (tid & 15));
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
      const int mel = it * 16 + (tid >> 4), c4 = tid & 15;
      if (mel < a.n_mel && 4 * c4 < frames_here)        // T % 4 == 0: a float4 is all-valid or all-out
#if WFE_ABLATE & 32
        if (row[it][0] == 12345.678f)
#endif
        *reinterpret_cast<f32x4*>(lo + (long long)mel * T + 4 * c4) = row[it];
    }
  } else {
    for (int idx = tid; idx < a.n_mel * kBlockFrames; idx += 256) {
      const int mel = idx >> 6, fb = idx & 63;
      if (fb < frames_here) lo[(long long)mel * T + fb] = tile[mel * kTileStride + fb];
    }
  }
  WFE_STAMP(5);
}

// ------------------------------------------------------------- kernel 2
// log_spec = maximum(log_spec, max - 8.0); (log_spec + 4.0) / 4.0   (HF:160-165).
// Kernel 1 already stored g(v) = (v + 4) / 4 and, per 64-frame tile, the maximum of each of its waves
// (16 frames) and the minimum of each wave's 16-row mel tiles, all in the stored scale; g is monotone, so the
// result is max(stored, g(max) - 2) bit for bit, and a (16 frames x 16 mel rows) block needs touching only when
// its minimum is below that floor.  kClampParts workgroups per utterance, each walking every kClampParts-th tile
// (round 3: one workgroup per tile was 12 032 workgroups that mostly exit -- 0.016 ms of dispatch -- and a
// half tile of ALL mel rows was read for one quiet row: 383 MB per 256 x 30 s at 128 mel).
constexpr int kClampParts = 8;
__global__ __launch_bounds__(256) void clamp_kernel(float* __restrict__ out, int out_kind,
                                                    const float* __restrict__ tile_stats, int n_mel, int T) {
  __shared__ float sh[4];
  const int b = blockIdx.y;
  const int nblk = (T + kBlockFrames - 1) / kBlockFrames;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* __restrict__ stats = tile_stats + (long long)b * nblk * kTileStats;
  // log_spec.max() of the utterance (HF:161): the maximum of the per-wave maxima kernel 1 left
  float gmax = -INFINITY;
  for (int i = tid; i < nblk * 4; i += 256) gmax = fmaxf(gmax, stats[(i >> 2) * kTileStats + 32 + (i & 3)]);
  gmax = wave_reduce<true>(gmax);
  if (lane == 0) sh[wv] = gmax;
  __syncthreads();
  gmax = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  // the statistics are in the stored scale g(v) = (v + 4) / 4: g(max - 8) = g(max) - 2
  const float floor_g = gmax - 2.0f;
  const float fill = fmaxf(-1.5f, floor_g);           // a silent tile: every value is max(g(-10), g(max-8))
  const bool bf = out_kind == 2;
  const unsigned hf = to16(floor_g, bf), hfill = to16(fill, bf);
  const float ff = from16(hf, bf);
  const int n_mt = (n_mel + 15) >> 4;
  // This workgroup's tiles are tx = blockIdx.x + kClampParts j.  Eight of them at a time: thread t looks at ONE
  // block (tile j = t >> 5, frame group w = (t >> 3) & 3, mel tile mt = t & 7) -- all the minima in one round trip
  // -- and each wave then walks the set bits of its own ballot (its 64 blocks = two tiles); most ballots are empty.
  const int row = lane >> 2, c4 = lane & 3;            // a block is 16 rows x 16 frames: lane -> row, frames 4 c4 .. + 3
  for (int j0 = 0; blockIdx.x + kClampParts * j0 < nblk; j0 += 8) {
    const int tj = blockIdx.x + kClampParts * (j0 + (tid >> 5));
    const int mt_t = tid & 7;
    float lowest_t = INFINITY;
    if (tj < nblk && mt_t < n_mt) lowest_t = stats[tj * kTileStats + (tid & 31)];
    // Tiles kernel 1 left unwritten (silence behind a short utterance: most of a ragged batch) are filled whole,
    // every wave 4 rows x 64 frames at a time (256-byte row segments); tiles at a ragged end of T or of an
    // unaligned output take the block path below.
    if (out_kind == 0 && (T & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0) {
      for (int j = 0; j < 8; ++j) {
        const int tx = blockIdx.x + kClampParts * (j0 + j);
        if (tx >= nblk || (tx + 1) * kBlockFrames > T) break;
        if (stats[tx * kTileStats] != -INFINITY) continue;          // (uniform)
        float* __restrict__ p = out + ((long long)b * n_mel) * T + tx * kBlockFrames + 4 * (tid & 15);
        for (int mel = tid >> 4; mel < n_mel; mel += 16)
          *reinterpret_cast<f32x4*>(p + (long long)mel * T) = f32x4{fill, fill, fill, fill};
        if ((tid >> 5) == j) lowest_t = INFINITY;                    // done: not a block of the ballot
      }
    }
    unsigned long long todo = __builtin_amdgcn_ballot_w64(lowest_t < floor_g);
    while (todo) {
      const int bit = __builtin_ctzll(todo);
      todo &= todo - 1;
      const int t = 64 * wv + bit;                      // (uniform) the thread whose block this is
      const int tx = blockIdx.x + kClampParts * (j0 + (t >> 5)), w = (t >> 3) & 3, mt = t & 7;
      const float lowest = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lowest_t), bit));
      const bool silent = lowest == -INFINITY;          // left unwritten by kernel 1
      const int mel = 16 * mt + row, f0 = tx * kBlockFrames + 16 * w + 4 * c4;
      if (mel >= n_mel || f0 >= T) continue;
      const long long at = ((long long)b * n_mel + mel) * T + f0;
      if (out_kind == 0) {
        float* __restrict__ p = out + at;
        if (f0 + 4 <= T && (reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
          f32x4* q4 = reinterpret_cast<f32x4*>(p);
          if (silent) {
            *q4 = f32x4{fill, fill, fill, fill};
          } else {
            // written back only where the clamp changes something
            const f32x4 v = *q4;
            if (fminf(fminf(v[0], v[1]), fminf(v[2], v[3])) < floor_g)
              *q4 = f32x4{fmaxf(v[0], floor_g), fmaxf(v[1], floor_g), fmaxf(v[2], floor_g), fmaxf(v[3], floor_g)};
          }
        } else {
          for (int e = 0; e < 4 && f0 + e < T; ++e) p[e] = silent ? fill : fmaxf(p[e], floor_g);
        }
      } else {
        // 16-bit output: max(h, round(floor)) on the stored halves == round(max(g(v), floor))
        unsigned short* __restrict__ p = reinterpret_cast<unsigned short*>(out) + at;
        if (f0 + 4 <= T && (reinterpret_cast<uintptr_t>(p) & 7u) == 0) {
          uint2* q2 = reinterpret_cast<uint2*>(p);
          uint2 v;
          v.x = v.y = hfill | (hfill << 16);
          if (silent) {
            *q2 = v;
          } else {
            const uint2 old = *q2;
            unsigned e[4] = {old.x & 0xffffu, old.x >> 16, old.y & 0xffffu, old.y >> 16};
#pragma unroll
            for (int k = 0; k < 4; ++k) e[k] = (from16(e[k], bf) > ff) ? e[k] : hf;
            v.x = e[0] | (e[1] << 16);
            v.y = e[2] | (e[3] << 16);
            if (v.x != old.x || v.y != old.y) *q2 = v;      // (only where the clamp changes something)
          }
        } else {
          for (int e = 0; e < 4 && f0 + e < T; ++e) {
            const unsigned old = p[e];
            p[e] = (unsigned short)(silent ? hfill : ((from16(old, bf) > ff) ? old : hf));
          }
        }
      }
    }
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

// Mean square of every noise clip: kNoiseParts workgroups per clip over a fixed
// partition write float64 partial sums of the float32 squares; noise_meansq() combines
// them in index order (scheduling cannot change the result).
__global__ __launch_bounds__(256) void noise_meansq_kernel(const float* __restrict__ noise,
                                                           const long long* __restrict__ off,
                                                           double* __restrict__ partial) {
  __shared__ double sh[4];
  const int j = blockIdx.y, part = blockIdx.x;
  const long long o0 = off[j], ln = off[j + 1] - o0;
  const long long per = (ln + kNoiseParts - 1) / kNoiseParts;
  const long long lo = part * per, hi = (lo + per < ln) ? lo + per : ln;
  double s = 0.0;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const float x = noise[o0 + i];
    s += (double)mul_rn(x, x);                    // float32 squares, as sample ** 2
  }
  s = block_sum<256>(s, sh);
  if (threadIdx.x == 0) partial[(long long)j * kNoiseParts + part] = s;
}
__device__ inline double noise_meansq(const FrontArgs& a, int j) {
  const long long ln = a.noise_off[j + 1] - a.noise_off[j];
  double s = 0.0;
  for (int p = 0; p < kNoiseParts; ++p) s += a.noise_meansq[(long long)j * kNoiseParts + p];
  return ln > 0 ? s / (double)ln : 0.0;
}

// rms_db (utils/reader.py:350-353) under NumPy-2 float32 scalar semantics
__device__ inline float rms_db_f32(double meansq) { return mul_rn(10.0f, log10f((float)meansq)); }

// (Round 3 also built the sums INTO kernel 1: the workgroups of utterance b summed utterance b + 16, a part each, the
// last of them resolved and published that chain with device-scope stores, and every workgroup waited -- bounded,
// with a sum-it-yourself way out -- for its own utterance's ready word.  Identical bits to this pass on a mixed batch
// and in the give-up path; and no faster: kernel 1 grew by 0.093 ms per 256 x 30 s, what this pass costs (0.100),
// because every workgroup's prologue gained three serial memory round trips -- its part's loads, the ticket, the
// ready word and then the head, which must come as device-scope loads -- that its partner workgroup only half
// hides.  A pass of its own at 4.5 TB/s is the cheaper way to read the waveform twice.  Removed.)
// Statistics of the signal ENTERING the first NOISE / AFFINE op of every utterance
// (the only one in the reference's configurations), kResolveParts workgroups per
// utterance so that the pass runs at HBM rate: float64 partial sums of x and x*x
// over a fixed partition, combined in a fixed order by resolve_utterance (the result
// does not depend on scheduling).
__device__ inline int first_resolvable(const Chain& c) {
  for (int k = 0; k < c.n_ops; ++k)
    if (c.ops[k].kind == WFE_AUG_NOISE || c.ops[k].kind == WFE_AUG_AFFINE) return k;
  return -1;
}

__device__ void resolve_utterance(const FrontArgs& a, const double* partials, int b, double* sh);

__global__ __launch_bounds__(256) void resolve_partial_kernel(const FrontArgs a, double* partials,
                                                              unsigned* __restrict__ tickets) {
  __shared__ double sh[4];
  const int b = blockIdx.y, part = blockIdx.x;
  const Source src = make_source(a, b);
  int raw_len = a.lengths[b];
  if (raw_len < 0) raw_len = 0;
  const Chain c = make_chain(a, b, src, raw_len);
  const int k0 = first_resolvable(c);
  __shared__ COp cops[kMaxCops];
  __shared__ int is_last;
  Chain cc = c;
  if (k0 >= 0) {
    const bool compiled = compile_chain(c, cops, k0, (int)threadIdx.x);
    __syncthreads();
    if (compiled) cc.cops = cops;
  }
  int L = k0 >= 0 ? chain_len(c, k0) : 0;
  if (k0 >= 0 && c.ops[k0].kind == WFE_AUG_AFFINE && L > a.n_samples) L = a.n_samples;
  const int per = (L + kResolveParts - 1) / kResolveParts;
  const int lo = part * per, hi = (lo + per < L) ? lo + per : L;
  double s1 = 0.0, s2 = 0.0;
  if (k0 == 0 && src.w) {
    // the signal entering the chain's first op is the source itself (the shipped order starts with NOISE):
    // 16-byte loads, four rows in flight per thread.  The loads ask for dword alignment only, so the partition
    // and the order of the sums -- and with them the last bit of the gains -- do not depend on where the
    // utterance starts in the caller's buffer (tests: unaligned offsets).
    const f32x4u* __restrict__ w4 = reinterpret_cast<const f32x4u*>(src.w);
    const int lo4 = (lo + 3) >> 2, hi4 = hi >> 2;                 // whole float4s inside [lo, hi)
    auto take = [&](float x) {
      s1 += (double)x;
      s2 += (double)mul_rn(x, x);                                 // float32 squares, as sample ** 2
    };
    if (lo4 >= hi4) {
      for (int i = lo + (int)threadIdx.x; i < hi; i += 256) take(src.w[i]);
    } else {
      if ((int)threadIdx.x < 4 * lo4 - lo) take(src.w[lo + threadIdx.x]);            // ragged head (< 4 samples)
      if ((int)threadIdx.x < hi - 4 * hi4) take(src.w[4 * hi4 + threadIdx.x]);       // ragged tail
      // Rows of 256 float4 in batches of kRows, every load of a batch in flight before the first sum: a part of a
      // 30 s utterance is 29.3 rows per thread, i.e. two batches -- two memory round trips per workgroup where
      // four-row batches plus a row-by-row tail made eleven (0.108 -> 0.0xx ms per 256 x 30 s).  A row beyond the
      // part loads the part's last float4 and contributes nothing; the order of the sums is the row order, as before.
      constexpr int kRows = 16;
      for (int i0 = lo4 + (int)threadIdx.x; i0 - (int)threadIdx.x < hi4; i0 += 256 * kRows) {
        f32x4 v[kRows];
#pragma unroll
        for (int r = 0; r < kRows; ++r) v[r] = w4[min(i0 + 256 * r, hi4 - 1)];
#pragma unroll
        for (int r = 0; r < kRows; ++r)
          if (i0 + 256 * r < hi4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) take(v[r][e]);
          }
      }
    }
  } else if (k0 < 0) {
    // nothing to sum: the chain has no data-dependent op
  } else if (cc.cops && (c.speed_at < 0 || c.speed_at >= k0)) {
    constexpr int U = 8;
    const int S = seg_shift_sum(cc, 0, k0);
    for (int i0 = lo + (int)threadIdx.x; i0 < hi; i0 += 256 * U) {
      int cur[U];
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = (i0 + 256 * u < hi) ? i0 + 256 * u + S : kDeadIndex;
      src.gather<U>(cur, c.len0, v);
      seg_apply<U>(cc, 0, k0, c.len0, cur, v);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (i0 + 256 * u < hi) {
          s1 += (double)v[u];
          s2 += (double)mul_rn(v[u], v[u]);         // float32 squares, as sample ** 2
        }
    }
  } else {
    for (int i = lo + (int)threadIdx.x; i < hi; i += 256) {
      const float x = chain_eval(c, k0, i);
      s1 += (double)x;
      s2 += (double)mul_rn(x, x);
    }
  }
  s1 = block_sum<256>(s1, sh);
  s2 = block_sum<256>(s2, sh);
  if (threadIdx.x == 0) {
    // The last of the utterance's kResolveParts workgroups to get here resolves the chain (what a second
    // launch did in round 1: 0.020 ms + a launch gap per call).  The partial sums and the ticket are
    // device-scope atomics (performed at the memory side, past the per-XCD L2s, no cache maintenance: a
    // __threadfence() pair per workgroup -- an L2 write-back and invalidate each -- cost 0.58 ms per call);
    // the sums have left the wave before the ticket is taken.
    __hip_atomic_store(&partials[((long long)b * kResolveParts + part) * 2 + 0], s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&partials[((long long)b * kResolveParts + part) * 2 + 1], s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(&tickets[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (t == (unsigned)kResolveParts - 1u);
    if (is_last) __hip_atomic_store(&tickets[b], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
  }
  __syncthreads();
  if (!is_last) return;
  resolve_utterance(a, partials, b, sh);
}

// One workgroup per utterance (the last of resolve_partial_kernel's to arrive): walks its chain in order
// and fills the data-dependent scalars of NOISE (gains, utils/reader.py:325-331) and AFFINE
// (mean / std, HF:172-191) ops.  An op's statistics are taken over the signal
// ENTERING it, so earlier ops -- including earlier resolved ones -- are applied.
// The first such op takes its sums from resolve_partial_kernel; any later one is
// summed here.
__device__ void resolve_utterance(const FrontArgs& a, const double* partials, int b, double* sh) {
  const Source src = make_source(a, b);
  int raw_len = a.lengths[b];
  if (raw_len < 0) raw_len = 0;
  const Chain c = make_chain(a, b, src, raw_len);
  Resolved* res = a.resolved + a.op_begin[b];
  bool first = true;
  for (int k = 0; k < c.n_ops; ++k) {
    const int kind = c.ops[k].kind;
    if (kind != WFE_AUG_NOISE && kind != WFE_AUG_AFFINE) continue;
    int L = chain_len(c, k);
    if (kind == WFE_AUG_AFFINE && L > a.n_samples) L = a.n_samples;   // stats over the kept prefix
    double s1 = 0.0, s2 = 0.0;
    if (first) {
      for (int p = 0; p < kResolveParts; ++p) {
        s1 += __hip_atomic_load(&partials[((long long)b * kResolveParts + p) * 2 + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s2 += __hip_atomic_load(&partials[((long long)b * kResolveParts + p) * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      for (int i = threadIdx.x; i < L; i += 256) {
        const float x = chain_eval(c, k, i);
        s1 += (double)x;
        s2 += (double)mul_rn(x, x);
      }
      s1 = block_sum<256>(s1, sh);
      s2 = block_sum<256>(s2, sh);
    }
    first = false;
    if (threadIdx.x == 0) {
      Resolved r{0.f, 1.f};
      if (kind == WFE_AUG_NOISE) {
        const double ms = L > 0 ? s2 / (double)L : 0.0;
        const float gain = fminf(300.0f, sub_rn(-20.0f, rms_db_f32(ms)));
        const float g1 = powf(10.0f, div_rn(gain, 20.0f));
        const float s_db = rms_db_f32(ms * (double)g1 * (double)g1);
        const float n_db = rms_db_f32(noise_meansq(a, c.ops[k].i0));
        const float ng = fminf(sub_rn(sub_rn(s_db, n_db), (float)c.ops[k].i2), 300.0f);
        r.a = g1;
        r.b = powf(10.0f, div_rn(ng, 20.0f));
      } else if (L > 0) {
        const double mean = s1 / (double)L;
        double var = s2 / (double)L - mean * mean;
        if (var < 0.0) var = 0.0;
        r.a = (float)mean;
        r.b = sqrtf(add_rn((float)var, 1e-7f));
      }
      res[k] = r;
    }
    __threadfence_block();                    // (read back below by this workgroup only: later ops, make_cop)
    __syncthreads();
  }
  // the chain, compiled with its resolved scalars, for kernel 1 (copy_cops)
  const bool fits = c.n_ops <= kMaxCops;
  if (fits && (int)threadIdx.x < c.n_ops)
    reinterpret_cast<COp*>(a.cops)[a.op_begin[b] + threadIdx.x] = make_cop(c, (int)threadIdx.x);
  __threadfence_block();                      // (thread 0 reads the compiled ops back below)
  __syncthreads();
  if (threadIdx.x == 0) {
    ChainHead h;
    h.n_ops = c.n_ops;
    h.speed_at = c.speed_at;
    h.len1 = c.len1;
    h.compiled = fits ? 1 : 0;
    h.step = c.step;
    h.shipped = 0;
    h.noise_k = -1;
    h.shift = 0;
    h.gain = 1.0f;
    h.noise_len = h.noise_start = 0;
    h.noise_o0 = 0;
    h.ra = 0.f;
    h.rb = 1.f;
    h.noise_inv = 0.0;
    if (fits) {
      Chain cc = c;
      cc.cops = reinterpret_cast<const COp*>(a.cops) + a.op_begin[b];
      const Shipped f = match_shipped(cc);
      if (f.ok) {
        h.shipped = 1;
        h.noise_k = f.noise_k;
        h.shift = f.shift;
        h.gain = f.gain;
        if (f.noise_k >= 0) {
          const COp nop = get_cop(cc, f.noise_k);
          h.noise_len = nop.ln;
          h.noise_start = nop.i1;
          h.noise_o0 = nop.o0;
          h.ra = nop.ra;
          h.rb = nop.rb;
          h.noise_inv = nop.inv;
        }
      }
    }
    a.heads[b] = h;
  }
}

// The augmented waveform itself (stage-level parity tests).  It runs kernel 1's own
// staging code, stage_augmented, over the same 10480-position spans and copies the
// LDS image out, so the bit-exact waveform tests exercise the production evaluator.
__global__ __launch_bounds__(256) void waveform_kernel(const FrontArgs a, float* __restrict__ out,
                                                       int* __restrict__ out_len) {
  constexpr int kSpan = kBlockFrames * kHop;            // output samples per workgroup (10240)
  __shared__ __attribute__((aligned(16))) float lds[kLdsWaveFloats + kStageAuxFloats];
  const int b = blockIdx.y;
  const Source src = make_source(a, b);
  int raw_len = a.lengths[b];
  if (raw_len < 0) raw_len = 0;
  __shared__ COp cops[kMaxCops];
  const ChainHead head = a.heads[b];
  Chain c = chain_from_head(a, head, b, src, raw_len);
  const int L = chain_len(c, c.n_ops);
  const int keep = L < a.n_samples ? L : a.n_samples;
  if (blockIdx.x == 0 && threadIdx.x == 0 && out_len) out_len[b] = L;
  const int p0 = blockIdx.x * kSpan;                    // staging position q holds sample p0 + q - 200
  stage_augmented(a, head, b, c, cops, lds, lds + kLdsWaveFloats, p0, a.n_samples, keep, threadIdx.x);
  __syncthreads();
  float* o = out + (long long)b * a.n_samples;
  for (int q = kHalf + threadIdx.x; q < kHalf + kSpan; q += 256) {
    const int i = p0 + q - kHalf;
    if (i < a.n_samples) o[i] = lds[pad_q(q)];
  }
}

// ------------------------------------------------------------- resampling
// Polyphase windowed-sinc resampler (filter: wfe_tables.cpp).  One workgroup =
// out_per_block consecutive outputs of one utterance; their input span is staged in
// LDS once (16-bit PCM converted and mixed down on the way), the filter table
// [taps][L] is streamed through LDS in chunks of rows shared by all waves, and every
// thread accumulates up to 8 outputs: acc += H[j][phase] * x[base - k0 + j], float32
// fmaf in tap order.  HBM traffic is the input once and the output once; the table
// stays in L2.
__global__ __launch_bounds__(256) void resample_kernel(const ResampleArgs a) {
  __shared__ float xs[kRsSpanFloats];
  __shared__ float hs[kRsChunkFloats];
  constexpr int U = 8;
  const int tid = threadIdx.x, b = blockIdx.y;
  const long long n0 = (long long)blockIdx.x * a.out_per_block;
  if (n0 >= a.out_stride) return;
  int n_out = a.out_lengths[b];
  if (n_out < 0) n_out = 0;
  if (n_out > a.out_stride) n_out = (int)a.out_stride;
  float* __restrict__ o = a.out + (long long)b * a.out_stride;
  long long n_end = n0 + a.out_per_block;
  if (n_end > a.out_stride) n_end = a.out_stride;
  if (n0 >= n_out) {                                   // behind the utterance: zeros
    for (long long n = n0 + tid; n < n_end; n += 256) o[n] = 0.f;
    return;
  }
  int len_in = a.lengths[b];
  if (len_in < 0) len_in = 0;
  const long long off = a.offsets ? a.offsets[b] : (long long)b * a.row_stride;
  Source src;
  src.w = a.pcm ? nullptr : a.wav + off;
  src.pcm = a.pcm ? a.pcm + off : nullptr;
  src.ch = a.channels;
  const long long n_last = (n_end < n_out ? n_end : n_out) - 1;
  const long long lo = n0 * a.M / a.L - a.k0;
  const long long hi = n_last * a.M / a.L - a.k0 + a.taps - 1;
  const int span = (int)(hi - lo + 1);                 // <= kRsSpanFloats (checked by the host)
  for (int i0 = tid; i0 < span; i0 += 256 * 8) {        // 8 loads in flight per thread, branch-free (Source::gather)
    int idx[8];
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long k = lo + i0 + 256 * u;
      idx[u] = (i0 + 256 * u < span && k >= 0 && k < len_in) ? (int)k : -1;
    }
    src.gather<8>(idx, len_in, v);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + 256 * u < span) xs[i0 + 256 * u] = v[u];
  }
  int phase[U], xoff[U];
  float acc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    long long n = n0 + tid + 256 * u;
    if (n > n_last) n = n_last;                        // idle slots redo the last output, in bounds
    const long long t = n * a.M;
    const long long base = t / a.L;
    phase[u] = (int)(t - base * a.L);
    xoff[u] = (int)(base - a.k0 - lo);
    acc[u] = 0.f;
  }
  const int rows = kRsChunkFloats / a.L;               // filter rows per chunk (>= 1)
  for (int j0 = 0; j0 < a.taps; j0 += rows) {
    const int r_here = (a.taps - j0 < rows) ? a.taps - j0 : rows;
    __syncthreads();                                   // previous chunk consumed (first pass: xs written)
    for (int i = tid; i < r_here * a.L; i += 256) hs[i] = a.table[(long long)j0 * a.L + i];
    __syncthreads();
    for (int r = 0; r < r_here; ++r) {
      const float* hrow = hs + r * a.L;
#pragma unroll
      for (int u = 0; u < U; ++u) acc[u] = fmaf(hrow[phase[u]], xs[xoff[u] + j0 + r], acc[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long long n = n0 + tid + 256 * u;
    if (256 * u < a.out_per_block && n < n_end) o[n] = (n < n_out) ? acc[u] : 0.f;
  }
}

// The same resampler with the outputs of a thread sharing ONE filter phase.  Outputs n and n + S have the same
// phase when L divides S, and their input windows start S M / L samples apart; with S = the smallest multiple
// of L that is >= 256 a workgroup of S threads (rounded up to whole waves) takes U S consecutive outputs,
// thread t the outputs n0 + t + S i.  Per tap a thread then reads the filter value once for its U outputs:
// (U + 1) / U LDS reads per fma where resample_kernel above needs two -- the kernel is LDS-bound, so that is
// its speed.  Every output still accumulates its taps in ascending order with fmaf: the same bits.
// Used when S <= 1024 (launch_resample); resample_kernel remains for other ratios.
__global__ __launch_bounds__(1024) void resample_phase_kernel(const ResampleArgs a, const int S, const int U) {
  __shared__ float xs[kRsSpanFloats];
  __shared__ float hs[kRsChunkFloats];
  constexpr int kU = 8;
  const int tid = threadIdx.x, nthreads = blockDim.x, b = blockIdx.y;
  const int opb = S * U;
  const long long n0 = (long long)blockIdx.x * opb;
  if (n0 >= a.out_stride) return;
  int n_out = a.out_lengths[b];
  if (n_out < 0) n_out = 0;
  if (n_out > a.out_stride) n_out = (int)a.out_stride;
  float* __restrict__ o = a.out + (long long)b * a.out_stride;
  long long n_end = n0 + opb;
  if (n_end > a.out_stride) n_end = a.out_stride;
  if (n0 >= n_out) {                                   // behind the utterance: zeros
    for (long long n = n0 + tid; n < n_end; n += nthreads) o[n] = 0.f;
    return;
  }
  int len_in = a.lengths[b];
  if (len_in < 0) len_in = 0;
  const long long off = a.offsets ? a.offsets[b] : (long long)b * a.row_stride;
  Source src;
  src.w = a.pcm ? nullptr : a.wav + off;
  src.pcm = a.pcm ? a.pcm + off : nullptr;
  src.ch = a.channels;
  const long long n_last = (n_end < n_out ? n_end : n_out) - 1;
  const long long lo = n0 * a.M / a.L - a.k0;
  const long long hi = n_last * a.M / a.L - a.k0 + a.taps - 1;
  const int span = (int)(hi - lo + 1);                 // <= kRsSpanFloats (checked by the host)
  for (int i0 = tid; i0 < span; i0 += nthreads * 8) {   // 8 loads in flight per thread, branch-free (Source::gather)
    int idx[8];
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long k = lo + i0 + nthreads * u;
      idx[u] = (i0 + nthreads * u < span && k >= 0 && k < len_in) ? (int)k : -1;
    }
    src.gather<8>(idx, len_in, v);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + nthreads * u < span) xs[i0 + nthreads * u] = v[u];
  }
  // this thread's phase and the window starts of its outputs (slots behind the last output redo output 0's window)
  const bool worker = tid < S && n0 + tid <= n_last;
  const long long t = (n0 + (worker ? tid : 0)) * a.M;
  const long long base = t / a.L;
  const int phase = (int)(t - base * a.L);
  const int stride = S / a.L * a.M;                    // window distance of two outputs S apart
  int xoff[kU];
  float acc[kU];
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    const bool live = worker && u < U && n0 + tid + (long long)S * u <= n_last;
    xoff[u] = (int)(base - a.k0 - lo) + (live ? stride * u : 0);
    acc[u] = 0.f;
  }
  const int rows = kRsChunkFloats / a.L;               // filter rows per chunk (>= 1)
  for (int j0 = 0; j0 < a.taps; j0 += rows) {
    const int r_here = (a.taps - j0 < rows) ? a.taps - j0 : rows;
    __syncthreads();                                   // previous chunk consumed (first pass: xs written)
    for (int i = tid; i < r_here * a.L; i += nthreads) hs[i] = a.table[(long long)j0 * a.L + i];
    __syncthreads();
    const float* hp = hs + phase;
    int r = 0;
    for (; r + 4 <= r_here; r += 4) {
      const float h0 = hp[r * a.L], h1 = hp[(r + 1) * a.L], h2 = hp[(r + 2) * a.L], h3 = hp[(r + 3) * a.L];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const float* x = xs + xoff[u] + j0 + r;
        acc[u] = fmaf(h0, x[0], acc[u]);
        acc[u] = fmaf(h1, x[1], acc[u]);
        acc[u] = fmaf(h2, x[2], acc[u]);
        acc[u] = fmaf(h3, x[3], acc[u]);
      }
    }
    for (; r < r_here; ++r) {
      const float h0 = hp[r * a.L];
#pragma unroll
      for (int u = 0; u < kU; ++u) acc[u] = fmaf(h0, xs[xoff[u] + j0 + r], acc[u]);
    }
  }
  if (tid < S) {
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const long long n = n0 + tid + (long long)S * u;
      if (u < U && n < n_end) o[n] = (n < n_out) ? acc[u] : 0.f;
    }
  }
}

// Few filter phases (L <= 4: 48 / 32 / 64 / 96 kHz -> 16 kHz with L = 1; 8 -> 16, 24 -> 16, 12 -> 16 kHz with L = 2, 4),
// sliding-window form.  The outputs of ONE phase p, n = p + L m, are a decimation by M of the input with the
// filter column (p M) mod L: output m reads the inputs M m + floor(p M / L) - k0 + j.  blockIdx.z is the phase; thread t
// owns 7 CONSECUTIVE outputs of it, whose inputs at one tap are M apart and move by one sample from tap to tap,
// so the 6 M + 1 samples they span live in registers (rotated through an unrolled block of 6 M + 1 taps) and a
// tap costs one new LDS read and one broadcast read of the tap value for 7 fma, where resample_kernel needs 14.
// (7, not 8, outputs: the lanes' reads are 7 M samples apart, an odd or at worst twice-odd stride over the
// banks.)  Taps in ascending order per output: the same bits.
template <int M>
__global__ __launch_bounds__(256) void resample_decimate_kernel(const ResampleArgs a) {
  constexpr int U = 7, W = (U - 1) * M + 1, kOut = 256 * U;
  constexpr int kTapsMax = 1024;
  __shared__ float xs[kRsSpanFloats];
  __shared__ float hs[kTapsMax];
  const int tid = threadIdx.x, b = blockIdx.y, L = a.L, ph = blockIdx.z;
  const long long m0 = (long long)blockIdx.x * kOut;   // first output of this workgroup, counted within the phase
  const long long n_first = ph + L * m0;
  if (n_first >= a.out_stride) return;
  int n_out = a.out_lengths[b];
  if (n_out < 0) n_out = 0;
  if (n_out > a.out_stride) n_out = (int)a.out_stride;
  float* __restrict__ o = a.out + (long long)b * a.out_stride;
  if (n_first >= n_out) {                              // behind the utterance: zeros
    for (int i = tid; i < kOut; i += 256) {
      const long long n = ph + L * (m0 + i);
      if (n < a.out_stride) o[n] = 0.f;
    }
    return;
  }
  int len_in = a.lengths[b];
  if (len_in < 0) len_in = 0;
  const long long off = a.offsets ? a.offsets[b] : (long long)b * a.row_stride;
  Source src;
  src.w = a.pcm ? nullptr : a.wav + off;
  src.pcm = a.pcm ? a.pcm + off : nullptr;
  src.ch = a.channels;
  const int taps_p = (a.taps + W - 1) / W * W;         // whole blocks of W taps, the tail padded with zeros
  const int col = ph * M % L;                          // the filter column of output n is (n M) mod L
  const long long lo = m0 * M + (long long)ph * M / L - a.k0;   // span position i is input lo + i
  const int span = (kOut - 1) * M + taps_p + W;        // every slot of the workgroup, live or not (host: <= kRsSpanFloats)
  for (int i0 = tid; i0 < span; i0 += 256 * 8) {       // 8 loads in flight per thread, branch-free (Source::gather)
    int idx[8];
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long k = lo + i0 + 256 * u;
      idx[u] = (i0 + 256 * u < span && k >= 0 && k < len_in) ? (int)k : -1;
    }
    src.gather<8>(idx, len_in, v);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + 256 * u < span) xs[i0 + 256 * u] = v[u];
  }
  for (int i = tid; i < taps_p; i += 256) hs[i] = i < a.taps ? a.table[(long long)i * L + col] : 0.f;
  __syncthreads();
  const float* __restrict__ xw = xs + U * M * tid;     // output i of this thread reads xw[j + M i] at tap j
  float w[W], acc[U];
#pragma unroll
  for (int r = 0; r < W; ++r) w[r] = xw[r];
#pragma unroll
  for (int i = 0; i < U; ++i) acc[i] = 0.f;
  for (int jb = 0; jb < taps_p; jb += W) {
#pragma unroll
    for (int r = 0; r < W; ++r) {
      const float hv = hs[jb + r];
      // position j + M i of the window sits in slot (j + M i) mod W; jb is a multiple of W
#pragma unroll
      for (int i = 0; i < U; ++i) acc[i] = fmaf(hv, w[(r + M * i) % W], acc[i]);
      w[r] = xw[jb + r + W];                           // the sample that enters as the one tap j used for output 0 leaves
    }
  }
#pragma unroll
  for (int i = 0; i < U; ++i) {
    const long long n = ph + L * (m0 + U * tid + i);
    if (n < a.out_stride) o[n] = (n < n_out) ? acc[i] : 0.f;
  }
}

// Rational ratios with many phases (44.1 / 22.05 / 11.025 kHz <-> 16 kHz) on the matrix cores.  The L outputs
// n = g L + p of group g read the inputs g M + floor(p M / L) - k0 + j with the filter column (p M) mod L -- the
// SAME banded L x K matrix T[p][c] = table[c - floor(p M / L)][(p M) mod L] (K = M - 1 + taps) for every group:
//   out[g L + p] = sum_c T[p][c] x[g M - k0 + c],
// a GEMM of T with the overlapping input windows of the groups.  A workgroup stages the input span of 16 GT
// groups in LDS once; wave w takes the tiles of 16 phases w, w + nwaves, ...: v_mfma_f32_16x16x4_f32 with T as
// the A operand (fragment order, built on the host at table creation: one 16-byte load per lane feeds 4 k-steps
// x GT MFMAs, only the k range where the tile's band is non-zero) and the windows as B, read straight from the
// span (lane (g, k) reads xs[g M + c + k]: one ds_read per MFMA, immediate offsets).  Products with T's zeros add
// exact zeros; every output still sums its taps in ascending order, in float32 (the order INSIDE one 4-wide
// k-step is the matrix core's).
// PAD (M a multiple of 8, e.g. 16 -> 44.1 kHz with M = 160): the 16 windows of a k-step would start a multiple of
// 8 floats apart and meet in 4 or fewer LDS banks, so the span is stored with 4 extra floats per M (span index i
// at i + 4 (i / M): window stride M + 4, 4 x odd); a k-step's 4 columns never cross a multiple of M, so its
// extra offset is one wave-uniform number.
// ---- the pieces of the matrix-core resampler, shared by its two kernels
// Stage the input span [lo, lo + span) of one utterance into LDS with the threads t = 0 .. nt - 1 of the caller.
template <bool PAD, int U = 4>
__device__ __forceinline__ void rs_stage_span(const ResampleArgs& a, const Source& src, int len_in, long long lo, int span,
                                              int M, float* __restrict__ xs, int t, int nt) {
  if (!a.pcm && lo >= 0) {
    // float32 source: 16-byte buffer loads (4-byte aligned is enough; dwords behind the utterance read as 0), U in
    // flight per thread -- a CU keeps a bounded NUMBER of loads in flight, so the load instruction is the unit of cost
    const __amdgpu_buffer_rsrc_t rx = f32_rsrc(src.w, lo, (long long)len_in - lo);
    const int nvec = (span + 3) >> 2;
    for (int v0 = t; v0 < nvec; v0 += nt * U) {
      f32x4 X[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
#if WFE_RS_ABLATE & 1
        X[u] = f32x4{(float)v0, 1.f, 2.f, 3.f};
#else
        X[u] = buf_f32x4(rx, (v0 + nt * u) * 16);
#endif
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (v0 + nt * u < nvec) {
          const int v = v0 + nt * u;
          reinterpret_cast<f32x4*>(xs)[PAD ? v + (4 * v) / M : v] = X[u];     // (M % 4 == 0: a quad stays inside one M)
        }
    }
  } else {
    for (int i0 = t; i0 < span; i0 += nt * 8) {        // 8 loads in flight per thread, branch-free (Source::gather)
      int idx[8];
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long long k = lo + i0 + nt * u;
        idx[u] = (i0 + nt * u < span && k >= 0 && k < len_in) ? (int)k : -1;
      }
      src.gather<8>(idx, len_in, v);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + nt * u < span) {
          const int i = i0 + nt * u;
          xs[PAD ? i + 4 * (i / M) : i] = v[u];
        }
    }
  }
}

// Column n of tile gt is group gt * 16 + n -- or, for odd M, tiles 2 t and 2 t + 1 take the even and the odd
// groups of a run of 32: the 16 lanes of one k then read 2 M floats apart, an even stride with M odd, so with
// the neighbouring k they cover 32 distinct LDS banks (M apart, the odd stride, half of them collide).
template <int GT>
__device__ __forceinline__ int rs_group_of(int gt, int lane, int M) {
  const bool interleave = GT >= 2 && (M & 1);
  return interleave ? (gt & 1) + 2 * ((lane & 15) + 16 * (gt >> 1)) : gt * 16 + (lane & 15);
}

// One unit of work: phase tile pt x the group tiles [G0, G0 + GN) of the span xs; store(group, phase p, v)
// receives the lane's results for the phases p .. p + 3 of that group.
template <int GT, bool PAD, int G0, int GN, class Store>
__device__ __forceinline__ void rs_unit(const ResampleArgs& a, const float* __restrict__ xs, int pt, int lane, Store store) {
  constexpr int P = PAD ? 4 : 0;
  const int M = a.M;
  const int kb_lo = __builtin_amdgcn_readfirstlane(a.frag_meta[3 * pt]);
  const int nq = __builtin_amdgcn_readfirstlane(a.frag_meta[3 * pt + 1]);       // quads of k-steps
  const int foff = __builtin_amdgcn_readfirstlane(a.frag_meta[3 * pt + 2]);     // in 64-lane float4 fragments
  const float* xb[GT];
#pragma unroll
  for (int gt = G0; gt < G0 + GN; ++gt)
    xb[gt] = xs + rs_group_of<GT>(gt, lane, M) * (M + P) + (lane >> 4) + (PAD ? 0 : 4 * kb_lo);
  f32x4 acc[GT];
#pragma unroll
  for (int gt = G0; gt < G0 + GN; ++gt) acc[gt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Fragments through a buffer resource (quad offset in an SGPR, one lane-offset VGPR), two register sets in
  // rotation with the loads pinned TWO quads ahead of their use (round 2's single-quad "prefetch" compiled to a
  // load and a full wait per quad).
  const __amdgpu_buffer_rsrc_t fr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.frag), 0, 0x7ffffff0, 0x00020000);
  const int flane = lane * 16;
  auto ldA = [&](int q) {                              // quad q of this tile (clamped: a repeat is never used)
    const int qq = q < nq ? q : nq - 1;
#if WFE_RS_ABLATE & 2
    return f32x4{(float)qq, 1.f, (float)flane, 3.f};
#else
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(fr, flane, (foff + qq) * 1024, 0));
#endif
  };
  if (!PAD) {
    // the quad loop unrolled by four: every window read is lane base + immediate, the bases move once per 16 GN MFMAs
    auto quad = [&](const f32x4& A, int base) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int gt = G0; gt < G0 + GN; ++gt) {
#if WFE_RS_ABLATE & 4
          const float bv = A[(i + 1) & 3];
#else
          const float bv = xb[gt][base + 4 * i];
#endif
#if WFE_RS_ABLATE & 16
          acc[gt][i] += A[i] * bv;
#else
          acc[gt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i], bv, acc[gt], 0, 0, 0);
#endif
        }
    };
    f32x4 X0 = ldA(0), X1 = ldA(1), Y0, Y1;
    int q = 0;
    for (; q + 4 <= nq; q += 4) {
      Y0 = ldA(q + 2), Y1 = ldA(q + 3);
      __builtin_amdgcn_sched_barrier(0);
      quad(X0, 0);
      quad(X1, 16);
      X0 = ldA(q + 4), X1 = ldA(q + 5);
      __builtin_amdgcn_sched_barrier(0);
      quad(Y0, 32);
      quad(Y1, 48);
#pragma unroll
      for (int gt = G0; gt < G0 + GN; ++gt) xb[gt] += 64;
    }
    if (q + 2 < nq) Y0 = ldA(q + 2);
    if (q < nq) quad(X0, 0);
    if (q + 1 < nq) quad(X1, 16);
    if (q + 2 < nq) quad(Y0, 32);
  } else {
    // PAD: the k-step's LDS offset is a wave-uniform running number (column + 4 per M crossed)
    int col = 4 * kb_lo, rem = col % M, cpad = col + P * (col / M);
    f32x4 A = ldA(0);
    for (int q = 0; q < nq; ++q) {
      const f32x4 An = ldA(q + 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int gt = G0; gt < G0 + GN; ++gt)
          acc[gt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i], xb[gt][cpad], acc[gt], 0, 0, 0);
        rem += 4;
        cpad += 4;
        if (rem >= M) rem -= M, cpad += P;
      }
      A = An;
    }
  }
#pragma unroll
  for (int gt = G0; gt < G0 + GN; ++gt) store(rs_group_of<GT>(gt, lane, M), 16 * pt + 4 * (lane >> 4), acc[gt]);
}

// The units of wave `wave` of `nwaves`.  Round 2 gave wave w the tiles w, w + nwaves, ... whole; with 10 tiles on
// 10 waves -- waves are dealt to the SIMDs round-robin, wave w runs on SIMD w & 3 -- the SIMDs carry 3 : 3 : 2 : 2
// tiles.  Balanced (nwaves a multiple of 4): the first W = ptiles - ptiles % 4 tiles go WHOLE to the SIMDs in turn
// (tile t to SIMD t & 3, there to that SIMD's waves in rotation); the remaining tiles are split along the group
// tiles into GT units each and dealt on in the same rotation: L = 160 on 12 waves is 2 whole tiles + 1 half tile
// per SIMD.  (A split tile's fragments are loaded once per unit.)
template <int GT, class F>
__device__ __forceinline__ void rs_deal_units(const ResampleArgs& a, int wave, int nwaves, F f) {
  using std::integral_constant;
  if ((a.tune & 256) || a.ptiles < 8 || (nwaves & 3)) {
    for (int pt = wave; pt < a.ptiles; pt += nwaves) f(pt, integral_constant<int, 0>{}, integral_constant<int, GT>{});
    return;
  }
  const int simd = wave & 3, slot = wave >> 2, c = nwaves >> 2;
  const int W = a.ptiles & ~3;
  for (int t = simd + 4 * slot; t < W; t += 4 * c) f(t, integral_constant<int, 0>{}, integral_constant<int, GT>{});
  const int units = (a.ptiles - W) * GT;
  for (int k = 0; simd + 4 * k < units; ++k) {
    if ((W / 4 + k) % c != slot) continue;
    const int v = simd + 4 * k, pt = W + v / GT, g = v % GT;
    if (GT == 1 || g == 0) f(pt, integral_constant<int, 0>{}, integral_constant<int, 1>{});
    else if (g == 1) f(pt, integral_constant<int, (GT > 1 ? 1 : 0)>{}, integral_constant<int, 1>{});
    else if (g == 2) f(pt, integral_constant<int, (GT > 2 ? 2 : 0)>{}, integral_constant<int, 1>{});
    else f(pt, integral_constant<int, (GT > 3 ? 3 : 0)>{}, integral_constant<int, 1>{});
  }
}

template <int GT, bool PAD>
__global__ __launch_bounds__(1024) void resample_mfma_kernel(const ResampleArgs a) {
  extern __shared__ __attribute__((aligned(16))) float xs_dyn[];
  float* xs = xs_dyn;
  constexpr int G = 16 * GT;
  const int tid = threadIdx.x, b = blockIdx.y, L = a.L, M = a.M;
  const int nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
  const long long g0 = (long long)blockIdx.x * G;
  const long long n_first = g0 * L;
  if (n_first >= a.out_stride) return;
  int n_out = a.out_lengths[b];
  if (n_out < 0) n_out = 0;
  if (n_out > a.out_stride) n_out = (int)a.out_stride;
  float* __restrict__ o = a.out + (long long)b * a.out_stride;
  long long n_end = n_first + (long long)G * L;
  if (n_end > a.out_stride) n_end = a.out_stride;
  if (n_first >= n_out) {                              // behind the utterance: zeros
    for (long long n = n_first + tid; n < n_end; n += nthreads) o[n] = 0.f;
    return;
  }
  int len_in = a.lengths[b];
  if (len_in < 0) len_in = 0;
  const long long off = a.offsets ? a.offsets[b] : (long long)b * a.row_stride;
  Source src;
  src.w = a.pcm ? nullptr : a.wav + off;
  src.pcm = a.pcm ? a.pcm + off : nullptr;
  src.ch = a.channels;
  const long long lo = g0 * M - a.k0;                  // span position i is input lo + i
  const int span = (G - 1) * M + a.kspan;              // (host: fits the dynamic LDS of the launch, rounded up to 4)
  rs_stage_span<PAD>(a, src, len_in, lo, span, M, xs, tid, nthreads);
  __syncthreads();
  // a lane's four consecutive outputs as ONE 16-byte store where they are whole and aligned: the 4 k-lanes of a
  // group then fill a 64-byte segment per instruction (four scattered 4-byte stores per lane before: 44.1 -> 16 kHz
  // 7 % slower, 22.05 -> 16 kHz 20 %)
  const bool vec_ok = (L & 3) == 0 && (a.out_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(o) & 15u) == 0;
  auto store = [&](int group, int p, const f32x4& v) {
    const long long n = (g0 + group) * L + p;
#if WFE_RS_ABLATE & 8
    if (p < L && n < a.out_stride && v[0] == 12345.678f) o[n] = 1.f;
#else
    if (vec_ok && p + 3 < L && n + 3 < a.out_stride) {
      *reinterpret_cast<f32x4*>(o + n) = f32x4{n < n_out ? v[0] : 0.f, n + 1 < n_out ? v[1] : 0.f,
                                                n + 2 < n_out ? v[2] : 0.f, n + 3 < n_out ? v[3] : 0.f};
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (p + i < L && n + i < a.out_stride) o[n + i] = (n + i < n_out) ? v[i] : 0.f;
    }
#endif
  };
  rs_deal_units<GT>(a, wave, nwaves, [&](int pt, auto g0c, auto gnc) {
    rs_unit<GT, PAD, decltype(g0c)::value, decltype(gnc)::value>(a, xs, pt, lane, store);
  });
}

// (Round 3 also built this product as a PERSISTENT, software-pipelined workgroup -- one per CU, two span buffers,
// 12 compute waves multiplying block i while 4 loader waves staged block i + 1 and wrote block i - 1 out of an LDS
// tile; identical bits, and slower: 44.1 -> 16 kHz 1.41 - 1.54 ms per 256 x 30 s against 1.27 for the kernel above.
// With one workgroup per CU nothing hides a block's serial latencies -- its scalar loads, the first fragment's L2
// round trip, the barrier behind the slowest wave -- that a second, independent workgroup hides for free.  Removed.)

// ------------------------------------------------------------- reverberation
// scipy.signal.fftconvolve(sample, rir, mode='same') restated as the direct sum
//   y[n] = sum_k rir[k] * x[n + (Lr - 1) / 2 - k],   n in [0, len),
// float32 fmaf in tap order (the reference's float32 FFT agrees to ~1e-6 of max|y|),
// then the peak normalisation y * (max|x| / max|y|) of reader.py:346-347 in a
// second pass.  One workgroup = 2048 outputs of one utterance (thread t owns
// n0 + t + 256 i): per chunk of 512 taps the taps and the input span they touch are
// staged in LDS; lanes read consecutive addresses.
// Sliding-window form (round 2).  reverb_kernel below reads one input sample from LDS per fma (thread t owns
// outputs n0 + t + 256 i) and is LDS-bound at 36 TFLOP/s.  Here thread t owns the 8 CONSECUTIVE outputs
// n0 + 8 t + i: from one tap to the next its 8 inputs shift by one sample, so the window lives in registers
// (rotated through an unrolled block of 8 taps) and a tap costs ONE new LDS read for 8 fma, plus the tap value
// (a broadcast read, four taps per ds_read_b128).  The lanes' reads are 8 samples apart: the span is stored with
// one pad word per 8 (index s + s / 8, lane stride 9) so that they fall on distinct banks.  Every output still
// accumulates its taps in ascending order with fmaf: the same bits as reverb_kernel (tools/reverb_ab.py).
constexpr int kRvOut = 2048, kRvTaps = 512;
__device__ __forceinline__ int rv_pad(int s) { return s + (s >> 3); }
__global__ __launch_bounds__(256) void reverb_window_kernel(const ReverbArgs a) {
  constexpr int kSpan = kRvOut + kRvTaps + 8;
  __shared__ float xs[kSpan + kSpan / 8 + 8];
  __shared__ __attribute__((aligned(16))) float hs[kRvTaps];
  const int tid = threadIdx.x, b = blockIdx.y;
  const long long n0 = (long long)blockIdx.x * kRvOut;
  if (n0 >= a.out_stride) return;
  int len = a.lengths[b];
  if (len < 0) len = 0;
  if (len > a.out_stride) len = (int)a.out_stride;
  float* __restrict__ o = a.out + (long long)b * a.out_stride;
  long long n_end = n0 + kRvOut;
  if (n_end > a.out_stride) n_end = a.out_stride;
  if (n0 >= len) {
    for (long long n = n0 + tid; n < n_end; n += 256) o[n] = 0.f;
    return;
  }
  const float* __restrict__ x = a.wav + (a.offsets ? a.offsets[b] : (long long)b * a.row_stride);
  const int ri = a.rir_index[b];
  if (ri < 0) {                                        // no reverb for this utterance
    for (long long n = n0 + tid; n < n_end; n += 256) o[n] = n < len ? x[n] : 0.f;
    return;
  }
  Source xsrc;
  xsrc.w = x;
  xsrc.pcm = nullptr;
  xsrc.ch = 1;
  const long long h0 = a.rir_off[ri];
  const int lr = (int)(a.rir_off[ri + 1] - h0);
  const int c = (lr - 1) / 2;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < lr; k0 += kRvTaps) {
    const int kc = (lr - k0 < kRvTaps) ? lr - k0 : kRvTaps;
    const int kc8 = (kc + 7) & ~7;                     // taps of the chunk, padded with zeros to whole blocks of 8
    // inputs touched by taps [k0, k0 + kc8): s = n + c - k, n in [n0, n0 + 2048); position i of the span is input s_lo + i
    const long long s_lo = n0 + c - (k0 + kc8 - 1);
    const int span = kRvOut + kc8 - 1;
    __syncthreads();
    for (int i0 = tid; i0 < span; i0 += 256 * 10) {    // the whole span in one batch of 10 loads per thread
      int idx[10];
      float v[10];
#pragma unroll
      for (int u = 0; u < 10; ++u) {
        const long long s2 = s_lo + i0 + 256 * u;
        idx[u] = (i0 + 256 * u < span && s2 >= 0 && s2 < len) ? (int)s2 : -1;
      }
      xsrc.gather<10>(idx, len, v);
#pragma unroll
      for (int u = 0; u < 10; ++u)
        if (i0 + 256 * u < span) xs[rv_pad(i0 + 256 * u)] = v[u];
    }
    for (int i = tid; i < kc8; i += 256) hs[i] = i < kc ? a.rir[h0 + k0 + i] : 0.f;   // (fmaf(0, x, acc) == acc: x is finite or the output is NaN anyway)
    __syncthreads();
    // tap kk of output 8 t + i reads span position 8 t + i + d, d = kc8 - 1 - kk; d runs over whole blocks 8 g + 7 .. 8 g
    const float* __restrict__ xw = xs + 9 * tid;       // rv_pad(8 t + e) = 9 t + e + (e >> 3)
    float w[8];
    {
      const int d = kc8 - 1;                           // == 7 mod 8
#pragma unroll
      for (int i = 0; i < 8; ++i) w[i] = xw[rv_pad(i + d)];
    }
    for (int kb = 0; kb < kc8; kb += 8) {
      const f32x4 ha = *reinterpret_cast<const f32x4*>(hs + kb), hb = *reinterpret_cast<const f32x4*>(hs + kb + 4);
      const int g9 = 9 * ((kc8 - 8 - kb) >> 3);        // the block's new samples sit at e = 8 g + 6 .. 8 g - 1
      const float* __restrict__ xn = xw + g9;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float hv = r < 4 ? ha[r & 3] : hb[r & 3];
        // window before tap kb + r: w[(i - r) & 7] holds the input of output i
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fmaf(hv, w[(i - r) & 7], acc[i]);
        // shift by one sample: output 0's next input replaces the slot output 7 used
        // e = 8 g + 6 - r; for r = 7: e = 8 g - 1, one pad word back (nothing follows the chunk's last tap)
        w[(7 - r) & 7] = (r < 7) ? xn[6 - r] : (g9 > 0 ? xn[-2] : 0.f);
      }
    }
  }
  float pk_y = 0.f, pk_x = 0.f;
  const long long nb = n0 + 8 * tid;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const long long n = nb + i;
    if (n < n_end) {
      const bool in = n < len;
      o[n] = in ? acc[i] : 0.f;
      if (in) {
        pk_y = fmaxf(pk_y, fabsf(acc[i]));
        pk_x = fmaxf(pk_x, fabsf(x[n]));
      }
    }
  }
  pk_y = wave_reduce<true>(pk_y);
  pk_x = wave_reduce<true>(pk_x);
  if ((tid & 63) == 0) {                               // NaN peaks (NaN input) are not propagated: max ignores them
    atomicMax(a.peaks + 2 * b, __float_as_uint(pk_x));
    atomicMax(a.peaks + 2 * b + 1, __float_as_uint(pk_y));
  }
}

__global__ __launch_bounds__(256) void reverb_kernel(const ReverbArgs a) {
  __shared__ float xs[kRvOut + kRvTaps];
  __shared__ float hs[kRvTaps];
  const int tid = threadIdx.x, b = blockIdx.y;
  const long long n0 = (long long)blockIdx.x * kRvOut;
  if (n0 >= a.out_stride) return;
  int len = a.lengths[b];
  if (len < 0) len = 0;
  if (len > a.out_stride) len = (int)a.out_stride;
  float* __restrict__ o = a.out + (long long)b * a.out_stride;
  long long n_end = n0 + kRvOut;
  if (n_end > a.out_stride) n_end = a.out_stride;
  if (n0 >= len) {
    for (long long n = n0 + tid; n < n_end; n += 256) o[n] = 0.f;
    return;
  }
  const float* __restrict__ x = a.wav + (a.offsets ? a.offsets[b] : (long long)b * a.row_stride);
  const int ri = a.rir_index[b];
  if (ri < 0) {                                        // no reverb for this utterance
    for (long long n = n0 + tid; n < n_end; n += 256) o[n] = n < len ? x[n] : 0.f;
    return;
  }
  Source xsrc;
  xsrc.w = x;
  xsrc.pcm = nullptr;
  xsrc.ch = 1;
  const long long h0 = a.rir_off[ri];
  const int lr = (int)(a.rir_off[ri + 1] - h0);
  const int c = (lr - 1) / 2;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < lr; k0 += kRvTaps) {
    const int kc = (lr - k0 < kRvTaps) ? lr - k0 : kRvTaps;
    // inputs touched by taps [k0, k0 + kc): s = n + c - k, n in [n0, n0 + 2048)
    const long long s_lo = n0 + c - (k0 + kc - 1);
    __syncthreads();
    for (int i0 = tid; i0 < kRvOut + kc - 1; i0 += 256 * 10) {   // the whole span in one batch of 10 loads per thread
      int idx[10];
      float v[10];
#pragma unroll
      for (int u = 0; u < 10; ++u) {
        const long long s2 = s_lo + i0 + 256 * u;
        idx[u] = (i0 + 256 * u < kRvOut + kc - 1 && s2 >= 0 && s2 < len) ? (int)s2 : -1;
      }
      xsrc.gather<10>(idx, len, v);
#pragma unroll
      for (int u = 0; u < 10; ++u)
        if (i0 + 256 * u < kRvOut + kc - 1) xs[i0 + 256 * u] = v[u];
    }
    for (int i = tid; i < kc; i += 256) hs[i] = a.rir[h0 + k0 + i];
    __syncthreads();
    for (int kk = 0; kk < kc; ++kk) {
      const float hv = hs[kk];
      const int base = tid + (kc - 1 - kk);            // n0 + tid + c - (k0 + kk) - s_lo
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = fmaf(hv, xs[base + 256 * i], acc[i]);
    }
  }
  float pk_y = 0.f, pk_x = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const long long n = n0 + tid + 256 * i;
    if (n < n_end) {
      const bool in = n < len;
      o[n] = in ? acc[i] : 0.f;
      if (in) {
        pk_y = fmaxf(pk_y, fabsf(acc[i]));
        pk_x = fmaxf(pk_x, fabsf(x[n]));
      }
    }
  }
  pk_y = wave_reduce<true>(pk_y);
  pk_x = wave_reduce<true>(pk_x);
  if ((tid & 63) == 0) {                               // NaN peaks (NaN input) are not propagated: max ignores them
    atomicMax(a.peaks + 2 * b, __float_as_uint(pk_x));
    atomicMax(a.peaks + 2 * b + 1, __float_as_uint(pk_y));
  }
}

// sample_convolved * (max|sample| / max|sample_convolved|): float32 division, float32 product
__global__ __launch_bounds__(256) void reverb_scale_kernel(const ReverbArgs a) {
  const int b = blockIdx.y;
  if (a.rir_index[b] < 0) return;
  int len = a.lengths[b];
  if (len > a.out_stride) len = (int)a.out_stride;
  const float rate = div_rn(__uint_as_float(a.peaks[2 * b]), __uint_as_float(a.peaks[2 * b + 1]));
  float* __restrict__ o = a.out + (long long)b * a.out_stride;
  for (long long n = (long long)blockIdx.x * 256 + threadIdx.x; n < len; n += (long long)gridDim.x * 256)
    o[n] = mul_rn(o[n], rate);
}

// ------------------------------------------------------------- launchers
template <int NMT, bool POWER, bool AUG, int BANK>
static hipError_t launch_k1(const FrontArgs& a, int batch, hipStream_t stream) {
  const dim3 grid((a.n_frames + kBlockFrames - 1) / kBlockFrames, batch);
  // (WFE_TUNE & 8, diagnostic: 36 KB of unused dynamic LDS leave room for ONE workgroup per CU -- phases without a partner)
  hipLaunchKernelGGL((frontend_kernel<NMT, POWER, AUG, BANK>), grid, dim3(256), (a.tune & 8) ? 36 * 1024 : 0, stream, a);
  return hipGetLastError();
}

// bank: 1 = the 80-mel Whisper bank, 2 = the 128-mel bank (compile-time block
// masks), 0 = any other feature_size (run-time masks).
hipError_t launch_frontend(const FrontArgs& a, int batch, int mode, bool aug, int bank, hipStream_t stream) {
#ifdef WFE_FAST_BUILD          // experiment builds (minutes less to compile): the two plain log-mel kernels only
#if WFE_FAST_BUILD + 0 == 2    // (-DWFE_FAST_BUILD=2: and the 80-mel augmented one with its helpers)
  if (mode == 0 && aug && bank == 1) return launch_k1<5, false, true, 1>(a, batch, stream);
#endif
  if (mode == 0 && !aug && bank == 1) return launch_k1<5, false, false, 1>(a, batch, stream);
  if (mode == 0 && !aug && bank == 2) return launch_k1<8, false, false, 2>(a, batch, stream);
  if (mode == 1 && !aug) return launch_k1<5, true, false, 0>(a, batch, stream);
  return hipErrorInvalidValue;
#else
  if (mode == 1)
    return aug ? launch_k1<5, true, true, 0>(a, batch, stream) : launch_k1<5, true, false, 0>(a, batch, stream);
  if (bank == 1)
    return aug ? launch_k1<5, false, true, 1>(a, batch, stream) : launch_k1<5, false, false, 1>(a, batch, stream);
  if (bank == 2)
    return aug ? launch_k1<8, false, true, 2>(a, batch, stream) : launch_k1<8, false, false, 2>(a, batch, stream);
  if (a.n_mel <= 80)
    return aug ? launch_k1<5, false, true, 0>(a, batch, stream) : launch_k1<5, false, false, 0>(a, batch, stream);
  return aug ? launch_k1<8, false, true, 0>(a, batch, stream) : launch_k1<8, false, false, 0>(a, batch, stream);
#endif
}

hipError_t launch_clamp(float* out, int out_kind, const float* tile_stats, int batch, int n_mel, int n_frames,
                        hipStream_t stream) {
  const dim3 grid(kClampParts, batch);
  hipLaunchKernelGGL(clamp_kernel, grid, dim3(256), 0, stream, out, out_kind, tile_stats, n_mel, n_frames);
  return hipGetLastError();
}

hipError_t launch_noise_meansq(const float* noise, const long long* off, int n_noise, double* meansq,
                               hipStream_t stream) {
  if (n_noise <= 0) return hipSuccess;
  hipLaunchKernelGGL(noise_meansq_kernel, dim3(kNoiseParts, n_noise), dim3(256), 0, stream, noise, off, meansq);
  return hipGetLastError();
}

hipError_t launch_resolve(const FrontArgs& a, int batch, double* partials, unsigned* tickets, hipStream_t stream) {
#if defined(WFE_FAST_BUILD) && WFE_FAST_BUILD + 0 != 2
  return hipErrorInvalidValue;
#endif
  hipLaunchKernelGGL(resolve_partial_kernel, dim3(kResolveParts, batch), dim3(256), 0, stream, a, partials, tickets);
  return hipGetLastError();
}

hipError_t launch_waveform(const FrontArgs& a, int batch, float* out, int* out_lengths, hipStream_t stream) {
  const int gx = (a.n_samples + kBlockFrames * kHop - 1) / (kBlockFrames * kHop);
#if defined(WFE_FAST_BUILD) && WFE_FAST_BUILD + 0 != 2
  return hipErrorInvalidValue;
#endif
  hipLaunchKernelGGL(waveform_kernel, dim3(gx < 1 ? 1 : gx, batch), dim3(256), 0, stream, a, out, out_lengths);
  return hipGetLastError();
}

hipError_t launch_reverb(const ReverbArgs& a, int batch, hipStream_t stream) {
  const long long gx = (a.out_stride + kRvOut - 1) / kRvOut;
  if (a.tune & 4)                                       // (WFE_TUNE & 4: round 1's kernel, for A/B)
    hipLaunchKernelGGL(reverb_kernel, dim3((unsigned)(gx < 1 ? 1 : gx), batch), dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL(reverb_window_kernel, dim3((unsigned)(gx < 1 ? 1 : gx), batch), dim3(256), 0, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return launch_reverb_scale(a, batch, stream);
}

hipError_t launch_reverb_scale(const ReverbArgs& a, int batch, hipStream_t stream) {
  long long sx = (a.out_stride + 256 * 8 - 1) / (256 * 8);
  if (sx > 256) sx = 256;
  hipLaunchKernelGGL(reverb_scale_kernel, dim3((unsigned)(sx < 1 ? 1 : sx), batch), dim3(256), 0, stream, a);
  return hipGetLastError();
}

// resample_mfma_kernel, when the ratio's fragments exist and a span fits: true if launched
static bool launch_resample_mfma(const ResampleArgs& a0, int batch, hipStream_t stream) {
  if (!a0.frag) return false;
  ResampleArgs a = a0;
  a.L = a0.L * a0.frag_group;                          // the kernel's group: frag_group * L outputs from frag_group * M inputs
  a.M = a0.M * a0.frag_group;
  // groups of L outputs per workgroup: 64, 32 or 16 -- the most whose LDS span leaves room for two workgroups
  // per CU (one stages while the other multiplies), else the most that fit at all
  const int span_pad = (a.M % 8 == 0 && !(a.tune & 64)) ? 4 : 0;   // resample_mfma_kernel<.., PAD> (WFE_TUNE & 64: without)
  auto span_of = [&](int gt) {
    const long long plain = (long long)(16 * gt - 1) * a.M + a.kspan;
    return plain + span_pad * ((plain - 1) / a.M) + 8;   // (+ 8: the last staged quad may reach into the next M)
  };
  int gt_pick = 0;
  for (int gt = 4; gt >= 1 && !gt_pick; gt >>= 1)
    if (span_of(gt) <= kRsMfmaSpanFloats / 2 + ((a.tune & 32) ? kRsMfmaSpanFloats : 0)) gt_pick = gt;
  for (int gt = 4; gt >= 1 && !gt_pick; gt >>= 1)
    if (span_of(gt) <= kRsMfmaSpanFloats) gt_pick = gt;
  if (gt_pick) {
    const int gt = gt_pick;
    const long long span = span_of(gt);
    const long long groups = (a.out_stride + a.L - 1) / a.L;
    const long long gx = (groups + 16 * gt - 1) / (16 * gt);
    const dim3 grid((unsigned)(gx < 1 ? 1 : gx), batch), block(64 * a.nwaves);
    const size_t lds = (size_t)((span + 3) / 4 * 4) * sizeof(float);
    using Kernel = void (*)(const ResampleArgs);
    const bool pad = span_pad != 0;
    const Kernel kernel = gt == 4 ? (pad ? resample_mfma_kernel<4, true> : resample_mfma_kernel<4, false>)
                        : gt == 2 ? (pad ? resample_mfma_kernel<2, true> : resample_mfma_kernel<2, false>)
                                  : (pad ? resample_mfma_kernel<1, true> : resample_mfma_kernel<1, false>);
    // dynamic LDS above 64 KB needs the attribute; it is a property of the kernel ON the current device, so it is
    // set per launch (a host-side table lookup) rather than remembered per process
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kRsMfmaSpanFloats * (int)sizeof(float)) != hipSuccess) {
      (void)hipGetLastError();                           // the fallback's launch check must not see this failure
      return false;
    }
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, a);
    return true;
  }
  return false;
}

hipError_t launch_resample(const ResampleArgs& a, int batch, hipStream_t stream) {
  // outputs S apart share a filter phase when L | S: S = the smallest multiple of L that fills 4 waves
  const int S = a.L >= 256 ? a.L : (256 + a.L - 1) / a.L * a.L;
  // first choice: the matrix-core kernel (WFE_TUNE & 16: without it, for A/B and to test the kernels below)
  if (!(a.tune & (4 | 16)) && launch_resample_mfma(a, batch, stream)) return hipGetLastError();
  if (a.L <= 4 && !(a.tune & 4) && (a.M == 1 || a.M == 2 || a.M == 3 || a.M == 4 || a.M == 6)) {
    // few phases: the sliding-window kernel, one grid layer per phase, when its span and taps fit
    const int W = 6 * a.M + 1, taps_p = (a.taps + W - 1) / W * W;
    if (taps_p <= 1024 && (long long)(256 * 7 - 1) * a.M + taps_p + W <= kRsSpanFloats) {
      const long long per_phase = (a.out_stride + a.L - 1) / a.L;
      const long long gx = (per_phase + 256 * 7 - 1) / (256 * 7);
      const dim3 grid((unsigned)(gx < 1 ? 1 : gx), batch, a.L);
      switch (a.M) {
        case 1: hipLaunchKernelGGL(resample_decimate_kernel<1>, grid, dim3(256), 0, stream, a); break;
        case 2: hipLaunchKernelGGL(resample_decimate_kernel<2>, grid, dim3(256), 0, stream, a); break;
        case 3: hipLaunchKernelGGL(resample_decimate_kernel<3>, grid, dim3(256), 0, stream, a); break;
        case 4: hipLaunchKernelGGL(resample_decimate_kernel<4>, grid, dim3(256), 0, stream, a); break;
        default: hipLaunchKernelGGL(resample_decimate_kernel<6>, grid, dim3(256), 0, stream, a); break;
      }
      return hipGetLastError();
    }
  }
  if (S <= 1024 && !(a.tune & 4)) {                   // (WFE_TUNE & 4: resample_kernel everywhere, for A/B)
    // U outputs per thread: as many (<= 8) as keep the workgroup's input span inside the LDS buffer
    int U = 8;
    while (U > 1 && (long long)S * U * a.M / a.L + a.taps + 2 > kRsSpanFloats) --U;
    if ((long long)S * U * a.M / a.L + a.taps + 2 <= kRsSpanFloats) {
      const long long opb = (long long)S * U;
      const long long gx = (a.out_stride + opb - 1) / opb;
      hipLaunchKernelGGL(resample_phase_kernel, dim3((unsigned)(gx < 1 ? 1 : gx), batch), dim3((S + 63) / 64 * 64), 0, stream, a,
                         S, U);
      return hipGetLastError();
    }
  }
  const long long gx = (a.out_stride + a.out_per_block - 1) / a.out_per_block;
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)(gx < 1 ? 1 : gx), batch), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace wfe
