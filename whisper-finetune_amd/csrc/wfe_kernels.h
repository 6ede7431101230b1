// wfe_kernels.h -- launch interface between the C ABI (wfe_capi.cpp) and the
// gfx950 kernels (wfe_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/wfe.h"

namespace wfe {

// Gains of one NOISE op resolved on the device (utils/reader.py:325-331), or
// mean / std of an AFFINE op.
struct Resolved {
  float a;
  float b;
};

constexpr int kTileStats = 36;   // floats kernel 1 leaves per 64-frame tile for kernel 2: [wave 4][mel tile 8] minima, [wave 4] maxima

// What the resolve pass (resolve_utterance) leaves per utterance for kernel 1: the chain's geometry, and (in `cops`)
// its ops compiled with their resolved scalars -- kernel 1 then needs two load round trips to
// set a chain up instead of walking the op list (five dependent ones).
struct ChainHead {
  int n_ops, speed_at, len1, compiled;   // compiled: cops hold the chain (n_ops <= 16)
  double step;
  // The chain shape the reference's own configuration draws ([NOISE]? [SPEED]? [SHIFT]? [VOLUME]?), matched
  // once per utterance: everything kernel 1's contiguous staging needs, in one uniform 72-byte load instead
  // of the op array copied through LDS behind a barrier (absent SHIFT / VOLUME are shift 0 / gain 1.0f).
  int shipped;                           // 1: the fields below describe the whole chain
  int noise_k;                           // index of the NOISE op or -1
  int shift;
  float gain;
  int noise_len, noise_start;            // the NOISE op's clip: length, start of the slice when longer than the utterance
  long long noise_o0;                    // its offset in the bank
  float ra, rb;                          // resolved gains
  double noise_inv;                      // 1 / noise_len
};
constexpr int kCopBytes = 48;            // sizeof(COp), wfe_kernels.hip

struct FrontArgs {
  // input
  const float* wav;              // float32 samples, or
  const short* pcm;              // interleaved 16-bit PCM (then wav is null); offsets / row_stride in int16 elements
  int channels;                  // PCM only: channels per frame, mixed down to mono (np.mean)
  const long long* offsets;      // optional
  const int* lengths;
  long long row_stride;
  int n_samples;                 // padded length (reflect padding applies to this length)
  int n_frames;                  // n_samples / 160
  int n_mel;
  // tables
  const float* dft_tab;
  const float* mel_tab;
  unsigned mel_mask[14];
  // output
  float* out;                    // (B, n_mel, T) log10 mel power, or (B, 201, T) power
  int out_kind;                  // log-mel element type: 0 float32, 1 float16, 2 bfloat16 (out then points to 2-byte elements)
  float* tile_stats;             // (B x 64-frame tiles x kTileStats): minima [wave 4][mel tile 8], maxima [wave 4], in the stored scale
  // augmentation (AUG kernels only)
  const wfe_aug_op* ops;
  const int* op_begin;
  const float* noise;
  const long long* noise_off;
  const double* noise_meansq;    // (n_noise x kNoiseParts) partial sums of squares of each noise clip
  Resolved* resolved;            // (total ops) filled by the reduce kernel
  void* cops;                    // (total ops x kCopBytes) compiled ops, written by the resolve pass
  ChainHead* heads;              // (batch) written by the resolve pass
  unsigned long long* stamps;    // diagnostic builds (-DWFE_STAMPS) only: 16 clock stamps per wave
  int tune;                      // experiment switches (WFE_TUNE, read at wfe_create); 0 in production
};

// mode 0: log10 mel + clip max.  mode 1: power spectrum.
hipError_t launch_frontend(const FrontArgs& a, int batch, int mode, bool augmented, int bank, hipStream_t stream);
// max(stored, g(max-8)) over the tiles whose minimum is below max-8 (kernel 1 stored g(v) = (v+4)/4)
hipError_t launch_clamp(float* out, int out_kind, const float* tile_stats, int batch, int n_mel, int n_frames,
                        hipStream_t stream);
// partial sums of squares of the noise clips: meansq has n_noise x kNoiseParts doubles
constexpr int kNoiseParts = 16;
hipError_t launch_noise_meansq(const float* noise, const long long* noise_off, int n_noise, double* meansq,
                               hipStream_t stream);
// resolve the data-dependent scalars of every utterance's chain (NOISE gains, AFFINE stats)
// partials: workspace of batch x kResolveParts x 2 doubles; tickets: batch counters, zero before the FIRST
// launch (every launch leaves them zero again)
constexpr int kResolveParts = 16;
hipError_t launch_resolve(const FrontArgs& a, int batch, double* partials, unsigned* tickets, hipStream_t stream);
// augmented + padded waveform (batch, n_samples) and final lengths
hipError_t launch_waveform(const FrontArgs& a, int batch, float* out, int* out_lengths, hipStream_t stream);

// ---- resampling (CustomDataset.resample, utils/reader.py:316-319)
constexpr int kRsSpanFloats = 14336;   // LDS floats for the input span of one workgroup
constexpr int kRsChunkFloats = 4096;   // LDS floats for one chunk of filter rows (so L <= 4096)
constexpr int kRsMfmaSpanFloats = 36864;   // LDS floats (144 KB) for the span of resample_mfma_kernel
struct ResampleArgs {
  const float* wav;              // float32 input, or
  const short* pcm;              // interleaved 16-bit PCM
  int channels;
  const long long* offsets;      // optional
  const int* lengths;            // input frames per utterance
  long long row_stride;
  float* out;                    // (batch, out_stride) float32, rows zero-filled behind out_lengths[b]
  long long out_stride;
  const int* out_lengths;        // output samples per utterance (librosa: ceil(n * out_rate / in_rate))
  const float* table;            // [taps][L] filter, wfe_build_resample_table
  int L, M, taps, k0;
  int out_per_block;             // outputs per workgroup: a multiple of 256, <= 2048, span fits kRsSpanFloats
  int tune;                      // experiment switches (WFE_TUNE)
  // the filter as the A operand of resample_mfma_kernel (wfe_capi.cpp: build_resample_fragments); null: not built
  const float* frag;             // per tile of 16 phases: [quads of k-steps][64 lanes][4] floats
  const int* frag_meta;          // per tile: first k-step, quads of k-steps, offset into frag (in 64 x float4)
  int ptiles, kspan, nwaves;     // tiles of 16 phases; LDS columns the widest tile reaches; waves per workgroup
  int frag_group;                // the fragments describe groups of frag_group * L outputs (frag_group * M inputs)
};
hipError_t launch_resample(const ResampleArgs& a, int batch, hipStream_t stream);

// ---- reverberation (CustomDataset.add_reverb, utils/reader.py:342-348)
struct ReverbArgs {
  const float* wav;              // float32 utterances
  const long long* offsets;      // optional
  const int* lengths;
  long long row_stride;
  const float* rir;              // all impulse responses back to back
  const long long* rir_off;      // (n_rir + 1) prefix offsets
  const int* rir_index;          // per utterance; < 0: copied through unchanged
  float* out;                    // (batch, out_stride), rows zero behind lengths[b]
  long long out_stride;
  unsigned* peaks;               // (batch x 2) max |x|, max |y| as float bits (non-negative floats order as uints)
  int tune;                      // experiment switches (WFE_TUNE)
};
hipError_t launch_reverb(const ReverbArgs& a, int batch, hipStream_t stream);
hipError_t launch_reverb_scale(const ReverbArgs& a, int batch, hipStream_t stream);   // (the second pass of launch_reverb alone)

// ---- the same for long impulse responses: partitioned overlap-save FFT convolution (wfe_reverb_fft.hip)
constexpr int kRvFftN = 4096;          // transform size; partitions and output blocks of N / 2
#ifndef WFE_RV_FFT_MIN_TAPS
#define WFE_RV_FFT_MIN_TAPS 512
#endif
constexpr int kRvFftMinTaps = WFE_RV_FFT_MIN_TAPS;   // banks whose longest response has at least this many taps take the FFT path
// tw: 128 float2 = exp(-2 pi i 64 a / N), a < 64, then exp(-2 pi i b / N), b < 64 (built in float64 on the host)
// H: (n_rir x p_max x N) spectra of the response partitions / N
hipError_t launch_reverb_spectra(const float* rir, const long long* rir_off, int n_rir, int p_max, const float2* tw, float2* H,
                                 hipStream_t stream);
// One kernel (responses of one or two partitions).  ring: (batch x segments x p_max x N) float2 workspace; a.out rows
// must be zero behind their lengths already
hipError_t launch_reverb_fft(const ReverbArgs& a, int batch, const float2* tw, const float2* H, int p_max, float2* ring,
                             int segments, int seg_blocks, hipStream_t stream);
// Three kernels (forward transforms -> products by the owner of a bin -> inverse transforms), for longer responses.
// blocks_cap = ceil(out_stride / (N / 2)) + 1, kh_cap = ceil(blocks_cap / 2), pairs_cap = kh_cap + p_max - 1;
// Z: (batch x pairs_cap x N), Y: (batch x kh_cap x N) float2 workspaces.  The products' kernel needs
// reverb_mac_lds_bytes(p_max) of LDS: at most kRvMacLdsMax.
constexpr int kRvMacLdsMax = 152 * 1024;
int reverb_mac_lds_bytes(int p_max);
hipError_t launch_reverb_fft3(const ReverbArgs& a, int batch, const float2* tw, const float2* H, int p_max, float2* Z,
                              int pairs_cap, float2* Y, int kh_cap, hipStream_t stream);

}  // namespace wfe
