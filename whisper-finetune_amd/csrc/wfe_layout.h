// wfe_layout.h -- constants and index maps shared by the host table builders
// and the gfx950 kernels.  Everything here is plain integer arithmetic so that
// host and device agree by construction.
//
// Folded DFT (DESIGN.md "Kernel 1"): a 400-point real DFT of a Hann-windowed
// frame x[0..399] is evaluated as four small GEMMs.
//   e[n] = x[n] + x[400-n],  o[n] = x[n] - x[400-n]          (n = 1..200)
//   Re X[k] =  sum_n w[n] e[n] cos(2 pi k n / 400)             (w[0] = 0, w symmetric)
//   Im X[k] = -sum_n w[n] o[n] sin(2 pi k n / 400)
// and, splitting n by parity, for k = 0..100:
//   Re X[k] = Ce[k] + Co[k]     Re X[200-k] = Ce[k] - Co[k]
//   |Im X[k]| = |Se[k] + So[k]| |Im X[200-k]| = |Se[k] - So[k]|
// Each of Ce, Se (even n = 2..200) and Co, So (odd n = 1..199) is a
// (frames x 100) . (100 x 101) product, run on v_mfma_f32_16x16x4_f32.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define WFE_HD __host__ __device__ inline
#else
#define WFE_HD inline
#endif

namespace wfe {

constexpr int kNfft = 400;
constexpr int kHop = 160;
constexpr int kHalf = 200;          // reflect pad on each side
constexpr int kNfreq = 201;
constexpr int kMaxMel = 128;

constexpr int kSteps = 25;          // k-steps of 4 per GEMM (K = 100)
constexpr int kGemms = 4;           // 0 Ce (even,cos) 1 Se (even,sin) 2 Co (odd,cos) 3 So (odd,sin)
constexpr int kBinTiles = 7;        // 7 x 16 = 112 >= 101 half-spectrum bins
constexpr int kBinTilesPad = 8;     // table stores 8 so that a lane fetches 2 x float4
constexpr int kFoldBins = 101;      // k = 0..100

// DFT table: [step][gemm][half(2)][lane(64)][4]  (half h, element j -> bin tile 4h + j)
constexpr int kDftStepFloats = kGemms * 2 * 64 * 4;             // 2048
constexpr int64_t kDftTableFloats = (int64_t)kSteps * kDftStepFloats;

// Mel table: [family(2)][bin tile(7)][mel tile(8)][lane(64)][reg(4)]
constexpr int kMelTiles = 8;
constexpr int64_t kMelTableFloats = 2LL * kBinTiles * kMelTiles * 64 * 4;

// Non-zero (bin tile, mel tile) blocks of the two standard Whisper banks, bit mt
// of entry family*7 + bin tile (what wfe_build_mel_table returns in mask14).
// Known at compile time so that the mel phase of the kernel is straight-line code
// whose weight loads the compiler can issue ahead of the MFMAs; any other
// feature_size takes the run-time-mask build.  wfe_create checks them against the
// masks it computes.
constexpr unsigned kMelMask80[14] = {0x3, 0x6, 0x4, 0xc, 0x8, 0x8, 0x8, 0x10, 0x10, 0x10, 0x10, 0x10, 0x18, 0x8};
constexpr unsigned kMelMask128[14] = {0x3, 0xe, 0x18, 0x30, 0x20, 0x60, 0x40,
                                      0x80, 0x80, 0x80, 0x80, 0xc0, 0x40, 0x40};

// Frames per workgroup: 4 waves x 1 frame tile x 16 frames.  Two workgroups are
// resident per CU (2 waves per SIMD, <= 256 registers each), so one workgroup's
// staging / epilogue overlaps the other's MFMA stream.
constexpr int kWaves = 4;
constexpr int kBlockFrames = kWaves * 16;                           // 64
constexpr int kBlockSamples = (kBlockFrames - 1) * kHop + kNfft;   // 10480 padded-signal positions

// LDS image of the waveform tile: position q -> q + q/160.  The extra word per
// hop makes the 16 frames of an MFMA B operand (stride 160 = 0 mod 32 banks)
// land on 16 consecutive banks.
WFE_HD int pad_q(int q) { return q + q / kHop; }
constexpr int kLdsWaveFloats = ((kBlockSamples - 1 + (kBlockSamples - 1) / kHop + 1) + 3) / 4 * 4;   // 10548
// The output tile [mel][64 frames] that follows it in LDS: 4 extra words per row
// keep rows 16-byte aligned and put the 4 mel rows of one MFMA register on
// disjoint bank halves.
constexpr int kTileStride = kBlockFrames + 4;

// n handled by k-lane kk (0..3) of k-step s (0..24) for parity class c
// (0: even n, 1: odd n).  Within lanes 0..31 (kk 0,1) and 32..63 (kk 2,3) the two
// n differ by 16, so the two 16-frame groups of a ds_read_b32 lane group sit on
// disjoint bank halves.  Step 24 takes the four left-over n (193..200).
WFE_HD int dft_n_of(int c, int s, int kk) {
  const int c0 = c ? 1 : 2;
  if (s < 24) return 32 * (s >> 2) + 4 * (s & 3) + 2 * (kk >> 1) + 16 * (kk & 1) + c0;
  return 192 + 2 * kk + c0;
}

}  // namespace wfe
