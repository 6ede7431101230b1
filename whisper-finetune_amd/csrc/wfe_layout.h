// wfe_layout.h -- constants and index maps shared by the host table builders
// and the gfx950 kernels.  Everything here is plain integer arithmetic so that
// host and device agree by construction.
//
// Folded DFT (DESIGN.md "Kernel 1"): a 400-point real DFT of a Hann-windowed
// frame x[0..399] is evaluated as six small GEMMs.
//   e[n] = x[n] + x[400-n],  o[n] = x[n] - x[400-n]          (n = 1..200)
//   Re X[k] =  sum_n w[n] e[n] cos(2 pi k n / 400)             (w[0] = 0, w symmetric)
//   Im X[k] = -sum_n w[n] o[n] sin(2 pi k n / 400)
// Level B: splitting n by parity (C = cos sums over e, S = sin sums over o), for k = 0..100:
//   Re X[k] = Ce[k] + Co[k]     Re X[200-k] = Ce[k] - Co[k]
//   |Im X[k]| = |Se[k] + So[k]| |Im X[200-k]| = |Se[k] - So[k]|
// Level C (round 3): the even class n = 2m folds once more, because its twiddles at bin 100 - k are
// (-1)^m times those at bin k.  With ee = {n = 0 mod 4}, eo = {n = 2 mod 4}:
//   Ce[k] = Cee[k] + Ceo[k]     Ce[100-k] = Cee[k] - Ceo[k]
//   Se[k] = See[k] + Seo[k]     Se[100-k] = Seo[k] - See[k]
// so the even class needs rows k = 0..52 only (K = 50 per sub-class) instead of 0..100 (K = 100): half its
// MFMAs.  (The odd class has no such fold: under k -> 100 - k its cos and sin twiddles swap.)
// GEMMs, all on v_mfma_f32_16x16x4_f32:
//   Co, So            odd n = 1..199       rows k = 0..100 in 7 tiles (below)   25 k-steps
//   Cee, See | Ceo, Seo   n = 4..200 | 2..198   rows k = 0..52 in 4 tiles       13 + 13 k-steps ("even-steps")
// Row order = tile POSITIONS: position p = 16 tile + row.
//   U tiles 0..3   p = 0..63 holds bin row k = p (p <= 52), giving spectrum bins p and 200 - p
//   V tiles 0..2   p = 0..47 holds bin row k = 100 - p,      giving spectrum bins 100 - p and 100 + p
// V's even-class terms come from the SAME position of the U tiles (Ce[100-p] = Cee[p] - Ceo[p]), so every
// butterfly pairs accumulators of one lane and register.  Bins 0..52 | 148..200 | 53..100 | 101..147.
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

constexpr int kSteps = 25;          // k-steps of 4 of the odd class (K = 100)
constexpr int kEvenSteps = 26;      // even-steps: 13 of class ee (n = 0 mod 4), then 13 of class eo (n = 2 mod 4)
constexpr int kOddTiles = 7;        // U0..U3, V0..V2
constexpr int kEvenTiles = 4;       // U0..U3
constexpr int kUTiles = 4, kVTiles = 3;
constexpr int kURows = 53;          // U positions 0..52 hold bin rows 0..52
constexpr int kVRows = 48;          // V positions 0..47 hold bin rows 100..53
constexpr int kPowerTiles = 14;     // power tiles: U lo (bins p) 0..3, U hi (200 - p) 4..7, V a (100 - p) 8..10, V b (100 + p) 11..13

// DFT table: [step(25)][quad(8)][lane(64)][4].  Quads of step s: 0 Co U0..3 | 1 Co V0..2 + pad | 2 So U0..3 |
// 3 So V0..2 + pad | 4 even cos U0..3 of even-step s | 5 even sin of even-step s | 6, 7 even cos / sin of
// even-step 25 (step 24 only; zero elsewhere).  One k-step = 8 KB, shared by the 4 waves through an LDS slot.
constexpr int kDftQuads = 8;
constexpr int kDftStepFloats = kDftQuads * 64 * 4;              // 2048
constexpr int64_t kDftTableFloats = (int64_t)kSteps * kDftStepFloats;

// Mel table: [power tile(14)][mel tile(8)][lane(64)][reg(4)]
constexpr int kMelTiles = 8;
constexpr int64_t kMelTableFloats = (int64_t)kPowerTiles * kMelTiles * 64 * 4;

// spectrum bin of position p (0..15 within the tile: 4 (lane >> 4) + reg) of power tile T, or -1
WFE_HD int power_bin_of(int T, int row) {
  if (T < 4) { const int p = 16 * T + row; return p < kURows ? p : -1; }
  if (T < 8) { const int p = 16 * (T - 4) + row; return p < kURows ? 200 - p : -1; }
  if (T < 11) { const int p = 16 * (T - 8) + row; return 100 - p; }
  const int p = 16 * (T - 11) + row;
  return p >= 1 ? 100 + p : -1;                                  // (p = 0 would repeat bin 100)
}
// bin row k (the twiddle frequency) of odd-class tile ti (0..6) / even-class tile (0..3), row 0..15; -1: unused
WFE_HD int odd_row_of(int ti, int row) {
  if (ti < kUTiles) { const int p = 16 * ti + row; return p < kURows ? p : -1; }
  return 100 - (16 * (ti - kUTiles) + row);
}
WFE_HD int even_row_of(int ti, int row) { const int p = 16 * ti + row; return p < kURows ? p : -1; }

// Non-zero (power tile, mel tile) blocks of the two standard Whisper banks, bit mt
// of entry power tile (what wfe_build_mel_table returns in mask14).
// Known at compile time so that the mel phase of the kernel is straight-line code
// whose weight loads the compiler can issue ahead of the MFMAs; any other
// feature_size takes the run-time-mask build.  wfe_create checks them against the
// masks it computes.
constexpr unsigned kMelMask80[14] = {0x3, 0x6, 0x4, 0x4, 0x10, 0x10, 0x10, 0x10, 0x8, 0x8, 0xc, 0x18, 0x10, 0x10};     // 18 blocks
constexpr unsigned kMelMask128[14] = {0x3, 0xe, 0x18, 0x10, 0x80, 0x80, 0x80, 0x80, 0x60, 0x20, 0x30, 0x40, 0x40, 0xc0};   // 21 blocks

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

// n handled by k-lane kk (0..3) of step s for parity class c: c = 1, the odd n, k-steps 0..24; c = 0, the even
// n, EVEN-steps 0..25 (0..12 class ee, 13..25 class eo).  Within lanes 0..31 (kk 0,1) and 32..63 (kk 2,3) the
// two n differ by 16, so the two 16-frame groups of a ds_read_b32 lane group sit on disjoint bank halves.
// Odd step 24 takes the four left-over n (193..199).  The last even-step of each class holds its two
// left-over n (196, 200 | 194, 198) on k-lanes 0 and 2; its k-lanes 1 and 3 name n > 200: DUMMIES whose table
// entries are zero (the samples they read exist, the products vanish).
WFE_HD int dft_n_of(int c, int s, int kk) {
  if (c) {
    if (s < 24) return 32 * (s >> 2) + 4 * (s & 3) + 2 * (kk >> 1) + 16 * (kk & 1) + 1;
    return 192 + 2 * kk + 1;
  }
  const int c0 = s < 13 ? 4 : 2, u = s < 13 ? s : s - 13;
  if (u < 12) return 32 * (u >> 1) + 8 * (u & 1) + 4 * (kk >> 1) + 16 * (kk & 1) + c0;
  return 192 + 4 * (kk >> 1) + 16 * (kk & 1) + c0;
}

}  // namespace wfe
