// wfe_tables.cpp -- host-side (float64) construction of the tables the kernels
// read: the slaney mel bank of WhisperFeatureExtractor.__init__
// (transformers/models/whisper/feature_extraction_whisper.py:95-103 ->
// transformers/audio_utils.py:638-729), the periodic Hann window
// (torch.hann_window, same file :141) folded into the DFT twiddles, and their
// MFMA-fragment packing.  Pure host code: callable without a GPU.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/wfe.h"
#include "wfe_layout.h"

namespace {

const double kPi = 3.14159265358979323846;

// audio_utils.py:470-481, mel_scale="slaney"
double hz_to_mel(double f) {
  if (f >= 1000.0) return 15.0 + std::log(f / 1000.0) * (27.0 / std::log(6.4));
  return 3.0 * f / 200.0;
}
// audio_utils.py:506-517
double mel_to_hz(double m) {
  if (m >= 15.0) return 1000.0 * std::exp((std::log(6.4) / 27.0) * (m - 15.0));
  return 200.0 * m / 3.0;
}

// np.linspace(a, b, n)[i]: arange * step + a, last element pinned to b
double linspace_at(double a, double b, int n, int i) {
  if (i == n - 1) return b;
  const double step = (b - a) / (double)(n - 1);
  return (double)i * step + a;
}

}  // namespace

extern "C" int wfe_build_mel_filters(int32_t n_mel, double* out) {
  if (!out || n_mel < 1 || n_mel > wfe::kMaxMel) return WFE_ERR_ARG;
  const int nf = wfe::kNfreq;
  const double mel_lo = hz_to_mel(0.0), mel_hi = hz_to_mel(8000.0);
  std::vector<double> hz(n_mel + 2);
  for (int i = 0; i < n_mel + 2; ++i) hz[i] = mel_to_hz(linspace_at(mel_lo, mel_hi, n_mel + 2, i));
  for (int b = 0; b < nf; ++b) {
    const double f = linspace_at(0.0, 8000.0, nf, b);          // linspace(0, sr//2, 201)
    for (int m = 0; m < n_mel; ++m) {
      // audio_utils.py:541-560: min(down, up) clipped at 0
      const double down = -(hz[m] - f) / (hz[m + 1] - hz[m]);
      const double up = (hz[m + 2] - f) / (hz[m + 2] - hz[m + 1]);
      double v = down < up ? down : up;
      if (v < 0.0) v = 0.0;
      v *= 2.0 / (hz[m + 2] - hz[m]);                           // slaney energy norm :717-720
      out[(size_t)b * n_mel + m] = v;
    }
  }
  return WFE_OK;
}

extern "C" int64_t wfe_dft_table_floats(void) { return wfe::kDftTableFloats; }
extern "C" int64_t wfe_mel_table_floats(void) { return wfe::kMelTableFloats; }
extern "C" int32_t wfe_dft_n_of(int32_t c, int32_t s, int32_t kk) { return wfe::dft_n_of(c, s, kk); }

// [step][gemm][half][lane][j]: A operand of v_mfma_f32_16x16x4_f32 -- lane l
// holds A[row = l & 15][k = l >> 4]; row = bin inside bin tile (4*half + j),
// k = the n assigned to k-lane l >> 4 of this step.
extern "C" int wfe_build_dft_table(float* out) {
  if (!out) return WFE_ERR_ARG;
  using namespace wfe;
  for (int s = 0; s < kSteps; ++s)
    for (int g = 0; g < kGemms; ++g)
      for (int half = 0; half < 2; ++half)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 4; ++j) {
            const int bt = 4 * half + j;
            const int k = 16 * bt + (lane & 15);
            const int n = dft_n_of(g >> 1, s, lane >> 4);
            double v = 0.0;
            if (bt < kBinTiles && k < kFoldBins) {
              const double w = 0.5 - 0.5 * std::cos(2.0 * kPi * (double)n / (double)kNfft);
              const int r = (k * n) % kNfft;                    // exact angle reduction
              const double ang = 2.0 * kPi * (double)r / (double)kNfft;
              v = w * ((g & 1) ? std::sin(ang) : std::cos(ang));
              if (n == kHalf) v = (g & 1) ? 0.0 : 0.5 * v;      // e[200] = 2 x[200]; o[200] = 0
            }
            out[((((size_t)s * kGemms + g) * 2 + half) * 64 + lane) * 4 + j] = (float)v;
          }
  return WFE_OK;
}

// [family][bin tile][mel tile][lane][r]: A operand of the mel MFMA -- lane l
// holds A[row = mel 16*mt + (l & 15)][k = l >> 4] for the power register r, whose
// lanes carry folded bin k = 16*bt + 4*(l >> 4) + r (the C/D layout of the DFT
// MFMA).  family 0 is spectrum bin k, family 1 is bin 200 - k.
extern "C" int wfe_build_mel_table(int32_t n_mel, float* out, uint32_t* mask14) {
  if (!out || !mask14 || n_mel < 1 || n_mel > wfe::kMaxMel) return WFE_ERR_ARG;
  using namespace wfe;
  std::vector<double> bank((size_t)kNfreq * n_mel);
  wfe_build_mel_filters(n_mel, bank.data());
  for (int i = 0; i < 2 * kBinTiles; ++i) mask14[i] = 0;
  for (int f = 0; f < 2; ++f)
    for (int bt = 0; bt < kBinTiles; ++bt)
      for (int mt = 0; mt < kMelTiles; ++mt)
        for (int lane = 0; lane < 64; ++lane)
          for (int r = 0; r < 4; ++r) {
            const int mel = 16 * mt + (lane & 15);
            const int k = 16 * bt + 4 * (lane >> 4) + r;
            float v = 0.f;
            const bool live = mel < n_mel && k < kFoldBins && !(f == 1 && k == kFoldBins - 1);
            if (live) v = (float)bank[(size_t)(f ? 200 - k : k) * n_mel + mel];
            if (v != 0.f) mask14[f * kBinTiles + bt] |= 1u << mt;
            out[((((size_t)f * kBinTiles + bt) * kMelTiles + mt) * 64 + lane) * 4 + r] = v;
          }
  return WFE_OK;
}
