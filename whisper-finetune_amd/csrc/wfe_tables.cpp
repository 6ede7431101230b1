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

// One entry of the folded-DFT matrices: Hann window times cos / sin of bin row k at sample n.
// k < 0 (unused row) and n > 200 (dummy k-lane of a class's last even-step) give 0.
namespace {
double twiddle(int k, int n, bool sine) {
  using namespace wfe;
  if (k < 0 || n > kHalf) return 0.0;
  const double w = 0.5 - 0.5 * std::cos(2.0 * kPi * (double)n / (double)kNfft);
  const int r = (k * n) % kNfft;                    // exact angle reduction
  const double ang = 2.0 * kPi * (double)r / (double)kNfft;
  double v = w * (sine ? std::sin(ang) : std::cos(ang));
  if (n == kHalf) v = sine ? 0.0 : 0.5 * v;         // e[200] = 2 x[200]; o[200] = 0
  return v;
}
}  // namespace

// [step][quad][lane][j] (wfe_layout.h): A operand of v_mfma_f32_16x16x4_f32 -- lane l holds
// A[row = l & 15][k-lane = l >> 4] of tile j of the quad; the row's bin frequency comes from
// odd_row_of / even_row_of, the k-lane's n from dft_n_of.
extern "C" int wfe_build_dft_table(float* out) {
  if (!out) return WFE_ERR_ARG;
  using namespace wfe;
  for (int s = 0; s < kSteps; ++s)
    for (int q = 0; q < kDftQuads; ++q)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 4; ++j) {
          const int row = lane & 15, kk = lane >> 4;
          double v = 0.0;
          if (q < 4) {                                         // odd class: Co (q 0, 1), So (q 2, 3)
            const int ti = 4 * (q & 1) + j;
            if (ti < kOddTiles) v = twiddle(odd_row_of(ti, row), dft_n_of(1, s, kk), q >= 2);
          } else if (q < 6) {                                  // even class, even-step s
            v = twiddle(even_row_of(j, row), dft_n_of(0, s, kk), q == 5);
          } else if (s == kSteps - 1) {                        // even-step 25 rides in the last k-step
            v = twiddle(even_row_of(j, row), dft_n_of(0, kEvenSteps - 1, kk), q == 7);
          }
          out[(((size_t)s * kDftQuads + q) * 64 + lane) * 4 + j] = (float)v;
        }
  return WFE_OK;
}

// [power tile][mel tile][lane][r]: A operand of the mel MFMA -- lane l holds
// A[row = mel 16*mt + (l & 15)][k = l >> 4] for the power register r, whose lanes carry the spectrum bin
// power_bin_of(T, 4 * (l >> 4) + r) (the C/D layout of the DFT MFMA).
extern "C" int wfe_build_mel_table(int32_t n_mel, float* out, uint32_t* mask14) {
  if (!out || !mask14 || n_mel < 1 || n_mel > wfe::kMaxMel) return WFE_ERR_ARG;
  using namespace wfe;
  std::vector<double> bank((size_t)kNfreq * n_mel);
  wfe_build_mel_filters(n_mel, bank.data());
  for (int i = 0; i < kPowerTiles; ++i) mask14[i] = 0;
  for (int T = 0; T < kPowerTiles; ++T)
    for (int mt = 0; mt < kMelTiles; ++mt)
      for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
          const int mel = 16 * mt + (lane & 15);
          const int bin = power_bin_of(T, 4 * (lane >> 4) + r);
          float v = 0.f;
          if (mel < n_mel && bin >= 0) v = (float)bank[(size_t)bin * n_mel + mel];
          if (v != 0.f) mask14[T] |= 1u << mt;
          out[(((size_t)T * kMelTiles + mt) * 64 + lane) * 4 + r] = v;
        }
  return WFE_OK;
}

// ---------------------------------------------------------------- resampling
// The reference resamples with librosa.resample (utils/reader.py:316-319,
// res_type 'soxr_hq'): soxr is a third-party native library that is not
// available to this build, so bit parity cannot be pinned.  The filter here is
// librosa's other high-quality design, resampy's 'kaiser_best' -- a Kaiser
// (beta 14.7697) windowed sinc with 64 zero crossings per side and roll-off
// 0.94759 of the lower Nyquist -- evaluated exactly at every rational phase
// instead of through resampy's linearly interpolated 512-per-crossing table:
//   h(tau) = s r sinc(r s tau) I0(beta sqrt(1 - (s tau / 64)^2)) / I0(beta),   |s tau| < 64,
// tau in input samples, s = min(1, out_rate / in_rate).  Output n sits at input
// position n M / L (L / M = out_rate / in_rate reduced), zero-phase, zeros outside.
namespace {

const int kRsZeros = 64;
const double kRsRolloff = 0.9475937167399596;
const double kRsBeta = 14.769656459379492;

long long gcd_ll(long long a, long long b) {
  while (b) {
    const long long t = a % b;
    a = b;
    b = t;
  }
  return a;
}

double bessel_i0(double x) {      // power series; x <= 15 here
  double sum = 1.0, term = 1.0;
  const double q = x * x / 4.0;
  for (int k = 1; k < 200; ++k) {
    term *= q / ((double)k * (double)k);
    sum += term;
    if (term < 1e-18 * sum) break;
  }
  return sum;
}

double rs_filter(double tau, double scale) {
  const double x = scale * tau;
  if (std::fabs(x) >= (double)kRsZeros) return 0.0;
  const double a = kPi * kRsRolloff * x;
  const double sinc = (a == 0.0) ? 1.0 : std::sin(a) / a;
  const double t = x / (double)kRsZeros;
  return scale * kRsRolloff * sinc * bessel_i0(kRsBeta * std::sqrt(1.0 - t * t)) / bessel_i0(kRsBeta);
}

}  // namespace

extern "C" int wfe_resample_geometry(int32_t in_rate, int32_t out_rate, int32_t* L, int32_t* M, int32_t* taps,
                                     int32_t* k0) {
  if (in_rate < 1 || out_rate < 1 || !L || !M || !taps || !k0) return WFE_ERR_ARG;
  const long long g = gcd_ll(in_rate, out_rate);
  *L = (int32_t)(out_rate / g);
  *M = (int32_t)(in_rate / g);
  const double scale = out_rate < in_rate ? (double)out_rate / (double)in_rate : 1.0;
  const double hw = (double)kRsZeros / scale;
  if (hw > 1e6) return WFE_ERR_ARG;
  *k0 = (int32_t)std::floor(hw);
  *taps = 2 * *k0 + 2;
  return WFE_OK;
}

// out[j * L + p] = h((j - k0) - p / L): the weight of input sample base - k0 + j for an
// output whose position is base + p / L.  Row-major over taps so that the 64 phases a
// wave needs for one tap lie in one short run.
extern "C" int wfe_build_resample_table(int32_t in_rate, int32_t out_rate, float* out) {
  int32_t L, M, taps, k0;
  if (!out || wfe_resample_geometry(in_rate, out_rate, &L, &M, &taps, &k0) != WFE_OK) return WFE_ERR_ARG;
  const double scale = out_rate < in_rate ? (double)out_rate / (double)in_rate : 1.0;
  for (int j = 0; j < taps; ++j)
    for (int p = 0; p < L; ++p)
      out[(size_t)j * L + p] = (float)rs_filter((double)(j - k0) - (double)p / (double)L, scale);
  return WFE_OK;
}
