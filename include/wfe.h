/*
 * wfe.h -- C ABI of the MI355X (gfx950) Whisper audio front end.
 *
 * The reference (cttailearn/Whisper-Finetune) has no FFI for this path: the
 * boundary there is two duck-typed Python objects.  These entry
 * points are what a binding for that boundary calls; each one names the
 * reference interface it stands in for.  "HF:" = transformers/models/whisper/
 * feature_extraction_whisper.py (third-party, called from utils/reader.py:192,
 * 195,198 and infer_tfs.py:37).
 *
 * Conventions: every function returns 0 on success and a negative wfe_status on
 * failure, never throws, never takes ownership of caller buffers, and is
 * stream-ordered (asynchronous) with respect to `stream` (a hipStream_t passed
 * as void*; NULL = the null stream).  Device pointers must belong to the
 * handle's device.  One host thread per handle at a time.  A handle owns small
 * device workspaces (utterance maxima, tile minima, resolved gains, filter
 * tables) that every call reuses: calls on ONE handle must be ordered -- issue
 * them on one stream, or order the streams with events; use one handle per
 * stream for concurrent launches.  Workspaces grow on demand (hipMalloc /
 * hipFree, which synchronise the device); after the first call at a given batch
 * shape the hot path allocates nothing.
 */
#ifndef WFE_H
#define WFE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WFE_N_FFT 400
#define WFE_HOP 160
#define WFE_N_FREQ 201
#define WFE_MAX_MEL 128

typedef enum {
  WFE_OK = 0,
  WFE_ERR_ARG = -1,       /* bad argument (null pointer, shape, unsupported size) */
  WFE_ERR_HIP = -2,       /* a HIP runtime call failed; see wfe_last_error */
  WFE_ERR_ALLOC = -3,     /* allocation failure */
  WFE_ERR_SAMPLING = -4   /* sampling rate is not the extractor's (HF:265-271 ValueError) */
} wfe_status;

typedef struct wfe_context* wfe_handle;

/* Mirrors WhisperFeatureExtractor.__init__ (HF:69-103): feature_size, sampling
 * rate, hop, n_fft.  Only n_fft=400 / hop=160 (every Whisper checkpoint) are
 * implemented by the kernels; anything else is WFE_ERR_ARG. */
typedef struct {
  int32_t n_mel;          /* feature_size: 1..128 (80 tiny..large-v2, 128 large-v3) */
  int32_t n_fft;          /* 400 */
  int32_t hop_length;     /* 160 */
  int32_t sampling_rate;  /* 16000 */
  int32_t device;         /* HIP device ordinal */
} wfe_config;

/* ---- augmentation (utils/reader.py:246-340) ------------------------------ */
/* One entry of a per-utterance op list.  The host draws the decisions in the
 * reference's order (utils/reader.py:246-278) and passes the results; the
 * kernels apply them while staging the waveform.  Ops are listed in
 * application order. */
typedef enum {
  WFE_AUG_SPEED = 1,   /* change_speed  utils/reader.py:281-292 */
  WFE_AUG_SHIFT = 2,   /* shift         utils/reader.py:295-307 */
  WFE_AUG_VOLUME = 3,  /* volume        utils/reader.py:310-313 */
  WFE_AUG_NOISE = 4,   /* add_noise     utils/reader.py:322-340 */
  WFE_AUG_AFFINE = 5   /* (x - mean) * rstd on the kept prefix: zero_mean_unit_var_norm HF:172-191 */
} wfe_aug_kind;

typedef struct {
  int32_t kind;        /* wfe_aug_kind */
  int32_t i0;          /* SPEED: new length int(old/rate).  SHIFT: shift in samples (signed,
                          int(ms*sr/1000)).  NOISE: noise clip index */
  int32_t i1;          /* NOISE: start frame inside the noise clip (0 unless the clip is longer
                          than the sample, utils/reader.py:336-338) */
  int32_t i2;          /* NOISE: snr in dB */
  float   f0;          /* VOLUME: the float32 multiplier float32(10.**(gain/20.)) */
  float   f1;          /* reserved, 0 */
  double  d0;          /* reserved, 0 */
} wfe_aug_op;          /* 32 bytes; at most one SPEED op per utterance */

/* ---- lifetime ------------------------------------------------------------ */
/* WhisperFeatureExtractor(feature_size=...) (HF:69-103): builds the Hann/DFT and
 * slaney mel tables in float64 on the host (audio_utils.py:638-729), rounds
 * once to float32 and uploads them. */
int wfe_create(const wfe_config* cfg, wfe_handle* out);
int wfe_destroy(wfe_handle h);
/* Text of the last failure on this handle (or of the last failed wfe_create
 * when h is NULL).  Valid until the next call on the same handle. */
const char* wfe_last_error(wfe_handle h);

/* self.mel_filters (HF:95-103): (201, n_mel) row-major float64, host memory. */
int wfe_mel_filters(wfe_handle h, double* host_out);

/* ---- hot path ------------------------------------------------------------ */
/* WhisperFeatureExtractor.__call__ minus the Python wrapping (HF:300-330):
 * truncate each utterance to n_samples, right-pad with zeros, reflect-padded
 * 400-point Hann STFT with hop 160 (last frame dropped), power, mel
 * projection, log10 floored at 1e-10, per-utterance max-8 clamp, (x+4)/4.
 *
 *   wav        device, float32.  Utterance b starts at wav + offsets[b]
 *              (offsets != NULL, device int64) or wav + b*row_stride.
 *   lengths    device int32[batch], samples present for each utterance (may
 *              exceed n_samples: truncated; may be 0).
 *   n_samples  padded length; 480000 for the default padding="max_length",
 *              the longest clip for padding="longest".  Must be > 200.
 *   out        device float32 (batch, n_mel, n_samples/160), fully overwritten.
 */
int wfe_logmel(wfe_handle h, const float* wav, const int64_t* offsets, const int32_t* lengths,
               int64_t row_stride, int32_t batch, int32_t n_samples, float* out, void* stream);

/* wfe_logmel for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for
 * callers that must not reach hipMalloc / hipFree -- code under HIP graph capture, tight
 * serving loops, harnesses that forbid allocation.  Same arguments, same kernels and bit for
 * bit the same output as wfe_logmel (it honours wfe_set_output_format too); the per-tile
 * statistics the first kernel leaves for the clamp kernel live in a static device array of the
 * library (1 732 608 bytes, held on every GPU the library runs on until the process ends) instead
 * of the handle's workspace, and the call itself allocates nothing, creates no event and is
 * never timed by wfe_kernel_timing.
 *   limit   batch * ceil((n_samples / 160) / 64) <= 12032 tiles (256 utterances of 30 s) and
 *           batch <= 4096; anything larger is WFE_ERR_ARG with the numbers in wfe_last_error.
 *   sharing every call on one GPU -- whatever the handle or stream -- uses that GPU's one array:
 *           calls that overlap on a GPU race; order them on one stream or with events.  The
 *           array keeps the last call's contents; nothing has to start at zero (kernel 1
 *           writes all 36 statistics of every tile before the clamp kernel reads them). */
int wfe_logmel_static(wfe_handle h, const float* wav, const int64_t* offsets, const int32_t* lengths,
                      int64_t row_stride, int32_t batch, int32_t n_samples, float* out, void* stream);

/* Same transform with the augmentation chain of CustomDataset.augment
 * (utils/reader.py:246-278) applied to each utterance while it is staged:
 * utterance b runs ops[op_begin[b] .. op_begin[b+1]).  `lengths` are the RAW
 * lengths; speed ops change the effective length before pad/trim.
 *   ops, op_begin   device arrays (op_begin has batch+1 entries; total_ops = op_begin[batch],
 *                   passed by the host so that no device read-back is needed)
 *   noise           device float32, all noise clips back to back
 *   noise_offsets   device int64[n_noise+1] prefix offsets into `noise`
 */
int wfe_augment_logmel(wfe_handle h, const float* wav, const int64_t* offsets, const int32_t* lengths,
                       int64_t row_stride, int32_t batch, int32_t n_samples,
                       const wfe_aug_op* ops, const int32_t* op_begin, int64_t total_ops,
                       const float* noise, const int64_t* noise_offsets, int32_t n_noise,
                       float* out, void* stream);

/* No reference counterpart (the reference decodes its noise file anew for every sample,
 * utils/reader.py:323).  The mean square of every clip of a noise bank -- the rms_db(noise) of
 * utils/reader.py:329 -- is computed on the first augmented call that sees the bank and kept while
 * later calls pass the same `noise` / `noise_offsets` pointers and `n_noise`.  A caller that
 * rewrites a bank IN PLACE (same pointers, new samples) calls this before the next augmented
 * call; a new allocation needs nothing.  The call also drops the spectra wfe_reverb keeps of
 * its impulse-response bank (same rule, same remedy). */
int wfe_noise_bank_changed(wfe_handle h);

/* The same two calls on 16-bit PCM as it lies in a WAV file, replacing the
 * decode-to-float32 and mono mix-down of CustomDataset._get_list_data
 * (utils/reader.py:144 soundfile.read(dtype='float32'), :148-150 librosa.to_mono):
 * sample = pcm / 32768 (exact), mono = float32 mean over the `channels`
 * interleaved values of a frame.  Halves the host-to-device bytes and removes the
 * host conversion pass.
 *   pcm        device int16, interleaved.  Utterance b starts at pcm + offsets[b]
 *              or pcm + b*row_stride (both counted in int16 ELEMENTS).
 *   lengths    FRAMES (samples per channel) of each utterance.
 *   channels   1..256, the same for the whole batch.
 * ops == NULL (then op_begin, noise are ignored) runs without augmentation. */
int wfe_logmel_pcm16(wfe_handle h, const int16_t* pcm, const int64_t* offsets, const int32_t* lengths,
                     int64_t row_stride, int32_t channels, int32_t batch, int32_t n_samples,
                     const wfe_aug_op* ops, const int32_t* op_begin, int64_t total_ops,
                     const float* noise, const int64_t* noise_offsets, int32_t n_noise,
                     float* out, void* stream);

/* CustomDataset.resample (utils/reader.py:316-319 = librosa.resample, used for the
 * resample augmentation :263-266 and the return to 16 kHz :157-158) for a batch:
 * out[b, n] = sum_k in_b[k] h(k - n in_rate/out_rate), zero-phase, zeros outside the
 * utterance.  PARITY UNPINNED: the reference's filter is soxr 'soxr_hq' (third-party,
 * not available); h is resampy's 'kaiser_best' design evaluated exactly
 * (whisper-finetune_amd/csrc/wfe_tables.cpp, oracle/resample.py).
 *   in           device float32 (pcm_channels == 0) or interleaved int16 PCM with
 *                pcm_channels channels (converted and mixed down as wfe_logmel_pcm16)
 *   lengths      device int32[batch], input frames
 *   out          device float32 (batch, out_stride); row b is zero behind out_lengths[b]
 *   out_lengths  device int32[batch], GIVEN by the caller: librosa's
 *                int(ceil(n * (float(out_rate) / in_rate)))
 * Ratios whose reduced numerator out_rate/gcd exceeds 4096, or that decimate by
 * more than ~100, are rejected with WFE_ERR_ARG. */
int wfe_resample(wfe_handle h, const void* in, int32_t pcm_channels, const int64_t* offsets,
                 const int32_t* lengths, int64_t row_stride, int32_t batch, int32_t in_rate, int32_t out_rate,
                 float* out, int64_t out_stride, const int32_t* out_lengths, void* stream);

/* CustomDataset.add_reverb (utils/reader.py:342-348) for a batch:
 * scipy.signal.fftconvolve(sample, rir, mode='same'), then y * (max|sample| / max|y|).
 * Utterances with rir_index[b] < 0 are copied through.  The convolution is a direct float32
 * sum while the bank's longest response is shorter than 512 taps, and a partitioned
 * overlap-save FFT convolution (4096-point float32 transforms) from there on; both agree with
 * the float64 convolution to 3e-6 max|y| sqrt(taps / 64).  The FFT form reads the bank's
 * lengths back ONCE per bank (a stream synchronisation on the first call that sees it) and
 * keeps the responses' spectra while later calls pass the same `rir` / `rir_offsets`
 * pointers and `n_rir`; a caller that rewrites a bank IN PLACE calls wfe_noise_bank_changed
 * (which forgets this cache too).  Workspace of the FFT form, kept by the handle:
 * about 0.26 MB per second of audio at 16 kHz and utterance of the batch from four partitions
 * of 2048 taps on (2 GB for 256 x 30 s with a 1 s response), less for shorter responses.
 *   wav / offsets / lengths / row_stride   as wfe_logmel (float32 only)
 *   rir, rir_offsets   device float32 impulse responses back to back and their
 *                      int64[n_rir + 1] prefix offsets (already at the utterances' rate:
 *                      librosa.load(rir_file, sr=sample_rate))
 *   rir_index          device int32[batch]
 *   out                device float32 (batch, out_stride), rows zero behind lengths[b] */
int wfe_reverb(wfe_handle h, const float* wav, const int64_t* offsets, const int32_t* lengths,
               int64_t row_stride, int32_t batch, const float* rir, const int64_t* rir_offsets, int32_t n_rir,
               const int32_t* rir_index, float* out, int64_t out_stride, void* stream);

/* The augmented, padded waveform itself, (batch, n_samples) float32: what
 * CustomDataset._get_list_data returns (utils/reader.py:134-162) after pad/trim.
 * Used by the parity tests of the augmentation stages. */
int wfe_augment_waveform(wfe_handle h, const float* wav, const int64_t* offsets, const int32_t* lengths,
                         int64_t row_stride, int32_t batch, int32_t n_samples,
                         const wfe_aug_op* ops, const int32_t* op_begin, int64_t total_ops,
                         const float* noise, const int64_t* noise_offsets, int32_t n_noise,
                         float* out, int32_t* out_lengths, void* stream);

/* Stage F4 alone: |STFT|^2 as (batch, 201, n_samples/160) float32 -- the
 * `magnitudes` tensor of HF:149-154.  For stage-level parity tests. */
int wfe_power_spectrum(wfe_handle h, const float* wav, const int64_t* offsets, const int32_t* lengths,
                       int64_t row_stride, int32_t batch, int32_t n_samples, float* out, void* stream);

/* ---- output element type ---------------------------------------------------- */
/* No reference counterpart: the reference's extractor returns float32 and the
 * Trainer casts input_features to the training dtype on the device (fp16=True,
 * finetune.py:122).  With WFE_OUT_F16 / WFE_OUT_BF16 the log-mel
 * entry points (wfe_logmel, wfe_augment_logmel, wfe_logmel_pcm16) write that
 * dtype directly -- `out` then points to batch * n_mel * (n_samples / 160) 2-byte
 * elements -- and the values are bit for bit the float32 result rounded to
 * nearest even (the clamp of kernel 2 commutes with the rounding).  The setting
 * belongs to the handle and stays until changed; wfe_power_spectrum and
 * wfe_augment_waveform are always float32. */
typedef enum { WFE_OUT_F32 = 0, WFE_OUT_F16 = 1, WFE_OUT_BF16 = 2 } wfe_out_format;
int wfe_set_output_format(wfe_handle h, int32_t format);

/* ---- measurement ---------------------------------------------------------- */
/* No reference counterpart (the reference has no timing code).
 * When enabled, every front-end launch is bracketed by HIP events on the
 * caller's stream; wfe_kernel_times waits for them, returns the summed device
 * time of the STFT+mel kernel and of the clamp kernel and the number of launches
 * since the last query, and resets the counters.  Used by bench.py's roofline. */
int wfe_kernel_timing(wfe_handle h, int32_t enable);
int wfe_kernel_times(wfe_handle h, double* frontend_ms, double* clamp_ms, int64_t* launches);

/* ---- host-only helpers (no GPU needed) ----------------------------------- */
/* The packed tables exactly as the kernels read them; exported so that CPU
 * tests can check the folded-DFT algebra and the packing without a device. */
int64_t wfe_dft_table_floats(void);
int64_t wfe_mel_table_floats(void);
int wfe_build_dft_table(float* host_out);
int wfe_build_mel_table(int32_t n_mel, float* host_out, uint32_t* mask14);
int wfe_build_mel_filters(int32_t n_mel, double* host_out /* (201, n_mel) */);
/* n for (parity class, k-step, k-lane) of the folded DFT. */
/* resampling filter: out_rate/in_rate = L/M reduced, taps per phase, index of the
 * centre tap; table [taps][L] float32 */
int wfe_resample_geometry(int32_t in_rate, int32_t out_rate, int32_t* L, int32_t* M, int32_t* taps, int32_t* k0);
int wfe_build_resample_table(int32_t in_rate, int32_t out_rate, float* out);
int32_t wfe_dft_n_of(int32_t parity_class, int32_t step, int32_t kk);

const char* wfe_version(void);

#ifdef __cplusplus
}
#endif
#endif /* WFE_H */
