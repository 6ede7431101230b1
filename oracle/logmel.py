"""CPU restatement of the Whisper log-mel transform (TEST INFRASTRUCTURE ONLY).

The arithmetic lives in the third-party ``transformers`` package (installed
5.15.0; the reference pins ``transformers>=4.31.0`` in requirements.txt:5) and
is reached from the reference at utils/reader.py:192,195,198 and
infer_tfs.py:37.  ``HF:`` below = transformers/models/whisper/
feature_extraction_whisper.py, ``AU:`` = transformers/audio_utils.py.

Two statements of the same maths are kept:
  * ``logmel_f64``   explicit framing + float64 rFFT (follows the numpy path,
                     HF:105-133 / AU:809-1017) -- the "truth" both fp32
                     implementations are compared with;
  * ``logmel_f32``   float32 torch.stft path (follows HF:135-168), the path the
                     reference actually executes because torch is importable
                     (HF:321-324).  This one is timed as the CPU baseline.

Pinned by tests/test_oracle_golden.py against tests/golden/*.npz, which were
produced by the installed WhisperFeatureExtractor (tests/golden/make_golden.py).
"""
from __future__ import annotations

import numpy as np

SAMPLING_RATE = 16000
N_FFT = 400
HOP = 160
N_SAMPLES = 480000          # HF:91  chunk_length * sampling_rate
N_FRAMES = 3000             # HF:92  n_samples // hop_length
N_FREQ = 1 + N_FFT // 2     # HF:96


# --------------------------------------------------------------------------- F1
def _hz_to_mel_slaney(f):
    """AU:470-481 (slaney branch): linear below 1 kHz, log above."""
    f = np.asarray(f, dtype=np.float64)
    lin = 3.0 * f / 200.0
    logstep = 27.0 / np.log(6.4)
    with np.errstate(divide="ignore", invalid="ignore"):
        lg = 15.0 + np.log(f / 1000.0) * logstep
    return np.where(f >= 1000.0, lg, lin)


def _mel_to_hz_slaney(m):
    """AU:506-517 (slaney branch)."""
    m = np.asarray(m, dtype=np.float64)
    lin = 200.0 * m / 3.0
    logstep = np.log(6.4) / 27.0
    lg = 1000.0 * np.exp(logstep * (m - 15.0))
    return np.where(m >= 15.0, lg, lin)


def mel_filter_bank(n_mel: int, n_freq: int = N_FREQ, fmin: float = 0.0,
                    fmax: float = 8000.0, sr: int = SAMPLING_RATE) -> np.ndarray:
    """(n_freq, n_mel) float64 slaney-normalised triangular bank.

    Follows AU:638-729 as called from HF:95-103 (norm="slaney",
    mel_scale="slaney", triangles in Hz space, AU:541-560; energy norm
    AU:717-720).
    """
    mel_pts = np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mel + 2)
    hz_pts = _mel_to_hz_slaney(mel_pts)
    fft_hz = np.linspace(0, sr // 2, n_freq)
    width = np.diff(hz_pts)
    d = hz_pts[None, :] - fft_hz[:, None]              # (n_freq, n_mel+2)
    down = -d[:, :-2] / width[:-1]
    up = d[:, 2:] / width[1:]
    bank = np.maximum(0.0, np.minimum(down, up))
    bank = bank * (2.0 / (hz_pts[2:n_mel + 2] - hz_pts[:n_mel]))[None, :]
    return bank


def hann_periodic(n: int = N_FFT) -> np.ndarray:
    """torch.hann_window(n) default periodic=True (HF:141): 0.5 - 0.5 cos(2 pi k / n)."""
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


# ------------------------------------------------------------------------ F2/F3
def pad_or_trim(wave, n_samples: int = N_SAMPLES):
    """Truncate to n_samples, right-pad zeros (HF:300-307 -> feature_extraction_
    sequence_utils.py _truncate :330-335, _pad :271-281).  Returns (padded f32, kept)."""
    w = np.asarray(wave, dtype=np.float32).reshape(-1)
    kept = min(w.shape[0], n_samples)
    out = np.zeros(n_samples, dtype=np.float32)
    out[:kept] = w[:kept]
    return out, kept


def zero_mean_unit_var(padded: np.ndarray, kept: int) -> np.ndarray:
    """HF:172-191 with an attention mask: statistics over the kept prefix only,
    pad region reset to 0.  numpy evaluates this in float32."""
    x = padded.astype(np.float32)
    head = x[:kept]
    y = (x - head.mean()) / np.sqrt(head.var() + 1e-7)
    y = y.astype(np.float32)
    if kept < y.shape[0]:
        y[kept:] = 0.0
    return y


def reflect_source_index(p: np.ndarray, n: int, half: int = N_FFT // 2) -> np.ndarray:
    """torch.stft(center=True, pad_mode='reflect'): padded position p -> index
    into the length-n signal (no edge repeat)."""
    i = p - half
    i = np.where(i < 0, -i, i)
    i = np.where(i >= n, 2 * (n - 1) - i, i)
    return i


def frame_matrix(padded: np.ndarray) -> np.ndarray:
    """(T, 400) frames; frame t covers padded positions [160 t, 160 t + 400)
    (SURVEY 8a F4).  T = n // hop, i.e. the last STFT frame is already dropped."""
    n = padded.shape[0]
    t = n // HOP
    p = (np.arange(t) * HOP)[:, None] + np.arange(N_FFT)[None, :]
    return padded[reflect_source_index(p, n)]


# --------------------------------------------------------------------- F4/F5/F6
def power_spectrum_f64(padded: np.ndarray) -> np.ndarray:
    """(201, T) float64 |STFT|^2."""
    fr = frame_matrix(padded).astype(np.float64) * hann_periodic()[None, :]
    spec = np.fft.rfft(fr, axis=1)
    return (spec.real ** 2 + spec.imag ** 2).T


def _finish(mel_power: np.ndarray) -> np.ndarray:
    """log10 with 1e-10 floor, per-utterance max-8 clamp, (x+4)/4 (HF:159-165)."""
    lg = np.log10(np.maximum(mel_power, 1e-10))
    lg = np.maximum(lg, lg.max() - 8.0)
    return (lg + 4.0) / 4.0


def logmel_f64(wave, n_mel: int = 80, n_samples: int = N_SAMPLES,
               do_normalize: bool = False) -> np.ndarray:
    """One utterance -> (n_mel, n_samples//160) float64."""
    padded, kept = pad_or_trim(wave, n_samples)
    if do_normalize:
        padded = zero_mean_unit_var(padded, kept)
    p = power_spectrum_f64(padded)
    mel = mel_filter_bank(n_mel).T @ p
    return _finish(mel)


def logmel_f32(waves, n_mel: int = 80, n_samples: int = N_SAMPLES,
               do_normalize: bool = False, num_threads: int | None = None) -> np.ndarray:
    """Batch of utterances -> (B, n_mel, T) float32 through torch.stft on the CPU,
    the path the reference executes (HF:135-168)."""
    import torch

    if num_threads is not None:
        torch.set_num_threads(num_threads)
    rows = []
    for w in waves:
        padded, kept = pad_or_trim(w, n_samples)
        if do_normalize:
            padded = zero_mean_unit_var(padded, kept)
        rows.append(padded)
    x = torch.from_numpy(np.stack(rows, axis=0))
    win = torch.hann_window(N_FFT)
    st = torch.stft(x, N_FFT, HOP, window=win, return_complex=True)
    mag = (st[..., :-1].abs() ** 2).contiguous()
    bank = torch.from_numpy(mel_filter_bank(n_mel)).to(torch.float32)
    mel = bank.T @ mag
    lg = torch.clamp(mel, min=1e-10).log10()
    peak = lg.amax(dim=(1, 2), keepdim=True)
    lg = torch.maximum(lg, peak - 8.0)
    lg = (lg + 4.0) / 4.0
    return lg.numpy()


# --------------------------------------------------------------------------- F7
def frame_attention_mask(kept: int, n_samples: int = N_SAMPLES) -> np.ndarray:
    """HF:332-341: sample mask[::160], last entry dropped when n_samples % 160 != 0."""
    m = np.zeros(n_samples, dtype=np.int32)
    m[:kept] = 1
    r = m[::HOP]
    if n_samples % HOP != 0:
        r = r[:-1]
    return r
