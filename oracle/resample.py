"""CPU restatement of the resampling step (TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline, never by the product).

Reference: ``CustomDataset.resample`` = ``librosa.resample(sample, orig_sr, target_sr)``
(/root/reference/utils/reader.py:316-319, default res_type 'soxr_hq'; called for the
resample augmentation :263-266 and for the return to 16 kHz :157-158).

PARITY UNPINNED: librosa and soxr are not installed here and the reference holds no
resampled fixture, so soxr's output cannot be reproduced or checked.  What is
restated is (a) librosa's length rule ``int(np.ceil(n * target / orig))`` with
zero-phase alignment (output n at input position n * orig / target) and (b) a
published filter of the same family and quality class: resampy's 'kaiser_best'
(librosa's documented high-quality alternative) -- Kaiser(beta = 14.7697) windowed
sinc, 64 zero crossings, roll-off 0.94759 -- evaluated exactly in float64.  The GPU
kernel uses the same filter rounded to float32; tests state their tolerance against
this oracle and sanity-check both against scipy.signal.resample_poly.
"""
from __future__ import annotations

from math import gcd

import numpy as np

ZEROS = 64
ROLLOFF = 0.9475937167399596
BETA = 14.769656459379492


def output_length(n: int, orig_sr: int, target_sr: int) -> int:
    """librosa.resample: n_samples = int(np.ceil(y.shape[-1] * ratio)), ratio = float(target_sr) / orig_sr."""
    return int(np.ceil(n * (float(target_sr) / orig_sr)))


def filter_value(tau: np.ndarray, scale: float) -> np.ndarray:
    x = scale * np.asarray(tau, dtype=np.float64)
    inside = np.abs(x) < ZEROS
    t = np.clip(x / ZEROS, -1.0, 1.0)
    win = np.i0(BETA * np.sqrt(1.0 - t * t)) / np.i0(BETA)
    return np.where(inside, scale * ROLLOFF * np.sinc(ROLLOFF * x) * win, 0.0)


def resample(x: np.ndarray, orig_sr: int, target_sr: int, dtype=np.float32) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    if orig_sr == target_sr:
        return x.astype(dtype)
    g = gcd(int(orig_sr), int(target_sr))
    L, M = target_sr // g, orig_sr // g
    scale = min(1.0, target_sr / orig_sr)
    k0 = int(np.floor(ZEROS / scale))
    n_out = output_length(len(x), orig_sr, target_sr)
    out = np.zeros(n_out, dtype=np.float64)
    xp = np.concatenate([np.zeros(k0 + 2), x, np.zeros(k0 + 4 + M)])       # zeros outside the signal
    j = np.arange(2 * k0 + 2)
    for n0 in range(0, n_out, 4096):
        n = np.arange(n0, min(n0 + 4096, n_out), dtype=np.int64)
        base = (n * M) // L
        frac = ((n * M) % L) / L
        k = base[:, None] - k0 + j[None, :]                                # input index of every tap
        k = np.minimum(k, len(x) + k0 + M)                                 # far right of the zero padding
        w = filter_value((j[None, :] - k0) - frac[:, None], scale)
        out[n - 0] = np.sum(w * xp[k + k0 + 2], axis=1)
    return out.astype(dtype)
