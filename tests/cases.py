"""Seeded input generators shared by tests/golden/make_golden.py (which runs in
the build container, next to the reference) and by the parity tests (which run
on the GPU box, where neither /root/reference nor its outputs exist).  Every
input is a pure function of (name, seed) so that only the expected OUTPUTS have
to be committed as fixtures."""
from __future__ import annotations

import os
import wave

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
SR = 16000
N30 = 480000


def synth_clip(i: int, n: int = N30) -> np.ndarray:
    """BASELINE.md section 2 / SURVEY 8d config 1: rng(1234+i), 0.1*N(0,1),
    amplitude-modulated by 0.5(1+sin(2 pi 3 t))."""
    rng = np.random.default_rng(1234 + i)
    x = 0.1 * rng.standard_normal(n)
    t = np.arange(n) / SR
    return (x * 0.5 * (1.0 + np.sin(2 * np.pi * 3.0 * t))).astype(np.float32)


def bench_clip(seed: int, n: int = N30) -> np.ndarray:
    """seeds 0..255, same recipe."""
    rng = np.random.default_rng(seed)
    x = 0.1 * rng.standard_normal(n)
    t = np.arange(n) / SR
    return (x * 0.5 * (1.0 + np.sin(2 * np.pi * 3.0 * t))).astype(np.float32)


def noise_clip(j: int, seconds: float = 10.0) -> np.ndarray:
    """SURVEY 8d config 3 noise bank: 16 kHz, seeds 9000+j."""
    rng = np.random.default_rng(9000 + j)
    return (0.05 * rng.standard_normal(int(seconds * SR))).astype(np.float32)


def rir_clip(seed: int, taps: int) -> np.ndarray:
    """A synthetic room impulse response: direct path + exponentially decaying noise tail."""
    rng = np.random.default_rng(7000 + seed)
    h = rng.standard_normal(taps) * np.exp(-np.arange(taps) / (0.2 * taps + 1.0))
    h[0] = 1.0
    return (0.5 * h).astype(np.float32)


def load_test_wav() -> np.ndarray:
    """The reference's dataset/test.wav (mono 16 kHz s16, 134240 frames) decoded
    as soundfile.read(dtype='float32') does: int16 / 32768."""
    with wave.open(os.path.join(GOLDEN, "test.wav")) as w:
        assert w.getnchannels() == 1 and w.getsampwidth() == 2 and w.getframerate() == SR
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    return (pcm.astype(np.float32) / 32768.0).astype(np.float32)


def _tone(freq, amp, n=N30):
    t = np.arange(n, dtype=np.float64) / SR
    return (amp * np.sin(2 * np.pi * freq * t)).astype(np.float32)


def logmel_cases() -> dict:
    """name -> waveform (float32, any length)."""
    rng = np.random.default_rng(77)
    hdr = (1e-4 * rng.standard_normal(N30)).astype(np.float32)
    hdr[100000:108000] += (0.8 * rng.standard_normal(8000)).astype(np.float32)
    speechlike = np.zeros(N30, dtype=np.float32)
    for k, f in enumerate([180.0, 360.0, 540.0, 1200.0, 2500.0, 3400.0]):
        speechlike += _tone(f, 0.2 / (k + 1))
    speechlike *= (0.5 * (1 + np.sin(2 * np.pi * 1.5 * np.arange(N30) / SR))).astype(np.float32)
    speechlike += (1e-3 * rng.standard_normal(N30)).astype(np.float32)
    return {
        "test_wav": load_test_wav(),
        "synth0": synth_clip(0),
        "synth1": synth_clip(1),
        "short_5p3s": synth_clip(2, 84811),
        "exact_30s": (0.3 * rng.standard_normal(N30)).astype(np.float32),
        "long_31p3s": synth_clip(3, 500123),
        "zeros": np.zeros(12345, dtype=np.float32),
        "tone440": _tone(440.0, 0.5),
        "tone_1001p7_short": _tone(1001.7, 0.25, 160 * 1000 + 37),
        "hdr_burst": hdr,
        "speechlike": speechlike,
        "tiny_100": (0.2 * rng.standard_normal(100)).astype(np.float32),
        "single_sample": np.array([0.5], dtype=np.float32),
        "len_199": (0.2 * rng.standard_normal(199)).astype(np.float32),
        "len_200": (0.2 * rng.standard_normal(200)).astype(np.float32),
        "len_201": (0.2 * rng.standard_normal(201)).astype(np.float32),
        "end_479999": (0.2 * rng.standard_normal(479999)).astype(np.float32),
    }


# frames kept in the committed fixtures: both edges (reflection / pad region)
# plus a stride through the middle
def kept_frames(t: int = 3000) -> np.ndarray:
    idx = set(range(0, min(24, t))) | set(range(max(0, t - 24), t)) | set(range(0, t, 11))
    return np.array(sorted(idx), dtype=np.int64)
