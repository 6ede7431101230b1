#!/usr/bin/env python3
"""Randomised parity sweep of the HIP log-mel path against the reference class itself (the installed
transformers.WhisperFeatureExtractor; the oracle's torch.stft restatement where that is not importable):
clips of many characters and lengths, both mel sizes.  Not a test (run time, no fixed fixture): a search for
inputs the golden vectors do not cover.  Prints the worst error per character and every point above 1e-4 with
the reference's own distance from the float64 result there (the tests' excuse rule).
    python tools/fuzz_parity.py [clips per mel size, default 240] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-finetune_amd")):
    sys.path.insert(0, p)
from oracle import logmel as ol  # noqa: E402  (checker)
from wfe_amd.frontend import HipFrontEnd, pack_ragged  # noqa: E402

TOL = 1e-4
SR = 16000


def make_clip(kind: int, n: int, rng) -> np.ndarray:
    t = np.arange(n, dtype=np.float64) / SR
    if kind == 0:
        x = 0.1 * rng.standard_normal(n)
    elif kind == 1:                                   # tones, some near the top of the band
        x = sum(a * np.sin(2 * np.pi * f * t + ph) for a, f, ph in
                zip(rng.uniform(0.01, 0.4, 3), rng.uniform(30, 7990, 3), rng.uniform(0, 6.28, 3)))
    elif kind == 2:                                   # chirp
        f0, f1 = rng.uniform(20, 2000), rng.uniform(2000, 7900)
        x = 0.3 * np.sin(2 * np.pi * (f0 * t + (f1 - f0) * t * t / (2 * max(t[-1], 1e-3))))
    elif kind == 3:                                   # bursts over near-silence: the max-8 floor
        x = 1e-5 * rng.standard_normal(n)
        for _ in range(4):
            s = int(rng.integers(0, max(n - 400, 1)))
            x[s:s + 800] += 0.8 * rng.standard_normal(len(x[s:s + 800]))
    elif kind == 4:                                   # silence gaps and a DC offset
        x = 0.05 * rng.standard_normal(n) + rng.uniform(-0.2, 0.2)
        for _ in range(3):
            s = int(rng.integers(0, max(n - 1, 1)))
            x[s:s + int(rng.integers(100, 20000))] = 0.0
    elif kind == 5:                                   # tiny amplitude
        x = 1e-6 * rng.standard_normal(n)
    elif kind == 6:                                   # clipped
        x = np.clip(3.0 * rng.standard_normal(n), -1.0, 1.0)
    elif kind == 7:                                   # impulses
        x = np.zeros(n)
        x[rng.integers(0, n, size=max(n // 5000, 1))] = rng.uniform(-1, 1, size=max(n // 5000, 1))
    else:                                             # speech-like: amplitude-modulated coloured noise
        w = rng.standard_normal(n)
        x = 0.2 * np.convolve(w, np.ones(8) / 8, mode="same") * (0.5 + 0.5 * np.sin(2 * np.pi * rng.uniform(1, 6) * t)) ** 2
    return x.astype(np.float32)


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 240
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    try:
        from transformers import WhisperFeatureExtractor
        ref_cls = WhisperFeatureExtractor
    except Exception:  # noqa: BLE001
        ref_cls = None
    print("reference:", "transformers.WhisperFeatureExtractor" if ref_cls else "oracle.logmel_f32")
    total_bad = 0
    for n_mel in (80, 128):
        fe = HipFrontEnd(n_mel=n_mel, device=0)
        ref = ref_cls(feature_size=n_mel) if ref_cls else None
        worst = {}
        for c0 in range(0, per, 24):
            kinds = [int(rng.integers(0, 9)) for _ in range(24)]
            lens = [int(rng.choice([rng.integers(1, 2000), rng.integers(2000, 480000), 480000,
                                    rng.integers(480001, 520000)], p=[0.1, 0.6, 0.15, 0.15])) for _ in kinds]
            clips = [make_clip(k, n, rng) for k, n in zip(kinds, lens)]
            flat, off, ln = pack_ragged(clips, fe.device)
            got = fe.logmel(flat, ln, offsets=off).cpu().numpy()
            for i, (k, x) in enumerate(zip(kinds, clips)):
                want = ref(x, sampling_rate=SR)["input_features"][0] if ref else ol.logmel_f32([x], n_mel=n_mel)[0]
                err = np.abs(got[i] - want)
                worst[k] = max(worst.get(k, 0.0), float(err.max()))
                if err.max() > TOL:
                    truth = ol.logmel_f64(x, n_mel=n_mel)
                    over = err > TOL
                    ref_noise = np.abs(want - truth)
                    ours = np.abs(got[i] - truth)
                    unexcused = over & ~((ref_noise > TOL / 2) & (ours <= 2 * ref_noise + TOL))
                    print(f"  n_mel {n_mel} kind {k} len {len(x)}: {int(over.sum())} point(s) above {TOL:g} "
                          f"(max {err.max():.2e}; reference off float64 there by up to {ref_noise[over].max():.2e}, "
                          f"ours by {ours[over].max():.2e}); not excused: {int(unexcused.sum())}")
                    total_bad += int(unexcused.sum())
                    dump = os.environ.get("WFE_FUZZ_DUMP")             # directory: inputs and outputs of the failing clips
                    if dump and unexcused.any():
                        os.makedirs(dump, exist_ok=True)
                        np.savez_compressed(os.path.join(dump, f"fail_{n_mel}_{c0 + i}.npz"), x=x, got=got[i], want=want)
        print(f"n_mel {n_mel}: worst max-abs-err per character", {k: f"{v:.2e}" for k, v in sorted(worst.items())})
        fe.close()
    print("points above tolerance that the reference's own float32 noise does not explain:", total_bad)
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
