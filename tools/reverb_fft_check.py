#!/usr/bin/env python3
"""wfe_reverb: the partitioned overlap-save FFT form (impulse-response banks whose longest response has 2048 taps or
more) against the direct sum (WFE_TUNE=8192 keeps it for every length) and against scipy.signal.fftconvolve in
float64; times of both.  Ragged utterances, one copied through, responses around the partition size."""
import os
import sys

import numpy as np
import torch
from scipy.signal import fftconvolve

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-finetune_amd")):
    sys.path.insert(0, p)
from wfe_amd.frontend import HipFrontEnd, NoiseBank  # noqa: E402

os.environ.pop("WFE_TUNE", None)
if os.environ.get("FFT_TUNE"):                      # (16384: the one-kernel form for every length)
    os.environ["WFE_TUNE"] = os.environ["FFT_TUNE"]
fe_fft = HipFrontEnd(n_mel=80, device=0)
os.environ["WFE_TUNE"] = "8192"
fe_dir = HipFrontEnd(n_mel=80, device=0)
dev = fe_fft.device
B, n = int(os.environ.get("B", 64)), 480000


def timed(fn, steps=3):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


g = torch.Generator(device=dev).manual_seed(1)
x = 0.1 * torch.randn((B, n), device=dev, generator=g)
lens = torch.full((B,), n, dtype=torch.int32, device=dev)
lens[1], lens[2], lens[3], lens[4], lens[6] = n // 3, 5, 2049, 0, 4096
host = [int(v) for v in lens.tolist()]
xh = x.cpu().numpy()
worst = 0.0
for taps in [int(t) for t in os.environ.get("TAPS", "2048,2049,4095,4096,4097,8000,16000,48000").split(",")]:
    rng = np.random.default_rng(taps)
    clips = [(rng.standard_normal(t) * np.exp(-np.arange(t) / max(t / 4, 1))).astype(np.float32)
             for t in (taps, max(taps - 1, 1), min(300, taps))]
    idx = [i % 3 for i in range(B)]
    idx[5] = -1
    rf, rd = NoiseBank(clips, dev), NoiseBank(clips, dev)
    y_fft = fe_fft.reverb(x, lens, rf, idx, host_lengths=host)
    y_dir = fe_dir.reverb(x, lens, rd, idx, host_lengths=host)
    torch.cuda.synchronize()
    yf, yd = y_fft.cpu().numpy(), y_dir.cpu().numpy()
    rel_dir = rel_ref = 0.0
    for b in range(min(B, 9)):
        m = host[b]
        assert not yf[b, m:].any()
        if m == 0:
            continue
        if idx[b] < 0:
            assert np.array_equal(yf[b, :m], xh[b, :m])
            continue
        h = clips[idx[b]].astype(np.float64)
        ref = fftconvolve(xh[b, :m].astype(np.float64), h, mode="same")
        ref *= np.abs(xh[b, :m]).max() / np.abs(ref).max()
        scale = np.abs(ref).max() * max(1.0, np.sqrt(len(h) / 64))
        rel_ref = max(rel_ref, np.abs(yf[b, :m] - ref).max() / scale)
        rel_dir = max(rel_dir, np.abs(yd[b, :m] - ref).max() / scale)
    worst = max(worst, rel_ref)
    t_fft = timed(lambda: fe_fft.reverb(x, lens, rf, idx, host_lengths=host))
    t_dir = timed(lambda: fe_dir.reverb(x, lens, rd, idx, host_lengths=host), steps=1)
    print(f"taps {taps:6d}: err/(max|y| sqrt(taps/64)) vs float64: fft {rel_ref:.2e}  direct {rel_dir:.2e}   "
          f"fft {t_fft:8.3f} ms   direct {t_dir:8.3f} ms   x{t_dir / t_fft:.1f}", flush=True)
assert worst <= 3e-6, worst
print("ok")
