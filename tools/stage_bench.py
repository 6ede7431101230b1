#!/usr/bin/env python3
"""Timings of the stage kernels that only multi-rate / reverb chains use
(wfe_resample, wfe_reverb), batch 256 x 30 s, device-resident input.  One JSON line
per case; HIP-event timing via torch (the kernels run on torch's current stream)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-finetune_amd")):
    sys.path.insert(0, p)
from wfe_amd.frontend import HipFrontEnd, NoiseBank  # noqa: E402


def timed(fn, steps=5, warmup=2):
    for _ in range(warmup):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def main():
    # python tools/stage_bench.py [rate ...]: only the resampler, from these rates (default: every case)
    # python tools/stage_bench.py reverb [taps ...]: only the reverberation, with responses of these lengths
    only_reverb = sys.argv[1:2] == ["reverb"]
    taps_list = tuple(int(a) for a in sys.argv[2:]) if only_reverb and len(sys.argv) > 2 else (1600, 8000, 16000, 48000)
    rates = () if only_reverb else tuple(int(a) for a in sys.argv[1:]) or (44100, 48000, 8000, 22050)
    B = 256
    fe = HipFrontEnd(n_mel=80, device=0)
    dev = fe.device
    for sr in rates:
        n = 30 * sr
        x = 0.1 * torch.randn((B, n), device=dev)
        lens = torch.full((B,), n, dtype=torch.int32, device=dev)
        host = [n] * B
        ms = timed(lambda: fe.resample(x, lens, sr, 16000, host_lengths=host))
        print(json.dumps({"kernel": "resample", "from": sr, "to": 16000, "batch": B, "clip_seconds": 30,
                          "ms": ms, "audio_hours_per_sec": B * 30 / 3600 / (ms / 1e3),
                          "in_plus_out_GBps": (B * n * 4 + B * 480000 * 4) / ms / 1e6}), flush=True)
        del x
    if len(sys.argv) > 1 and not only_reverb:
        return
    x = 0.1 * torch.randn((B, 480000), device=dev)
    lens = torch.full((B,), 480000, dtype=torch.int32, device=dev)
    for taps in taps_list:
        rng = np.random.default_rng(taps)
        rirs = NoiseBank([(rng.standard_normal(taps) * np.exp(-np.arange(taps) / (0.2 * taps))).astype(np.float32)
                          for _ in range(4)], dev)
        idx = [i % 4 for i in range(B)]
        ms = timed(lambda: fe.reverb(x, lens, rirs, idx, out_stride=480000), steps=3, warmup=1)
        print(json.dumps({"kernel": "reverb", "taps": taps, "batch": B, "clip_seconds": 30, "ms": ms,
                          "audio_hours_per_sec": B * 30 / 3600 / (ms / 1e3),
                          "direct_sum_TFLOPs": 2.0 * B * 480000 * taps / ms / 1e9,       # (what the direct sum would need)
                          "form": "direct sum" if taps < 512 or int(os.environ.get("WFE_TUNE", "0")) & 8192 else
                                  "fft, one kernel" if taps <= 3 * 2048 else "fft, three kernels"}), flush=True)


if __name__ == "__main__":
    main()
