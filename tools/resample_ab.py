#!/usr/bin/env python3
"""wfe_resample: the phase-sharing kernel against resample_kernel (WFE_TUNE=4), same bits expected; times of both.
Batch 64 x 30 s per rate, device-resident input."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-finetune_amd")):
    sys.path.insert(0, p)
from wfe_amd.frontend import HipFrontEnd  # noqa: E402

os.environ.pop("WFE_TUNE", None)
fe_new = HipFrontEnd(n_mel=80, device=0)
os.environ["WFE_TUNE"] = os.environ.get("AB_OLD_TUNE", "4")
fe_old = HipFrontEnd(n_mel=80, device=0)
dev = fe_new.device
B = 64


def timed(fn, steps=5):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


for sr, tgt in ((44100, 16000), (48000, 16000), (8000, 16000), (22050, 16000), (11025, 16000), (32000, 16000),
                (24000, 16000), (96000, 16000), (64000, 16000), (16000, 44100), (16000, 8000), (12345, 16000)):
    n = 30 * sr
    g = torch.Generator(device=dev).manual_seed(sr)
    x = 0.1 * torch.randn((B, n), device=dev, generator=g)
    lens = torch.full((B,), n, dtype=torch.int32, device=dev)
    lens[1] = n // 3
    lens[2] = 7
    host = [int(v) for v in lens.tolist()]
    try:
        y_new = fe_new.resample(x, lens, sr, tgt, host_lengths=host)
        y_old = fe_old.resample(x, lens, sr, tgt, host_lengths=host)
    except ValueError as e:
        print(f"{sr} -> {tgt}: {e}")
        continue
    y_new = y_new[0] if isinstance(y_new, tuple) else y_new
    y_old = y_old[0] if isinstance(y_old, tuple) else y_old
    same = torch.equal(y_new, y_old)
    diff = float((y_new - y_old).abs().max())
    t_new = timed(lambda: fe_new.resample(x, lens, sr, tgt, host_lengths=host))
    t_old = timed(lambda: fe_old.resample(x, lens, sr, tgt, host_lengths=host))
    print(f"{sr:6d} -> {tgt:6d}: identical {same!s:5} (max diff {diff:.1e})   round 2 {t_new:7.3f} ms   resample_kernel {t_old:7.3f} ms   x{t_old / t_new:.2f}")
