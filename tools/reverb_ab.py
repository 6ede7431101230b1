#!/usr/bin/env python3
"""wfe_reverb: the sliding-window kernel against round 1's reverb_kernel (WFE_TUNE=4), same bits expected; times
of both.  Batch 64 x 30 s, impulse responses of several lengths (odd, even, shorter than a tap block, longer
than a chunk), ragged utterances, one utterance copied through."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-finetune_amd")):
    sys.path.insert(0, p)
from wfe_amd.frontend import HipFrontEnd, NoiseBank  # noqa: E402

os.environ.pop("WFE_TUNE", None)
fe_new = HipFrontEnd(n_mel=80, device=0)
os.environ["WFE_TUNE"] = "4"
fe_old = HipFrontEnd(n_mel=80, device=0)
dev = fe_new.device
B, n = 64, 480000


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
lens[1], lens[2], lens[3] = n // 3, 5, 2049
host = [int(v) for v in lens.tolist()]
for taps in (1, 3, 8, 100, 511, 512, 513, 1600, 8000):
    rng = np.random.default_rng(taps)
    clips = [(rng.standard_normal(t) * np.exp(-np.arange(t) / max(t / 4, 1))).astype(np.float32) for t in (taps, max(taps - 1, 1))]
    idx = [i % 2 for i in range(B)]
    idx[5] = -1
    y_new = fe_new.reverb(x, lens, NoiseBank(clips, dev), idx, host_lengths=host)
    y_old = fe_old.reverb(x, lens, NoiseBank(clips, dev), idx, host_lengths=host)
    same = torch.equal(y_new, y_old)
    rn, ro = NoiseBank(clips, dev), NoiseBank(clips, dev)
    t_new = timed(lambda: fe_new.reverb(x, lens, rn, idx, host_lengths=host))
    t_old = timed(lambda: fe_old.reverb(x, lens, ro, idx, host_lengths=host))
    print(f"taps {taps:5d}: identical {same}   window {t_new:8.3f} ms   reverb_kernel {t_old:8.3f} ms   x{t_old / t_new:.2f}")
