#!/usr/bin/env python3
"""A/B of kernel-1 experiment switches (WFE_TUNE, read at wfe_create) in one process, interleaved:
    python tools/ab_tune.py 0 1 3 [--n-mel 80] [--rounds 3] [--augment]
Prints the mean device time of kernel 1 and of the whole step per switch value."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "whisper-finetune_amd"))
import bench  # noqa: E402
from wfe_amd.frontend import HipFrontEnd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("tunes", nargs="+", type=int)
ap.add_argument("--n-mel", type=int, default=80)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=150)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--lengths", default="full")
args = ap.parse_args()
dev = torch.device("cuda", 0)
wav = bench.device_clips(args.batch, dev, 0)
lens = torch.full((args.batch,), 480000, dtype=torch.int32, device=dev)
if args.lengths == "speech":
    lens = torch.from_numpy(np.random.default_rng(99).integers(16000, 15 * 16000, size=args.batch).astype(np.int32)).to(dev)
fes = {}
for t in args.tunes:
    os.environ["WFE_TUNE"] = str(t)
    fes[t] = HipFrontEnd(n_mel=args.n_mel, device=0)
out = torch.empty((args.batch, args.n_mel, 3000), device=dev)
ref = None
res = {t: [] for t in args.tunes}
for r in range(args.rounds + 1):
    for t in args.tunes:
        fe = fes[t]
        for _ in range(40):
            fe.logmel(wav, lens, out=out)
        torch.cuda.synchronize()
        fe.kernel_timing(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fe.logmel(wav, lens, out=out)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / args.steps * 1e3
        k1, k2, n = fe.kernel_times()
        fe.kernel_timing(False)
        if r > 0:
            res[t].append((k1 / n, k2 / n, el))
        if ref is None:
            ref = out.clone()
        if t < 32:
            assert float((out - ref).abs().max()) <= 2e-5, f"tune {t} changes the output"  # (kernel 1 | 1R: mel sum order)
for t in args.tunes:
    a = np.array(res[t])
    print(f"tune {t:3d}: k1 {a[:, 0].mean():.4f} ms (min {a[:, 0].min():.4f})  k2 {a[:, 1].mean():.4f}  step {a[:, 2].mean():.4f} ms", flush=True)
