#!/usr/bin/env python3
"""Diagnostic: cost of kernel 1's augmentation staging alone, through waveform_kernel
(the same stage_augmented code, no DFT).  256 x 30 s, device resident; one line per
chain.  The NOISE chains include the resolve kernels (~0.2 ms)."""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-finetune_amd")):
    sys.path.insert(0, p)
from wfe_amd.augment import AugmentPlanner  # noqa: E402
from wfe_amd.frontend import HipFrontEnd, NoiseBank  # noqa: E402

ENTRY = {
    "noise": {"type": "noise", "prob": 1.0, "params": {"min_snr_dB": 10, "max_snr_dB": 50}},
    "speed": {"type": "speed", "prob": 1.0, "params": {"min_speed_rate": 0.9, "max_speed_rate": 1.1, "num_rates": 3}},
    "shift": {"type": "shift", "prob": 1.0, "params": {"min_shift_ms": -5, "max_shift_ms": 5}},
    "volume": {"type": "volume", "prob": 1.0, "params": {"min_gain_dBFS": -15, "max_gain_dBFS": 15}},
}


def main():
    B = 256
    fe = HipFrontEnd(n_mel=80, device=0)
    wav = 0.1 * torch.randn((B, 480000), device=fe.device)
    lens = torch.full((B,), 480000, dtype=torch.int32, device=fe.device)
    bank = [(0.05 * np.random.default_rng(9000 + j).standard_normal(160000)).astype(np.float32) for j in range(16)]
    nb = NoiseBank(bank, fe.device)
    for chain in ([], ["volume"], ["shift", "volume"], ["noise"], ["speed"], ["noise", "speed"],
                  ["noise", "speed", "shift", "volume"]):
        planner = AugmentPlanner([ENTRY[k] for k in chain], noise_lengths=[len(c) for c in bank])
        rnd = random.Random(0)
        packed = fe.pack_plans([planner.draw(480000, 16000, rnd)[0] for _ in range(B)])
        run = lambda: fe.augment_waveform(wav, lens, packed, nb)     # noqa: E731
        for _ in range(2):
            run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(5):
            run()
        b.record()
        torch.cuda.synchronize()
        print(f"{'+'.join(chain) or 'none':32s} {a.elapsed_time(b) / 5:8.3f} ms", flush=True)


if __name__ == "__main__":
    main()
