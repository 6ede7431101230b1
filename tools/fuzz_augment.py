#!/usr/bin/env python3
"""Randomised sweep of the fused augmentation chain against the CPU oracle (oracle/augment.py, pinned bit-exact on
the reference's own outputs): random configurations -- any subset and ORDER of noise / speed / shift / volume
entries, random probabilities and ranges -- random utterance lengths, noise clips shorter and longer than the
utterances.  The planner replays the oracle's draws (same `random` stream), the kernels must give the oracle's
waveform: bit-exact without a noise op, within 2e-6 * max|x| per noise op with one (the gains pass
through float32 log10 / pow).  Not a test (no fixed fixture): a search for chains the tests do not cover.
    python tools/fuzz_augment.py [rounds, default 40] [seed]"""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-finetune_amd")):
    sys.path.insert(0, p)
from oracle import augment as oa  # noqa: E402  (checker)
from wfe_amd import _lib  # noqa: E402
from wfe_amd.augment import AugmentPlanner, is_host_op  # noqa: E402
from wfe_amd.frontend import HipFrontEnd, NoiseBank, pack_ragged  # noqa: E402

SR = 16000


def random_config(rng):
    entries = []
    kinds = ["noise", "speed", "shift", "volume", "noise", "shift", "volume"]
    rng.shuffle(kinds)
    seen_speed = False
    for k in kinds[: rng.randint(1, 5)]:
        prob = rng.choice([1.0, 1.0, 0.6, 0.0])
        if k == "noise":
            lo = rng.randint(0, 40)
            entries.append({"type": k, "prob": prob, "params": {"min_snr_dB": lo, "max_snr_dB": lo + rng.randint(0, 30)}})
        elif k == "speed":
            if seen_speed:
                continue
            seen_speed = True
            lo = rng.choice([0.5, 0.8, 0.9, 0.97])
            entries.append({"type": k, "prob": prob, "params": {"min_speed_rate": lo, "max_speed_rate": lo + rng.choice([0.2, 0.4, 1.3]),
                                                                 "num_rates": rng.randint(2, 5)}})
        elif k == "shift":
            lo = rng.randint(-30, 10)
            entries.append({"type": k, "prob": prob, "params": {"min_shift_ms": lo, "max_shift_ms": lo + rng.randint(0, 40)}})
        else:
            lo = rng.randint(-20, 5)
            entries.append({"type": k, "prob": prob, "params": {"min_gain_dBFS": lo, "max_gain_dBFS": lo + rng.randint(0, 20)}})
    return entries


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = random.Random(seed)
    nrng = np.random.default_rng(seed)
    fe = HipFrontEnd(n_mel=80, device=0)
    bad = 0
    worst_noise = 0.0
    for r in range(rounds):
        cfg = random_config(rng)
        bank = [(0.05 * nrng.standard_normal(n)).astype(np.float32)
                for n in (rng.randint(900, 3000), rng.randint(20000, 60000), rng.randint(100000, 200000))]
        nb = NoiseBank(bank, fe.device)
        waves, plans, wants, n_noise = [], [], [], []
        for u in range(12):
            n = rng.choice([rng.randint(600, 5000), rng.randint(5000, 90000), rng.randint(90000, 170000)])
            x = (0.1 * nrng.standard_normal(n)).astype(np.float32) * np.float32(rng.choice([1.0, 1e-3, 3.0]))
            s = rng.randrange(1 << 30)
            try:
                want, _, _ = oa.augment(x.copy(), SR, cfg, random.Random(s), bank)
                planner = AugmentPlanner(cfg, noise_lengths=[len(c) for c in bank])
                ops, n_out = planner.draw(n, SR, random.Random(s))
            except ValueError:
                continue                                         # (a shift beyond the clip: both raise)
            assert not any(is_host_op(o) for o in ops)
            assert n_out == len(want), (cfg, n, n_out, len(want))
            waves.append(x)
            plans.append(ops)
            wants.append(want)
            n_noise.append(sum(1 for o in ops if o[0] == _lib.AUG_NOISE))
        if not waves:
            continue
        n_samples = max(len(w) for w in wants) + 13
        flat, off, lens = pack_ragged(waves, fe.device)
        got, glen = fe.augment_waveform(flat, lens, plans, nb, n_samples=n_samples, offsets=off)
        mel = fe.augment_logmel(flat, lens, plans, nb, n_samples=(n_samples + 159) // 160 * 160, offsets=off)
        torch.cuda.synchronize()
        assert torch.isfinite(mel).all()
        got = got.cpu().numpy()
        for i, w in enumerate(wants):
            assert int(glen[i]) == len(w), (cfg, plans[i])
            err = float(np.abs(got[i, :len(w)] - w).max()) if len(w) else 0.0
            tail = bool(got[i, len(w):].any())
            tol = 2e-6 * n_noise[i] * max(float(np.abs(w).max()), 1e-30)
            if n_noise[i]:
                worst_noise = max(worst_noise, err / max(float(np.abs(w).max()), 1e-30))
            if err > tol or tail:
                bad += 1
                print(f"MISMATCH round {r} utt {i}: err {err:.3e} (tol {tol:.3e}) tail {tail} len {len(waves[i])} ops {plans[i]} cfg {cfg}")
    print(f"{rounds} rounds: mismatches {bad}; worst relative error on chains with noise {worst_noise:.2e}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
