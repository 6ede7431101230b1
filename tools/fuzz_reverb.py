#!/usr/bin/env python3
"""Random batches through wfe_reverb's FFT forms against add_reverb in float64 (fftconvolve 'same' + peak
normalisation): utterance lengths 0 .. 100 000 (log-uniform, and within 3 samples of a block boundary up to 40 blocks), response lengths 1 .. 20 000 taps
(log-uniform, with extra weight on the partition boundaries), banks of 1 - 4 responses, random copy-through rows,
contiguous and ragged (offsets) inputs, both forms (one kernel / three kernels) and forced small workspaces.
Prints the worst error in units of max|y| sqrt(taps / 64) (bound of the tests: 3e-6).

  python tools/fuzz_reverb.py [batches=300] [seed=0]
"""
import os
import sys

import numpy as np
import torch
from scipy.signal import fftconvolve

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-finetune_amd")):
    sys.path.insert(0, p)
from wfe_amd.frontend import HipFrontEnd, NoiseBank, pack_ragged  # noqa: E402


def main():
    n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    fes = {}
    for name, env in (("default", {}), ("one kernel", {"WFE_TUNE": "16384"}), ("sliced", {"WFE_RV_WS_MB": "1"})):
        for k in ("WFE_TUNE", "WFE_RV_WS_MB"):
            os.environ.pop(k, None)
        os.environ.update(env)
        fes[name] = HipFrontEnd(n_mel=80, device=0)
    worst, cases, forms = 0.0, 0, {k: 0 for k in fes}
    edges = [1, 2, 511, 512, 513, 2047, 2048, 2049, 4095, 4096, 4097, 6143, 6144, 6145, 8192, 8193, 16384, 16385]
    for it in range(n_batches):
        name = list(fes)[it % len(fes)]
        fe = fes[name]
        n_rir = int(rng.integers(1, 5))
        taps = [int(rng.choice(edges)) if rng.random() < 0.4 else int(np.exp(rng.uniform(0, np.log(20000)))) for _ in range(n_rir)]
        if max(taps) < 512:
            taps[0] = int(rng.integers(512, 9000))            # (shorter banks take the direct sum: not what is fuzzed here)
        rirs = []
        for t in taps:
            h = rng.standard_normal(t) * np.exp(-np.arange(t) / max(0.25 * t, 1.0))
            h[0] = 1.0
            rirs.append((0.5 * h).astype(np.float32))
        batch = int(rng.integers(1, 9))
        lens = []
        for _ in range(batch):
            r = rng.random()
            if r < 0.1:
                lens.append(0)
            elif r < 0.5:
                lens.append(max(0, int(rng.integers(1, 41)) * 2048 + int(rng.integers(-3, 4))))
            else:
                lens.append(int(np.exp(rng.uniform(0, np.log(100000)))))
        waves = [(0.1 * rng.standard_normal(n)).astype(np.float32) for n in lens]
        index = [int(rng.integers(-1, n_rir)) for _ in range(batch)]
        bank = NoiseBank(rirs, fe.device)
        if rng.random() < 0.5:
            flat, off, dl = pack_ragged(waves, fe.device)
            out = fe.reverb(flat, dl, bank, index, offsets=off, host_lengths=lens)
        else:
            width = max(max(lens), 1)
            x = np.zeros((batch, width), np.float32)
            for b, w in enumerate(waves):
                x[b, :len(w)] = w
            out = fe.reverb(torch.from_numpy(x).to(fe.device), torch.tensor(lens, dtype=torch.int32, device=fe.device),
                            bank, index, host_lengths=lens)
        out = out.cpu().numpy()
        for b, (w, j) in enumerate(zip(waves, index)):
            m = len(w)
            assert not out[b, m:].any(), (it, b)
            if m == 0:
                continue
            if j < 0:
                assert np.array_equal(out[b, :m], w), (it, b)
                continue
            y = fftconvolve(w.astype(np.float64), rirs[j].astype(np.float64), mode="same")
            pk = np.abs(y).max()
            if pk == 0:
                continue
            y *= np.abs(w).max() / pk
            err = np.abs(out[b, :m] - y).max() / (np.abs(y).max() * max(1.0, np.sqrt(len(rirs[j]) / 64)))
            if err > 3e-6:
                print("FAIL", it, name, "len", m, "taps", len(rirs[j]), "bank", taps, "err", err, flush=True)
                sys.exit(1)
            worst = max(worst, err)
            cases += 1
        forms[name] += 1
    print(f"{n_batches} batches, {cases} convolved utterances, forms {forms}: worst error {worst:.2e} "
          f"of max|y| sqrt(taps / 64) (bound 3e-6)")


if __name__ == "__main__":
    main()
