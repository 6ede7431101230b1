#!/usr/bin/env python3
"""Diagnostic: where a wave of kernel 1 spends its cycles.  Loads the stamped
build (make -C whisper-finetune_amd/csrc stamps), runs one batch and prints the
median cycles per phase.  Stamps go to a buffer of their own."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "whisper-finetune_amd"))
from wfe_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "build", os.environ.get("WFE_STAMP_LIB", "libwfe_hip_stamps.so"))
from wfe_amd.frontend import HipFrontEnd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_mel = int(sys.argv[2]) if len(sys.argv) > 2 else 80
fe = HipFrontEnd(n_mel=n_mel, device=0)
lib = _lib.lib()
lib.wfe_debug_stamps.restype = C.c_int
lib.wfe_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
blocks = B * 47
stamps = torch.zeros((blocks * 4, 16), dtype=torch.int64, device="cuda")
wav = 0.1 * torch.randn((B, 480000), device="cuda")
lens = torch.full((B,), 480000, dtype=torch.int32, device="cuda")
run = lambda: fe.logmel(wav, lens)                                   # noqa: E731
if os.environ.get("WFE_PHASE_AUG") == "1":                           # BASELINE config 3 chain (see bench.py --augment)
    import random
    from wfe_amd.augment import AugmentPlanner
    from wfe_amd.frontend import NoiseBank
    bank = [(0.05 * np.random.default_rng(9000 + j).standard_normal(160000)).astype(np.float32) for j in range(16)]
    planner = AugmentPlanner([
        {"type": "noise", "prob": 1.0, "params": {"min_snr_dB": 10, "max_snr_dB": 50}},
        {"type": "speed", "prob": 1.0, "params": {"min_speed_rate": 0.9, "max_speed_rate": 1.1, "num_rates": 3}},
        {"type": "shift", "prob": 1.0, "params": {"min_shift_ms": -5, "max_shift_ms": 5}},
        {"type": "volume", "prob": 1.0, "params": {"min_gain_dBFS": -15, "max_gain_dBFS": 15}},
    ], noise_lengths=[len(c) for c in bank])
    rnd = random.Random(0)
    packed = fe.pack_plans([planner.draw(480000, 16000, rnd)[0] for _ in range(B)])
    nb = NoiseBank(bank, fe.device)
    run = lambda: fe.augment_logmel(wav, lens, packed, nb)           # noqa: E731
for _ in range(3):
    run()
torch.cuda.synchronize()
assert lib.wfe_debug_stamps(fe._h, C.c_void_p(stamps.data_ptr())) == 0
run()
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.int64)
s = s[s[:, 0] != 0]
names = ["stage(issue+write)", "barrier", "dft k-loop", "butterfly+mel", "log+store+max"]
d = np.diff(s[:, :6], axis=1)
print("waves", len(s), "kernel span (cycles)", s[:, 5].max() - s[:, 0].min())
for i, nme in enumerate(names):
    print(f"{nme:22s} median {np.median(d[:, i]):9.0f}  p10 {np.percentile(d[:, i], 10):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f}")
tot = s[:, 5] - s[:, 0]
print(f"{'  k-loop barrier waits':22s} median {np.median(s[:, 7]):9.0f}  (sum over 24 barriers incl. 2 stamps each)")
print(f"{'total':22s} median {np.median(tot):9.0f}  p10 {np.percentile(tot, 10):9.0f}  p90 {np.percentile(tot, 90):9.0f}")

if os.environ.get("WFE_PHASE_AUG") == "1":           # slots 8..11 carry the staging passes of SPEED chains
    t0, t1, t2, nb, t3 = s[:, 8], s[:, 9], s[:, 10], s[:, 11], s[:, 12]
    print("(WFE_TUNE & 1: gather form -- indices / gathers / arithmetic; else contiguous form of SPEED tiles -- "
          "set-up / fetches' wait + mix + LDS / next fetch issue + barrier / interpolation + barrier)")
    for nme, sel in (("tiles staged in >= 3 batches (SPEED)", nb >= 3), ("tiles staged in 2 batches", nb == 2), ("tiles staged in 1 batch", nb == 1)):
        if not sel.any():
            continue
        print(f"{nme}: {sel.sum()} of {len(s)} waves; staging phase median {np.median(d[sel, 0]):.0f}")
        pre, post = s[:, 13] - s[:, 0], s[:, 1] - s[:, 14]      # chain set-up before the staging; twiddle request + slot writes after
        for lab, v in (("before (chain load)", pre), ("part 0", t0), ("part 1", t1), ("part 2", t2), ("part 3", t3), ("after (twiddles)", post)):
            print(f"  {lab:26s} median {np.median(v[sel]):9.0f}  p10 {np.percentile(v[sel], 10):9.0f}  p90 {np.percentile(v[sel], 90):9.0f}")
    sys.exit(0)

# co-residency: group waves by (xcc, se/sh/cu, simd) and measure how much of each wave's
# k-loop overlaps another wave's k-loop on the same SIMD
hw = s[:, 6] & 0xffffffff
xcc = s[:, 6] >> 32
key = (xcc << 16) | (((hw >> 8) & 255) << 4) | ((hw >> 4) & 3)
print("distinct (xcc,cu,simd):", len(np.unique(key)), " distinct (xcc,cu):", len(np.unique(key >> 4)),
     )
if len(np.unique(key)) < 8:          # stamps 6 / 7 not written by this kernel variant: no per-SIMD analysis
    sys.exit(0)
ov, tot_k = 0, 0
for k in np.unique(key)[:200]:
    w = s[key == k]
    w = w[np.argsort(w[:, 2])]
    for i in range(len(w)):
        a0, a1 = w[i, 2], w[i, 3]
        tot_k += a1 - a0
        for j in range(len(w)):
            if j != i:
                ov += max(0, min(a1, w[j, 3]) - max(a0, w[j, 2]))
print("fraction of k-loop time overlapped by another k-loop on the same SIMD: %.3f" % (ov / max(tot_k, 1)))

# residency: how long each SIMD holds 0 / 1 / 2 waves of this kernel (stamp 0 .. stamp 5 of a
# wave; the time between one workgroup's end and its successor's start shows up as "1 resident").
# Each XCD has its own clock base, so every SIMD is measured against its own first / last stamp.
level_time = np.zeros(4)
gaps = []
for k in np.unique(key)[:400]:
    w = s[key == k]
    ev = sorted([(t, 1) for t in w[:, 0]] + [(t, -1) for t in w[:, 5]])
    lvl, last = 0, ev[0][0]
    for t, dlt in ev:
        level_time[min(lvl, 3)] += t - last
        lvl, last = lvl + dlt, t
    ends = np.sort(w[:, 5])
    starts = np.sort(w[:, 0])[2:]                       # the first two start at launch
    if len(starts):
        gaps.append(np.median(starts - ends[:len(starts)]))
level_time /= level_time.sum()
print("SIMD residency between its first and last stamp: 0 waves %.3f, 1 wave %.3f, 2 waves %.3f" % tuple(level_time[:3]))
print("median (successor start - predecessor end) per SIMD slot: %.0f cycles" % np.median(gaps))

# wave slots: HW_ID[3:0] of the waves that shared a SIMD, in start order (round 3: can the slot's parity tell the two
# resident workgroups of a CU apart?)
wid = hw & 15
print("wave_id histogram:", np.bincount(wid, minlength=16).tolist())
tg = (hw >> 16) & 15
print("tg_id histogram:", np.bincount(tg, minlength=16).tolist())
# do the 4 waves of a workgroup agree on the slot, and do co-resident workgroups differ?
wg_rows = s[:len(s) // 4 * 4].reshape(-1, 4, s.shape[1])
wid4 = (wg_rows[:, :, 6] & 15)
print("workgroups whose 4 waves share one wave_id: %.3f; share its parity: %.3f" % (
    (wid4 == wid4[:, :1]).all(axis=1).mean(), ((wid4 & 1) == (wid4[:, :1] & 1)).all(axis=1).mean()))
same_par, n_pairs = 0, 0
for k in np.unique(key)[:400]:
    w = s[key == k]
    w = w[np.argsort(w[:, 0])]
    ids = w[:, 6] & 15
    for i in range(len(w)):
        for j in range(i + 1, len(w)):
            if w[j, 0] < w[i, 5]:                       # j started before i ended: co-resident
                n_pairs += 1
                same_par += int((ids[i] & 1) == (ids[j] & 1))
            else:
                break
print("co-resident wave pairs on a SIMD: %d, with EQUAL wave_id parity: %d" % (n_pairs, same_par))
k0 = np.unique(key)[0]
w = s[key == k0]
w = w[np.argsort(w[:, 0])]
print("first SIMD, (start, end, wave_id, tg_id):", [(int(a - w[0, 0]), int(b - w[0, 0]), int(c & 15), int((c >> 16) & 15)) for a, b, c in w[:12, [0, 5, 6]]])

# time each SIMD spends with 2 / 1 / 0 waves inside the k-loop, and the k-loop MFMA rate that implies
# (round 3: 558 MFMAs of 32 cycles per wave and k-loop; the mel MFMAs run outside it)
X = Y = Z = 0.0
work = 0.0
for k in np.unique(key)[:400]:
    w = s[key == k]
    ev = sorted([(t, 1) for t in w[:, 2]] + [(t, -1) for t in w[:, 3]])
    t_first, t_last = w[:, 0].min(), w[:, 5].max()
    lvl, last = 0, t_first
    for t, dlt in ev:
        if lvl >= 2: X += t - last
        elif lvl == 1: Y += t - last
        else: Z += t - last
        lvl, last = lvl + dlt, t
    Z += t_last - last
    work += len(w) * float(os.environ.get("WFE_KLOOP_MFMAS", "558")) * 32
tot_t = X + Y + Z
print("per SIMD: both in k-loop %.3f, one %.3f, none %.3f of the time; k-loop MFMA cycles / time = %.3f" % (X / tot_t, Y / tot_t, Z / tot_t, work / tot_t))
print("  if the pipe is full while both are in the k-loop, a k-loop running beside a non-k-loop partner gets %.3f of the pipe" % ((work - X) / max(Y, 1)))
