#!/usr/bin/env python3
"""Diagnostic: cycles per section of the resident-table kernel (frontend_r_kernel), per wave role.
Loads the stamped build (make -C whisper-finetune_amd/csrc stamps).  Usage: phase_breakdown_r.py [batch] [n_mel]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "whisper-finetune_amd"))
from wfe_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "whisper-finetune_amd", "wfe_amd", "libwfe_hip_stamps.so")
import bench  # noqa: E402
from wfe_amd.frontend import HipFrontEnd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_mel = int(sys.argv[2]) if len(sys.argv) > 2 else 80
fe = HipFrontEnd(n_mel=n_mel, device=0)
lib = _lib.lib()
lib.wfe_debug_stamps.restype = C.c_int
lib.wfe_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
n_wg = torch.cuda.get_device_properties(0).multi_processor_count
stamps = torch.zeros((n_wg * 8, 16), dtype=torch.int64, device="cuda")
wav = bench.device_clips(B, torch.device("cuda", 0), 0)
lens = torch.full((B,), 480000, dtype=torch.int32, device="cuda")
for _ in range(20):
    fe.logmel(wav, lens)
torch.cuda.synchronize()
assert lib.wfe_debug_stamps(fe._h, C.c_void_p(stamps.data_ptr())) == 0
fe.logmel(wav, lens)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(n_wg, 8, 16).astype(np.float64)
names = ["top", "gemm w>=4", "fin-rest", "st-load", "gemm w<4", "barrier", "mel-sum", "log+store", "fold", "st-write"]
ph = s[:, :, 10]
print("phases per wave: median", np.median(ph))
print("wave  " + "  ".join(f"{n:>9s}" for n in names) + "      total   (cycles per phase, mean over workgroups)")
for w in range(8):
    per = s[:, w, :10] / np.maximum(ph[:, w:w + 1], 1)
    m = per.mean(axis=0)
    print(f"  {w}   " + "  ".join(f"{x:9.0f}" for x in m) + f"  {m.sum():9.0f}")
