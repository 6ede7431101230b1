#!/usr/bin/env python3
"""End-to-end input pipeline: WAV files -> CustomDataset -> DataLoader -> collator ->
input_features, this repository's way (num_workers=0, 16-bit PCM memory-mapped, one
pinned copy + one fused launch per batch) beside the reference's way
(utils/reader.py:134-216: float32 decode + WhisperFeatureExtractor per item inside
num_workers forked workers, then a stacking collator).  Synthetic 16 kHz mono WAVs of
5..30 s in a temporary directory (page-cache resident); stub tokenizer.  One JSON
line per mode.  Indicative: the host side depends on the box's cores and storage."""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time
import wave

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-finetune_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


class ReferenceStyle(torch.utils.data.Dataset):
    """What CustomDataset.__getitem__ of the reference does per item, with the stdlib
    wave reader standing in for soundfile (not installed here)."""

    def __init__(self, recs, n_mel):
        from transformers import WhisperFeatureExtractor
        self.recs, self.fe = recs, WhisperFeatureExtractor(feature_size=n_mel)

    def __len__(self):
        return len(self.recs)

    def __getitem__(self, i):
        from wfe_amd.data import read_audio
        x, sr = read_audio(self.recs[i]["audio"]["path"])
        return {"input_features": self.fe(x, sampling_rate=sr).input_features, "labels": [1, 2, 3], "id": str(i)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--n-mel", type=int, default=80)
    ap.add_argument("--workers", type=int, default=8, help="DataLoader workers of the reference-style mode")
    ap.add_argument("--modes", default="gpu,gpu-f32,cpu")
    args = ap.parse_args()
    import stubs
    from wfe_amd import data as wd
    tmp = tempfile.mkdtemp(prefix="wfe_pipe_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    rng = np.random.default_rng(0)
    base = (3000 * rng.standard_normal(30 * 16000)).astype("<i2")
    recs, seconds = [], 0.0
    for i in range(args.clips):
        n = int(rng.integers(5, 31)) * 16000
        p = os.path.join(tmp, f"{i}.wav")
        with wave.open(p, "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(16000)
            w.writeframes(np.roll(base, i * 37)[:n].tobytes())
        recs.append({"audio": {"path": p}, "sentence": "hello world", "duration": n / 16000})
        seconds += n / 16000
    mf = os.path.join(tmp, "list.json")
    with open(mf, "w") as f:
        f.write("\n".join(json.dumps(r) for r in recs))
    proc = stubs.StubProcessor(n_mel=args.n_mel)
    try:
        for mode in args.modes.split(","):
            if mode.startswith("gpu"):
                os.environ["WFE_PCM16"] = "0" if mode == "gpu-f32" else "1"
                ds = wd.CustomDataset(mf, proc)
                coll = wd.DataCollatorSpeechSeq2SeqWithPadding(processor=proc, device_output=True)
                loader = torch.utils.data.DataLoader(ds, batch_size=args.batch, collate_fn=coll, num_workers=0)
            else:
                ds = ReferenceStyle(recs, args.n_mel)
                coll = wd.DataCollatorSpeechSeq2SeqWithPadding(processor=proc)
                loader = torch.utils.data.DataLoader(ds, batch_size=args.batch, collate_fn=coll,
                                                     num_workers=args.workers, pin_memory=True)
            for epoch in range(2):                     # epoch 0 warms caches, workers and the allocator
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 0
                for batch in loader:
                    n += batch["input_features"].shape[0]
                    feats = batch["input_features"].to("cuda", non_blocking=True)
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
            print(json.dumps({"metric": "input pipeline audio-hours/sec (files -> input_features on the GPU)",
                              "mode": {"gpu": "this repo, int16 PCM to the GPU, num_workers=0",
                                       "gpu-f32": "this repo, float32 host decode, num_workers=0",
                                       "cpu": f"reference style, CPU extractor in {args.workers} workers"}[mode],
                              "value": seconds / 3600.0 / el, "clips_per_sec": n / el, "clips": n,
                              "batch": args.batch, "mean_clip_seconds": seconds / args.clips,
                              "n_mel": args.n_mel}), flush=True)
            del loader
    finally:
        for r in recs:
            os.remove(r["audio"]["path"])
        os.remove(mf)
        os.rmdir(tmp)


if __name__ == "__main__":
    main()
