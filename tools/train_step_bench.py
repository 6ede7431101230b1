#!/usr/bin/env python3
"""BASELINE config 5 (indicative): is a whisper-large-v3 LoRA training step
input-bound?  Compares the reference's CPU front end inside DataLoader workers
(transformers.WhisperFeatureExtractor per item, utils/reader.py:195, num_workers=8,
finetune.py:26,129) with the MI355X front end in the collator (num_workers=0), on
the same synthetic utterances, same model, same optimiser.

The model side is NOT part of this repository's scope: it is a random-init
`WhisperForConditionalGeneration` with the large-v3 geometry (no checkpoint is
available offline) and a minimal LoRA (r=32, alpha=64 on q/k/v/out/fc1/fc2 as
finetune.py:98-104; peft is not installed), bf16 autocast, AdamW on the adapters.
Run with torch.distributed.run for N > 1 (DDP over RCCL; the front end adds no
collective).  Prints one JSON line per mode on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-finetune_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

SIZES = {   # d_model, layers, heads, ffn, n_mel, vocab
    "tiny": (384, 4, 6, 1536, 80, 51865),
    "small": (768, 12, 12, 3072, 80, 51865),
    "large-v3": (1280, 32, 20, 5120, 128, 51866),
}


class LoRALinear(nn.Module):
    def __init__(self, base: nn.Linear, r=32, alpha=64, dropout=0.05):
        super().__init__()
        self.base = base
        for p in self.base.parameters():
            p.requires_grad_(False)
        self.a = nn.Linear(base.in_features, r, bias=False)
        self.b = nn.Linear(r, base.out_features, bias=False)
        nn.init.zeros_(self.b.weight)
        self.scale = alpha / r
        self.drop = nn.Dropout(dropout)

    def forward(self, x):
        return self.base(x) + self.b(self.a(self.drop(x))) * self.scale


def add_lora(model):
    targets = ("k_proj", "q_proj", "v_proj", "out_proj", "fc1", "fc2")
    for p in model.parameters():
        p.requires_grad_(False)
    for mod in list(model.modules()):
        for name, child in list(mod.named_children()):
            if name in targets and isinstance(child, nn.Linear):
                setattr(mod, name, LoRALinear(child))
    return model


class Utterances(torch.utils.data.Dataset):
    """Synthetic corpus: utterance i = seeded noise of 5..25 s, 48 label tokens."""

    def __init__(self, n, n_mel, mode):
        self.n, self.mode = n, mode
        prng = np.random.default_rng(1)      # the waveforms come from a pool: generating noise is not decode work
        self.pool = [(0.1 * prng.standard_normal(25 * 16000)).astype(np.float32) for _ in range(32)]
        if mode == "cpu":
            from transformers import WhisperFeatureExtractor
            self.fe = WhisperFeatureExtractor(feature_size=n_mel)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        rng = np.random.default_rng(i)
        wav = self.pool[i % 32][:int(rng.integers(5, 25) * 16000)]
        labels = rng.integers(1000, 50000, size=48).tolist()
        if self.mode == "cpu":     # the reference: features computed in the worker (utils/reader.py:195)
            return {"input_features": self.fe(wav, sampling_rate=16000).input_features, "labels": labels, "id": str(i)}
        return {"waveform": wav, "aug_ops": [], "labels": labels, "id": str(i), "specaug": False}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="large-v3", choices=list(SIZES))
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--modes", default="none,gpu,gpu+prefetch,cpu")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as td
        td.init_process_group("nccl", device_id=dev)

    from transformers import WhisperConfig, WhisperForConditionalGeneration
    import stubs
    from wfe_amd import data as wd
    d, layers, heads, ffn, n_mel, vocab = SIZES[args.size]
    cfg = WhisperConfig(vocab_size=vocab, num_mel_bins=n_mel, d_model=d, encoder_layers=layers, decoder_layers=layers,
                        encoder_attention_heads=heads, decoder_attention_heads=heads, encoder_ffn_dim=ffn,
                        decoder_ffn_dim=ffn, max_source_positions=1500, max_target_positions=448)
    torch.manual_seed(0)
    model = add_lora(WhisperForConditionalGeneration(cfg)).to(dev)
    model.config.use_cache = False
    if world > 1:
        model = nn.parallel.DistributedDataParallel(model, device_ids=[local])
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4)
    proc = stubs.StubProcessor(n_mel=n_mel)
    for mode in args.modes.split(","):
        # modes: "gpu" = the MI355X front end in the collator, serial with the step (num_workers=0);
        # "gpu+prefetch" = the same, iterated one batch ahead on a thread (wfe_amd.data.PrefetchIterator, what
        # wfe_amd.launch installs); "cpu" = the reference's structure (extractor per item in 8 workers);
        # "none" = no input pipeline at all (one precomputed batch reused): the step's own time
        gpu = mode.startswith("gpu")
        # GPU modes keep the features on the device (no pinning of CUDA tensors); the CPU mode is the reference's
        coll = wd.DataCollatorSpeechSeq2SeqWithPadding(processor=proc, device_output=gpu,
                                                       private_handle=(mode == "gpu+prefetch"))
        ds = Utterances((args.steps + args.warmup) * args.batch * world, n_mel, "cpu" if mode == "cpu" else "gpu")
        sampler = torch.utils.data.distributed.DistributedSampler(ds, world, rank, shuffle=False) if world > 1 else None
        loader = torch.utils.data.DataLoader(ds, batch_size=args.batch, sampler=sampler, collate_fn=coll,
                                             num_workers=8 if mode == "cpu" else 0, pin_memory=(mode == "cpu"),
                                             persistent_workers=False)
        it = iter(loader)
        if mode == "gpu+prefetch":
            it = wd.PrefetchIterator(it, device=local)
        elif mode == "none":
            one = next(it)
            one = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in one.items()}
            it = iter(lambda: one, None)
        t_data = t0 = 0.0
        for step in range(args.steps + args.warmup):
            if step == args.warmup:
                torch.cuda.synchronize()
                t0, t_data = time.perf_counter(), 0.0
            td0 = time.perf_counter()
            batch = next(it)
            feats = batch["input_features"].to(dev, non_blocking=True)
            labels = batch["labels"].to(dev, non_blocking=True)
            t_data += time.perf_counter() - td0
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = model(input_features=feats, labels=labels).loss
            loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if rank == 0:
            print(json.dumps({"metric": "LoRA train samples/sec", "mode": mode + " front end", "model": args.size,
                              "value": args.steps * args.batch * world / el, "n_gpus": world,
                              "per_device_batch": args.batch, "ms_per_step": el / args.steps * 1e3,
                              "ms_per_step_waiting_for_data": t_data / args.steps * 1e3,
                              "note": "random-init weights, synthetic utterances 5-25 s, bf16 autocast, minimal LoRA "
                                      "r=32; indicative only"}), flush=True)
        del it, loader
    if world > 1:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
