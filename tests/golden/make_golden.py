#!/usr/bin/env python3
"""Generate the committed golden vectors.  Runs ONLY in the build container
(needs /root/reference and the installed third-party ``transformers``); the
parity tests read the .npz files it writes and never import this script.

Sources of truth exercised here:
  * transformers.WhisperFeatureExtractor (the class the reference calls at
    utils/reader.py:192,195,198) on the seeded inputs of tests/cases.py;
  * the reference's own utils/reader.py, imported with EMPTY stub modules for
    the two packages that are not installed (librosa, soundfile) -- only the
    pure-numpy static methods and ``augment``/``add_noise`` are executed, with
    ``librosa.load`` returning a synthetic clip (audio decode is outside the
    pinned path).

Usage:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import hashlib
import json
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import cases  # noqa: E402


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def logmel_goldens():
    from transformers import WhisperFeatureExtractor

    out = {}
    meta = {}
    inputs = cases.logmel_cases()
    for n_mel in (80, 128):
        fe = WhisperFeatureExtractor(feature_size=n_mel)
        out[f"mel_filters_{n_mel}"] = fe.mel_filters.astype(np.float64)
        keep = cases.kept_frames(3000)
        for name, wav in inputs.items():
            feats = fe(wav, sampling_rate=16000)["input_features"]
            assert feats.shape == (1, n_mel, 3000) and feats.dtype == np.float32
            out[f"{name}_{n_mel}"] = feats[0][:, keep]
            meta[f"{name}_{n_mel}"] = {"sha256": sha(feats[0]), "min": float(feats.min()),
                                       "max": float(feats.max()), "len": int(wav.shape[0])}
        # one ragged batched call: per-utterance max must stay per-utterance
        names = ["short_5p3s", "tone440", "zeros", "hdr_burst"]
        feats = fe([inputs[n] for n in names], sampling_rate=16000)["input_features"]
        for i, n in enumerate(names):
            out[f"batched_{n}_{n_mel}"] = feats[i][:, keep]
        # do_normalize=True (infer_tfs.py:37) and the frame attention mask
        r = fe(inputs["test_wav"], sampling_rate=16000, do_normalize=True,
               return_attention_mask=True)
        out[f"normalized_test_wav_{n_mel}"] = r["input_features"][0][:, keep]
        out[f"attention_mask_test_wav_{n_mel}"] = np.asarray(r["attention_mask"][0], dtype=np.int32)
        # padding="longest": frame count follows the longest clip
        r = fe([inputs["short_5p3s"], inputs["tiny_100"]], sampling_rate=16000, padding="longest",
               return_attention_mask=True)
        f = r["input_features"]
        meta[f"longest_{n_mel}"] = {"shape": list(f.shape)}
        out[f"longest_short_5p3s_{n_mel}"] = f[0]
        out[f"longest_tiny_100_{n_mel}"] = f[1]
        out[f"longest_mask_{n_mel}"] = np.asarray(r["attention_mask"], dtype=np.int32)
    # stage intermediates for one clip (float64 numpy path of the same class)
    fe = WhisperFeatureExtractor(feature_size=80)
    np_path = fe._np_extract_fbank_features(
        np.stack([np.pad(inputs["test_wav"], (0, 480000 - inputs["test_wav"].shape[0]))]), "cpu")
    out["np_path_test_wav_80"] = np_path[0][:, cases.kept_frames(3000)].astype(np.float64)
    np.savez_compressed(os.path.join(HERE, "logmel_golden.npz"), **out)
    with open(os.path.join(HERE, "logmel_golden_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("logmel goldens:", len(out), "arrays")


def augment_goldens():
    # empty stand-ins so that ``import utils.reader`` resolves its two missing imports
    for missing in ("librosa", "soundfile"):
        if missing not in sys.modules:
            sys.modules[missing] = types.ModuleType(missing)
    sys.path.insert(0, "/root/reference")
    from utils.reader import CustomDataset  # the reference itself
    import librosa

    out = {}
    rng = np.random.default_rng(4242)
    base = (0.1 * rng.standard_normal(16000)).astype(np.float32)
    odd = (0.1 * rng.standard_normal(12347)).astype(np.float32)
    out["in_base"] = base
    out["in_odd"] = odd
    for rate in (0.9, 1.1, 0.95, 1.0):
        tag = str(rate).replace(".", "p")
        out[f"speed_{tag}_base"] = CustomDataset.change_speed(base.copy(), rate)
        out[f"speed_{tag}_odd"] = CustomDataset.change_speed(odd.copy(), rate)
    for ms in (-5, -1, 0, 3, 5, 250):
        out[f"shift_{ms}_base"] = CustomDataset.shift(base.copy(), 16000, ms)
    for g in (-15, -3, 0, 7, 15):
        out[f"volume_{g}_base"] = CustomDataset.volume(base.copy(), g)
    out["rms_db_base"] = np.asarray(CustomDataset.rms_db(base))
    out["rms_db_odd"] = np.asarray(CustomDataset.rms_db(odd))

    # add_noise mixing with the decode replaced by a synthetic clip
    ds = object.__new__(CustomDataset)
    short_noise = cases.noise_clip(0, 0.4)      # shorter than the sample -> wrap padding
    long_noise = cases.noise_clip(1, 2.5)       # longer -> random.randint start
    for tag, nz in (("short", short_noise), ("long", long_noise)):
        librosa.load = lambda path, sr, _nz=nz: (_nz.copy(), sr)
        for snr in (10, 33, 50):
            random.seed(123)
            out[f"noise_{tag}_{snr}"] = ds.add_noise(base.copy(), 16000, "x.wav", snr)

    # the whole chain in shipped file order, probabilities forced to 1 for the
    # four north-star stages, decisions from random.seed(s)
    with open("/root/reference/configs/augmentation.json") as f:
        cfg = json.load(f)
    for c in cfg:
        if c["type"] in ("speed", "shift", "volume", "noise"):
            c["prob"] = 1.0
        if c["type"] == "reverb":
            c["prob"] = 0.0
    out["chain_config_json"] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
    bank = [cases.noise_clip(j, 0.3 + 0.35 * j) for j in range(4)]
    ds.augment_configs = cfg
    ds.noise_files = [f"{j}.wav" for j in range(4)]
    ds.rir_files = None
    ds.speed_rates = None
    librosa.load = lambda path, sr: (bank[int(os.path.basename(path)[0])].copy(), sr)
    for s in range(6):
        random.seed(s)
        src = (0.1 * np.random.default_rng(500 + s).standard_normal(9000 + 1111 * s)).astype(np.float32)
        res, sr = ds.augment(src.copy(), 16000)
        assert sr == 16000
        out[f"chain_{s}"] = res
        out[f"chain_{s}_next_random"] = np.asarray(random.random())   # pins the number of draws
    # shipped probabilities (all gates drawn, nothing but noise can fire)
    with open("/root/reference/configs/augmentation.json") as f:
        ds.augment_configs = json.load(f)
    for s in range(4):
        random.seed(100 + s)
        src = (0.1 * np.random.default_rng(600 + s).standard_normal(8000)).astype(np.float32)
        res, sr = ds.augment(src.copy(), 16000)
        out[f"shipped_{s}"] = res
        out[f"shipped_{s}_next_random"] = np.asarray(random.random())
    np.savez_compressed(os.path.join(HERE, "augment_golden.npz"), **out)
    print("augment goldens:", len(out), "arrays")


if __name__ == "__main__":
    logmel_goldens()
    augment_goldens()
