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
from transformers import WhisperFeatureExtractor  # noqa: F401  (before the librosa stand-in exists: its
                                                  # availability probe rejects a module without a spec)

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


def _reference_dataset_class():
    """The reference's CustomDataset, imported with empty stand-in modules for the two
    packages that are not installed; the few functions of them the executed code reaches
    are set by the callers below (decode of dataset/test.wav through the stdlib ``wave``
    module = what soundfile returns for 16-bit PCM; seeded clips for librosa.load)."""
    for missing in ("librosa", "soundfile"):
        if missing not in sys.modules:
            sys.modules[missing] = types.ModuleType(missing)
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    from utils.reader import CustomDataset
    return CustomDataset


def reverb_goldens():
    """CustomDataset.add_reverb (utils/reader.py:342-348) and the whole augment() chain with the
    reverb entry firing, run by the reference itself; librosa.load returns seeded impulse
    responses / noise clips (decode is outside the pinned path)."""
    CustomDataset = _reference_dataset_class()
    import librosa
    out = {}
    rng = np.random.default_rng(4242)
    base = (0.1 * rng.standard_normal(16000)).astype(np.float32)
    odd = (0.1 * rng.standard_normal(12347)).astype(np.float32)
    out["in_base"], out["in_odd"] = base, odd
    ds = object.__new__(CustomDataset)
    for tag, taps in (("t64", 64), ("t801", 801), ("t1600", 1600), ("t4000", 4000)):
        rir = cases.rir_clip(taps, taps)
        out[f"rir_{tag}"] = rir
        librosa.load = lambda path, sr, _r=rir: (_r.copy(), sr)
        for name, x in (("base", base), ("odd", odd)):
            y = ds.add_reverb(x.copy(), 16000, "rir.wav")
            assert y.dtype == np.float32 and y.shape == x.shape
            out[f"reverb_{tag}_{name}"] = y
    # the chain in shipped file order with noise, reverb, speed, shift, volume all firing
    with open("/root/reference/configs/augmentation.json") as f:
        cfg = json.load(f)
    for c in cfg:
        if c["type"] in ("speed", "shift", "volume", "noise", "reverb"):
            c["prob"] = 1.0
    out["chain_config_json"] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
    bank = [cases.noise_clip(j, 0.3 + 0.35 * j) for j in range(4)]
    rirs = [cases.rir_clip(100 + j, 300 + 250 * j) for j in range(3)]
    for j, r in enumerate(rirs):
        out[f"chain_rir_{j}"] = r
    ds.augment_configs = cfg
    ds.noise_files = [f"n{j}.wav" for j in range(4)]
    ds.rir_files = [f"r{j}.wav" for j in range(3)]
    ds.speed_rates = None

    def load(path, sr):
        name = os.path.basename(path)
        return ((bank if name[0] == "n" else rirs)[int(name[1])].copy(), sr)
    librosa.load = load
    for s in range(6):
        random.seed(40 + s)
        src = (0.1 * np.random.default_rng(700 + s).standard_normal(9000 + 1111 * s)).astype(np.float32)
        res, sr = ds.augment(src.copy(), 16000)
        assert sr == 16000 and res.dtype == np.float32
        out[f"chain_{s}"] = res
        out[f"chain_{s}_next_random"] = np.asarray(random.random())
    np.savez_compressed(os.path.join(HERE, "reverb_golden.npz"), **out)
    print("reverb goldens:", len(out), "arrays")


SPEED2_CONFIG = [
    {"type": "speed", "params": {"min_speed_rate": 0.9, "max_speed_rate": 1.1, "num_rates": 3}, "prob": 1.0},
    {"type": "shift", "params": {"min_shift_ms": -5, "max_shift_ms": 5}, "prob": 1.0},
    # a second speed entry (utils/reader.py:248-254 applies every one; its rates are the FIRST entry's:
    # self.speed_rates is built once)
    {"type": "speed", "params": {"min_speed_rate": 0.5, "max_speed_rate": 0.7, "num_rates": 2}, "prob": 1.0},
    {"type": "volume", "params": {"min_gain_dBFS": -15, "max_gain_dBFS": 15}, "prob": 1.0},
]


def speed2_goldens():
    """A config with TWO speed entries, run by the reference's own augment() (no noise / reverb: the
    chain is index arithmetic and single float32 roundings, so the outputs pin the GPU chain bit for bit)."""
    CustomDataset = _reference_dataset_class()
    out = {"config_json": np.frombuffer(json.dumps(SPEED2_CONFIG).encode(), dtype=np.uint8)}
    ds = object.__new__(CustomDataset)
    ds.augment_configs = json.loads(json.dumps(SPEED2_CONFIG))
    ds.noise_files, ds.rir_files, ds.speed_rates = [], [], None
    n_two = 0
    for s in range(8):
        random.seed(60 + s)
        src = (0.1 * np.random.default_rng(800 + s).standard_normal(9000 + 1111 * s)).astype(np.float32)
        res, sr = ds.augment(src.copy(), 16000)
        assert sr == 16000 and res.dtype == np.float32
        out[f"chain_{s}"] = res
        out[f"chain_{s}_next_random"] = np.asarray(random.random())
        n_two += int(len(res) != len(src))
    assert list(ds.speed_rates) == [0.9, 1.0, 1.1]
    np.savez_compressed(os.path.join(HERE, "speed2_golden.npz"), **out)
    print("two-speed goldens:", len(out), "arrays;", n_two, "of 8 change length")


def dataset_goldens():
    """The reference's CustomDataset end to end on dataset/test.wav: manifests written by the
    reference's own DatasetWriter (utils/binary.py:9-35), items from its __getitem__
    (timestamp labels reader.py:164-179, the no-text branch :196-199, slices :222-243) and its
    spec_augmentation (:355-389) on the real (1, n_mel, 3000) layout.  Tokenizer = tests/stubs.py
    (no Whisper vocabulary exists offline): the label ARITHMETIC is the reference's."""
    import wave
    import stubs
    CustomDataset = _reference_dataset_class()
    import librosa
    import soundfile
    from utils.binary import DatasetWriter

    wav_rel = "tests/golden/test.wav"                      # manifest paths are relative to the repository root
    root = os.path.dirname(os.path.dirname(HERE))

    class SoundFile:                                       # the four members slice_from_file uses
        def __init__(self, path):
            self._w = wave.open(os.path.join(root, path))
            self.samplerate = self._w.getframerate()
        def __len__(self):
            return self._w.getnframes()
        def seek(self, frame):
            self._w.setpos(frame)
        def read(self, frames, dtype):
            assert dtype == "float32" and self._w.getsampwidth() == 2 and self._w.getnchannels() == 1
            return np.frombuffer(self._w.readframes(frames), dtype="<i2").astype(np.float32) / np.float32(32768.0)

    def sf_read(path, dtype):
        f = SoundFile(path)
        return f.read(len(f), dtype), f.samplerate
    soundfile.read, soundfile.SoundFile = sf_read, SoundFile
    librosa.to_mono = lambda y: y if y.ndim == 1 else np.mean(y, axis=tuple(range(y.ndim - 1)))

    segs = [{"start": 0.0, "end": 1.51, "text": "ab"}, {"start": 1.53, "end": 2.9, "text": "c"},
            {"start": 3.07, "end": 8.39, "text": "whisper"}]
    recs = [
        {"audio": {"path": wav_rel}, "sentence": "hello world", "duration": 8.39, "sentences": segs},
        {"audio": {"path": wav_rel}, "sentence": "too short", "duration": 0.3, "sentences": segs[:1]},
        {"audio": {"path": wav_rel, "start_time": 1.2345, "end_time": 4.5678}, "sentence": "a slice", "duration": 3.33,
         "sentences": segs[:2], "language": "zh", "id": "slice-1"},
        {"audio": {"path": wav_rel}, "sentence": "", "duration": 8.39, "sentences": []},
        {"audio": {"path": wav_rel}, "sentence": "too long", "duration": 31.0, "sentences": segs},
        {"audio": {"path": wav_rel, "start_time": -3.0, "end_time": -0.5}, "sentence": "from the end", "duration": 2.5,
         "sentences": [{"start": 0.01, "end": 2.49, "text": "z"}]},
    ]
    writer = DatasetWriter(os.path.join(HERE, "manifest"))          # -> manifest.data / manifest.header
    for r in recs:
        writer.add_data(json.dumps(r))
    writer.close()
    with open(os.path.join(HERE, "manifest.jsonl"), "w", encoding="utf-8") as f:
        f.write("\n".join(json.dumps(r) for r in recs))

    cwd = os.getcwd()
    os.chdir(root)
    try:
        meta, arrays = {}, {}
        keep = cases.kept_frames(3000)
        for n_mel in (80, 128):
            for mode, ts in (("text", False), ("timestamps", True)):
                for src in ("header", "jsonl"):
                    ds = CustomDataset(os.path.join("tests/golden", "manifest." + src), stubs.StubProcessor(n_mel),
                                       timestamps=ts, language="en")
                    items = []
                    for i in range(len(ds)):
                        it = ds[i]
                        feats = np.asarray(it["input_features"])
                        assert feats.shape == (1, n_mel, 3000) and feats.dtype == np.float32
                        items.append({"id": it["id"], "labels": [int(t) for t in it["labels"]]})
                        if src == "header":
                            arrays[f"{mode}_{n_mel}_{i}"] = feats[0][:, keep]
                    meta[f"{mode}_{n_mel}_{src}"] = items
        # spec_augmentation on the real layout, called directly and through __getitem__
        for n_mel in (80, 128):
            ds = CustomDataset("tests/golden/manifest.jsonl", stubs.StubProcessor(n_mel), language="en")
            feats = np.asarray(ds[0]["input_features"])
            for s in range(3):
                random.seed(900 + s)
                got = ds.spec_augmentation({"input_features": feats.copy()})["input_features"]
                assert tuple(got.shape) == (1, n_mel, 3000)
                arrays[f"specaug_direct_{n_mel}_{s}"] = got.numpy()
                meta[f"specaug_direct_{n_mel}_{s}_next_random"] = random.random()
        cfgp = os.path.join(HERE, "specaug_config.json")
        with open(cfgp, "w") as f:
            json.dump([{"type": "volume", "params": {"min_gain_dBFS": -15, "max_gain_dBFS": 15}, "prob": 0.0},
                       {"type": "specaug", "params": {"num_t_mask": 2, "num_f_mask": 2, "max_t": 50, "max_f": 10},
                        "prob": 0.5}], f)
        ds = CustomDataset("tests/golden/manifest.jsonl", stubs.StubProcessor(80), language="en",
                           augment_config_path=cfgp)
        random.seed(77)
        fired = []
        for rep in range(8):
            it = ds[rep % len(ds)]
            f = it["input_features"]
            is_tensor = not isinstance(f, np.ndarray)
            fired.append(bool(is_tensor))
            arr = f.numpy() if is_tensor else np.asarray(f)
            arrays[f"specaug_item_{rep}"] = arr[0][:, keep]
        meta["specaug_item_fired"] = fired
        meta["specaug_item_next_random"] = random.random()
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, "dataset_golden.npz"), **arrays)
    with open(os.path.join(HERE, "dataset_golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("dataset goldens:", len(arrays), "arrays,", len(meta), "records")


if __name__ == "__main__":
    which = sys.argv[1:] or ["logmel", "augment", "reverb", "dataset", "speed2"]
    if "logmel" in which:
        logmel_goldens()
    if "augment" in which:
        augment_goldens()
    if "reverb" in which:
        reverb_goldens()
    if "dataset" in which:
        dataset_goldens()
    if "speed2" in which:
        speed2_goldens()
