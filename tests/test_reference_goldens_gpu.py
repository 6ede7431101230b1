"""GPU parity against outputs of the REFERENCE ITSELF (tests/golden/make_golden.py ``reverb`` /
``dataset``): wfe_reverb against add_reverb, the staged chain with the reverb entry firing
against augment(), CustomDataset -> collator against the reference's __getitem__ features,
labels and spec_augmentation.  Tolerances: log-mel 1e-4 (north_star); reverb waveform
3e-6 * max|y| * sqrt(taps / 64) (direct float32 sum vs scipy's float32 FFT)."""
import json
import os
import random

import numpy as np
import pytest
import torch

import cases
import stubs
from oracle import logmel as ol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4


@pytest.fixture(scope="module")
def g_reverb():
    return np.load(os.path.join(cases.GOLDEN, "reverb_golden.npz"))


@pytest.fixture(scope="module")
def g_dataset():
    with open(os.path.join(cases.GOLDEN, "dataset_golden.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(cases.GOLDEN, "dataset_golden.npz"))


def test_reverb_kernel_matches_reference_add_reverb(g_reverb):
    from wfe_amd.frontend import HipFrontEnd, NoiseBank, pack_ragged
    fe = HipFrontEnd(n_mel=80, device=0)
    tags = ("t64", "t801", "t1600", "t4000")
    rirs = [g_reverb[f"rir_{t}"] for t in tags]
    waves, index, want = [], [], []
    for j, t in enumerate(tags):
        for name in ("base", "odd"):
            waves.append(g_reverb[f"in_{name}"])
            index.append(j)
            want.append(g_reverb[f"reverb_{t}_{name}"])
    flat, off, lens = pack_ragged(waves, fe.device)
    out = fe.reverb(flat, lens, NoiseBank(rirs, fe.device), index, offsets=off)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    for i, (w, j) in enumerate(zip(want, index)):
        err = np.abs(out[i, :len(w)] - w).max()
        bound = 3e-6 * np.abs(w).max() * max(1.0, np.sqrt(len(rirs[j]) / 64))
        assert err <= bound, (tags[j], err, bound)
        assert not out[i, len(w):].any()


@pytest.mark.parametrize("n_mel", [80, 128])
def test_chain_with_reverb_matches_reference_augment(g_reverb, n_mel):
    """noise -> reverb -> speed -> shift -> volume all firing: the collator's staged execution
    (fused chain -> waveform, wfe_reverb, fused chain + log-mel) against the log-mel of the
    waveform the reference's augment() produced for the same random stream."""
    from wfe_amd import data as wd
    from wfe_amd.augment import AugmentPlanner
    cfg = json.loads(bytes(g_reverb["chain_config_json"]).decode())
    bank = [cases.noise_clip(j, 0.3 + 0.35 * j) for j in range(4)]
    rirs = [g_reverb[f"chain_rir_{j}"] for j in range(3)]
    clip_set = wd.ClipSet(bank, None, rirs, None)
    planner = AugmentPlanner(cfg, [len(c) for c in bank], n_rir_files=3)
    items, want = [], []
    for s in range(6):
        rnd = random.Random(40 + s)
        src = (0.1 * np.random.default_rng(700 + s).standard_normal(9000 + 1111 * s)).astype(np.float32)
        ops, n = planner.draw(len(src), 16000, rnd)
        assert rnd.random() == float(g_reverb[f"chain_{s}_next_random"]) and n == len(g_reverb[f"chain_{s}"])
        items.append({"waveform": src, "aug_ops": ops, "labels": [1], "id": str(s), "sample_rate": 16000,
                      "target_rate": 16000, "aug_bank": clip_set})
        want.append(g_reverb[f"chain_{s}"])
    coll = wd.DataCollatorSpeechSeq2SeqWithPadding(processor=stubs.StubProcessor(n_mel))
    got = coll(items)["input_features"].numpy()
    ref = ol.logmel_f32(want, n_mel=n_mel)
    err = np.abs(got - ref).reshape(len(items), -1).max(axis=1)
    print("chain with reverb, log-mel max-abs-err vs reference waveforms:", err)
    assert (err <= TOL).all(), err


@pytest.mark.parametrize("mode", ["text", "timestamps"])
def test_dataset_items_match_reference_getitem(g_dataset, monkeypatch, mode):
    """Reference-written binary manifest -> CustomDataset -> eval collator (prompt kept): features
    against the reference's processor output for the same records, labels token for token."""
    from wfe_amd import data as wd
    meta, arrays = g_dataset
    monkeypatch.chdir(ROOT)
    keep = cases.kept_frames(3000)
    for n_mel in (80, 128):
        proc = stubs.StubProcessor(n_mel)
        ds = wd.CustomDataset("tests/golden/manifest.header", proc, timestamps=(mode == "timestamps"), language="en")
        coll = wd.DataCollatorSpeechSeq2SeqWithPaddingforEval(processor=proc)
        batch = coll([ds[i] for i in range(len(ds))])
        want = meta[f"{mode}_{n_mel}_header"]
        assert batch["id"] == [w["id"] for w in want]
        feats = batch["input_features"].numpy()
        for i, w in enumerate(want):
            row = batch["labels"][i].tolist()
            assert row[:len(w["labels"])] == w["labels"] and set(row[len(w["labels"]):]) <= {-100}
            err = np.abs(feats[i][:, keep] - arrays[f"{mode}_{n_mel}_{i}"]).max()
            assert err <= TOL, (mode, n_mel, i, err)


def test_specaug_through_collator_matches_reference(g_dataset, monkeypatch):
    """The reference's own spec_augmentation output for 8 items drawn from one random stream: the
    items whose gate fired are zero everywhere (its quirk on the (1, M, 3000) layout), the others
    are the plain log-mel."""
    from wfe_amd import data as wd
    meta, arrays = g_dataset
    monkeypatch.chdir(ROOT)
    keep = cases.kept_frames(3000)
    proc = stubs.StubProcessor(80)
    ds = wd.CustomDataset("tests/golden/manifest.jsonl", proc, language="en",
                          augment_config_path=os.path.join(cases.GOLDEN, "specaug_config.json"))
    coll = wd.DataCollatorSpeechSeq2SeqWithPadding(processor=proc)
    random.seed(77)
    items = [ds[rep % len(ds)] for rep in range(8)]
    assert random.random() == meta["specaug_item_next_random"]
    got = coll(items)["input_features"].numpy()
    for rep in range(8):
        want = arrays[f"specaug_item_{rep}"]
        if meta["specaug_item_fired"][rep]:
            assert not want.any() and not got[rep].any()
        else:
            assert np.abs(got[rep][:, keep] - want).max() <= TOL, rep


@pytest.fixture(scope="module")
def g_speed2():
    return np.load(os.path.join(cases.GOLDEN, "speed2_golden.npz"))


def test_two_speed_entries_match_reference_augment(g_speed2):
    """A config with two speed entries (utils/reader.py:248-254 applies every one): the staged execution --
    [chain with the first speed change -> waveform] then [chain with the second -> waveform / log-mel] --
    against the reference's own augment() outputs.  No noise in this chain, so the waveform is BIT-EXACT;
    the collator's log-mel of it is within the log-mel tolerance."""
    from wfe_amd import data as wd
    from wfe_amd.augment import AugmentPlanner, split_stages
    from wfe_amd.frontend import HipFrontEnd
    cfg = json.loads(bytes(g_speed2["config_json"]).decode())
    planner = AugmentPlanner(cfg)
    fe = HipFrontEnd(n_mel=80, device=0)
    items, want = [], []
    for s in range(8):
        src = (0.1 * np.random.default_rng(800 + s).standard_normal(9000 + 1111 * s)).astype(np.float32)
        ops, n = planner.draw(len(src), 16000, random.Random(60 + s))
        w = g_speed2[f"chain_{s}"]
        assert n == len(w)
        # the waveform, stage by stage, through the same entry point the collator's staging uses
        wav = torch.from_numpy(src).to(fe.device)[None]
        lens = torch.tensor([len(src)], dtype=torch.int32, device=fe.device)
        for st in split_stages(ops, len(src), 16000, 16000):
            assert st["post"] in (None, ("pass", 0))
            if st["ops"]:
                wav, lens = fe.augment_waveform(wav, lens, [st["ops"]], None, n_samples=(st["len_out"] + 3) // 4 * 4)
        torch.cuda.synchronize()
        assert int(lens[0]) == len(w)
        got = wav[0, :len(w)].cpu().numpy()
        assert np.array_equal(got, w), (s, ops, np.abs(got - w).max())
        items.append({"waveform": src, "aug_ops": ops, "labels": [1], "id": str(s), "sample_rate": 16000, "target_rate": 16000})
        want.append(w)
    coll = wd.DataCollatorSpeechSeq2SeqWithPadding(processor=stubs.StubProcessor(80))
    got = coll(items)["input_features"].numpy()
    ref = ol.logmel_f32(want, n_mel=80)
    err = np.abs(got - ref).reshape(len(items), -1).max(axis=1)
    assert (err <= TOL).all(), err
