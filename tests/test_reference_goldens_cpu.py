"""Pins the host logic and the CPU oracle to outputs of the REFERENCE ITSELF
(tests/golden/make_golden.py ``reverb`` and ``dataset``): add_reverb and the full chain with
the reverb entry firing (utils/reader.py:246-278,342-348), the manifests written by the
reference's DatasetWriter (utils/binary.py:9-35), item ids / labels of its __getitem__
(timestamp arithmetic :164-179, no-text branch :196-199, slices :222-243) and its
spec_augmentation (:355-389) on the real (1, n_mel, 3000) layout."""
import json
import os
import random

import numpy as np
import pytest

import cases
import stubs
from oracle import augment as oa
from wfe_amd import data as wd
from wfe_amd.augment import HOST_REVERB, AugmentPlanner

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def g_reverb():
    return np.load(os.path.join(cases.GOLDEN, "reverb_golden.npz"))


@pytest.fixture(scope="module")
def g_dataset():
    with open(os.path.join(cases.GOLDEN, "dataset_golden.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(cases.GOLDEN, "dataset_golden.npz"))


# scipy's float32 FFT may take another SIMD path on another host CPU: a few ulps at the peak
FFT_RTOL = 4e-7


def test_oracle_add_reverb_matches_reference(g_reverb):
    for tag in ("t64", "t801", "t1600", "t4000"):
        assert np.array_equal(g_reverb[f"rir_{tag}"], cases.rir_clip(int(tag[1:]), int(tag[1:])))
        for name in ("base", "odd"):
            x = g_reverb[f"in_{name}"]
            want = g_reverb[f"reverb_{tag}_{name}"]
            got = oa.add_reverb(x.copy(), g_reverb[f"rir_{tag}"])
            assert got.dtype == np.float32 and got.shape == want.shape
            assert np.abs(got - want).max() <= FFT_RTOL * np.abs(want).max(), (tag, name)
            assert np.isclose(np.abs(got).max(), np.abs(x).max(), rtol=1e-6)          # peak normalisation


def test_oracle_chain_with_reverb_matches_reference(g_reverb):
    cfg = json.loads(bytes(g_reverb["chain_config_json"]).decode())
    bank = [cases.noise_clip(j, 0.3 + 0.35 * j) for j in range(4)]
    rirs = [g_reverb[f"chain_rir_{j}"] for j in range(3)]
    fired = set()
    for s in range(6):
        rnd = random.Random(40 + s)
        src = (0.1 * np.random.default_rng(700 + s).standard_normal(9000 + 1111 * s)).astype(np.float32)
        got, sr, trace = oa.augment(src.copy(), 16000, cfg, rnd, bank, rir_bank=rirs)
        want = g_reverb[f"chain_{s}"]
        assert sr == 16000 and got.shape == want.shape, (s, trace)
        assert np.abs(got - want).max() <= 2 * FFT_RTOL * max(np.abs(want).max(), 1e-30), (s, trace)
        assert rnd.random() == float(g_reverb[f"chain_{s}_next_random"])             # same number of draws
        fired |= {t[0] for t in trace}
    assert {"noise", "reverb", "speed", "shift", "volume"} <= fired


def test_planner_draws_with_reverb_match_reference_stream(g_reverb):
    cfg = json.loads(bytes(g_reverb["chain_config_json"]).decode())
    bank = [cases.noise_clip(j, 0.3 + 0.35 * j) for j in range(4)]
    planner = AugmentPlanner(cfg, [len(c) for c in bank], n_rir_files=3)
    for s in range(6):
        rnd = random.Random(40 + s)
        ops, n = planner.draw(9000 + 1111 * s, 16000, rnd)
        assert n == len(g_reverb[f"chain_{s}"])
        assert any(op[0] == HOST_REVERB for op in ops)
        assert rnd.random() == float(g_reverb[f"chain_{s}_next_random"])


def test_binary_manifest_written_by_the_reference(monkeypatch):
    """tests/golden/manifest.{header,data}: written by the reference's DatasetWriter."""
    monkeypatch.chdir(ROOT)
    with open(os.path.join(cases.GOLDEN, "manifest.jsonl"), encoding="utf-8") as f:
        recs = [json.loads(line) for line in f]
    m = wd._BinaryManifest(os.path.join(cases.GOLDEN, "manifest.header"), 0.5, 30)
    kept = [r for r in recs if 0.5 <= r["duration"] <= 30]                            # reader.py:127-130 / binary.py:50-53
    assert len(m.keys) == len(kept) == 4
    assert [m.get(k) for k in m.keys] == kept
    everything = wd._BinaryManifest(os.path.join(cases.GOLDEN, "manifest.header"), 0.0, -1)
    assert [everything.get(k) for k in everything.keys] == recs


@pytest.mark.parametrize("src", ["header", "jsonl"])
@pytest.mark.parametrize("mode", ["text", "timestamps"])
def test_item_ids_and_labels_match_reference(g_dataset, monkeypatch, src, mode):
    meta, _ = g_dataset
    monkeypatch.chdir(ROOT)                                 # manifest paths are relative to the repository root
    for n_mel in (80, 128):
        ds = wd.CustomDataset(os.path.join("tests/golden", "manifest." + src), stubs.StubProcessor(n_mel),
                              timestamps=(mode == "timestamps"), language="en")
        want = meta[f"{mode}_{n_mel}_{src}"]
        assert len(ds) == len(want) == 4
        for i, w in enumerate(want):
            it = ds[i]
            assert it["id"] == w["id"], i
            assert [int(t) for t in it["labels"]] == w["labels"], (mode, i)
    # what the slices hand to the GPU: frames int(start * sr) .. int(end * sr) of reader.py:222-243
    it = ds[1]
    assert it["waveform"].shape[0] == int(round(4.5678, 3) * 16000) - int(round(1.2345, 3) * 16000)
    it = ds[3]                                              # negative times count from the end
    dur = round(134240 / 16000, 3)
    assert it["waveform"].shape[0] == int((dur - 0.5) * 16000) - int((dur - 3.0) * 16000)


def _apply_specaug(feat, sa):
    out = feat.copy()
    if sa["all"]:
        out[:] = 0
    for r0, r1 in sa["rows"]:
        out[r0:r1] = 0
    return out


def test_specaug_draws_and_result_match_reference(g_dataset, monkeypatch, tmp_path):
    meta, arrays = g_dataset
    monkeypatch.chdir(ROOT)
    cfgp = tmp_path / "sa.json"
    cfgp.write_text(json.dumps([{"type": "specaug", "params": {}, "prob": 0.5}]))
    for n_mel in (80, 128):
        ds = wd.CustomDataset("tests/golden/manifest.jsonl", stubs.StubProcessor(n_mel), language="en",
                              augment_config_path=str(cfgp))
        feat = np.ones((n_mel, 3000), np.float32)
        for s in range(3):
            random.seed(900 + s)
            sa = ds._draw_spec_augmentation()
            assert random.random() == meta[f"specaug_direct_{n_mel}_{s}_next_random"]   # same draws consumed
            want = arrays[f"specaug_direct_{n_mel}_{s}"]
            assert want.shape == (1, n_mel, 3000)
            assert np.array_equal(_apply_specaug(feat, sa), np.where(want[0] == 0, 0, 1).astype(np.float32) * feat)
            assert not want.any()                            # the quirk itself: the reference zeroes everything


def test_specaug_gate_sequence_matches_reference(g_dataset, monkeypatch):
    """__getitem__ with the shipped-style config (one augment gate + the specaug gate per item):
    which items fire and where the random stream stands afterwards, reader.py:201-207."""
    meta, _ = g_dataset
    monkeypatch.chdir(ROOT)
    ds = wd.CustomDataset("tests/golden/manifest.jsonl", stubs.StubProcessor(80), language="en",
                          augment_config_path=os.path.join(cases.GOLDEN, "specaug_config.json"))
    random.seed(77)
    fired = [bool(ds[rep % len(ds)]["specaug"]) for rep in range(8)]
    assert fired == meta["specaug_item_fired"] and any(fired) and not all(fired)
    assert random.random() == meta["specaug_item_next_random"]


# ---- a config with TWO speed entries (utils/reader.py:248-254), reference outputs in speed2_golden.npz
@pytest.fixture(scope="module")
def g_speed2():
    return np.load(os.path.join(cases.GOLDEN, "speed2_golden.npz"))


def test_two_speed_entries_oracle_and_planner_match_reference(g_speed2):
    """The oracle restatement reproduces the reference's augment() bit for bit on a config with two speed
    entries (the second entry draws from the FIRST entry's rates: speed_rates is built once), the planner
    consumes the same random stream and predicts the same lengths, and every chain with two speed changes
    is split into stages of at most one each."""
    from wfe_amd import _lib
    from wfe_amd.augment import split_stages
    cfg = json.loads(bytes(g_speed2["config_json"]).decode())
    planner = AugmentPlanner(cfg)
    speed_rates = None
    two = 0
    for s in range(8):
        src = (0.1 * np.random.default_rng(800 + s).standard_normal(9000 + 1111 * s)).astype(np.float32)
        want = g_speed2[f"chain_{s}"]
        rnd = random.Random(60 + s)
        if speed_rates is None:
            speed_rates = np.linspace(0.9, 1.1, 3, endpoint=True)        # reader.py:249-252: from the first entry
        got, _, _ = oa.augment(src.copy(), 16000, cfg, rnd, [], speed_rates=speed_rates)
        assert np.array_equal(got, want), s
        assert rnd.random() == float(g_speed2[f"chain_{s}_next_random"])
        rnd = random.Random(60 + s)
        ops, n = planner.draw(len(src), 16000, rnd)
        assert rnd.random() == float(g_speed2[f"chain_{s}_next_random"]) and n == len(want)
        stages = split_stages(ops, len(src), 16000, 16000)
        n_speed = sum(1 for o in ops if o[0] == _lib.AUG_SPEED)
        assert len(stages) == max(1, n_speed)
        assert all(sum(1 for o in st["ops"] if o[0] == _lib.AUG_SPEED) <= 1 for st in stages)
        assert stages[0]["len_in"] == len(src) and stages[-1]["len_out"] == len(want) and stages[-1]["post"] is None
        assert all(st["post"] == ("pass", 0) for st in stages[:-1])
        two += int(n_speed == 2)
    assert two >= 2
