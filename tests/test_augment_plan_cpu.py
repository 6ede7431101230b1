"""Host-side augmentation planner (no GPU): the draw order must replay the
reference's ``random`` stream exactly (goldens produced by the reference's own
CustomDataset.augment), and the op list must describe what the reference did."""
import json
import random

import numpy as np
import pytest

import cases
from oracle import augment as oa
from wfe_amd import _lib
from wfe_amd.augment import AugmentPlanner


def _bank():
    return [cases.noise_clip(j, 0.3 + 0.35 * j) for j in range(4)]


def test_planner_replays_reference_draws(golden_augment):
    g = golden_augment
    cfg = json.loads(bytes(g["chain_config_json"]).decode())
    bank = _bank()
    planner = AugmentPlanner(cfg, [len(b) for b in bank])
    for s in range(6):
        n = 9000 + 1111 * s
        rnd = random.Random(s)
        ops, final_len = planner.draw(n, 16000, rnd)
        # the reference consumed exactly the same number of draws
        assert rnd.random() == float(g[f"chain_{s}_next_random"]), s
        assert final_len == g[f"chain_{s}"].shape[0], s
        # and the oracle (bit-exact with the reference) made the same decisions
        src = (0.1 * np.random.default_rng(500 + s).standard_normal(n)).astype(np.float32)
        _, _, trace = oa.augment(src.copy(), 16000, cfg, random.Random(s), bank)
        kinds = {"noise": _lib.AUG_NOISE, "speed": _lib.AUG_SPEED, "shift": _lib.AUG_SHIFT,
                 "volume": _lib.AUG_VOLUME}
        want = [t for t in trace if not (t[0] == "speed" and t[1] == 1.0) and not (t[0] == "shift" and int(t[1] * 16) == 0)]
        assert [o[0] for o in ops] == [kinds[t[0]] for t in want], (s, ops, trace)
        for o, t in zip(ops, want):
            if t[0] == "noise":
                assert (o[1], o[3]) == (t[1], t[2])
            if t[0] == "shift":
                assert o[1] == oa.shift_samples(t[1], 16000)
            if t[0] == "volume":
                assert o[3] == t[1] and np.float32(o[4]) == np.float32(10. ** (t[1] / 20.))


def test_shipped_config_draws_every_gate(golden_augment):
    g = golden_augment
    shipped = [
        {"type": "resample", "params": {"new_sample_rates": [8000, 32000, 44100]}, "prob": 0.0},
        {"type": "noise", "params": {"min_snr_dB": 10, "max_snr_dB": 50}, "prob": 0.5},
        {"type": "reverb", "params": {}, "prob": 0.5},
        {"type": "speed", "params": {"min_speed_rate": 0.9, "max_speed_rate": 1.1, "num_rates": 3}, "prob": 0.0},
        {"type": "shift", "params": {"min_shift_ms": -5, "max_shift_ms": 5}, "prob": 0.0},
        {"type": "volume", "params": {"min_gain_dBFS": -15, "max_gain_dBFS": 15}, "prob": 0.0},
        {"type": "specaug", "params": {}, "prob": 0.5},
    ]
    planner = AugmentPlanner(shipped, [len(b) for b in _bank()])
    for s in range(4):
        rnd = random.Random(100 + s)
        planner.draw(8000, 16000, rnd)
        assert rnd.random() == float(g[f"shipped_{s}_next_random"]), s


def test_errors_match_reference():
    p = AugmentPlanner([{"type": "shift", "params": {"min_shift_ms": 2000, "max_shift_ms": 2000}, "prob": 1.0}])
    with pytest.raises(ValueError):
        p.draw(16000, 16000, random.Random(0))            # 2 s shift of a 1 s clip (reader.py:298-299)
    p = AugmentPlanner([{"type": "speed", "params": {"min_speed_rate": 0.0, "max_speed_rate": 0.0, "num_rates": 1},
                         "prob": 1.0}])
    with pytest.raises(ValueError):
        p.draw(16000, 16000, random.Random(0))            # reader.py:285-286
    from wfe_amd.augment import HOST_REVERB
    p = AugmentPlanner([{"type": "reverb", "params": {}, "prob": 1.0}], n_rir_files=2)
    r = random.Random(0)
    r.random()
    assert p.draw(16000, 16000, random.Random(0)) == ([(HOST_REVERB, r.choice(range(2)), 0, 0, 0.0)], 16000)
    assert AugmentPlanner([{"type": "reverb", "params": {}, "prob": 1.0}]).draw(16000, 16000, random.Random(0)) == ([], 16000)


def test_resample_entry_splits_the_chain_into_stages():
    """reader.py:263-267: np.random.choice(new_sample_rates) (numpy's RNG, one draw),
    the rate changes for the entries that follow (shift in samples of the NEW rate,
    noise clips as librosa.load(path, sr=new rate) would return them), and
    _get_list_data returns to 16 kHz at the end (:157-158)."""
    from wfe_amd import _lib
    from wfe_amd.augment import HOST_RESAMPLE, HOST_REVERB, resampled_length, split_stages
    cfg = [
        {"type": "volume", "params": {"min_gain_dBFS": 3, "max_gain_dBFS": 3}, "prob": 1.0},
        {"type": "resample", "params": {"new_sample_rates": [8000, 32000, 44100]}, "prob": 1.0},
        {"type": "noise", "params": {"min_snr_dB": 10, "max_snr_dB": 10}, "prob": 1.0},
        {"type": "shift", "params": {"min_shift_ms": 5, "max_shift_ms": 5}, "prob": 1.0},
    ]
    p = AugmentPlanner(cfg, noise_lengths=[480000], noise_rates=[48000])
    g = np.random.default_rng(5)
    want_rate = int(np.random.default_rng(5).choice([8000, 32000, 44100]))
    rnd = random.Random(1)
    ops, n = p.draw(16000, 16000, rnd, g)
    assert n == resampled_length(16000, 16000, want_rate)
    assert [o[0] for o in ops] == [_lib.AUG_VOLUME, HOST_RESAMPLE, _lib.AUG_NOISE, _lib.AUG_SHIFT]
    assert ops[1][1] == want_rate and ops[3][1] == int(5 * want_rate / 1000)
    # the 10 s / 48 kHz noise clip is longer than the utterance at any of the rates: a start offset was drawn
    r = random.Random(1)
    r.random(), r.randint(3, 3)            # volume gate + gain
    r.random()                             # resample gate (the rate comes from numpy's RNG)
    r.random(), r.choice(range(1)), r.randint(10, 10)      # noise gate, clip, SNR
    assert ops[2][2] == r.randint(0, resampled_length(480000, 48000, want_rate) - n)
    st = split_stages(ops, 16000, 16000, 16000)
    assert [(s["sr"], s["post"], len(s["ops"])) for s in st] == [(16000, ("resample", want_rate), 1),
                                                                 (want_rate, ("resample", 16000), 2), (16000, None, 0)]
    assert st[1]["len_in"] == n and st[2]["len_in"] == resampled_length(n, want_rate, 16000)
    # a 44.1 kHz file without augmentation: one resampling stage, then the log-mel stage
    st = split_stages([], 44100, 44100, 16000)
    assert [(s["sr"], s["len_in"], s["post"]) for s in st] == [(44100, 44100, ("resample", 16000)), (16000, 16000, None)]
    # a reverb entry ends a stage without changing rate or length
    st = split_stages([(_lib.AUG_VOLUME, 0, 0, 3, 1.4), (HOST_REVERB, 1, 0, 0, 0.0), (_lib.AUG_SHIFT, 8, 0, 0, 0.0)],
                      500, 16000, 16000)
    assert [(s["sr"], s["post"], len(s["ops"]), s["len_in"]) for s in st] == [(16000, ("reverb", 1), 1, 500),
                                                                              (16000, None, 1, 500)]
    # same-rate choice is not a stage boundary
    p2 = AugmentPlanner([{"type": "resample", "params": {"new_sample_rates": [16000]}, "prob": 1.0}])
    assert p2.draw(1000, 16000, random.Random(0), np.random.default_rng(0)) == ([], 1000)
