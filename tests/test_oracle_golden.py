"""Pins the CPU oracle (oracle/) to the golden vectors: outputs of the installed
transformers.WhisperFeatureExtractor and of the reference's utils/reader.py
(tests/golden/make_golden.py), plus the reference's own convert-ggml/mel_filters.npz."""
import json
import os
import random

import numpy as np
import pytest

import cases
from oracle import augment as oa
from oracle import logmel as ol

LOGMEL_TOL = 3e-5     # fp32 torch.stft restatement vs the HF fp32 torch path (same kernels)
F64_TOL = 5e-5        # float64 restatement vs HF fp32 (SURVEY: 0.95e-5 / 1.35e-5 on test.wav)


def test_mel_bank_matches_hf_and_reference_npz(golden_logmel):
    ref = np.load(os.path.join(cases.GOLDEN, "mel_filters.npz"))
    for m in (80, 128):
        bank = ol.mel_filter_bank(m)
        assert bank.shape == (201, m) and bank.dtype == np.float64
        assert np.array_equal(bank, golden_logmel[f"mel_filters_{m}"])
        assert np.abs(bank.T - ref[f"mel_{m}"]).max() < 1e-8


def test_reflect_index_matches_numpy_pad():
    for n in (1000, 480000):
        x = np.arange(n, dtype=np.float64)
        want = np.pad(x, 200, mode="reflect")
        got = x[ol.reflect_source_index(np.arange(n + 400), n)]
        assert np.array_equal(want, got)


@pytest.mark.parametrize("n_mel", [80, 128])
def test_logmel_f32_matches_hf(golden_logmel, n_mel):
    keep = cases.kept_frames(3000)
    inputs = cases.logmel_cases()
    names = list(inputs)
    got = ol.logmel_f32([inputs[n] for n in names], n_mel=n_mel)
    assert got.shape == (len(names), n_mel, 3000) and got.dtype == np.float32
    for i, n in enumerate(names):
        err = np.abs(got[i][:, keep] - golden_logmel[f"{n}_{n_mel}"]).max()
        assert err <= LOGMEL_TOL, (n, err)
        # batched HF call == per-utterance HF call (max is per utterance)
        if f"batched_{n}_{n_mel}" in golden_logmel:
            err = np.abs(got[i][:, keep] - golden_logmel[f"batched_{n}_{n_mel}"]).max()
            assert err <= LOGMEL_TOL, ("batched", n, err)


@pytest.mark.parametrize("name", ["test_wav", "synth0", "short_5p3s", "zeros", "tiny_100",
                                  "single_sample", "long_31p3s", "end_479999"])
def test_logmel_f64_matches_hf(golden_logmel, name):
    keep = cases.kept_frames(3000)
    wav = cases.logmel_cases()[name]
    for n_mel in (80, 128):
        got = ol.logmel_f64(wav, n_mel=n_mel)
        err = np.abs(got[:, keep] - golden_logmel[f"{name}_{n_mel}"]).max()
        assert err <= F64_TOL, (name, n_mel, err)


def test_f64_path_matches_hf_numpy_path(golden_logmel):
    keep = cases.kept_frames(3000)
    got = ol.logmel_f64(cases.load_test_wav(), n_mel=80)
    assert np.abs(got[:, keep] - golden_logmel["np_path_test_wav_80"]).max() < 2e-6


def test_normalize_and_attention_mask(golden_logmel):
    keep = cases.kept_frames(3000)
    wav = cases.load_test_wav()
    for n_mel in (80, 128):
        got = ol.logmel_f32([wav], n_mel=n_mel, do_normalize=True)[0]
        assert np.abs(got[:, keep] - golden_logmel[f"normalized_test_wav_{n_mel}"]).max() <= LOGMEL_TOL
        mask = ol.frame_attention_mask(wav.shape[0])
        assert np.array_equal(mask, golden_logmel[f"attention_mask_test_wav_{n_mel}"])


def test_padding_longest(golden_logmel):
    inputs = cases.logmel_cases()
    a, b = inputs["short_5p3s"], inputs["tiny_100"]
    longest = max(a.shape[0], b.shape[0])
    for n_mel in (80, 128):
        got = ol.logmel_f32([a, b], n_mel=n_mel, n_samples=longest)
        assert got.shape == golden_logmel[f"longest_short_5p3s_{n_mel}"].shape[:0] + (2,) + got.shape[1:]
        assert got.shape[2] == longest // 160
        assert np.abs(got[0] - golden_logmel[f"longest_short_5p3s_{n_mel}"]).max() <= LOGMEL_TOL
        assert np.abs(got[1] - golden_logmel[f"longest_tiny_100_{n_mel}"]).max() <= LOGMEL_TOL
        m = np.stack([ol.frame_attention_mask(a.shape[0], longest), ol.frame_attention_mask(b.shape[0], longest)])
        assert np.array_equal(m, golden_logmel[f"longest_mask_{n_mel}"])


# ------------------------------------------------------------------ augmentation
def test_speed_shift_volume_rms_bit_exact(golden_augment):
    g = golden_augment
    for src in ("base", "odd"):
        x = g[f"in_{src}"]
        for rate in (0.9, 1.1, 0.95, 1.0):
            tag = str(rate).replace(".", "p")
            got = oa.change_speed(x.copy(), rate)
            assert got.dtype == np.float32 and np.array_equal(got, g[f"speed_{tag}_{src}"]), (src, rate)
    x = g["in_base"]
    assert oa.change_speed(x, 0.9).shape[0] == 17777 and oa.change_speed(x, 1.1).shape[0] == 14545
    for ms in (-5, -1, 0, 3, 5, 250):
        assert np.array_equal(oa.shift(x.copy(), 16000, ms), g[f"shift_{ms}_base"]), ms
    for gain in (-15, -3, 0, 7, 15):
        assert np.array_equal(oa.volume(x.copy(), gain), g[f"volume_{gain}_base"]), gain
    assert np.array_equal(np.asarray(oa.rms_db(x)), g["rms_db_base"])
    assert np.array_equal(np.asarray(oa.rms_db(g["in_odd"])), g["rms_db_odd"])
    with pytest.raises(ValueError):
        oa.shift(x.copy(), 16000, 1001)
    with pytest.raises(ValueError):
        oa.change_speed(x.copy(), 0.0)


def test_noise_mixing_bit_exact(golden_augment):
    g = golden_augment
    x = g["in_base"]
    for tag, nz in (("short", cases.noise_clip(0, 0.4)), ("long", cases.noise_clip(1, 2.5))):
        for snr in (10, 33, 50):
            rnd = random.Random(123)
            got = oa.add_noise(x.copy(), nz, snr, rnd.randint)
            assert np.array_equal(got, g[f"noise_{tag}_{snr}"]), (tag, snr)


def test_chain_draw_order_and_result_bit_exact(golden_augment):
    g = golden_augment
    cfg = json.loads(bytes(g["chain_config_json"]).decode())
    bank = [cases.noise_clip(j, 0.3 + 0.35 * j) for j in range(4)]
    for s in range(6):
        rnd = random.Random(s)
        src = (0.1 * np.random.default_rng(500 + s).standard_normal(9000 + 1111 * s)).astype(np.float32)
        got, sr, trace = oa.augment(src.copy(), 16000, cfg, rnd, bank)
        assert sr == 16000 and [t[0] for t in trace] == ["noise", "speed", "shift", "volume"]
        assert np.array_equal(got, g[f"chain_{s}"]), s
        assert rnd.random() == float(g[f"chain_{s}_next_random"]), s


def test_shipped_probabilities_draw_every_gate(golden_augment):
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
    bank = [cases.noise_clip(j, 0.3 + 0.35 * j) for j in range(4)]
    for s in range(4):
        rnd = random.Random(100 + s)
        src = (0.1 * np.random.default_rng(600 + s).standard_normal(8000)).astype(np.float32)
        got, _, _ = oa.augment(src.copy(), 16000, shipped, rnd, bank)
        assert np.array_equal(got, g[f"shipped_{s}"]), s
        assert rnd.random() == float(g[f"shipped_{s}_next_random"]), s
