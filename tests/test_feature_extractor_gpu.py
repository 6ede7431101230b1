"""GPU: the drop-in extractor's __call__ against the reference extractor's golden
outputs, and (when transformers is importable on the box) against the
reference class itself on the same inputs."""
import numpy as np
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def fe80():
    from wfe_amd.feature_extractor import MI355XWhisperFeatureExtractor
    return MI355XWhisperFeatureExtractor(feature_size=80)


def test_call_contract(fe80, golden_logmel):
    keep = cases.kept_frames(3000)
    wav = cases.load_test_wav()
    out = fe80(wav, sampling_rate=16000)
    f = out["input_features"]
    assert isinstance(f, np.ndarray) and f.shape == (1, 80, 3000) and f.dtype == np.float32   # always batched
    assert np.abs(f[0][:, keep] - golden_logmel["test_wav_80"]).max() <= TOL
    out = fe80(wav.astype(np.float64).tolist(), sampling_rate=16000, return_tensors="pt")      # list[float] input
    assert isinstance(out["input_features"], torch.Tensor) and out["input_features"].device.type == "cpu"
    assert np.abs(out["input_features"][0].numpy()[:, keep] - golden_logmel["test_wav_80"]).max() <= TOL
    r = fe80(wav, sampling_rate=16000, do_normalize=True, return_attention_mask=True)
    assert np.abs(r["input_features"][0][:, keep] - golden_logmel["normalized_test_wav_80"]).max() <= TOL
    assert np.array_equal(r["attention_mask"][0], golden_logmel["attention_mask_test_wav_80"])


def test_batched_ragged_and_longest(fe80, golden_logmel):
    keep = cases.kept_frames(3000)
    inputs = cases.logmel_cases()
    names = ["short_5p3s", "tone440", "zeros", "hdr_burst"]
    f = fe80([inputs[n] for n in names], sampling_rate=16000)["input_features"]
    for i, n in enumerate(names):
        assert np.abs(f[i][:, keep] - golden_logmel[f"batched_{n}_80"]).max() <= TOL, n
    r = fe80([inputs["short_5p3s"], inputs["tiny_100"]], sampling_rate=16000, padding="longest",
             return_attention_mask=True)
    assert r["input_features"].shape == golden_logmel["longest_short_5p3s_80"][None].repeat(2, 0).shape
    assert np.abs(r["input_features"][0] - golden_logmel["longest_short_5p3s_80"]).max() <= TOL
    assert np.abs(r["input_features"][1] - golden_logmel["longest_tiny_100_80"]).max() <= TOL
    assert np.array_equal(r["attention_mask"], golden_logmel["longest_mask_80"])


def test_device_output_stays_on_gpu():
    from wfe_amd.feature_extractor import MI355XWhisperFeatureExtractor
    fe = MI355XWhisperFeatureExtractor(feature_size=128, device_output=True)
    out = fe([cases.synth_clip(0), cases.synth_clip(1, 4000)], sampling_rate=16000, return_tensors="pt")
    assert out["input_features"].is_cuda and out["input_features"].shape == (2, 128, 3000)


def test_against_installed_reference_class(fe80):
    transformers = pytest.importorskip("transformers")
    ref = transformers.WhisperFeatureExtractor(feature_size=80)
    rng = np.random.default_rng(5)
    batch = [(0.2 * rng.standard_normal(n)).astype(np.float32) for n in (480000, 31234, 512345, 1600)]
    for kw in ({}, {"padding": "longest"}, {"do_normalize": True}, {"max_length": 160000},
               {"max_length": 100000, "pad_to_multiple_of": 160, "return_attention_mask": True}):
        want = ref(batch, sampling_rate=16000, **kw)
        got = fe80(batch, sampling_rate=16000, **kw)
        assert got["input_features"].shape == want["input_features"].shape, kw
        err = np.abs(got["input_features"] - want["input_features"]).max()
        assert err <= TOL, (kw, err)
        if "attention_mask" in want:
            assert np.array_equal(got["attention_mask"], np.asarray(want["attention_mask"]))


def test_other_feature_sizes():
    """feature_size outside {80, 128} takes the run-time-mask kernel build."""
    from oracle import logmel as ol
    from wfe_amd.feature_extractor import MI355XWhisperFeatureExtractor
    wav = cases.load_test_wav()
    for m in (40, 64, 100, 1):
        fe = MI355XWhisperFeatureExtractor(feature_size=m)
        got = fe(wav, sampling_rate=16000)["input_features"]
        want = ol.logmel_f32([wav], n_mel=m)
        assert got.shape == want.shape == (1, m, 3000)
        assert np.abs(got - want).max() <= TOL, m


def test_dither_matches_installed_reference_class():
    """HF:146-147: dither * torch.randn on the padded batch, drawn from torch's global CPU generator -- a seeded
    run sees the same noise with either class (do_normalize comes before the dither, the padding is dithered too)."""
    transformers = pytest.importorskip("transformers")
    from wfe_amd.feature_extractor import MI355XWhisperFeatureExtractor
    ref = transformers.WhisperFeatureExtractor(feature_size=80, dither=1e-3)
    fe = MI355XWhisperFeatureExtractor(feature_size=80, dither=1e-3)
    rng = np.random.default_rng(11)
    batch = [(0.2 * rng.standard_normal(n)).astype(np.float32) for n in (480000, 31234, 1600)]
    for kw in ({}, {"do_normalize": True}, {"padding": "longest"}):
        torch.manual_seed(99)
        want = ref(batch, sampling_rate=16000, **kw)["input_features"]
        after_ref = torch.rand(1).item()
        torch.manual_seed(99)
        got = fe(batch, sampling_rate=16000, **kw)["input_features"]
        assert torch.rand(1).item() == after_ref, kw              # the same number of draws
        assert got.shape == want.shape, kw
        err = np.abs(got - want).max()
        assert err <= TOL, (kw, err)
        silent = fe(batch, sampling_rate=16000, **kw)["input_features"]
        assert np.abs(silent - got).max() > 1e-3                  # the next draw is another noise
