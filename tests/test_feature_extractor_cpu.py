"""Host logic of the drop-in extractor (no GPU): length planning against the
installed transformers class, persistence, and loud failure without a device."""
import numpy as np
import pytest
import torch
from transformers import WhisperFeatureExtractor

from wfe_amd.feature_extractor import MI355XWhisperFeatureExtractor, install, plan_lengths


def _hf_lengths(lens, **kw):
    fe = WhisperFeatureExtractor(feature_size=80)
    from transformers.feature_extraction_utils import BatchFeature
    raw = [np.zeros((n, 1), dtype=np.float32) for n in lens]
    out = fe.pad(BatchFeature({"input_features": raw}), return_attention_mask=True,
                 max_length=kw.get("max_length") or fe.n_samples, padding=kw.get("padding", "max_length"),
                 truncation=kw.get("truncation", True), pad_to_multiple_of=kw.get("pad_to_multiple_of"))
    x = out["input_features"]
    return x.shape[1], [int(m.sum()) for m in out["attention_mask"]]


@pytest.mark.parametrize("kw", [
    {}, {"padding": "longest"}, {"max_length": 1000}, {"max_length": 1000, "pad_to_multiple_of": 160},
    {"padding": "longest", "pad_to_multiple_of": 128}, {"padding": True, "max_length": 700},
])
def test_plan_lengths_matches_transformers(kw):
    rng = np.random.default_rng(1)
    for _ in range(5):
        lens = [int(x) for x in rng.integers(1, 3000, size=4)]
        want_t, want_k = _hf_lengths(lens, **kw)
        got_t, got_k = plan_lengths(lens, 480000, kw.get("truncation", True), kw.get("pad_to_multiple_of"),
                                    kw.get("padding", "max_length"), kw.get("max_length"))
        assert (got_t, got_k) == (want_t, want_k), (kw, lens)


def test_ragged_result_is_rejected():
    with pytest.raises(ValueError):
        plan_lengths([10, 20], 480000, True, None, "do_not_pad", None)
    with pytest.raises(ValueError):
        plan_lengths([10, 2000], 480000, False, None, "max_length", 1000)


def test_is_a_whisper_feature_extractor_and_round_trips(tmp_path):
    fe = MI355XWhisperFeatureExtractor(feature_size=128)
    assert isinstance(fe, WhisperFeatureExtractor)
    ref = WhisperFeatureExtractor(feature_size=128)
    for name in ("feature_size", "sampling_rate", "n_fft", "hop_length", "chunk_length", "n_samples",
                 "nb_max_frames", "padding_value", "padding_side", "model_input_names"):
        assert getattr(fe, name) == getattr(ref, name), name
    assert np.array_equal(fe.mel_filters, ref.mel_filters)
    fe.save_pretrained(tmp_path)                       # merge_lora.py:46, Trainer checkpoints (finetune.py:149)
    stock = WhisperFeatureExtractor.from_pretrained(tmp_path)          # loadable by stock transformers
    assert type(stock) is WhisperFeatureExtractor and stock.feature_size == 128
    again = MI355XWhisperFeatureExtractor.from_pretrained(tmp_path)
    assert isinstance(again, MI355XWhisperFeatureExtractor) and again.feature_size == 128
    d = fe.to_dict()
    assert "_device" not in d and d["feature_extractor_type"] == "WhisperFeatureExtractor"


def test_install_swaps_the_extractor():
    class P:
        feature_extractor = WhisperFeatureExtractor(feature_size=80)
    p = install(P())
    assert isinstance(p.feature_extractor, MI355XWhisperFeatureExtractor)


def test_errors_before_any_device_work():
    fe = MI355XWhisperFeatureExtractor(feature_size=80)
    with pytest.raises(ValueError):
        fe(np.zeros(100, dtype=np.float32), sampling_rate=8000)        # HF:265-271
    with pytest.raises(ValueError):
        fe(np.zeros((2, 2, 100), dtype=np.float32), sampling_rate=16000)   # HF:279-280
    with pytest.raises(ValueError):
        MI355XWhisperFeatureExtractor(feature_size=80, n_fft=512)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_gpu():
    from wfe_amd._lib import WfeError
    fe = MI355XWhisperFeatureExtractor(feature_size=80)
    with pytest.raises(WfeError):
        fe(np.zeros(16000, dtype=np.float32), sampling_rate=16000)
