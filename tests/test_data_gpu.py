"""GPU: CustomDataset -> DataLoader(num_workers=0) -> collator end to end, against
the CPU oracle replaying the same random stream (oracle-augmented waveform ->
oracle log-mel)."""
import json
import random
import wave

import numpy as np
import pytest
import torch

import cases
import stubs
from oracle import augment as oa
from oracle import logmel as ol

pytestmark = pytest.mark.gpu


def _write_wav(path, x, sr=16000):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(np.round(np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())


def _read_wav(path):
    with wave.open(str(path)) as w:
        return np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").astype(np.float32) / 32768.0


@pytest.mark.parametrize("n_mel", [80, 128])
def test_dataset_collator_end_to_end(tmp_path, n_mel):
    from wfe_amd import data as wd
    rng = np.random.default_rng(11)
    (tmp_path / "noise").mkdir()
    for j in range(3):
        _write_wav(tmp_path / "noise" / f"n{j}.wav", cases.noise_clip(j, 0.7 + j))
    recs = []
    for i, dur in enumerate([1.0, 2.5, 4.2, 0.8, 29.9, 3.3]):
        p = tmp_path / f"a{i}.wav"
        _write_wav(p, 0.1 * rng.standard_normal(int(dur * 16000)))
        recs.append({"audio": {"path": str(p)}, "sentence": f"utt {i}", "duration": dur})
    mf = tmp_path / "list.json"
    mf.write_text("\n".join(json.dumps(r) for r in recs))
    cfg = [
        {"type": "noise", "params": {"min_snr_dB": 10, "max_snr_dB": 50, "noise_dir": str(tmp_path / "noise")}, "prob": 0.7},
        {"type": "speed", "params": {"min_speed_rate": 0.9, "max_speed_rate": 1.1, "num_rates": 3}, "prob": 0.7},
        {"type": "shift", "params": {"min_shift_ms": -5, "max_shift_ms": 5}, "prob": 0.7},
        {"type": "volume", "params": {"min_gain_dBFS": -15, "max_gain_dBFS": 15}, "prob": 0.7},
    ]
    cfgp = tmp_path / "aug.json"
    cfgp.write_text(json.dumps(cfg))
    proc = stubs.StubProcessor(n_mel=n_mel)
    ds = wd.CustomDataset(str(mf), proc, augment_config_path=str(cfgp))
    coll = wd.DataCollatorSpeechSeq2SeqWithPadding(processor=proc)
    loader = torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False, num_workers=0, collate_fn=coll)

    random.seed(2024)
    batches = list(loader)
    assert [tuple(b["input_features"].shape) for b in batches] == [(3, n_mel, 3000)] * 2
    assert all(b["input_features"].device.type == "cpu" and b["labels"].dtype == torch.int64 for b in batches)

    # oracle replay of the same stream, in the same (sequential) order
    random.seed(2024)
    bank = [_read_wav(p) for p in ds.noise_files]
    want = []
    for r in recs:
        x = _read_wav(r["audio"]["path"])
        aug, _, _ = oa.augment(x.copy(), 16000, cfg, random, bank)
        want.append(aug)
    ref = ol.logmel_f32(want, n_mel=n_mel)
    got = torch.cat([b["input_features"] for b in batches]).numpy()
    err = np.abs(got - ref).reshape(len(recs), -1).max(axis=1)
    print("end-to-end max-abs-err per utterance:", err)
    assert (err <= 1e-4).all(), err
    lab = batches[0]["labels"]
    assert lab[0].tolist()[:4] == [stubs.SOT, stubs.EN, stubs.TRANSCRIBE, stubs.NOTIMESTAMPS]


def test_specaug_and_eval_collator(tmp_path):
    from wfe_amd import data as wd
    rng = np.random.default_rng(12)
    recs = []
    for i in range(2):
        p = tmp_path / f"a{i}.wav"
        _write_wav(p, 0.1 * rng.standard_normal(16000 * (i + 1)))
        recs.append({"audio": {"path": str(p)}, "sentence": "x", "duration": float(i + 1), "id": f"utt{i}"})
    recs.append({"audio": {"path": str(tmp_path / "missing.wav")}, "sentence": "x", "duration": 1.0})
    mf = tmp_path / "list.json"
    mf.write_text("\n".join(json.dumps(r) for r in recs))
    cfgp = tmp_path / "aug.json"
    cfgp.write_text(json.dumps([{"type": "specaug", "params": {}, "prob": 1.0}]))
    proc = stubs.StubProcessor()
    ds = wd.CustomDataset(str(mf), proc, augment_config_path=str(cfgp))
    ev = wd.DataCollatorSpeechSeq2SeqWithPaddingforEval(processor=proc, device_output=True)
    b = ev([ds[0], ds[1], ds[2]])
    assert b["id"] == ["utt0", "utt1"] and b["input_features"].is_cuda and b["input_features"].shape == (2, 80, 3000)
    assert not b["input_features"].any()                   # the reference's specaug zeroes the whole tensor (A7)
    plain = wd.CustomDataset(str(mf), proc)
    b2 = ev([plain[0], plain[1]])
    want = ol.logmel_f32([_read_wav(r["audio"]["path"]) for r in recs[:2]], n_mel=80)
    assert np.abs(b2["input_features"].cpu().numpy() - want).max() <= 1e-4
