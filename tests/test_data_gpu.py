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
    # training collator, upstream default: the labels keep their <|startoftranscript|>
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


def test_other_sampling_rates_end_to_end(tmp_path):
    """Files that are not at 16 kHz (reader.py:157-158), a resample augmentation
    (:263-267) and noise clips at a third rate (:323), through CustomDataset -> collator
    with the GPU resampler, against the oracle chain.  Resampling is PARITY UNPINNED
    w.r.t. the reference's soxr filter (oracle/resample.py); the tolerance here is the
    log-mel bar, 1e-4, against the oracle's restated filter."""
    from oracle import resample as rs
    from wfe_amd import data as wd
    rng = np.random.default_rng(21)
    (tmp_path / "noise").mkdir()
    for j in range(2):
        _write_wav(tmp_path / "noise" / f"n{j}.wav", cases.noise_clip(j, 1.5 + j)[:int(16000 * (1.5 + j))], sr=22050)
    recs, rates = [], [44100, 8000, 16000, 48000, 16000]
    for i, (sr, dur) in enumerate(zip(rates, [2.0, 3.1, 1.7, 29.0, 4.0])):
        p = tmp_path / f"a{i}.wav"
        t = np.arange(int(dur * sr)) / sr
        x = 0.2 * np.sin(2 * np.pi * (300 + 150 * i) * t) * (0.6 + 0.4 * np.sin(2 * np.pi * 2 * t)) \
            + 0.02 * rng.standard_normal(len(t))
        _write_wav(p, x, sr=sr)
        recs.append({"audio": {"path": str(p)}, "sentence": f"utt {i}", "duration": dur})
    mf = tmp_path / "list.json"
    mf.write_text("\n".join(json.dumps(r) for r in recs))
    proc = stubs.StubProcessor(n_mel=80)
    read = lambda r: _read_wav(r["audio"]["path"])                      # noqa: E731

    # 1. no augmentation: decode -> resample to 16 kHz -> log-mel
    ds = wd.CustomDataset(str(mf), proc)
    coll = wd.DataCollatorSpeechSeq2SeqWithPadding(processor=proc)
    got = coll([ds[i] for i in range(len(recs))])["input_features"].numpy()
    want = [rs.resample(read(r), sr, 16000) if sr != 16000 else read(r) for r, sr in zip(recs, rates)]
    err = np.abs(got - ol.logmel_f32(want, n_mel=80)).reshape(len(recs), -1).max(axis=1)
    print("multi-rate max-abs-err:", err)
    assert (err <= 1e-4).all(), err

    # 2. with the shipped entry order and everything firing, incl. the resample and reverb entries
    (tmp_path / "rir").mkdir()
    for j in range(2):
        h = np.random.default_rng(40 + j).standard_normal(2000 + 777 * j) * np.exp(-np.arange(2000 + 777 * j) / 500.0)
        _write_wav(tmp_path / "rir" / f"r{j}.wav", 0.3 * h / np.abs(h).max(), sr=16000)
    cfg = [
        {"type": "resample", "params": {"new_sample_rates": [8000, 32000, 44100]}, "prob": 1.0},
        {"type": "noise", "params": {"min_snr_dB": 10, "max_snr_dB": 50, "noise_dir": str(tmp_path / "noise")}, "prob": 1.0},
        {"type": "reverb", "params": {"rir_dir": str(tmp_path / "rir")}, "prob": 1.0},
        {"type": "speed", "params": {"min_speed_rate": 0.9, "max_speed_rate": 1.1, "num_rates": 3}, "prob": 1.0},
        {"type": "shift", "params": {"min_shift_ms": -5, "max_shift_ms": 5}, "prob": 1.0},
        {"type": "volume", "params": {"min_gain_dBFS": -15, "max_gain_dBFS": 15}, "prob": 1.0},
    ]
    cfgp = tmp_path / "aug.json"
    cfgp.write_text(json.dumps(cfg))
    ds = wd.CustomDataset(str(mf), proc, augment_config_path=str(cfgp))
    coll = wd.DataCollatorSpeechSeq2SeqWithPadding(processor=proc)
    random.seed(7)
    np.random.seed(7)
    got = coll([ds[i] for i in range(len(recs))])["input_features"].numpy()
    random.seed(7)
    np.random.seed(7)
    bank = [_read_wav(p) for p in ds.noise_files]
    rirs = [_read_wav(p) for p in ds.rir_files]
    want = []
    for r, sr in zip(recs, rates):
        aug, sr2, trace = oa.augment(read(r).copy(), sr, cfg, random, bank, np_rnd=np.random, noise_rates=[22050, 22050],
                                     rir_bank=rirs, rir_rates=[16000, 16000])
        assert [t[0] for t in trace][:3] == ["resample", "noise", "reverb"]
        want.append(rs.resample(aug, sr2, 16000) if sr2 != 16000 else aug)
    err = np.abs(got - ol.logmel_f32(want, n_mel=80)).reshape(len(recs), -1).max(axis=1)
    print("multi-rate augmented max-abs-err:", err)
    assert (err <= 1e-4).all(), err


def test_collator_output_dtype(tmp_path):
    """output_dtype / WFE_OUTPUT_DTYPE: input_features arrive in the training dtype and equal
    the float32 batch rounded; the shared front end is float32 again afterwards."""
    from wfe_amd import data as wd
    rng = np.random.default_rng(31)
    recs = []
    for i, dur in enumerate([1.2, 3.4, 0.6]):
        p = tmp_path / f"a{i}.wav"
        _write_wav(p, 0.1 * rng.standard_normal(int(dur * 16000)))
        recs.append({"audio": {"path": str(p)}, "sentence": f"utt {i}", "duration": dur})
    mf = tmp_path / "list.json"
    mf.write_text("\n".join(json.dumps(r) for r in recs))
    proc = stubs.StubProcessor(n_mel=80)
    ds = wd.CustomDataset(str(mf), proc)
    items = [ds[i] for i in range(3)]
    want = wd.DataCollatorSpeechSeq2SeqWithPadding(processor=proc)(items)["input_features"]
    for name in ("float16", "bfloat16"):
        got = wd.DataCollatorSpeechSeq2SeqWithPadding(processor=proc, output_dtype=name)(items)["input_features"]
        assert got.dtype == getattr(torch, name) and torch.equal(got, want.to(got.dtype))
    again = wd.DataCollatorSpeechSeq2SeqWithPadding(processor=proc)(items)["input_features"]
    assert again.dtype == torch.float32 and torch.equal(again, want)
    with pytest.raises(ValueError):
        wd.DataCollatorSpeechSeq2SeqWithPadding(processor=proc, output_dtype="int8")


def test_prefetch_iterator_with_the_gpu_collator(tmp_path, monkeypatch):
    """The one-batch-ahead iterator the launcher installs (wfe_amd.data.install_prefetch): the collator runs on a
    thread of this process, on a stream and a front-end handle of its own, batch i + 1 is launched while batch i
    is consumed.  Same draws in the same order, so the batches must be the bits of the plain in-line iteration --
    with features left on the GPU (WFE_DEVICE_OUTPUT=1, the launcher's default) and read on ANOTHER stream than
    the one they were produced on."""
    from wfe_amd import data as wd
    rng = np.random.default_rng(17)
    recs = []
    for i in range(23):
        p = tmp_path / f"p{i}.wav"
        _write_wav(p, 0.1 * rng.standard_normal(int((0.6 + 0.37 * i) * 16000)))
        recs.append({"audio": {"path": str(p)}, "sentence": f"utt {i}", "duration": 0.6 + 0.37 * i})
    mf = tmp_path / "list.json"
    mf.write_text("\n".join(json.dumps(r) for r in recs))
    cfgp = tmp_path / "aug.json"
    cfgp.write_text(json.dumps([
        {"type": "speed", "params": {"min_speed_rate": 0.9, "max_speed_rate": 1.1, "num_rates": 3}, "prob": 0.8},
        {"type": "shift", "params": {"min_shift_ms": -5, "max_shift_ms": 5}, "prob": 0.8},
        {"type": "volume", "params": {"min_gain_dBFS": -15, "max_gain_dBFS": 15}, "prob": 0.8}]))
    monkeypatch.setenv("WFE_DEVICE_OUTPUT", "1")
    proc = stubs.StubProcessor(n_mel=80)

    def batches(prefetch):
        ds = wd.CustomDataset(str(mf), proc, augment_config_path=str(cfgp))
        coll = wd.DataCollatorSpeechSeq2SeqWithPadding(processor=proc, private_handle=prefetch)
        random.seed(99)                                      # (before the producer thread starts drawing)
        it = iter(torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False, num_workers=0, collate_fn=coll))
        if prefetch:
            it = wd.PrefetchIterator(it, device=0)
        out = []
        side = torch.cuda.Stream()
        for b in it:
            assert b["input_features"].is_cuda
            with torch.cuda.stream(side):                    # a consumer on a stream of its own
                side.wait_stream(torch.cuda.current_stream())
                out.append((b["input_features"].float().clone(), b["labels"].clone()))
        torch.cuda.synchronize()
        return out

    plain = batches(False)
    ahead = batches(True)
    assert len(plain) == len(ahead) == 6
    for (fa, la), (fb, lb) in zip(plain, ahead):
        assert torch.equal(fa, fb) and torch.equal(la.cpu(), lb.cpu())
