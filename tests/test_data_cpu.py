"""Dataset / collator host logic (no GPU): manifest filtering, slicing, label
arithmetic, null records, draw order, label padding.  Collator behaviour is
PARITY UNPINNED (no reference source); it is checked against the call-site
contract restated in oracle/collate.py."""
import json
import os
import random
import wave

import numpy as np
import pytest
import torch

import cases
import stubs
from oracle import collate as oc
from wfe_amd import data as wd


def _write_wav(path, x, sr=16000):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())


@pytest.fixture()
def corpus(tmp_path):
    rng = np.random.default_rng(0)
    recs = []
    for i, dur in enumerate([1.0, 0.3, 2.5, 31.0, 1.7]):
        p = tmp_path / f"a{i}.wav"
        _write_wav(p, 0.1 * rng.standard_normal(int(dur * 16000)))
        recs.append({"audio": {"path": str(p)}, "sentence": f"utt {i}", "duration": dur,
                     "sentences": [{"start": 0.0, "end": 0.51, "text": "ab"}, {"start": 0.53, "end": 0.9, "text": "c"}]})
    recs[4]["audio"].update(start_time=0.5, end_time=1.25)
    recs[4]["language"] = "zh"
    recs[2]["sentence"] = ""
    mf = tmp_path / "list.json"
    mf.write_text("\n".join(json.dumps(r) for r in recs))
    return str(mf), recs


def test_manifest_filter_and_items(corpus, monkeypatch):
    mf, recs = corpus
    proc = stubs.StubProcessor()
    monkeypatch.setenv("WFE_PCM16", "0")                  # float32 host decode, as the reference (reader.py:144)
    ds = wd.CustomDataset(mf, proc)
    assert len(ds) == 3                                   # 0.3 s (< min) and 31 s (> max) dropped (reader.py:127-130)
    it = ds[0]
    assert it["waveform"].dtype == np.float32 and it["waveform"].shape == (16000,) and it["aug_ops"] == []
    assert it["labels"] == proc.tokenizer("utt 0").input_ids and it["id"] == recs[0]["audio"]["path"]
    it = ds[1]                                            # empty transcript -> [sot, nospeech, eot] (reader.py:196-199)
    assert it["labels"] == [stubs.SOT, stubs.NOSPEECH, stubs.EOT]
    it = ds[2]                                            # sliced read int(0.5*sr)..int(1.25*sr), per-item language
    assert it["waveform"].shape == (12000,) and it["labels"][1] == stubs.ZH
    batch = ds.__getitems__([0, 1, 2])                    # threaded decode path gives the same items
    assert all(np.array_equal(a["waveform"], ds[i]["waveform"]) for i, a in enumerate(batch))


def test_pcm16_items_are_the_file_samples(corpus, tmp_path, monkeypatch):
    """Default decode of 16-bit PCM: the int16 samples as they lie in the file
    (frames, channels); x / 32768 and the mono mean happen on the GPU.  Here: the
    values equal the float32 decode times 32768, slices agree, a stereo file stays
    interleaved, a non-16-bit file falls back to float32."""
    from wfe_amd.frontend import RaggedPacker
    mf, recs = corpus
    proc = stubs.StubProcessor()
    ds16 = wd.CustomDataset(mf, proc)
    monkeypatch.setenv("WFE_PCM16", "0")
    ds32 = wd.CustomDataset(mf, proc)
    for i in range(3):
        a, b = ds16[i], ds32[i]
        assert isinstance(a["waveform"], wd.PcmRef)
        assert a["waveform"].dtype == np.int16 and a["waveform"].shape == (len(b["waveform"]), 1)
        assert np.array_equal(a["waveform"][:, 0].astype(np.float32) / np.float32(32768), b["waveform"])
        assert a["labels"] == b["labels"] and a["id"] == b["id"]
    st = tmp_path / "st.wav"
    rng = np.random.default_rng(5)
    pcm = rng.integers(-32768, 32767, size=(8000, 2), dtype=np.int16)
    with wave.open(str(st), "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes(pcm.tobytes())
    got, sr = wd.read_pcm16(str(st))
    assert sr == 16000 and got.shape == (8000, 2) and np.array_equal(np.asarray(got), pcm)
    got, _ = wd.read_pcm16(str(st), 0.1, -0.1)             # negative end counts from the end (reader.py:229-232)
    assert np.array_equal(np.asarray(got), pcm[1600:6400])
    dst = np.full(4800 * 2 + 3, 7, dtype=np.int16)          # read_into fills exactly frames * channels elements
    got.read_into(dst[:9600])
    assert np.array_equal(dst[:9600], pcm[1600:6400].reshape(-1)) and (dst[9600:] == 7).all()
    ref, _ = wd.read_audio(str(st))
    assert np.array_equal(RaggedPacker._as_float_mono(pcm), np.mean(ref.T, axis=0))
    b8 = tmp_path / "b8.wav"
    with wave.open(str(b8), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(1)
        w.setframerate(16000)
        w.writeframes(bytes(range(200)))
    assert wd.read_pcm16(str(b8)) is None and wd.read_pcm16(str(tmp_path / "missing.wav")) is None


def test_timestamp_labels(corpus):
    mf, _ = corpus
    proc = stubs.StubProcessor()
    ds = wd.CustomDataset(mf, proc, timestamps=True)
    got = ds[0]["labels"]
    tb = stubs.NOTIMESTAMPS + 1
    # 0.51 s: 51 cs odd -> 0.50 -> 25 ; 0.53 -> 0.54 -> 27 ; 0.90 -> 45   (reader.py:170-173)
    want = [stubs.SOT, stubs.EN, stubs.TRANSCRIBE, tb + 0, 1000 + ord("a"), 1000 + ord("b"), tb + 25,
            tb + 27, 1000 + ord("c"), tb + 45, stubs.EOT]
    assert got == want
    segs = [{"start": 0.0, "end": 0.51, "text": "ab"}, {"start": 0.53, "end": 0.9, "text": "c"}]
    assert got == oc.timestamp_labels(segs, proc.tokenizer.prefix_tokens[:3], tb, stubs.EOT,
                                      lambda t: proc.tokenizer(t).input_ids)


def test_errors_become_null_records(corpus, capsys):
    mf, recs = corpus
    os.remove(recs[0]["audio"]["path"])
    ds = wd.CustomDataset(mf, stubs.StubProcessor())
    assert ds[0] == {"input_features": None, "labels": None, "id": "error_0"}     # reader.py:209-216
    assert "error reading sample 0" in capsys.readouterr().err


def test_augment_draw_order_matches_reference_stream(corpus, tmp_path, golden_augment):
    """With the shipped probabilities every gate is drawn per sample, then the
    specaug gate (reader.py:201-202): the number of random draws per item is pinned
    by the goldens of the reference's own augment()."""
    mf, _ = corpus
    cfgp = tmp_path / "aug.json"
    shipped = [
        {"type": "resample", "params": {"new_sample_rates": [8000, 32000, 44100]}, "prob": 0.0},
        {"type": "noise", "params": {"min_snr_dB": 10, "max_snr_dB": 50, "noise_dir": str(tmp_path / "none")}, "prob": 0.5},
        {"type": "reverb", "params": {"rir_dir": str(tmp_path / "none")}, "prob": 0.5},
        {"type": "speed", "params": {"min_speed_rate": 0.9, "max_speed_rate": 1.1, "num_rates": 3}, "prob": 0.0},
        {"type": "shift", "params": {"min_shift_ms": -5, "max_shift_ms": 5}, "prob": 0.0},
        {"type": "volume", "params": {"min_gain_dBFS": -15, "max_gain_dBFS": 15}, "prob": 0.0},
        {"type": "specaug", "params": {"num_t_mask": 2, "num_f_mask": 2, "max_t": 50, "max_f": 10}, "prob": 0.0},
    ]
    cfgp.write_text(json.dumps(shipped))
    ds = wd.CustomDataset(mf, stubs.StubProcessor(), augment_config_path=str(cfgp))
    random.seed(100)
    it = ds[0]
    assert it["aug_ops"] == [] and it["specaug"] is False
    random.random()                                        # the specaug gate was drawn by __getitem__: one more
    r = random.Random(100)
    for _ in range(6 + 1 + 1):                             # 6 augment gates + specaug gate + the draw above
        last = r.random()
    assert random.random() == r.random()


def test_specaug_quirk_zeroes_everything(corpus, tmp_path):
    mf, _ = corpus
    cfgp = tmp_path / "aug.json"
    cfgp.write_text(json.dumps([{"type": "specaug", "params": {}, "prob": 1.0}]))
    ds = wd.CustomDataset(mf, stubs.StubProcessor(), augment_config_path=str(cfgp))
    random.seed(0)
    it = ds[0]
    assert it["specaug"]["all"] is True and len(it["specaug"]["rows"]) == 2
    # same draws as reader.py:377-387 on a (1, 80, 3000) tensor
    r = random.Random(0)
    r.random()
    for _ in range(2):
        assert r.randint(0, 0) == 0
        r.randint(1, 50)
    rows = []
    for _ in range(2):
        s = r.randint(0, 79)
        rows.append((s, min(80, s + r.randint(1, 10))))
    assert it["specaug"]["rows"] == rows


def test_collator_contract_on_precomputed_features():
    proc = stubs.StubProcessor()
    rng = np.random.default_rng(3)
    items = [
        {"input_features": rng.standard_normal((1, 80, 3000)).astype(np.float32), "labels": [stubs.SOT, 5, 6, stubs.EOT], "id": "a"},
        {"input_features": None, "labels": None, "id": "error_1"},
        {"input_features": torch.zeros((1, 80, 3000)), "labels": [stubs.SOT, 7, stubs.EOT], "id": "c"},
    ]
    for cls, strip, ids, kw in ((wd.DataCollatorSpeechSeq2SeqWithPadding, (stubs.EOT,), False, {}),
                                (wd.DataCollatorSpeechSeq2SeqWithPadding, (stubs.EOT, stubs.SOT), False, {"strip_sot": True}),
                                (wd.DataCollatorSpeechSeq2SeqWithPaddingforEval, (), True, {})):
        got = cls(processor=proc, **kw)(items)
        want = oc.collate(items, 80, 3000, strip_tokens=strip, with_ids=ids)
        assert got["input_features"].dtype == torch.float32 and got["labels"].dtype == torch.int64
        assert np.array_equal(got["input_features"].numpy(), want["input_features"])
        assert np.array_equal(got["labels"].numpy(), want["labels"])
        assert got.get("id") == want.get("id")
    assert got["labels"][1].tolist() == [stubs.SOT, 7, stubs.EOT, -100]          # evaluation.py:106,111
    # training, upstream default: only bos_token_id (= <|endoftext|> for Whisper) is ever stripped, so
    # the labels START WITH <|startoftranscript|>; padding is -100
    train = wd.DataCollatorSpeechSeq2SeqWithPadding(processor=proc)(items)["labels"].tolist()
    assert train == [[stubs.SOT, 5, 6, stubs.EOT], [stubs.SOT, 7, stubs.EOT, -100]]
    # the opt-in deviation (strip_sot=True or WFE_STRIP_SOT=1): the shared <|startoftranscript|> is dropped
    train = wd.DataCollatorSpeechSeq2SeqWithPadding(processor=proc, strip_sot=True)(items)["labels"].tolist()
    assert train == [[5, 6, stubs.EOT], [7, stubs.EOT, -100]]
    # a shared leading BOS is stripped by the training collator only
    bos_items = [{"input_features": np.zeros((1, 80, 3000), np.float32), "labels": [stubs.EOT, 1, 2], "id": "x"}]
    assert wd.DataCollatorSpeechSeq2SeqWithPadding(processor=proc)(bos_items)["labels"].tolist() == [[1, 2]]
    assert wd.DataCollatorSpeechSeq2SeqWithPaddingforEval(processor=proc)(bos_items)["labels"].tolist() == [[stubs.EOT, 1, 2]]
    # all records null -> empty batch the evaluation loop skips (evaluation.py:94-99)
    empty = wd.DataCollatorSpeechSeq2SeqWithPaddingforEval(processor=proc)([items[1]])
    assert empty["id"] == [] and empty["input_features"].size(0) == 0


def test_binary_manifest(corpus, tmp_path):
    """utils/binary.py layout written by hand: key, data as length-prefixed records;
    header lines key \\t offset \\t length."""
    import struct
    mf, recs = corpus
    with open(tmp_path / "m.data", "wb") as d, open(tmp_path / "m.header", "wb") as h:
        off = 0
        for i, r in enumerate(recs):
            key, payload = str(i).encode(), json.dumps(r).encode()
            d.write(struct.pack("I", len(key)) + key + struct.pack("I", len(payload)) + payload)
            off += 4 + len(key) + 4
            h.write(key + b"\t" + str(off).encode() + b"\t" + str(len(payload)).encode() + b"\n")
            off += len(payload)
    ds = wd.CustomDataset(str(tmp_path / "m.header"), stubs.StubProcessor())
    ref = wd.CustomDataset(mf, stubs.StubProcessor())
    assert len(ds) == len(ref) == 3
    assert np.array_equal(ds[2]["waveform"], ref[2]["waveform"]) and ds[2]["labels"] == ref[2]["labels"]


def test_text_helpers():
    assert wd.remove_punctuation("你好，世界！") == "你好世界"
    assert wd.remove_punctuation(["a, b.", "(c)"]) == ["a b", "c"]
    assert wd.to_lower(["AbC"]) == ["abc"] and wd.to_lower("X") == "x"
