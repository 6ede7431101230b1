"""BASELINE configs 4 and 5 at test size, on the GPU.

config 4  the sharded driver (wfe_amd.sharded.run_sharded, world size 1) with the real HIP
          transform at 128 mel over 40 clips in uneven batches, against the oracle on the same
          clips: per-element <= 1e-4 and the checksum of checksums.
config 5  finetune.py's step (finetune.py:98-104,144-155) at toy size: DataLoader(num_workers=0,
          collate_fn=DataCollatorSpeechSeq2SeqWithPadding) -> a random-init
          WhisperForConditionalGeneration forward + backward on the GPU.  input_features against
          the oracle <= 1e-4; labels as the Trainer expects them (label_names=["labels"],
          finetune.py:131-132): -100 padding, and the decoder sees ONE start token."""
import json
import wave

import numpy as np
import pytest
import torch

import stubs
from oracle import logmel as ol

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _clip(i):
    n = 16000 + (7919 * i) % 32000                                  # 1 .. 3 s, a function of i alone
    return (0.1 * np.random.default_rng(8100 + i).standard_normal(n)).astype(np.float32)


def test_config4_sharded_driver_with_hip_transform_128_mel():
    from wfe_amd.frontend import HipFrontEnd, pack_ragged
    from wfe_amd.sharded import run_sharded
    fe = HipFrontEnd(n_mel=128, device=0)
    n_items, batch, n_samples = 40, 7, 48000                         # 5 batches of 7 and one of 5
    seen = []

    def make_batch(begin, count):
        return pack_ragged([_clip(i) for i in range(begin, begin + count)], fe.device)

    def transform(flat, off, lens):
        out = fe.logmel(flat, lens, n_samples=n_samples, offsets=off)
        seen.append(out)
        return out

    res = run_sharded(n_items, batch, make_batch, transform, sync=torch.cuda.synchronize, rank=0, world=1)
    assert res["items"] == n_items and res["shard"] == (0, n_items) and [o.shape[0] for o in seen] == [7] * 5 + [5]
    want = ol.logmel_f32([_clip(i) for i in range(n_items)], n_mel=128, n_samples=n_samples)
    got = torch.cat(seen).cpu().numpy()
    assert got.shape == want.shape == (n_items, 128, n_samples // 160)
    assert np.abs(got - want).max() <= TOL
    assert res["checksum"] == pytest.approx(float(want.astype(np.float64).sum()), abs=TOL * 0.05 * want.size)
    # the sampled checksum bench.py uses is a function of the clips only, not of the batching
    a = run_sharded(n_items, 7, make_batch, transform, sync=torch.cuda.synchronize, rank=0, world=1, checksum_stride=11)
    b = run_sharded(n_items, 16, make_batch, transform, sync=torch.cuda.synchronize, rank=0, world=1, checksum_stride=11)
    assert a["checksum"] == pytest.approx(b["checksum"], rel=1e-12)


def _write_wav(path, x, sr=16000):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(np.round(np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())


def _read_wav(path):
    with wave.open(str(path)) as w:
        return np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").astype(np.float32) / 32768.0


@pytest.mark.parametrize("strip_sot", [False, True])
def test_config5_collator_feeds_a_whisper_training_step(tmp_path, strip_sot):
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    from wfe_amd import data as wd
    recs = []
    for i, text in enumerate(["ab", "hello", "x", "whisper on mi355x", "", "cd"]):
        p = tmp_path / f"a{i}.wav"
        _write_wav(p, _clip(i))
        recs.append({"audio": {"path": str(p)}, "sentence": text, "duration": len(_clip(i)) / 16000})
    mf = tmp_path / "list.json"
    mf.write_text("\n".join(json.dumps(r) for r in recs))
    proc = stubs.StubProcessor(80)
    ds = wd.CustomDataset(str(mf), proc, language="en")
    # default = the upstream collator (labels keep <|startoftranscript|>); strip_sot=True = the opt-in HF recipe
    coll = wd.DataCollatorSpeechSeq2SeqWithPadding(processor=proc, strip_sot=strip_sot)
    loader = torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False, num_workers=0, collate_fn=coll)

    torch.manual_seed(0)
    cfg = WhisperConfig(vocab_size=51865, num_mel_bins=80, d_model=64, encoder_layers=2, decoder_layers=2,
                        encoder_attention_heads=2, decoder_attention_heads=2, encoder_ffn_dim=128, decoder_ffn_dim=128,
                        max_source_positions=1500, max_target_positions=448, decoder_start_token_id=stubs.SOT,
                        pad_token_id=stubs.EOT, bos_token_id=stubs.EOT, eos_token_id=stubs.EOT)
    model = WhisperForConditionalGeneration(cfg).to("cuda").train()
    want = ol.logmel_f32([_read_wav(r["audio"]["path"]) for r in recs], n_mel=80)
    losses = []
    for k, batch in enumerate(loader):
        feats, labels = batch["input_features"], batch["labels"]
        assert feats.shape == (3, 80, 3000) and feats.dtype == torch.float32 and labels.dtype == torch.int64
        assert np.abs(feats.numpy() - want[3 * k:3 * k + 3]).max() <= TOL
        rows = [ds[3 * k + j]["labels"] for j in range(3)]
        drop = 1 if strip_sot else 0
        width = max(len(r) for r in rows) - drop
        for j, r in enumerate(rows):                                 # -100 behind the row
            assert r[0] == stubs.SOT
            assert labels[j].tolist() == r[drop:] + [-100] * (width - (len(r) - drop))
        out = model(input_features=feats.to("cuda"), labels=labels.to("cuda"))
        # what the decoder was fed: the labels shifted right behind ONE start token
        from transformers.models.whisper.modeling_whisper import shift_tokens_right
        dec = shift_tokens_right(labels, cfg.pad_token_id, cfg.decoder_start_token_id)
        assert (dec[:, 0] == stubs.SOT).all() and bool((dec[:, 1] == stubs.SOT).all()) == (not strip_sot)
        assert torch.isfinite(out.loss)
        out.loss.backward()
        losses.append(float(out.loss))
    g = model.model.encoder.conv1.weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
    assert all(np.isfinite(losses)) and len(losses) == 2
