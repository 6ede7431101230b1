"""16-bit PCM input (include/wfe.h wfe_logmel_pcm16): the device does
soundfile's x / 32768 and librosa.to_mono's float32 mean.  Both are exact
(power-of-two scale; sums of <= 256 16-bit values fit float32), so the log-mel of
PCM input must be BIT-IDENTICAL to the log-mel of the host-decoded float32."""
import random

import numpy as np
import pytest
import torch

import cases
from oracle import augment as oa
from oracle import logmel as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fe():
    from wfe_amd.frontend import HipFrontEnd
    return HipFrontEnd(n_mel=80, device=0)


def _pcm(seed, frames, ch=1):
    rng = np.random.default_rng(seed)
    t = np.arange(frames) / 16000.0
    x = 9000 * rng.standard_normal((frames, ch)) * (0.5 * (1 + np.sin(2 * np.pi * 3 * t)))[:, None]
    return np.clip(np.round(x), -32768, 32767).astype(np.int16)


def _host_decode(p):
    f = p.astype(np.float32) / np.float32(32768.0)          # soundfile.read(dtype='float32')
    return f[:, 0].copy() if p.shape[1] == 1 else np.mean(f.T, axis=0)       # librosa.to_mono


@pytest.mark.parametrize("ch", [1, 2, 5])
def test_pcm16_bit_identical_to_float_decode(fe, ch):
    from wfe_amd.frontend import RaggedPacker, pack_ragged
    clips = [_pcm(i, n, ch) for i, n in enumerate((480000, 123457, 600000, 1, 0, 30011, 479999))]
    flat, off, lens, got_ch = RaggedPacker(fe.device).pack(clips)
    assert got_ch == ch and flat.dtype == torch.int16
    got = fe.logmel(flat, lens, offsets=off, channels=ch)
    f32 = [_host_decode(p) for p in clips]
    fflat, foff, flens = pack_ragged(f32, fe.device)
    want = fe.logmel(fflat, flens, offsets=foff)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    ref = ol.logmel_f32(f32[:2], n_mel=80)
    assert np.abs(got[:2].cpu().numpy() - ref).max() <= 1e-4


def test_pcm16_dense_rows_and_unaligned_offsets(fe):
    clips = [_pcm(20 + i, 200000) for i in range(3)]
    dense = torch.from_numpy(np.stack([c[:, 0] for c in clips])).to(fe.device)
    lens = torch.tensor([200000, 150001, 7], dtype=torch.int32, device=fe.device)
    got = fe.logmel(dense, lens, n_samples=160000)
    want = fe.logmel(dense.to(torch.float32) / 32768.0, lens, n_samples=160000)
    # utterances at odd element offsets: the 16-byte interior path must not be taken
    flat = torch.zeros(3 * 200000 + 16, dtype=torch.int16, device=fe.device)
    offs = [3, 200000 + 5, 400000 + 9]
    for o, c in zip(offs, clips):
        flat[o:o + 200000] = torch.from_numpy(c[:, 0]).to(fe.device)
    got2 = fe.logmel(flat, lens, n_samples=160000, offsets=torch.tensor(offs, dtype=torch.int64, device=fe.device))
    torch.cuda.synchronize()
    assert torch.equal(got, want) and torch.equal(got2, want)


def test_pcm16_with_augmentation(fe):
    """The fused chain on PCM input equals the chain on the float32 decode."""
    from wfe_amd.augment import AugmentPlanner
    from wfe_amd.frontend import NoiseBank, RaggedPacker, pack_ragged
    cfg = [
        {"type": "noise", "params": {"min_snr_dB": 10, "max_snr_dB": 50}, "prob": 1.0},
        {"type": "speed", "params": {"min_speed_rate": 0.9, "max_speed_rate": 1.1, "num_rates": 3}, "prob": 1.0},
        {"type": "shift", "params": {"min_shift_ms": -5, "max_shift_ms": 5}, "prob": 1.0},
        {"type": "volume", "params": {"min_gain_dBFS": -15, "max_gain_dBFS": 15}, "prob": 1.0},
    ]
    bank = [cases.noise_clip(j, 3.0) for j in range(3)]
    planner = AugmentPlanner(cfg, [len(b) for b in bank])
    clips = [_pcm(40 + i, n, 2) for i, n in enumerate((480000, 100000, 333333))]
    plans = [planner.draw(len(c), 16000, random.Random(900 + i))[0] for i, c in enumerate(clips)]
    nb = NoiseBank(bank, fe.device)
    flat, off, lens, ch = RaggedPacker(fe.device).pack(clips)
    got = fe.augment_logmel(flat, lens, plans, nb, offsets=off, channels=ch)
    f32 = [_host_decode(p) for p in clips]
    fflat, foff, flens = pack_ragged(f32, fe.device)
    want = fe.augment_logmel(fflat, flens, plans, nb, offsets=foff)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    aug = [oa.augment(w.copy(), 16000, cfg, random.Random(900 + i), bank)[0] for i, w in enumerate(f32)]
    assert np.abs(got.cpu().numpy() - ol.logmel_f32(aug, n_mel=80)).max() <= 1e-4


def test_pcm16_argument_errors(fe):
    x = torch.zeros((1, 480000), dtype=torch.int16, device=fe.device)
    lens = torch.tensor([480000], dtype=torch.int32, device=fe.device)
    with pytest.raises(ValueError):
        fe.logmel(x, lens, channels=0)
    with pytest.raises(ValueError):
        fe.logmel(x, lens, channels=257)
    with pytest.raises(ValueError):
        fe.logmel(x.to(torch.int32), lens)
