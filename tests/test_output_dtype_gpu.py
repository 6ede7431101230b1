"""GPU: float16 / bfloat16 log-mel output (include/wfe.h wfe_set_output_format).
The 16-bit result must be, bit for bit, the float32 result rounded to nearest even --
torch's .to(dtype) -- including the tiles kernel 2 clamps or fills (the max-8 clamp
commutes with the rounding).  The float32 result itself is checked against the
reference goldens elsewhere (test_logmel_gpu.py)."""
import ctypes as C
import random

import numpy as np
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu
DTYPES = [torch.float16, torch.bfloat16]


def _fe(n_mel):
    from wfe_amd.frontend import HipFrontEnd
    return HipFrontEnd(n_mel=n_mel, device=0)


def _clips():
    # a full clip, short ones (silent tiles behind them), loud + quiet halves (clamped tiles), silence
    rng = np.random.default_rng(77)
    loud_quiet = np.concatenate([0.5 * rng.standard_normal(80000), 1e-5 * rng.standard_normal(80000)]).astype(np.float32)
    return [cases.bench_clip(0), cases.bench_clip(1, 37000), loud_quiet, np.zeros(5000, np.float32),
            cases.bench_clip(2, 199), cases.bench_clip(3, 480000 + 999)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n_mel", [80, 128])
@pytest.mark.parametrize("n_samples", [480000, 16160])          # 3000 frames | 101 frames (no vector stores)
def test_sixteen_bit_output_is_rounded_float32(dtype, n_mel, n_samples):
    from wfe_amd.frontend import pack_ragged
    fe = _fe(n_mel)
    flat, off, lens = pack_ragged(_clips(), fe.device)
    want = fe.logmel(flat, lens, n_samples=n_samples, offsets=off)
    fe.set_output_dtype(dtype)
    got = fe.logmel(flat, lens, n_samples=n_samples, offsets=off)
    fe.set_output_dtype(torch.float32)
    again = fe.logmel(flat, lens, n_samples=n_samples, offsets=off)
    torch.cuda.synchronize()
    assert got.dtype == dtype and got.shape == want.shape
    assert torch.equal(got, want.to(dtype))
    assert torch.equal(again, want)                              # the setting does not stick past its reset


@pytest.mark.parametrize("dtype", DTYPES)
def test_sixteen_bit_output_augmented_and_pcm(dtype):
    from wfe_amd.augment import AugmentPlanner
    from wfe_amd.frontend import NoiseBank, RaggedPacker, pack_ragged
    fe = _fe(80)
    cfg = [
        {"type": "noise", "params": {"min_snr_dB": 10, "max_snr_dB": 50}, "prob": 1.0},
        {"type": "speed", "params": {"min_speed_rate": 0.9, "max_speed_rate": 1.1, "num_rates": 3}, "prob": 1.0},
        {"type": "shift", "params": {"min_shift_ms": -5, "max_shift_ms": 5}, "prob": 1.0},
        {"type": "volume", "params": {"min_gain_dBFS": -15, "max_gain_dBFS": 15}, "prob": 1.0},
    ]
    bank = [cases.noise_clip(j, 3.0) for j in range(3)]
    planner = AugmentPlanner(cfg, [len(b) for b in bank])
    waves = [cases.bench_clip(20 + i, n) for i, n in enumerate((480000, 90000, 250001))]
    plans = [planner.draw(len(w), 16000, random.Random(60 + i))[0] for i, w in enumerate(waves)]
    nb = NoiseBank(bank, fe.device)
    flat, off, lens = pack_ragged(waves, fe.device)
    want = fe.augment_logmel(flat, lens, plans, nb, offsets=off)
    pcm = [np.round(np.clip(w, -1, 1) * 32767).astype(np.int16)[:, None] for w in waves]
    pflat, poff, plens, ch = RaggedPacker(fe.device).pack(pcm)
    want_pcm = fe.logmel(pflat, plens, offsets=poff, channels=ch)
    fe.set_output_dtype(dtype)
    got = fe.augment_logmel(flat, lens, plans, nb, offsets=off)
    got_pcm = fe.logmel(pflat, plens, offsets=poff, channels=ch)
    fe.set_output_dtype(torch.float32)
    torch.cuda.synchronize()
    assert torch.equal(got, want.to(dtype)) and torch.equal(got_pcm, want_pcm.to(dtype))


def test_output_format_errors():
    from wfe_amd import _lib
    fe = _fe(80)
    with pytest.raises(ValueError):
        fe.set_output_dtype(torch.float64)
    assert _lib.lib().wfe_set_output_format(fe._h, 7) == -1      # WFE_ERR_ARG
    assert b"output format 7" in _lib.lib().wfe_last_error(fe._h)
    x = torch.zeros((1, 16000), device=fe.device)
    lens = torch.tensor([16000], dtype=torch.int32, device=fe.device)
    fe.set_output_dtype(torch.float16)
    with pytest.raises(ValueError):                              # an out tensor of the wrong dtype
        fe.logmel(x, lens, out=torch.empty((1, 80, 3000), device=fe.device))
    fe.set_output_dtype(torch.float32)
