"""GPU: geometries off the benchmark's happy path -- unaligned ragged offsets (scalar
staging path), padded lengths that are not multiples of 4 or 160 (scalar epilogue),
zero-length utterances, a batch larger than one launch slice, dense rows with a
stride larger than the clip."""
import numpy as np
import pytest
import torch

import cases
from oracle import logmel as ol

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def fe():
    from wfe_amd.frontend import HipFrontEnd
    return HipFrontEnd(n_mel=80, device=0)


def test_unaligned_offsets_and_odd_padded_lengths(fe):
    rng = np.random.default_rng(21)
    waves = [(0.2 * rng.standard_normal(n)).astype(np.float32) for n in (30011, 17, 48000, 0, 25999)]
    # pack with arbitrary (odd) offsets: float4 staging must not be taken
    gaps = [1, 3, 2, 5, 7]
    total = sum(len(w) + g for w, g in zip(waves, gaps)) + 8
    flat = np.zeros(total, dtype=np.float32)
    offs, o = [], 0
    for w, g in zip(waves, gaps):
        o += g
        offs.append(o)
        flat[o:o + len(w)] = w
        o += len(w)
    dflat = torch.from_numpy(flat).cuda()
    doff = torch.tensor(offs, dtype=torch.int64, device="cuda")
    dlen = torch.tensor([len(w) for w in waves], dtype=torch.int32, device="cuda")
    for n_samples in (30011, 48000, 40001, 201, 160 * 64 + 2):
        got = fe.logmel(dflat, dlen, n_samples=n_samples, offsets=doff).cpu().numpy()
        want = ol.logmel_f32(waves, n_mel=80, n_samples=n_samples)
        assert got.shape == want.shape == (5, 80, n_samples // 160)
        assert np.abs(got - want).max() <= TOL, n_samples
        assert (got[3] == np.float32(-1.5)).all()              # the empty utterance


def test_dense_rows_with_stride(fe):
    rng = np.random.default_rng(22)
    x = (0.2 * rng.standard_normal((3, 20000))).astype(np.float32)
    lens = [20000, 12345, 1]
    d = torch.from_numpy(x).cuda()
    dl = torch.tensor(lens, dtype=torch.int32, device="cuda")
    got = fe.logmel(d, dl, n_samples=16000).cpu().numpy()     # truncation below the row stride
    want = ol.logmel_f32([x[i, :lens[i]] for i in range(3)], n_mel=80, n_samples=16000)
    assert np.abs(got - want).max() <= TOL


def test_batch_larger_than_one_launch_slice(fe):
    """4096 utterances per launch slice (wfe_capi.cpp): 5000 short clips cross it."""
    B, n = 5000, 4000
    g = torch.Generator(device="cuda").manual_seed(5)
    x = 0.1 * torch.randn((B, n), generator=g, device="cuda")
    lens = torch.full((B,), n, dtype=torch.int32, device="cuda")
    lens[4500] = 1234
    got = fe.logmel(x, lens, n_samples=n)
    ref = fe.logmel(x[4090:4110].contiguous(), lens[4090:4110].contiguous(), n_samples=n)
    assert torch.equal(got[4090:4110], ref)
    idx = [0, 4095, 4096, 4500, 4999]
    want = ol.logmel_f32([x[i, :int(lens[i])].cpu().numpy() for i in idx], n_mel=80, n_samples=n)
    assert np.abs(got[idx].cpu().numpy() - want).max() <= TOL


def test_stream_ordering(fe):
    """Launches are asynchronous on the caller's current stream."""
    s = torch.cuda.Stream()
    x = torch.from_numpy(np.stack([cases.synth_clip(0), cases.synth_clip(1)])).cuda()
    lens = torch.full((2,), 480000, dtype=torch.int32, device="cuda")
    base = fe.logmel(x, lens)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        y = x * 2.0                                            # produced on s, consumed on s
        out = fe.logmel(y, lens)
    s.synchronize()
    want = ol.logmel_f32([2.0 * cases.synth_clip(0), 2.0 * cases.synth_clip(1)], n_mel=80)
    assert np.abs(out.cpu().numpy() - want).max() <= TOL
    assert not torch.equal(out, base)


def test_augmented_batch_larger_than_one_launch_slice(fe):
    """The resolve pass works per launch slice of 4096 utterances (arrival tickets, partial sums and chain heads
    are indexed inside the slice and reused by the next one): 4200 short utterances with a NOISE op cross it, and
    a second, smaller call follows on the same handle.  Rows equal the same utterances run in a small batch."""
    from wfe_amd import _lib
    from wfe_amd.frontend import NoiseBank
    B, n = 4200, 6000
    g = torch.Generator(device="cuda").manual_seed(6)
    x = 0.1 * torch.randn((B, n), generator=g, device="cuda")
    lens = torch.full((B,), n, dtype=torch.int32, device="cuda")
    lens[4100] = 2345
    bank = [cases.noise_clip(0, 0.2), cases.noise_clip(1, 1.0)]
    nb = NoiseBank(bank, fe.device)
    gain = float(np.float32(10. ** (3 / 20.)))

    def plan(i):
        j = i % 2
        start = (i * 7) % (len(bank[j]) - n) if len(bank[j]) > n else 0
        return [(_lib.AUG_NOISE, j, start, 10 + i % 30, 0.0), (_lib.AUG_VOLUME, 0, 0, 3, gain)]

    plans = [plan(i) for i in range(B)]
    got = fe.augment_logmel(x, lens, plans, nb, n_samples=n)
    sel = list(range(4090, 4110)) + [0, 4199]
    idx = torch.tensor(sel, device="cuda")
    ref = fe.augment_logmel(x[idx].contiguous(), lens[idx].contiguous(), [plans[i] for i in sel], nb, n_samples=n)
    again = fe.augment_logmel(x, lens, plans, nb, n_samples=n)
    torch.cuda.synchronize()
    assert torch.equal(got[idx], ref)
    assert torch.equal(again, got)
