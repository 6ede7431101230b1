"""CPU checks of the host-side table builders exported by the C ABI (no GPU):
the library loads, exports every symbol of include/wfe.h, and the packed
DFT / mel tables reproduce the oracle when the kernel's folded-DFT algebra is
replayed in numpy with the kernel's own index maps."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import cases
from oracle import logmel as ol
from wfe_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    header = open(os.path.join(ROOT, "include", "wfe.h")).read()
    declared = set(re.findall(r"\b(wfe_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.wfe_version()


def _tables(n_mel):
    lib = _lib.lib()
    dft = np.zeros(lib.wfe_dft_table_floats(), dtype=np.float32)
    mel = np.zeros(lib.wfe_mel_table_floats(), dtype=np.float32)
    mask = np.zeros(14, dtype=np.uint32)
    assert lib.wfe_build_dft_table(dft.ctypes.data) == 0
    assert lib.wfe_build_mel_table(n_mel, mel.ctypes.data, mask.ctypes.data) == 0
    return dft.reshape(25, 4, 2, 64, 4), mel.reshape(2, 7, 8, 64, 4), mask


@pytest.mark.parametrize("n_mel", [80, 128, 1, 37])
def test_mel_filters_match_oracle(n_mel, golden_logmel):
    lib = _lib.lib()
    out = np.zeros((201, n_mel), dtype=np.float64)
    assert lib.wfe_build_mel_filters(n_mel, out.ctypes.data) == 0
    want = ol.mel_filter_bank(n_mel)
    assert np.abs(out - want).max() < 1e-15
    assert np.array_equal(out.astype(np.float32), want.astype(np.float32))
    if n_mel in (80, 128):
        ref = np.load(os.path.join(cases.GOLDEN, "mel_filters.npz"))[f"mel_{n_mel}"]
        assert np.abs(out.T - ref).max() < 1e-8
    assert lib.wfe_build_mel_filters(0, out.ctypes.data) == _lib.ERR_ARG
    assert lib.wfe_build_mel_filters(129, out.ctypes.data) == _lib.ERR_ARG


def test_k_step_map_covers_each_n_once():
    lib = _lib.lib()
    for c, want in ((0, set(range(2, 201, 2))), (1, set(range(1, 200, 2)))):
        got = [lib.wfe_dft_n_of(c, s, kk) for s in range(25) for kk in range(4)]
        assert len(got) == 100 and set(got) == want
        # bank-conflict rule: k-lanes 0/1 and 2/3 differ by 16 in all but the last step
        for s in range(24):
            assert lib.wfe_dft_n_of(c, s, 1) - lib.wfe_dft_n_of(c, s, 0) == 16
            assert lib.wfe_dft_n_of(c, s, 3) - lib.wfe_dft_n_of(c, s, 2) == 16


def _emulate(frames, dft, mel, mask, n_mel):
    """Replays frontend_kernel's arithmetic for a (F, 400) float frame matrix in
    float64, reading the packed float32 tables with the kernel's lane maps."""
    lib = _lib.lib()
    F = frames.shape[0]
    acc = np.zeros((4, 112, F))                       # [gemm][folded bin][frame]
    lane = np.arange(64)
    for s in range(25):
        for g in range(4):
            c = g >> 1
            n = np.array([lib.wfe_dft_n_of(c, s, kk) for kk in range(4)])
            xf, xb = frames[:, n], frames[:, 400 - n]            # (F, 4)
            bop = (xf - xb) if (g & 1) else (xf + xb)
            for bt in range(7):
                a = dft[s, g, bt >> 2, :, bt & 3].astype(np.float64)   # (64,) lane l: bin l&15, k l>>4
                amat = np.zeros((16, 4))
                amat[lane & 15, lane >> 4] = a
                acc[g, 16 * bt:16 * bt + 16, :] += amat @ bop.T
    ce, se, co, so = acc
    P = np.stack([(ce + co) ** 2 + (se + so) ** 2, (ce - co) ** 2 + (se - so) ** 2])   # [family][k][frame]
    out = np.zeros((128, F))
    for f in range(2):
        for bt in range(7):
            for mt in range(8):
                blk = mel[f, bt, mt].astype(np.float64)           # (64 lanes, 4 regs)
                if not (mask[f * 7 + bt] >> mt) & 1:
                    assert not blk.any()
                    continue
                for r in range(4):
                    amat = np.zeros((16, 4))
                    amat[lane & 15, lane >> 4] = blk[:, r]        # mel row l&15, k-lane l>>4
                    kbins = 16 * bt + 4 * np.arange(4) + r        # folded bins of the 4 k-lanes
                    out[16 * mt:16 * mt + 16, :] += amat @ P[f][kbins, :]
    return P, out[:n_mel]


@pytest.mark.parametrize("n_mel", [80, 128])
def test_packed_tables_reproduce_oracle(n_mel):
    dft, mel, mask = _tables(n_mel)
    wav = cases.load_test_wav()
    padded, _ = ol.pad_or_trim(wav)
    sel = np.r_[0:3, 400:420, 2997:3000]                # reflect edges + voiced middle + zero tail
    frames = ol.frame_matrix(padded)[sel].astype(np.float64)
    P, melp = _emulate(frames, dft, mel, mask, n_mel)
    want_p = ol.power_spectrum_f64(padded)[:, sel]
    got_p = np.zeros_like(want_p)
    got_p[:101] = P[0][:101]
    got_p[101:] = P[1][:100][::-1]                      # family 1 holds bin 200 - k
    scale = want_p.max()
    assert np.abs(got_p - want_p).max() <= 2e-6 * scale
    want_m = ol.mel_filter_bank(n_mel).T @ want_p
    assert np.abs(melp - want_m).max() <= 2e-6 * want_m.max()


def test_resample_table_matches_oracle_filter():
    """The float32 polyphase table the kernel reads ([taps][L]) against the float64
    filter of oracle/resample.py, for down-, up- and awkward ratios; DC gain 1."""
    import ctypes as C
    from oracle import resample as rs
    from wfe_amd import _lib
    lib = _lib.lib()
    for orig, target in ((44100, 16000), (8000, 16000), (48000, 16000), (22050, 16000), (16000, 44100), (32000, 16000)):
        L, M, taps, k0 = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        assert lib.wfe_resample_geometry(orig, target, C.byref(L), C.byref(M), C.byref(taps), C.byref(k0)) == 0
        from math import gcd
        g = gcd(orig, target)
        assert (L.value, M.value) == (target // g, orig // g)
        scale = min(1.0, target / orig)
        assert k0.value == int(np.floor(rs.ZEROS / scale)) and taps.value == 2 * k0.value + 2
        tab = np.zeros((taps.value, L.value), dtype=np.float32)
        assert lib.wfe_build_resample_table(orig, target, tab.ctypes.data) == 0
        j = np.arange(taps.value)[:, None]
        p = np.arange(L.value)[None, :]
        want = rs.filter_value((j - k0.value) - p / L.value, scale)
        assert np.abs(tab - want).max() <= 1e-7
        assert np.abs(tab.astype(np.float64).sum(axis=0) - 1.0).max() <= 2e-5      # every phase: unit DC gain
    bad = C.c_int32()
    assert lib.wfe_resample_geometry(0, 16000, C.byref(bad), C.byref(bad), C.byref(bad), C.byref(bad)) != 0
