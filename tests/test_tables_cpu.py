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
    return dft.reshape(25, 8, 64, 4), mel.reshape(14, 8, 64, 4), mask      # [step][quad][lane][4], [power tile][mel tile][lane][reg]


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
    """wfe_layout.h dft_n_of: class 1 = the odd n over 25 k-steps; class 0 = the even n over 26 even-steps
    (13 of n = 0 mod 4, then 13 of n = 2 mod 4), whose last step per sub-class pads with dummies n > 200."""
    lib = _lib.lib()
    odd = [lib.wfe_dft_n_of(1, s, kk) for s in range(25) for kk in range(4)]
    assert len(odd) == 100 and set(odd) == set(range(1, 200, 2))
    ee = [lib.wfe_dft_n_of(0, s, kk) for s in range(13) for kk in range(4)]
    eo = [lib.wfe_dft_n_of(0, s, kk) for s in range(13, 26) for kk in range(4)]
    assert sorted(n for n in ee if n <= 200) == list(range(4, 201, 4))
    assert sorted(n for n in eo if n <= 200) == list(range(2, 200, 4))
    assert sorted(n for n in ee + eo if n > 200) == [210, 212, 214, 216]      # dummy k-lanes: zero table entries
    # bank-conflict rule: k-lanes 0/1 and 2/3 differ by 16 (all odd steps but the last, every even-step)
    for c, steps in ((1, range(24)), (0, range(26))):
        for s in steps:
            assert lib.wfe_dft_n_of(c, s, 1) - lib.wfe_dft_n_of(c, s, 0) == 16
            assert lib.wfe_dft_n_of(c, s, 3) - lib.wfe_dft_n_of(c, s, 2) == 16


def _power_bin_of(T, row):
    """wfe_layout.h power_bin_of: U lo (bins p), U hi (200 - p), V a (100 - p), V b (100 + p)."""
    if T < 4:
        p = 16 * T + row
        return p if p < 53 else -1
    if T < 8:
        p = 16 * (T - 4) + row
        return 200 - p if p < 53 else -1
    if T < 11:
        return 100 - (16 * (T - 8) + row)
    p = 16 * (T - 11) + row
    return 100 + p if p >= 1 else -1


def _emulate(frames, dft, mel, mask, n_mel):
    """Replays frontend_kernel's arithmetic for a (F, 400) float frame matrix in
    float64, reading the packed float32 tables with the kernel's lane maps: the odd class's two GEMMs over
    the 7 tiles U0..3, V0..2, the even class's four (cos | sin) x (ee | eo) over U0..3, the level-C and
    level-B butterflies, and the mel projection of the 14 power tiles."""
    lib = _lib.lib()
    F = frames.shape[0]
    lane = np.arange(64)

    def amat(vec):                                    # lane l: row l & 15, k-lane l >> 4
        a = np.zeros((16, 4))
        a[lane & 15, lane >> 4] = vec
        return a

    def bops(c, step):
        n = np.array([lib.wfe_dft_n_of(c, step, kk) for kk in range(4)])
        xf, xb = frames[:, n], frames[:, 400 - n]     # (F, 4); dummies n > 200 read real samples, weights are 0
        return xf + xb, xf - xb

    co, so = np.zeros((7, 16, F)), np.zeros((7, 16, F))
    ce, se = np.zeros((2, 4, 16, F)), np.zeros((2, 4, 16, F))        # [ee | eo][tile]
    for s in range(25):
        e1, o1 = bops(1, s)
        for ti in range(7):
            co[ti] += amat(dft[s, ti >> 2, :, ti & 3].astype(np.float64)) @ e1.T
            so[ti] += amat(dft[s, 2 + (ti >> 2), :, ti & 3].astype(np.float64)) @ o1.T
        assert not dft[s, 1, :, 3].any() and not dft[s, 3, :, 3].any()         # the padding tile of quads 1 and 3
        for t, qc, qs in [(s, 4, 5)] + ([(25, 6, 7)] if s == 24 else []):
            e0, o0 = bops(0, t)
            cls = 0 if t < 13 else 1
            for ti in range(4):
                ce[cls, ti] += amat(dft[s, qc, :, ti].astype(np.float64)) @ e0.T
                se[cls, ti] += amat(dft[s, qs, :, ti].astype(np.float64)) @ o0.T
        if s != 24:
            assert not dft[s, 6:].any()
    ce_p, ce_m = ce[0] + ce[1], ce[0] - ce[1]                        # Ce[p], Ce[100 - p]
    se_p, se_m = se[0] + se[1], se[1] - se[0]                        # Se[p], Se[100 - p]
    P = np.zeros((14, 16, F))
    for ti in range(4):
        P[ti] = (ce_p[ti] + co[ti]) ** 2 + (se_p[ti] + so[ti]) ** 2
        P[4 + ti] = (ce_p[ti] - co[ti]) ** 2 + (se_p[ti] - so[ti]) ** 2
    for vt in range(3):
        P[8 + vt] = (ce_m[vt] + co[4 + vt]) ** 2 + (se_m[vt] + so[4 + vt]) ** 2
        P[11 + vt] = (ce_m[vt] - co[4 + vt]) ** 2 + (se_m[vt] - so[4 + vt]) ** 2
    spec = np.full((201, F), np.nan)
    for T in range(14):
        for row in range(16):
            b = _power_bin_of(T, row)
            if b >= 0:
                assert np.isnan(spec[b]).all()                           # every bin comes from exactly one position
                spec[b] = P[T, row]
    assert not np.isnan(spec).any()
    out = np.zeros((128, F))
    for T in range(14):
        for mt in range(8):
            blk = mel[T, mt].astype(np.float64)                       # (64 lanes, 4 regs)
            if not (mask[T] >> mt) & 1:
                assert not blk.any()
                continue
            for r in range(4):
                out[16 * mt:16 * mt + 16, :] += amat(blk[:, r]) @ P[T][4 * np.arange(4) + r, :]
    return spec, out[:n_mel]


@pytest.mark.parametrize("n_mel", [80, 128])
def test_packed_tables_reproduce_oracle(n_mel):
    dft, mel, mask = _tables(n_mel)
    wav = cases.load_test_wav()
    padded, _ = ol.pad_or_trim(wav)
    sel = np.r_[0:3, 400:420, 2997:3000]                # reflect edges + voiced middle + zero tail
    frames = ol.frame_matrix(padded)[sel].astype(np.float64)
    got_p, melp = _emulate(frames, dft, mel, mask, n_mel)
    want_p = ol.power_spectrum_f64(padded)[:, sel]
    scale = want_p.max()
    assert np.abs(got_p - want_p).max() <= 2e-6 * scale
    want_m = ol.mel_filter_bank(n_mel).T @ want_p
    assert np.abs(melp - want_m).max() <= 2e-6 * want_m.max()


def test_compile_time_mel_masks_match_the_builder():
    """wfe_layout.h kMelMask80 / kMelMask128 (the kernel's straight-line mel phase) against the masks the
    table builder computes; wfe_create checks the same and falls back to run-time masks otherwise."""
    layout = open(os.path.join(ROOT, "whisper-finetune_amd", "csrc", "wfe_layout.h")).read()
    for n_mel in (80, 128):
        m = re.search(r"kMelMask%d\[14\] = \{([^}]*)\}" % n_mel, layout)
        want = [int(x, 16) for x in m.group(1).replace(" ", "").split(",")]
        assert _tables(n_mel)[2].tolist() == want


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
