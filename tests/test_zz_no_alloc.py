"""wfe_logmel_static (include/wfe.h): the allocation-free variant of wfe_logmel, called through
the C ABI and compared bit for bit with wfe_logmel on two calls in a row -- the second with a
smaller input, so that the static scratch still holds the first call's statistics behind it."""
import ctypes as C

import numpy as np
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu


def _call(fn, fe, wav, off, lens, stride, batch, n_samples, out):
    from wfe_amd import _lib
    rc = fn(fe._h, C.c_void_p(wav.data_ptr()), C.c_void_p(off.data_ptr()) if off is not None else None,
            C.c_void_p(lens.data_ptr()), stride, batch, n_samples, C.c_void_p(out.data_ptr()),
            C.c_void_p(torch.cuda.current_stream(fe.device).cuda_stream))
    return rc, (_lib.lib().wfe_last_error(fe._h) or b"").decode()


@pytest.mark.parametrize("n_mel", [80, 128])
def test_static_entry_matches_wfe_logmel_twice(n_mel):
    from wfe_amd import _lib
    from wfe_amd.frontend import HipFrontEnd, pack_ragged
    lib = _lib.lib()
    fe = HipFrontEnd(n_mel=n_mel, device=0)
    inputs = cases.logmel_cases()
    first = [inputs[n] for n in ("test_wav", "synth0", "short_5p3s", "long_31p3s", "zeros", "hdr_burst", "tiny_100")]
    second = [inputs[n] for n in ("tone440", "short_5p3s")]           # smaller batch AND a shorter padded length
    for waves, n_samples in ((first, 480000), (second, 96000)):
        flat, off, lens = pack_ragged(waves, fe.device)
        want = torch.full((len(waves), n_mel, n_samples // 160), 7.0, device=fe.device)
        got = torch.full_like(want, -7.0)
        rc, msg = _call(lib.wfe_logmel, fe, flat, off, lens, 0, len(waves), n_samples, want)
        assert rc == 0, msg
        rc, msg = _call(lib.wfe_logmel_static, fe, flat, off, lens, 0, len(waves), n_samples, got)
        assert rc == 0, msg
        torch.cuda.synchronize()
        assert torch.equal(got, want)                                   # same kernels, same bits
        assert bool(torch.isfinite(got).all())
    # 16-bit output format is honoured as by wfe_logmel
    fe.set_output_dtype(torch.bfloat16)
    flat, off, lens = pack_ragged(second, fe.device)
    want = torch.empty((2, n_mel, 600), dtype=torch.bfloat16, device=fe.device)
    got = torch.zeros_like(want)
    assert _call(lib.wfe_logmel, fe, flat, off, lens, 0, 2, 96000, want)[0] == 0
    assert _call(lib.wfe_logmel_static, fe, flat, off, lens, 0, 2, 96000, got)[0] == 0
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    fe.set_output_dtype(torch.float32)


def test_static_entry_rejects_sizes_beyond_its_scratch():
    from wfe_amd import _lib
    from wfe_amd.frontend import HipFrontEnd
    lib = _lib.lib()
    fe = HipFrontEnd(n_mel=80, device=0)
    batch = 257                                                         # 257 x 47 tiles > 12032
    wav = torch.zeros((batch, 480000), device=fe.device)
    lens = torch.full((batch,), 480000, dtype=torch.int32, device=fe.device)
    out = torch.empty((batch, 80, 3000), device=fe.device)
    rc, msg = _call(lib.wfe_logmel_static, fe, wav, None, lens, 480000, batch, 480000, out)
    assert rc == _lib.ERR_ARG and "exceeds the static scratch" in msg
    rc, msg = _call(lib.wfe_logmel_static, fe, wav, None, lens, 480000, 256, 480000, out[:256])
    assert rc == 0, msg                                                 # exactly the limit is fine
    torch.cuda.synchronize()
    assert float(out[0, 0, 0]) == -1.5
