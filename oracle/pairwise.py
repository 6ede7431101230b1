"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

What ``np.mean(sample ** 2)`` computes for a contiguous float32 array -- the reduction inside
the reference's ``rms_db`` (utils/reader.py:350-353), which decides the gains of ``add_noise``
(:325-331).  Third-party arithmetic: NumPy (reference pin ``numpy>=1.23.1``, requirements.txt;
installed here 2.2.6), ``numpy/_core/src/umath/loops_utils.h.src`` (``@TYPE@_pairwise_sum``) driven
by the buffered reduction iterator.  Restated from its published algorithm and pinned against
the installed NumPy by tests/test_pairwise_cpu.py (bit-exact on every length tried):

* the iterator hands the inner loop chunks of 8192 elements (the default buffer size); their
  partial sums are accumulated sequentially, in float32:  ((c0 + c1) + c2) + ...
* one chunk is summed pairwise: n < 8 sequentially; n <= 128 with 8 interleaved accumulators
  r[j] += a[i + j], combined as ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)), the n % 8 tail added
  sequentially; larger n split at n2 = n/2 - (n/2) % 8 and the two halves added.
* mean = float32(sum / float32(n)).

The GPU resolve kernels sum in float64 over a fixed 16-way partition instead, so their gains can
differ from NumPy's by an ulp (tests allow 2e-6 of the waveform); a kernel following THIS order
would make the noise path bit-identical -- DESIGN.md, next steps.
"""
import numpy as np

BUFSIZE = 8192
BLOCK = 128


def _leaf(a: np.ndarray) -> np.float32:
    n = len(a)
    if n < 8:
        r = np.float32(0.0)
        for v in a:
            r = np.float32(r + v)
        return r
    r = a[:8].astype(np.float32)
    m = n - (n % 8)
    for i in range(8, m, 8):
        r = (r + a[i:i + 8]).astype(np.float32)
    res = np.float32(np.float32(np.float32(r[0] + r[1]) + np.float32(r[2] + r[3]))
                     + np.float32(np.float32(r[4] + r[5]) + np.float32(r[6] + r[7])))
    for v in a[m:]:
        res = np.float32(res + v)
    return res


def pairwise_sum(a: np.ndarray) -> np.float32:
    n = len(a)
    if n <= BLOCK:
        return _leaf(a)
    n2 = n // 2
    n2 -= n2 % 8
    return np.float32(pairwise_sum(a[:n2]) + pairwise_sum(a[n2:]))


def sum_f32(a: np.ndarray) -> np.float32:
    """np.sum of a contiguous 1-D float32 array."""
    assert a.dtype == np.float32 and a.ndim == 1
    res = None
    for i in range(0, len(a), BUFSIZE):
        p = pairwise_sum(a[i:i + BUFSIZE])
        res = p if res is None else np.float32(res + p)
    return np.float32(0.0) if res is None else res


def mean_f32(a: np.ndarray) -> np.float32:
    """np.mean of a contiguous 1-D float32 array (float32 accumulation, one division)."""
    return np.float32(sum_f32(a) / np.float32(len(a)))
