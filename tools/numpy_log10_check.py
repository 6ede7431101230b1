#!/usr/bin/env python3
"""Why the NOISE gains are compared with a tolerance and not bit for bit.
The reference computes them (utils/reader.py:325-331, 350-353) from 10 * np.log10(np.mean(x ** 2)) and
10. ** (gain / 20.) on float32 scalars.  NumPy evaluates float32 log10 / power with whatever its build and
the host CPU dispatch to (SVML on AVX-512 hosts, libm elsewhere); neither is correctly rounded, so the
gains of "the reference" are themselves host dependent in the last few ulp.  This script counts, on the
host it runs on, how often NumPy's float32 log10 / power differ from the correctly rounded result."""
import numpy as np

rng = np.random.default_rng(0)
x = np.exp(rng.uniform(-25, 5, 200000)).astype(np.float32)
ref = np.log10(x.astype(np.float64)).astype(np.float32)
got = np.log10(x)
ulp = np.abs(got.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))
print(f"numpy {np.__version__}: float32 log10 differs from the correctly rounded value in "
      f"{(got != ref).mean():.1%} of {len(x)} samples (max {ulp.max()} ulp)")
y = rng.uniform(-15, 15, 200000).astype(np.float32)
p = np.power(np.float32(10.0), y)
q = np.power(10.0, y.astype(np.float64)).astype(np.float32)
print(f"float32 power(10, y) differs in {(p != q).mean():.2%}")
