"""concrete_ntt::native128 (src/native128.rs)."""
from ._native import _make

Plan32 = _make(2, 10, 16, 4, False, "native128::Plan32 (src/native128.rs): 10 x 30-bit primes")
