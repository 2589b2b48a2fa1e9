"""concrete_ntt::native_binary128 (src/native_binary128.rs)."""
from ._native import _make

Plan32 = _make(5, 5, 16, 4, True, "native_binary128::Plan32 (src/native_binary128.rs): 5 x 30-bit primes")
