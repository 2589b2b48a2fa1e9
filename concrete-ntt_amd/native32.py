"""concrete_ntt::native32 (src/native32.rs)."""
from ._native import _make

Plan32 = _make(0, 3, 4, 4, False, "native32::Plan32 (src/native32.rs): 3 x 30-bit primes")
# The reference offers Plan52 only with AVX-512 IFMA (nightly); here it runs on the u64 HIP kernels.
Plan52 = _make(6, 2, 4, 8, False, "native32::Plan52 (src/native32.rs): 2 x 50-bit primes")
