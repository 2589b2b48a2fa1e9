"""concrete_ntt::prime64 (src/prime64.rs)."""
from ._prime import PrimePlan


class Plan(PrimePlan):
    """Negacyclic NTT plan for 64bit primes (src/prime64.rs:220-236)."""
    BITS = 64


class Solinas:
    """prime64::Solinas (src/prime64/generic_solinas.rs:35-40)."""
    P = (1 << 64) - (1 << 32) + 1
