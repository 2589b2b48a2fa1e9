"""concrete_ntt::prime32 (src/prime32.rs)."""
from ._prime import PrimePlan


class Plan(PrimePlan):
    """Negacyclic NTT plan for 32bit primes (src/prime32.rs:600-616)."""
    BITS = 32
