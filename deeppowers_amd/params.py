"""FheParams - the one POD configuration of the FHE hot path (SURVEY.md section 5 "Config / flags").

The reference has no FHE parameter set (its examples carry none:
/root/reference/examples/basic_generation.cpp:11-17, examples/quantization_example.cpp:71-84), so
the pinned constants below are the BUILD-SPEC of SURVEY.md Appendix A:

  limb i = i-th largest prime < 2^60 with q = 1 (mod 16384); psi = smallest primitive 2N-th root.

tests/test_params.py re-derives every entry (primality, psi^N = -1, minimality) instead of
trusting the table.
"""
from __future__ import annotations

from dataclasses import dataclass

# (q, psi for N=4096, psi for N=8192)
PRIMES_60 = (
    (1152921504606830593, 116777451583545, 25959043411404),
    (1152921504606748673, 271802498405390, 100406242475323),
    (1152921504606683137, 134367042585739, 45474351589225),
    (1152921504606601217, 276147373136904, 92707844590835),
    (1152921504606584833, 317490233586139, 23981819781494),
    (1152921504606109697, 279138086580908, 253932030982881),
)
# BASELINE.json configs[0]: N=1024, one 30-bit limb
PRIME_30 = 1073707009
PSI_30_N1024 = 169871

MAX_MODULUS_BITS = 60  # lazy butterflies keep values < 16q in a u64 word
MIN_LOG2N, MAX_LOG2N = 3, 16


@dataclass(frozen=True)
class FheParams:
    """log2N, L limbs, the primes q_i = 1 (mod 2N) and psi_i (a primitive 2N-th root mod q_i)."""

    log2_n: int
    moduli: tuple
    psi: tuple

    def __post_init__(self):
        if not (MIN_LOG2N <= self.log2_n <= MAX_LOG2N):
            raise ValueError(f"log2_n={self.log2_n} outside [{MIN_LOG2N},{MAX_LOG2N}]")
        if len(self.moduli) == 0 or len(self.moduli) != len(self.psi):
            raise ValueError("moduli and psi must be non-empty and of equal length")
        n = 1 << self.log2_n
        for q, w in zip(self.moduli, self.psi):
            if q < 3 or q >> MAX_MODULUS_BITS:
                raise ValueError(f"modulus {q} not in [3, 2^{MAX_MODULUS_BITS})")
            if (q - 1) % (2 * n):
                raise ValueError(f"modulus {q} is not 1 mod 2N")
            if not (0 < w < q) or pow(w, n, q) != q - 1:
                raise ValueError(f"psi {w} is not a primitive 2N-th root mod {q}")

    @property
    def n(self) -> int:
        return 1 << self.log2_n

    @property
    def n_limbs(self) -> int:
        return len(self.moduli)

    def words_per_rns_poly(self) -> int:
        return self.n_limbs * self.n

    def words_per_ct(self, components: int = 2) -> int:
        return components * self.n_limbs * self.n

    def drop_last_limb(self) -> "FheParams":
        """the next level after a rescale"""
        return FheParams(self.log2_n, tuple(self.moduli[:-1]), tuple(self.psi[:-1]))

    # ---- the BASELINE.json parameter sets -----------------------------------------------------
    @staticmethod
    def config1() -> "FheParams":
        """configs[0]: N=1024, 1 limb, 30-bit q (the CPU-runnable bit-exact check)."""
        return FheParams(10, (PRIME_30,), (PSI_30_N1024,))

    @staticmethod
    def n4096_l4() -> "FheParams":
        """configs[1..3]: N=4096, 4 x 60-bit limbs (the metric configuration)."""
        return FheParams(12, tuple(p[0] for p in PRIMES_60[:4]), tuple(p[1] for p in PRIMES_60[:4]))

    @staticmethod
    def n8192_l6() -> "FheParams":
        """configs[4] sizes: N=8192, 6 x 60-bit limbs."""
        return FheParams(13, tuple(p[0] for p in PRIMES_60), tuple(p[2] for p in PRIMES_60))

    @staticmethod
    def n8192(n_limbs: int) -> "FheParams":
        """N=8192 on the first n_limbs primes of the same descending chain (n8192(6) == n8192_l6()): deeper levels and the workspaces of
        exact multiplies at higher levels (include/deeppowers/fhe.hpp FheParams::n8192 tabulates the first 20)."""
        return ntt_primes(13, n_limbs)


    @staticmethod
    def generic_n4096_l4() -> "FheParams":
        """N=4096 on four GENERIC primes (the largest primes = 1 mod 8192 below 2^59, 2^50, 2^40, 2^33; none of the 2^60 - d shape): the library's
        generic-prime arithmetic (Harvey/Shoup butterflies, 128-bit Barrett products) at the headline shape - bench.py other_configs.shoup_n4096_l4."""
        qs = tuple(ntt_primes(12, 1, bits).moduli[0] for bits in (59, 50, 40, 33))
        return FheParams(12, qs, tuple(min_primitive_2n_root(4096, q) for q in qs))


# ---- building other parameter sets ---------------------------------------------------------------------------------
def is_prime(n: int) -> bool:
    """Deterministic Miller-Rabin for n < 3.3e24 (the first 13 primes as witnesses)."""
    if n < 2:
        return False
    small = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41)
    for p in small:
        if n % p == 0:
            return n == p
    d, r = n - 1, 0
    while d % 2 == 0:
        d //= 2
        r += 1
    for a in small:
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(r - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def min_primitive_2n_root(n: int, q: int) -> int:
    """Smallest psi with psi^n = -1 (mod q), i.e. of order exactly 2n (n a power of two, q = 1 mod 2n prime) - Appendix A's rule."""
    if (q - 1) % (2 * n):
        raise ValueError("q is not 1 mod 2N")
    for g in range(2, 2 + 4096):                # g^((q-1)/2n) has full order for half of all g: the search ends after a few candidates
        w = pow(g, (q - 1) // (2 * n), q)
        if pow(w, n, q) == q - 1:
            x, w2, best = w, w * w % q, w
            for _ in range(n):                  # the 2N-th roots of full order are the odd powers of any one of them
                if x < best:
                    best = x
                x = x * w2 % q
            return best
    raise ValueError("no primitive 2N-th root found (q not prime?)")


def ntt_primes(log2_n: int, count: int, bits: int = 60) -> "FheParams":
    """`count` largest primes below 2^bits with q = 1 (mod 2N) and their smallest primitive 2N-th roots.  For bits = 60 these are of the
    form 2^60 - d with small d, the shape the library's fold-reduction kernels take (dpfhe_ctx_uses_fold)."""
    n = 1 << log2_n
    qs, q = [], (1 << bits) - ((1 << bits) - 1) % (2 * n)
    while len(qs) < count:
        if q < 3:
            raise ValueError("not enough primes")
        if is_prime(q):
            qs.append(q)
        q -= 2 * n
    return FheParams(log2_n, tuple(qs), tuple(min_primitive_2n_root(n, v) for v in qs))
