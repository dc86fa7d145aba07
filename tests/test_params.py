"""Re-derives SURVEY.md Appendix A instead of trusting the pinned table (CPU)."""
import pytest

from deeppowers_amd.params import PRIMES_60, PRIME_30, PSI_30_N1024, FheParams
from oracle import pyoracle as po

sympy = pytest.importorskip("sympy")


def test_primes_are_the_largest_below_2_60_congruent_1_mod_16384():
    found, q = [], (1 << 60) - ((1 << 60) - 1) % 16384  # largest value = 1 mod 16384 below 2^60
    assert q % 16384 == 1
    while len(found) < 6:
        if sympy.isprime(q):
            found.append(q)
        q -= 16384
    assert found == [p[0] for p in PRIMES_60]
    assert all(po.is_prime(p) for p in found)


def test_prime30_is_largest_30bit_prime_1_mod_2048():
    q = (1 << 30) - ((1 << 30) - 1) % 2048
    while not sympy.isprime(q):
        q -= 2048
    assert q == PRIME_30


@pytest.mark.parametrize("row", range(6))
def test_psi_are_minimal_primitive_roots(row):
    q, psi4096, psi8192 = PRIMES_60[row]
    assert po.is_primitive_2n_root(psi4096, 4096, q) and po.is_primitive_2n_root(psi8192, 8192, q)
    assert po.min_primitive_2n_root(4096, q) == psi4096
    assert po.min_primitive_2n_root(8192, q) == psi8192


def test_psi30():
    assert po.min_primitive_2n_root(1024, PRIME_30) == PSI_30_N1024


def test_params_validation():
    FheParams.config1(); FheParams.n4096_l4(); FheParams.n8192_l6()
    with pytest.raises(ValueError):
        FheParams(12, (PRIMES_60[0][0],), (PRIMES_60[0][2],))  # psi of the wrong order
    with pytest.raises(ValueError):
        FheParams(12, (PRIMES_60[0][0] + 2,), (3,))  # not 1 mod 2N
    with pytest.raises(ValueError):
        FheParams(12, (), ())
    with pytest.raises(ValueError):
        FheParams(12, ((1 << 61) + 1,), (3,))  # too wide
    p = FheParams.n4096_l4()
    assert p.n == 4096 and p.n_limbs == 4 and p.words_per_ct(2) * 8 == 256 * 1024 and p.words_per_ct(3) * 8 == 384 * 1024


def test_parameter_search_reproduces_the_pinned_table_and_the_oracle():
    """deeppowers_amd.params.ntt_primes / is_prime / min_primitive_2n_root (product-side helpers for other parameter sets) against
    Appendix A's pinned constants and against the oracle's independent implementations."""
    from deeppowers_amd.params import PRIMES_60, is_prime, min_primitive_2n_root, ntt_primes
    from oracle import pyoracle as po
    p13 = ntt_primes(13, 6)
    assert p13.moduli == tuple(x[0] for x in PRIMES_60) and p13.psi == tuple(x[2] for x in PRIMES_60)
    p12 = ntt_primes(12, 4)
    assert p12.moduli == tuple(x[0] for x in PRIMES_60[:4]) and p12.psi == tuple(x[1] for x in PRIMES_60[:4])
    for log2n, bits in ((8, 60), (10, 30), (16, 60), (11, 45)):
        p = ntt_primes(log2n, 2, bits)
        for q, w in zip(p.moduli, p.psi):
            assert po.is_prime(q) and is_prime(q) and q < (1 << bits) and (q - 1) % (2 << log2n) == 0
            assert w == po.min_primitive_2n_root(1 << log2n, q) == min_primitive_2n_root(1 << log2n, q)
    for n in (1, 4, 9, 561, 1105, 2**61 - 1, (2**31 - 1) * (2**31 - 1), 3215031751):
        assert is_prime(n) == po.is_prime(n), n
