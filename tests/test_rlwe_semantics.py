"""Semantic end-to-end check (test-side only): a toy RLWE encryption built from
Python big ints shows that the tensor product computed on the ring level is the product under decryption:
    phase(ct) = c0 + c1 s (+ c2 s^2)  and  phase(ct_a (x) ct_b) = phase(ct_a) * phase(ct_b)  in R_Q, Q = prod q_i.
CPU leg uses the C oracle; the GPU leg (-m gpu) runs the HIP fused multiply through the C ABI."""
import numpy as np
import pytest

from deeppowers_amd.params import PRIMES_60, FheParams
from oracle import pyoracle as po
from oracle.cbind import Oracle


def small_params(log2n=8, limbs=3):
    n = 1 << log2n
    qs = tuple(PRIMES_60[i][0] for i in range(limbs))
    return FheParams(log2n, qs, tuple(pow(PRIMES_60[i][2], 8192 // n, qs[i]) for i in range(limbs)))


def crt_compose(residues, moduli):
    """residues[l][k] -> list of big ints mod Q (Garner-free: direct CRT with big ints)."""
    Q = 1
    for q in moduli:
        Q *= q
    out = [0] * len(residues[0])
    for l, q in enumerate(moduli):
        Ql = Q // q
        inv = pow(Ql, -1, q)
        for k, r in enumerate(residues[l]):
            out[k] = (out[k] + int(r) * inv % q * Ql) % Q
    return out, Q


def encrypt(rng, p, s, msg, delta):
    """(c0, c1) = (-a s + e + delta m, a) per limb; returns arrays [2][L][N] and the exact integer phase."""
    n, L = p.n, p.n_limbs
    e = rng.integers(-8, 9, n)
    ct = np.zeros((2, L, n), np.uint64)
    for l, q in enumerate(p.moduli):
        a = [int(rng.integers(0, 2**62)) % q for _ in range(n)]
        a_s = po.negacyclic_schoolbook(a, [int(v) % q for v in s], q)
        ct[0, l] = [(-a_s[k] + int(e[k]) + delta * int(msg[k])) % q for k in range(n)]
        ct[1, l] = a
    return ct, [int(e[k]) + delta * int(msg[k]) for k in range(n)]


def phase(p, ct, s):
    """c0 + c1 s + c2 s^2 per limb, CRT-composed to integers mod Q."""
    res = []
    for l, q in enumerate(p.moduli):
        sq = [int(v) % q for v in s]
        acc = [int(v) for v in ct[0, l]]
        spow = sq
        for comp in range(1, ct.shape[0]):
            term = po.negacyclic_schoolbook([int(v) for v in ct[comp, l]], spow, q)
            acc = po.poly_add(acc, term, q)
            spow = po.negacyclic_schoolbook(spow, sq, q)
        res.append(acc)
    return crt_compose(res, p.moduli)


def negacyclic_int(a, b, Q):
    n = len(a)
    out = [0] * n
    for i, ai in enumerate(a):
        for j, bj in enumerate(b):
            k = i + j
            if k < n:
                out[k] += ai * bj
            else:
                out[k - n] -= ai * bj
    return [v % Q for v in out]


def run_semantic(multiply):
    p = small_params()
    rng = np.random.default_rng(2024)
    s = rng.integers(-1, 2, p.n)  # ternary secret
    m1, m2 = rng.integers(0, 1000, p.n), rng.integers(0, 1000, p.n)
    delta = 1 << 40
    ct1, ph1 = encrypt(rng, p, s, m1, delta)
    ct2, ph2 = encrypt(rng, p, s, m2, delta)
    got1, Q = phase(p, ct1, s)
    assert got1 == [v % Q for v in ph1]                      # decryption of a fresh ciphertext is exact
    ct3 = multiply(p, ct1, ct2)                              # [3][L][N]
    got3, _ = phase(p, ct3, s)
    assert got3 == negacyclic_int(ph1, ph2, Q)               # phase(ct1 (x) ct2) == phase(ct1) * phase(ct2)


def test_tensor_product_is_multiplicative_under_decryption_oracle():
    def mul(p, a, b):
        return Oracle.from_params(p).ct_mul(np.ascontiguousarray(a), np.ascontiguousarray(b))[0]
    run_semantic(mul)


@pytest.mark.gpu
def test_tensor_product_is_multiplicative_under_decryption_hip():
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host

    def mul(p, a, b):
        ctx = Context(p, 0)
        ev = Evaluator(ctx)
        c = ev.multiply(Ciphertext(to_device(a[None], ctx.device)), Ciphertext(to_device(b[None], ctx.device)))
        out = to_host(c.data)[0]
        ctx.close()
        return out
    run_semantic(mul)
