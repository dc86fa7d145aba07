"""Generates tests/golden/*.json from the Python big-int oracle (oracle/pyoracle.py) - and, round 6, one fixture from sympy alone
(sympy_restatement: a third, builder-independent route to the same words).

The reference holds no golden vectors for this path ("Golden vectors /
known-answer tests: none of any kind"), so these are known answers of the mathematical definition,
produced by pure-Python big-int schoolbook / O(N^2) direct evaluation - code that shares nothing
with the C oracle or the HIP kernels.  Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as po  # noqa: E402
from deeppowers_amd.params import PRIMES_60, PRIME_30, PSI_30_N1024  # noqa: E402


def sha(words):
    return hashlib.sha256(po.words_to_bytes(words)).hexdigest()


def small_ntt_vectors():
    """Full forward-NTT vectors at N=16, 64 by direct O(N^2) evaluation (several primes)."""
    out = []
    for log2n in (3, 4, 6):
        n = 1 << log2n
        for q, psi8192 in ((PRIME_30, None), (PRIMES_60[0][0], PRIMES_60[0][2]), (PRIMES_60[3][0], PRIMES_60[3][2])):
            if psi8192 is None:
                psi = pow(PSI_30_N1024, 1024 // n, q)
            else:
                psi = pow(psi8192, 8192 // n, q)
            assert po.is_primitive_2n_root(psi, n, q)
            g = po.SplitMix64(1000 * log2n + (q & 0xFF))
            a = g.words_mod(n, q)
            ahat = po.ntt_forward_definition(a, q, psi)
            assert po.ntt_forward(a, q, psi) == ahat and po.ntt_inverse(ahat, q, psi) == a
            out.append({"log2n": log2n, "q": q, "psi": psi, "a": a, "ntt": ahat})
    return out


def config1_ct_mul():
    """SURVEY.md Appendix B config-1 vector: N=1024, q=1073707009, splitmix64(seed=1), a0,a1,b0,b1."""
    q, n = PRIME_30, 1024
    g = po.SplitMix64(1)
    a0, a1, b0, b1 = (g.words_mod(n, q) for _ in range(4))
    c = po.ct_mul_schoolbook([[a0], [a1]], [[b0], [b1]], [q])
    c0, c1, c2 = c[0][0], c[1][0], c[2][0]
    return {
        "log2n": 10, "q": q, "psi": PSI_30_N1024, "seed": 1,
        "a0_head": a0[:4], "c0_head": c0[:4], "c1_head": c1[:4], "c2_head": c2[:4],
        "sha256": {"c0": sha(c0), "c1": sha(c1), "c2": sha(c2), "c0c1c2": sha(c0 + c1 + c2)},
        "c0": c0, "c1": c1, "c2": c2,
    }


def rns_ct_mul_small():
    """A 2-limb, N=64, batch=2 ct x ct product with 60-bit primes (layout [batch][comp][limb][N])."""
    log2n, n = 6, 64
    moduli = [PRIMES_60[0][0], PRIMES_60[1][0]]
    psis = [pow(PRIMES_60[0][2], 8192 // n, moduli[0]), pow(PRIMES_60[1][2], 8192 // n, moduli[1])]
    g = po.SplitMix64(42)
    batch = 2
    A, B, Cc = [], [], []
    for _ in range(batch):
        a = [[g.words_mod(n, q) for q in moduli] for _ in range(2)]
        b = [[g.words_mod(n, q) for q in moduli] for _ in range(2)]
        c = po.ct_mul_schoolbook(a, b, moduli)
        assert po.ct_mul_ntt(a, b, moduli, psis) == c
        A += po.flatten_ct(a); B += po.flatten_ct(b); Cc += po.flatten_ct(c)
    return {"log2n": log2n, "moduli": moduli, "psi": psis, "batch": batch, "a": A, "b": B, "c": Cc}


def rns_ct_mul_n256():
    """2 limbs (one fold-eligible 60-bit prime, the 30-bit prime), N=256, batch=3: smallest size the
    HIP C-ABI dispatches (log2_n >= 8), so the GPU tests can replay a pure big-int vector."""
    log2n, n = 8, 256
    moduli = [PRIMES_60[2][0], PRIME_30]
    psis = [pow(PRIMES_60[2][2], 8192 // n, moduli[0]), pow(PSI_30_N1024, 1024 // n, moduli[1])]
    g = po.SplitMix64(256)
    batch = 3
    A, B, Cc = [], [], []
    for _ in range(batch):
        a = [[g.words_mod(n, q) for q in moduli] for _ in range(2)]
        b = [[g.words_mod(n, q) for q in moduli] for _ in range(2)]
        c = po.ct_mul_schoolbook(a, b, moduli)
        A += po.flatten_ct(a); B += po.flatten_ct(b); Cc += po.flatten_ct(c)
    ntt_a0 = [po.ntt_forward(A[l * n:(l + 1) * n], moduli[l], psis[l]) for l in range(2)]
    return {"log2n": log2n, "moduli": moduli, "psi": psis, "batch": batch, "a": A, "b": B,
            "c_sha256": sha(Cc), "c_head": Cc[:8], "c_tail": Cc[-8:], "ntt_a0_limb0_head": ntt_a0[0][:8],
            "ntt_a0_sha256": sha(ntt_a0[0] + ntt_a0[1])}


def identities():
    """Hand-checkable products + NTT(delta_0), NTT(X)."""
    n, q = 8, 17
    out = {
        "n8_q17_a": list(range(1, 9)), "n8_q17_b": list(range(8, 0, -1)),
        "n8_q17_ab": po.negacyclic_schoolbook(list(range(1, 9)), list(range(8, 0, -1)), q),
        "n8_q17_1pX_times_X7": po.negacyclic_schoolbook([1, 1, 0, 0, 0, 0, 0, 0], [0] * 7 + [1], q),
    }
    q, psi, n = PRIMES_60[0][0], PRIMES_60[0][1], 4096
    out["ntt_X_n4096_q0_head"] = po.ntt_forward([0, 1] + [0] * (n - 2), q, psi)[:8]
    out["ntt_X_n4096_q0_sha256"] = sha(po.ntt_forward([0, 1] + [0] * (n - 2), q, psi))
    return out


def n4096_ntt_digest():
    """One full-size residue polynomial per limb of the metric configuration: digest of NTT(a)."""
    n = 4096
    res = []
    for l in range(4):
        q, psi = PRIMES_60[l][0], PRIMES_60[l][1]
        a = po.SplitMix64(2000 + l).words_mod(n, q)
        ah = po.ntt_forward(a, q, psi)
        assert po.ntt_inverse(ah, q, psi) == a
        res.append({"limb": l, "q": q, "psi": psi, "seed": 2000 + l, "a_head": a[:4], "ntt_head": ah[:4], "ntt_sha256": sha(ah)})
    return res


def sympy_restatement():
    """A THIRD restatement, written by neither this build's oracle authors nor its kernel authors: sympy's polynomial ring over GF(q) and sympy's
    number-theoretic transform (sympy 1.14, already in the image; tests/test_params.py uses it for primality).  Nothing of oracle/ is called to
    PRODUCE these values (pyoracle only draws the inputs and is compared afterwards):
      * N = 256, two limbs (a pinned 60-bit prime and the 30-bit prime): the ct x ct tensor product as sympy Poly products over GF(q) reduced by X^N + 1;
      * N = 64, the same two primes: the forward negacyclic NTT (i) by evaluating the sympy polynomial at psi^(2 brv(k) + 1) and (ii) through
        sympy.discrete.transforms.ntt of the psi-twisted sequence, re-indexed from sympy's root of unity to ours."""
    import sympy
    from sympy import GF, Poly, symbols
    from sympy.discrete.transforms import ntt as sympy_ntt
    from sympy.ntheory import primitive_root
    x = symbols("x")

    def poly(c, q):
        return Poly(list(reversed(c)), x, domain=GF(q, symmetric=False))

    def coeffs(pl, n, q):
        c = [int(v) % q for v in reversed(pl.all_coeffs())]
        return c + [0] * (n - len(c))

    out = {"sympy_version": sympy.__version__}
    # ---- ct x ct at N = 256, two limbs, batch 2 ----
    log2n, n = 8, 256
    moduli = [PRIMES_60[2][0], PRIME_30]
    g = po.SplitMix64(2560)
    A, B, Cc = [], [], []
    for _ in range(2):
        a = [[g.words_mod(n, q) for q in moduli] for _ in range(2)]
        b = [[g.words_mod(n, q) for q in moduli] for _ in range(2)]
        c = [[None] * len(moduli) for _ in range(3)]
        for l, q in enumerate(moduli):
            m = Poly(x**n + 1, x, domain=GF(q, symmetric=False))
            a0, a1, b0, b1 = (poly(v[l], q) for v in (a[0], a[1], b[0], b[1]))
            c[0][l] = coeffs((a0 * b0).rem(m), n, q)
            c[1][l] = coeffs((a0 * b1 + a1 * b0).rem(m), n, q)
            c[2][l] = coeffs((a1 * b1).rem(m), n, q)
        A += po.flatten_ct(a); B += po.flatten_ct(b); Cc += po.flatten_ct(c)
    out["ct_mul_n256"] = {"log2n": log2n, "moduli": moduli, "psi": [pow(PRIMES_60[2][2], 8192 // n, moduli[0]), pow(PSI_30_N1024, 1024 // n, moduli[1])],
                          "batch": 2, "a": A, "b": B, "c": Cc}
    # ---- forward NTT at N = 64 ----
    log2n, n = 6, 64
    vecs = []
    for q, psi in ((PRIMES_60[2][0], pow(PRIMES_60[2][2], 8192 // n, PRIMES_60[2][0])), (PRIME_30, pow(PSI_30_N1024, 1024 // n, PRIME_30))):
        a = po.SplitMix64(640 + (q & 0xFF)).words_mod(n, q)
        pa = poly(a, q)
        brv = lambda k: int(format(k, "0%db" % log2n)[::-1], 2)
        by_eval = [int(pa.eval(pow(psi, 2 * brv(k) + 1, q))) % q for k in range(n)]
        # (ii) cyclic transform of the twisted sequence with sympy's own root rt = g^((q-1)/n); ours is omega = psi^2 = rt^t
        twisted = [a[j] * pow(psi, j, q) % q for j in range(n)]
        cyc = [int(v) % q for v in sympy_ntt(twisted, prime=q)]
        rt = pow(primitive_root(q), (q - 1) // n, q)
        omega = psi * psi % q
        t = next(t for t in range(1, n, 2) if pow(rt, t, q) == omega)      # omega^(j k) = rt^(j (t k)):  ours[k] = sympy[t k mod n]
        by_ntt = [cyc[(t * brv(k)) % n] for k in range(n)]
        assert by_eval == by_ntt, "sympy's two routes disagree"
        vecs.append({"log2n": log2n, "q": q, "psi": psi, "a": a, "ntt": by_eval})
    out["ntt_n64"] = vecs
    return out


if __name__ == "__main__":
    data = {
        "small_ntt": small_ntt_vectors(),
        "config1_ct_mul": config1_ct_mul(),
        "rns_ct_mul_small": rns_ct_mul_small(),
        "identities": identities(),
        "rns_ct_mul_n256": rns_ct_mul_n256(),
        "n4096_ntt_digest": n4096_ntt_digest(),
        "sympy_restatement": sympy_restatement(),
    }
    for k, v in data.items():
        with open(os.path.join(HERE, k + ".json"), "w") as f:
            json.dump(v, f, separators=(",", ":"))
        print(k, os.path.getsize(os.path.join(HERE, k + ".json")), "bytes")
