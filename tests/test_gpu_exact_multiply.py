"""GPU (-m gpu): dpfhe_base_extend / dpfhe_scale_round bit-exact against the oracle (itself pinned to the big-integer definition in
tests/test_base_extend_oracle.py), and Evaluator.multiply_exact - the BFV-style exact multiply around the fused ct x ct kernel -
decrypting to the product of the plaintexts mod t under a toy RLWE scheme in Python integers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from deeppowers_amd import _cabi  # noqa: E402
from deeppowers_amd.evaluator import Context, Evaluator, to_device, to_host  # noqa: E402
from deeppowers_amd.params import FheParams, ntt_primes  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from oracle.cbind import Oracle  # noqa: E402


def fold_params(log2n, limbs):
    """the first `limbs` primes of the N = 8192 chain (FheParams.n8192), on ring degree 2^log2n"""
    n = 1 << log2n
    big = ntt_primes(13, limbs)
    return FheParams(log2n, big.moduli, tuple(pow(s, 8192 // n, q) for s, q in zip(big.psi, big.moduli)))


def generic_params(log2n):
    n = 1 << log2n

    def prime(bits):
        q = (1 << bits) - ((1 << bits) - 1) % (2 * n)
        while not po.is_prime(q):
            q -= 2 * n
        return q
    qs = [prime(59), prime(50), prime(33), prime(58), prime(45)]
    return FheParams(log2n, tuple(qs), tuple(po.min_primitive_2n_root(n, q) for q in qs))


@pytest.mark.parametrize("make,log2n", [(fold_params, 13), (fold_params, 9), (generic_params, 10)])
def test_base_extend_and_scale_round_bit_exact(make, log2n):
    p = make(log2n, 5) if make is fold_params else make(log2n)
    orc, ctx = Oracle.from_params(p), Context(p, 0)
    ev = Evaluator(ctx)
    rng = np.random.default_rng(11)
    L, n = p.n_limbs, p.n
    for ns, src0, dst0, nd in ((1, 0, 0, 5), (2, 0, 0, 5), (3, 2, 0, 2), (4, 1, 0, 5), (2, 3, 1, 3)):
        src = p.moduli[src0:src0 + ns]
        x = np.stack([rng.integers(0, q, (7, n), dtype=np.uint64) for q in src], axis=1)           # [7][ns][N]
        x[0, :, : n // 2] = np.array(src, np.uint64)[:, None] - np.uint64(1)                       # X = -1: the sign path of every lane
        Qs = int(np.prod([int(q) for q in src], dtype=object))
        for k, val in enumerate((Qs // 2, Qs // 2 + 1, Qs // 2 - 1, 0)):
            x[1, :, k] = [val % q for q in src]
        got = to_host(ev.base_extend(to_device(x, ctx.device), src0, dst0, nd))
        assert np.array_equal(got, orc.base_extend(x, src0, dst0, nd)), (ns, src0, dst0, nd)
    Q = int(np.prod([int(q) for q in p.moduli], dtype=object))
    for drop0, ndrop, keep0, nkeep, mul in ((0, 2, 2, 3, 65537), (4, 1, 0, 4, 1), (1, 3, 4, 1, 12289)):
        vals = [(int(rng.integers(0, 2**62)) * int(rng.integers(0, 2**62)) * int(rng.integers(0, 2**62)) * int(rng.integers(0, 2**60))) % (Q // (4 * mul)) * (1 if i % 2 else -1)
                for i in range(3 * n)]
        x = np.array([[[v % q for v in vals[b * n:(b + 1) * n]] for q in p.moduli] for b in range(3)], dtype=np.uint64)
        got = to_host(ev.scale_round(to_device(x, ctx.device), drop0, ndrop, keep0, nkeep, mul))
        assert np.array_equal(got, orc.scale_round(x, drop0, ndrop, keep0, nkeep, mul)), (drop0, ndrop, keep0, nkeep, mul)
    # argument checks
    t = to_device(np.zeros((1, 5, n), np.uint64), ctx.device)
    with pytest.raises(_cabi.DpfheError):
        ev.base_extend(t, 1, 0, 5)                      # source limbs 1 .. 5 of a five-limb context
    with pytest.raises(_cabi.DpfheError):
        ev.scale_round(t, 0, 2, 1, 3, 3)                # kept limbs overlap the dropped ones
    ctx.close()


@pytest.mark.parametrize("log2n", [13, 10])
def test_deep_level_limb_counts_bit_exact(log2n):
    """What a multiply at a FIVE-limb level needs (examples/encrypted_gpt2_stack.cpp: the first block of a two-block stack): extension 5 -> 11
    limbs, scale-and-round dropping 5 and keeping 6, extension back 6 -> 5 - and the entries' limits, 10 source and 20 destination limbs - each
    against the oracle; then the whole multiply_exact at that level against the oracle's pipeline."""
    p = fold_params(log2n, 20)
    orc, ctx = Oracle.from_params(p), Context(p, 0)
    ev = Evaluator(ctx)
    rng = np.random.default_rng(21)
    n = p.n
    for ns, src0, dst0, nd in ((5, 0, 0, 11), (6, 5, 0, 5), (10, 0, 0, 20), (7, 8, 1, 7), (10, 9, 0, 9)):
        src = p.moduli[src0:src0 + ns]
        x = np.stack([rng.integers(0, q, (3, n), dtype=np.uint64) for q in src], axis=1)
        x[0, :, : n // 2] = np.array(src, np.uint64)[:, None] - np.uint64(1)
        Qs = int(np.prod([int(q) for q in src], dtype=object))
        for k, val in enumerate((Qs // 2, Qs // 2 + 1, Qs // 2 - 1, 0)):
            x[1, :, k] = [val % q for q in src]
        got = to_host(ev.base_extend(to_device(x, ctx.device), src0, dst0, nd))
        assert np.array_equal(got, orc.base_extend(x, src0, dst0, nd)), (ns, src0, dst0, nd)
    x = np.stack([rng.integers(0, q, (2, n), dtype=np.uint64) for q in p.moduli], axis=1)        # any residues: the integer is what they represent
    for drop0, ndrop, keep0, nkeep, mul in ((0, 5, 5, 6, 65537), (0, 9, 9, 10, 65537), (10, 10, 0, 8, 1)):
        got = to_host(ev.scale_round(to_device(x, ctx.device), drop0, ndrop, keep0, nkeep, mul))
        assert np.array_equal(got, orc.scale_round(x, drop0, ndrop, keep0, nkeep, mul)), (drop0, ndrop, keep0, nkeep, mul)
    with pytest.raises(_cabi.DpfheError):
        ev.base_extend(to_device(np.zeros((1, 11, n), np.uint64), ctx.device), 0, 0, 4)           # 11 source limbs
    ctx.close()
    p = fold_params(log2n, 11)
    orc, ctx = Oracle.from_params(p), Context(p, 0)
    ev = Evaluator(ctx)
    ll, t = 5, 65537
    a = np.stack([rng.integers(0, q, (2, 2, n), dtype=np.uint64) for q in p.moduli[:ll]], axis=2)   # [2][2][ll][N]
    b = np.stack([rng.integers(0, q, (2, 2, n), dtype=np.uint64) for q in p.moduli[:ll]], axis=2)
    got = to_host(ev.multiply_exact(to_device(a, ctx.device), to_device(b, ctx.device), ll, t))
    A, B = orc.base_extend(a, 0, 0, 11), orc.base_extend(b, 0, 0, 11)
    T = orc.ct_mul(np.ascontiguousarray(A), np.ascontiguousarray(B), threads=0)
    want = orc.base_extend(orc.scale_round(T, 0, ll, ll, 11 - ll, t), ll, 0, ll)
    assert got.shape == (2, 3, ll, n) and np.array_equal(got, want)
    ctx.close()


def test_multiply_exact_at_n16384_equals_the_oracle_pipeline():
    """Round 5: the exact multiply at N = 16384 (five primes = 1 mod 2^15, ciphertexts at the two-limb level: examples/encrypted_gpt2_block_act ... 14),
    where the tensor product is composed from the batched transforms: base extension, multiply, scale-and-round and the extension back, each and
    together, word for word against the oracle."""
    p = ntt_primes(14, 5)
    orc, ctx = Oracle.from_params(p), Context(p, 0)
    ev = Evaluator(ctx)
    rng = np.random.default_rng(14)
    ll, t, n = 2, 65537, p.n
    a = np.stack([rng.integers(0, q, (3, 2, n), dtype=np.uint64) for q in p.moduli[:ll]], axis=2)   # [3][2][ll][N]
    b = np.stack([rng.integers(0, q, (3, 2, n), dtype=np.uint64) for q in p.moduli[:ll]], axis=2)
    A, B = orc.base_extend(a, 0, 0, 5), orc.base_extend(b, 0, 0, 5)
    assert np.array_equal(to_host(ev.base_extend(to_device(a, ctx.device), 0, 0, 5)), A)
    T = orc.ct_mul(np.ascontiguousarray(A), np.ascontiguousarray(B), threads=0)
    W = orc.scale_round(T, 0, ll, ll, 5 - ll, t)
    assert np.array_equal(to_host(ev.scale_round(to_device(T, ctx.device), 0, ll, ll, 5 - ll, t)), W)
    want = orc.base_extend(W, ll, 0, ll)
    got = to_host(ev.multiply_exact(to_device(a, ctx.device), to_device(b, ctx.device), ll, t))
    assert got.shape == (3, 3, ll, n) and np.array_equal(got, want)
    sq = to_host(ev.multiply_exact(to_device(a, ctx.device), to_device(a, ctx.device), ll, t))      # (squaring shares one extension: the activation's shape)
    Ts = orc.ct_mul(np.ascontiguousarray(A), np.ascontiguousarray(A), threads=0)
    assert np.array_equal(sq, orc.base_extend(orc.scale_round(Ts, 0, ll, ll, 5 - ll, t), ll, 0, ll))
    ctx.close()


def test_multiply_exact_equals_the_oracle_pipeline_and_decrypts_to_the_product():
    """N = 8192, five 60-bit limbs, ciphertexts at the two-limb level, t = 65537 (the configuration of the activated FFN example): the GPU
    pipeline equals the oracle's word for word; on a small ring the result decrypts to m1 * m2 mod (X^N + 1, t) under a toy BFV scheme."""
    p = fold_params(13, 5)
    orc, ctx = Oracle.from_params(p), Context(p, 0)
    ev = Evaluator(ctx)
    rng = np.random.default_rng(12)
    ll, t, n = 2, 65537, p.n
    a = np.stack([rng.integers(0, q, (3, 2, n), dtype=np.uint64) for q in p.moduli[:ll]], axis=2)   # [3][2][ll][N]
    b = np.stack([rng.integers(0, q, (3, 2, n), dtype=np.uint64) for q in p.moduli[:ll]], axis=2)
    got = to_host(ev.multiply_exact(to_device(a, ctx.device), to_device(b, ctx.device), ll, t))
    A, B = orc.base_extend(a, 0, 0, 5), orc.base_extend(b, 0, 0, 5)
    T = orc.ct_mul(np.ascontiguousarray(A), np.ascontiguousarray(B), threads=0)
    want = orc.base_extend(orc.scale_round(T, 0, ll, ll, 5 - ll, t), ll, 0, ll)
    assert got.shape == (3, 3, ll, n) and np.array_equal(got, want)
    with pytest.raises(_cabi.DpfheError):
        ev.multiply_exact(to_device(a, ctx.device)[:, :, :1].contiguous(), to_device(b, ctx.device), ll, t)   # shape mismatch
    ctx.close()

    # decryption-level check on a small ring (Python integers): Delta = floor(q / t), phase(ct) = Delta m + e
    p = fold_params(8, 5)
    ctx = Context(p, 0)
    ev = Evaluator(ctx)
    n = p.n
    q = p.moduli[0] * p.moduli[1]
    delta = q // t
    s = rng.integers(-1, 2, n)

    def encrypt(m):
        e = rng.integers(-8, 9, n)
        ct = np.zeros((2, ll, n), np.uint64)
        for i, qi in enumerate(p.moduli[:ll]):
            a_ = [int(rng.integers(0, 2**62)) % qi for _ in range(n)]
            a_s = po.negacyclic_schoolbook(a_, [int(v) % qi for v in s], qi)
            ct[0, i] = [(-a_s[k] + int(e[k]) + delta * int(m[k])) % qi for k in range(n)]
            ct[1, i] = a_
        return ct
    m1, m2 = rng.integers(0, t, n), rng.integers(0, t, n)
    c1, c2 = encrypt(m1), encrypt(m2)
    c3 = to_host(ev.multiply_exact(to_device(c1[None], ctx.device), to_device(c2[None], ctx.device), ll, t))[0]    # [3][ll][N]
    ph = None
    for i, qi in enumerate(p.moduli[:ll]):          # phase = c0 + c1 s + c2 s^2 per limb, then CRT
        sq = [int(v) % qi for v in s]
        s2 = po.negacyclic_schoolbook(sq, sq, qi)
        acc = po.poly_add(po.poly_add([int(v) for v in c3[0, i]], po.negacyclic_schoolbook([int(v) for v in c3[1, i]], sq, qi), qi),
                          po.negacyclic_schoolbook([int(v) for v in c3[2, i]], s2, qi), qi)
        ph = [acc] if ph is None else ph + [acc]
    want_m = po.negacyclic_schoolbook([int(v) for v in m1], [int(v) for v in m2], t)
    for k in range(n):
        x = po.crt_centered([ph[i][k] for i in range(ll)], p.moduli[:ll])
        assert ((2 * t * x + q) // (2 * q)) % t == want_m[k], k
    ctx.close()
