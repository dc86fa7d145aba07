"""-m gpu: randomised parameter sets (ring degree, limb count, prime widths, batch sizes) through the C ABI against the oracle.

The pinned BASELINE configurations have their own tests; this one walks the corners between them - every fused-kernel geometry
(N = 256 ... 8192) with 1 ... 7 limbs, primes of mixed widths (the fold-reduction kernels need every prime of the form 2^60 - d, one
narrower prime sends the whole context to the generic kernels), odd batch sizes - so that a dispatch rule (quad / pair / single-transform
multiply, shared-digit key switch, XCD-aware block ids, split transforms) is never exercised at one size only.  Seeds are fixed:
a failure reproduces."""
import numpy as np
import pytest

from deeppowers_amd.params import FheParams, ntt_primes
from oracle.cbind import Oracle

SEEDS = list(range(20))


def _random_params(rng):
    log2n = int(rng.integers(8, 14))
    limbs = int(rng.integers(1, 8))
    if rng.integers(0, 2):
        return ntt_primes(log2n, limbs, 60)                       # all of the form 2^60 - d: fold-reduction kernels
    widths = [int(rng.integers(31, 61)) for _ in range(limbs)]    # mixed widths: generic (Shoup/Barrett) kernels
    qs, psis = [], []
    for w in widths:
        p = ntt_primes(log2n, limbs, w)                           # `limbs` candidates per width: take the first one not used yet
        q, psi = next((q, s) for q, s in zip(p.moduli, p.psi) if q not in qs)
        qs.append(q)
        psis.append(psi)
    return FheParams(log2n, tuple(qs), tuple(psis))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_random_parameter_set_matches_the_oracle(seed):
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host
    rng = np.random.default_rng(1000 + seed)
    p = _random_params(rng)
    L, n = p.n_limbs, p.n
    orc = Oracle.from_params(p)
    ctx = Context(p, 0)
    ev = Evaluator(ctx)
    try:
        batch = int(rng.integers(1, 6))
        qcol = np.array(p.moduli, np.uint64)[None, :, None]
        # transforms: forward, inverse, round trip (odd number of RNS polynomials)
        x = orc.fill(batch, 7000 + seed).reshape(batch, L, n)
        x[0, :, : n // 8] = qcol[0] - np.uint64(1)
        X = ev.ntt_forward(to_device(x, ctx.device))
        assert np.array_equal(to_host(X), orc.ntt_fwd(x, threads=0))
        assert np.array_equal(to_host(ev.ntt_inverse(to_device(x, ctx.device))), orc.ntt_inv(x, threads=0))
        assert np.array_equal(to_host(ev.ntt_inverse(X)), x)
        # the metric op in both output domains
        a = orc.fill(batch * 2, 7100 + seed).reshape(batch, 2, L, n)
        b = orc.fill(batch * 2, 7200 + seed).reshape(batch, 2, L, n)
        a[0, :, :, -(n // 8):] = qcol - np.uint64(1)
        b[0, :, :, -(n // 8):] = qcol - np.uint64(1)
        want = orc.ct_mul(a, b, threads=0)
        A, B = Ciphertext(to_device(a, ctx.device)), Ciphertext(to_device(b, ctx.device))
        c = ev.multiply(A, B)
        assert np.array_equal(to_host(c.data), want)
        assert np.array_equal(to_host(ev.multiply(A, B, out_ntt=True).data), orc.ntt_fwd(want.reshape(-1, L, n), threads=0).reshape(want.shape))
        # relinearisation of the products (RNS-digit keys)
        evk = orc.fill(L * 2, 7300 + seed).reshape(L, 2, L, n)
        got = to_host(ev.relinearize(c, to_device(evk, ctx.device)).data)
        assert np.array_equal(got, orc.relinearize(want, evk, threads=0))
        # hybrid key switch (last limb = special prime) where there is a data limb to switch
        if L >= 2:
            Ld = L - 1
            data = Oracle(p.log2_n, p.moduli[:-1], p.psi[:-1])
            key = orc.fill(Ld * 2, 7400 + seed).reshape(Ld, 2, L, n)
            for comps in (2, 3):
                ct = data.fill(batch * comps, 7500 + seed + comps).reshape(batch, comps, Ld, n)
                w = orc.keyswitch_hybrid(ct, key, comps, threads=0)
                g = to_host(ev.keyswitch_hybrid(Ciphertext(to_device(ct, ctx.device)), to_device(key, ctx.device)).data)
                assert np.array_equal(g, w), comps
        # shard-local reduce over an odd batch
        r = to_host(ev.reduce_sum(c).data)
        assert np.array_equal(r.reshape(3, L, n), orc.reduce_sum(want, 3).reshape(3, L, n))
    finally:
        ctx.close()


# ---- round 6: hypothesis-driven limb widths (every arithmetic class and every mixture of them) ---------------------------------------------------------
from hypothesis import HealthCheck, given, settings, strategies as st  # noqa: E402


def _kth_prime(log2n, bits, k):
    p = ntt_primes(log2n, k + 1, bits)
    return p.moduli[k], p.psi[k]


@pytest.mark.gpu
@settings(max_examples=25, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(log2n=st.integers(8, 13), widths=st.lists(st.tuples(st.integers(20, 60), st.integers(0, 3)), min_size=1, max_size=6, unique=True), seed=st.integers(0, 2**31))
def test_random_limb_widths_every_class_matches_the_oracle(log2n, widths, seed):
    """(width k, index j) -> the j-th largest prime = 1 mod 2N below 2^k: 20 ... 60 bits, so every limb class (f64, f64_wide, fold_scaled, fold, shoup) and every
    mixture of them turns up; transforms, the fused multiply in both output domains and relinearisation against the oracle, extreme residues included"""
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host
    n = 1 << log2n
    qs, psis = [], []
    for bits, j in widths:
        if bits < log2n + 8:                      # too few primes = 1 mod 2N below 2^bits
            bits = log2n + 8 + bits % 8
        q, psi = _kth_prime(log2n, bits, j)
        if q in qs:
            continue
        qs.append(q)
        psis.append(psi)
    p = FheParams(log2n, tuple(qs), tuple(psis))
    L = p.n_limbs
    orc = Oracle.from_params(p)
    ctx = Context(p, 0)
    ev = Evaluator(ctx)
    try:
        qcol = np.array(p.moduli, np.uint64)[None, :, None]
        batch = 1 + seed % 3
        x = orc.fill(batch, seed % 100000).reshape(batch, L, n)
        x[0, :, : n // 4] = qcol[0] - np.uint64(1)
        X = ev.ntt_forward(to_device(x, ctx.device))
        assert np.array_equal(to_host(X), orc.ntt_fwd(x, threads=0)), ctx.limb_classes
        assert np.array_equal(to_host(ev.ntt_inverse(to_device(x, ctx.device))), orc.ntt_inv(x, threads=0)), ctx.limb_classes
        a = orc.fill(batch * 2, seed % 100000 + 1).reshape(batch, 2, L, n)
        b = orc.fill(batch * 2, seed % 100000 + 2).reshape(batch, 2, L, n)
        a[0, :, :, -(n // 4):] = qcol - np.uint64(1)
        b[0, :, :, -(n // 4):] = qcol - np.uint64(1)
        want = orc.ct_mul(a, b, threads=0)
        A, B = Ciphertext(to_device(a, ctx.device)), Ciphertext(to_device(b, ctx.device))
        c = ev.multiply(A, B)
        assert np.array_equal(to_host(c.data), want), ctx.limb_classes
        assert np.array_equal(to_host(ev.ntt_inverse(ev.multiply(A, B, out_ntt=True).data)), want), ctx.limb_classes
        evk = orc.fill(L * 2, seed % 100000 + 3).reshape(L, 2, L, n)
        assert np.array_equal(to_host(ev.relinearize(c, to_device(evk, ctx.device)).data), orc.relinearize(want, evk, threads=0)), ctx.limb_classes
    finally:
        ctx.close()
