"""-m gpu: the per-limb arithmetic classes (round 6; include/dpfhe.h dpfhe_ctx_limb_class) through the C ABI against the oracle.

A context's limbs no longer share one arithmetic: the batched transforms and the fused multiply run each limb on the fastest policy its prime admits -
fold (2^60 - d), f64 (any prime below 2^47: doubles inside a transform), fold_scaled (2^k - d0 carried as 2^60 - d), f64_wide (the other primes below 2^50),
shoup (the rest) - in one launch for the transforms of most mixtures, one launch per class present otherwise.  Results must be the same words whatever the class: every case below compares whole buffers with the oracle (radix-2 Harvey NTT + u128
schoolbook, a different algorithm from all four)."""
import numpy as np
import pytest

from deeppowers_amd.params import FheParams, PRIMES_60, ntt_primes
from oracle import pyoracle as po
from oracle.cbind import Oracle


def primes_of(log2n, widths):
    """one prime = 1 mod 2N per requested width, distinct (the j-th largest below 2^w for the j-th request of width w); a NEGATIVE width -w asks for a
    prime of w bits that only the generic (Shoup) class takes"""
    qs, ps, seen = [], [], {}
    for w in widths:
        j = seen.get(w, 0)
        seen[w] = j + 1
        if w < 0:
            sq, sp = shoup_class_primes(log2n, -w, j + 1)
            qs.append(sq[j])
            ps.append(sp[j])
            continue
        p = ntt_primes(log2n, j + 1, w)
        qs.append(p.moduli[j])
        ps.append(p.psi[j])
    return FheParams(log2n, tuple(qs), tuple(ps))


def expected_class(q):
    if q < (1 << 60) and (1 << 60) - q < (1 << 24):
        return "fold"
    if q < (1 << 47):
        return "f64"
    k = q.bit_length()
    if 48 <= k <= 59 and (((1 << k) - q) << (60 - k)) < (1 << 24):
        return "fold_scaled"
    if q < (1 << 50):
        return "f64_wide"
    return "shoup"


def shoup_class_primes(log2n, bits, count):
    """`count` primes = 1 mod 2N of `bits` (51 ... 59) bits that NO fast class takes: too wide for the doubles, too far below 2^bits for the scaled fold"""
    from deeppowers_amd.params import is_prime, min_primitive_2n_root
    n = 1 << log2n
    qs, q = [], (1 << bits) - ((1 << bits) - 1) % (2 * n)
    while len(qs) < count:
        if is_prime(q) and expected_class(q) == "shoup":
            qs.append(q)
        q -= 2 * n
    return qs, [min_primitive_2n_root(n, v) for v in qs]


def worst_case(x, qcol, n):
    """stripes of extreme residues in the first item: q - 1 everywhere in one stretch, alternating q - 1 / 0 in another, the half point in a third"""
    x[0, ..., : n // 8] = qcol - np.uint64(1)
    x[0, ..., n // 8: n // 4: 2] = qcol - np.uint64(1)
    x[0, ..., n // 8 + 1: n // 4: 2] = 0
    x[0, ..., n // 4: n // 4 + n // 8] = qcol // np.uint64(2)
    return x


CASES = [
    # (name, log2n, widths)
    ("f64_30x4", 12, (30, 30, 30, 30)),
    ("f64_mixed_widths", 12, (20, 31, 40, 46)),
    ("f64_46x3_n8192", 13, (46, 46, 46)),
    ("f64_n16384", 14, (36, 45)),
    ("f64_n256", 8, (30, 46)),
    ("fscaled_59x4", 12, (59, 59, 59, 59)),
    ("fscaled_widths", 12, (59, 58, 57, 56)),
    ("fscaled_n8192", 13, (59, 58)),
    ("fscaled_n16384", 14, (59, 57)),
    ("shoup_55x2", 12, (-55, -55)),
    ("f64_wide_49_48_50", 12, (49, 48, 50)),
    ("f64_wide_n8192", 13, (50, 49)),
    ("f64_wide_n16384", 14, (49, 50)),
    ("f64_wide_n256", 8, (50, 48)),
    ("all_five_classes", 12, (60, 40, 59, -53, 49)),
    ("wide_next_to_f64_one_launch_per_class", 12, (49, 30)),
    ("seal_like_60_40_40_60", 12, (60, 40, 40, 60)),
    ("bench_mixed_59_50_40_33", 12, (59, 50, 40, 33)),
    ("config1_like_30bit_n1024", 10, (30,)),
    ("all_five_classes_n8192", 13, (60, 33, 59, 49, -56, 46)),
    ("all_five_classes_n2048", 11, (49, 60, 45, 58, -52)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,log2n,widths", CASES, ids=[c[0] for c in CASES])
def test_limb_classes_transforms_and_fused_multiply_match_the_oracle(name, log2n, widths):
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host
    p = primes_of(log2n, widths)
    L, n = p.n_limbs, p.n
    orc = Oracle.from_params(p)
    ctx = Context(p, 0)
    ev = Evaluator(ctx)
    try:
        want_cls = tuple(expected_class(q) for q in p.moduli)
        if all(c == "shoup" for c in want_cls) or all(c == "fold" for c in want_cls):
            assert ctx.limb_classes == want_cls   # uniform contexts report the context-wide policy
        else:
            assert ctx.limb_classes == want_cls, (ctx.limb_classes, want_cls)
        qcol = np.array(p.moduli, np.uint64)[:, None]
        batch = 5
        x = worst_case(orc.fill(batch, 4100).reshape(batch, L, n), qcol, n)
        X = ev.ntt_forward(to_device(x, ctx.device))
        assert np.array_equal(to_host(X), orc.ntt_fwd(x, threads=0))
        assert np.array_equal(to_host(ev.ntt_inverse(to_device(x, ctx.device))), orc.ntt_inv(x, threads=0))   # inverse of non-image data
        assert np.array_equal(to_host(ev.ntt_inverse(X)), x)
        if log2n <= 13:
            a = worst_case(orc.fill(batch * 2, 4200).reshape(batch, 2, L, n), qcol, n)
            b = worst_case(orc.fill(batch * 2, 4300).reshape(batch, 2, L, n), qcol, n)
            want = orc.ct_mul(a, b, threads=0)
            A, B = Ciphertext(to_device(a, ctx.device)), Ciphertext(to_device(b, ctx.device))
            assert np.array_equal(to_host(ev.multiply(A, B).data), want)
            want_ntt = orc.ntt_fwd(want.reshape(-1, L, n), threads=0).reshape(want.shape)
            assert np.array_equal(to_host(ev.multiply(A, B, out_ntt=True).data), want_ntt)
            An = Ciphertext(ev.ntt_forward(A.data.view(-1, L, n)).view(batch, 2, L, n), is_ntt=True)
            Bn = Ciphertext(ev.ntt_forward(B.data.view(-1, L, n)).view(batch, 2, L, n), is_ntt=True)
            assert np.array_equal(to_host(ev.multiply(An, Bn, out_ntt=False).data), want)
            assert np.array_equal(to_host(ev.multiply(An, Bn, out_ntt=True).data), want_ntt)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("widths", [(30, 30, 30, 30), (59, 50, 40, 33), (59, 59, 58, 58), (49, 49, 50, 48)], ids=["f64", "mixed", "fold_scaled", "f64_wide"])
def test_limb_classes_full_size_whole_buffer_oracle(widths):
    """BASELINE configs[1]'s batch (1024 RNS polynomials = 4096 residue polynomials, N = 4096) and 1024 ciphertext pairs, every word against the oracle,
    plus the size-independent properties: round trip, linearity of the transform"""
    import torch
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host
    p = primes_of(12, widths)
    L, n = p.n_limbs, p.n
    orc = Oracle.from_params(p)
    ctx = Context(p, 0)
    ev = Evaluator(ctx)
    try:
        qcol = np.array(p.moduli, np.uint64)[:, None]
        x = worst_case(orc.fill(1024, 5100).reshape(1024, L, n), qcol, n)
        dx = to_device(x, ctx.device)
        X = ev.ntt_forward(dx)
        assert np.array_equal(to_host(X), orc.ntt_fwd(x, threads=0))
        assert torch.equal(ev.ntt_inverse(X), dx)
        y = orc.fill(1024, 5200).reshape(1024, L, n)
        s = ((x.astype(object) + y.astype(object)) % qcol.astype(object)).astype(np.uint64)
        Y, S = ev.ntt_forward(to_device(y, ctx.device)), ev.ntt_forward(to_device(s, ctx.device))
        assert torch.equal(ev.add_words(X, Y), S)   # NTT(x + y) = NTT(x) + NTT(y)
        pairs = 1024
        a = worst_case(orc.fill(pairs * 2, 5300).reshape(pairs, 2, L, n), qcol, n)
        b = worst_case(orc.fill(pairs * 2, 5400).reshape(pairs, 2, L, n), qcol, n)
        got = to_host(ev.multiply(Ciphertext(to_device(a, ctx.device)), Ciphertext(to_device(b, ctx.device))).data)
        assert np.array_equal(got, orc.ct_mul(a, b, threads=0))
    finally:
        ctx.close()


@pytest.mark.gpu
def test_key_switching_on_a_context_with_classes_still_matches_the_oracle():
    """the key-switching kernels of a non-uniform context read its complete generic tables; the transforms and the fused multiply around them run per class"""
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host
    p = primes_of(12, (60, 40, 59, -54, 49))
    L, n = p.n_limbs, p.n
    orc = Oracle.from_params(p)
    ctx = Context(p, 0)
    ev = Evaluator(ctx)
    try:
        assert ctx.limb_classes == ("fold", "f64", "fold_scaled", "shoup", "f64_wide")
        a = orc.fill(6, 6100).reshape(3, 2, L, n)
        b = orc.fill(6, 6200).reshape(3, 2, L, n)
        want = orc.ct_mul(a, b, threads=0)
        c = ev.multiply(Ciphertext(to_device(a, ctx.device)), Ciphertext(to_device(b, ctx.device)))
        assert np.array_equal(to_host(c.data), want)
        evk = orc.fill(L * 2, 6300).reshape(L, 2, L, n)
        assert np.array_equal(to_host(ev.relinearize(c, to_device(evk, ctx.device)).data), orc.relinearize(want, evk, threads=0))
        Ld = L - 1
        data = Oracle(p.log2_n, p.moduli[:-1], p.psi[:-1])
        key = orc.fill(Ld * 2, 6400).reshape(Ld, 2, L, n)
        ct = data.fill(3 * 3, 6500).reshape(3, 3, Ld, n)
        got = to_host(ev.keyswitch_hybrid(Ciphertext(to_device(ct, ctx.device)), to_device(key, ctx.device)).data)
        assert np.array_equal(got, orc.keyswitch_hybrid(ct, key, 3, threads=0))
        # hoisted rotations (hoisted_ks_kernel, one launch per class) and the Galois inverse transform (ntt_inv_galois_kernel, per class) on the mixture
        k, T = 70, 2
        elts = [pow(3, i + 1, 2 * n) for i in range(k)]
        keys = orc.fill(k * Ld * 2, 6600).reshape(k, Ld, 2, L, n)
        cts = data.fill(T * 2, 6700).reshape(T, 2, Ld, n)
        got = to_host(ev.rotate_hybrid_hoisted(Ciphertext(to_device(cts, ctx.device)), elts, to_device(keys, ctx.device)).data).reshape(k, T, 2, Ld, n)
        for r in (0, 22, 63, 64, 69):
            for t in range(T):
                assert np.array_equal(got[r, t], orc.rotate_hoisted(cts[t], [elts[r]], keys[r][None], threads=0)[0]), ("hoisted", r, t)
        ge = [pow(3, 5 * i, 2 * n) for i in range(70)]
        x = orc.fill(70 * 2, 6800).reshape(70, 2, L, n)
        wantg = np.stack([orc.apply_galois(orc.ntt_inv(x[e]), ge[e]) for e in range(70)])
        d = to_device(x, ctx.device)
        assert np.array_equal(to_host(ev.ntt_inverse_galois(d, ge)), wantg)
        ev.ntt_inverse_galois(d, ge, out=d)
        assert np.array_equal(to_host(d), wantg)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_sympy_fixture_replayed_on_the_device(golden_dir):
    """tests/golden/sympy_restatement.json (sympy's GF(q)[X] arithmetic alone) against the HIP path directly: a 60-bit fold limb next to the 30-bit prime
    (f64 class), N = 256 - the device words must be sympy's words, with no oracle of this build in between."""
    import json
    import os
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host
    with open(os.path.join(golden_dir, "sympy_restatement.json")) as f:
        v = json.load(f)["ct_mul_n256"]
    p = FheParams(v["log2n"], tuple(v["moduli"]), tuple(v["psi"]))
    ctx = Context(p, 0)
    ev = Evaluator(ctx)
    try:
        assert ctx.limb_classes == ("fold", "f64")
        n, L = p.n, p.n_limbs
        a = np.array(v["a"], np.uint64).reshape(v["batch"], 2, L, n)
        b = np.array(v["b"], np.uint64).reshape(v["batch"], 2, L, n)
        c = np.array(v["c"], np.uint64).reshape(v["batch"], 3, L, n)
        A, B = Ciphertext(to_device(a, ctx.device)), Ciphertext(to_device(b, ctx.device))
        assert np.array_equal(to_host(ev.multiply(A, B).data), c)
        got = ev.ntt_inverse(ev.multiply(A, B, out_ntt=True).data)
        assert np.array_equal(to_host(got), c)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,log2n,widths", [("f64", 12, (30, 31, 40, 46)), ("f64_n8192", 13, (45, 46, 33)), ("f64_wide", 12, (49, 49, 48)), ("fold_scaled", 12, (59, 58, 59)),
                                               ("fold_scaled_n8192", 13, (59, 58, 57)), ("f64_n1024", 10, (30, 30))], ids=lambda v: v if isinstance(v, str) else None)
def test_key_switching_runs_on_the_class_of_a_uniform_context(name, log2n, widths):
    """a context whose limbs all share ONE class key-switches on that class's kernels (dpfhe_cabi.hip with_policy: the generic forms of relin_kernel /
    hoisted_ks_kernel / ntt_inv_galois_kernel instantiated for the class): relinearisation, the plain key switch behind an automorphism, hybrid key switching
    with 2- and 3-component inputs, hoisted rotations - every word against the oracle"""
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host
    p = primes_of(log2n, widths)
    L, n = p.n_limbs, p.n
    orc = Oracle.from_params(p)
    ctx = Context(p, 0)
    ev = Evaluator(ctx)
    try:
        assert len(set(ctx.limb_classes)) == 1 and ctx.limb_classes[0] == name.split("_n")[0]
        batch = 3
        a = orc.fill(batch * 2, 8100).reshape(batch, 2, L, n)
        b = orc.fill(batch * 2, 8200).reshape(batch, 2, L, n)
        want = orc.ct_mul(a, b, threads=0)
        c = ev.multiply(Ciphertext(to_device(a, ctx.device)), Ciphertext(to_device(b, ctx.device)))
        assert np.array_equal(to_host(c.data), want)
        evk = orc.fill(L * 2, 8300).reshape(L, 2, L, n)
        assert np.array_equal(to_host(ev.relinearize(c, to_device(evk, ctx.device)).data), orc.relinearize(want, evk, threads=0))
        got = ev.apply_galois(Ciphertext(to_device(a, ctx.device)), 5, to_device(evk, ctx.device))        # automorphism + dpfhe_switch_key
        assert np.array_equal(to_host(got.data), orc.switch_key(orc.apply_galois(a, 5), evk, threads=0))
        Ld = L - 1
        data = Oracle(p.log2_n, p.moduli[:-1], p.psi[:-1])
        key = orc.fill(Ld * 2, 8400).reshape(Ld, 2, L, n)
        for comps in (2, 3):
            ct = data.fill(batch * comps, 8500 + comps).reshape(batch, comps, Ld, n)
            got = to_host(ev.keyswitch_hybrid(Ciphertext(to_device(ct, ctx.device)), to_device(key, ctx.device)).data)
            assert np.array_equal(got, orc.keyswitch_hybrid(ct, key, comps, threads=0)), comps
        # rotations: one key per item (batched), hoisted (hoisted_ks_kernel: permuted digits, 70 rotations = two launch groups), grouped (key-major giant steps)
        k, T = 70, 2
        elts = [pow(3, i + 1, 2 * n) for i in range(k)]
        elts[-1] = 2 * n - 1
        keys = orc.fill(k * Ld * 2, 8600).reshape(k, Ld, 2, L, n)
        dk = to_device(keys, ctx.device)
        cts = data.fill(T * 2, 8700).reshape(T, 2, Ld, n)
        got = to_host(ev.rotate_hybrid_batch(Ciphertext(to_device(cts[:1], ctx.device)), elts, dk).data)
        for i in (0, 1, 63, 64, 69):
            assert np.array_equal(got[i], orc.keyswitch_hybrid(data.apply_galois(cts[:1], elts[i]), keys[i], 2, threads=0)[0]), ("batch", i)
        got = to_host(ev.rotate_hybrid_hoisted(Ciphertext(to_device(cts, ctx.device)), elts, dk).data).reshape(k, T, 2, Ld, n)
        for r in (0, 22, 63, 64, 69):
            for t in range(T):
                assert np.array_equal(got[r, t], orc.rotate_hoisted(cts[t], [elts[r]], keys[r][None], threads=0)[0]), ("hoisted", r, t)
        items = data.fill(k * T * 2, 8800).reshape(k * T, 2, Ld, n)
        got = to_host(ev.rotate_hybrid_grouped(Ciphertext(to_device(items, ctx.device)), elts, T, dk).data)
        for i in (0, 1, 2, 127, 128, k * T - 1):
            assert np.array_equal(got[i], orc.keyswitch_hybrid(data.apply_galois(items[i][None], elts[i // T]), keys[i // T], 2, threads=0)[0]), ("grouped", i)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("widths", [(36, 45, 40), (59, 57, 45), (49, 50)], ids=["f64", "mixed", "f64_wide_and_scaled"])
def test_large_ring_composed_operations_on_limb_classes(widths):
    """N = 16384 has no fused kernels: dpfhe_ct_mul / dpfhe_relinearize / hybrid key switching are composed from the batched transforms (per limb class, the
    mixtures in one launch) and the streaming kernels (generic products on canonical words) - every word against the oracle"""
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host
    p = primes_of(14, widths)
    L, n = p.n_limbs, p.n
    orc = Oracle.from_params(p)
    ctx = Context(p, 0)
    ev = Evaluator(ctx)
    try:
        assert ctx.limb_classes == tuple(expected_class(q) for q in p.moduli)
        qcol = np.array(p.moduli, np.uint64)[:, None]
        a = worst_case(orc.fill(4, 9100).reshape(2, 2, L, n), qcol, n)
        b = worst_case(orc.fill(4, 9200).reshape(2, 2, L, n), qcol, n)
        want = orc.ct_mul(a, b, threads=0)
        A, B = Ciphertext(to_device(a, ctx.device)), Ciphertext(to_device(b, ctx.device))
        c = ev.multiply(A, B)
        assert np.array_equal(to_host(c.data), want)
        assert np.array_equal(to_host(ev.ntt_inverse(ev.multiply(A, B, out_ntt=True).data)), want)
        evk = orc.fill(L * 2, 9300).reshape(L, 2, L, n)
        assert np.array_equal(to_host(ev.relinearize(c, to_device(evk, ctx.device)).data), orc.relinearize(want, evk, threads=0))
        Ld = L - 1
        data = Oracle(p.log2_n, p.moduli[:-1], p.psi[:-1])
        key = orc.fill(Ld * 2, 9400).reshape(Ld, 2, L, n)
        ct = data.fill(2 * 2, 9500).reshape(2, 2, Ld, n)
        got = to_host(ev.keyswitch_hybrid(Ciphertext(to_device(ct, ctx.device)), to_device(key, ctx.device)).data)
        assert np.array_equal(got, orc.keyswitch_hybrid(ct, key, 2, threads=0))
        ctx.release_scratch(all_streams=True)
        assert ctx.scratch_bytes == 0
    finally:
        ctx.close()
