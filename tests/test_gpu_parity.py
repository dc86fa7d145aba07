"""GPU parity tests (-m gpu): the HIP path through the C ABI vs the CPU oracle and the committed golden
vectors.  Integer work: every comparison is BIT-EXACT.  Nothing here reads /root/reference."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from deeppowers_amd import _cabi  # noqa: E402
from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, Plaintext, to_device, to_host  # noqa: E402
from deeppowers_amd.params import PRIMES_60, PRIME_30, PSI_30_N1024, FheParams  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from oracle.cbind import Oracle  # noqa: E402


def sha(arr):
    return hashlib.sha256(np.ascontiguousarray(arr, dtype="<u8").tobytes()).hexdigest()


def load(golden_dir, name):
    with open(os.path.join(golden_dir, name + ".json")) as f:
        return json.load(f)


def generic_prime(bits, two_n):
    """largest prime < 2^bits that is 1 mod two_n (NOT of the 2^60 - d form: exercises ShoupArith)."""
    q = (1 << bits) - ((1 << bits) - 1) % two_n
    while not po.is_prime(q):
        q -= two_n
    return q


def fold_primes(two_n, count):
    """primes 2^60 - d (d < 2^24) that are 1 mod two_n: FoldArith-eligible limbs for any ring degree"""
    out = []
    k = 1
    while len(out) < count:
        c = (1 << 60) - (k * two_n - 1)
        assert (1 << 60) - c < (1 << 24), "no more fold-eligible primes for this ring degree"
        if po.is_prime(c):
            out.append(c)
        k += 1
    return out


class Rig:
    def __init__(self, params):
        self.p = params
        self.orc = Oracle.from_params(params)
        self.ctx = Context(params, 0)
        self.ev = Evaluator(self.ctx)

    def dev(self, a):
        return to_device(a, self.ctx.device)


@pytest.fixture(scope="module")
def rigs():
    cache = {}

    def get(name):
        if name not in cache:
            if name == "config1":
                p = FheParams.config1()
            elif name == "n4096":
                p = FheParams.n4096_l4()
            elif name == "n8192":
                p = FheParams.n8192_l6()
            elif name in ("fold15", "fold16"):  # split transform (column stages + 4096-point kernels)
                log2n = int(name[4:])
                qs = fold_primes(2 << log2n, 2)
                p = FheParams(log2n, tuple(qs), tuple(po.min_primitive_2n_root(1 << log2n, q) for q in qs))
            elif name == "fold14":  # N = 16384: the pinned primes that are 1 mod 32768 (transform / streaming kernels only)
                qs = [PRIMES_60[i][0] for i in (1, 2, 4)]
                p = FheParams(14, tuple(qs), tuple(po.min_primitive_2n_root(16384, q) for q in qs))
            elif name.startswith("shoup"):  # generic primes of several widths, 3 limbs
                log2n = int(name[5:])
                n = 1 << log2n
                qs = [generic_prime(59, 2 * n), generic_prime(50, 2 * n), generic_prime(33, 2 * n)]
                p = FheParams(log2n, tuple(qs), tuple(po.min_primitive_2n_root(n, q) for q in qs))
            else:  # foldN / foldNx3: pinned 60-bit primes at another N
                log2n = int(name[4:].split("x")[0])
                n = 1 << log2n
                qs = [PRIMES_60[i][0] for i in (0, 2, 5)]
                p = FheParams(log2n, tuple(qs), tuple(pow(PRIMES_60[i][2], 8192 // n, PRIMES_60[i][0]) for i in (0, 2, 5)))
            cache[name] = Rig(p)
        return cache[name]

    yield get
    for r in cache.values():
        r.ctx.close()


ALL = ["config1", "n4096", "n8192", "fold8", "fold9", "fold10", "fold11", "shoup8", "shoup10", "shoup12", "shoup13"]
NTT_ONLY = ["fold14", "shoup14", "fold15", "fold16", "shoup16"]   # N >= 16384: no fused ct x ct / key-switch kernels (composed ones: see the large-ring test)


def test_arithmetic_policy_selection(rigs):
    assert rigs("n4096").ctx.uses_fold and rigs("n8192").ctx.uses_fold
    assert not rigs("config1").ctx.uses_fold and not rigs("shoup12").ctx.uses_fold


# ---- golden vectors -------------------------------------------------------------------------------------
def test_config1_ct_mul_matches_appendix_b_digest(rigs, golden_dir):
    d = load(golden_dir, "config1_ct_mul")
    r = rigs("config1")
    ab = r.orc.fill(4, 1).reshape(4, 1, 1024)
    assert ab[0, 0, :4].tolist() == d["a0_head"]
    a = Ciphertext(r.dev(ab[:2].reshape(1, 2, 1, 1024)))
    b = Ciphertext(r.dev(ab[2:].reshape(1, 2, 1, 1024)))
    c = to_host(r.ev.multiply(a, b).data)
    assert sha(c) == d["sha256"]["c0c1c2"] == "9e97bb5cf219416d6cb3683b99de66ee8aaec17496504a97bf0471fdf43ab66b"
    assert c[0, 0, 0].tolist() == d["c0"] and c[0, 1, 0].tolist() == d["c1"] and c[0, 2, 0].tolist() == d["c2"]


def test_n4096_ntt_golden_digests(rigs, golden_dir):
    r = rigs("n4096")
    x = np.zeros((1, 4, 4096), np.uint64)
    vs = load(golden_dir, "n4096_ntt_digest")
    for v in vs:
        x[0, v["limb"]] = po.SplitMix64(v["seed"]).words_mod(4096, v["q"])
    y = to_host(r.ev.ntt_forward(r.dev(x)))
    for v in vs:
        assert y[0, v["limb"], :4].tolist() == v["ntt_head"] and sha(y[0, v["limb"]]) == v["ntt_sha256"]


def test_n256_bigint_fixture_mixed_fold_and_30bit_limbs(golden_dir):
    v = load(golden_dir, "rns_ct_mul_n256")
    p = FheParams(v["log2n"], tuple(v["moduli"]), tuple(v["psi"]))
    ctx = Context(p, 0)
    ev = Evaluator(ctx)
    assert not ctx.uses_fold and ctx.limb_classes == ("fold", "f64")   # round 6: the arithmetic is chosen per limb - the 30-bit prime no longer drags the 60-bit one along
    shape = (v["batch"], 2, 2, 256)
    a = np.array(v["a"], np.uint64).reshape(shape)
    b = np.array(v["b"], np.uint64).reshape(shape)
    c = to_host(ev.multiply(Ciphertext(to_device(a, ctx.device)), Ciphertext(to_device(b, ctx.device))).data)
    assert sha(c) == v["c_sha256"] and c.ravel()[:8].tolist() == v["c_head"] and c.ravel()[-8:].tolist() == v["c_tail"]
    na = to_host(ev.ntt_forward(to_device(a[0, 0], ctx.device)))
    assert na[0, :8].tolist() == v["ntt_a0_limb0_head"] and sha(na) == v["ntt_a0_sha256"]
    ctx.close()


def test_identities_on_device(rigs):
    r = rigs("n4096")
    n, L = 4096, 4
    delta = np.zeros((1, L, n), np.uint64); delta[:, :, 0] = 1
    assert np.all(to_host(r.ev.ntt_forward(r.dev(delta))) == 1)
    X = np.zeros((1, L, n), np.uint64); X[:, :, 1] = 1
    nx = to_host(r.ev.ntt_forward(r.dev(X)))
    for l, (q, psi) in enumerate(zip(r.p.moduli, r.p.psi)):
        for k in (0, 1, 2, 1234, 4095):
            assert int(nx[0, l, k]) == pow(psi, 2 * po.bit_reverse(k, 12) + 1, q)
    # (1 + X) * X^(N-1) = X^(N-1) - 1 through the fused multiply (a1 = b1 = 0)
    a = np.zeros((1, 2, L, n), np.uint64); a[0, 0, :, 0] = 1; a[0, 0, :, 1] = 1
    b = np.zeros((1, 2, L, n), np.uint64); b[0, 0, :, n - 1] = 1
    c = to_host(r.ev.multiply(Ciphertext(r.dev(a)), Ciphertext(r.dev(b))).data)
    for l, q in enumerate(r.p.moduli):
        want = np.zeros(n, np.uint64); want[0] = q - 1; want[n - 1] = 1
        assert np.array_equal(c[0, 0, l], want) and not c[0, 1, l].any() and not c[0, 2, l].any()


# ---- NTT vs oracle, every geometry and both arithmetic policies ---------------------------------------------
@pytest.mark.parametrize("name", ALL + NTT_ONLY)
def test_ntt_forward_inverse_vs_oracle(rigs, name):
    r = rigs(name)
    L, n = r.p.n_limbs, r.p.n
    for npolys, seed in ((1, 3), (5, 4)):
        x = r.orc.fill(npolys, seed)
        # extreme residues in the first polynomial
        x[0, :, : n // 2] = (np.array(r.p.moduli, np.uint64) - np.uint64(1))[:, None]
        if npolys > 1:
            x[1] = 0
            x[2] = (np.array(r.p.moduli, np.uint64) - np.uint64(1))[:, None]
        want = r.orc.ntt_fwd(x, threads=0)
        d = r.dev(x)
        got = to_host(r.ev.ntt_forward(d))
        assert np.array_equal(got, want)
        assert np.array_equal(to_host(d), x), "out-of-place transform modified its input"
        r.ev.ntt_forward_(d)
        assert np.array_equal(to_host(d), want)
        assert np.array_equal(to_host(r.ev.ntt_inverse(d)), x)
        r.ev.ntt_inverse_(d)
        assert np.array_equal(to_host(d), x)
        # inverse of arbitrary canonical data (not an NTT image) is the oracle's inverse too
        assert np.array_equal(to_host(r.ev.ntt_inverse(r.dev(x))), r.orc.ntt_inv(x, threads=0))
    assert L == r.p.n_limbs


@pytest.mark.parametrize("name", ["config1", "n4096", "n8192", "fold8", "shoup10", "shoup13", "fold14", "fold16"])
def test_dyadic_ops_vs_oracle(rigs, name):
    r = rigs(name)
    x = r.orc.fill(9, 17)
    a, b, acc = x[0:3].copy(), x[3:6].copy(), x[6:9].copy()
    qs = np.array(r.p.moduli, np.uint64)[:, None]
    a[0, :, :8] = 0; b[0, :, :8] = 0
    a[1, :, :8] = qs - np.uint64(1); b[1, :, :8] = qs - np.uint64(1); acc[1, :, :8] = qs - np.uint64(1)
    A, Bd, ACC = r.dev(a), r.dev(b), r.dev(acc)
    assert np.array_equal(to_host(r.ev.dyadic_mul(A, Bd)), r.orc.dyadic("mul", a, b))
    assert np.array_equal(to_host(r.ev.add_words(A, Bd)), r.orc.dyadic("add", a, b))
    assert np.array_equal(to_host(r.ev.sub_words(A, Bd)), r.orc.dyadic("sub", a, b))
    assert np.array_equal(to_host(r.ev.negate_words(A)), r.orc.dyadic("negate", a))
    r.ev.dyadic_mul_add_(ACC, A, Bd)
    assert np.array_equal(to_host(ACC), r.orc.dyadic("mul_add", a, b, acc=acc))
    out = A.clone()  # out aliases a
    r.ev.dyadic_mul(out, Bd, out=out)
    assert np.array_equal(to_host(out), r.orc.dyadic("mul", a, b))


@pytest.mark.parametrize("name", ["n4096", "shoup12"])
def test_dyadic_ops_beyond_the_infinity_cache_take_the_non_temporal_path(rigs, name):
    """A launch that touches more than the 256 MiB Infinity Cache holds runs the non-temporal variant of the dyadic kernel (dpfhe_cabi.hip
    launch_dy): same words.  768 x L residue polynomials of 32 KiB per operand = 96 MiB (72 with three limbs) per stream."""
    r = rigs(name)
    L, n = r.p.n_limbs, r.p.n
    nb = 768 if L == 4 else 1100
    a, b, acc = (r.orc.fill(nb, seed).reshape(nb, L, n) for seed in (91, 92, 93))
    assert 3 * a.nbytes > (256 << 20) and 2 * a.nbytes <= (256 << 20)   # three streams cross the threshold, two (negate) do not
    A, Bd, ACC = r.dev(a), r.dev(b), r.dev(acc)
    assert np.array_equal(to_host(r.ev.dyadic_mul(A, Bd)), r.orc.dyadic("mul", a, b, threads=0))
    assert np.array_equal(to_host(r.ev.sub_words(A, Bd)), r.orc.dyadic("sub", a, b, threads=0))
    assert np.array_equal(to_host(r.ev.negate_words(A)), r.orc.dyadic("negate", a, threads=0))
    r.ev.dyadic_mul_add_(ACC, A, Bd)
    assert np.array_equal(to_host(ACC), r.orc.dyadic("mul_add", a, b, acc=acc, threads=0))
    pt = r.orc.fill(1, 94).reshape(L, n)
    big = np.concatenate([a, b], 0)   # one plaintext over a 2 x 96 MiB batch: two streams + the cached plaintext
    got = r.ev.multiply_plain(Ciphertext(r.dev(big.reshape(-1, 2, L, n)), True), Plaintext(r.dev(pt), True))
    assert np.array_equal(to_host(got.data).reshape(big.shape), r.orc.dyadic("mul", big, np.ascontiguousarray(np.broadcast_to(pt, big.shape)), threads=0))


# ---- ct x ct (the metric op) ---------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ALL)
def test_ct_mul_vs_oracle_all_domains(rigs, name):
    r = rigs(name)
    L, n = r.p.n_limbs, r.p.n
    for batch in (1, 3):
        a = r.orc.fill(batch * 2, 31).reshape(batch, 2, L, n)
        b = r.orc.fill(batch * 2, 32).reshape(batch, 2, L, n)
        qs = np.array(r.p.moduli, np.uint64)[None, :, None]
        a[0, :, :, : n // 4] = qs - np.uint64(1)  # worst-case magnitudes for the lazy arithmetic
        b[0, :, :, : n // 4] = qs - np.uint64(1)
        want = r.orc.ct_mul(np.ascontiguousarray(a), np.ascontiguousarray(b), threads=0)
        A, Bc = Ciphertext(r.dev(a)), Ciphertext(r.dev(b))
        c = r.ev.multiply(A, Bc)
        assert not c.is_ntt and np.array_equal(to_host(c.data), want)
        want_ntt = r.orc.ntt_fwd(want.reshape(-1, L, n), threads=0).reshape(want.shape)
        c2 = r.ev.multiply(A, Bc, out_ntt=True)
        assert c2.is_ntt and np.array_equal(to_host(c2.data), want_ntt)
        An = Ciphertext(r.ev.ntt_forward(A.data), True)
        Bn = Ciphertext(r.ev.ntt_forward(Bc.data), True)
        c3 = r.ev.multiply(An, Bn)
        assert c3.is_ntt and np.array_equal(to_host(c3.data), want_ntt)
        c4 = r.ev.multiply(An, Bn, out_ntt=False)
        assert np.array_equal(to_host(c4.data), want)
    if n <= 1024:  # and the oracle's NTT path itself equals schoolbook convolution here
        sb = r.orc.ct_mul(np.ascontiguousarray(a), np.ascontiguousarray(b), threads=0, schoolbook=True)
        assert np.array_equal(sb, want)


@pytest.mark.parametrize("name", ["n4096", "n8192", "fold12x3", "shoup12", "fold11"])
def test_every_form_of_the_fused_multiply_is_bit_exact_and_tunable(rigs, name):
    """include/dpfhe.h "A0, continued": quad / dual give the same words as the oracle; context creation measures nothing (default form, no
    probe times); dpfhe_ctx_autotune on caller scratch reports its measurements, leaves a usable choice and seeds the process-wide cache that
    a later context of the same shape starts from; contexts with one form say so."""
    r = rigs(name)
    L, n = r.p.n_limbs, r.p.n
    info = r.ctx.tune_info()
    if not (r.ctx.uses_fold and r.p.log2_n in (12, 13)):
        assert info["n_variants"] == 0 and info["probe_us"] == {}
        with pytest.raises(_cabi.DpfheError):
            r.ctx.set_ct_mul_variant("dual")
        return
    forms = ("quad", "dual")
    default = "quad" if r.p.log2_n == 12 else "dual"
    assert info["n_variants"] == len(forms) and info["chosen"] in forms
    # a context built before any explicit probe of its shape says "default"; one built after it (another test's rig, this test re-run) says "cached"
    assert (info["source"] == "default" and info["chosen"] == default and info["probe_us"] == {}) or info["source"].startswith("cached")
    batch = 29
    a = r.orc.fill(batch * 2, 33).reshape(batch, 2, L, n)
    b = r.orc.fill(batch * 2, 34).reshape(batch, 2, L, n)
    qs = np.array(r.p.moduli, np.uint64)[None, :, None]
    a[0, :, :, : n // 2] = qs - np.uint64(1)
    b[0, :, :, n // 2:] = qs - np.uint64(1)
    want = r.orc.ct_mul(np.ascontiguousarray(a), np.ascontiguousarray(b), threads=0)
    A, Bc = Ciphertext(r.dev(a)), Ciphertext(r.dev(b))
    try:
        for form in forms:
            r.ctx.set_ct_mul_variant(form)
            assert r.ctx.tune_info()["chosen"] == form and r.ctx.tune_info()["source"] == "forced"
            assert np.array_equal(to_host(r.ev.multiply(A, Bc).data), want), form
            assert np.array_equal(to_host(r.ev.multiply(A, A).data), r.orc.ct_mul(np.ascontiguousarray(a), np.ascontiguousarray(a), threads=0)), form + " (squaring)"
        with pytest.raises(_cabi.DpfheError):
            r.ctx.set_ct_mul_variant("octo")
        work = torch.empty(7 * 64 * L * n, dtype=torch.int64, device=r.ctx.device)
        info = r.ctx.autotune(work, reps=2)
        assert info["source"] == "dpfhe_ctx_autotune" and info["probe_pairs"] == 64 and info["probe_reps"] == 2 and len(info["probe_us"]) == len(forms)
        best = min(info["probe_us"], key=info["probe_us"].get)
        assert info["chosen"] in (best, default)          # the default stays unless another form is >= 3 % faster
        assert np.array_equal(to_host(r.ev.multiply(A, Bc).data), want)
        second = Context(r.p, r.ctx.device_id)            # same (device, log2 N, L): starts from the cached probe, launches nothing
        try:
            i2 = second.tune_info()
            assert i2["source"].startswith("cached") and i2["chosen"] == info["chosen"] and i2["probe_us"] == info["probe_us"]
        finally:
            second.close()
        with pytest.raises(_cabi.DpfheError):
            r.ctx.autotune(work[:100])
    finally:
        r.ctx.set_ct_mul_variant("quad" if r.p.log2_n == 12 else "dual")


def test_ct_mul_empty_batch_and_errors(rigs):
    r = rigs("n4096")
    lib = _cabi.load()
    assert lib.dpfhe_ct_mul(r.ctx.handle, None, None, None, 0, 0, None) == 0  # empty batch is a no-op
    assert lib.dpfhe_ntt_fwd(r.ctx.handle, None, 0, None) == 0
    t = r.ctx.empty(1, components=2)
    assert lib.dpfhe_ntt_fwd(r.ctx.handle, t.data_ptr() + 8, 1, None) == 2000  # misaligned
    assert lib.dpfhe_ct_mul(r.ctx.handle, t.data_ptr(), t.data_ptr(), t.data_ptr(), 1, 4, None) == 2000  # unknown flag
    big = r.ctx.empty(2, components=3)
    assert lib.dpfhe_ct_mul(r.ctx.handle, big.data_ptr(), big.data_ptr(), t.data_ptr(), 1, 0, None) == 2000  # output overlaps an operand
    assert "overlaps" in lib.dpfhe_last_error().decode()
    # squaring (a and b the same buffer) is fine
    sq = Ciphertext(torch.randint(0, 2**59, (1, 2, r.p.n_limbs, r.p.n), dtype=torch.int64, device=r.ctx.device))
    assert lib.dpfhe_ct_mul(r.ctx.handle, big.data_ptr(), sq.data.data_ptr(), sq.data.data_ptr(), 1, 0, None) == 0
    assert np.array_equal(to_host(big[:1]), r.orc.ct_mul(to_host(sq.data), to_host(sq.data), threads=0))
    with pytest.raises(_cabi.DpfheError) as e:
        r.ev.multiply(Ciphertext(t, True), Ciphertext(t, False))
    assert e.value.code == 2002
    with pytest.raises(_cabi.DpfheError):
        r.ev.ntt_forward_(torch.zeros(4, 4095, dtype=torch.int64, device=r.ctx.device))
    with pytest.raises(_cabi.DpfheError):
        Context(FheParams.__new__(FheParams), 0) if False else r.ev.multiply(Ciphertext(r.ctx.empty(components=3)), Ciphertext(r.ctx.empty(components=3)))


# ---- ct x pt matvec, reduce ------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,rows,cols", [("n4096", 5, 7), ("fold8", 3, 300), ("shoup10", 4, 130), ("n8192", 2, 3)])
def test_matvec_plain_vs_oracle(rigs, name, rows, cols):
    r = rigs(name)
    L, n = r.p.n_limbs, r.p.n
    W = r.orc.fill(rows * cols, 41).reshape(rows, cols, L, n)
    x = r.orc.fill(cols * 2, 42).reshape(cols, 2, L, n)
    qs = np.array(r.p.moduli, np.uint64)[:, None]
    W[0, :, :, :16] = qs - np.uint64(1); x[:, :, :, :16] = qs - np.uint64(1)  # max accumulation
    want = r.orc.matvec_plain(W.ravel(), x.ravel(), rows, cols, threads=0)
    y = r.ev.matvec_plain(Plaintext(r.dev(W), True), Ciphertext(r.dev(x), True))
    assert y.is_ntt and np.array_equal(to_host(y.data), want)


def test_matvec_plain_multi_rhs_vs_oracle(rigs):
    """dpfhe_matvec_plain_multi: [cols][n_rhs] right-hand sides against one oracle matvec per right-hand side; n_rhs = 7 exercises
    the 4 + 2 + 1 grouping, ragged row counts the tile tails.  Whole row tiles and whole 8-column periods (8 x 16, 4 x 8, 16 x 64 - the shape
    of a packed layer's products - and 12 x 24 with an odd right-hand-side count) take the branch-free form of the kernel (raw buffer loads, operands
    requested a column ahead); the 16 x 64 case carries q - 1 in every word of two rows and two columns: the lazy columns at their largest."""
    r = rigs("n4096")
    L, n = r.p.n_limbs, r.p.n
    qm1 = (np.array(r.p.moduli, np.uint64) - np.uint64(1))[:, None]
    for rows, cols, n_rhs in ((5, 3, 7), (8, 16, 4), (3, 2, 2), (2, 5, 1), (4, 8, 2), (16, 64, 8), (12, 24, 5)):
        W = r.orc.fill(rows * cols, 11).reshape(rows, cols, L, n)
        x = r.orc.fill(cols * n_rhs * 2, 12).reshape(cols, n_rhs, 2, L, n)
        if rows == 16:
            W[3, :], W[:, 5], x[:, 1], x[7] = qm1, qm1, qm1, qm1
        got = to_host(r.ev.matvec_plain_multi(Plaintext(r.dev(W), True), r.dev(x), n_rhs))
        for t in range(n_rhs):
            want = r.orc.matvec_plain(W.ravel(), np.ascontiguousarray(x[:, t]).ravel(), rows, cols, threads=0)
            assert np.array_equal(got[:, t], want), (rows, cols, n_rhs, t)


@pytest.mark.parametrize("name,rows,cols", [("n4096", 13, 9), ("fold8", 3, 300), ("shoup10", 9, 130), ("config1", 8, 5)])
def test_matvec_scalar_vs_oracle(rigs, name, rows, cols):
    r = rigs(name)
    L, n = r.p.n_limbs, r.p.n
    rng = np.random.default_rng(5)
    qs = np.array(r.p.moduli, np.uint64)
    w = (rng.integers(0, 2**62, (rows, cols, L), dtype=np.uint64) % qs[None, None, :]).astype(np.uint64)
    w[0, :, :] = qs - np.uint64(1)
    x = r.orc.fill(cols * 2, 43).reshape(cols, 2, L, n)
    x[:, :, :, :16] = (qs - np.uint64(1))[None, None, :, None]
    want = r.orc.matvec_scalar(w, x, rows, cols, threads=0)
    y = r.ev.matvec_scalar(r.dev(w), Ciphertext(r.dev(x), True))
    assert y.is_ntt and np.array_equal(to_host(y.data), want)


@pytest.mark.parametrize("name", ["n4096", "shoup10", "n8192"])
def test_multiply_plain_and_ct_add_sub_negate(rigs, name):
    r = rigs(name)
    L, n = r.p.n_limbs, r.p.n
    a = r.orc.fill(4, 51).reshape(2, 2, L, n)
    b = r.orc.fill(4, 52).reshape(2, 2, L, n)
    pt = r.orc.fill(1, 53).reshape(L, n)
    A, Bc = Ciphertext(r.dev(a), True), Ciphertext(r.dev(b), True)
    assert np.array_equal(to_host(r.ev.add(A, Bc).data), r.orc.dyadic("add", a, b))
    assert np.array_equal(to_host(r.ev.sub(A, Bc).data), r.orc.dyadic("sub", a, b))
    assert np.array_equal(to_host(r.ev.negate(A).data), r.orc.dyadic("negate", a))
    got = to_host(r.ev.multiply_plain(A, Plaintext(r.dev(pt), True)).data)
    want = r.orc.dyadic("mul", a, np.ascontiguousarray(np.broadcast_to(pt, a.shape)))
    assert np.array_equal(got, want)
    # ONE launch for the whole batch (dpfhe_multiply_plain: the plaintext broadcast inside the kernel), also in place
    io, ptd = r.dev(a), r.dev(pt)
    assert r.ev._lib.dpfhe_multiply_plain(r.ctx.handle, io.data_ptr(), io.data_ptr(), ptd.data_ptr(), 4, None) == 0
    assert np.array_equal(to_host(io), want)
    with pytest.raises(_cabi.DpfheError):
        r.ev.multiply_plain(A, Plaintext(r.dev(np.ascontiguousarray(np.broadcast_to(pt, a.shape))), True))   # per-item plaintexts: dyadic_mul's job


@pytest.mark.parametrize("name,count,comps", [("n4096", 37, 3), ("config1", 5, 2), ("shoup13", 1, 3)])
def test_reduce_sum_vs_oracle(rigs, name, count, comps):
    r = rigs(name)
    L, n = r.p.n_limbs, r.p.n
    x = r.orc.fill(count * comps, 61).reshape(count, comps, L, n)
    x[:, :, :, :4] = (np.array(r.p.moduli, np.uint64) - np.uint64(1))[None, None, :, None]
    got = to_host(r.ev.reduce_sum(Ciphertext(r.dev(x))).data)
    assert np.array_equal(got, r.orc.reduce_sum(x.ravel(), comps))


# ---- full BASELINE sizes: size-independent properties + EVERY output word against the oracle -------------------
def test_config2_full_size_roundtrip_linearity_and_whole_buffer_oracle(rigs):
    """configs[1]: batch = 1024 RNS polys x 4 limbs, N = 4096 (128 MiB)."""
    r = rigs("n4096")
    L, n, batch = 4, 4096, 1024
    g = torch.Generator(device="cpu").manual_seed(7)
    q = torch.tensor(r.p.moduli, dtype=torch.int64).view(1, L, 1)
    x = (torch.randint(0, 2**62, (batch, L, n), generator=g, dtype=torch.int64) % q).to(r.ctx.device)
    y = (torch.randint(0, 2**62, (batch, L, n), generator=g, dtype=torch.int64) % q).to(r.ctx.device)
    X = r.ev.ntt_forward(x)
    assert torch.equal(r.ev.ntt_inverse(X), x)                                   # NTT o INTT = id
    s = r.ev.add_words(x, y)
    assert torch.equal(r.ev.ntt_forward(s), r.ev.add_words(X, r.ev.ntt_forward(y)))  # linearity
    assert int(X.min()) >= 0 and bool((X < q.to(X.device)).all())              # canonical outputs
    want = r.orc.ntt_fwd(to_host(x), threads=0)                                # all 4096 residue polynomials, every word
    assert np.array_equal(to_host(X), want)
    assert np.array_equal(to_host(r.ev.ntt_inverse(y)), r.orc.ntt_inv(to_host(y), threads=0))   # inverse of non-image data, whole buffer


def test_n8192_large_batches_take_the_halves_form_and_match_the_oracle(rigs):
    """N = 8192, L = 6: from 2304 residue polynomials per launch the batched transforms run in "halves" form (launch.h kHalvesMinPolys: a register column stage
    + two 4096-point sub-transforms through one LDS buffer, 256-thread workgroups); below, the 512-thread kernels.  Both sides of the threshold, out of place
    and in place, EVERY word against the oracle; round trip; the two forms agree on a shared prefix of the batch."""
    r = rigs("n8192")
    L, n = r.p.n_limbs, r.p.n
    g = torch.Generator(device="cpu").manual_seed(29)
    q = torch.tensor(r.p.moduli, dtype=torch.int64).view(1, L, 1)
    big, small = 512, 383                                  # 3072 and 2298 residue polynomials: above and just below the threshold
    x = (torch.randint(0, 2**62, (big, L, n), generator=g, dtype=torch.int64) % q)
    x[0, :, : n // 2] = q.view(L, 1) - 1                   # worst-case residues in one polynomial
    xd = x.to(r.ctx.device)
    X = r.ev.ntt_forward(xd)
    want = r.orc.ntt_fwd(to_host(xd), threads=0)
    assert np.array_equal(to_host(X), want)
    assert torch.equal(r.ev.ntt_inverse(X), xd)                                           # round trip through the halves inverse
    assert np.array_equal(to_host(r.ev.ntt_inverse(xd)), r.orc.ntt_inv(to_host(xd), threads=0))   # inverse of non-image data
    Xs = r.ev.ntt_forward(xd[:small].contiguous())                                        # the 512-thread kernels on a prefix: the same words
    assert torch.equal(Xs, X[:small])
    y = xd.clone()
    r.ev.ntt_forward_(y)
    assert torch.equal(y, X)
    r.ev.ntt_inverse_(y)
    assert torch.equal(y, xd)


def test_n16384_large_batches_take_the_quarters_form_and_match_the_oracle(rigs):
    """N = 16384, 3 limbs of the pinned chain: from 768 residue polynomials per launch the batched transforms run in "quarters" form (launch.h kQuartersMinPolys:
    two register column stages + four 4096-point sub-transforms through one LDS buffer, 256-thread workgroups, two to a CU); below, the 1024-thread kernel.
    Both sides of the threshold, out of place and in place, EVERY word against the oracle; round trip; the two forms agree on a shared prefix of the batch."""
    r = rigs("fold14")
    L, n = r.p.n_limbs, r.p.n
    g = torch.Generator(device="cpu").manual_seed(31)
    q = torch.tensor(r.p.moduli, dtype=torch.int64).view(1, L, 1)
    big, small = 300, 255                                  # 900 and 765 residue polynomials: above and just below the threshold
    x = (torch.randint(0, 2**62, (big, L, n), generator=g, dtype=torch.int64) % q)
    x[0, :, : n // 2] = q.view(L, 1) - 1                   # worst-case residues in one polynomial
    x[1, :, 1::2] = q.view(L, 1) - 1
    xd = x.to(r.ctx.device)
    X = r.ev.ntt_forward(xd)
    want = r.orc.ntt_fwd(to_host(xd), threads=0)
    assert np.array_equal(to_host(X), want)
    assert torch.equal(r.ev.ntt_inverse(X), xd)                                           # round trip through the quarters inverse
    assert np.array_equal(to_host(r.ev.ntt_inverse(xd)), r.orc.ntt_inv(to_host(xd), threads=0))   # inverse of non-image data
    Xs = r.ev.ntt_forward(xd[:small].contiguous())                                        # the 1024-thread kernel on a prefix: the same words
    assert torch.equal(Xs, X[:small])
    assert torch.equal(r.ev.ntt_inverse(X[:small].contiguous()), xd[:small])
    y = xd.clone()
    r.ev.ntt_forward_(y)
    assert torch.equal(y, X)
    r.ev.ntt_inverse_(y)
    assert torch.equal(y, xd)


def test_ct_mul_large_batch_checksum_of_checksums(rigs):
    """2048 ct-muls at N=4096/L=4: sum of outputs == output of ... (bilinearity): sum_i a_i (x) b == (sum_i a_i) (x) b."""
    r = rigs("n4096")
    L, n, batch = 4, 4096, 2048
    g = torch.Generator(device="cpu").manual_seed(11)
    q = torch.tensor(r.p.moduli, dtype=torch.int64).view(1, 1, L, 1)
    a = (torch.randint(0, 2**62, (batch, 2, L, n), generator=g, dtype=torch.int64) % q).to(r.ctx.device)
    b1 = (torch.randint(0, 2**62, (1, 2, L, n), generator=g, dtype=torch.int64) % q).to(r.ctx.device)
    b = b1.expand(batch, 2, L, n).contiguous()
    c = r.ev.multiply(Ciphertext(a), Ciphertext(b))
    lhs = r.ev.reduce_sum(c)                                       # sum_i (a_i (x) b)
    asum = r.ev.reduce_sum(Ciphertext(a))                          # sum_i a_i
    rhs = r.ev.multiply(Ciphertext(asum.data.unsqueeze(0)), Ciphertext(b1))
    assert torch.equal(lhs.data, rhs.data[0])
    want = r.orc.ct_mul(to_host(a), to_host(b), threads=0)           # every pair, every word
    assert np.array_equal(to_host(c.data), want)


def test_generic_primes_full_size_whole_buffer_oracle():
    """The generic-prime arithmetic (Harvey/Shoup butterflies, 128-bit Barrett products) at the headline shape - N = 4096, four primes of 59 / 50 / 40 / 33
    bits, none of the 2^60 - d form (bench.py other_configs.shoup_n4096_l4): configs[1]'s 1024 RNS polynomials through both transforms and 2048
    ciphertext pairs through the fused multiply, EVERY output word against the oracle; round trip and canonical range on the device."""
    p = FheParams.generic_n4096_l4()
    r = Rig(p)
    try:
        assert not r.ctx.uses_fold and [q.bit_length() for q in p.moduli] == [59, 50, 40, 33]
        L, n = 4, 4096
        g = torch.Generator(device="cpu").manual_seed(23)
        q = torch.tensor(p.moduli, dtype=torch.int64)
        x = (torch.randint(0, 2**62, (1024, L, n), generator=g, dtype=torch.int64) % q.view(1, L, 1)).to(r.ctx.device)
        X = r.ev.ntt_forward(x)
        assert torch.equal(r.ev.ntt_inverse(X), x) and int(X.min()) >= 0 and bool((X < q.view(1, L, 1).to(X.device)).all())
        assert np.array_equal(to_host(X), r.orc.ntt_fwd(to_host(x), threads=0))
        assert np.array_equal(to_host(r.ev.ntt_inverse(x)), r.orc.ntt_inv(to_host(x), threads=0))
        batch = 2048
        a = (torch.randint(0, 2**62, (batch, 2, L, n), generator=g, dtype=torch.int64) % q.view(1, 1, L, 1)).to(r.ctx.device)
        b = (torch.randint(0, 2**62, (batch, 2, L, n), generator=g, dtype=torch.int64) % q.view(1, 1, L, 1)).to(r.ctx.device)
        c = r.ev.multiply(Ciphertext(a), Ciphertext(b))
        assert np.array_equal(to_host(c.data), r.orc.ct_mul(to_host(a), to_host(b), threads=0))
    finally:
        r.ctx.close()


def test_config4_per_gpu_shard_full_size_through_the_bench_step(rigs):
    """BASELINE configs[3], one GPU's share: 8192 ct-muls at N=4096 / L=4 through deeppowers_amd.sharding.ShardedMultiplyReduce
    - the very object bench.py times (multiply on the main stream, shard-local reduce -> all-gather -> final sum on the side
    stream, double-buffered).  Size-independent properties at FULL size + all 8192 products against the oracle:
      * canonical range of every output word;
      * bilinearity: sum_i a_i (x) b == (sum_i a_i) (x) b  (all 8192 pairs share b in the second step);
      * the pipelined total equals a serial recomputation, and the serial helper sharded_multiply_reduce agrees;
      * steps are independent: running two steps back to back (buffer reuse, overlap) changes nothing."""
    from deeppowers_amd.sharding import ShardedMultiplyReduce, sharded_multiply_reduce
    r = rigs("n4096")
    L, n, batch = 4, 4096, 8192
    dev = r.ctx.device
    g = torch.Generator(device=dev).manual_seed(41)
    q = torch.tensor(r.p.moduli, dtype=torch.int64, device=dev).view(1, 1, L, 1)
    a = Ciphertext(torch.randint(0, 2**62, (batch, 2, L, n), generator=g, dtype=torch.int64, device=dev) % q)
    b = Ciphertext(torch.randint(0, 2**62, (batch, 2, L, n), generator=g, dtype=torch.int64, device=dev) % q)
    b1 = b.data[:1].clone()
    bsame = Ciphertext(b1.expand(batch, 2, L, n).contiguous())
    pipe = ShardedMultiplyReduce(r.ev, batch)
    k0 = pipe.step(a, b)
    k1 = pipe.step(a, bsame)            # overlaps the reduce of step 0
    torch.cuda.synchronize()
    out0, tot0, out1, tot1 = pipe.outs[k0], pipe.totals[k0], pipe.outs[k1], pipe.totals[k1]
    for t in (out0, out1, tot0, tot1):
        assert int(t.min()) >= 0 and bool((t < q.view(1, L, 1) if t.dim() == 3 else t < q).all())    # canonical
    assert torch.equal(tot0, r.ev.reduce_sum(Ciphertext(out0)).data)                                     # pipelined == serial
    asum = r.ev.reduce_sum(a)
    rhs = r.ev.multiply(Ciphertext(asum.data.unsqueeze(0)), Ciphertext(b1))
    assert torch.equal(tot1, rhs.data[0])                                                                 # bilinearity at full size
    for lo in range(0, batch, 1024):                                                                      # ALL 8192 products vs the oracle, every word
        want = r.orc.ct_mul(to_host(a.data[lo:lo + 1024]), to_host(b.data[lo:lo + 1024]), threads=0)     # (1024 pairs at a time: 384 MiB of host memory per slice)
        assert np.array_equal(to_host(out0[lo:lo + 1024]), want), f"pairs {lo}..{lo + 1023}"
    local, total = sharded_multiply_reduce(r.ev, a, b)
    assert torch.equal(local.data, out0) and torch.equal(total.data, tot0)
    del pipe, local, total


def test_native_comm_world_size_one_through_the_step(rigs):
    """dpfhe_comm_* (RCCL behind the C ABI) as the transport of the same step, on a non-default stream - world size 1 because the
    test box has one GPU; the multi-rank form is examples/sharded_ct_mul.cpp and bench.py --native-comm."""
    from deeppowers_amd.sharding import NativeComm, ShardedMultiplyReduce
    r = rigs("n4096")
    L, n, batch = 4, 4096, 64
    ah, bh = r.orc.fill(batch * 2, 91).reshape(batch, 2, L, n), r.orc.fill(batch * 2, 92).reshape(batch, 2, L, n)
    a, b = Ciphertext(r.dev(ah)), Ciphertext(r.dev(bh))
    comm = NativeComm(0, 1, 0, NativeComm.new_unique_id())
    pipe = ShardedMultiplyReduce(r.ev, batch, comm=comm)
    k = pipe.step(a, b)
    torch.cuda.synchronize()
    want = r.orc.ct_mul(ah, bh, threads=0)
    assert np.array_equal(to_host(pipe.outs[k]), want)
    assert np.array_equal(to_host(pipe.totals[k]), r.orc.reduce_sum(want.ravel(), 3))
    g = comm.allgather(pipe.partials[k], stream=pipe.side)
    torch.cuda.synchronize()
    assert g.shape == (1, 3, L, n) and torch.equal(g[0], pipe.partials[k])
    # SURVEY.md 8(e)'s alternative exchange through the same communicator: ncclAllReduce(u64, sum) in place + one mod-q pass (dpfhe_comm_allreduce_sum).
    # World size 1: the sum of one partial is the partial; fed a LAZY word (3 q + r) the mod-q pass must bring it back to r.
    lazy = pipe.partials[k].clone()
    qcol = torch.tensor(r.p.moduli, dtype=torch.int64, device=lazy.device).view(1, L, 1)
    lazy += 3 * qcol
    comm.allreduce_sum(r.ctx, lazy, stream=pipe.side)
    torch.cuda.synchronize()
    assert torch.equal(lazy, pipe.partials[k])
    pipe2 = ShardedMultiplyReduce(r.ev, batch, comm=comm, collective="allreduce")   # (world 1: the step takes the plain path, same totals)
    k2 = pipe2.step(a, b)
    torch.cuda.synchronize()
    assert torch.equal(pipe2.totals[k2], pipe.totals[k])
    comm.close()


def test_evaluator_temporaries_follow_the_stream(rigs):
    """default outputs and temporaries are allocated on the stream the kernels run on.  Run key switching /
    multiply_plain on a side stream while the main stream churns the caching allocator with same-sized blocks: results must
    equal the default-stream run."""
    r = rigs("n4096")
    L, n = 4, 4096
    ch = r.orc.fill(8 * 3, 55).reshape(8, 3, L, n)
    evkh = r.orc.fill(L * 2, 56).reshape(L, 2, L, n)
    c3, evk = Ciphertext(r.dev(ch)), r.dev(evkh)
    want = r.ev.relinearize(c3, evk)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    outs = []
    for _ in range(8):
        outs.append(r.ev.relinearize(c3, evk, stream=side))
        junk = [torch.full_like(want.data, 7) for _ in range(4)]    # same size class, main stream
        del junk
    torch.cuda.synchronize()
    assert all(torch.equal(o.data, want.data) for o in outs)


def test_streams_and_concurrent_contexts(rigs):
    r = rigs("n4096")
    x = r.orc.fill(8, 71)
    want = r.orc.ntt_fwd(x, threads=0)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    d1, d2 = r.dev(x), r.dev(x)
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        r.ev.ntt_forward_(d1)
    r.ev.ntt_forward_(d2, stream=s2)
    torch.cuda.synchronize()
    assert np.array_equal(to_host(d1), want) and np.array_equal(to_host(d2), want)


def test_composed_large_ring_operations_on_two_streams_keep_their_scratch_apart(rigs):
    """N = 16384: the composed multiply and key switch take their scratch from an arena the context keeps PER STREAM (round 5; a stream-ordered pool before).
    Two streams issue them interleaved, several times, with different batch sizes (the second call on a stream makes its arena grow): every result equals
    the oracle's - a shared arena, or a growth that frees a block another stream still uses, would corrupt one of them."""
    r = rigs("fold14")
    L, n = r.p.n_limbs, r.p.n
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    cases = []
    for i, batch in enumerate((2, 5, 3, 7)):
        a = r.orc.fill(batch * 2, 300 + i).reshape(batch, 2, L, n)
        b = r.orc.fill(batch * 2, 400 + i).reshape(batch, 2, L, n)
        cases.append((a, b, r.orc.ct_mul(np.ascontiguousarray(a), np.ascontiguousarray(b), threads=0)))
    evk = r.orc.fill(L * 2, 7).reshape(L, 2, L, n)
    devk = r.dev(evk)
    dev = [(Ciphertext(r.dev(a)), Ciphertext(r.dev(b))) for a, b, _ in cases]
    torch.cuda.synchronize()
    outs = []
    for rep in range(2):
        for i, (A, Bc) in enumerate(dev):
            st = s1 if i % 2 == 0 else s2
            c = r.ev.multiply(A, Bc, stream=st)
            outs.append((i, c, r.ev.relinearize(c, devk, stream=st)))
    torch.cuda.synchronize()
    for i, c, rl in outs:
        want = cases[i][2]
        assert np.array_equal(to_host(c.data), want), i
        assert np.array_equal(to_host(rl.data), r.orc.relinearize(want, evk, threads=0)), i


def test_rccl_allgather_world_size_one(rigs):
    import ctypes as C
    r = rigs("n4096")
    lib = _cabi.load()
    uid = (C.c_uint8 * 128)()
    _cabi.check(lib.dpfhe_comm_unique_id(uid), "unique_id")
    comm = C.c_void_p()
    _cabi.check(lib.dpfhe_comm_create(C.byref(comm), uid, 0, 1, 0), "comm_create")
    send = r.dev(r.orc.fill(3, 81))
    recv = torch.zeros_like(send)
    _cabi.check(lib.dpfhe_comm_allgather(comm, recv.data_ptr(), send.data_ptr(), send.numel(), torch.cuda.current_stream().cuda_stream), "allgather")
    torch.cuda.synchronize()
    assert torch.equal(send, recv)
    lib.dpfhe_comm_destroy(comm)


def test_bench_distributed_path_world_size_one():
    """bench.py launched the way the driver launches N>1 (torch.distributed.run, RCCL process group, barrier,
    all-gather on the side stream) - at world size 1 because the test box has one GPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DPFHE_FORCE_DIST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--batch-per-gpu", "64", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["reduce_consistent"] is True and d["value"] > 0 and d["unit"] == "ct-mul/s"
    assert set(("roofline", "config", "metric", "ms_per_step", "scaling", "dtype", "data", "allgather_us")) <= set(d)
    assert d["roofline"]["bound"] == "hbm" and "valu" in d["roofline"]["limited_by"] and d["roofline"]["frac_hbm"] == d["roofline"]["frac"] and d["allgather_us"]["median"] > 0
    # what the driver's record keeps: the NTT verdict inside `roofline`, the form of the multiply and its measurements inside `config`
    nv, at = d["roofline"]["ntt"], d["config"]["autotune"]
    assert 0 < nv["fwd_frac"] < 1 and 0 < nv["inv_frac"] < 1 and nv["round_trip_exact"] is True and "sustained_2s" in nv
    lt = d["roofline"]["traffic_live"]   # measured by rocprofv3 in this very run (or an error string when the box has no profiler): never silently absent
    assert lt is not None and ("error" in lt or 0.99 < lt["hbm_bytes_per_ct_mul"] / lt["algorithmic_bytes_per_ct_mul"] < 1.02), lt
    assert at["chosen"] in ("quad", "dual") and set(at["step_probe_ms"]) == {"quad", "dual"} and at["at_ctx_create"]["source"] == "default" and at["at_ctx_create"]["probe_us"] == {}
    assert at["chosen"] in d["roofline"]["kernel"] and "regime" in d["roofline"] and len(line) < 12000
    rf = d["roofline"]   # the metric's second half as SCALARS (the driver's record keeps scalars of `roofline` only)
    assert rf["autotune_chosen"] == at["chosen"] and rf["ntt_fwd_frac"] == nv["fwd_frac"] and rf["ntt_inv_frac"] == nv["inv_frac"] and rf["ntt_fwd_us"] > 0 and 0 < rf["copy_frac"] < 1
    # the same launch with the library's own communicator as the transport
    out = subprocess.run(cmd[:-1] + ["--no-cpu-baseline", "--native-comm", "--no-live-traffic"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["reduce_consistent"] is True and "dpfhe_comm_allgather" in d["config"]["collective"]


def test_config3_full_size_matvec_linearity_and_whole_buffer_oracle(rigs):
    """BASELINE configs[2]: hidden = 768 rows, 64 input ciphertexts, N=4096, L=4 (6 GiB of plaintext weights)."""
    r = rigs("n4096")
    L, n, rows, cols = 4, 4096, 768, 64
    g = torch.Generator(device="cpu").manual_seed(21)
    q = torch.tensor(r.p.moduli, dtype=torch.int64, device=r.ctx.device).view(1, 1, L, 1)
    dg = torch.Generator(device=r.ctx.device).manual_seed(22)
    W = torch.randint(0, 2**62, (rows, cols, L, n), generator=dg, dtype=torch.int64, device=r.ctx.device) % q
    x1 = torch.randint(0, 2**62, (cols, 2, L, n), generator=dg, dtype=torch.int64, device=r.ctx.device) % q
    x2 = torch.randint(0, 2**62, (cols, 2, L, n), generator=dg, dtype=torch.int64, device=r.ctx.device) % q
    Wp = Plaintext(W, True)
    y1 = r.ev.matvec_plain(Wp, Ciphertext(x1, True))
    y2 = r.ev.matvec_plain(Wp, Ciphertext(x2, True))
    y12 = r.ev.matvec_plain(Wp, Ciphertext(r.ev.add_words(x1, x2), True))
    assert torch.equal(y12.data, r.ev.add_words(y1.data, y2.data))                    # linearity in x at full size
    x1h = to_host(x1).ravel()
    for lo in range(0, rows, 96):                                                     # ALL 768 rows vs the oracle (96 rows = 768 MiB of W per slice)
        want = r.orc.matvec_plain(to_host(W[lo:lo + 96]).ravel(), x1h, 96, cols, threads=0)
        assert np.array_equal(to_host(y1.data[lo:lo + 96]), want.reshape(96, 2, L, n)), f"rows {lo}..{lo + 95}"
    # scalar-weight variant on the same shape: equals the polynomial variant with constant polynomials' NTT = constants
    w = torch.randint(0, 2**62, (rows, cols, L), generator=dg, dtype=torch.int64, device=r.ctx.device) % q.view(1, 1, L)
    ys = r.ev.matvec_scalar(w, Ciphertext(x1, True))
    want_s = r.orc.matvec_scalar(to_host(w), to_host(x1), rows, cols, threads=0)     # scalar weights: all rows
    assert np.array_equal(to_host(ys.data), np.asarray(want_s).reshape(rows, 2, L, n))
    del W


@pytest.mark.parametrize("name", ["fold14", "shoup14", "fold15", "fold16"])
def test_large_rings_multiply_and_key_switch_composed_behind_the_c_abi(rigs, name):
    """N > 8192 has no fused kernels: dpfhe_ct_mul / dpfhe_relinearize / dpfhe_switch_key compose the batched transforms with one-pass
    streaming kernels (kernels_large.h; scratch from the stream-ordered allocator) and match the oracle in every domain combination,
    also for a squaring (one transformed copy) and on a side stream."""
    r = rigs(name)
    L, n = r.p.n_limbs, r.p.n
    ah, bh = r.orc.fill(6, 1).reshape(3, 2, L, n), r.orc.fill(6, 2).reshape(3, 2, L, n)
    ah[0, :, :, :8] = np.array(r.p.moduli, np.uint64)[None, :, None] - np.uint64(1)   # worst-case residues
    a, b = Ciphertext(r.dev(ah)), Ciphertext(r.dev(bh))
    want = r.orc.ct_mul(ah, bh)
    assert np.array_equal(to_host(r.ev.multiply(a, b).data), want)
    an, bn = Ciphertext(r.ev.ntt_forward(a.data), True), Ciphertext(r.ev.ntt_forward(b.data), True)
    assert np.array_equal(to_host(r.ev.multiply(an, bn, out_ntt=False).data), want)
    assert np.array_equal(to_host(r.ev.ntt_inverse(r.ev.multiply(an, bn).data)), want)
    assert np.array_equal(to_host(r.ev.ntt_inverse(r.ev.multiply(a, b, out_ntt=True).data)), want)
    assert np.array_equal(to_host(r.ev.multiply(a, a).data), r.orc.ct_mul(ah, ah))
    side = torch.cuda.Stream(device=r.ctx.device)
    side.wait_stream(torch.cuda.current_stream(r.ctx.device))
    got = r.ev.multiply(a, b, stream=side)
    side.synchronize()
    assert np.array_equal(to_host(got.data), want)
    # relinearisation and the plain key switch with RNS-digit keys (any words < q serve as a key for a bit-exactness check)
    evk = r.orc.fill(L * 2, 7).reshape(L, 2, L, n)
    got = r.ev.relinearize(Ciphertext(r.dev(want)), r.dev(evk))
    assert np.array_equal(to_host(got.data), r.orc.relinearize(want, evk, threads=0))
    got = r.ev.apply_galois(Ciphertext(r.dev(ah)), 5, r.dev(evk))   # automorphism + dpfhe_switch_key
    assert np.array_equal(to_host(got.data), r.orc.switch_key(r.orc.apply_galois(ah, 5), evk, threads=0))
    out = r.ctx.empty(3, components=2)
    assert r.ev._lib.dpfhe_relinearize(r.ctx.handle, out.data_ptr(), out.data_ptr(), r.dev(evk).data_ptr(), 1, None) == 2000   # output over its input
    # hybrid key switching with this context read as the EXTENDED one (last limb = special prime): data on the first L - 1 limbs
    Ld = L - 1
    hkey = r.orc.fill(Ld * 2, 8).reshape(Ld, 2, L, n)
    for comps in (3, 2):
        cth = np.ascontiguousarray(r.orc.fill(3 * comps, 9 + comps).reshape(3, comps, L, n)[:, :, :Ld])
        got = r.ev.keyswitch_hybrid(Ciphertext(r.dev(cth)), r.dev(hkey))
        assert np.array_equal(to_host(got.data), r.orc.keyswitch_hybrid(cth, hkey, comps, threads=0))


def test_large_ring_batches_are_sliced_to_bound_the_scratch():
    """The composed operations above N = 8192 take their scratch in slices (1 GiB by default): with dpfhe_ctx_set_scratch_limit(2 MiB) a batch of 5
    items at N = 16384 runs as 5 (multiply: 1.5 MiB of scratch per item) and 3 (key switch: 1.1 MiB) slices - same words as the oracle.
    Run in a child so that the context's memory pool goes with the process."""
    code = r"""
import numpy as np, torch
from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host
from deeppowers_amd.params import PRIMES_60, FheParams
from oracle import pyoracle as po
from oracle.cbind import Oracle
qs = [PRIMES_60[i][0] for i in (1, 2, 4)]
p = FheParams(14, tuple(qs), tuple(po.min_primitive_2n_root(16384, q) for q in qs))
ctx = Context(p, 0); ctx.set_scratch_limit(2); ev = Evaluator(ctx); orc = Oracle.from_params(p)
L, n = p.n_limbs, p.n
ah, bh = orc.fill(10, 1).reshape(5, 2, L, n), orc.fill(10, 2).reshape(5, 2, L, n)
want = orc.ct_mul(ah, bh, threads=0)
assert np.array_equal(to_host(ev.multiply(Ciphertext(to_device(ah, ctx.device)), Ciphertext(to_device(bh, ctx.device))).data), want)
evk = orc.fill(L * 2, 7).reshape(L, 2, L, n)
got = ev.relinearize(Ciphertext(to_device(want, ctx.device)), to_device(evk, ctx.device))
assert np.array_equal(to_host(got.data), orc.relinearize(want, evk, threads=0))
print("SLICED-OK")
"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert run.returncode == 0 and "SLICED-OK" in run.stdout, run.stdout + run.stderr


def test_sum_in_ntt_domain_gives_the_same_total():
    """ShardedMultiplyReduce(sum_in_ntt_domain=True): products left in the NTT domain (dpfhe_ct_mul, DPFHE_OUT_NTT), summed there, ONE inverse transform of
    the total - the inverse transform is linear and every word canonical, so the total equals the coefficient-domain pipeline's and the oracle's word for word."""
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host
    from deeppowers_amd.sharding import ShardedMultiplyReduce
    p = FheParams.n4096_l4()
    orc = Oracle.from_params(p)
    batch = 37
    a = orc.fill(batch * 2, 311).reshape(batch, 2, p.n_limbs, p.n)
    b = orc.fill(batch * 2, 312).reshape(batch, 2, p.n_limbs, p.n)
    a[0, :, :, : p.n // 2] = np.array(p.moduli, np.uint64)[None, :, None] - np.uint64(1)
    ctx = Context(p, 0)
    ev = Evaluator(ctx)
    try:
        da, db = Ciphertext(to_device(a, ctx.device)), Ciphertext(to_device(b, ctx.device))
        plain = ShardedMultiplyReduce(ev, batch)
        k0 = plain.step(da, db)
        lazy = ShardedMultiplyReduce(ev, batch, sum_in_ntt_domain=True)
        k1 = k1b = lazy.step(da, db)
        k1b = lazy.step(da, db)          # second buffer as well
        torch.cuda.synchronize()
        want = orc.reduce_sum(orc.ct_mul(a, b, threads=0).ravel(), 3)
        got0, got1, got1b = (to_host(t).reshape(want.shape) for t in (plain.totals[k0], lazy.totals[k1], lazy.totals[k1b]))
        assert np.array_equal(got0, want) and np.array_equal(got1, want) and np.array_equal(got1b, want)
    finally:
        ctx.close()
