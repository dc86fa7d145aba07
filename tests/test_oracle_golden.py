"""CPU tests: the oracle (Python big-int + C restatement) against known answers.

The reference holds no vectors for this path, so the pins are:
  * the literal known answers of SURVEY.md Appendix B (computed independently by the survey),
  * tests/golden/*.json (pure-Python big-int schoolbook / O(N^2) evaluation, make_golden.py),
  * first-principles identities (X^N = -1, NTT(delta_0) = 1, round trips, linearity).
"""
import hashlib
import json
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from deeppowers_amd.params import PRIMES_60, PRIME_30, PSI_30_N1024, FheParams
from oracle import pyoracle as po
from oracle.cbind import Oracle


def load(golden_dir, name):
    with open(os.path.join(golden_dir, name + ".json")) as f:
        return json.load(f)


def u64(x):
    return np.ascontiguousarray(np.array(x, dtype=np.uint64))


def sha(arr):
    return hashlib.sha256(np.ascontiguousarray(arr, dtype="<u8").tobytes()).hexdigest()


# ---- SURVEY.md Appendix B literals ------------------------------------------------------------
def test_appendix_b_hand_checkable():
    assert po.negacyclic_schoolbook(list(range(1, 9)), list(range(8, 0, -1)), 17) == [10, 9, 12, 0, 5, 8, 7, 0]
    assert po.negacyclic_schoolbook([1, 1] + [0] * 6, [0] * 7 + [1], 17) == [16, 0, 0, 0, 0, 0, 0, 1]


def test_appendix_b_splitmix_stream():
    g = po.SplitMix64(1)
    assert g.words_mod(4, PRIME_30) == [895327288, 351381262, 898803898, 651882458]


APPENDIX_B_SHA = {
    "c0": "dfca37d9297b2e1aa0da41ed3922e8d24a0093a224ec78c8c45af12d8f8ebdbe",
    "c1": "409a1fda06de0a6c7e74bbf8a68e03981ff01e80629ef0dfed57a875a560f87b",
    "c2": "fdd1e1cfe69e681376f8578e67cc9992a7836a389fe6f1eecf220a910cd27599",
    "c0c1c2": "9e97bb5cf219416d6cb3683b99de66ee8aaec17496504a97bf0471fdf43ab66b",
}


def test_config1_fixture_matches_appendix_b(golden_dir):
    d = load(golden_dir, "config1_ct_mul")
    assert d["sha256"] == APPENDIX_B_SHA
    assert d["c0_head"] == [54834692, 176261304, 149787807, 840201699]
    assert d["c1_head"] == [170487726, 929643531, 91750252, 616303861]
    assert d["c2_head"] == [280408825, 925878884, 456576842, 698931630]
    assert sha(u64(d["c0"] + d["c1"] + d["c2"])) == APPENDIX_B_SHA["c0c1c2"]


def test_c_oracle_config1_ct_mul_both_paths(golden_dir):
    """C schoolbook AND C Harvey-NTT path reproduce the Appendix-B digest bit for bit."""
    d = load(golden_dir, "config1_ct_mul")
    orc = Oracle.from_params(FheParams.config1())
    ab = orc.fill(4, 1).reshape(4, 1024)  # a0, a1, b0, b1 drawn in this order
    assert ab[0, :4].tolist() == d["a0_head"]
    a2, b2 = np.ascontiguousarray(ab[:2]), np.ascontiguousarray(ab[2:])
    for schoolbook in (True, False):
        c = orc.ct_mul(a2, b2, schoolbook=schoolbook)
        assert sha(c) == APPENDIX_B_SHA["c0c1c2"], f"schoolbook={schoolbook}"
        assert c[0, 0, 0].tolist() == d["c0"] and c[0, 1, 0].tolist() == d["c1"] and c[0, 2, 0].tolist() == d["c2"]


# ---- golden fixtures ------------------------------------------------------------------------------
def test_small_ntt_vectors(golden_dir):
    for v in load(golden_dir, "small_ntt"):
        orc = Oracle(v["log2n"], [v["q"]], [v["psi"]])
        a, want = u64(v["a"]), u64(v["ntt"])
        got = orc.ntt_fwd(a)
        assert np.array_equal(got, want), (v["log2n"], v["q"])
        assert np.array_equal(orc.ntt_inv(got), a)
        assert po.ntt_forward(v["a"], v["q"], v["psi"]) == v["ntt"]
        assert po.ntt_forward_definition(v["a"], v["q"], v["psi"]) == v["ntt"]


def test_rns_ct_mul_small(golden_dir):
    v = load(golden_dir, "rns_ct_mul_small")
    orc = Oracle(v["log2n"], v["moduli"], v["psi"])
    a, b, c = u64(v["a"]), u64(v["b"]), u64(v["c"])
    assert np.array_equal(orc.ct_mul(a, b).ravel(), c)
    assert np.array_equal(orc.ct_mul(a, b, schoolbook=True).ravel(), c)
    assert np.array_equal(orc.ct_mul(a, b, threads=0).ravel(), c)  # all threads


def test_sympy_restatement_agrees_with_both_oracles(golden_dir):
    """tests/golden/sympy_restatement.json was produced by sympy ALONE (Poly products over GF(q) reduced by X^N + 1; Poly.eval and
    sympy.discrete.transforms.ntt for the transform - make_golden.py sympy_restatement): a third, builder-independent route to the same words.
    The Python big-int oracle and the C oracle (both of its multiply paths, single- and multi-threaded) must reproduce it word for word.
    (It cannot turn "parity unpinned" into "pinned" - the reference holds no vector - but the oracle no longer rests on one author's arithmetic.)"""
    v = load(golden_dir, "sympy_restatement")
    cm = v["ct_mul_n256"]
    n, moduli, L = 1 << cm["log2n"], cm["moduli"], len(cm["moduli"])
    orc = Oracle(cm["log2n"], moduli, cm["psi"])
    a, b, c = u64(cm["a"]), u64(cm["b"]), u64(cm["c"])
    assert np.array_equal(orc.ct_mul(a, b).ravel(), c)
    assert np.array_equal(orc.ct_mul(a, b, schoolbook=True).ravel(), c)
    assert np.array_equal(orc.ct_mul(a, b, threads=0).ravel(), c)
    A, B, C3 = a.reshape(cm["batch"], 2, L, n), b.reshape(cm["batch"], 2, L, n), c.reshape(cm["batch"], 3, L, n)
    for i in range(cm["batch"]):   # the pure-Python big-int oracle, both of ITS routes
        ai = [[[int(w) for w in A[i, comp, l]] for l in range(L)] for comp in range(2)]
        bi = [[[int(w) for w in B[i, comp, l]] for l in range(L)] for comp in range(2)]
        want = [[[int(w) for w in C3[i, comp, l]] for l in range(L)] for comp in range(3)]
        assert po.ct_mul_schoolbook(ai, bi, moduli) == want
        assert po.ct_mul_ntt(ai, bi, moduli, cm["psi"]) == want
    for t in v["ntt_n64"]:
        o1 = Oracle(t["log2n"], [t["q"]], [t["psi"]])
        assert np.array_equal(o1.ntt_fwd(u64(t["a"])), u64(t["ntt"]))
        assert np.array_equal(o1.ntt_inv(u64(t["ntt"])), u64(t["a"]))
        assert po.ntt_forward(t["a"], t["q"], t["psi"]) == t["ntt"] and po.ntt_forward_definition(t["a"], t["q"], t["psi"]) == t["ntt"]


def test_n4096_ntt_digests(golden_dir):
    orc = Oracle.from_params(FheParams.n4096_l4())
    for v in load(golden_dir, "n4096_ntt_digest"):
        one = Oracle(12, [v["q"]], [v["psi"]])
        a = u64(po.SplitMix64(v["seed"]).words_mod(4096, v["q"]))
        assert a[:4].tolist() == v["a_head"]
        ah = one.ntt_fwd(a)
        assert ah[:4].tolist() == v["ntt_head"] and sha(ah) == v["ntt_sha256"]
    # and the 4-limb context applies limb l's tables to slot l
    x = orc.fill(1, 7)
    y = orc.ntt_fwd(x)
    for l in range(4):
        one = Oracle(12, [orc.moduli[l]], [orc.psi[l]])
        assert np.array_equal(one.ntt_fwd(x[0, l]), y[0, l])


def test_identities(golden_dir):
    d = load(golden_dir, "identities")
    assert d["n8_q17_ab"] == [10, 9, 12, 0, 5, 8, 7, 0] and d["n8_q17_1pX_times_X7"] == [16, 0, 0, 0, 0, 0, 0, 1]
    q, psi, n = PRIMES_60[0][0], PRIMES_60[0][1], 4096
    one = Oracle(12, [q], [psi])
    delta = np.zeros(n, np.uint64); delta[0] = 1
    assert np.all(one.ntt_fwd(delta) == 1)  # NTT(1) = all-ones
    X = np.zeros(n, np.uint64); X[1] = 1
    nx = one.ntt_fwd(X)  # NTT(X)[k] = psi^(2 brv(k) + 1)
    assert nx[:8].tolist() == d["ntt_X_n4096_q0_head"] and sha(nx) == d["ntt_X_n4096_q0_sha256"]
    for k in (0, 1, 2, 77, 4095):
        assert int(nx[k]) == pow(psi, 2 * po.bit_reverse(k, 12) + 1, q)
    rp, irp = one.root_powers(0)
    assert [int(v) for v in rp[:64]] == po.root_powers_bitrev(n, q, psi)[:64]
    assert all(int(r) * int(i) % q == 1 for r, i in zip(rp[:64], irp[:64]))


# ---- properties (C oracle) ------------------------------------------------------------------------
@pytest.mark.parametrize("params", [FheParams.config1(), FheParams.n4096_l4(), FheParams.n8192_l6()], ids=["n1024", "n4096l4", "n8192l6"])
def test_roundtrip_and_convolution_theorem(params):
    orc = Oracle.from_params(params)
    x = orc.fill(3, 11)
    xh = orc.ntt_fwd(x, threads=0)
    assert np.array_equal(orc.ntt_inv(xh, threads=0), x)
    assert all(int(xh[..., l, :].max()) < q for l, q in enumerate(params.moduli))
    if params.n <= 1024:  # NTT-path product == schoolbook product
        a, b = x[0], x[1]
        prod = orc.ntt_inv(orc.dyadic("mul", xh[0], xh[1]))
        for l, q in enumerate(params.moduli):
            assert np.array_equal(prod[l], orc.schoolbook(np.ascontiguousarray(a[l]), np.ascontiguousarray(b[l]), q))


def test_dyadic_ops_against_bigint():
    p = FheParams.n4096_l4()
    orc = Oracle.from_params(p)
    x = orc.fill(3, 5)
    a, b, acc = x[0], x[1], x[2]
    # force edge residues
    a[:, 0], b[:, 0] = 0, 0
    for l, q in enumerate(p.moduli):
        a[l, 1], b[l, 1], a[l, 2], b[l, 2], acc[l, 1] = q - 1, q - 1, q - 1, 1, q - 1
    got = {op: orc.dyadic(op, a, b) for op in ("mul", "add", "sub")}
    got["negate"] = orc.dyadic("negate", a)
    got["mul_add"] = orc.dyadic("mul_add", a, b, acc=acc)
    for l, q in enumerate(p.moduli):
        A, B, C = [int(v) for v in a[l]], [int(v) for v in b[l]], [int(v) for v in acc[l]]
        assert got["mul"][l].tolist() == po.dyadic_mul(A, B, q)
        assert got["add"][l].tolist() == po.poly_add(A, B, q)
        assert got["sub"][l].tolist() == po.poly_sub(A, B, q)
        assert got["negate"][l].tolist() == po.poly_negate(A, q)
        assert got["mul_add"][l].tolist() == po.dyadic_mul_add(C, A, B, q)


@settings(max_examples=25, deadline=None)
@given(st.integers(3, 7), st.integers(0, 2**63), st.sampled_from([0, 1, 2, 3, 4, 5]))
def test_property_schoolbook_vs_ntt_path(log2n, seed, limb):
    n = 1 << log2n
    q = PRIMES_60[limb][0]
    psi = pow(PRIMES_60[limb][2], 8192 // n, q)
    orc = Oracle(log2n, [q], [psi])
    g = po.SplitMix64(seed)
    a, b = g.words_mod(n, q), g.words_mod(n, q)
    want = po.negacyclic_schoolbook(a, b, q)
    A, B = orc.ntt_fwd(u64(a)), orc.ntt_fwd(u64(b))
    assert orc.ntt_inv(orc.dyadic("mul", A, B)).tolist() == want
    assert orc.schoolbook(u64(a), u64(b), q).tolist() == want
    assert A.tolist() == po.ntt_forward(a, q, psi)


def test_linearity_and_xn_minus_one():
    p = FheParams.n4096_l4()
    orc = Oracle.from_params(p)
    x = orc.fill(2, 3)
    s = orc.dyadic("add", x[0], x[1])
    assert np.array_equal(orc.ntt_fwd(s), orc.dyadic("add", orc.ntt_fwd(x[0]), orc.ntt_fwd(x[1])))
    # multiplying by X^(N-1) then by X gives -a  (X^N = -1)
    n = p.n
    xm = np.zeros((p.n_limbs, n), np.uint64); xm[:, n - 1] = 1
    x1 = np.zeros((p.n_limbs, n), np.uint64); x1[:, 1] = 1
    t = orc.dyadic("mul", orc.dyadic("mul", orc.ntt_fwd(x[0]), orc.ntt_fwd(xm)), orc.ntt_fwd(x1))
    assert np.array_equal(orc.ntt_inv(t), orc.dyadic("negate", x[0]))


def test_matvec_and_reduce_small():
    log2n, n = 4, 16
    moduli = [PRIMES_60[0][0], PRIMES_60[5][0]]
    psis = [pow(PRIMES_60[0][2], 8192 // n, moduli[0]), pow(PRIMES_60[5][2], 8192 // n, moduli[1])]
    orc = Oracle(log2n, moduli, psis)
    rows, cols = 3, 5
    W = orc.fill(rows * cols, 1)           # [rows*cols][L][N]
    x = orc.fill(cols * 2, 2)              # [cols][2][L][N]
    y = orc.matvec_plain(W.ravel(), x.ravel(), rows, cols)
    Wl = [[[W[i * cols + j, l].tolist() for l in range(2)] for j in range(cols)] for i in range(rows)]
    xl = [[[x[j * 2 + c, l].tolist() for l in range(2)] for c in range(2)] for j in range(cols)]
    want = po.matvec_plain(Wl, xl, moduli)
    for i in range(rows):
        for c in range(2):
            for l in range(2):
                assert y[i, c, l].tolist() == want[i][c][l]
    red = orc.reduce_sum(x.ravel(), comps=2)
    for c in range(2):
        for l, q in enumerate(moduli):
            assert red[c, l].tolist() == [sum(int(x[j * 2 + c, l, k]) for j in range(cols)) % q for k in range(n)]
