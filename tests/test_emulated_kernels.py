"""CPU: runs the per-thread code of the HIP NTT kernels (deeppowers_amd/csrc/ntt_core.h) in a
thread-by-thread emulator (tools/emulate.cpp) and checks it bit for bit against the oracle.
Covers every geometry the C-ABI dispatches, both arithmetic policies, both directions, extreme
inputs, and counts 64-bit wrap-arounds of the lazy (uncorrected) arithmetic - must be zero."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from deeppowers_amd.params import PRIMES_60, PRIME_30, PSI_30_N1024, ntt_primes
from oracle import pyoracle as po
from oracle.cbind import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
U = C.POINTER(C.c_uint64)


@pytest.fixture(scope="module")
def emu():
    so = os.path.join(ROOT, "tools", "libemu.so")
    src = os.path.join(ROOT, "tools", "emulate.cpp")
    deps = [src] + [os.path.join(ROOT, "deeppowers_amd", "csrc", f) for f in ("ntt_core.h", "ntt_top.h", "ntt_halves.h", "ntt_quarters.h", "modarith.h", "tables.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, src])   # (F64Arith: nothing fused behind the explicit fma calls)
    lib = C.CDLL(so)
    lib.emu_ntt.argtypes = [C.c_int] * 4 + [C.c_uint64, C.c_uint64, U, U]
    lib.emu_overflows.restype = C.c_long
    return lib


def run(emu, arith, ln, le, inv, q, psi, a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.zeros_like(a)
    rc = emu.emu_ntt(arith, ln, le, inv, q, psi, a.ctypes.data_as(U), out.ctypes.data_as(U))
    return rc, out


GEOS = [(8, 4), (10, 4), (11, 4), (12, 4), (13, 4), (13, 5), (14, 4), (14, 5)]


@pytest.mark.parametrize("ln,le", GEOS)
@pytest.mark.parametrize("arith", [0, 1], ids=["shoup", "fold"])
def test_emulated_ntt_matches_oracle(emu, ln, le, arith):
    n = 1 << ln
    before = emu.emu_overflows()
    for limb in ((0, 3, 5) if ln <= 13 else (1, 4)):   # N = 16384 needs q = 1 mod 32768: two of the pinned primes
        q = PRIMES_60[limb][0]
        psi = pow(PRIMES_60[limb][2], 8192 // n, q) if ln <= 13 else po.min_primitive_2n_root(n, q)
        orc = Oracle(ln, [q], [psi])
        pats = [orc.fill(1, 77 + limb).ravel().copy(), np.full(n, q - 1, np.uint64), np.zeros(n, np.uint64),
                np.where(np.arange(n) % 2 == 0, q - 1, 0).astype(np.uint64),
                np.where(np.arange(n) < n // 2, q - 1, 1).astype(np.uint64)]
        for a in pats:
            rc, got = run(emu, arith, ln, le, 0, q, psi, a)
            assert rc == 0 and np.array_equal(got, orc.ntt_fwd(a))
            rc, got = run(emu, arith, ln, le, 1, q, psi, a)
            assert rc == 0 and np.array_equal(got, orc.ntt_inv(a))
    assert emu.emu_overflows() == before, "lazy arithmetic wrapped around 2^64"


@pytest.mark.parametrize("ln,le", GEOS + [(9, 4), (12, 3), (12, 5)])
def test_lds_regions_are_wave_private(emu, ln, le):
    """The kernels run every exchange but the first (forward) / last (inverse) without a workgroup barrier.  That is only
    legal if those exchanges read and write nothing outside the issuing wave's private LDS region, if the forward all-to-all
    exchange is READ inside the own region and the inverse one WRITTEN inside it, and if every address map is injective:
    tools/emulate.cpp enumerates every (thread, word) address of every exchange in both directions."""
    assert emu.emu_check_lds_regions(ln, le) == 0


def test_emulated_30bit_prime_uses_shoup_only(emu):
    orc = Oracle(10, [PRIME_30], [PSI_30_N1024])
    a = orc.fill(1, 1).ravel().copy()
    rc, got = run(emu, 0, 10, 4, 0, PRIME_30, PSI_30_N1024, a)
    assert rc == 0 and np.array_equal(got, orc.ntt_fwd(a))
    rc, back = run(emu, 0, 10, 4, 1, PRIME_30, PSI_30_N1024, got)
    assert rc == 0 and np.array_equal(back, a)
    assert run(emu, 1, 10, 4, 0, PRIME_30, PSI_30_N1024, a)[0] == 2000  # not fold-eligible


@pytest.mark.parametrize("ln", [8, 10, 12, 13])
def test_emulated_ct_mul_lazy_path_matches_oracle_without_wraps(emu, ln):
    """kernels.h ct_mul_kernel's lazy FoldArith data path (same transform code, same dyadic sequence) on the CPU with the
    wrap-around and mul60-precondition counters armed: worst-case inputs (all q-1) and random ones."""
    n = 1 << ln
    emu.emu_ct_mul.argtypes = [C.c_int, C.c_uint64, C.c_uint64, U, U, U, U, U]
    emu.emu_ct_mul.restype = C.c_int
    before = emu.emu_overflows()
    for limb in (0, 5):
        q = PRIMES_60[limb][0]
        psi = pow(PRIMES_60[limb][2], 8192 // n, q)
        orc = Oracle(ln, [q], [psi])
        rnd = orc.fill(4, 900 + limb).reshape(4, n)
        cases = [rnd, np.full((4, n), q - 1, np.uint64),
                 np.stack([np.full(n, q - 1, np.uint64), np.zeros(n, np.uint64), np.ones(n, np.uint64), np.full(n, q - 1, np.uint64)])]
        for polys in cases:
            polys = np.ascontiguousarray(polys, dtype=np.uint64)
            a0, a1, b0, b1 = (np.ascontiguousarray(polys[i]) for i in range(4))
            out = np.zeros(3 * n, np.uint64)
            rc = emu.emu_ct_mul(ln, q, psi, a0.ctypes.data_as(U), a1.ctypes.data_as(U), b0.ctypes.data_as(U), b1.ctypes.data_as(U), out.ctypes.data_as(U))
            assert rc == 0
            a = np.stack([a0, a1]).reshape(1, 2, 1, n)
            b = np.stack([b0, b1]).reshape(1, 2, 1, n)
            want = orc.ct_mul(a, b).reshape(3 * n)
            assert np.array_equal(out, want)
    assert emu.emu_overflows() == before, "lazy arithmetic wrapped around 2^64 or broke a mul60 precondition"


@pytest.mark.parametrize("ln", [15, 16])
@pytest.mark.parametrize("arith", [0, 1], ids=["shoup", "fold"])
def test_emulated_split_transform_matches_oracle(emu, ln, arith):
    """N = 2^15, 2^16: column stages (ntt_top.h) + 4096-point kernels on sub-tree tables == the oracle's one-piece transform"""
    n = 1 << ln
    emu.emu_ntt_split.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, U, U]
    emu.emu_ntt_split.restype = C.c_int
    before = emu.emu_overflows()
    # fold-eligible prime = 1 mod 2N: 2^60 - d with d = -1 mod 2N; the 5th pinned prime is 1 mod 2^16 only
    q = next(c for c in ((1 << 60) - (k * 2 * n - 1) for k in range(1, 1 << (24 - ln - 1))) if c % (2 * n) == 1 and po.is_prime(c))
    psi = po.min_primitive_2n_root(n, q)
    orc = Oracle(ln, [q], [psi])
    for a in (orc.fill(1, 31).ravel().copy(), np.full(n, q - 1, np.uint64), np.where(np.arange(n) % 3 == 0, q - 1, 1).astype(np.uint64)):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        for inv, ref in ((0, orc.ntt_fwd), (1, orc.ntt_inv)):
            out = np.zeros_like(a)
            rc = emu.emu_ntt_split(arith, ln, inv, q, psi, a.ctypes.data_as(U), out.ctypes.data_as(U))
            assert rc == 0 and np.array_equal(out, ref(a))
    assert emu.emu_overflows() == before, "lazy arithmetic wrapped around 2^64"


@pytest.mark.parametrize("arith", [0, 1], ids=["shoup", "fold"])
def test_emulated_halves_transform_matches_oracle(emu, arith):
    """N = 8192 as a register column stage + two 4096-point sub-transforms through one LDS buffer (ntt_halves.h, the 256-thread kernels of
    kernels_halves.h) == the oracle's one-piece transform, worst-case residues included, no 64-bit wrap of the lazy arithmetic."""
    n = 8192
    emu.emu_ntt_halves.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint64, U, U]
    emu.emu_ntt_halves.restype = C.c_int
    before = emu.emu_overflows()
    for limb in (0, 2, 5):
        q, psi = PRIMES_60[limb][0], PRIMES_60[limb][2]
        orc = Oracle(13, [q], [psi])
        pats = [orc.fill(1, 177 + limb).ravel().copy(), np.full(n, q - 1, np.uint64), np.zeros(n, np.uint64),
                np.where(np.arange(n) % 2 == 0, q - 1, 0).astype(np.uint64), np.where(np.arange(n) < n // 2, q - 1, 1).astype(np.uint64),
                np.where(np.arange(n) < n // 2, 0, q - 1).astype(np.uint64)]
        for a in pats:
            a = np.ascontiguousarray(a, dtype=np.uint64)
            for inv, ref in ((0, orc.ntt_fwd), (1, orc.ntt_inv)):
                out = np.zeros_like(a)
                rc = emu.emu_ntt_halves(arith, inv, q, psi, a.ctypes.data_as(U), out.ctypes.data_as(U))
                assert rc == 0 and np.array_equal(out, ref(a)), (limb, inv)
    assert emu.emu_overflows() == before, "lazy arithmetic wrapped around 2^64"


@pytest.mark.parametrize("ln", [8, 10, 12, 13])
def test_emulated_forward_transform_of_words_below_2_60(emu, ln):
    """The key-switch kernels hand the digits (residues of ANOTHER limb: any word < 2^60) to the forward transform without canonicalising them
    (NttBody FWD_IN = kRedB): same result as the oracle's transform of the reduced words, no 64-bit wrap, no broken multiply-add precondition."""
    n = 1 << ln
    emu.emu_ntt_fwd_any60.argtypes = [C.c_int, C.c_uint64, C.c_uint64, U, U]
    emu.emu_ntt_fwd_any60.restype = C.c_int
    before = emu.emu_overflows()
    rng = np.random.default_rng(ln)
    for limb in (0, 3, 5):
        q = PRIMES_60[limb][0]
        psi = pow(PRIMES_60[limb][2], 8192 // n, q)
        orc = Oracle(ln, [q], [psi])
        top = (1 << 60) - 1
        pats = [np.full(n, top, np.uint64), rng.integers(0, 1 << 60, n, dtype=np.uint64), np.where(np.arange(n) % 2 == 0, top, q).astype(np.uint64),
                np.where(np.arange(n) < n // 2, top, 0).astype(np.uint64), np.where(np.arange(n) < n // 2, q - 1, top).astype(np.uint64)]
        for a in pats:
            a = np.ascontiguousarray(a, dtype=np.uint64)
            out = np.zeros_like(a)
            assert emu.emu_ntt_fwd_any60(ln, q, psi, a.ctypes.data_as(U), out.ctypes.data_as(U)) == 0
            assert np.array_equal(out, orc.ntt_fwd(np.ascontiguousarray(a % np.uint64(q))))
    assert emu.emu_overflows() == before, "lazy arithmetic wrapped around 2^64"


def test_dot30_column_accumulators_match_128_bit_arithmetic(emu):
    """The FoldArith matvec kernels accumulate products of canonical residues in three 64-bit columns (operands split at bit 30)
    and fold every 8 terms: same result as exact integer arithmetic, no 64-bit wrap, for random and worst-case operands."""
    emu.emu_dot30.argtypes = [C.c_uint64, U, U, C.c_size_t]
    emu.emu_dot30.restype = C.c_uint64
    rng = np.random.default_rng(5)
    before = emu.emu_overflows()
    for q, _, _ in PRIMES_60[:6]:
        for n in (1, 7, 8, 9, 16, 31, 32, 64, 127, 1000):
            cases = [(rng.integers(0, q, n, dtype=np.uint64), rng.integers(0, q, n, dtype=np.uint64)),
                     (np.full(n, q - 1, np.uint64), np.full(n, q - 1, np.uint64)),
                     (np.full(n, (1 << 30) - 1, np.uint64), np.full(n, q - 1, np.uint64)),
                     (np.full(n, q - 1, np.uint64), np.full(n, ((1 << 30) - 1) << 30, np.uint64) % np.uint64(q))]
            for a, b in cases:
                a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
                want = sum(int(x) * int(y) for x, y in zip(a, b)) % q
                assert emu.emu_dot30(q, a.ctypes.data_as(U), b.ctypes.data_as(U), n) == want
    assert emu.emu_overflows() == before


# ---- round 6: the per-limb arithmetic classes (modarith.h F64Arith, FoldScaledArith) -------------------------------------------------------------------
CLASS_CASES = [(2, bits) for bits in (20, 30, 31, 33, 40, 45, 47)] + [(3, bits) for bits in (54, 56, 57, 58, 59)] + [(4, bits) for bits in (30, 47, 48, 49, 50)]
CLASS_NAMES = {2: "f64_", 3: "fold_scaled_", 4: "f64_wide_"}


def _class_patterns(orc, n, q):
    return [orc.fill(1, 77).ravel().copy(), np.full(n, q - 1, np.uint64), np.zeros(n, np.uint64),
            np.where(np.arange(n) % 2 == 0, q - 1, 0).astype(np.uint64), np.where(np.arange(n) < n // 2, q - 1, 1).astype(np.uint64),
            np.where(np.arange(n) % 2 == 0, q // 2, q // 2 + 1).astype(np.uint64)]


@pytest.mark.parametrize("ln,le", [(8, 4), (10, 4), (12, 4), (13, 4), (14, 4)])
@pytest.mark.parametrize("arith,bits", CLASS_CASES, ids=[CLASS_NAMES[a] + str(b) for a, b in CLASS_CASES])
def test_emulated_class_transforms_match_oracle(emu, ln, le, arith, bits):
    """F64Arith / F64WideArith (residues as doubles, error-free FMA products; the |y| < 2^51 precondition of every product is armed in the emulator build;
    the wide form's static plans reduce words inside the transforms) and FoldScaledArith (2^k - d0 carried as 2^60 - d) through the very per-thread code the
    kernels run, both directions, extreme residues included."""
    n = 1 << ln
    before = emu.emu_overflows()
    P = ntt_primes(ln, 2, bits)
    ran = 0
    for q, psi in zip(P.moduli, P.psi):
        orc = Oracle(ln, [q], [psi])
        for a in _class_patterns(orc, n, q):
            rc, got = run(emu, arith, ln, le, 0, q, psi, a)
            if rc == 2000:          # a prime of this width that the class does not take (d0 2^(60-k) >= 2^24): the library runs it on another class
                assert arith == 3
                break
            assert rc == 0 and np.array_equal(got, orc.ntt_fwd(a))
            rc, got = run(emu, arith, ln, le, 1, q, psi, a)
            assert rc == 0 and np.array_equal(got, orc.ntt_inv(a))
            ran += 1
    assert ran or arith == 3
    assert emu.emu_overflows() == before, "a lazy-arithmetic precondition was broken"


@pytest.mark.parametrize("ln", [8, 12, 13])
@pytest.mark.parametrize("arith,bits,lazy", [(2, 30, 1), (2, 47, 1), (2, 40, 0), (3, 59, 1), (3, 57, 1), (3, 59, 0), (1, 60, 1), (4, 50, 1), (4, 48, 1), (4, 49, 0)],
                         ids=["f64_30_lazy", "f64_47_lazy", "f64_40_generic", "fscaled_59_lazy", "fscaled_57_lazy", "fscaled_59_generic", "fold_lazy", "f64w_50_lazy", "f64w_48_lazy",
                              "f64w_49_generic"])
def test_emulated_class_fused_multiply_matches_oracle(emu, ln, arith, bits, lazy):
    """the fused multiply's two data paths for the classes: lazy products of forward outputs straight into the inverse (ct_mul_quad / ct_mul_dual, ntt_core.h
    NttBody::prod; FoldScaledArith's products carry the scale twice and end on last2) and the generic path through canonical words (ct_mul_kernel)"""
    n = 1 << ln
    fn = emu.emu_ct_mul_lazy_class if lazy else emu.emu_ct_mul_class
    fn.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint64, U, U, U, U, U]
    fn.restype = C.c_int
    before = emu.emu_overflows()
    P = ntt_primes(ln, 1, bits)
    q, psi = P.moduli[0], P.psi[0]
    orc = Oracle(ln, [q], [psi])
    for polys in (orc.fill(4, 900).reshape(4, n), np.full((4, n), q - 1, np.uint64)):
        polys = np.ascontiguousarray(polys, dtype=np.uint64)
        a0, a1, b0, b1 = (np.ascontiguousarray(polys[i]) for i in range(4))
        out = np.zeros(3 * n, np.uint64)
        assert fn(arith, ln, q, psi, *(v.ctypes.data_as(U) for v in (a0, a1, b0, b1, out))) == 0
        want = orc.ct_mul(np.stack([a0, a1]).reshape(1, 2, 1, n), np.stack([b0, b1]).reshape(1, 2, 1, n)).reshape(3 * n)
        assert np.array_equal(out, want)
    assert emu.emu_overflows() == before


def test_emulated_quarters_transform_matches_oracle(emu):
    """N = 16384 as two register column stages + four 4096-point sub-transforms through one LDS buffer (ntt_quarters.h, the 256-thread kernels of
    kernels_quarters.h) == the oracle's one-piece transform, worst-case residues included, no 64-bit wrap of the lazy arithmetic."""
    n = 16384
    emu.emu_ntt_quarters.argtypes = [C.c_int, C.c_uint64, C.c_uint64, U, U]
    emu.emu_ntt_quarters.restype = C.c_int
    before = emu.emu_overflows()
    for limb in (1, 2, 4):   # the pinned primes that are 1 mod 32768
        q = PRIMES_60[limb][0]
        psi = po.min_primitive_2n_root(n, q)
        orc = Oracle(14, [q], [psi])
        for a in (orc.fill(1, 91 + limb).ravel().copy(), np.full(n, q - 1, np.uint64), np.where(np.arange(n) % 2 == 0, q - 1, 0).astype(np.uint64),
                  np.where(np.arange(n) < n // 2, q - 1, 1).astype(np.uint64), np.where(np.arange(n) % 4 < 2, q - 1, 0).astype(np.uint64)):
            a = np.ascontiguousarray(a, dtype=np.uint64)
            for inv, ref in ((0, orc.ntt_fwd), (1, orc.ntt_inv)):
                out = np.zeros_like(a)
                assert emu.emu_ntt_quarters(inv, q, psi, a.ctypes.data_as(U), out.ctypes.data_as(U)) == 0
                assert np.array_equal(out, ref(a)), (limb, inv)
    assert emu.emu_overflows() == before, "lazy arithmetic wrapped around 2^64"
    assert emu.emu_ntt_quarters(0, PRIME_30, PSI_30_N1024, None, None) == 2000   # FoldArith only


def test_emulated_fold_products_through_the_twiddle_chain(emu):
    """FoldArith::prod_tw + mul_ptw_add (round 6: the fused multiply's tensor step and the key products): addend + y b mod q for ANY 64-bit y, b up to
    2^60 + 2^29 - 1 and addends up to 2^62 - 2^58, at the extremes and at random, with the exact no-wrap checks of the chain armed; d up to 2^24 - 1
    (the scaled-fold moduli 2^60 - d are not prime: only the arithmetic modulo 2^60 - d is checked)."""
    emu.emu_fold_ptw.argtypes = [C.c_uint64] * 4
    emu.emu_fold_ptw.restype = C.c_uint64
    before = emu.emu_overflows()
    rng = np.random.default_rng(61)
    ds = [(1 << 60) - PRIMES_60[0][0], (1 << 60) - PRIMES_60[5][0], (1 << 24) - 1, 1, (1 << 24) - (1 << 10)]
    for d in ds:
        q = (1 << 60) - d
        ys = [0, 1, (1 << 64) - 1, 15 << 60, (1 << 63) + 12345, (1 << 32) - 1, ((1 << 32) - 1) << 32] + [int(v) for v in rng.integers(0, 1 << 63, 40, dtype=np.uint64)] \
            + [int(v) | (1 << 63) for v in rng.integers(0, 1 << 63, 20, dtype=np.uint64)]
        bs = [0, 1, q - 1, q, (1 << 60) - 1, (1 << 60) + (1 << 29) - 1, (1 << 60), (1 << 30) - 1, ((1 << 30) - 1) << 30, (1 << 29) - 1, ((1 << 31) - 1) << 29] \
            + [int(v) for v in rng.integers(0, q, 40, dtype=np.uint64)]
        adds = [0, (1 << 60) + 16 * d, (1 << 62) - (1 << 58) - 1, q - 1]
        for y in ys:
            for b in bs:
                for add in adds:
                    assert emu.emu_fold_ptw(d, y, b, add) == (add + y * b) % q, (d, y, b, add)
    assert emu.emu_overflows() == before, "the chain wrapped around 2^64"
