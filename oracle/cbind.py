"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE ONLY - see oracle/oracle.c).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_U64P = C.POINTER(C.c_uint64)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liboracle.so"])
    return _LIB


def _load():
    lib = C.CDLL(build())
    lib.orc_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, _U64P, _U64P]
    lib.orc_ctx_create.restype = C.c_int
    lib.orc_ctx_destroy.argtypes = [C.c_void_p]
    lib.orc_get_root_powers.argtypes = [C.c_void_p, C.c_uint32, _U64P, _U64P]
    lib.orc_schoolbook_negacyclic.argtypes = [_U64P, _U64P, _U64P, C.c_uint64, C.c_uint64]
    lib.orc_max_threads.restype = C.c_int
    lib.orc_ntt_fwd.argtypes = [C.c_void_p, _U64P, C.c_size_t, C.c_int]
    lib.orc_ntt_inv.argtypes = [C.c_void_p, _U64P, C.c_size_t, C.c_int]
    lib.orc_dyadic.argtypes = [C.c_void_p, C.c_int, _U64P, _U64P, _U64P, C.c_size_t, C.c_int]
    lib.orc_ct_mul.argtypes = [C.c_void_p, _U64P, _U64P, _U64P, C.c_size_t, C.c_int]
    lib.orc_ct_mul_timed.argtypes = [C.c_void_p, _U64P, _U64P, _U64P, C.c_size_t, C.c_int, C.c_int]
    lib.orc_ct_mul_timed.restype = C.c_double
    lib.orc_isa.restype = C.c_char_p
    lib.orc_ct_mul_schoolbook.argtypes = [C.c_void_p, _U64P, _U64P, _U64P, C.c_size_t, C.c_int]
    lib.orc_relinearize.argtypes = [C.c_void_p, _U64P, _U64P, _U64P, C.c_size_t, C.c_int]
    lib.orc_keyswitch_hybrid.argtypes = [C.c_void_p, _U64P, _U64P, _U64P, C.c_size_t, C.c_int, C.c_int]
    lib.orc_rotate_hoisted.argtypes = [C.c_void_p, _U64P, _U64P, C.POINTER(C.c_uint32), _U64P, C.c_size_t, C.c_int]
    lib.orc_rotate_hoisted.restype = None
    lib.orc_rotate_hoisted_qp.argtypes = [C.c_void_p, _U64P, _U64P, C.POINTER(C.c_uint32), _U64P, C.c_size_t, C.c_int]
    lib.orc_rotate_hoisted_qp.restype = None
    lib.orc_switch_key_qp.argtypes = [C.c_void_p, _U64P, _U64P, _U64P, C.c_size_t, C.c_int]
    lib.orc_switch_key_qp.restype = None
    lib.orc_rescale.argtypes = [C.c_void_p, _U64P, _U64P, C.c_size_t]
    lib.orc_base_extend.argtypes = [C.c_void_p, C.c_int, _U64P, C.c_size_t, _U64P, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_size_t]
    lib.orc_base_extend.restype = None
    lib.orc_apply_galois.argtypes = [C.c_void_p, _U64P, _U64P, C.c_size_t, C.c_uint32]
    lib.orc_switch_key.argtypes = [C.c_void_p, _U64P, _U64P, _U64P, C.c_size_t, C.c_int]
    lib.orc_matvec_plain.argtypes = [C.c_void_p, _U64P, _U64P, _U64P, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
    lib.orc_matvec_scalar.argtypes = [C.c_void_p, _U64P, _U64P, _U64P, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
    lib.orc_reduce_sum.argtypes = [C.c_void_p, _U64P, _U64P, C.c_size_t, C.c_size_t]
    lib.orc_fill_splitmix.argtypes = [C.c_void_p, _U64P, C.c_size_t, C.c_uint64]
    for f in ("orc_ctx_destroy", "orc_get_root_powers", "orc_schoolbook_negacyclic", "orc_ntt_fwd", "orc_ntt_inv",
              "orc_dyadic", "orc_ct_mul", "orc_ct_mul_schoolbook", "orc_relinearize", "orc_matvec_scalar", "orc_apply_galois", "orc_switch_key", "orc_rescale", "orc_keyswitch_hybrid", "orc_matvec_plain", "orc_reduce_sum", "orc_fill_splitmix"):
        getattr(lib, f).restype = None
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_U64P)


class Oracle:
    """CPU evaluator over numpy uint64 arrays laid out [batch][component][limb][N]."""

    DYADIC = {"mul": 0, "mul_add": 1, "add": 2, "sub": 3, "negate": 4}

    def __init__(self, log2_n: int, moduli, psi):
        self.log2_n, self.n, self.L = log2_n, 1 << log2_n, len(moduli)
        self.moduli, self.psi = tuple(int(q) for q in moduli), tuple(int(w) for w in psi)
        m = (C.c_uint64 * self.L)(*self.moduli)
        w = (C.c_uint64 * self.L)(*self.psi)
        h = C.c_void_p()
        rc = lib().orc_ctx_create(C.byref(h), log2_n, self.L, m, w)
        if rc:
            raise ValueError(f"orc_ctx_create failed rc={rc}")
        self._h = h

    @classmethod
    def from_params(cls, p):
        return cls(p.log2_n, p.moduli, p.psi)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_ctx_destroy(self._h)
            self._h = None

    def max_threads(self) -> int:
        return lib().orc_max_threads()

    def root_powers(self, limb: int):
        rp, irp = np.empty(self.n, np.uint64), np.empty(self.n, np.uint64)
        lib().orc_get_root_powers(self._h, limb, _p(rp), _p(irp))
        return rp, irp

    def schoolbook(self, a, b, q):
        out = np.empty_like(a)
        lib().orc_schoolbook_negacyclic(_p(out), _p(a), _p(b), a.size, int(q))
        return out

    def _npolys(self, a):
        assert a.size % (self.L * self.n) == 0
        return a.size // (self.L * self.n)

    def ntt_fwd(self, a, threads=1):
        out = np.ascontiguousarray(a).copy()
        lib().orc_ntt_fwd(self._h, _p(out), self._npolys(out), threads)
        return out

    def ntt_inv(self, a, threads=1):
        out = np.ascontiguousarray(a).copy()
        lib().orc_ntt_inv(self._h, _p(out), self._npolys(out), threads)
        return out

    def dyadic(self, op, a, b=None, acc=None, threads=1):
        out = np.empty_like(a) if acc is None else np.ascontiguousarray(acc).copy()
        lib().orc_dyadic(self._h, self.DYADIC[op], _p(out), _p(a), _p(b) if b is not None else None, self._npolys(a), threads)
        return out

    def ct_mul(self, a2, b2, threads=1, schoolbook=False):
        batch = a2.size // (2 * self.L * self.n)
        out = np.empty(batch * 3 * self.L * self.n, np.uint64)
        f = lib().orc_ct_mul_schoolbook if schoolbook else lib().orc_ct_mul
        f(self._h, _p(out), _p(a2), _p(b2), batch, threads)
        return out.reshape(batch, 3, self.L, self.n)

    def ct_mul_timed(self, a2, b2, threads=1, reps=3):
        """bench.py's CPU column: (result, best wall seconds of `reps` passes) with NUMA-local, pre-touched working copies."""
        batch = a2.size // (2 * self.L * self.n)
        out = np.empty(batch * 3 * self.L * self.n, np.uint64)
        t = lib().orc_ct_mul_timed(self._h, _p(out), _p(np.ascontiguousarray(a2)), _p(np.ascontiguousarray(b2)), batch, threads, reps)
        if t < 0:
            raise MemoryError("orc_ct_mul_timed: allocation failed")
        return out.reshape(batch, 3, self.L, self.n), t

    @staticmethod
    def isa() -> str:
        return lib().orc_isa().decode()

    def relinearize(self, ct3, evk, threads=1):
        batch = ct3.size // (3 * self.L * self.n)
        out = np.empty(batch * 2 * self.L * self.n, np.uint64)
        lib().orc_relinearize(self._h, _p(out), _p(np.ascontiguousarray(ct3)), _p(np.ascontiguousarray(evk)), batch, threads)
        return out.reshape(batch, 2, self.L, self.n)

    def keyswitch_hybrid(self, ct, key, in_comps, threads=1):
        """self is the EXTENDED context (last limb = special prime); ct: [batch][in_comps][L-1][N]."""
        ct = np.ascontiguousarray(ct)
        batch = ct.size // (in_comps * (self.L - 1) * self.n)
        out = np.empty(batch * 2 * (self.L - 1) * self.n, np.uint64)
        lib().orc_keyswitch_hybrid(self._h, _p(out), _p(ct), _p(np.ascontiguousarray(key)), batch, in_comps, threads)
        return out.reshape(batch, 2, self.L - 1, self.n)

    def rotate_hoisted(self, ct2, elts, keys, threads=1):
        """self is the EXTENDED context; ct2: [2][L-1][N] (one item); keys: [k][L-1][2][L][N] -> [k][2][L-1][N]"""
        k = len(elts)
        out = np.empty(k * 2 * (self.L - 1) * self.n, np.uint64)
        e = (C.c_uint32 * k)(*[int(g) for g in elts])
        lib().orc_rotate_hoisted(self._h, _p(out), _p(np.ascontiguousarray(ct2)), e, _p(np.ascontiguousarray(keys)), k, threads)
        return out.reshape(k, 2, self.L - 1, self.n)

    def rotate_hoisted_qp(self, ct2, elts, keys, threads=1):
        """self is the EXTENDED context; ct2: [2][L-1][N] (one item) -> [1 + k][2][L][N], NTT domain over Q P, not divided by P:
        block 0 = P ct, block 1 + r = P sigma_g(ct) + the key-switching term of rotation r."""
        k = len(elts)
        out = np.empty((k + 1) * 2 * self.L * self.n, np.uint64)
        e = (C.c_uint32 * max(k, 1))(*[int(g) for g in elts])
        kk = np.ascontiguousarray(keys) if k else np.zeros(1, np.uint64)
        lib().orc_rotate_hoisted_qp(self._h, _p(out), _p(np.ascontiguousarray(ct2)), e, _p(kk), k, threads)
        return out.reshape(k + 1, 2, self.L, self.n)

    def switch_key_qp(self, ct2, key, threads=1):
        """self is the EXTENDED context; ct2: [batch][2][L-1][N] -> [batch][2][L][N]: sum_j NTT(lift([c1]_j)) (.) key_j, NTT domain."""
        ct2 = np.ascontiguousarray(ct2)
        batch = ct2.size // (2 * (self.L - 1) * self.n)
        out = np.empty(batch * 2 * self.L * self.n, np.uint64)
        lib().orc_switch_key_qp(self._h, _p(out), _p(ct2), _p(np.ascontiguousarray(key)), batch, threads)
        return out.reshape(batch, 2, self.L, self.n)

    def rescale(self, x):
        x = np.ascontiguousarray(x)
        npolys = self._npolys(x)
        out = np.empty(npolys * (self.L - 1) * self.n, np.uint64)
        lib().orc_rescale(self._h, _p(out), _p(x), npolys)
        return out.reshape(x.shape[:-2] + (self.L - 1, self.n))

    def base_extend(self, x, src_limb0, dst_limb0, n_dst):
        """x: [..., n_src, N] residues mod limbs src_limb0.. -> the same centred integers mod limbs dst_limb0.. ([..., n_dst, N])"""
        x = np.ascontiguousarray(x)
        ns = x.shape[-2]
        npolys = x.size // (ns * self.n)
        out = np.empty(x.shape[:-2] + (n_dst, self.n), np.uint64)
        lib().orc_base_extend(self._h, 0, _p(out), n_dst, _p(x), ns, src_limb0, ns, dst_limb0, n_dst, 1, npolys)
        return out

    def scale_round(self, x, drop_limb0, n_drop, keep_limb0, n_keep, multiplier):
        """x: [..., L, N] on all limbs -> round(multiplier * X / prod(dropped limbs)) on the kept limbs ([..., n_keep, N])"""
        x = np.ascontiguousarray(x)
        npolys = self._npolys(x)
        out = np.empty(x.shape[:-2] + (n_keep, self.n), np.uint64)
        lib().orc_base_extend(self._h, 1, _p(out), n_keep, _p(x), self.L, drop_limb0, n_drop, keep_limb0, n_keep, int(multiplier), npolys)
        return out

    def apply_galois(self, x, galois_elt):
        x = np.ascontiguousarray(x)
        out = np.empty_like(x)
        lib().orc_apply_galois(self._h, _p(out), _p(x), self._npolys(x), int(galois_elt))
        return out

    def switch_key(self, ct2, key, threads=1):
        batch = ct2.size // (2 * self.L * self.n)
        out = np.empty(batch * 2 * self.L * self.n, np.uint64)
        lib().orc_switch_key(self._h, _p(out), _p(np.ascontiguousarray(ct2)), _p(np.ascontiguousarray(key)), batch, threads)
        return out.reshape(batch, 2, self.L, self.n)

    def matvec_plain(self, W, x, rows, cols, comps=2, threads=1):
        y = np.empty(rows * comps * self.L * self.n, np.uint64)
        lib().orc_matvec_plain(self._h, _p(y), _p(W), _p(x), rows, cols, comps, threads)
        return y.reshape(rows, comps, self.L, self.n)

    def matvec_scalar(self, w, x, rows, cols, comps=2, threads=1):
        y = np.empty(rows * comps * self.L * self.n, np.uint64)
        lib().orc_matvec_scalar(self._h, _p(y), _p(np.ascontiguousarray(w)), _p(np.ascontiguousarray(x)), rows, cols, comps, threads)
        return y.reshape(rows, comps, self.L, self.n)

    def reduce_sum(self, cts, comps):
        words = comps * self.L * self.n
        count = cts.size // words
        out = np.empty(words, np.uint64)
        lib().orc_reduce_sum(self._h, _p(out), _p(cts), count, comps)
        return out.reshape(comps, self.L, self.n)

    def fill(self, n_rns_polys: int, seed: int):
        out = np.empty(n_rns_polys * self.L * self.n, np.uint64)
        lib().orc_fill_splitmix(self._h, _p(out), n_rns_polys, seed)
        return out.reshape(n_rns_polys, self.L, self.n)
