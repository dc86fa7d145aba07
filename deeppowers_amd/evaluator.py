"""deeppowers.fhe - Context / Plaintext / Ciphertext / Evaluator over the HIP C ABI (Python mirror).

The reference has no Ciphertext/Evaluator; this mirrors the BUILD-SPEC operator API
of SURVEY.md section 8(a)/(b), in the reference's house style (the C++ twin is include/deeppowers/fhe.hpp):
heavy objects own their buffers (/root/reference/src/core/execution/model.hpp:88-89), the device is
chosen by id (/root/reference/src/api/cpp/src/deeppowers.cpp:15), errors are exceptions carrying a
deeppowers::common::ErrorCode (/root/reference/src/common/error.hpp:42-53), the stream is optional and last
(/root/reference/src/core/hal/hal.hpp:95).

PyTorch is used only for device memory and streams; every operation is a HIP kernel launched through
libdpfhe_hip.so.  Words are u64 residues stored in int64 tensors (bit pattern), layout
[batch...][component][limb][N].
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _cabi
from .params import FheParams


def _resolve_stream(stream, device):
    """`stream=None` means the CURRENT stream OF THE CONTEXT'S DEVICE (not of whatever device is current)."""
    return torch.cuda.current_stream(device) if stream is None else stream


def to_device(words: np.ndarray, device) -> torch.Tensor:
    """numpy uint64 -> int64 device tensor (same bits)."""
    a = np.ascontiguousarray(words, dtype=np.uint64)
    return torch.from_numpy(a.view(np.int64)).to(device)


def to_host(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().contiguous().numpy().view(np.uint64)


class Context:
    """A0: immutable per-device tables (twiddles, Shoup/Barrett/fold constants) for one FheParams."""

    def __init__(self, params: FheParams, device_id: int = 0):
        self.params = params
        self.device_id = int(device_id)
        self.device = torch.device("cuda", self.device_id)
        self._lib = _cabi.load()
        L = params.n_limbs
        h = C.c_void_p()
        m = (C.c_uint64 * L)(*params.moduli)
        w = (C.c_uint64 * L)(*params.psi)
        _cabi.check(self._lib.dpfhe_ctx_create(C.byref(h), params.log2_n, L, m, w, self.device_id), "dpfhe_ctx_create")
        self._h = h

    @property
    def handle(self):
        return self._h

    @property
    def uses_fold(self) -> bool:
        return bool(self._lib.dpfhe_ctx_uses_fold(self._h))

    def release_scratch(self, stream=None, all_streams: bool = False) -> None:
        """hand back the scratch arena the composed large-ring operations keep for `stream` (a torch stream, None = the current one), or every arena of the
        context (dpfhe_ctx_release_scratch); scratch_bytes = what the context holds"""
        sp = 0 if all_streams else (stream.cuda_stream if stream is not None else torch.cuda.current_stream(self.device).cuda_stream)
        _cabi.check(self._lib.dpfhe_ctx_release_scratch(self._h, C.c_void_p(sp), 1 if all_streams else 0), "dpfhe_ctx_release_scratch")

    @property
    def scratch_bytes(self) -> int:
        return int(self._lib.dpfhe_ctx_scratch_bytes(self._h))

    ARITH_NAMES = ("shoup", "fold", "f64", "fold_scaled", "f64_wide")

    @property
    def limb_classes(self) -> tuple:
        """per limb, the arithmetic the transforms and the fused multiply run it on (dpfhe_ctx_limb_class): 'shoup' | 'fold' | 'f64' | 'fold_scaled' | 'f64_wide'"""
        return tuple(self.ARITH_NAMES[self._lib.dpfhe_ctx_limb_class(self._h, i)] for i in range(self.params.n_limbs))

    # ---- which form of the fused multiply this context launches (include/dpfhe.h "A0, continued") ----
    def tune_info(self) -> dict:
        """{"chosen": name, "source": ..., "probe_us": {name: us}, "probe_pairs": n, "probe_reps": r}; forms are bit-identical."""
        t = _cabi.TuneInfo()
        _cabi.check(self._lib.dpfhe_ctx_tune_info(self._h, C.byref(t)), "dpfhe_ctx_tune_info")
        name = lambda v: self._lib.dpfhe_ct_mul_variant_name(v).decode()
        return {"chosen": name(t.chosen), "n_variants": t.n_variants, "source": _cabi.TUNE_SOURCES[t.source] if 0 <= t.source < len(_cabi.TUNE_SOURCES) else str(t.source),
                "probe_pairs": t.probe_pairs, "probe_reps": t.probe_reps,
                "probe_us": {name(v): round(float(t.probe_us[v]), 2) for v in range(t.n_variants) if t.probe_us[v] >= 0}}

    def autotune(self, work: torch.Tensor, reps: int = 3, stream=None) -> dict:
        """Re-measures the forms on the caller's scratch tensor (contents are overwritten; synchronises the stream)."""
        if work.dtype != torch.int64 or not work.is_contiguous() or work.device != self.device:
            raise _cabi.DpfheError(2000, "autotune scratch must be a contiguous int64 tensor on the context's device")
        s = _resolve_stream(stream, self.device)
        _cabi.check(self._lib.dpfhe_ctx_autotune(self._h, work.data_ptr(), work.numel(), reps, s.cuda_stream), "dpfhe_ctx_autotune")
        return self.tune_info()

    def variants(self):
        return [self._lib.dpfhe_ct_mul_variant_name(v).decode() for v in range(self.tune_info()["n_variants"])]

    def set_ct_mul_variant(self, name: str):
        names = [self._lib.dpfhe_ct_mul_variant_name(v).decode() for v in range(8)]
        if name not in names or not name:
            raise _cabi.DpfheError(2000, f"unknown form {name!r}")
        _cabi.check(self._lib.dpfhe_ctx_set_ct_mul_variant(self._h, names.index(name)), "dpfhe_ctx_set_ct_mul_variant")

    def set_scratch_limit(self, mib: int):
        """Slice size (MiB of scratch) of the composed large-ring operations (log2_n >= 14); a set-up call."""
        _cabi.check(self._lib.dpfhe_ctx_set_scratch_limit(self._h, int(mib)), "dpfhe_ctx_set_scratch_limit")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dpfhe_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # buffers
    def empty(self, *lead, components: int) -> torch.Tensor:
        p = self.params
        return torch.empty(*lead, components, p.n_limbs, p.n, dtype=torch.int64, device=self.device)


class Plaintext:
    """A5: one RNS polynomial [L][N] (optionally batched), with an is_ntt flag."""

    def __init__(self, data: torch.Tensor, is_ntt: bool = False):
        self.data, self.is_ntt = data, bool(is_ntt)


class Ciphertext:
    """A4: `size` in {2,3} RNS polynomials, tensor [..., size, L, N] + is_ntt."""

    def __init__(self, data: torch.Tensor, is_ntt: bool = False):
        if data.dim() < 3 or data.shape[-3] not in (2, 3):
            raise _cabi.DpfheError(2000, "Ciphertext tensor must be [..., 2|3, L, N]")
        self.data, self.is_ntt = data, bool(is_ntt)

    @property
    def size(self) -> int:
        return self.data.shape[-3]

    @property
    def batch(self) -> int:
        return int(np.prod(self.data.shape[:-3])) if self.data.dim() > 3 else 1


class Evaluator:
    """Operator API over one Context.  Every method only enqueues work on `stream` (default: current)."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self._lib = ctx._lib

    # ---- streams and stream-ordered temporaries ---------------------------------------------------------
    def _sp(self, stream) -> int:
        return _resolve_stream(stream, self.ctx.device).cuda_stream

    def _on(self, stream):
        """Context manager: torch work and allocations inside it belong to `stream` on the context's device, so the caching
        allocator never hands a temporary to another stream while the kernels enqueued here still use it."""
        return torch.cuda.stream(_resolve_stream(stream, self.ctx.device))

    def _empty(self, shape, stream) -> torch.Tensor:
        with self._on(stream):
            return torch.empty(tuple(shape), dtype=torch.int64, device=self.ctx.device)

    def _empty_like(self, t: torch.Tensor, stream) -> torch.Tensor:
        return self._empty(t.shape, stream)

    # ---- checks -------------------------------------------------------------------------------------
    def _chk(self, *tensors):
        p = self.ctx.params
        for t in tensors:
            if t.dtype != torch.int64 or not t.is_cuda or not t.is_contiguous():
                raise _cabi.DpfheError(2000, "buffers must be contiguous int64 (u64 bits) CUDA tensors")
            if t.device != self.ctx.device:
                raise _cabi.DpfheError(2000, f"buffer on {t.device}, context on {self.ctx.device}")
            if t.dim() < 2 or t.shape[-1] != p.n or t.shape[-2] != p.n_limbs:
                raise _cabi.DpfheError(2000, f"trailing dims must be [L={p.n_limbs}][N={p.n}]")

    def _npolys(self, t) -> int:
        return t.numel() // self.ctx.params.words_per_rns_poly()

    # ---- A1 / A2 --------------------------------------------------------------------------------------
    def ntt_forward_(self, t: torch.Tensor, stream=None) -> torch.Tensor:
        self._chk(t)
        _cabi.check(self._lib.dpfhe_ntt_fwd(self.ctx.handle, t.data_ptr(), self._npolys(t), self._sp(stream)), "dpfhe_ntt_fwd")
        return t

    def ntt_inverse_(self, t: torch.Tensor, stream=None) -> torch.Tensor:
        self._chk(t)
        _cabi.check(self._lib.dpfhe_ntt_inv(self.ctx.handle, t.data_ptr(), self._npolys(t), self._sp(stream)), "dpfhe_ntt_inv")
        return t

    def ntt_forward(self, t: torch.Tensor, out: torch.Tensor | None = None, stream=None) -> torch.Tensor:
        out = self._empty_like(t, stream) if out is None else out
        self._chk(t, out)
        _cabi.check(self._lib.dpfhe_ntt_fwd_oop(self.ctx.handle, out.data_ptr(), t.data_ptr(), self._npolys(t), self._sp(stream)), "dpfhe_ntt_fwd_oop")
        return out

    def ntt_inverse(self, t: torch.Tensor, out: torch.Tensor | None = None, stream=None) -> torch.Tensor:
        out = self._empty_like(t, stream) if out is None else out
        self._chk(t, out)
        _cabi.check(self._lib.dpfhe_ntt_inv_oop(self.ctx.handle, out.data_ptr(), t.data_ptr(), self._npolys(t), self._sp(stream)), "dpfhe_ntt_inv_oop")
        return out

    def transform_to_ntt_(self, x, stream=None):
        if not x.is_ntt:
            self.ntt_forward_(x.data, stream)
            x.is_ntt = True
        return x

    def transform_from_ntt_(self, x, stream=None):
        if x.is_ntt:
            self.ntt_inverse_(x.data, stream)
            x.is_ntt = False
        return x

    # ---- A3 -------------------------------------------------------------------------------------------
    def _dy(self, fn, name, out, a, b, stream):
        self._chk(out, a, *(() if b is None else (b,)))
        if a.shape != out.shape or (b is not None and b.shape != a.shape):
            raise _cabi.DpfheError(2000, "operand shapes differ")
        args = (self.ctx.handle, out.data_ptr(), a.data_ptr()) + (() if b is None else (b.data_ptr(),))
        _cabi.check(fn(*args, self._npolys(a), self._sp(stream)), name)
        return out

    def dyadic_mul(self, a, b, out=None, stream=None):
        return self._dy(self._lib.dpfhe_dyadic_mul, "dpfhe_dyadic_mul", self._empty_like(a, stream) if out is None else out, a, b, stream)

    def dyadic_mul_add_(self, acc, a, b, stream=None):
        return self._dy(self._lib.dpfhe_dyadic_mul_add, "dpfhe_dyadic_mul_add", acc, a, b, stream)

    def add_words(self, a, b, out=None, stream=None):
        return self._dy(self._lib.dpfhe_add, "dpfhe_add", self._empty_like(a, stream) if out is None else out, a, b, stream)

    def sub_words(self, a, b, out=None, stream=None):
        return self._dy(self._lib.dpfhe_sub, "dpfhe_sub", self._empty_like(a, stream) if out is None else out, a, b, stream)

    def negate_words(self, a, out=None, stream=None):
        return self._dy(self._lib.dpfhe_negate, "dpfhe_negate", self._empty_like(a, stream) if out is None else out, a, None, stream)

    # ---- A8: ciphertext add/sub/negate/reduce --------------------------------------------------------------
    def add(self, a: Ciphertext, b: Ciphertext, stream=None) -> Ciphertext:
        self._same_domain(a, b)
        return Ciphertext(self.add_words(a.data, b.data, stream=stream), a.is_ntt)

    def sub(self, a: Ciphertext, b: Ciphertext, stream=None) -> Ciphertext:
        self._same_domain(a, b)
        return Ciphertext(self.sub_words(a.data, b.data, stream=stream), a.is_ntt)

    def negate(self, a: Ciphertext, stream=None) -> Ciphertext:
        return Ciphertext(self.negate_words(a.data, stream=stream), a.is_ntt)

    def canonicalize_sum_(self, t: torch.Tensor, stream=None) -> torch.Tensor:
        """in place: words that are sums of at most 15 canonical residues -> canonical residues (dpfhe_canonicalize_sum: the pass after an all-reduce of partials)"""
        self._chk(t)
        _cabi.check(self._lib.dpfhe_canonicalize_sum(self.ctx.handle, t.data_ptr(), self._npolys(t), self._sp(stream)), "dpfhe_canonicalize_sum")
        return t

    def reduce_sum(self, cts: Ciphertext, out: torch.Tensor | None = None, stream=None) -> Ciphertext:
        """Modular sum over the batch dimension(s) -> one ciphertext (the shard-local partial)."""
        self._chk(cts.data)
        comps = cts.size
        if out is None:
            out = self._empty((comps, self.ctx.params.n_limbs, self.ctx.params.n), stream)
        self._chk(out)
        _cabi.check(self._lib.dpfhe_reduce_sum(self.ctx.handle, out.data_ptr(), cts.data.data_ptr(), cts.batch, comps, self._sp(stream)), "dpfhe_reduce_sum")
        return Ciphertext(out, cts.is_ntt)

    @staticmethod
    def _same_domain(a, b):
        if a.is_ntt != b.is_ntt:
            raise _cabi.DpfheError(2002, "operands are in different domains (is_ntt mismatch)")
        if a.data.shape != b.data.shape:
            raise _cabi.DpfheError(2000, "operand shapes differ")

    # ---- A6: THE METRIC OP -----------------------------------------------------------------------------
    def multiply(self, a: Ciphertext, b: Ciphertext, out: torch.Tensor | None = None, out_ntt: bool | None = None, stream=None) -> Ciphertext:
        """(a0,a1) (x) (b0,b1) -> (a0 b0, a0 b1 + a1 b0, a1 b1); no relinearisation.  Batched over leading dims."""
        self._same_domain(a, b)
        if a.size != 2 or b.size != 2:
            raise _cabi.DpfheError(2000, "multiply expects 2-component ciphertexts")
        self._chk(a.data, b.data)
        out_ntt = a.is_ntt if out_ntt is None else bool(out_ntt)
        lead = a.data.shape[:-3]
        if out is None:
            out = self._empty(tuple(lead) + (3, self.ctx.params.n_limbs, self.ctx.params.n), stream)
        self._chk(out)
        flags = (_cabi.IN_NTT if a.is_ntt else 0) | (_cabi.OUT_NTT if out_ntt else 0)
        _cabi.check(self._lib.dpfhe_ct_mul(self.ctx.handle, out.data_ptr(), a.data.data_ptr(), b.data.data_ptr(), a.batch, flags, self._sp(stream)), "dpfhe_ct_mul")
        return Ciphertext(out, out_ntt)

    def matvec_scalar(self, w: torch.Tensor, x: Ciphertext, out: torch.Tensor | None = None, stream=None) -> Ciphertext:
        """y_i = sum_j w_ij * x_j with scalar weights w: [rows][cols][L] residues (either domain)."""
        p = self.ctx.params
        self._chk(x.data)
        if (w.dtype != torch.int64 or not w.is_cuda or not w.is_contiguous() or w.dim() != 3 or w.shape[2] != p.n_limbs
                or x.data.dim() != 4 or x.size != 2 or w.shape[1] != x.data.shape[0]):
            raise _cabi.DpfheError(2000, "matvec_scalar: w [rows][cols][L] int64 cuda, x [cols][2][L][N]")
        rows, cols = w.shape[0], w.shape[1]
        if out is None:
            out = self._empty((rows, 2, self.ctx.params.n_limbs, self.ctx.params.n), stream)
        self._chk(out)
        _cabi.check(self._lib.dpfhe_matvec_scalar(self.ctx.handle, out.data_ptr(), w.data_ptr(), x.data.data_ptr(), rows, cols, self._sp(stream)), "dpfhe_matvec_scalar")
        return Ciphertext(out, x.is_ntt)

    # ---- N1: relinearisation ---------------------------------------------------------------
    def relinearize(self, ct3: Ciphertext, evk: torch.Tensor, out: torch.Tensor | None = None, stream=None) -> Ciphertext:
        """3 -> 2 components.  evk: [L digits][2][L][N], NTT domain (see include/dpfhe.h); ct3 coefficient domain."""
        if ct3.size != 3 or ct3.is_ntt:
            raise _cabi.DpfheError(2002 if ct3.is_ntt else 2000, "relinearize expects a 3-component coefficient-domain ciphertext")
        p = self.ctx.params
        self._chk(ct3.data, evk)
        if tuple(evk.shape) != (p.n_limbs, 2, p.n_limbs, p.n):
            raise _cabi.DpfheError(2000, "evk must be [L][2][L][N]")
        lead = ct3.data.shape[:-3]
        if out is None:
            out = self._empty(tuple(lead) + (2, self.ctx.params.n_limbs, self.ctx.params.n), stream)
        self._chk(out)
        _cabi.check(self._lib.dpfhe_relinearize(self.ctx.handle, out.data_ptr(), ct3.data.data_ptr(), evk.data_ptr(), ct3.batch, self._sp(stream)), "dpfhe_relinearize")
        return Ciphertext(out, False)

    def keyswitch_hybrid(self, ct: Ciphertext, key: torch.Tensor, stream=None) -> Ciphertext:
        """Hybrid key switching; THIS evaluator's context is the extended one (last limb = special prime P) while `ct`
        lives on the first L-1 limbs.  3 components: relinearisation; 2 components: key switch after an automorphism.
        key: [L-1][2][L][N] NTT domain, key_j = (-(a_j s) + e_j + P g_j T, a_j)."""
        p = self.ctx.params
        L, Ld, n = p.n_limbs, p.n_limbs - 1, p.n
        d = ct.data
        if ct.is_ntt or ct.size not in (2, 3) or d.dim() != 4 or d.shape[-2] != Ld or d.shape[-1] != n or d.dtype != torch.int64 or not d.is_contiguous():
            raise _cabi.DpfheError(2000, "keyswitch_hybrid: coefficient-domain [batch][2|3][L-1][N] ciphertexts on the extended context")
        if tuple(key.shape) != (Ld, 2, L, n):
            raise _cabi.DpfheError(2000, "key must be [L-1][2][L][N]")
        batch = d.shape[0]
        out = self._empty((batch, 2, Ld, n), stream)
        work = self._empty((batch, 2, L, n), stream)
        fn = self._lib.dpfhe_relinearize_hybrid if ct.size == 3 else self._lib.dpfhe_switch_key_hybrid
        _cabi.check(fn(self.ctx.handle, out.data_ptr(), d.data_ptr(), key.data_ptr(), work.data_ptr(), batch, self._sp(stream)), "dpfhe_*_hybrid")
        return Ciphertext(out, False)

    def rotate_hybrid_batch(self, ct: Ciphertext, galois_elts, keys: torch.Tensor, stream=None) -> Ciphertext:
        """N3, batched rotations on the extended context: output item i = key-switched sigma_{galois_elts[i]} of input item
        i, or of THE input item when ct holds one.  keys: [len(elts)][L-1][2][L][N] (key i switches sigma_{g_i}(s) -> s)."""
        import ctypes as C
        p = self.ctx.params
        L, Ld, n = p.n_limbs, p.n_limbs - 1, p.n
        d = ct.data
        k = len(galois_elts)
        if ct.is_ntt or ct.size != 2 or d.dim() != 4 or d.shape[-2] != Ld or d.shape[-1] != n or d.dtype != torch.int64 or not d.is_contiguous() or d.shape[0] not in (1, k):
            raise _cabi.DpfheError(2000, "rotate_hybrid_batch: coefficient-domain [1 or k][2][L-1][N] ciphertexts on the extended context")
        if tuple(keys.shape) != (k, Ld, 2, L, n) or keys.dtype != torch.int64 or not keys.is_contiguous():
            raise _cabi.DpfheError(2000, "keys must be [k][L-1][2][L][N]")
        out = self._empty((k, 2, Ld, n), stream)
        work = self._empty((k, 2, L, n), stream)
        rotated = self._empty((k, 2, Ld, n), stream)
        elts = (C.c_uint32 * k)(*[int(g) for g in galois_elts])
        _cabi.check(self._lib.dpfhe_rotate_hybrid_batch(self.ctx.handle, out.data_ptr(), d.data_ptr(), d.shape[0], elts, keys.data_ptr(), work.data_ptr(),
                                                        rotated.data_ptr(), k, self._sp(stream)), "dpfhe_rotate_hybrid_batch")
        return Ciphertext(out, False)

    def rotate_hybrid_hoisted(self, ct: Ciphertext, galois_elts, keys: torch.Tensor, stream=None) -> Ciphertext:
        """N3, hoisted rotations on the extended context: output item r * T + t = key-switched sigma_{galois_elts[r]} of input item t
        (rotation-major), the digit decomposition of each input and its forward transforms shared by all rotations.
        keys: [len(elts)][L-1][2][L][N]."""
        p = self.ctx.params
        L, Ld, n = p.n_limbs, p.n_limbs - 1, p.n
        d = ct.data
        k = len(galois_elts)
        if ct.is_ntt or ct.size != 2 or d.dim() != 4 or d.shape[-2] != Ld or d.shape[-1] != n or d.dtype != torch.int64 or not d.is_contiguous():
            raise _cabi.DpfheError(2000, "rotate_hybrid_hoisted: coefficient-domain [T][2][L-1][N] ciphertexts on the extended context")
        T = d.shape[0]
        if tuple(keys.shape) != (k, Ld, 2, L, n) or keys.dtype != torch.int64 or not keys.is_contiguous():
            raise _cabi.DpfheError(2000, "keys must be [k][L-1][2][L][N]")
        out = self._empty((k * T, 2, Ld, n), stream)
        work = self._empty((k * T, 2, L, n), stream)
        rotated0 = self._empty((k * T, Ld, n), stream)
        digits = self._empty((T, Ld, L, n), stream)
        elts = (C.c_uint32 * k)(*[int(g) for g in galois_elts])
        _cabi.check(self._lib.dpfhe_rotate_hybrid_hoisted(self.ctx.handle, out.data_ptr(), d.data_ptr(), T, elts, keys.data_ptr(), work.data_ptr(), rotated0.data_ptr(),
                                                          digits.data_ptr(), k, self._sp(stream)), "dpfhe_rotate_hybrid_hoisted")
        return Ciphertext(out, False)

    def rotate_hybrid_grouped(self, ct: Ciphertext, galois_elts, group: int, keys: torch.Tensor, stream=None) -> Ciphertext:
        """N3, grouped rotations: ct holds len(elts) * group items, item i is rotated by galois_elts[i // group] with key i // group."""
        p = self.ctx.params
        L, Ld, n = p.n_limbs, p.n_limbs - 1, p.n
        d = ct.data
        k = len(galois_elts)
        if ct.is_ntt or ct.size != 2 or d.dim() != 4 or d.shape[0] != k * group or d.shape[-2] != Ld or d.shape[-1] != n or d.dtype != torch.int64 or not d.is_contiguous():
            raise _cabi.DpfheError(2000, "rotate_hybrid_grouped: coefficient-domain [k * group][2][L-1][N] ciphertexts on the extended context")
        if tuple(keys.shape) != (k, Ld, 2, L, n) or keys.dtype != torch.int64 or not keys.is_contiguous():
            raise _cabi.DpfheError(2000, "keys must be [k][L-1][2][L][N]")
        out = self._empty((k * group, 2, Ld, n), stream)
        work = self._empty((k * group, 2, L, n), stream)
        rotated = self._empty((k * group, 2, Ld, n), stream)
        elts = (C.c_uint32 * k)(*[int(g) for g in galois_elts])
        _cabi.check(self._lib.dpfhe_rotate_hybrid_grouped(self.ctx.handle, out.data_ptr(), d.data_ptr(), elts, k, group, keys.data_ptr(), work.data_ptr(),
                                                          rotated.data_ptr(), self._sp(stream)), "dpfhe_rotate_hybrid_grouped")
        return Ciphertext(out, False)

    def device_copy(self, dst: torch.Tensor, src: torch.Tensor, stream=None) -> torch.Tensor:
        """Plain device-to-device copy through the library's streaming access shape (dpfhe_copy): the practical HBM ceiling."""
        if dst.numel() != src.numel() or dst.dtype != torch.int64 or src.dtype != torch.int64 or not dst.is_contiguous() or not src.is_contiguous():
            raise _cabi.DpfheError(2000, "device_copy: two contiguous int64 tensors of the same size")
        _cabi.check(self._lib.dpfhe_copy(self.ctx.handle, dst.data_ptr(), src.data_ptr(), src.numel(), self._sp(stream)), "dpfhe_copy")
        return dst

    # ---- N3, round 3: baby-step / giant-step with the division by P deferred (include/dpfhe.h) ---------------------------
    def rotate_hoisted_qp(self, ct: Ciphertext, galois_elts, keys: torch.Tensor, stream=None) -> torch.Tensor:
        """[T][2][L-1][N] coefficient-domain inputs on the extended context -> [1 + k][T][2][L][N], NTT domain over Q P:
        block 0 = P * ct, block 1 + r = P * sigma_{g_r}(ct) + its key-switching term (not yet divided by P)."""
        p = self.ctx.params
        L, Ld, n = p.n_limbs, p.n_limbs - 1, p.n
        d = ct.data
        k = len(galois_elts)
        if ct.is_ntt or ct.size != 2 or d.dim() != 4 or d.shape[-2] != Ld or d.shape[-1] != n or d.dtype != torch.int64 or not d.is_contiguous():
            raise _cabi.DpfheError(2000, "rotate_hoisted_qp: coefficient-domain [T][2][L-1][N] ciphertexts on the extended context")
        T = d.shape[0]
        if k and (tuple(keys.shape) != (k, Ld, 2, L, n) or keys.dtype != torch.int64 or not keys.is_contiguous()):
            raise _cabi.DpfheError(2000, "keys must be [k][L-1][2][L][N]")
        out = self._empty((k + 1, T, 2, L, n), stream)
        in_ntt = self._empty((T, 2, Ld, n), stream)
        digits = self._empty((T, Ld, L, n), stream)
        elts = (C.c_uint32 * max(k, 1))(*[int(g) for g in galois_elts])
        _cabi.check(self._lib.dpfhe_rotate_hoisted_qp(self.ctx.handle, out.data_ptr(), d.data_ptr(), T, elts, keys.data_ptr() if k else None, in_ntt.data_ptr(),
                                                      digits.data_ptr(), k, self._sp(stream)), "dpfhe_rotate_hoisted_qp")
        return out

    def ntt_inverse_galois(self, t: torch.Tensor, galois_elts, out: torch.Tensor | None = None, stream=None) -> torch.Tensor:
        """t: [len(elts)][...][L][N] NTT domain -> sigma_{g_e}(INTT(t[e])) (the automorphism as a gather in the NTT domain)."""
        self._chk(t)
        k = len(galois_elts)
        if t.shape[0] != k:
            raise _cabi.DpfheError(2000, "ntt_inverse_galois: leading dimension = number of elements")
        out = self._empty_like(t, stream) if out is None else out
        elts = (C.c_uint32 * k)(*[int(g) for g in galois_elts])
        _cabi.check(self._lib.dpfhe_ntt_inv_galois(self.ctx.handle, out.data_ptr(), t.data_ptr(), self._npolys(t) // k, elts, k, self._sp(stream)), "dpfhe_ntt_inv_galois")
        return out

    def switch_key_qp(self, ct: Ciphertext, keys: torch.Tensor, group: int, stream=None) -> torch.Tensor:
        """[k * group][2][L-1][N] coefficient-domain items (item i with key i // group) -> [k * group][2][L][N]: the key inner
        products sum_j NTT(lift([c1]_{q_j})) (.) key_j in the NTT domain over Q P, nothing added, not divided by P."""
        p = self.ctx.params
        L, Ld, n = p.n_limbs, p.n_limbs - 1, p.n
        d = ct.data
        k = keys.shape[0]
        if ct.is_ntt or ct.size != 2 or d.dim() != 4 or d.shape[0] != k * group or d.shape[-2] != Ld or d.shape[-1] != n or d.dtype != torch.int64 or not d.is_contiguous():
            raise _cabi.DpfheError(2000, "switch_key_qp: coefficient-domain [k * group][2][L-1][N] ciphertexts on the extended context")
        if tuple(keys.shape) != (k, Ld, 2, L, n) or keys.dtype != torch.int64 or not keys.is_contiguous():
            raise _cabi.DpfheError(2000, "keys must be [k][L-1][2][L][N]")
        out = self._empty((k * group, 2, L, n), stream)
        _cabi.check(self._lib.dpfhe_switch_key_qp(self.ctx.handle, out.data_ptr(), d.data_ptr(), keys.data_ptr(), k, group, self._sp(stream)), "dpfhe_switch_key_qp")
        return out

    def rescale_bsgs(self, t_qp: torch.Tensor, addends: torch.Tensor, stream=None) -> torch.Tensor:
        """t_qp: [batch][2][L][N] (coefficient domain over Q P), addends: [n_add][batch][2][L-1][N] -> [batch][2][L-1][N] =
        round(t / P) + (sum over ALL addends of component 0, component 1 of addend 0)."""
        self._chk(t_qp)
        p = self.ctx.params
        L, Ld, n = p.n_limbs, p.n_limbs - 1, p.n
        batch = t_qp.shape[0]
        if t_qp.dim() != 4 or t_qp.shape[1] != 2 or addends.dim() != 5 or tuple(addends.shape[1:]) != (batch, 2, Ld, n) or not addends.is_contiguous():
            raise _cabi.DpfheError(2000, "rescale_bsgs: t [batch][2][L][N], addends [n_add][batch][2][L-1][N]")
        out = self._empty((batch, 2, Ld, n), stream)
        _cabi.check(self._lib.dpfhe_rescale_bsgs(self.ctx.handle, out.data_ptr(), t_qp.data_ptr(), addends.data_ptr(), addends.shape[0], batch, self._sp(stream)),
                    "dpfhe_rescale_bsgs")
        return out

    def matvec_plain_multi(self, W: Plaintext, x: torch.Tensor, n_rhs: int, stream=None) -> torch.Tensor:
        """y[i][t] = sum_j W[i][j] (.) x[j][t].  W.data: [rows][cols][L][N] (NTT), x: [cols][n_rhs][2][L][N] (NTT) -> [rows][n_rhs][2][L][N]."""
        self._chk(W.data, x)
        if W.data.dim() != 4 or x.dim() != 5 or x.shape[1] != n_rhs or x.shape[2] != 2 or W.data.shape[1] != x.shape[0]:
            raise _cabi.DpfheError(2000, "matvec_plain_multi: W [rows][cols][L][N], x [cols][n_rhs][2][L][N]")
        rows, cols = W.data.shape[0], W.data.shape[1]
        out = self._empty((rows, n_rhs, 2, self.ctx.params.n_limbs, self.ctx.params.n), stream)
        _cabi.check(self._lib.dpfhe_matvec_plain_multi(self.ctx.handle, out.data_ptr(), W.data.data_ptr(), x.data_ptr(), rows, cols, n_rhs, self._sp(stream)),
                    "dpfhe_matvec_plain_multi")
        return out

    def rescale_words(self, t: torch.Tensor, stream=None) -> torch.Tensor:
        """[..., L, N] -> [..., L-1, N]: round(x / q_last) limb by limb (coefficient domain).  The result belongs to the
        context of the first L-1 moduli."""
        self._chk(t)
        p = self.ctx.params
        out = self._empty(tuple(t.shape[:-2]) + (p.n_limbs - 1, p.n), stream)
        _cabi.check(self._lib.dpfhe_rescale(self.ctx.handle, out.data_ptr(), t.data_ptr(), self._npolys(t), self._sp(stream)), "dpfhe_rescale")
        return out

    # ---- round 4: exact base extension, scale-and-round, and the exact (BFV-style) multiply built on them ----------------------
    def base_extend(self, t: torch.Tensor, src_limb0: int, dst_limb0: int, n_dst: int, stream=None) -> torch.Tensor:
        """t: [..., n_src, N] residues modulo this context's limbs src_limb0 .. -> [..., n_dst, N] residues of the SAME centred integers
        modulo limbs dst_limb0 ..  (exact; include/dpfhe.h dpfhe_base_extend)."""
        p = self.ctx.params
        if t.dtype != torch.int64 or not t.is_contiguous() or t.device != self.ctx.device or t.dim() < 2 or t.shape[-1] != p.n:
            raise _cabi.DpfheError(2000, "base_extend: contiguous int64 [..., n_src, N] on the context's device")
        ns = t.shape[-2]
        out = self._empty(tuple(t.shape[:-2]) + (n_dst, p.n), stream)
        npolys = t.numel() // (ns * p.n)
        _cabi.check(self._lib.dpfhe_base_extend(self.ctx.handle, out.data_ptr(), n_dst, t.data_ptr(), ns, src_limb0, ns, dst_limb0, n_dst, npolys, self._sp(stream)), "dpfhe_base_extend")
        return out

    def scale_round(self, t: torch.Tensor, drop_limb0: int, n_drop: int, keep_limb0: int, n_keep: int, multiplier: int, stream=None) -> torch.Tensor:
        """t: [..., L, N] on all limbs -> [..., n_keep, N]: round(multiplier * X / (product of the dropped limbs)) on the kept limbs."""
        self._chk(t)
        p = self.ctx.params
        out = self._empty(tuple(t.shape[:-2]) + (n_keep, p.n), stream)
        _cabi.check(self._lib.dpfhe_scale_round(self.ctx.handle, out.data_ptr(), n_keep, t.data_ptr(), drop_limb0, n_drop, keep_limb0, n_keep, multiplier, self._npolys(t),
                                                self._sp(stream)), "dpfhe_scale_round")
        return out

    def multiply_exact(self, a: torch.Tensor, b: torch.Tensor, level_limbs: int, plain_modulus: int, stream=None) -> torch.Tensor:
        """EXACT ciphertext x ciphertext multiply of BFV-style ciphertexts (scale floor(q / t), Encryptor::encrypt_exact) whose modulus q is the
        product of this context's FIRST `level_limbs` limbs: a, b [batch][2][level_limbs][N] -> [batch][3][level_limbs][N], the tensor product
        over the integers scaled by t / q and rounded, reduced mod q.  The context's remaining limbs are the workspace: it needs
        Q > 2 N t q^2 (N = 8192, t = 65537, two 60-bit level limbs: five limbs).  Steps: dpfhe_base_extend (both operands to all limbs), the
        fused dpfhe_ct_mul (THE METRIC OP) on all limbs, dpfhe_scale_round (x t / q on the workspace limbs), dpfhe_base_extend back."""
        p = self.ctx.params
        L, ll = p.n_limbs, level_limbs
        # the limits of dpfhe_base_extend / dpfhe_scale_round (base_ext.h kBxMaxSrc = 10, kBxMaxDst = 20, kBxMaxSrcGeneric = 8), checked BEFORE anything is
        # enqueued (ExactMultiplier's constructor in fhe_api.cpp makes the same checks)
        max_src = 9 if self.ctx.uses_fold else 8
        if a.shape != b.shape or a.dim() != 4 or a.shape[1] != 2 or a.shape[2] != ll or not (0 < ll < L and ll <= max_src and L <= 20 and L - ll <= (10 if self.ctx.uses_fold else 8)):
            raise _cabi.DpfheError(2000, "multiply_exact: [batch][2][level_limbs][N] operands, 0 < level_limbs <= min(9, L - 1) (8 with generic primes), L <= 20, L - level_limbs <= 10 (8)")
        # workspace check (exactness): log2 Q - 1 > log2(N) + log2(t) + 2 log2(q) + 1
        import math
        lq = sum(math.log2(m) for m in p.moduli[:ll])
        if sum(math.log2(m) for m in p.moduli) - 1 <= math.log2(p.n) + math.log2(plain_modulus) + 2 * lq + 1:
            raise _cabi.DpfheError(2002, "multiply_exact: the context's limbs cannot hold the integer tensor product of this level")
        if sum(math.log2(m) for m in p.moduli[ll:]) - 1 <= math.log2(p.n) + math.log2(plain_modulus) + lq + 2:
            raise _cabi.DpfheError(2002, "multiply_exact: the workspace limbs cannot hold the scaled product")
        A = self.base_extend(a, 0, 0, L, stream=stream)
        B = A if b is a else self.base_extend(b, 0, 0, L, stream=stream)
        T = self.multiply(Ciphertext(A), Ciphertext(B), stream=stream).data                       # [batch][3][L][N]
        W = self.scale_round(T, 0, ll, ll, L - ll, plain_modulus, stream=stream)                    # [batch][3][L - ll][N]
        return self.base_extend(W, ll, 0, ll, stream=stream)                                        # [batch][3][ll][N]

    # ---- N3: Galois automorphism + key switch ----------------------------------------------------
    def apply_galois_words(self, t: torch.Tensor, galois_elt: int, out: torch.Tensor | None = None, stream=None) -> torch.Tensor:
        """a(X) -> a(X^galois_elt) on every RNS polynomial of t (coefficient domain, out of place)."""
        out = self._empty_like(t, stream) if out is None else out
        self._chk(t, out)
        _cabi.check(self._lib.dpfhe_apply_galois(self.ctx.handle, out.data_ptr(), t.data_ptr(), self._npolys(t), int(galois_elt), self._sp(stream)), "dpfhe_apply_galois")
        return out

    def apply_galois(self, ct: Ciphertext, galois_elt: int, key: torch.Tensor, stream=None) -> Ciphertext:
        """ct (2 components, coefficient domain) encrypting m(X) under s  ->  ciphertext of m(X^g) under s.
        key: [L][2][L][N] NTT-domain switching key from sigma_g(s) to s."""
        if ct.size != 2 or ct.is_ntt:
            raise _cabi.DpfheError(2002 if ct.is_ntt else 2000, "apply_galois expects a 2-component coefficient-domain ciphertext")
        p = self.ctx.params
        self._chk(key)
        if tuple(key.shape) != (p.n_limbs, 2, p.n_limbs, p.n):
            raise _cabi.DpfheError(2000, "key must be [L][2][L][N]")
        rotated = self.apply_galois_words(ct.data, galois_elt, stream=stream)
        out = self._empty_like(rotated, stream)
        _cabi.check(self._lib.dpfhe_switch_key(self.ctx.handle, out.data_ptr(), rotated.data_ptr(), key.data_ptr(), ct.batch, self._sp(stream)), "dpfhe_switch_key")
        return Ciphertext(out, False)

    # ---- A7 -------------------------------------------------------------------------------------------
    def multiply_plain(self, a: Ciphertext, p: Plaintext, stream=None) -> Ciphertext:
        """ct (.) pt, both in the NTT domain: every component times the plaintext polynomial."""
        if not (a.is_ntt and p.is_ntt):
            raise _cabi.DpfheError(2002, "multiply_plain needs NTT-domain operands")
        pr = self.ctx.params
        if tuple(p.data.shape[-2:]) != (pr.n_limbs, pr.n) or p.data.numel() != pr.n_limbs * pr.n:
            raise _cabi.DpfheError(2000, "multiply_plain takes ONE plaintext ([L][N]); use dyadic_mul for per-item plaintexts")
        self._chk(a.data, p.data)
        out = self._empty_like(a.data, stream)
        _cabi.check(self._lib.dpfhe_multiply_plain(self.ctx.handle, out.data_ptr(), a.data.data_ptr(), p.data.data_ptr(), self._npolys(a.data), self._sp(stream)), "dpfhe_multiply_plain")
        return Ciphertext(out, True)

    def matvec_plain(self, W: Plaintext, x: Ciphertext, out: torch.Tensor | None = None, stream=None) -> Ciphertext:
        """y_i = sum_j W_ij (.) x_j.  W.data: [rows][cols][L][N] (NTT), x.data: [cols][2][L][N] (NTT)."""
        if not (W.is_ntt and x.is_ntt):
            raise _cabi.DpfheError(2002, "matvec_plain needs NTT-domain operands")
        self._chk(W.data, x.data)
        if W.data.dim() != 4 or x.data.dim() != 4 or x.size != 2 or W.data.shape[1] != x.data.shape[0]:
            raise _cabi.DpfheError(2000, "matvec_plain: W [rows][cols][L][N], x [cols][2][L][N]")
        rows, cols = W.data.shape[0], W.data.shape[1]
        if out is None:
            out = self._empty((rows, 2, self.ctx.params.n_limbs, self.ctx.params.n), stream)
        self._chk(out)
        _cabi.check(self._lib.dpfhe_matvec_plain(self.ctx.handle, out.data_ptr(), W.data.data_ptr(), x.data.data_ptr(), rows, cols, self._sp(stream)), "dpfhe_matvec_plain")
        return Ciphertext(out, True)
