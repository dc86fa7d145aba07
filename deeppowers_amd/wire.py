"""Ciphertext wire format - the Python twin of PolyBuffer::save/load in
include/deeppowers/fhe.hpp.  Little-endian: b"DPFHEv1\0", u32 log2_n, u32 n_limbs, u64 batch, u64 components,
u32 is_ntt, u32 reserved, u64 moduli[n_limbs], then the u64 words [batch][component][limb][N]."""
from __future__ import annotations

import struct

import numpy as np

from .params import FheParams

MAGIC = b"DPFHEv1\0"
_HDR = struct.Struct("<8sIIQQII")


def dumps(words: np.ndarray, params: FheParams, is_ntt: bool) -> bytes:
    """words: uint64 array shaped [batch][components][L][N] (canonical residues)."""
    a = np.ascontiguousarray(words, dtype="<u8")
    if a.ndim != 4 or a.shape[2] != params.n_limbs or a.shape[3] != params.n:
        raise ValueError("words must be [batch][components][L][N]")
    q = np.array(params.moduli, dtype=np.uint64)[None, None, :, None]
    if (a >= q).any():
        raise ValueError("non-canonical residue")
    hdr = _HDR.pack(MAGIC, params.log2_n, params.n_limbs, a.shape[0], a.shape[1], 1 if is_ntt else 0, 0)
    return hdr + np.array(params.moduli, dtype="<u8").tobytes() + a.tobytes()


def loads(blob: bytes, params: FheParams):
    """-> (words [batch][components][L][N] uint64, is_ntt).  Raises ValueError on any mismatch with `params`."""
    if len(blob) < _HDR.size:
        raise ValueError("truncated header")
    magic, log2_n, n_limbs, batch, comps, is_ntt, _ = _HDR.unpack_from(blob, 0)
    if magic != MAGIC:
        raise ValueError("not a DPFHEv1 stream")
    if log2_n != params.log2_n or n_limbs != params.n_limbs:
        raise ValueError("header does not match the parameters")
    off = _HDR.size
    moduli = np.frombuffer(blob, dtype="<u8", count=n_limbs, offset=off)
    if tuple(int(m) for m in moduli) != tuple(params.moduli):
        raise ValueError("moduli differ")
    off += 8 * n_limbs
    count = batch * comps * n_limbs * params.n
    if len(blob) != off + 8 * count:
        raise ValueError("payload size does not match the header")
    words = np.frombuffer(blob, dtype="<u8", count=count, offset=off).reshape(batch, comps, n_limbs, params.n).astype(np.uint64)
    if (words >= np.array(params.moduli, dtype=np.uint64)[None, None, :, None]).any():
        raise ValueError("non-canonical residue")
    return words, bool(is_ntt)
