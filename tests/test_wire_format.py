"""CPU: ciphertext wire format (N4) - Python twin round trip, header validation, and the exact byte layout the C++
facade writes (tests/cpp/test_fhe_api.cpp checks the C++ side against the same layout on the GPU)."""
import numpy as np
import pytest

from deeppowers_amd import wire
from deeppowers_amd.params import FheParams
from oracle.cbind import Oracle


def test_roundtrip_and_layout():
    p = FheParams.config1()
    w = Oracle.from_params(p).fill(6, 3).reshape(3, 2, 1, 1024)
    blob = wire.dumps(w, p, is_ntt=True)
    assert blob[:8] == b"DPFHEv1\0" and len(blob) == 40 + 8 * 1 + w.size * 8
    assert int.from_bytes(blob[8:12], "little") == 10 and int.from_bytes(blob[12:16], "little") == 1
    assert int.from_bytes(blob[16:24], "little") == 3 and int.from_bytes(blob[24:32], "little") == 2
    assert int.from_bytes(blob[40:48], "little") == p.moduli[0]
    got, is_ntt = wire.loads(blob, p)
    assert is_ntt and np.array_equal(got, w)


def test_rejects_mismatches():
    p = FheParams.config1()
    w = Oracle.from_params(p).fill(2, 4).reshape(1, 2, 1, 1024)
    blob = wire.dumps(w, p, False)
    with pytest.raises(ValueError):
        wire.loads(blob[:-8], p)
    with pytest.raises(ValueError):
        wire.loads(b"X" + blob[1:], p)
    with pytest.raises(ValueError):
        wire.loads(blob, FheParams.n4096_l4())
    bad = bytearray(blob); bad[-8:] = (p.moduli[0]).to_bytes(8, "little")   # a word == q is not canonical
    with pytest.raises(ValueError):
        wire.loads(bytes(bad), p)
    with pytest.raises(ValueError):
        wire.dumps(w + np.uint64(p.moduli[0]), p, False)
