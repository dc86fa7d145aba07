"""the Generate-style RPC carrying ciphertexts (deeppowers_amd/rpc.py, dpfhe_rpc.proto).

CPU: the .proto text and the run-time descriptors agree; a transport-only server (echo model, no device) exercises framing,
message sizes beyond gRPC's default, status mapping and the metrics counters over a real localhost channel.
GPU (-m gpu): a client that holds the secret key encrypts, the server evaluates with the HIP kernels, the client decrypts:
the encrypted linear layer (integer weights) and multiply + relinearise with keys registered for the session; both are also
compared bit for bit with the oracle on the same words."""
import os
import re

import grpc
import numpy as np
import pytest

from deeppowers_amd import rpc
from deeppowers_amd.params import FheParams
from test_rlwe_semantics import encrypt, keygen_relin, negacyclic_int, phase, small_params


def _proto_fields():
    text = open(os.path.join(os.path.dirname(rpc.__file__), "dpfhe_rpc.proto")).read()
    text = re.sub(r"//[^\n]*", "", text)
    msgs = {}
    for m in re.finditer(r"message\s+(\w+)\s*\{([^}]*)\}", text):
        msgs[m.group(1)] = [(f.group(2), int(f.group(3)), f.group(1)) for f in re.finditer(r"(\w+)\s+(\w+)\s*=\s*(\d+)\s*;", m.group(2))]
    rpcs = re.findall(r"rpc\s+(\w+)\s*\(\s*(\w+)\s*\)\s*returns\s*\(\s*(\w+)\s*\)", text)
    return msgs, rpcs


def test_proto_file_matches_runtime_descriptors():
    T = rpc._T
    names = {T.TYPE_STRING: "string", T.TYPE_BYTES: "bytes", T.TYPE_FLOAT: "float", T.TYPE_INT32: "int32", T.TYPE_BOOL: "bool"}
    msgs, rpcs = _proto_fields()
    assert set(msgs) == {n for n, _ in rpc.MESSAGES}
    for name, fields in rpc.MESSAGES:
        want = [(f, num, tn if t == T.TYPE_MESSAGE else names[t]) for f, num, t, tn in fields]
        assert msgs[name] == want, name
    assert [tuple(r) for r in rpcs] == [tuple(m) for m in rpc.METHODS]
    # and the classes really serialise what they declare
    req = rpc.pb["EncryptedGenerateRequest"](request_id="r1", model="m", ciphertext=b"\x01\x02", session_id="s")
    back = rpc.pb["EncryptedGenerateRequest"].FromString(req.SerializeToString())
    assert (back.request_id, back.model, back.ciphertext, back.session_id, back.ciphertext_b) == ("r1", "m", b"\x01\x02", "s", b"")


def _canonical(rng, p, batch, comps):
    q = np.array(p.moduli, dtype=np.uint64)[None, None, :, None]
    return rng.integers(0, 2**62, (batch, comps, p.n_limbs, p.n), dtype=np.uint64) % q


def test_transport_echo_errors_and_metrics():
    p = FheParams.n4096_l4()
    server = rpc.EncryptedInferenceServer(None, params=p)
    server.register_model("echo", rpc.Passthrough())
    server.register_model("needs_device", rpc.MultiplyRelinearize(relinearize=False))
    port = server.start("127.0.0.1:0")
    client = rpc.EncryptedClient(f"127.0.0.1:{port}", p)
    try:
        rng = np.random.default_rng(3)
        x = _canonical(rng, p, 24, 2)                      # 6 MiB: above gRPC's 4 MiB default in both directions
        y, is_ntt = client.generate("echo", x, is_ntt=True)
        assert is_ntt and np.array_equal(x, y)
        assert client.last_response.server_ms > 0
        with pytest.raises(grpc.RpcError) as e:
            client.generate("no_such_model", x[:1])
        assert e.value.code() == grpc.StatusCode.NOT_FOUND
        # a stream for other parameters is the client's mistake
        other = FheParams.config1()
        bad = rpc.EncryptedClient(f"127.0.0.1:{port}", other)
        with pytest.raises(grpc.RpcError) as e:
            bad.generate("echo", _canonical(rng, other, 1, 2))
        assert e.value.code() == grpc.StatusCode.INVALID_ARGUMENT and "header" in e.value.details()
        bad.close()
        # truncated payload
        call = client._channel.unary_unary(rpc.method_path("EncryptedGenerate"), request_serializer=rpc.pb["EncryptedGenerateRequest"].SerializeToString,
                                           response_deserializer=rpc.pb["EncryptedGenerateResponse"].FromString)
        with pytest.raises(grpc.RpcError) as e:
            call(rpc.pb["EncryptedGenerateRequest"](request_id="x", model="echo", ciphertext=b"DPFHEv1\0abc"))
        assert e.value.code() == grpc.StatusCode.INVALID_ARGUMENT
        # no device behind this server: a compute model must fail loudly, as INTERNAL (there is no CPU fallback)
        with pytest.raises(grpc.RpcError) as e:
            client.generate("needs_device", x[:1], x[:1])
        assert e.value.code() == grpc.StatusCode.INTERNAL and "no device context" in e.value.details()
        # key registration validates shape and domain
        with pytest.raises(grpc.RpcError) as e:
            client.register_relin_keys(_canonical(rng, p, 2, 2))
        assert e.value.code() == grpc.StatusCode.INVALID_ARGUMENT
        assert client.register_relin_keys(_canonical(rng, p, p.n_limbs, 2))
        # sessions are bounded: the least recently used one is dropped
        server.max_sessions = 2
        for sid in ("s1", "s2", "s3"):
            c2 = rpc.EncryptedClient(f"127.0.0.1:{port}", p, session_id=sid)
            k3 = _canonical(rng, p, p.n_limbs, 2)
            assert c2.register_relin_keys(k3)
            c2.close()
        assert list(server.sessions) == ["s2", "s3"] and set(server._key_digests) == {"s2", "s3"}
        # a retry with the SAME keys (a deadline whose first call did land) is idempotent: ok, no new upload, no error counted
        owner = rpc.EncryptedClient(f"127.0.0.1:{port}", p, session_id="s3")
        assert owner.register_relin_keys(k3)
        owner.close()
        # a session id is the bearer token of its keys: nobody can replace the keys registered under an existing id
        thief = rpc.EncryptedClient(f"127.0.0.1:{port}", p, session_id="s3")
        with pytest.raises(grpc.RpcError) as e:
            thief.register_relin_keys(_canonical(rng, p, p.n_limbs, 2))
        assert e.value.code() == grpc.StatusCode.ALREADY_EXISTS
        thief.close()
        # message sizes are bounded by the parameters and the server's maximum batch
        assert rpc.message_limit(p, server.max_batch) < 2**31 and rpc.message_limit(p, 1) > 3 * p.n_limbs * p.n * 8
        assert rpc.message_limit(p, 1) > 2 * p.n_limbs * p.n_limbs * p.n * 8      # a small max_batch still admits one set of relinearisation keys
        m = client.metrics()
        assert m.total_requests == 7 and m.errors.total_errors == 6 and m.errors.internal_errors == 1 and m.errors.invalid_argument_errors == 5
        assert m.latency.p50_ms > 0 and m.throughput.ciphertexts_per_second > 0
    finally:
        client.close()
        server.stop()


@pytest.mark.gpu
def test_encrypted_linear_layer_over_rpc():
    from deeppowers_amd.evaluator import Context
    from oracle.cbind import Oracle
    p = small_params()
    rng = np.random.default_rng(11)
    s = rng.integers(-1, 2, p.n)
    rows, cols, delta = 5, 3, 1 << 40
    W = rng.integers(-50, 51, (rows, cols))
    msgs = rng.integers(0, 1000, (cols, p.n))            # feature j of n samples (one per coefficient)
    cts, phases = zip(*(encrypt(rng, p, s, msgs[j], delta) for j in range(cols)))
    x = np.stack(cts)
    ctx = Context(p, 0)
    server = rpc.EncryptedInferenceServer(ctx)
    server.register_model("linear", rpc.ScalarLinear(ctx, W))
    port = server.start("127.0.0.1:0")
    client = rpc.EncryptedClient(f"127.0.0.1:{port}", p)
    try:
        y, is_ntt = client.generate("linear", x)
        assert not is_ntt and y.shape == (rows, 2, p.n_limbs, p.n)
        assert client.last_response.device_ms > 0
        # bit-exact against the oracle's scalar matvec on the same words
        wres = np.array([[[int(v) % q for q in p.moduli] for v in row] for row in W.tolist()], dtype=np.uint64)
        assert np.array_equal(y, Oracle.from_params(p).matvec_scalar(wres, x, rows, cols))
        # and under decryption: phase(y_i) = sum_j w_ij phase(x_j) exactly, so round(phase / delta) = (W m)_i
        for i in range(rows):
            ph, Q = phase(p, y[i], s)
            want = [sum(int(W[i, j]) * phases[j][k] for j in range(cols)) % Q for k in range(p.n)]
            assert ph == want
            dec = [((v if v < Q // 2 else v - Q) + delta // 2) // delta for v in ph]
            assert dec == [int(v) for v in (W[i] @ msgs)]
        with pytest.raises(grpc.RpcError) as e:
            client.generate("linear", x[:2])
        assert e.value.code() == grpc.StatusCode.INVALID_ARGUMENT
    finally:
        client.close()
        server.stop()
        ctx.close()


@pytest.mark.gpu
def test_multiply_relinearize_over_rpc_with_session_keys():
    from deeppowers_amd.evaluator import Context
    from oracle.cbind import Oracle
    p = small_params()
    rng = np.random.default_rng(12)
    s = rng.integers(-1, 2, p.n)
    delta = 1 << 40
    m1, m2 = rng.integers(0, 1000, (2, p.n)), rng.integers(0, 1000, (2, p.n))
    a = np.stack([encrypt(rng, p, s, m1[i], delta)[0] for i in range(2)])
    b = np.stack([encrypt(rng, p, s, m2[i], delta)[0] for i in range(2)])
    evk, _ = keygen_relin(rng, p, s)
    ctx = Context(p, 0)
    server = rpc.EncryptedInferenceServer(ctx)
    server.register_model("multiply", rpc.MultiplyRelinearize())
    port = server.start("127.0.0.1:0")
    client = rpc.EncryptedClient(f"127.0.0.1:{port}", p)
    try:
        with pytest.raises(grpc.RpcError) as e:            # keys first
            client.generate("multiply", a, b)
        assert e.value.code() == grpc.StatusCode.INVALID_ARGUMENT and "keys" in e.value.details()
        assert client.register_relin_keys(evk)
        y, is_ntt = client.generate("multiply", a, b)
        assert not is_ntt and y.shape == (2, 2, p.n_limbs, p.n)
        orc = Oracle.from_params(p)
        assert np.array_equal(y, orc.relinearize(orc.ct_mul(a, b), evk))
        for i in range(2):
            ph, Q = phase(p, y[i], s)
            dec = [((v if v < Q // 2 else v - Q) + (delta * delta) // 2) // (delta * delta) for v in ph]
            assert dec == [v if v < Q // 2 else v - Q for v in negacyclic_int([int(v) for v in m1[i]], [int(v) for v in m2[i]], Q)]
        # another session has no keys
        other = rpc.EncryptedClient(f"127.0.0.1:{port}", p)
        with pytest.raises(grpc.RpcError) as e:
            other.generate("multiply", a, b)
        assert e.value.code() == grpc.StatusCode.INVALID_ARGUMENT
        other.close()
    finally:
        client.close()
        server.stop()
        ctx.close()
