"""a `Generate`-style gRPC service whose payloads are ciphertexts.

The reference serves text (`/root/reference/src/core/api/deeppowers.proto:6-32`, handler
`/root/reference/src/core/api/grpc_server.cpp:163-226`: build a request, run the model, fill the response, record the
latency, map exceptions to `INTERNAL`).  This is the encrypted twin: the request carries DPFHEv1 ciphertext streams
(`wire.py`), the server runs a registered *encrypted model* - a callable built from `Evaluator` operations, i.e. HIP
kernels through the C ABI - and answers with a ciphertext stream.  The service never holds a secret key.

Messages are declared in `dpfhe_rpc.proto` (next to this file).  The image has grpcio and protobuf but no protoc plugin,
so the descriptors are built here at run time; tests/test_rpc.py keeps the two in step.

    server = EncryptedInferenceServer(ctx)                       # ctx: deeppowers_amd.evaluator.Context
    server.register_model("lm_head_tile", ScalarLinear(ctx, W))  # y_i = sum_j w_ij x_j, one ciphertext per feature
    server.register_model("multiply", MultiplyRelinearize())     # the metric op + relinearisation with the session's keys
    port = server.start("127.0.0.1:0")
    client = EncryptedClient(f"127.0.0.1:{port}", params)
    y = client.generate("lm_head_tile", x_words)                 # numpy uint64 [batch][2][L][N] in, same out
"""
from __future__ import annotations

import collections
import threading
import time
import uuid
from concurrent import futures

import grpc
import numpy as np
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

from . import wire
from .params import FheParams

PACKAGE = "deeppowers"
SERVICE = "EncryptedDeepPowers"
_T = descriptor_pb2.FieldDescriptorProto

# (message, [(field, number, type, type_name)]) - the run-time statement of dpfhe_rpc.proto
MESSAGES = [
    ("EncryptedGenerateRequest", [("request_id", 1, _T.TYPE_STRING, None), ("model", 2, _T.TYPE_STRING, None), ("ciphertext", 3, _T.TYPE_BYTES, None),
                                  ("session_id", 4, _T.TYPE_STRING, None), ("ciphertext_b", 5, _T.TYPE_BYTES, None)]),
    ("EncryptedGenerateResponse", [("request_id", 1, _T.TYPE_STRING, None), ("ciphertext", 2, _T.TYPE_BYTES, None), ("server_ms", 3, _T.TYPE_FLOAT, None),
                                   ("device_ms", 4, _T.TYPE_FLOAT, None)]),
    ("RegisterKeysRequest", [("session_id", 1, _T.TYPE_STRING, None), ("relin_keys", 2, _T.TYPE_BYTES, None)]),
    ("RegisterKeysResponse", [("ok", 1, _T.TYPE_BOOL, None)]),
    ("MetricsRequest", []),
    ("LatencyMetrics", [("avg_ms", 1, _T.TYPE_FLOAT, None), ("p50_ms", 2, _T.TYPE_FLOAT, None), ("p90_ms", 3, _T.TYPE_FLOAT, None), ("p99_ms", 4, _T.TYPE_FLOAT, None)]),
    ("ThroughputMetrics", [("requests_per_second", 1, _T.TYPE_FLOAT, None), ("ciphertexts_per_second", 2, _T.TYPE_FLOAT, None)]),
    ("ErrorMetrics", [("total_errors", 1, _T.TYPE_INT32, None), ("invalid_argument_errors", 2, _T.TYPE_INT32, None), ("internal_errors", 3, _T.TYPE_INT32, None)]),
    ("MetricsResponse", [("latency", 1, _T.TYPE_MESSAGE, "LatencyMetrics"), ("throughput", 2, _T.TYPE_MESSAGE, "ThroughputMetrics"),
                         ("errors", 3, _T.TYPE_MESSAGE, "ErrorMetrics"), ("total_requests", 4, _T.TYPE_INT32, None)]),
]
METHODS = [("EncryptedGenerate", "EncryptedGenerateRequest", "EncryptedGenerateResponse"),
           ("RegisterKeys", "RegisterKeysRequest", "RegisterKeysResponse"),
           ("GetMetrics", "MetricsRequest", "MetricsResponse")]
# ciphertext batches are MBs (256 KiB per ciphertext at N=4096, L=4): gRPC's 4 MiB default is lifted - to a BOUND derived from the
# parameters and a maximum batch on the server (one oversized request must not be able to exhaust the host or the GPU), without a
# bound on the client's side of the channel
CHANNEL_OPTIONS = [("grpc.max_receive_message_length", -1), ("grpc.max_send_message_length", -1)]


def message_limit(params: FheParams, max_batch: int) -> int:
    """bytes of the largest legal message: `max_batch` three-component ciphertexts or one set of relinearisation keys ([L][2][L][N] words,
    RegisterKeys) - whichever is larger, so that a server with a small max_batch still accepts keys - + the wire header + protobuf framing slack"""
    L, n = params.n_limbs, params.n
    return max(max_batch * 3 * L * n * 8, 2 * L * L * n * 8) + 4096 + 8 * L


def _build_messages():
    fd = descriptor_pb2.FileDescriptorProto(name="dpfhe_rpc.proto", package=PACKAGE, syntax="proto3")
    for name, fields in MESSAGES:
        m = fd.message_type.add(name=name)
        for fname, number, ftype, tname in fields:
            f = m.field.add(name=fname, number=number, type=ftype, label=_T.LABEL_OPTIONAL)
            if tname:
                f.type_name = f".{PACKAGE}.{tname}"
    svc = fd.service.add(name=SERVICE)
    for mname, req, resp in METHODS:
        svc.method.add(name=mname, input_type=f".{PACKAGE}.{req}", output_type=f".{PACKAGE}.{resp}")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return {name: message_factory.GetMessageClass(pool.FindMessageTypeByName(f"{PACKAGE}.{name}")) for name, _ in MESSAGES}


pb = _build_messages()   # pb["EncryptedGenerateRequest"](request_id=..., ...)


def method_path(name: str) -> str:
    return f"/{PACKAGE}.{SERVICE}/{name}"


class RpcError(Exception):
    """Raised by models / the server for errors the client caused; mapped to INVALID_ARGUMENT (everything else: INTERNAL)."""


# ---- encrypted models: callables (server, session_keys, a_words, b_words) -> device Ciphertext -------------------------
class ScalarLinear:
    """y_i = sum_j w_ij x_j with public integer weights: one ciphertext per input feature, one sample per coefficient - the
    encrypted stand-in for the reference's matvec sites (gpt_model.cpp:793,848,883).  Input batch = cols, output batch = rows."""

    def __init__(self, ctx, weights):
        import torch
        w = np.asarray(weights, dtype=np.int64)
        if w.ndim != 2:
            raise ValueError("weights: [rows][cols] integers")
        q = np.array(ctx.params.moduli, dtype=object)
        res = np.array([[[int(v) % int(m) for m in q] for v in row] for row in w.tolist()], dtype=np.uint64)
        self.rows, self.cols = w.shape
        self.w = torch.from_numpy(res.view(np.int64)).to(ctx.device).contiguous()

    def __call__(self, server, keys, a, b):
        if b is not None:
            raise RpcError("this model takes one operand")
        if a.size != 2 or a.data.dim() != 4 or a.data.shape[0] != self.cols:
            raise RpcError(f"expected a batch of {self.cols} two-component ciphertexts")
        return server.ev.matvec_scalar(self.w, a)


class MultiplyRelinearize:
    """The metric op over the wire: (a_i (x) b_i) relinearised with the session's keys; `relinearize=False` returns the
    3-component tensor products."""

    def __init__(self, relinearize: bool = True):
        self.relinearize = relinearize

    def __call__(self, server, keys, a, b):
        if b is None:
            raise RpcError("this model takes two operands (ciphertext, ciphertext_b)")
        if a.size != 2 or b.size != 2 or a.data.shape != b.data.shape or a.is_ntt or b.is_ntt:
            raise RpcError("operands must be equally shaped two-component coefficient-domain ciphertexts")
        prod = server.ev.multiply(a, b)
        if not self.relinearize:
            return prod
        if keys is None:
            raise RpcError("no relinearisation keys registered for this session")
        return server.ev.relinearize(prod, keys)


class Passthrough:
    """Echo (transport tests; touches no device)."""
    host_only = True

    def __call__(self, server, keys, a, b):
        return a


class _Metrics:
    def __init__(self):
        self.lock = threading.Lock()
        self.lat_ms, self.cts, self.t0, self.requests = [], 0, time.time(), 0
        self.invalid, self.internal = 0, 0

    def record(self, ms, n_ct):
        with self.lock:
            self.requests += 1
            self.lat_ms.append(ms)
            if len(self.lat_ms) > 4096:          # percentiles over a sliding window: a long-running server must not grow without bound
                del self.lat_ms[:2048]
            self.cts += n_ct

    def error(self, invalid):
        with self.lock:
            if invalid:
                self.invalid += 1
            else:
                self.internal += 1

    def snapshot(self):
        with self.lock:
            lat = sorted(self.lat_ms)
            el = max(time.time() - self.t0, 1e-9)
            pick = lambda f: lat[min(len(lat) - 1, int(f * len(lat)))] if lat else 0.0
            return dict(avg=sum(lat) / len(lat) if lat else 0.0, p50=pick(0.5), p90=pick(0.9), p99=pick(0.99), rps=self.requests / el, cps=self.cts / el,
                        invalid=self.invalid, internal=self.internal, total=self.requests + self.invalid + self.internal)


class EncryptedInferenceServer:
    """gRPC front of one Context.  `ctx` may be None for a transport-only server (host_only models)."""

    def __init__(self, ctx, params: FheParams | None = None, max_workers: int = 4, max_sessions: int = 64, max_batch: int = 1024):
        self.ctx = ctx
        self.params = params if params is not None else ctx.params
        self.max_batch = max_batch   # ciphertexts per request: bounds the message size (start()) and what _to_device may upload
        self.ev = None
        if ctx is not None:
            from .evaluator import Evaluator
            self.ev = Evaluator(ctx)
        # evaluation keys are MBs of device memory per session: least-recently-used sessions are dropped beyond `max_sessions`
        self.models, self.sessions, self.max_sessions = {}, collections.OrderedDict(), max_sessions
        self._key_digests = {}   # session id -> sha256 of its registered key words (idempotent re-registration, rpc RegisterKeys)
        self.metrics = _Metrics()
        self._gpu_lock = threading.Lock()   # one evaluation at a time per context: requests queue here, kernels fill the chip anyway
        self._server, self._workers = None, max_workers

    def register_model(self, name: str, model):
        self.models[name] = model

    # -- handlers ----------------------------------------------------------------------------------------------
    def _to_device(self, blob):
        from .evaluator import Ciphertext, to_device
        words, is_ntt = wire.loads(blob, self.params)
        if words.shape[0] > self.max_batch:
            raise ValueError(f"batch of {words.shape[0]} ciphertexts exceeds this server's limit of {self.max_batch}")
        return Ciphertext(to_device(words, self.ctx.device), is_ntt)

    def _generate(self, request, context):
        code, result = self._generate_impl(request)
        if code is not grpc.StatusCode.OK:
            context.abort(code, result)
        return result

    def _generate_impl(self, request):
        """-> (StatusCode.OK, response) or (error code, message).  Client mistakes (unknown model, malformed stream, wrong
        shapes, missing keys) are INVALID_ARGUMENT / NOT_FOUND; every failure of the evaluation itself is INTERNAL with the
        exception's text, as in the reference (grpc_server.cpp:220-225)."""
        t0 = time.time()
        model = self.models.get(request.model)
        if model is None:
            self.metrics.error(True)
            return grpc.StatusCode.NOT_FOUND, f"no model named '{request.model}'"
        try:
            if getattr(model, "host_only", False):
                words, is_ntt = wire.loads(request.ciphertext, self.params)
                out_blob, n_ct, dev_ms = wire.dumps(words, self.params, is_ntt), words.shape[0], 0.0
            else:
                if self.ctx is None:
                    raise RuntimeError("this server has no device context")
                import torch
                from .evaluator import to_host
                with self._gpu_lock:
                    a = self._to_device(request.ciphertext)
                    b = self._to_device(request.ciphertext_b) if request.ciphertext_b else None
                    keys = self.sessions.get(request.session_id)
                    if keys is not None:
                        self.sessions.move_to_end(request.session_id)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s = torch.cuda.current_stream(self.ctx.device)
                    e0.record(s)
                    y = model(self, keys, a, b)
                    e1.record(s)
                    out = to_host(y.data)
                    dev_ms = e0.elapsed_time(e1)
                out = out.reshape((-1,) + out.shape[-3:])
                out_blob, n_ct = wire.dumps(out, self.params, y.is_ntt), out.shape[0]
        except (ValueError, RpcError) as ex:
            self.metrics.error(True)
            return grpc.StatusCode.INVALID_ARGUMENT, str(ex)
        except Exception as ex:
            self.metrics.error(False)
            return grpc.StatusCode.INTERNAL, f"Internal error: {ex}"
        ms = (time.time() - t0) * 1e3
        self.metrics.record(ms, n_ct)
        return grpc.StatusCode.OK, pb["EncryptedGenerateResponse"](request_id=request.request_id, ciphertext=out_blob, server_ms=ms, device_ms=dev_ms)

    def _register_keys(self, request, context):
        try:
            words, is_ntt = wire.loads(request.relin_keys, self.params)
            L = self.params.n_limbs
            if words.shape[:2] != (L, 2) or not is_ntt:
                raise ValueError("relin_keys: batch = n_limbs, 2 components, NTT domain")
            if not request.session_id:
                raise ValueError("session_id is empty")
        except ValueError as ex:
            self.metrics.error(True)
            context.abort(grpc.StatusCode.INVALID_ARGUMENT, str(ex))
        import hashlib
        digest = hashlib.sha256(words.tobytes()).digest()
        # The session id is the bearer token of its keys (clients draw 128 random bits, EncryptedClient): a second registration under an
        # existing id - another tenant guessing or replaying it - must not replace the keys its owner's requests rely on.  Checked BEFORE
        # the upload (a replay costs no GPU copy); re-sending the SAME keys (a client retrying after a deadline whose first call did land)
        # is idempotent and answers ok.
        with self._gpu_lock:
            known = self._key_digests.get(request.session_id) if request.session_id in self.sessions else None
        if known is not None:
            if known == digest:
                return pb["RegisterKeysResponse"](ok=True)
            self.metrics.error(True)
            context.abort(grpc.StatusCode.ALREADY_EXISTS, "session already holds other keys: register under a fresh session id")
        if self.ctx is None:
            keys = words
        else:
            from .evaluator import to_device
            keys = to_device(words, self.ctx.device)
        with self._gpu_lock:
            if request.session_id in self.sessions:   # lost a race against a concurrent registration of the same id
                if self._key_digests.get(request.session_id) == digest:
                    return pb["RegisterKeysResponse"](ok=True)
                self.metrics.error(True)
                context.abort(grpc.StatusCode.ALREADY_EXISTS, "session already holds other keys: register under a fresh session id")
            self.sessions[request.session_id] = keys
            self._key_digests[request.session_id] = digest
            self.sessions.move_to_end(request.session_id)
            while len(self.sessions) > self.max_sessions:
                old_id, _ = self.sessions.popitem(last=False)
                self._key_digests.pop(old_id, None)
        return pb["RegisterKeysResponse"](ok=True)

    def _get_metrics(self, request, context):
        m = self.metrics.snapshot()
        return pb["MetricsResponse"](latency=pb["LatencyMetrics"](avg_ms=m["avg"], p50_ms=m["p50"], p90_ms=m["p90"], p99_ms=m["p99"]),
                                     throughput=pb["ThroughputMetrics"](requests_per_second=m["rps"], ciphertexts_per_second=m["cps"]),
                                     errors=pb["ErrorMetrics"](total_errors=m["invalid"] + m["internal"], invalid_argument_errors=m["invalid"], internal_errors=m["internal"]),
                                     total_requests=m["total"])

    # -- lifecycle ---------------------------------------------------------------------------------------------
    def start(self, address: str = "127.0.0.1:0", credentials=None) -> int:
        """`credentials`: grpc.ssl_server_credentials(...) for TLS (the reference's optional init_ssl, grpc_server.cpp); None = plain TCP"""
        impl = {"EncryptedGenerate": self._generate, "RegisterKeys": self._register_keys, "GetMetrics": self._get_metrics}
        handlers = {name: grpc.unary_unary_rpc_method_handler(impl[name], request_deserializer=pb[req].FromString, response_serializer=pb[resp].SerializeToString)
                    for name, req, resp in METHODS}
        limit = message_limit(self.params, self.max_batch)
        self._server = grpc.server(futures.ThreadPoolExecutor(max_workers=self._workers),
                                   options=[("grpc.max_receive_message_length", limit), ("grpc.max_send_message_length", limit)])
        self._server.add_generic_rpc_handlers((grpc.method_handlers_generic_handler(f"{PACKAGE}.{SERVICE}", handlers),))
        port = self._server.add_secure_port(address, credentials) if credentials is not None else self._server.add_insecure_port(address)
        if port == 0:
            raise RuntimeError(f"could not bind {address}")
        self._server.start()
        return port

    def stop(self, grace: float = 0.5):
        if self._server is not None:
            self._server.stop(grace).wait()
            self._server = None


class EncryptedClient:
    """Client stub.  Ciphertexts travel as numpy uint64 arrays [batch][components][L][N] (what Encryptor / PolyBuffer::save produce)."""

    def __init__(self, target: str, params: FheParams, session_id: str | None = None):
        self.params = params
        self.session_id = session_id or uuid.uuid4().hex
        self._channel = grpc.insecure_channel(target, options=CHANNEL_OPTIONS)
        self._calls = {name: self._channel.unary_unary(method_path(name), request_serializer=pb[req].SerializeToString, response_deserializer=pb[resp].FromString)
                       for name, req, resp in METHODS}
        self.last_response = None

    def register_relin_keys(self, evk_ntt: np.ndarray, timeout: float = 60.0) -> bool:
        blob = wire.dumps(np.asarray(evk_ntt, dtype=np.uint64), self.params, True)
        return self._calls["RegisterKeys"](pb["RegisterKeysRequest"](session_id=self.session_id, relin_keys=blob), timeout=timeout).ok

    def generate(self, model: str, a: np.ndarray, b: np.ndarray | None = None, is_ntt: bool = False, timeout: float = 120.0):
        """-> (words [batch][components][L][N], is_ntt)"""
        req = pb["EncryptedGenerateRequest"](request_id=uuid.uuid4().hex, model=model, session_id=self.session_id,
                                             ciphertext=wire.dumps(a, self.params, is_ntt),
                                             ciphertext_b=wire.dumps(b, self.params, is_ntt) if b is not None else b"")
        resp = self._calls["EncryptedGenerate"](req, timeout=timeout)
        if resp.request_id != req.request_id:
            raise RuntimeError("response does not belong to this request")
        self.last_response = resp
        return wire.loads(resp.ciphertext, self.params)

    def metrics(self, timeout: float = 10.0):
        return self._calls["GetMetrics"](pb["MetricsRequest"](), timeout=timeout)

    def close(self):
        self._channel.close()
