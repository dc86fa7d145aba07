"""-m gpu: the C ABI driven from several HOST THREADS at once ("concurrent calls from different host threads on different
streams"; the reference's callers are worker pools - /root/reference/src/core/execution/execution_engine.hpp:96-97 - each with its own stream -
/root/reference/src/core/hal/cuda/cuda_device.hpp:35).

ctypes releases the GIL for the duration of a foreign call, so the threads below really are inside libdpfhe_hip.so together.  What only such a caller
exercises: the per-stream scratch arenas behind one mutex (creation, growth under contention, LRU eviction, explicit release), the thread-local
dpfhe_last_error, the process-wide tune cache and the per-context atomics.  Every result is compared word for word with the oracle."""
import threading

import numpy as np
import pytest

from deeppowers_amd.params import FheParams, PRIMES_60
from oracle import pyoracle as po
from oracle.cbind import Oracle

N_THREADS = 6


def _large_params():
    qs = [PRIMES_60[i][0] for i in (1, 2, 4)]     # the pinned primes that are 1 mod 32768
    return FheParams(14, tuple(qs), tuple(po.min_primitive_2n_root(16384, q) for q in qs))


@pytest.mark.gpu
def test_threads_share_contexts_each_on_its_own_stream():
    import torch
    from deeppowers_amd import _cabi
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host
    lib = _cabi.load()
    ps, pl = FheParams.n4096_l4(), _large_params()
    cs, cl = Context(ps, 0), Context(pl, 0)       # ONE context per ring, shared by all threads
    os_, ol = Oracle.from_params(ps), Oracle.from_params(pl)
    dev = cs.device
    try:
        # operands and expected words, per thread (different seeds: a thread that reads another's scratch shows up)
        jobs = []
        for t in range(N_THREADS):
            j = {}
            Ls, ns, Ll, nl = ps.n_limbs, ps.n, pl.n_limbs, pl.n
            j["a_s"], j["b_s"] = os_.fill(4, 100 + t).reshape(2, 2, Ls, ns), os_.fill(4, 200 + t).reshape(2, 2, Ls, ns)
            j["want_s"] = os_.ct_mul(j["a_s"], j["b_s"], threads=0)
            j["evk_s"] = os_.fill(Ls * 2, 300 + t).reshape(Ls, 2, Ls, ns)
            j["relin_s"] = os_.relinearize(j["want_s"], j["evk_s"], threads=0)
            nb = 1 + t % 3                         # different batch sizes: arenas of different sizes, growth on the second round
            j["a_l"], j["b_l"] = ol.fill(2 * nb, 400 + t).reshape(nb, 2, Ll, nl), ol.fill(2 * nb, 500 + t).reshape(nb, 2, Ll, nl)
            j["want_l"] = ol.ct_mul(j["a_l"], j["b_l"], threads=0)
            j["evk_l"] = ol.fill(Ll * 2, 600 + t).reshape(Ll, 2, Ll, nl)
            j["relin_l"] = ol.relinearize(j["want_l"], j["evk_l"], threads=0)
            nb2 = nb + 2                           # a LARGER batch afterwards: the stream's arena has to grow while other threads use theirs
            j["a_l2"], j["b_l2"] = ol.fill(2 * nb2, 700 + t).reshape(nb2, 2, Ll, nl), ol.fill(2 * nb2, 800 + t).reshape(nb2, 2, Ll, nl)
            j["want_l2"] = ol.ct_mul(j["a_l2"], j["b_l2"], threads=0)
            jobs.append(j)
        errors, barrier = [], threading.Barrier(N_THREADS)

        def worker(t):
            try:
                torch.cuda.set_device(dev)
                stream = torch.cuda.Stream(device=dev)
                evs, evl = Evaluator(cs), Evaluator(cl)
                j = jobs[t]
                with torch.cuda.stream(stream):
                    d = {k: to_device(v, dev) for k, v in j.items() if not k.startswith(("want", "relin"))}
                stream.synchronize()
                barrier.wait()
                for rep in range(3):
                    c_s = evs.multiply(Ciphertext(d["a_s"]), Ciphertext(d["b_s"]), stream=stream)
                    c_l = evl.multiply(Ciphertext(d["a_l"]), Ciphertext(d["b_l"]), stream=stream)           # composed: arena of this stream
                    r_s = evs.relinearize(c_s, d["evk_s"], stream=stream)
                    r_l = evl.relinearize(c_l, d["evk_l"], stream=stream)                                   # composed: the arena is reused (L^2 / 2 x the input)
                    c_l2 = evl.multiply(Ciphertext(d["a_l2"]), Ciphertext(d["b_l2"]), stream=stream) if rep else None   # growth from the second round on
                    # a deliberate error in the middle: its message must stay THIS thread's (thread-local dpfhe_last_error)
                    rc = lib.dpfhe_ct_mul(cs.handle, 0, d["a_s"].data_ptr(), d["b_s"].data_ptr(), 1, 0, stream.cuda_stream)
                    msg = lib.dpfhe_last_error().decode()
                    assert rc == 2000 and "dpfhe_ct_mul" in msg, (rc, msg)
                    if t % 2:   # half of the threads provoke a DIFFERENT error and must read their own text back, whatever the others did meanwhile
                        rc = lib.dpfhe_relinearize(cl.handle, 0, 0, 0, 1, stream.cuda_stream)
                        assert rc == 2000 and "relinearize" in lib.dpfhe_last_error().decode()
                    stream.synchronize()
                    assert np.array_equal(to_host(c_s.data), j["want_s"]), ("ct_mul N=4096", t, rep)
                    assert np.array_equal(to_host(r_s.data), j["relin_s"]), ("relinearize N=4096", t, rep)
                    assert np.array_equal(to_host(c_l.data), j["want_l"]), ("ct_mul N=16384", t, rep)
                    assert np.array_equal(to_host(r_l.data), j["relin_l"]), ("relinearize N=16384", t, rep)
                    if c_l2 is not None:
                        assert np.array_equal(to_host(c_l2.data), j["want_l2"]), ("ct_mul N=16384, grown arena", t, rep)
                    if rep == 1 and t % 3 == 0:   # hand this stream's arena back in the middle: the next round re-creates it under contention
                        cl.release_scratch(stream)
            except BaseException as e:   # noqa: BLE001 - reported by the main thread
                errors.append((t, repr(e)))
                try:
                    barrier.abort()
                except Exception:
                    pass

        threads = [threading.Thread(target=worker, args=(t,)) for t in range(N_THREADS)]
        for th in threads:
            th.start()
        for th in threads:
            th.join(600)
        assert not errors, errors
        assert not any(th.is_alive() for th in threads)
        held = cl.scratch_bytes
        assert held > 0                                   # the arenas of the streams that did not release theirs
        cl.release_scratch(all_streams=True)
        assert cl.scratch_bytes == 0 and cs.scratch_bytes == 0   # (N = 4096 never composes: no arena)
    finally:
        cs.close()
        cl.close()


@pytest.mark.gpu
def test_scratch_arenas_are_capped_and_released():
    """a caller that rotates through a pool of streams keeps at most 16 arenas per context (least recently used evicted); release works per stream"""
    import torch
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host
    p = _large_params()
    ctx = Context(p, 0)
    ev = Evaluator(ctx)
    orc = Oracle.from_params(p)
    try:
        L, n = p.n_limbs, p.n
        ah, bh = orc.fill(2, 1).reshape(1, 2, L, n), orc.fill(2, 2).reshape(1, 2, L, n)
        want = orc.ct_mul(ah, bh, threads=0)
        a, b = Ciphertext(to_device(ah, ctx.device)), Ciphertext(to_device(bh, ctx.device))
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(device=ctx.device) for _ in range(20)]
        per = None
        for i, s in enumerate(streams):
            got = ev.multiply(a, b, stream=s)
            s.synchronize()
            assert np.array_equal(to_host(got.data), want)
            if per is None:
                per = ctx.scratch_bytes
                assert per > 0
            assert ctx.scratch_bytes == per * min(i + 1, 16)      # capped at 16 arenas
        ctx.release_scratch(streams[-1])
        assert ctx.scratch_bytes == per * 15
        ctx.release_scratch(streams[0])                            # evicted long ago: nothing to release, no error
        assert ctx.scratch_bytes == per * 15
        got = ev.multiply(a, b, stream=streams[0])                 # and it simply gets a new arena
        streams[0].synchronize()
        assert np.array_equal(to_host(got.data), want) and ctx.scratch_bytes == per * 16
        ctx.release_scratch(all_streams=True)
        assert ctx.scratch_bytes == 0
    finally:
        ctx.close()
