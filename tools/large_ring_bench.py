"""NTT and composed ct x ct / relinearisation timing at ring degrees above 8192 (tool): N = 16384 (one 1024-thread workgroup per polynomial) and the split transforms
N = 32768 / 65536, ~1 GiB of residues each, kernel time by HIP events.  usage: python tools/large_ring_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeppowers_amd._cabi as _cabi  # noqa: E402

if os.environ.get("DPFHE_AB_LIB"):
    _cabi.LIB_PATH = os.path.abspath(os.environ["DPFHE_AB_LIB"])
from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator  # noqa: E402
from deeppowers_amd.params import FheParams  # noqa: E402
from deeppowers_amd.params import ntt_primes  # noqa: E402

for ln, polys in ((14, 6144), (15, 4096), (16, 2048)):
    n = 1 << ln
    L = 2
    p = ntt_primes(ln, L)
    qs = p.moduli
    ctx = Context(p, 0)
    ev = Evaluator(ctx)
    q = torch.tensor(qs, dtype=torch.int64, device=ctx.device).view(1, L, 1)
    x = torch.randint(0, 2**62, (polys // L, L, n), dtype=torch.int64, device=ctx.device) % q
    y = torch.empty_like(x)
    nbytes = 2 * n * 8 * polys
    for name, fn in (("fwd", lambda: ev.ntt_forward(x, out=y)), ("inv", lambda: ev.ntt_inverse(x, out=y))):
        for _ in range(3):
            fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
        for s, e in evs:
            s.record(); fn(); e.record()
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) * 1e3 for s, e in evs)
        print(f"{os.path.basename(os.environ.get('DPFHE_AB_LIB', 'HEAD')):14s} N={n:6d} {polys} residue polys ntt_{name}: median {ts[7]:8.1f} us  = {nbytes / ts[7] / 8e6 * 100:5.1f}% of 8 TB/s")
    del x, y
    # the composed multiply / relinearisation (kernels_large.h): 4 + 3 transforms and one tensor pass; L (L + 2) transforms and three passes
    pairs = max(1, 256 >> (ln - 14))
    a = Ciphertext(torch.randint(0, 2**62, (pairs, 2, L, n), dtype=torch.int64, device=ctx.device) % q.view(1, 1, L, 1))
    b = Ciphertext(torch.randint(0, 2**62, (pairs, 2, L, n), dtype=torch.int64, device=ctx.device) % q.view(1, 1, L, 1))
    evk = torch.randint(0, 2**62, (L, 2, L, n), dtype=torch.int64, device=ctx.device) % q.view(1, 1, L, 1)
    c = ev.multiply(a, b)
    for name, fn, alg in (("ct_mul", lambda: ev.multiply(a, b, out=c.data), 7 * L * n * 8 * pairs), ("relinearize", lambda: ev.relinearize(c, evk), (3 + 2) * L * n * 8 * pairs)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fn()
        e.record(); torch.cuda.synchronize()
        t = s.elapsed_time(e) * 1e3 / 10
        print(f"{os.path.basename(os.environ.get('DPFHE_AB_LIB', 'HEAD')):14s} N={n:6d} {pairs} ciphertexts, L={L}: composed {name}: {t:8.1f} us = {pairs / t * 1e6:9.0f} /s, algorithmic bytes at {alg / t / 8e6 * 100:5.1f}% of 8 TB/s")
    ctx.close()
