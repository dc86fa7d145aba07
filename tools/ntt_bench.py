"""Quick kernel timing (HIP events) for A/B experiments: NTT fwd/inv at BASELINE configs[1] and ct_mul.
usage: python tools/ntt_bench.py [batch_rns_polys=1024] [ct_batch=2048]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeppowers_amd._cabi as _cabi  # noqa: E402

if os.environ.get("DPFHE_AB_LIB"):  # A/B experiments: time another build of the library (tool only)
    _cabi.LIB_PATH = os.path.abspath(os.environ["DPFHE_AB_LIB"])
from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator  # noqa: E402
from deeppowers_amd.params import FheParams  # noqa: E402


def timeit(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    cb = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    tag = os.environ.get("TAG", "")
    for name, params in (("n4096", FheParams.n4096_l4()), ("n8192", FheParams.n8192_l6())):
        ctx = Context(params, 0)
        ev = Evaluator(ctx)
        L, N = params.n_limbs, params.n
        q = torch.tensor(params.moduli, dtype=torch.int64, device=ctx.device).view(1, L, 1)
        b = nb if name == "n4096" else nb // 4
        x = torch.randint(0, 2**62, (b, L, N), dtype=torch.int64, device=ctx.device) % q
        y = torch.empty_like(x)
        nbytes = 2 * N * 8 * b * L
        for d, fn in (("fwd", lambda: ev.ntt_forward(x, out=y)), ("inv", lambda: ev.ntt_inverse(x, out=y))):
            med, mn = timeit(fn)
            print(f"{tag} {name} ntt_{d}: median {med:8.1f} us  min {mn:8.1f} us  {nbytes / med / 1e6:7.1f} GB/s  = {nbytes / med / 8e6 * 100:5.1f}% of 8 TB/s")
        bb = cb if name == "n4096" else cb // 4
        a = Ciphertext(torch.randint(0, 2**62, (bb, 2, L, N), dtype=torch.int64, device=ctx.device) % q.view(1, 1, L, 1))
        c = Ciphertext(torch.randint(0, 2**62, (bb, 2, L, N), dtype=torch.int64, device=ctx.device) % q.view(1, 1, L, 1))
        out = ctx.empty(bb, components=3)
        med, mn = timeit(lambda: ev.multiply(a, c, out=out), reps=15, warm=3)
        alg = 7 * L * N * 8 * bb
        print(f"{tag} {name} ct_mul x{bb}: median {med:8.1f} us  -> {bb / med:7.3f} M ct-mul/s   {alg / med / 1e6:7.1f} GB/s = {alg / med / 8e6 * 100:5.1f}% of 8 TB/s")
        evk = torch.randint(0, 2**62, (L, 2, L, N), dtype=torch.int64, device=ctx.device) % q.view(1, 1, L, 1)
        c3 = Ciphertext(out)
        out2 = ctx.empty(bb, components=2)
        med, mn = timeit(lambda: ev.relinearize(c3, evk, out=out2), reps=10, warm=2)
        alg = (L + 4) * L * N * 8 * bb   # per (ct, limb): L digit reads + c0, c1 reads + 2 writes
        print(f"{tag} {name} relinearize x{bb}: median {med:8.1f} us  -> {bb / med:7.3f} M relin/s   {alg / med / 1e6:7.1f} GB/s")
        ctx.close()


if __name__ == "__main__":
    main()
