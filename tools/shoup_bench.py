"""Generic-prime (ShoupArith) arm next to the fold arm at BASELINE's N = 4096, L = 4 (tool): forward / inverse NTT at configs[1]'s batch (1024 RNS
polynomials) and the fused multiply at `pairs` ciphertext pairs, HIP events per launch, launches enqueued back to back.
    python tools/shoup_bench.py [pairs=2048]
The generic limbs are the largest primes = 1 mod 8192 below 2^59, 2^50, 2^40 and 2^33 (none of the form 2^60 - d)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator
from deeppowers_amd.params import FheParams

def timed(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) * 1e3 for s, e in evs)
    return ts[len(ts) // 2], ts[0]

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
out = {}
for name, params in (("fold", FheParams.n4096_l4()), ("shoup", FheParams.generic_n4096_l4())):
    ctx = Context(params, 0); ev = Evaluator(ctx)
    assert ctx.uses_fold == (name == "fold")
    L, N = params.n_limbs, params.n
    q = torch.tensor(params.moduli, dtype=torch.int64, device=ctx.device).view(1, L, 1)
    g = torch.Generator(device=ctx.device).manual_seed(7)
    x = torch.randint(0, 2**62, (1024, L, N), generator=g, dtype=torch.int64, device=ctx.device) % q
    y = torch.empty_like(x)
    nbytes = 2 * N * 8 * 1024 * L
    r = {}
    for d, fn in (("ntt_fwd", lambda: ev.ntt_forward(x, out=y)), ("ntt_inv", lambda: ev.ntt_inverse(x, out=y))):
        med, mn = timed(fn, 30)
        r[d] = med
        print(f"{name:6s} {d}: median {med:7.1f} us  min {mn:7.1f} us  = {nbytes / med / 8e6 * 100:5.1f} % of 8 TB/s")
    a = Ciphertext(torch.randint(0, 2**62, (pairs, 2, L, N), generator=g, dtype=torch.int64, device=ctx.device) % q.view(1, 1, L, 1))
    b = Ciphertext(torch.randint(0, 2**62, (pairs, 2, L, N), generator=g, dtype=torch.int64, device=ctx.device) % q.view(1, 1, L, 1))
    o = ctx.empty(pairs, components=3)
    med, mn = timed(lambda: ev.multiply(a, b, out=o), 10)
    r["ct_mul"] = med
    print(f"{name:6s} ct_mul x{pairs}: median {med:8.1f} us -> {pairs / med:6.3f} M ct-mul/s = {7 * L * N * 8 * pairs / med / 8e6 * 100:5.1f} % of 8 TB/s")
    out[name] = r
    del a, b, o, x, y
    ctx.close()
print("fold over shoup: " + "  ".join(f"{k} {out['shoup'][k] / out['fold'][k]:.2f}x" for k in out["fold"]))
