"""Per-limb arithmetic classes at BASELINE's N = 4096, L = 4 (tool): forward / inverse NTT at configs[1]'s batch (1024 RNS polynomials) and the
fused multiply at `pairs` ciphertext pairs for one context per class - fold (2^60 - d), f64 (30- and 45-bit primes), fold_scaled (59-bit),
f64_wide (49-bit), shoup (55-bit primes too far below 2^55 for the scaled fold) - and the mixed 59/50/40/33 context.  HIP events per launch.
    python tools/class_bench.py [pairs=2048] [log2n=12]"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator
from deeppowers_amd.params import FheParams, ntt_primes

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
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
L = 4
def mixed():
    qs, ps = [], []
    for bits in (59, 50, 40, 33):
        p = ntt_primes(log2n, 1, bits); qs.append(p.moduli[0]); ps.append(p.psi[0])
    return FheParams(log2n, tuple(qs), tuple(ps))
def shoup55():   # 55-bit primes that no fast class takes (too far below 2^55 for the scaled fold)
    from deeppowers_amd.params import is_prime, min_primitive_2n_root
    n, qs, q = 1 << log2n, [], (1 << 55) - ((1 << 55) - 1) % (2 << log2n)
    while len(qs) < L:
        if is_prime(q) and (((1 << 55) - q) << 5) >= (1 << 24):
            qs.append(q)
        q -= 2 * n
    return FheParams(log2n, tuple(qs), tuple(min_primitive_2n_root(n, v) for v in qs))
sets = (("fold", ntt_primes(log2n, L, 60)), ("f64_30", ntt_primes(log2n, L, 30)), ("f64_45", ntt_primes(log2n, L, 45)), ("fscaled_59", ntt_primes(log2n, L, 59)),
        ("f64wide_49", ntt_primes(log2n, L, 49)), ("shoup_55", shoup55()), ("mixed", mixed()))
res = {}
for name, params in sets:
    ctx = Context(params, 0); ev = Evaluator(ctx)
    N = params.n
    q = torch.tensor(params.moduli, dtype=torch.int64, device=ctx.device).view(1, L, 1)
    g = torch.Generator(device=ctx.device).manual_seed(7)
    npoly = 1024 * 4096 // N
    x = torch.randint(0, 2**62, (npoly, L, N), generator=g, dtype=torch.int64, device=ctx.device) % q
    y = torch.empty_like(x)
    nbytes = 2 * N * 8 * npoly * L
    r = {"classes": ctx.limb_classes}
    for d, fn in (("ntt_fwd", lambda: ev.ntt_forward(x, out=y)), ("ntt_inv", lambda: ev.ntt_inverse(x, out=y))):
        med, mn = timed(fn, 30)
        r[d] = {"median_us": med, "frac": nbytes / med / 8e6}
        print(f"{name:11s} {d}: median {med:7.1f} us  min {mn:7.1f} us  = {nbytes / med / 8e6 * 100:5.1f} % of 8 TB/s   {ctx.limb_classes}", flush=True)
    if log2n <= 13:
        P = pairs * 4096 // N
        a = Ciphertext(torch.randint(0, 2**62, (P, 2, L, N), generator=g, dtype=torch.int64, device=ctx.device) % q.view(1, 1, L, 1))
        b = Ciphertext(torch.randint(0, 2**62, (P, 2, L, N), generator=g, dtype=torch.int64, device=ctx.device) % q.view(1, 1, L, 1))
        o = ctx.empty(P, components=3)
        med, mn = timed(lambda: ev.multiply(a, b, out=o), 10)
        r["ct_mul"] = {"median_us": med, "per_s": P / med * 1e6, "frac": 7 * L * N * 8 * P / med / 8e6}
        print(f"{name:11s} ct_mul x{P}: median {med:8.1f} us -> {P / med:6.3f} M ct-mul/s = {7 * L * N * 8 * P / med / 8e6 * 100:5.1f} % of 8 TB/s", flush=True)
        del a, b, o
    res[name] = r
    del x, y
    ctx.close()
print("CLASS_BENCH " + json.dumps(res))
