"""NTT at BASELINE configs[1] in place (one 128 MiB buffer, what SURVEY.md 8(d) specifies) against out of place (a second buffer, 256 MiB
touched = the size of the Infinity Cache), and at a 1 GiB batch where nothing is cached (tool)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppowers_amd.evaluator import Context, Evaluator
from deeppowers_amd.params import FheParams
p = FheParams.n4096_l4(); ctx = Context(p, 0); ev = Evaluator(ctx)
L, N = 4, 4096
q = torch.tensor(p.moduli, dtype=torch.int64, device=ctx.device).view(1, L, 1)
for nb in (1024, 8192):
    x = torch.randint(0, 2**62, (nb, L, N), dtype=torch.int64, device=ctx.device) % q
    y = torch.empty_like(x)
    def run(fns, reps=30):
        for _ in range(5):
            for _, fn in fns: fn()
        evs = {n: [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)] for n, _ in fns}
        for i in range(reps):
            for n, fn in fns:
                s, e = evs[n][i]; s.record(); fn(); e.record()
        torch.cuda.synchronize()
        out = {}
        for n, _ in fns:
            ts = sorted(s.elapsed_time(e) * 1e3 for s, e in evs[n]); out[n] = (ts[len(ts)//2], ts[0])
        return out
    nbytes = 2 * N * 8 * nb * L
    for label, fns in (("out-of-place", (("fwd", lambda: ev.ntt_forward(x, out=y)), ("inv", lambda: ev.ntt_inverse(x, out=y)))),
                       ("in-place", (("fwd", lambda: ev.ntt_forward_(x)), ("inv", lambda: ev.ntt_inverse_(x))))):
        r = run(fns)
        print(f"nb={nb} {label:13s} fwd median {r['fwd'][0]:7.1f} us ({nbytes/r['fwd'][0]/8e6*100:5.1f} %) min {r['fwd'][1]:7.1f}   inv median {r['inv'][0]:7.1f} us ({nbytes/r['inv'][0]/8e6*100:5.1f} %) min {r['inv'][1]:7.1f}")
