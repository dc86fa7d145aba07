"""N = 8192, L = 6: forward / inverse transform time against the batch size (tool; same-box A/B of library builds with DPFHE_AB_LIB).  Launches are enqueued back to
back, one HIP-event pair around `reps` of them, per-launch time reported; out of place."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeppowers_amd._cabi as _cabi
if os.environ.get("DPFHE_AB_LIB"):
    _cabi.LIB_PATH = os.path.abspath(os.environ["DPFHE_AB_LIB"])
from deeppowers_amd.evaluator import Context, Evaluator
from deeppowers_amd.params import FheParams

tag = os.path.basename(os.environ.get("DPFHE_AB_LIB", "HEAD"))
p = FheParams.n8192_l6(); ctx = Context(p, 0); ev = Evaluator(ctx)
L, N = p.n_limbs, p.n
q = torch.tensor(p.moduli, dtype=torch.int64, device=ctx.device).view(1, L, 1)
for nb in (int(a) for a in (sys.argv[1:] or ["40", "128", "256", "384", "512", "1024", "2048"])):
    x = torch.randint(0, 2**62, (nb, L, N), dtype=torch.int64, device=ctx.device) % q
    y = torch.empty_like(x)
    row = []
    for fn in (lambda: ev.ntt_forward(x, out=y), lambda: ev.ntt_inverse(x, out=y)):
        for _ in range(5):
            fn()
        reps = max(10, min(200, int(20000 / max(nb, 1))))
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
        row.append(best)
    nbytes = 2 * N * 8 * nb * L
    print(f"SWEEP13 {tag:16s} {nb:5d} RNS polys ({nb * L:6d} workgroups): fwd {row[0]:8.1f} us = {nbytes / row[0] / 8e6 * 100:5.1f} %   inv {row[1]:8.1f} us = {nbytes / row[1] / 8e6 * 100:5.1f} %")
    del x, y
ctx.close()
