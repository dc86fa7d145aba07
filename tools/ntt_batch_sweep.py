"""NTT kernel time vs batch size (fixed overhead vs per-polynomial slope)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppowers_amd.evaluator import Context, Evaluator
from deeppowers_amd.params import FheParams
from tools.ntt_bench import timeit

p = FheParams.n4096_l4(); ctx = Context(p, 0); ev = Evaluator(ctx)
q = torch.tensor(p.moduli, dtype=torch.int64, device=ctx.device).view(1, 4, 1)
for nb in (64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384):
    x = torch.randint(0, 2**62, (nb, 4, 4096), dtype=torch.int64, device=ctx.device) % q
    y = torch.empty_like(x)
    mf, nf = timeit(lambda: ev.ntt_forward(x, out=y), reps=20)
    mi, ni = timeit(lambda: ev.ntt_inverse(x, out=y), reps=20)
    nbytes = 2 * 4096 * 8 * nb * 4
    print(f"batch {nb:6d} RNS polys ({nb*4:6d} residue polys): fwd med {mf:8.1f} min {nf:8.1f} us ({nbytes/nf/8e6*100:5.1f}% peak)   inv med {mi:8.1f} min {ni:8.1f} us ({nbytes/ni/8e6*100:5.1f}% peak)")
