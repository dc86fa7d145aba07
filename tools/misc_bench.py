"""Streaming kernels: dyadic ops (A3), reduce_sum (A8), ct x pt matvec (A7, BASELINE configs[2])."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeppowers_amd._cabi as _cabi
if os.environ.get("DPFHE_AB_LIB"):
    _cabi.LIB_PATH = os.path.abspath(os.environ["DPFHE_AB_LIB"])
from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, Plaintext
from deeppowers_amd.params import FheParams
from tools.ntt_bench import timeit

p = FheParams.n4096_l4(); ctx = Context(p, 0); ev = Evaluator(ctx)
L, N = 4, 4096
q = torch.tensor(p.moduli, dtype=torch.int64, device=ctx.device).view(1, L, 1)
def rnd(*lead):
    return torch.randint(0, 2**62, (*lead, L, N), dtype=torch.int64, device=ctx.device) % q
nb = 4096  # 512 MiB per operand: beyond the 256 MiB Infinity Cache
a, b, c = rnd(nb), rnd(nb), rnd(nb)
o = torch.empty_like(a)
poly = L * N * 8 * nb
for name, fn, streams in (("dyadic_mul", lambda: ev.dyadic_mul(a, b, out=o), 3), ("dyadic_mul_add", lambda: ev.dyadic_mul_add_(c, a, b), 4),
                          ("add", lambda: ev.add_words(a, b, out=o), 3), ("sub", lambda: ev.sub_words(a, b, out=o), 3), ("negate", lambda: ev.negate_words(a, out=o), 2)):
    med, mn = timeit(fn, reps=15, warm=3)
    print(f"{name:16s} {nb} RNS polys: median {med:8.1f} us  {streams * poly / med / 1e6:7.2f} TB/s = {streams * poly / med / 8e6 * 100:5.1f}% of 8 TB/s")
del a, b, c, o
cts = Ciphertext(rnd(8192, 3))
out = ctx.empty(components=3)
med, mn = timeit(lambda: ev.reduce_sum(cts, out=out), reps=15, warm=3)
print(f"reduce_sum 8192 x 3-comp cts: median {med:8.1f} us  {8192 * 3 * L * N * 8 / med / 1e6:7.2f} TB/s = {8192 * 3 * L * N * 8 / med / 8e6 * 100:5.1f}% of 8 TB/s")
del cts
rows, cols = 768, 64
W = Plaintext(rnd(rows, cols), True)
x = Ciphertext(rnd(cols, 2), True)
y = ctx.empty(rows, components=2)
med, mn = timeit(lambda: ev.matvec_plain(W, x, out=y), reps=8, warm=2)
byts = (rows * cols + cols * 2 + rows * 2) * L * N * 8
print(f"matvec_plain rows={rows} cols={cols} (W {rows*cols*L*N*8/2**30:.1f} GiB): median {med:9.1f} us  {byts / med / 1e6:7.1f} GB/s = {byts / med / 8e6 * 100:5.1f}% of 8 TB/s; {rows*cols*2*L*N/med/1e3:.2f} G mod-FMA/s")

w = torch.randint(0, 2**62, (rows, cols, L), dtype=torch.int64, device=ctx.device) % q.view(1, 1, L)
med, mn = timeit(lambda: ev.matvec_scalar(w, x, out=y), reps=8, warm=2)
print(f"matvec_scalar rows={rows} cols={cols}: median {med:9.1f} us  {rows*cols*2*L*N/med/1e3:.2f} G mod-FMA/s")
