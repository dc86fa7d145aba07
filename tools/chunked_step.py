"""Experiment (tool): the bench step (multiply of 8192 pairs || shard-local reduce) with the multiply launched in K chunks and each chunk's reduce chasing it on the
side stream (the products may still sit in the 256 MiB Infinity Cache when the reduce reads them), against the step as bench.py runs it (K = 1: the reduce of
step i overlaps the multiply of step i + 1).  DPFHE_AB_LIB selects another build (e.g. one whose multiply stores are not non-temporal).
    python tools/chunked_step.py [K ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeppowers_amd._cabi as _cabi
if os.environ.get("DPFHE_AB_LIB"):
    _cabi.LIB_PATH = os.path.abspath(os.environ["DPFHE_AB_LIB"])
from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator
from deeppowers_amd.params import FheParams
from deeppowers_amd.sharding import ShardedMultiplyReduce

tag = os.path.basename(os.environ.get("DPFHE_AB_LIB", "HEAD"))
Ks = [int(v) for v in sys.argv[1:]] or [1, 8, 16, 32]
p = FheParams.n4096_l4(); ctx = Context(p, 0); ev = Evaluator(ctx); dev = ctx.device
L, N, B = p.n_limbs, p.n, 8192
g = torch.Generator(device=dev).manual_seed(11)
q = torch.tensor(p.moduli, dtype=torch.int64, device=dev).view(1, 1, L, 1)
a = Ciphertext(torch.randint(0, 2**62, (B, 2, L, N), generator=g, dtype=torch.int64, device=dev) % q)
b = Ciphertext(torch.randint(0, 2**62, (B, 2, L, N), generator=g, dtype=torch.int64, device=dev) % q)
main = torch.cuda.Stream(device=dev)
if hasattr(ctx, "autotune"):
    pass
pipe = ShardedMultiplyReduce(ev, B, main=main)
outs = pipe.outs

class Chunked:
    def __init__(self, K):
        self.K, self.side = K, torch.cuda.Stream(device=dev)
        self.parts = [torch.empty((K, 3, L, N), dtype=torch.int64, device=dev) for _ in range(2)]
        self.totals = [ctx.empty(components=3) for _ in range(2)]
        self.ev_mul = [[torch.cuda.Event() for _ in range(K)] for _ in range(2)]
        self.red_done = [torch.cuda.Event() for _ in range(2)]
        self.n = 0
        self.bounds = [(c * B // K, (c + 1) * B // K) for c in range(K)]
    def step(self):
        k = self.n & 1; self.n += 1
        main.wait_event(self.red_done[k])
        for c, (lo, hi) in enumerate(self.bounds):
            o = outs[k][lo:hi]
            ev.multiply(Ciphertext(a.data[lo:hi]), Ciphertext(b.data[lo:hi]), out=o, stream=main)
            self.ev_mul[k][c].record(main)
            self.side.wait_event(self.ev_mul[k][c])
            with torch.cuda.stream(self.side):
                ev.reduce_sum(Ciphertext(o), out=self.parts[k][c], stream=self.side)
        with torch.cuda.stream(self.side):
            ev.reduce_sum(Ciphertext(self.parts[k]), out=self.totals[k], stream=self.side)
        self.red_done[k].record(self.side)
        return k

def timed(stepfn, steps=20):
    for _ in range(3):
        stepfn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        k = stepfn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, k

with torch.cuda.stream(main):
    ref_ms, k = timed(lambda: pipe.step(a, b))
    want = pipe.totals[k].clone()
    for rnd in range(3):
        ms, k = timed(lambda: pipe.step(a, b))
        print(f"CHUNK {tag:12s} K=1 (bench step)  {ms:7.3f} ms/step -> {B / ms / 1e3:6.3f} M ct-mul/s", flush=True)
        for K in Ks:
            if K == 1:
                continue
            ch = Chunked(K)
            ms, k = timed(ch.step)
            ok = torch.equal(ch.totals[k], want)
            print(f"CHUNK {tag:12s} K={K:<3d} chased reduce {ms:7.3f} ms/step -> {B / ms / 1e3:6.3f} M ct-mul/s  same_total={ok}", flush=True)
ctx.close()
