"""N = 16384 forward / inverse transform time against the batch size (tool): the library under DPFHE_AB_LIB (or HEAD), 3 limbs of the pinned chain, out of place.
    python tools/ntt14_batch_sweep.py [rns polys ...]      prints SWEEP14 lines; the parity of both forms is tests/test_gpu_parity.py's"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeppowers_amd._cabi as _cabi
if os.environ.get("DPFHE_AB_LIB"):
    _cabi.LIB_PATH = os.path.abspath(os.environ["DPFHE_AB_LIB"])
from deeppowers_amd.evaluator import Context, Evaluator
from deeppowers_amd.params import ntt_primes

tag = os.path.basename(os.environ.get("DPFHE_AB_LIB", "HEAD"))
sizes = [int(v) for v in sys.argv[1:]] or [16, 64, 128, 256, 512, 1024, 2048]
p = ntt_primes(14, 3)
ctx = Context(p, 0); ev = Evaluator(ctx); dev = ctx.device
L, N = p.n_limbs, p.n
q = torch.tensor(p.moduli, dtype=torch.int64, device=dev).view(1, L, 1)
g = torch.Generator(device=dev).manual_seed(11)
for nb in sizes:
    x = torch.randint(0, 2**62, (nb, L, N), generator=g, dtype=torch.int64, device=dev) % q
    y = torch.empty_like(x)
    nbytes = 2 * N * 8 * nb * L
    row = []
    for name, fn in (("fwd", lambda: ev.ntt_forward(x, out=y)), ("inv", lambda: ev.ntt_inverse(x, out=y))):
        for _ in range(3):
            fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
        for s, e in evs:
            s.record(); fn(); e.record()
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) * 1e3 for s, e in evs)
        row.append(f"{name} {ts[7]:8.1f} us = {nbytes / ts[7] / 8e6 * 100:5.1f} %")
    print(f"SWEEP14 {tag:16s} {nb:5d} RNS polys x {L} limbs ({nb * L:5d} workgroups)  " + "   ".join(row), flush=True)
    del x, y
ctx.close()
