"""forward NTT only, configs[1] and a steady-state batch (tool; DPFHE_AB_LIB selects the build)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeppowers_amd._cabi as _cabi
if os.environ.get("DPFHE_AB_LIB"): _cabi.LIB_PATH = os.path.abspath(os.environ["DPFHE_AB_LIB"])
from deeppowers_amd.evaluator import Context, Evaluator
from deeppowers_amd.params import FheParams
params = FheParams.n4096_l4(); ctx = Context(params, 0); ev = Evaluator(ctx)
L, N = 4, 4096
q = torch.tensor(params.moduli, dtype=torch.int64, device=ctx.device).view(1, L, 1)
for nb in (1024, 8192):
    x = torch.randint(0, 2**62, (nb, L, N), dtype=torch.int64, device=ctx.device) % q
    y = torch.empty_like(x)
    for _ in range(5): ev.ntt_forward(x, out=y)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for s, e in evs:
        s.record(); ev.ntt_forward(x, out=y); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) * 1e3 for s, e in evs)
    nbytes = 2 * N * 8 * nb * L
    print(f"{os.path.basename(os.environ.get('DPFHE_AB_LIB','HEAD')):14s} nb={nb}: median {ts[15]:.1f} us min {ts[0]:.1f} us = {nbytes/ts[15]/8e6*100:.1f}% / {nbytes/ts[0]/8e6*100:.1f}%  chk {int(y.sum().item()) & 0xffffffff:x}")
