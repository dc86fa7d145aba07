"""Where a workgroup of the batched FORWARD transform spends its life (tool): needs a diagnostic build
    bash tools/ab_variant.sh ntttrace -DDPFHE_NTT_TRACE=1     (what the library launches: 256 threads at N = 4096; at N = 8192 512 threads below 2304 polynomials, halves form above)
    bash tools/ab_variant.sh ntttraceh -DDPFHE_NTT_TRACE=2    (N = 8192 in halves form at every batch size: 256 threads, two sub-transforms through one LDS buffer)
Every workgroup stamps s_memrealtime (100 MHz) at: start, first operand word in registers, all operand words arrived, transform done, stores
issued, stores drained (deeppowers_amd/csrc/kernels_trace.h).  usage: DPFHE_AB_LIB=<build> python tools/ntt_trace.py [n4096|n8192] [rns_polys]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeppowers_amd._cabi as _cabi  # noqa: E402
_cabi.LIB_PATH = os.path.abspath(os.environ.get("DPFHE_AB_LIB", "deeppowers_amd/csrc/build/var_ntttrace.so"))
from deeppowers_amd.evaluator import Context, Evaluator  # noqa: E402
from deeppowers_amd.params import FheParams  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "n8192"
p = FheParams.n4096_l4() if which == "n4096" else FheParams.n8192_l6()
nb = int(sys.argv[2]) if len(sys.argv) > 2 else (1024 if which == "n4096" else 256)
ctx = Context(p, 0)
ev = Evaluator(ctx)
L, N = p.n_limbs, p.n
q = torch.tensor(p.moduli, dtype=torch.int64, device=ctx.device).view(1, L, 1)
x = torch.randint(0, 2**62, (nb, L, N), dtype=torch.int64, device=ctx.device) % q
y = torch.empty_like(x)
for _ in range(20):
    ev.ntt_forward(x, out=y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ev.ntt_forward(x, out=y); e1.record(); torch.cuda.synchronize()
lib = C.CDLL(_cabi.LIB_PATH)
lib.dpfhe_debug_ntt_trace_read.argtypes = [C.c_void_p, C.c_size_t]
buf = np.zeros((65536, 8), np.uint64)
got = lib.dpfhe_debug_ntt_trace_read(buf.ctypes.data, 65536)
assert got == nb * L, got
t = (buf[:got, :6].astype(np.int64) - int(buf[:got, 0].min())) / 100.0      # microseconds
hw = buf[:got, 6]
life = t[:, 5] - t[:, 0]
span = t[:, 5].max()
steady = (t[:, 0] > 0.15 * span) & (t[:, 5] < 0.85 * span)
if steady.sum() < 32:
    steady[:] = True
seg = {"wait for the first operand word": t[:, 1] - t[:, 0], "rest of the operand arrives": t[:, 2] - t[:, 1], "transform + canonicalise": t[:, 3] - t[:, 2],
       "stores issued": t[:, 4] - t[:, 3], "stores drained": t[:, 5] - t[:, 4]}
tag = os.path.basename(_cabi.LIB_PATH)
print(f"NTTTRACE {tag} {which}: {got} workgroups, launch {e0.elapsed_time(e1) * 1e3:.1f} us (traced build), span {span:.1f} us, {int(steady.sum())} steady-state workgroups; "
      f"lifetime median {np.median(life[steady]):.2f} us (p10 {np.percentile(life[steady], 10):.2f}, p90 {np.percentile(life[steady], 90):.2f})")
for nm, v in seg.items():
    print(f"NTTTRACE   {nm:34s} median {np.median(v[steady]):6.2f} us = {100 * v[steady].mean() / life[steady].mean():5.1f} % of the lifetime")
ph = np.exp(2j * np.pi * t[steady, 0] / np.median(life[steady]))
pts = np.linspace(0.2 * span, 0.8 * span, 200)
alive = [int(((t[:, 0] <= v) & (t[:, 5] > v)).sum()) for v in pts]
cus = len(set((((hw >> np.uint64(8)) & np.uint64(0xff)).astype(np.int64) | (((hw >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64) << 8)).tolist()))
print(f"NTTTRACE   start-phase concentration R = {abs(ph.mean()):.3f} (0 = de-phased, 1 = lockstep); resident workgroups mean {np.mean(alive):.0f} (min {min(alive)}, max {max(alive)}) on {cus} CUs "
      f"= {np.mean(alive) / max(cus, 1):.2f} per CU")
ctx.close()
