"""Same-box A/B of the giant-step key inner products (dpfhe_switch_key_qp, relin_kernel MODE 4) at N = 8192, L = 6: 15 keys x 8 tokens,
the shape of the QKV layer's giant steps.  Run twice: plain, and with DPFHE_RELIN13_LOGE3=1 (8 words per thread, 4 waves per SIMD)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeppowers_amd._cabi as _cabi  # noqa: E402
if os.environ.get("DPFHE_AB_LIB"):  # A/B experiments: time another build of the library (tool only)
    _cabi.LIB_PATH = os.path.abspath(os.environ["DPFHE_AB_LIB"])
from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator  # noqa: E402
from deeppowers_amd.params import FheParams  # noqa: E402

pe = FheParams.n8192_l6()
ctx = Context(pe, 0)
ev = Evaluator(ctx)
L, Ld, n = pe.n_limbs, pe.n_limbs - 1, pe.n
k, group = 15, 8
g = torch.Generator(device=ctx.device).manual_seed(5)
q = torch.tensor(pe.moduli, dtype=torch.int64, device=ctx.device)
keys = torch.randint(0, 2**62, (k, Ld, 2, L, n), generator=g, dtype=torch.int64, device=ctx.device) % q.view(1, 1, 1, L, 1)
items = torch.randint(0, 2**62, (k * group, 2, Ld, n), generator=g, dtype=torch.int64, device=ctx.device) % q[:Ld].view(1, 1, Ld, 1)
ct = Ciphertext(items)
out = ev.switch_key_qp(ct, keys, group)
torch.cuda.synchronize()
chk = int(out.sum().item()) & 0xFFFFFFFFFFFF
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ev.switch_key_qp(ct, keys, group)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3 / 20)
print(f"RELIN13 {os.path.basename(os.environ.get('DPFHE_AB_LIB', 'HEAD')):14s} us per call (5 x 20): {' '.join(f'{t:.1f}' for t in ts)}  median {np.median(ts):.1f}  checksum {chk:x}")
