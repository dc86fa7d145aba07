"""Where a workgroup of the giant steps' key-inner-product kernel (relin_kernel MODE 4, N = 8192, L = 6) spends its life: needs the diagnostic
build  bash tools/ab_variant.sh reltrace -DDPFHE_RELIN_TRACE  (s_memrealtime stamps held in scalar registers, 8 words per workgroup).
15 keys x 8 tokens, the shape of the QKV layer's giant steps."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeppowers_amd._cabi as _cabi  # noqa: E402
_cabi.LIB_PATH = os.path.abspath(os.environ.get("DPFHE_AB_LIB", "deeppowers_amd/csrc/build/var_reltrace.so"))
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
for _ in range(3):
    ev.switch_key_qp(ct, keys, group)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ev.switch_key_qp(ct, keys, group); e1.record(); torch.cuda.synchronize()
lib = C.CDLL(_cabi.LIB_PATH)
lib.dpfhe_debug_relin_trace_read.argtypes = [C.c_void_p, C.c_size_t]
buf = np.zeros((4096, 8), np.uint64)
nb = lib.dpfhe_debug_relin_trace_read(buf.ctypes.data, 4096)
assert nb > 0, nb
t = buf[:nb].astype(np.float64)
t = t[t[:, 1] > 0]            # padding workgroups of the XCD-aligned grid return before the loop
tick = 0.01                   # s_memrealtime: 100 MHz
life = (t[:, 1] - t[:, 0]) * tick
names = ["digit words arrived", "forward transform", "first key tile arrived", "its products", "second key tile arrived", "its products"]
print(f"RELINTRACE entry {e0.elapsed_time(e1) * 1e3:.1f} us (traced build), {len(t)} workgroups; digit loop per workgroup: median {np.median(life):.2f} us  (p10 {np.percentile(life, 10):.2f}, p90 {np.percentile(life, 90):.2f})")
for i, nm in enumerate(names):
    v = t[:, 2 + i] * tick
    print(f"RELINTRACE   {nm:26s} median {np.median(v):6.2f} us  = {100 * np.median(v) / np.median(life):5.1f} % of the loop   (sum over the 5 digits)")
span = (t[:, 1].max() - t[:, 0].min()) * tick
print(f"RELINTRACE first start -> last end {span:.1f} us; workgroup-microseconds / (256 CUs x span) = {life.sum() / (256 * span):.2f} workgroups resident per CU on average")
