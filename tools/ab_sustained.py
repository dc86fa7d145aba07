"""Same-box A/B of the batched NTT kernels by the sustained method of bench.py (tool): >= `seconds` of back-to-back launches per entry,
one HIP event pair around the whole window.  Each arm runs in its own process (the library is chosen at import):
    python tools/ab_sustained.py [seconds=1.5]                          # the in-tree library
    DPFHE_AB_LIB=deeppowers_amd/csrc/build/var_x.so python tools/ab_sustained.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeppowers_amd._cabi as _cabi
if os.environ.get("DPFHE_AB_LIB"):
    _cabi.LIB_PATH = os.path.abspath(os.environ["DPFHE_AB_LIB"])
from deeppowers_amd.evaluator import Context, Evaluator
from deeppowers_amd.params import FheParams

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
tag = os.path.basename(os.environ.get("DPFHE_AB_LIB", "HEAD"))
for name, params, nb in (("n4096 configs[1]", FheParams.n4096_l4(), 1024), ("n4096 1 GiB oop", FheParams.n4096_l4(), 8192), ("n8192", FheParams.n8192_l6(), 256)):
    ctx = Context(params, 0); ev = Evaluator(ctx)
    L, N = params.n_limbs, params.n
    q = torch.tensor(params.moduli, dtype=torch.int64, device=ctx.device).view(1, L, 1)
    x = torch.randint(0, 2**62, (nb, L, N), dtype=torch.int64, device=ctx.device) % q
    y = torch.empty_like(x) if nb > 1024 else x
    nbytes = 2 * N * 8 * nb * L
    for d, fn in (("fwd", lambda: ev.ntt_forward(x, out=y) if y is not x else ev.ntt_forward_(x)), ("inv", lambda: ev.ntt_inverse(x, out=y) if y is not x else ev.ntt_inverse_(x))):
        for _ in range(20):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fn()
        e.record(); torch.cuda.synchronize()
        n = max(20, int(seconds / (s.elapsed_time(e) * 1e-3 / 10)))
        s.record()
        for _ in range(n):
            fn()
        e.record(); torch.cuda.synchronize()
        per = s.elapsed_time(e) * 1e3 / n
        print(f"{tag:18s} {name:18s} ntt_{d}: {per:8.2f} us/launch = {nbytes / per / 8e6 * 100:5.1f} % of 8 TB/s  ({n} launches)")
    ctx.close()
