"""Same-box timing of the three forms of the fused multiply (tool): alone (12 launches, one event pair each, median / min) and inside
the bench step (multiply || reduce; 6 steps per form, twice).  usage: python tools/ab_forms.py [pairs=8192]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeppowers_amd._cabi as _cabi  # noqa: E402
if os.environ.get("DPFHE_AB_LIB"):
    _cabi.LIB_PATH = os.path.abspath(os.environ["DPFHE_AB_LIB"])
from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator  # noqa: E402
from deeppowers_amd.params import FheParams  # noqa: E402
from deeppowers_amd.sharding import ShardedMultiplyReduce  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
tag = os.path.basename(os.environ.get("DPFHE_AB_LIB", "HEAD")) + " loop_pairs=" + os.environ.get("DPFHE_CTMUL_LOOP_PAIRS", "default")
for params in ((FheParams.n4096_l4(),) if len(sys.argv) > 2 and sys.argv[2] == "n4096" else (FheParams.n4096_l4(), FheParams.n8192_l6())):
    B = pairs if params.log2_n == 12 else pairs // 8
    ctx = Context(params, 0)
    ev = Evaluator(ctx)
    L, N = params.n_limbs, params.n
    dev = ctx.device
    print(f"# N={N} L={L} {B} pairs; probe at context creation: {ctx.tune_info()}")
    g = torch.Generator(device=dev).manual_seed(5)
    q = torch.tensor(params.moduli, dtype=torch.int64, device=dev).view(1, 1, L, 1)
    a = Ciphertext(torch.randint(0, 2**62, (B, 2, L, N), generator=g, dtype=torch.int64, device=dev) % q)
    b = Ciphertext(torch.randint(0, 2**62, (B, 2, L, N), generator=g, dtype=torch.int64, device=dev) % q)
    pipe = ShardedMultiplyReduce(ev, B)
    print(f"# dpfhe_ctx_autotune on {pipe.outs[1].numel()} words of scratch: {ctx.autotune(pipe.outs[1].view(-1), 3)}")
    ref = None
    for rnd in range(2):
        for form in (ctx.variants() if rnd == 0 else ctx.variants()[::-1]):
            ctx.set_ct_mul_variant(form)
            o = pipe.outs[0]
            for _ in range(2):
                ev.multiply(a, b, out=o)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
            for s_, e_ in evs:
                s_.record(); ev.multiply(a, b, out=o); e_.record()
            torch.cuda.synchronize()
            ts = sorted(s_.elapsed_time(e_) for s_, e_ in evs)
            chk = int(o.view(-1)[:: 4099].sum().item())
            ref = chk if ref is None else ref
            pipe.step(a, b); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(6):
                pipe.step(a, b)
            torch.cuda.synchronize()
            st = (time.perf_counter() - t0) / 6 * 1e3
            print(f"{tag} pass {rnd} {form:8s} alone: median {ts[6]:7.3f} ms  min {ts[0]:7.3f} ms = {B / ts[6] / 1e3:6.3f} M ct-mul/s | in the step: {st:7.3f} ms = {B / st / 1e3:6.3f} M ct-mul/s | same words: {chk == ref}")
    del pipe, a, b
    ctx.close()
