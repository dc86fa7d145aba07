"""A/B timing of the bench.py step (8192 ct x ct multiplies + overlapped shard-local reduce) for another build of the
library: DPFHE_AB_LIB=path python tools/step_bench.py [steps=8] [mode=overlap|serial|mulonly]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeppowers_amd._cabi as _cabi  # noqa: E402

if os.environ.get("DPFHE_AB_LIB"):
    _cabi.LIB_PATH = os.path.abspath(os.environ["DPFHE_AB_LIB"])
from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator  # noqa: E402
from deeppowers_amd.params import FheParams  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mode = sys.argv[2] if len(sys.argv) > 2 else "overlap"
params = FheParams.n4096_l4()
L, N, B = params.n_limbs, params.n, 8192
ctx = Context(params, 0); ev = Evaluator(ctx); dev = ctx.device
q = torch.tensor(params.moduli, dtype=torch.int64, device=dev).view(1, 1, L, 1)
a = Ciphertext(torch.randint(0, 2**62, (B, 2, L, N), dtype=torch.int64, device=dev) % q)
b = Ciphertext(torch.randint(0, 2**62, (B, 2, L, N), dtype=torch.int64, device=dev) % q)
outs = [ctx.empty(B, components=3) for _ in range(2)]
parts = [ctx.empty(components=3) for _ in range(2)]
prio = os.environ.get("STEP_PRIO")   # "hi_main": the multiply on a high-priority stream, the reduce on a normal one
main = torch.cuda.Stream(device=dev, priority=-1) if prio == "hi_main" else torch.cuda.current_stream()
side = torch.cuda.Stream(device=dev)
if prio == "hi_main":
    torch.cuda.set_stream(main)
mul_done = [torch.cuda.Event() for _ in range(2)]; red_done = [torch.cuda.Event() for _ in range(2)]


chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 32
cpart = [ctx.empty(chunks, components=3) for _ in range(2)]


def step_chunked(i):
    """multiply and reduce chunk by chunk so that the reduce reads what the multiply just wrote (Infinity Cache)"""
    k = i & 1
    per = B // chunks
    for c in range(chunks):
        sl = slice(c * per, (c + 1) * per)
        o = ev.multiply(Ciphertext(a.data[sl]), Ciphertext(b.data[sl]), out=outs[k][sl], stream=main)
        if mode == "chunked":
            ev.reduce_sum(o, out=cpart[k][c], stream=main)
        else:  # chunked_side: reduce of chunk c on the side stream while chunk c+1 multiplies
            e = torch.cuda.Event(); e.record(main); side.wait_event(e)
            with torch.cuda.stream(side):
                ev.reduce_sum(o, out=cpart[k][c], stream=side)
    if mode == "chunked":
        ev.reduce_sum(Ciphertext(cpart[k]), out=parts[k], stream=main)
    else:
        with torch.cuda.stream(side):
            ev.reduce_sum(Ciphertext(cpart[k]), out=parts[k], stream=side)
        red_done[k].record(side)


mul_streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
chunk_done = [[torch.cuda.Event() for _ in range(64)] for _ in range(2)]
step_start = torch.cuda.Event()


def step_chunked_2s(i):
    """chunks alternate between two multiply streams (the tail of one launch overlaps the head of the next); the reduce of chunk c
    runs on a third stream as soon as chunk c is complete, while its outputs are still in the Infinity Cache"""
    k = i & 1
    per = B // chunks
    step_start.record(main)
    for ms in mul_streams:
        ms.wait_event(step_start)
    side.wait_event(step_start)
    for c in range(chunks):
        sl = slice(c * per, (c + 1) * per)
        ms = mul_streams[c & 1]
        o = ev.multiply(Ciphertext(a.data[sl]), Ciphertext(b.data[sl]), out=outs[k][sl], stream=ms)
        chunk_done[k][c].record(ms)
        side.wait_event(chunk_done[k][c])
        with torch.cuda.stream(side):
            ev.reduce_sum(o, out=cpart[k][c], stream=side)
    with torch.cuda.stream(side):
        ev.reduce_sum(Ciphertext(cpart[k]), out=parts[k], stream=side)
    red_done[k].record(side)
    for ms in mul_streams:
        main.wait_stream(ms)


def step(i):
    if mode == "chunked_2s":
        main.wait_event(red_done[i & 1])
        return step_chunked_2s(i)
    if mode.startswith("chunked"):
        if mode == "chunked_side":
            main.wait_event(red_done[i & 1])
        return step_chunked(i)
    k = i & 1
    main.wait_event(red_done[k])
    if mode == "mulonly1":      # one output buffer instead of two alternating ones (footprint / TLB reach)
        ev.multiply(a, b, out=outs[0], stream=main)
        return
    c = ev.multiply(a, b, out=outs[k], stream=main)
    if mode == "mulonly":
        return
    if mode == "serial":
        ev.reduce_sum(c, out=parts[k], stream=main)
        return
    mul_done[k].record(main)
    side.wait_event(mul_done[k])
    with torch.cuda.stream(side):
        ev.reduce_sum(c, out=parts[k], stream=side)
    red_done[k].record(side)


for i in range(2):
    step(i)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(main)
for i in range(steps):
    step(i)
main.wait_stream(side)
e.record(main)
torch.cuda.synchronize()
ms = s.elapsed_time(e) / steps
print(f"{os.environ.get('TAG', '')} {mode}: {ms:.3f} ms/step  -> {B / ms / 1e3:.3f} M ct-mul/s")
