"""Same-box A/B timing (tool): NTT fwd/inv at BASELINE configs[1] and ct_mul at 8192 pairs, launches enqueued back to back with one
HIP-event pair each (bench.py's method).  DPFHE_AB_LIB=<path to another build of libdpfhe_hip.so> selects the arm."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeppowers_amd._cabi as _cabi
if os.environ.get("DPFHE_AB_LIB"):
    _cabi.LIB_PATH = os.path.abspath(os.environ["DPFHE_AB_LIB"])
from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator
from deeppowers_amd.params import FheParams

def timed(fns, reps):
    for _ in range(3):
        for _, fn in fns: fn()
    evs = {n: [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)] for n, _ in fns}
    for i in range(reps):
        for n, fn in fns:
            s, e = evs[n][i]; s.record(); fn(); e.record()
    torch.cuda.synchronize()
    out = {}
    for n, _ in fns:
        ts = sorted(s.elapsed_time(e) * 1e3 for s, e in evs[n])
        out[n] = (ts[len(ts) // 2], ts[0])
    return out

tag = os.path.basename(os.environ.get("DPFHE_AB_LIB", "HEAD"))
for name, params, nb, cb in (("n4096", FheParams.n4096_l4(), 1024, 8192), ("n8192", FheParams.n8192_l6(), 256, 1024)):
    ctx = Context(params, 0); ev = Evaluator(ctx)
    L, N = params.n_limbs, params.n
    q = torch.tensor(params.moduli, dtype=torch.int64, device=ctx.device).view(1, L, 1)
    g = torch.Generator(device=ctx.device).manual_seed(5)
    x = torch.randint(0, 2**62, (nb, L, N), generator=g, dtype=torch.int64, device=ctx.device) % q
    y = torch.empty_like(x)
    r = timed((("fwd", lambda: ev.ntt_forward(x, out=y)), ("inv", lambda: ev.ntt_inverse(x, out=y))), 40)
    nbytes = 2 * N * 8 * nb * L
    for d in ("fwd", "inv"):
        print(f"{tag:16s} {name} ntt_{d}: median {r[d][0]:7.1f} us  min {r[d][0 + 1]:7.1f} us  = {nbytes / r[d][0] / 8e6 * 100:5.1f}% / {nbytes / r[d][1] / 8e6 * 100:5.1f}% of 8 TB/s")
    a = Ciphertext(torch.randint(0, 2**62, (cb, 2, L, N), generator=g, dtype=torch.int64, device=ctx.device) % q.view(1, 1, L, 1))
    b = Ciphertext(torch.randint(0, 2**62, (cb, 2, L, N), generator=g, dtype=torch.int64, device=ctx.device) % q.view(1, 1, L, 1))
    o = ctx.empty(cb, components=3)
    r = timed((("mul", lambda: ev.multiply(a, b, out=o)),), 12)
    print(f"{tag:16s} {name} ct_mul x{cb}: median {r['mul'][0]:8.1f} us min {r['mul'][1]:8.1f} us -> {cb / r['mul'][0]:6.3f} M ct-mul/s  checksum {int(o.data.sum().item()) & 0xffffffff:x}")
    # relinearisation (RNS-digit keys) on the products, and the hybrid key-switch inner product (special prime = last limb)
    rb = min(cb, 2048)
    evk = torch.randint(0, 2**62, (L, 2, L, N), generator=g, dtype=torch.int64, device=ctx.device) % q.view(1, 1, L, 1)
    o2 = ctx.empty(rb, components=2)
    c3 = Ciphertext(o.data[:rb].contiguous())
    r = timed((("relin", lambda: ev.relinearize(c3, evk, out=o2)),), 12)
    print(f"{tag:16s} {name} relinearize x{rb}: median {r['relin'][0]:8.1f} us min {r['relin'][1]:8.1f} us  checksum {int(o2.sum().item()) & 0xffffffff:x}")
    Ld = L - 1
    key = torch.randint(0, 2**62, (Ld, 2, L, N), generator=g, dtype=torch.int64, device=ctx.device) % q.view(1, 1, L, 1)
    qd = q.view(1, 1, L, 1)[:, :, :Ld]
    ctd = Ciphertext((torch.randint(0, 2**62, (rb, 2, Ld, N), generator=g, dtype=torch.int64, device=ctx.device) % qd).contiguous())
    r = timed((("ks", lambda: ev.keyswitch_hybrid(ctd, key)),), 12)
    ks = ev.keyswitch_hybrid(ctd, key)
    print(f"{tag:16s} {name} keyswitch_hybrid x{rb}: median {r['ks'][0]:8.1f} us min {r['ks'][1]:8.1f} us  checksum {int(ks.data.sum().item()) & 0xffffffff:x}")
    del a, b, o, x, y, c3, o2, ctd, ks
    ctx.close()
