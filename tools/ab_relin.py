"""Same-box A/B of the key-switching kernels on the pinned primes (tool): relinearisation (RNS-digit keys) at N = 4096, L = 4 (2048 ciphertexts) and N = 8192,
L = 6 (512), hoisted rotations (31 rotations x 8 tokens at N = 8192, 5 + 1 limbs).  DPFHE_AB_LIB=<path to another build of libdpfhe_hip.so> selects the arm."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeppowers_amd._cabi as _cabi
if os.environ.get("DPFHE_AB_LIB"):
    _cabi.LIB_PATH = os.path.abspath(os.environ["DPFHE_AB_LIB"])
from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator
from deeppowers_amd.params import FheParams

tag = os.path.basename(os.environ.get("DPFHE_AB_LIB", "HEAD"))

def timed(fn, reps=9):
    for _ in range(3):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) * 1e3 for s, e in evs)
    return ts[len(ts) // 2], ts[0]

for name, p, nb in (("n4096_l4", FheParams.n4096_l4(), 2048), ("n8192_l6", FheParams.n8192_l6(), 512)):
    ctx = Context(p, 0); ev = Evaluator(ctx); dev = ctx.device
    L, N = p.n_limbs, p.n
    g = torch.Generator(device=dev).manual_seed(3)
    q = torch.tensor(p.moduli, dtype=torch.int64, device=dev)
    rnd = lambda *shape: torch.randint(0, 2**62, (*shape, L, N), generator=g, dtype=torch.int64, device=dev) % q.view(*([1] * len(shape)), L, 1)
    c3 = Ciphertext(rnd(nb, 3)); evk = rnd(L, 2); o2 = ctx.empty(nb, components=2)
    med, mn = timed(lambda: ev.relinearize(c3, evk, out=o2))
    print(f"ABRELIN {tag:12s} {name} relinearize x{nb}: median {med:8.1f} us min {mn:8.1f} -> {nb / med:6.3f} M/s  checksum {int(o2.data.sum().item()) & 0xffffffff:x}", flush=True)
    if name == "n8192_l6":
        Ld, T, k = L - 1, 8, 31
        qd = q[:Ld]
        cts = Ciphertext(torch.randint(0, 2**62, (T, 2, Ld, N), generator=g, dtype=torch.int64, device=dev) % qd.view(1, 1, Ld, 1))
        keys = rnd(k, Ld, 2)
        elts = [pow(3, i + 1, 2 * N) for i in range(k)]
        out = ev.rotate_hybrid_hoisted(cts, elts, keys)
        med, mn = timed(lambda: ev.rotate_hybrid_hoisted(cts, elts, keys))
        print(f"ABRELIN {tag:12s} {name} rotate_hybrid_hoisted {k}x{T}: median {med:8.1f} us min {mn:8.1f}  checksum {int(out.data.sum().item()) & 0xffffffff:x}", flush=True)
        ks = Ciphertext(torch.randint(0, 2**62, (64, 2, Ld, N), generator=g, dtype=torch.int64, device=dev) % qd.view(1, 1, Ld, 1))
        key1 = rnd(Ld, 2)
        out = ev.keyswitch_hybrid(ks, key1)
        med, mn = timed(lambda: ev.keyswitch_hybrid(ks, key1))
        print(f"ABRELIN {tag:12s} {name} keyswitch_hybrid x64: median {med:8.1f} us min {mn:8.1f}  checksum {int(out.data.sum().item()) & 0xffffffff:x}", flush=True)
    ctx.close()
