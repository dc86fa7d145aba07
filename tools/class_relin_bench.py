"""Relinearisation throughput per limb class at N = 4096, L = 4 (tool): 2048 three-component ciphertexts, RNS-digit keys - the key switch that follows the
metric op - on contexts whose limbs all share one class (the class's own key-switching kernels, dpfhe_cabi.hip with_policy) and on the generic class."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator
from deeppowers_amd.params import FheParams, is_prime, min_primitive_2n_root, ntt_primes

def shoup55():
    qs, q = [], (1 << 55) - ((1 << 55) - 1) % 8192
    while len(qs) < 4:
        if is_prime(q) and (((1 << 55) - q) << 5) >= (1 << 24):
            qs.append(q)
        q -= 8192
    return FheParams(12, tuple(qs), tuple(min_primitive_2n_root(4096, v) for v in qs))

nb, L, N = 2048, 4, 4096
for name, p in (("fold", ntt_primes(12, L, 60)), ("f64_31", ntt_primes(12, L, 31)), ("f64_45", ntt_primes(12, L, 45)), ("f64wide_49", ntt_primes(12, L, 49)),
                ("fscaled_59", ntt_primes(12, L, 59)), ("shoup_55", shoup55()), ("mixed", FheParams.generic_n4096_l4()), ("fold", ntt_primes(12, L, 60))):
    ctx = Context(p, 0); ev = Evaluator(ctx); dev = ctx.device
    g = torch.Generator(device=dev).manual_seed(3)
    q = torch.tensor(p.moduli, dtype=torch.int64, device=dev)
    c3 = Ciphertext(torch.randint(0, 2**62, (nb, 3, L, N), generator=g, dtype=torch.int64, device=dev) % q.view(1, 1, L, 1))
    evk = torch.randint(0, 2**62, (L, 2, L, N), generator=g, dtype=torch.int64, device=dev) % q.view(1, 1, L, 1)
    o2 = ctx.empty(nb, components=2)
    for _ in range(3):
        ev.relinearize(c3, evk, out=o2)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
    for s, e in evs:
        s.record(); ev.relinearize(c3, evk, out=o2); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) * 1e3 for s, e in evs)
    print(f"RELIN {name:11s} {'/'.join(sorted(set(ctx.limb_classes))):20s} median {ts[3]:8.1f} us -> {nb / ts[3]:6.3f} M relin/s", flush=True)
    ctx.close()
