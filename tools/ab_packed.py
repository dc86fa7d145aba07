"""Same-box A/B timing of the packed layer's stages (tool): the plaintext product, the baby-step pass and the giant steps' key inner
products at N = 8192, 5 + 1 limbs, 8 tokens, 64 x 16 split.  DPFHE_AB_LIB=<path to another build of libdpfhe_hip.so> selects the arm."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeppowers_amd._cabi as _cabi
if os.environ.get("DPFHE_AB_LIB"):
    _cabi.LIB_PATH = os.path.abspath(os.environ["DPFHE_AB_LIB"])
from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, Plaintext
from deeppowers_amd.params import FheParams

tag = os.path.basename(os.environ.get("DPFHE_AB_LIB", "HEAD"))
pe = FheParams.n8192_l6()
ctx = Context(pe, 0); ev = Evaluator(ctx); dev = ctx.device
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n1 = int(sys.argv[2]) if len(sys.argv) > 2 else 64
L, Ld, N, n2 = pe.n_limbs, pe.n_limbs - 1, pe.n, 1024 // n1
g = torch.Generator(device=dev).manual_seed(7)
q = torch.tensor(pe.moduli, dtype=torch.int64, device=dev)
rnd = lambda *shape, limbs: torch.randint(0, 2**62, shape + (limbs, N), generator=g, dtype=torch.int64, device=dev) % q[:limbs].view(*([1] * len(shape)), limbs, 1)

def timed(fn, reps=7):
    fn(); fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) * 1e3 for s, e in evs)
    return ts[len(ts) // 2], ts[0]

keys = rnd(n1 - 1, Ld, 2, limbs=L)
xin = Ciphertext(rnd(T, 2, limbs=Ld))
elts = [pow(3, j + 1, 2 * N) for j in range(n1 - 1)]
diag = Plaintext(rnd(n2, n1, limbs=L), True)
babies = ev.rotate_hoisted_qp(xin, elts, keys)
print(f"{tag:16s} rotate_hoisted_qp  median %8.1f us  min %8.1f us  checksum %x" % (*timed(lambda: ev.rotate_hoisted_qp(xin, elts, keys)), int(babies.sum().item()) & 0xffffffff))
inner = ev.matvec_plain_multi(diag, babies, T)
print(f"{tag:16s} matvec_plain_multi median %8.1f us  min %8.1f us  checksum %x" % (*timed(lambda: ev.matvec_plain_multi(diag, babies, T)), int(inner.sum().item()) & 0xffffffff))
ielts = [1] + [pow(3, n1 * i, 2 * N) for i in range(1, n2)]
print(f"{tag:16s} ntt_inverse_galois median %8.1f us  min %8.1f us" % timed(lambda: ev.ntt_inverse_galois(inner, ielts, out=inner)))
print(f"{tag:16s} rescale            median %8.1f us  min %8.1f us" % timed(lambda: ev.rescale_words(inner)))
rot = ev.rescale_words(inner)
gkeys = rnd(n2 - 1, Ld, 2, limbs=L)
gin = Ciphertext(rot[1:].reshape((n2 - 1) * T, 2, Ld, N))
terms = ev.switch_key_qp(gin, gkeys, T)
print(f"{tag:16s} switch_key_qp      median %8.1f us  min %8.1f us  checksum %x" % (*timed(lambda: ev.switch_key_qp(gin, gkeys, T)), int(terms.sum().item()) & 0xffffffff))
ksum = torch.empty((T, 2, L, N), dtype=torch.int64, device=dev)
def tail():
    _cabi.check(ctx._lib.dpfhe_reduce_sum(ctx.handle, ksum.data_ptr(), terms.data_ptr(), n2 - 1, T * 2, ev._sp(None)), "dpfhe_reduce_sum")
    ev.ntt_inverse_(ksum)
    return ev.rescale_bsgs(ksum, rot)
print(f"{tag:16s} reduce+intt+bsgs   median %8.1f us  min %8.1f us" % timed(tail))
print(f"{tag:16s} T={T} split {n1}x{n2}")
x1 = torch.randint(0, 2**62, (1 << 27,), generator=g, dtype=torch.int64, device=dev); y1 = torch.empty_like(x1)
m, mn = timed(lambda: ev.device_copy(y1, x1))
print(f"{tag:16s} dpfhe_copy 1 GiB   median %8.1f us  = %.2f TB/s" % (m, 2 * x1.numel() * 8 / m / 1e6))
ctx.close()
