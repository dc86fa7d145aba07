"""Where the giant-step key inner products (relin_kernel<..., MODE 4>, dpfhe_switch_key_qp) fetch more than their algorithmic bytes (tool, round 6).
Runs the kernel at N = 8192, 5 + 1 limbs in shapes that separate the three streams - keys (shared by the tokens of a giant step), digits (per item, read by
every limb's workgroup), results - so that FETCH_SIZE per dispatch under `rocprofv3 --pmc` can be differenced:
    (keys, tokens) = (15, 8) the packed layer's shape | (15, 1) no sharing of key tiles | (1, 8) one key | (8, 8) one key per XCD, a single round | (15, 4) | (15, 2)
Each shape launches its own grid size, which is how tools/pmc_summary.py tells them apart.  DPFHE_AB_LIB=<path> selects another build of the library.
Prints the expected read / write bytes per shape next to the timing."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deeppowers_amd._cabi as _cabi
if os.environ.get("DPFHE_AB_LIB"):
    _cabi.LIB_PATH = os.path.abspath(os.environ["DPFHE_AB_LIB"])
from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator
from deeppowers_amd.params import FheParams

tag = os.path.basename(os.environ.get("DPFHE_AB_LIB", "HEAD"))
pe = FheParams.n8192_l6()
ctx = Context(pe, 0); ev = Evaluator(ctx); dev = ctx.device
L, Ld, N = pe.n_limbs, pe.n_limbs - 1, pe.n
g = torch.Generator(device=dev).manual_seed(7)
q = torch.tensor(pe.moduli, dtype=torch.int64, device=dev)
rnd = lambda *shape, limbs: torch.randint(0, 2**62, shape + (limbs, N), generator=g, dtype=torch.int64, device=dev) % q[:limbs].view(*([1] * len(shape)), limbs, 1)
W = N * 8
for K, T in ((15, 8), (15, 1), (1, 8), (8, 8), (15, 4), (15, 2)):
    keys = rnd(K, Ld, 2, limbs=L)
    gin = Ciphertext(rnd(K * T, 2, limbs=Ld))
    for _ in range(2):
        ev.switch_key_qp(gin, keys, T)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for s, e in evs:
        s.record(); ev.switch_key_qp(gin, keys, T); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) * 1e3 for s, e in evs)
    rd_keys, rd_dig, wr = K * Ld * 2 * L * W, K * T * Ld * W, K * T * 2 * L * W
    blocks = (K + 7) // 8 * 8 * L * T if K >= 8 else K * T * L
    print(f"GIANT {tag:14s} keys {K:2d} tokens {T}  median {ts[2]:7.1f} us  expected reads: keys {rd_keys / 2**10:9.1f} KiB + digits {rd_dig / 2**10:9.1f} KiB = {(rd_keys + rd_dig) / 2**10:9.1f} KiB,"
          f" writes {wr / 2**10:9.1f} KiB  (grid ~{blocks * 512} threads)", flush=True)
    del keys, gin
ctx.close()
