#!/bin/bash
# round 6, GPU call X: ct_mul_dual_kernel with NTT-domain output on the lazy tensor step (FoldArith): parity of every domain combination, timing at N = 8192
OUT=gpurun_out/r06x; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_limb_classes.py -q -p no:cacheprovider -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" | tail -3 | tee $OUT/pytest_subset.txt
python - <<'PY' 2>&1 | tee $OUT/out_ntt_n8192.txt
import torch
from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator
from deeppowers_amd.params import FheParams
for name, p, nb in (("n4096_l4", FheParams.n4096_l4(), 4096), ("n8192_l6", FheParams.n8192_l6(), 1024)):
    ctx = Context(p, 0); ev = Evaluator(ctx); dev = ctx.device; L, N = p.n_limbs, p.n
    g = torch.Generator(device=dev).manual_seed(3)
    q = torch.tensor(p.moduli, dtype=torch.int64, device=dev).view(1, 1, L, 1)
    a = Ciphertext(torch.randint(0, 2**62, (nb, 2, L, N), generator=g, dtype=torch.int64, device=dev) % q)
    b = Ciphertext(torch.randint(0, 2**62, (nb, 2, L, N), generator=g, dtype=torch.int64, device=dev) % q)
    o = ctx.empty(nb, components=3)
    for out_ntt in (False, True):
        for _ in range(3): ev.multiply(a, b, out=o, out_ntt=out_ntt)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(9)]
        for s, e in evs:
            s.record(); ev.multiply(a, b, out=o, out_ntt=out_ntt); e.record()
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) * 1e3 for s, e in evs)
        print(f"OUTNTT {name} out_ntt={out_ntt}: median {ts[4]:8.1f} us -> {nb / ts[4]:6.3f} M pairs/s")
    ctx.close()
PY
