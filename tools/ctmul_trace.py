"""Where a workgroup of the fused multiply spends its life (tool; diagnostics of the "memory-parked" regime, MEASUREMENTS.md section 5).

Launches dpfhe_debug_ct_mul_trace (the quad form with s_memrealtime stamps at its milestones, one record per workgroup) over `pairs`
ciphertext pairs at N=4096 / L=4 and prints, per segment, median / p10 / p90 microseconds over all workgroups of the steady state:
  wait_first_load   start -> first operand word in registers          (HBM latency under the kernel's own load)
  forward           4 forward transforms (FwdChain4)                    (includes waiting for the other three operands)
  product           tensor product in registers
  inverse           3 inverse transforms (InvChain3)
  store_issue       canonicalise + issue the 48 stores per thread
  store_drain       s_waitcnt vmcnt(0): until the last store is acknowledged
plus the lifetime, how many workgroups are resident over time, and how de-phased the workgroup starts are (R = |mean exp(2 pi i
start / median lifetime)|: 0 = uniformly spread, 1 = lockstep generations).  usage: python tools/ctmul_trace.py [pairs=2048] [json]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppowers_amd import _cabi  # noqa: E402
from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator  # noqa: E402
from deeppowers_amd.params import FheParams  # noqa: E402


def run(pairs=2048, reps=3):
    p = FheParams.n4096_l4()
    ctx = Context(p, 0)
    ev = Evaluator(ctx)
    lib = ctx._lib
    L, N = p.n_limbs, p.n
    dev = ctx.device
    g = torch.Generator(device=dev).manual_seed(3)
    q = torch.tensor(p.moduli, dtype=torch.int64, device=dev).view(1, 1, L, 1)
    a = torch.randint(0, 2**62, (pairs, 2, L, N), generator=g, dtype=torch.int64, device=dev) % q
    b = torch.randint(0, 2**62, (pairs, 2, L, N), generator=g, dtype=torch.int64, device=dev) % q
    o = ctx.empty(pairs, components=3)
    o2 = ctx.empty(pairs, components=3)
    trace = torch.zeros(pairs * L * 12, dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream(dev).cuda_stream

    def traced():
        _cabi.check(lib.dpfhe_debug_ct_mul_trace(ctx.handle, o.data_ptr(), a.data_ptr(), b.data_ptr(), pairs, trace.data_ptr(), s), "trace")

    def timed(fn):
        fn(); fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    ctx.set_ct_mul_variant("quad")
    plain = lambda: ev.multiply(Ciphertext(a), Ciphertext(b), out=o2)
    for _ in range(6):   # both forms on a warm chip, alternating
        plain(); traced()
    t_plain, t_trace = timed(plain), timed(traced)
    t_plain, t_trace = min(t_plain, timed(plain)), min(t_trace, timed(traced))
    same = bool(torch.equal(o, o2))
    tr = trace.cpu().numpy().view(np.uint64).reshape(pairs * L, 12)
    fine = (tr[:, 8:12].astype(np.int64) - tr[:, 0:1].astype(np.int64)) / 100.0   # prologue / x issued / all issued / x complete, us after start
    t = tr[:, :7].astype(np.int64)
    t0 = t[:, 0].min()
    t = (t - t0) / 100.0                                    # 100 MHz -> microseconds
    life = t[:, 6] - t[:, 0]
    seg = {"wait_first_load": t[:, 1] - t[:, 0], "forward": t[:, 2] - t[:, 1], "product": t[:, 3] - t[:, 2], "inverse": t[:, 4] - t[:, 3],
           "store_issue": t[:, 5] - t[:, 4], "store_drain": t[:, 6] - t[:, 5], "lifetime": life}
    span = t[:, 6].max()
    steady = (t[:, 0] > 0.15 * span) & (t[:, 6] < 0.85 * span)   # drop the first generation (all start together) and the tail
    if steady.sum() < 64:
        steady[:] = True
    out = {"pairs": pairs, "workgroups": int(pairs * L), "kernel_us_plain": t_plain, "kernel_us_traced": t_trace, "traced_equals_plain_output": same,
           "span_us": float(span), "steady_workgroups": int(steady.sum()), "segments_us": {}}
    for k, v in seg.items():
        w = v[steady]
        out["segments_us"][k] = {"median": float(np.median(w)), "p10": float(np.percentile(w, 10)), "p90": float(np.percentile(w, 90)), "mean": float(w.mean())}
    out["first_operand_us_after_start"] = {nm: {"median": float(np.median(fine[steady, i])), "p90": float(np.percentile(fine[steady, i], 90))}
                                           for i, nm in enumerate(("prologue_done", "first_operand_issued", "all_loads_issued", "first_operand_complete"))}
    med = out["segments_us"]["lifetime"]["median"]
    out["share_of_lifetime"] = {k: out["segments_us"][k]["mean"] / out["segments_us"]["lifetime"]["mean"] for k in seg if k != "lifetime"}
    ph = np.exp(2j * np.pi * t[steady, 0] / med)
    out["start_phase_concentration_R"] = float(abs(ph.mean()))
    # residency over time: workgroups alive at 200 sample points of the steady window
    pts = np.linspace(0.2 * span, 0.8 * span, 200)
    alive = [(int(((t[:, 0] <= x) & (t[:, 6] > x)).sum())) for x in pts]
    out["resident_workgroups"] = {"mean": float(np.mean(alive)), "min": int(min(alive)), "max": int(max(alive))}
    # co-residents: workgroups by CU (HW_ID: cu_id[11:8] sh_id[12] se_id[15:13]; XCC_ID[3:0]); offset between the two residents' starts
    hw = tr[:, 7]
    cu_key = ((hw >> np.uint64(8)) & np.uint64(0xff)).astype(np.int64) | (((hw >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64) << 8)
    offs = []
    for key in np.unique(cu_key):
        idx = np.where((cu_key == key) & steady)[0]
        st = np.sort(t[idx, 0])
        if len(st) > 4:
            d = np.diff(st)
            offs.extend((d % med) / med)
    if offs:
        offs = np.array(offs)
        out["per_cu_start_gap_over_lifetime"] = {"median": float(np.median(offs)), "p10": float(np.percentile(offs, 10)), "p90": float(np.percentile(offs, 90)),
                                                  "cus_seen": int(len(np.unique(cu_key)))}
    ctx.close()
    return out


if __name__ == "__main__":
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    r = run(pairs)
    if len(sys.argv) > 2 and sys.argv[2] == "json":
        print(json.dumps(r))
    else:
        print(f"# tools/ctmul_trace.py: {r['pairs']} pairs, {r['workgroups']} workgroups; kernel {r['kernel_us_plain']:.1f} us plain, {r['kernel_us_traced']:.1f} us traced "
              f"(same output: {r['traced_equals_plain_output']}); steady-state workgroups {r['steady_workgroups']}")
        for k, v in r["segments_us"].items():
            sh = r["share_of_lifetime"].get(k)
            print(f"{k:16s} median {v['median']:8.2f} us  p10 {v['p10']:8.2f}  p90 {v['p90']:8.2f}  mean {v['mean']:8.2f}" + (f"  = {100 * sh:5.1f} % of a lifetime" if sh is not None else ""))
        print("inside wait_first_load, us after the workgroup's start: " + ", ".join(f"{k} {v['median']:.2f} (p90 {v['p90']:.2f})" for k, v in r["first_operand_us_after_start"].items()))
        print(f"resident workgroups (512 slots): mean {r['resident_workgroups']['mean']:.0f}, min {r['resident_workgroups']['min']}, max {r['resident_workgroups']['max']}")
        print(f"start-phase concentration R = {r['start_phase_concentration_R']:.3f} (0 = uniformly de-phased, 1 = lockstep generations)")
        if "per_cu_start_gap_over_lifetime" in r:
            g_ = r["per_cu_start_gap_over_lifetime"]
            print(f"gap between consecutive workgroup starts on one CU, as a fraction of a lifetime: median {g_['median']:.2f} (0.5 = the two residents perfectly interleaved), p10 {g_['p10']:.2f}, p90 {g_['p90']:.2f}; {g_['cus_seen']} CUs seen")
