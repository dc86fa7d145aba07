#!/usr/bin/env python3
"""bench.py - ciphertext-mul/s at N=4096, 4 RNS limbs (BASELINE.json metric) on N GPUs of one node.

A "step" = one pass of the hot path over this rank's shard of synthetic ciphertext pairs:
    fused ct x ct multiply (coefficient domain in/out)  ->  shard-local modular sum to ONE partial
    ciphertext  ->  (N>1) RCCL all-gather of the partials  ->  local sum of the gathered partials.
Weak scaling: --batch-per-gpu ciphertext pairs per GPU (default 8192 = BASELINE configs[3] per-GPU
share of 65536), inputs resident in HBM before the timed region.  `value` = ct-muls by all ranks / time.

Extra objects on the JSON line:
  roofline     - the dominant kernel of the timed region (ct_mul_kernel): algorithmic bytes
                 (7*L*N*8 = 917504 B per ct-mul) / HIP-event launch duration / 8 TB/s.
  ntt          - BASELINE configs[1] (batch = 1024 RNS polys x 4 limbs, N=4096): forward / inverse NTT
                 kernel time and fraction of HBM peak (2*N*8 algorithmic bytes per residue polynomial).
  cpu_baseline - the CPU oracle ("port": reference has no CPU evaluator) timed on
                 this host's cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: required by RCCL across processes on this driver

HBM_PEAK = 8.0e12  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-per-gpu", type=int, default=8192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline sample duration")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_host
    from deeppowers_amd.params import FheParams
    from deeppowers_amd.sharding import allgather_partials

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    force_dist = bool(os.environ.get("DPFHE_FORCE_DIST"))  # exercise the RCCL path at world size 1 (tests)
    if world > 1 or force_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    params = FheParams.n4096_l4()
    L, N = params.n_limbs, params.n
    B = args.batch_per_gpu
    ctx = Context(params, local_rank)
    ev = Evaluator(ctx)
    dev = ctx.device

    # synthetic inputs: uniform residues, generated on the device, resident in HBM before timing
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    q = torch.tensor(params.moduli, dtype=torch.int64, device=dev).view(1, 1, L, 1)
    a = Ciphertext(torch.randint(0, 2**62, (B, 2, L, N), generator=g, dtype=torch.int64, device=dev) % q)
    b = Ciphertext(torch.randint(0, 2**62, (B, 2, L, N), generator=g, dtype=torch.int64, device=dev) % q)
    # Two output buffers and two HIP streams: the VALU-bound multiply of step i+1 runs on the main stream
    # while the HBM-bound shard-local reduce + all-gather + final sum of step i run on the side stream.
    outs = [ctx.empty(B, components=3) for _ in range(2)]
    partials = [ctx.empty(components=3) for _ in range(2)]
    totals = [ctx.empty(components=3) for _ in range(2)]
    main = torch.cuda.current_stream()
    side = torch.cuda.Stream(device=dev)
    mul_done = [torch.cuda.Event() for _ in range(2)]
    red_done = [torch.cuda.Event() for _ in range(2)]

    ev_start = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev_end = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    counter = [0]

    def step(i=None):
        k = counter[0] & 1
        counter[0] += 1
        main.wait_event(red_done[k])              # the reduce that read outs[k] two steps ago has finished
        if i is not None:
            ev_start[i].record(main)
        c = ev.multiply(a, b, out=outs[k], stream=main)
        if i is not None:
            ev_end[i].record(main)
        mul_done[k].record(main)
        side.wait_event(mul_done[k])
        with torch.cuda.stream(side):
            p = ev.reduce_sum(c, out=partials[k], stream=side)
            gathered = allgather_partials(p.data)  # RCCL all-gather of one partial per rank (no-op at N=1)
            ev.reduce_sum(Ciphertext(gathered), out=totals[k], stream=side)
        red_done[k].record(side)
        return k

    # one-off initialisation that is not part of any step: RCCL communicator creation and code-object loading
    if dist.is_initialized():
        allgather_partials(partials[0])
        dist.barrier()
    ev.reduce_sum(ev.multiply(Ciphertext(a.data[:1]), Ciphertext(b.data[:1])))
    torch.cuda.synchronize()

    def measure_ntt():
        # BASELINE configs[1]: NTT / INTT HBM-roofline run, batch 1024 RNS polys x 4 limbs
        nb = 1024
        x = torch.randint(0, 2**62, (nb, L, N), generator=g, dtype=torch.int64, device=dev) % q.view(1, L, 1)
        y = torch.empty_like(x)
        ntt = {}
        fns = (("fwd", ev.ntt_forward), ("inv", ev.ntt_inverse))
        for _ in range(5):
            for _, fn in fns:
                fn(x, out=y)
        # 30 launches of each direction, interleaved (the clocks drift by ~10 % within a second of sustained load, so
        # measuring one direction after the other would penalise the second), enqueued back to back, each bracketed by
        # its own pair of HIP events; one host sync at the end (a host sync after every launch reads 10-15 % slower)
        evs = {name: [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)] for name, _ in fns}
        for i in range(30):
            for name, fn in fns:
                s_, e_ = evs[name][i]
                s_.record(); fn(x, out=y); e_.record()
        torch.cuda.synchronize()
        for name, _ in fns:
            ts = sorted(s_.elapsed_time(e_) * 1e-3 for s_, e_ in evs[name])
            med = ts[len(ts) // 2]
            nbytes = 2 * N * 8 * nb * L
            ntt[name] = {"median_us": med * 1e6, "min_us": ts[0] * 1e6, "GBps": nbytes / med / 1e9, "frac_of_hbm_peak": nbytes / med / HBM_PEAK}
        ntt["algorithmic_bytes"] = 2 * N * 8 * nb * L
        ntt["workload"] = "BASELINE configs[1]: batch=1024 RNS polys x 4 limbs, N=4096, out-of-place"
        return ntt


    def measure_other_configs():
        """BASELINE configs[2] (ct x pt matvec, hidden=768, 64 input ciphertexts) and the N1 relinearisation, kernel-only."""
        from deeppowers_amd.evaluator import Plaintext
        other = {}

        def timed(fn, reps):
            fn(); fn()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for s_, e_ in evs:
                s_.record(); fn(); e_.record()
            torch.cuda.synchronize()
            ts = sorted(s_.elapsed_time(e_) * 1e-3 for s_, e_ in evs)
            return ts[len(ts) // 2]

        rows, cols = 768, 64
        W = Plaintext(torch.randint(0, 2**62, (rows, cols, L, N), generator=g, dtype=torch.int64, device=dev) % q, True)
        xs = Ciphertext(torch.randint(0, 2**62, (cols, 2, L, N), generator=g, dtype=torch.int64, device=dev) % q, True)
        y = ctx.empty(rows, components=2)
        t = timed(lambda: ev.matvec_plain(W, xs, out=y), 6)
        nbytes = (rows * cols + cols * 2 + rows * 2) * L * N * 8
        other["matvec_plain"] = {"workload": "BASELINE configs[2]: 768 rows x 64 input ciphertexts, polynomial weights (6 GiB), N=4096, L=4",
                                 "median_us": t * 1e6, "GBps": nbytes / t / 1e9, "frac_of_hbm_peak": nbytes / t / HBM_PEAK,
                                 "mod_fma_per_s": rows * cols * 2 * L * N / t}
        del W
        w = torch.randint(0, 2**62, (rows, cols, L), generator=g, dtype=torch.int64, device=dev) % q.view(1, 1, L)
        t = timed(lambda: ev.matvec_scalar(w, xs, out=y), 6)
        other["matvec_scalar"] = {"workload": "same shape, scalar weights in Z_q", "median_us": t * 1e6, "mod_fma_per_s": rows * cols * 2 * L * N / t}
        nb = 2048
        c3 = Ciphertext(torch.randint(0, 2**62, (nb, 3, L, N), generator=g, dtype=torch.int64, device=dev) % q)
        evk = torch.randint(0, 2**62, (L, 2, L, N), generator=g, dtype=torch.int64, device=dev) % q
        o2 = ctx.empty(nb, components=2)
        t = timed(lambda: ev.relinearize(c3, evk, out=o2), 6)
        other["relinearize"] = {"workload": f"{nb} three-component ciphertexts, RNS-digit keys, N=4096, L=4", "median_us": t * 1e6, "per_s": nb / t}
        del c3, evk, o2, xs, y, w
        # BASELINE configs[4] sizes (N=8192, 6 limbs): NTT / INTT / ct x ct micro-benchmarks on a second context
        p5 = FheParams.n8192_l6()
        ctx5 = Context(p5, local_rank)
        ev5 = Evaluator(ctx5)
        L5, N5 = p5.n_limbs, p5.n
        q5 = torch.tensor(p5.moduli, dtype=torch.int64, device=dev)
        nb5 = 256
        x5 = torch.randint(0, 2**62, (nb5, L5, N5), generator=g, dtype=torch.int64, device=dev) % q5.view(1, L5, 1)
        y5 = torch.empty_like(x5)
        c5 = {"workload": "BASELINE configs[4] sizes: N=8192, 6 x 60-bit limbs (micro-benchmarks; the GPT-2 forward itself is out of reach, SURVEY.md section 7)"}
        for name, fn in (("ntt_fwd", ev5.ntt_forward), ("ntt_inv", ev5.ntt_inverse)):
            t = timed(lambda: fn(x5, out=y5), 10)
            nbytes = 2 * N5 * 8 * nb5 * L5
            c5[name] = {"rns_polys": nb5, "median_us": t * 1e6, "GBps": nbytes / t / 1e9, "frac_of_hbm_peak": nbytes / t / HBM_PEAK}
        bb = 1024
        a5 = Ciphertext(torch.randint(0, 2**62, (bb, 2, L5, N5), generator=g, dtype=torch.int64, device=dev) % q5.view(1, 1, L5, 1))
        b5 = Ciphertext(torch.randint(0, 2**62, (bb, 2, L5, N5), generator=g, dtype=torch.int64, device=dev) % q5.view(1, 1, L5, 1))
        o5 = ctx5.empty(bb, components=3)
        t = timed(lambda: ev5.multiply(a5, b5, out=o5), 6)
        alg5 = 7 * L5 * N5 * 8 * bb
        c5["ct_mul"] = {"pairs": bb, "median_us": t * 1e6, "per_s": bb / t, "GBps": alg5 / t / 1e9, "frac_of_hbm_peak": alg5 / t / HBM_PEAK}
        other["n8192_l6"] = c5
        del a5, b5, o5, x5, y5
        ctx5.close()
        return other

    ntt_result = measure_ntt() if world == 1 else None   # before the long multiply loop heats the chip into lower clocks
    other_result = measure_other_configs() if world == 1 else None

    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = step(i)
    fence()
    elapsed = time.perf_counter() - t0
    out = outs[last]
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # HBM traffic of the dominant kernel from the committed PMC passes (collected with rocprofv3 --pmc in their own
    # runs, corrected as MI355X_MICROARCH.md prescribes); scaled per ct-mul because traffic is linear in the batch.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            traffic = json.load(f)["ct_mul_kernel<FoldArith,12,4>"]["hbm_bytes_per_ct_mul"] * B
    except Exception:
        pass
    # secondary (ALU) roofline: butterflies per second against the register-only butterfly loop of tools/ubench on this chip
    alu_peak = None
    try:
        import re as _re
        with open(os.path.join(ROOT, "profiles", "r01_ubench.log")) as f:
            m = _re.findall(r"butterflies fold\s+8 blk/CU:.*?([0-9.]+) T bfly/s", f.read())
        alu_peak = float(m[-1]) * 1e12 if m else None
    except Exception:
        pass
    kernel_ms = [s.elapsed_time(e) for s, e in zip(ev_start, ev_end)]
    k_avg = sum(kernel_ms) / len(kernel_ms) * 1e-3
    alg_bytes = 7 * L * N * 8 * B
    achieved = alg_bytes / k_avg

    result = {
        "metric": "ciphertext-mul/s (N=4096, 4 RNS limbs)",
        "value": world * B * args.steps / elapsed,
        "unit": "ct-mul/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": f"ct x ct multiply (tensor product, coeff domain in/out), N=4096, L=4 x 60-bit limbs, "
                        f"{B} ciphertext pairs per GPU (BASELINE configs[3] per-GPU shard), shard-local reduce + "
                        f"all-gather of one partial ct per GPU (reduce/gather of step i overlapped with the multiply of step i+1 on a second stream)",
            "log2_n": 12, "n_limbs": 4, "batch_per_gpu": B, "global_batch": B * world,
            "parallelism": f"batch-sharded x{world}, one process per GPU" + (", RCCL all-gather" if world > 1 else ""),
            "arith": "fold(2^60-d)" if ctx.uses_fold else "shoup",
        },
        "roofline": {
            "kernel": "ct_mul_kernel<FoldArith,12,4>", "bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": achieved / HBM_PEAK, "traffic": traffic,
            "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)" if traffic else None,
            "alu_roofline_note": "VALU issue, not HBM, bounds this kernel: it issues one VALU instruction per SIMD every ~5 cycles, the rate of the register-only butterfly loop (tools/ubench: 2.35-2.6 T butterflies/s = 78-87% of 8 TB/s NTT-equivalent; profiles/r01_ubench.log, r01_pmc_sq_ntt_bench.txt)",
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": k_avg * 1e3,
            "alu": {"unit": "butterflies/s", "achieved": 7 * L * (N // 2) * 12 * B / k_avg, "peak": alu_peak,
                    "frac": (7 * L * (N // 2) * 12 * B / k_avg / alu_peak) if alu_peak else None,
                    "peak_source": "profiles/r01_ubench.log: register-only radix-2 butterflies, FoldArith, 8 workgroups per CU",
                    "note": "7 transforms x L limbs x (N/2) log2 N butterflies per ct-mul; the dyadic products, canonicalisation and the lower clock under HBM load are not in the peak"},
        },
        # "report compute-only and end-to-end": `value` is end-to-end (multiply + shard-local reduce +
        # all-gather + final sum); this is the multiply kernel alone, timed inside the same overlapped steps
        "compute_only_ct_mul_per_s": world * B / k_avg,
    }

    if ntt_result is not None:
        result["ntt"] = ntt_result
        result["other_configs"] = other_result
    if world == 1:
        # the reduced result of the last step equals a recomputation of the reduction on the main stream
        chk = ev.reduce_sum(Ciphertext(out), stream=main)
        torch.cuda.synchronize()
        result["reduce_consistent"] = bool(torch.equal(chk.data, totals[last]))

    if world == 1:
        if not args.no_cpu_baseline:
            from oracle.cbind import Oracle
            orc = Oracle.from_params(params)
            cores = orc.max_threads()
            probe = max(cores, 4)
            ah, bh = to_host(a.data[:probe]), to_host(b.data[:probe])
            t1 = time.perf_counter(); orc.ct_mul(ah, bh, threads=cores); t_probe = time.perf_counter() - t1
            n_s = int(min(B, max(probe, probe * args.cpu_seconds / max(t_probe, 1e-6))))
            ah, bh = to_host(a.data[:n_s]), to_host(b.data[:n_s])
            t1 = time.perf_counter(); cpu_out = orc.ct_mul(ah, bh, threads=cores); t_all = time.perf_counter() - t1
            # the CPU baseline leg doubles as the checker of the timed GPU output: every word of the sample must match
            result["bit_exact_sample"] = bool(np.array_equal(to_host(out[:n_s]), cpu_out))
            n_1 = max(1, n_s // cores)
            t1 = time.perf_counter(); orc.ct_mul(ah[:n_1], bh[:n_1], threads=1); t_one = time.perf_counter() - t1
            result["cpu_baseline"] = {
                "value": n_s / t_all, "unit": "ct-mul/s", "cores": cores, "kind": "port",
                "sample": f"first {n_s} ciphertext pairs of the same batch, oracle/oracle.c Harvey-NTT evaluator, OpenMP over "
                          f"{cores} threads, {t_all:.1f} s (reference has no CPU evaluator: build CPU evaluator)",
                "single_thread_value": n_1 / t_one,
            }
            result["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]

    if rank == 0:
        print(json.dumps(result))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
