#!/usr/bin/env python3
"""bench.py - ciphertext-mul/s at N=4096, 4 RNS limbs (BASELINE.json metric) on N GPUs of one node.

A "step" = one pass of the hot path over this rank's shard of synthetic ciphertext pairs:
    fused ct x ct multiply (coefficient domain in/out)  ->  shard-local modular sum to ONE partial
    ciphertext  ->  (N>1) RCCL all-gather of the partials  ->  local sum of the gathered partials.
Weak scaling: --batch-per-gpu ciphertext pairs per GPU (default 8192 = BASELINE configs[3] per-GPU
share of 65536), inputs resident in HBM before the timed region.  `value` = ct-muls by all ranks / time.

Extra objects on the JSON line (the driver's stored record keeps `config`, `roofline` and `cpu_baseline` whole and only the NAMES of
other keys, so everything a reader of that record needs is inside those three; full detail also goes to gpurun_out/bench_detail.json):
  roofline     - the dominant kernel of the timed region (the fused multiply, in the form config.autotune names): algorithmic bytes
                 (7*L*N*8 = 917504 B per ct-mul) / HIP-event launch duration / 8 TB/s.
  roofline.ntt - BASELINE configs[1] (the metric's second half): forward / inverse NTT fraction of HBM peak (region-bracketed medians
                 and the 2 s windows), against the copy kernel, with board power / cap / shader clock over the windows.
  roofline.regime - what kind of box this is: CU count, compute / memory partition, clocks (rocm-smi), the multiply's stall share.
  config.autotune - which form of the fused multiply ran, and every measurement behind the choice (dpfhe_ctx_autotune on
                 8192-pair-sized scratch - the explicit opt-in; context creation measures nothing -, three timed steps per form).
  roofline.* scalars - the driver's record keeps only SCALARS of `roofline`: ntt_fwd_frac / ntt_inv_frac (configs[1], the metric's second
                 half), their 2 s windows, copy_frac, board_w / cap_w / sclk_mhz, the N = 8192 fractions, the generic-prime (Shoup)
                 figures and autotune_chosen sit there as plain numbers next to the nested objects.
  ntt          - BASELINE configs[1] (batch = 1024 RNS polys x 4 limbs, N=4096): forward / inverse NTT
                 kernel time and fraction of HBM peak (2*N*8 algorithmic bytes per residue polynomial).
  sustained    - >= 2 s of back-to-back forward / inverse NTT launches (configs[1] in place, and 1 GiB out of place) and of the
                 library's plain copy kernel, each with board power / cap / shader clock sampled over THAT window: the
                 driver-visible evidence for what bounds the transforms (MEASUREMENTS.md section 5).
  cpu_baseline - the CPU oracle ("port": reference has no CPU evaluator) timed on
                 this host's cores on a bounded sample of the same workload (rank 0, N=1 only).

  multi_gpu_programs - N>1 only, rank 0, after the timed region: the pure-C++ multi-process programs for configs[3] and configs[4]
                 (token-sharded transformer block) over the same GPUs, each checked against its world-size-1 / plaintext answer.

`--dry-run` (CPU, gloo): walks every collective, barrier and rank-0 report of the N>1 path with the kernels stood in by host
arithmetic - `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 --dry-run` (tests/test_bench_dry_run.py).
"""
import argparse
import glob
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: required by RCCL across processes on this driver

HBM_PEAK = 8.0e12  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


class PowerSampler(threading.Thread):
    """Reads power1_average|power1_input, power1_cap and freq1_input (sclk) of one HIP device from sysfs every 2 ms.  The
    transforms run at the board's power cap (MEASUREMENTS.md section 5, profiles/*_power_probe.txt); the bench line carries the evidence."""

    def __init__(self, device):
        super().__init__(daemon=True)
        self.rows, self.halt, self.dir = [], False, None
        try:
            import torch
            p = torch.cuda.get_device_properties(device)
            want = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{getattr(p, 'pci_device_id', 0):02x}."
            for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
                if os.path.basename(os.path.realpath(os.path.join(d, "..", ".."))).startswith(want):
                    self.dir = d
        except Exception:
            pass
        self.pfile = next((f for f in ("power1_average", "power1_input") if self.dir and os.path.exists(os.path.join(self.dir, f))), None)

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return int(f.read().strip())
        except Exception:
            return None

    def run(self):
        while not self.halt and self.pfile:
            self.rows.append((self._read(os.path.join(self.dir, self.pfile)), self._read(os.path.join(self.dir, "freq1_input"))))
            time.sleep(0.002)

    def finish(self):
        self.halt = True
        if self.is_alive():
            self.join(timeout=1.0)
        ps = [p / 1e6 for p, _ in self.rows if p]
        fs = [f / 1e6 for _, f in self.rows if f]
        cap = self._read(os.path.join(self.dir, "power1_cap")) if self.dir else None
        if not ps:
            return None
        return {"board_w_mean": sum(ps) / len(ps), "board_w_max": max(ps), "cap_w": cap / 1e6 if cap else None,
                "sclk_mhz_mean": sum(fs) / len(fs) if fs else None, "samples": len(ps),
                "source": "hwmon power1/freq1 of this GPU, 2 ms sampling over the timed steps"}


def box_regime(device_index):
    """rank 0, outside the timed region: what rocm-smi says about this GPU (partition modes, clock levels, power cap).  The pool's
    boxes differ in how the fused multiply runs (MEASUREMENTS.md section 5); the record names the box it was measured on."""
    import subprocess
    out = {}
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        out.update({"name": p.name, "compute_units": p.multi_processor_count, "hbm_gib": round(p.total_memory / 2**30, 1),
                    "l2_mib": round(getattr(p, "L2_cache_size", 0) / 2**20, 1)})
    except Exception:
        pass
    try:
        run = subprocess.run(["rocm-smi", "-d", str(device_index), "--showcomputepartition", "--showmemorypartition", "--showclocks", "--showmaxpower", "--json"],
                             capture_output=True, text=True, timeout=20)
        j = json.loads(run.stdout[run.stdout.index("{"):])
        card = next(iter(j.values()))
        for k_, v in card.items():
            kl = k_.lower()
            if "compute partition" in kl:
                out["compute_partition"] = v
            elif "memory partition" in kl:
                out["memory_partition"] = v
            elif "max graphics package power" in kl:
                out["cap_w"] = v
            elif any(c in kl for c in ("sclk clock", "mclk clock", "fclk clock", "socclk clock")) and "level" not in kl:
                out[kl.split()[0]] = v
    except Exception as e:
        out["rocm_smi"] = repr(e)[:120]
    return out


def on_rank0_while_others_wait(rank, fn, key="dpfhe_bench_rank0_programs", timeout_s=600):
    """fn() on rank 0 while the other ranks wait on the HOST (a key of the process group's store), not inside a collective: a RCCL
    barrier would park a spinning kernel on every waiting GPU for as long as rank 0's programs use those GPUs."""
    import datetime

    import torch.distributed as dist
    store = dist.distributed_c10d._get_default_store()
    if rank == 0:
        try:
            return fn()
        finally:
            store.set(key, b"1")
    store.wait([key], datetime.timedelta(seconds=timeout_s))
    return None


def run_program(argv, timeout):
    """(returncode, output) of examples/<argv[0]> (built by __graft_entry__.build()); the program gets its own session: it forks one process
    per GPU, and a timeout must take the whole group down, not only the parent."""
    import signal
    import subprocess
    exe = os.path.join(ROOT, "examples", argv[0])
    if not os.path.exists(exe):
        raise FileNotFoundError(f"{exe}: run `python -c 'import __graft_entry__ as g; g.build()'` first")
    proc = subprocess.Popen([exe] + argv[1:], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    try:
        text, _ = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        text, _ = proc.communicate()
        return -9, (text or "") + f"\n[killed after {timeout} s]"
    return proc.returncode, text


def multi_gpu_programs(world, run=None):
    """N>1, rank 0 only, after the timed region: the two pure-C++ multi-process programs (one process per GPU, forked by the program itself,
    RCCL through the C ABI, no Python on the data path) over the same `world` GPUs:
      configs[3]  examples/sharded_ct_mul: batch-sharded multiply + all-gather of one partial per rank; the global sum must equal a
                  world-size-1 recomputation of the same global batch;
      configs[4]  examples/encrypted_gpt2_block: 8 tokens per rank through one transformer block's linear skeleton (N=8192, 5+1 limbs),
                  tokens sharded over the ranks, no collective until the all-gather of the output ciphertexts; every stage decrypted;
                  examples/sharded_ffn: ONE token, the FFN's 3072 inner features split over the ranks, all-gather of one partial each.
    A failure is reported in the entry; the headline metric does not depend on it.  `run(argv, timeout) -> (returncode, stdout)` is
    replaceable (the dry run passes a stand-in)."""
    run = run or run_program
    out = {}
    programs = [("configs3_cpp_host", ["sharded_ct_mul", str(world), "2048", "5"]),
                ("configs4_token_sharded_block", ["encrypted_gpt2_block", str(8 * world), "2", "json", str(world)])]
    if world in (1, 2, 4, 8):   # the single-token shape of configs[4]: the FFN's inner features split over the ranks (tensor parallel)
        programs.append(("configs4_single_token_tensor_parallel_ffn", ["sharded_ffn", str(world), "3"]))
    for key, argv in programs:
        try:
            t0 = time.perf_counter()
            rc, text = run(argv, 150)   # these take 5-30 s; a hung program must not hold the bench line back for long
            got = [json.loads(l) for l in text.splitlines() if l.startswith("{")]
            out[key] = dict(got[0], wall_s=round(time.perf_counter() - t0, 1)) if got and rc == 0 else {"error": text[-300:], "returncode": rc}
        except Exception as e:
            out[key] = {"error": repr(e)[:300]}
    return out


def dry_run(args):
    """CPU, gloo: the N>1 host path of this file with the kernels stood in by host arithmetic - rendezvous, the native communicator's id
    shipping, barriers, the all-gather of one partial per rank, the MAX / MIN reductions of the report, the per-rank rates - so that the
    first multi-GPU lease needs no code change.  The "products" are a fixed function of the GLOBAL pair index, so the gathered sum must
    equal a world-size-1 recomputation of the same global batch (checked on every rank)."""
    import hashlib

    import numpy as np
    import torch
    import torch.distributed as dist

    from deeppowers_amd.params import FheParams
    from deeppowers_amd.sharding import NativeComm, allgather_partials, shard_bounds

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world
    if world > 1:
        dist.init_process_group("gloo")
    params = FheParams.n4096_l4()
    L, N, B = params.n_limbs, params.n, min(args.batch_per_gpu, 4)
    q = np.array(params.moduli, np.uint64).reshape(1, L, 1)
    id_ok = None
    if args.native_comm:   # the id-shipping half of NativeComm (dpfhe_comm_unique_id works without a GPU; dpfhe_comm_create does not)
        r_, w_, uid = NativeComm.rendezvous()
        digest = int.from_bytes(hashlib.sha256(uid).digest()[:7], "little")
        ids = allgather_partials(torch.tensor([digest], dtype=torch.int64))
        id_ok = bool((ids == ids[0]).all()) and (r_, w_) == (rank, world) and len(uid) == 128

    def product(i):   # stands in for ct_mul of global pair i: 3 components of canonical residues
        seed = ((i + 1) * 0x9E3779B97F4A7C15) & ((1 << 64) - 1)
        v = (np.arange(3 * L * N, dtype=np.uint64).reshape(3, L, N) * np.uint64(2 * i + 1) + np.uint64(seed)) & np.uint64((1 << 59) - 1)   # wraps mod 2^64 by design
        return v % q

    def msum(items):
        acc = np.zeros((3, L, N), np.uint64)
        for v in items:
            acc = (acc + v) % q
        return acc
    lo = rank * B

    def step():
        partial = msum(product(lo + i) for i in range(B))                                      # shard-local reduce
        gathered = allgather_partials(torch.from_numpy(partial.view(np.int64)))                # the one collective
        return msum(gathered.numpy().view(np.uint64)), gathered

    def step_allreduce():   # SURVEY.md 8(e)'s alternative: 64-bit sum all-reduce of the partials (world * q < 2^64), then one mod-q pass
        partial = msum(product(lo + i) for i in range(B))
        t = torch.from_numpy(partial.view(np.int64).copy())
        if dist.is_initialized():
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy().view(np.uint64) % q

    def fence():
        if dist.is_initialized():
            dist.barrier()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    gather_us = []
    for _ in range(args.steps):
        t1 = time.perf_counter()
        total, gathered = step()
        gather_us.append((time.perf_counter() - t1) * 1e6)
    fence()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64)
        every = torch.empty(world, dtype=torch.float64)
        dist.all_gather_into_tensor(every, t)
        per_rank = [float(v) for v in every.tolist()]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    want = msum(product(i) for i in range(world * B))   # world-size-1 recomputation of the same global batch
    assert shard_bounds(world * B, world, rank) == (lo, lo + B)
    ar_total = step_allreduce()
    ok = torch.tensor([int(np.array_equal(total, want) and gathered.shape[0] == world and np.array_equal(ar_total, want))])
    if dist.is_initialized():
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    rates = [B * args.steps / e for e in per_rank]
    gather_us.sort()
    result = {"metric": "ciphertext-mul/s (N=4096, 4 RNS limbs)", "value": world * B * args.steps / elapsed, "unit": "ct-mul/s", "n_gpus": world, "steps": args.steps,
              "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
              "data": "synthetic", "dry_run": True,
              "config": {"workload": "DRY RUN on CPU over gloo: host path of the sharded multiply-reduce only, kernels stood in by host arithmetic (not a measurement)",
                         "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"batch-sharded x{world}, one process per rank", "collective": "torch.distributed all_gather_into_tensor (gloo)"},
              "per_rank_ct_mul_per_s": {"min": min(rates), "max": max(rates), "ranks": len(rates)},
              "allgather_us": {"median": gather_us[len(gather_us) // 2], "min": gather_us[0], "max": gather_us[-1]},
              "reduce_consistent": bool(ok.item()), "global_sum_matches_world1": bool(ok.item()), "native_comm_id_shipped": id_ok,
              "collective_ab": {"allreduce_total_equals_allgather_total": bool(ok.item()), "what": "all-gather + local sum against sum all-reduce + one mod-q pass (gloo)"}}
    if dist.is_initialized() and world > 1:   # the rank-0 programs of the N>1 report, stood in by canned output; the host-side wait is real
        def fake(argv, timeout):
            if argv[0] == "sharded_ct_mul":
                return 0, json.dumps({"host": "c++", "world": int(argv[1]), "matches_world1_recomputation": True, "ct_mul_per_s": 0.0, "dry_run": True}) + "\nOK\n"
            if argv[0] == "sharded_ffn":
                return 0, json.dumps({"block": "ffn_linear_tensor_parallel", "world": int(argv[1]), "correct": True, "dry_run": True}) + "\nOK\n"
            return 0, json.dumps({"block": "transformer_linear_skeleton", "tokens": int(argv[1]), "ranks": int(argv[4]), "correct": True, "dry_run": True}) + "\nOK\n"
        progs = on_rank0_while_others_wait(rank, lambda: multi_gpu_programs(world, fake))
        if rank == 0:
            result["multi_gpu_programs"] = progs
    if rank == 0:
        print(json.dumps(result))
    if dist.is_initialized():
        dist.destroy_process_group()


def digest_other(o):
    """other_configs in < 2 kB for the JSON line (per layer ms per token + all_correct, the stage table's fractions); the full block goes
    to gpurun_out/bench_detail.json."""
    r3 = lambda v: None if v is None else round(v, 4)
    d = {}
    for k_ in ("matvec_plain", "matvec_scalar", "relinearize"):
        if k_ in o:
            d[k_] = {x: r3(o[k_].get(x)) for x in ("median_us", "frac_of_hbm_peak", "per_s") if o[k_].get(x) is not None}
    if "n8192_l6" in o:
        c5 = o["n8192_l6"]
        d["n8192_l6"] = {"ntt_fwd_frac": r3(c5["ntt_fwd"]["frac_of_hbm_peak"]), "ntt_inv_frac": r3(c5["ntt_inv"]["frac_of_hbm_peak"]),
                         "ct_mul_per_s": round(c5["ct_mul"]["per_s"]), "ct_mul_frac": r3(c5["ct_mul"]["frac_of_hbm_peak"])}
    for key in ("shoup_n4096_l4", "p31_n4096_l4", "wide49_n4096_l4", "shoup55_n4096_l4"):   # the per-limb arithmetic classes at the headline shape
        if isinstance(o.get(key), dict) and "ct_mul" in o[key]:
            sg = o[key]
            d[key] = {"classes": sg.get("limb_classes"), "ntt_fwd_frac": r3(sg["ntt_fwd"]["frac_of_hbm_peak"]), "ntt_inv_frac": r3(sg["ntt_inv"]["frac_of_hbm_peak"]),
                      "ct_mul_per_s": round(sg["ct_mul"]["per_s"]), "ct_mul_frac": r3(sg["ct_mul"]["frac_of_hbm_peak"]), "fold_over_this_ct_mul": r3(sg["fold_over_shoup_ct_mul"]),
                      "relin_per_s": round(sg["relinearize"]["per_s"]) if "relinearize" in sg else None}
    pl = o.get("packed_linear") or {}
    if pl:
        e = {"all_correct": pl.get("all_correct")}
        if pl.get("layers"):
            e["ms_per_token"] = {f"{l.get('layer', '?')}@{l.get('tokens_per_apply', l.get('tokens', '?'))}": r3(l.get("ms_per_token")) for l in pl["layers"]}
        for blk in ("ffn_block", "transformer_block", "activated_ffn", "activated_block", "activated_block_n16384", "activated_block_two_tokens_per_ct",
                    "activated_block_n16384_two_tokens_per_ct", "activated_stack", "activated_stack_two_tokens_per_ct"):
            if isinstance(pl.get(blk), dict):
                e[blk] = {x: pl[blk].get(x) for x in ("ms_per_token", "ms_per_token_per_block", "correct", "blocks", "correct_blocks", "data_limbs", "tokens", "tokens_per_ciphertext", "error", "budget_bits", "log2_n",
                                                      "modulus_bits_under_key_switching", "he_standard_128bit_budget_bits",
                                                      "levels", "limbs_per_level") if pl[blk].get(x) is not None}
        ks = pl.get("kernels") or {}
        if ks.get("stages"):
            e["stages"] = [{"stage": st["stage"].split(" (")[0][:40], "us": round(st["median_us"], 1), "frac_hbm": r3(st["frac_of_hbm_peak"]),
                            "frac_bfly": r3(st.get("frac_of_butterfly_ceiling"))} for st in ks["stages"]]
            e["sum_of_stages_ms_per_token"] = r3(ks.get("sum_of_stages_ms_per_token"))
        elif ks.get("error"):
            e["stages_error"] = ks["error"][:120]
        if pl.get("error"):
            e["error"] = pl["error"][:160]
        d["packed_linear"] = e
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-per-gpu", type=int, default=8192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline sample duration")
    ap.add_argument("--native-comm", action="store_true",
                    help="all-gather through the library's own communicator (dpfhe_comm_*, RCCL behind the C ABI) instead of torch.distributed")
    ap.add_argument("--sustained-seconds", type=float, default=2.0, help="length of each sustained NTT / copy window (0 = skip the block)")
    ap.add_argument("--skip-other", action="store_true", help="skip the other_configs block (profiling runs: every launch is serialised under rocprofv3)")
    ap.add_argument("--ct-mul-form", default="auto", choices=["auto", "quad", "dual"],
                    help="form of the fused multiply: auto = measured on this box (library probe + three timed steps per form), or forced")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not profile the multiply's HBM traffic with rocprofv3 after the timed region (roofline.traffic then quotes the committed profile)")
    ap.add_argument("--dry-run", action="store_true", help="CPU + gloo: walk the multi-rank host path (collectives, barriers, report) without kernels")
    args = ap.parse_args()
    if args.dry_run:
        return dry_run(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_host
    from deeppowers_amd.params import FheParams
    from deeppowers_amd.sharding import NativeComm, ShardedMultiplyReduce

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
    autotune = {"at_ctx_create": ctx.tune_info()}   # dpfhe_ctx_create measures nothing: the default form (include/dpfhe.h "A0, continued")

    # synthetic inputs: uniform residues, generated on the device, resident in HBM before timing
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    q = torch.tensor(params.moduli, dtype=torch.int64, device=dev).view(1, 1, L, 1)
    a = Ciphertext(torch.randint(0, 2**62, (B, 2, L, N), generator=g, dtype=torch.int64, device=dev) % q)
    b = Ciphertext(torch.randint(0, 2**62, (B, 2, L, N), generator=g, dtype=torch.int64, device=dev) % q)
    # The step is deeppowers_amd.sharding.ShardedMultiplyReduce (also what tests/test_gpu_parity.py checks at this size):
    # multiply on the main stream; shard-local reduce -> all-gather of one partial per rank -> final sum on a side stream,
    # double-buffered so that they overlap the next step's multiply.
    comm = NativeComm.from_process_group(local_rank) if args.native_comm else None
    pipe = ShardedMultiplyReduce(ev, B, comm=comm)
    main = pipe.main
    outs, totals = pipe.outs, pipe.totals

    ev_start = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev_end = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev_gather = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i=None):
        pipe.gather_events = ev_gather[i] if i is not None else None
        return pipe.step(a, b, timing=(ev_start[i], ev_end[i]) if i is not None else None)

    # one-off initialisation that is not part of any step: RCCL communicator creation and code-object loading
    if dist.is_initialized() or comm is not None:
        warm = ShardedMultiplyReduce(ev, 1, comm=comm)
        warm.step(Ciphertext(a.data[:1]), Ciphertext(b.data[:1]))
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        del warm
    ev.reduce_sum(ev.multiply(Ciphertext(a.data[:1]), Ciphertext(b.data[:1])))
    torch.cuda.synchronize()

    def measure_ntt():
        # BASELINE configs[1] as SURVEY.md 8(d) states it ("Config 2: 1024 RNS polys (4096 residue polys, 128 MiB), forward then inverse,
        # timed separately, >= 20 iterations after 5 warm-ups, median"): ONE 128 MiB buffer transformed in place, forward then inverse
        # (so the data round-trips and the last inverse must reproduce the input - checked).  The 128 MiB fit the chip's 256 MiB
        # Infinity Cache; the same batch out of place (256 MiB touched) and a 1 GiB batch (nothing cached) are reported next to it.
        nb = 1024
        x = torch.randint(0, 2**62, (nb, L, N), generator=g, dtype=torch.int64, device=dev) % q.view(1, L, 1)
        x0 = x.clone()
        y = torch.empty_like(x)

        def timed_pair(fwd, inv, reps, warm):
            # `reps` launches of each direction, interleaved (the clocks drift by ~10 % within a second of sustained load, so
            # measuring one direction after the other would penalise the second), enqueued back to back, each bracketed by its
            # own pair of HIP events; one host sync at the end (a host sync after every launch reads 10-15 % slower)
            fns = (("fwd", fwd), ("inv", inv))
            for _ in range(warm):
                for _, fn in fns:
                    fn()
            evs = {name: [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)] for name, _ in fns}
            for i in range(reps):
                for name, fn in fns:
                    s_, e_ = evs[name][i]
                    s_.record(); fn(); e_.record()
            torch.cuda.synchronize()
            return {name: sorted(s_.elapsed_time(e_) * 1e-3 for s_, e_ in evs[name]) for name, _ in fns}

        def entry(ts, nbytes):
            med = ts[len(ts) // 2]
            return {"median_us": med * 1e6, "min_us": ts[0] * 1e6, "GBps": nbytes / med / 1e9, "frac_of_hbm_peak": nbytes / med / HBM_PEAK}

        def timed_runs(fn, samples, burst, warm):
            # one direction alone ("timed separately"): `samples` samples, each = `burst` back-to-back launches bracketed by ONE
            # pair of HIP events (the contract's "HIP events over the timed region" / launches), per-launch time = sample / burst
            for _ in range(warm):
                fn()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(samples)]
            for s_, e_ in evs:
                s_.record()
                for _ in range(burst):
                    fn()
                e_.record()
            torch.cuda.synchronize()
            return sorted(s_.elapsed_time(e_) * 1e-3 / burst for s_, e_ in evs)

        nbytes = 2 * N * 8 * nb * L
        ntt = {}
        t_in = timed_pair(lambda: ev.ntt_forward_(x), lambda: ev.ntt_inverse_(x), 30, 5)
        ntt["round_trip_exact"] = bool(torch.equal(x, x0))
        ntt["per_launch_events"] = {name: entry(t_in[name], nbytes) for name in ("fwd", "inv")}
        ntt["per_launch_events"]["note"] = ("one HIP event pair around EVERY launch, forward and inverse interleaved (rounds 1-2's figure): a lone ~60 us launch between two "
                                            "event records also pays its own dispatch latency, which back-to-back launches hide")
        ntt["fwd"] = entry(timed_runs(lambda: ev.ntt_forward_(x), 30, 32, 5), nbytes)
        ntt["inv"] = entry(timed_runs(lambda: ev.ntt_inverse_(x), 30, 32, 5), nbytes)
        t_out = {"fwd": timed_runs(lambda: ev.ntt_forward(x, out=y), 30, 32, 5), "inv": timed_runs(lambda: ev.ntt_inverse(x, out=y), 30, 32, 5)}
        ntt["out_of_place"] = {name: entry(t_out[name], nbytes) for name in ("fwd", "inv")}
        ntt["algorithmic_bytes"] = nbytes
        ntt["workload"] = ("BASELINE configs[1]: batch=1024 RNS polys x 4 limbs, N=4096 (4096 residue polynomials, 128 MiB), in place; forward and inverse timed "
                           "separately, 5 warm-ups, median of 30 samples of 32 back-to-back launches each (one HIP event pair per sample: the pair costs ~15 us of exposed dispatch latency, 0.5 us per launch at this sample size); `out_of_place`: the same batch into "
                           "a second buffer; `steady_state`: 8192 RNS polys (1 GiB) out of place; `per_launch_events`: the interleaved one-event-pair-per-launch figure")
        nb2 = 8192
        if a.data.numel() >= nb2 * L * N:
            x2 = a.data.view(-1)[: nb2 * L * N].view(nb2, L, N)    # canonical residues already resident (the multiply's operand)
            y2 = outs[0].view(-1)[: nb2 * L * N].view(nb2, L, N)
        else:                                                      # small --batch-per-gpu runs: dedicated 1 GiB buffers
            x2 = torch.randint(0, 2**62, (nb2, L, N), generator=g, dtype=torch.int64, device=dev) % q.view(1, L, 1)
            y2 = torch.empty_like(x2)
        t_big = timed_pair(lambda: ev.ntt_forward(x2, out=y2), lambda: ev.ntt_inverse(x2, out=y2), 20, 3)
        ntt["steady_state"] = {name: entry(t_big[name], 2 * N * 8 * nb2 * L) for name in ("fwd", "inv")}
        fns = (("fwd", None), ("inv", None))
        # "also report a measured device-copy bandwidth as the practical ceiling" - the library's own copy kernel
        # (dpfhe_copy: 16 bytes per lane, eight loads in flight per thread, the streaming kernels' access shape) over the multiply's
        # 2 GiB operand (far beyond the 256 MiB Infinity Cache) into its output buffer, same event bracketing; torch's copy_ next to it
        src = a.data.view(-1) if a.data.numel() >= nb2 * L * N else x2.view(-1)
        dst = outs[0].view(-1)[: src.numel()] if outs[0].numel() >= src.numel() else y2.view(-1)
        cb = 2 * src.numel() * 8

        def timed_copy(fn):
            for _ in range(2):
                fn()
            cev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
            for s_, e_ in cev:
                s_.record(); fn(); e_.record()
            torch.cuda.synchronize()
            ts = sorted(s_.elapsed_time(e_) * 1e-3 for s_, e_ in cev)
            return ts[len(ts) // 2]
        t_own = timed_copy(lambda: ev.device_copy(dst, src))
        t_torch = timed_copy(lambda: dst.copy_(src))
        ntt["device_copy"] = {"bytes_read_plus_written": cb, "median_us": t_own * 1e6, "GBps": cb / t_own / 1e9, "frac_of_hbm_peak": cb / t_own / HBM_PEAK,
                              "kernel": "dpfhe_copy (copy_kernel: 16 B per lane, 8 loads in flight per thread, one 32 KiB tile per workgroup)",
                              "torch_copy_GBps": cb / t_torch / 1e9,
                              "note": "hand-written copy of the multiply operand (2 GiB at the default batch): the practical HBM ceiling the NTT's fraction should be read against"}
        for name, _ in fns:
            for blk in (ntt, ntt["out_of_place"], ntt["steady_state"]):
                blk[name]["frac_of_device_copy"] = blk[name]["GBps"] / ntt["device_copy"]["GBps"]
        return ntt


    def measure_sustained(seconds):
        """>= `seconds` of back-to-back launches per entry, board power / cap / shader clock sampled over that very window (hwmon, 2 ms).
        North-star reading: the transforms run at the board's power cap with the shader clock pulled down, and move their bytes at
        the stated fraction of what a plain copy kernel sustains in the same run."""
        nb = 1024
        x = torch.randint(0, 2**62, (nb, L, N), generator=g, dtype=torch.int64, device=dev) % q.view(1, L, 1)
        nb2 = 8192
        if a.data.numel() >= nb2 * L * N:
            x2 = a.data.view(-1)[: nb2 * L * N].view(nb2, L, N)
            y2 = outs[0].view(-1)[: nb2 * L * N].view(nb2, L, N)
        else:
            x2 = torch.randint(0, 2**62, (nb2, L, N), generator=g, dtype=torch.int64, device=dev) % q.view(1, L, 1)
            y2 = torch.empty_like(x2)

        def window(fn, nbytes):
            for _ in range(3):
                fn()
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            for _ in range(10):
                fn()
            e_.record(); torch.cuda.synchronize()
            est = max(s_.elapsed_time(e_) * 1e-3 / 10, 1e-6)
            n = max(20, int(seconds / est))
            ps = PowerSampler(dev)
            ps.start()
            t0 = time.perf_counter()
            s_.record()
            for _ in range(n):
                fn()
            e_.record(); torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            pw = ps.finish()
            per = s_.elapsed_time(e_) * 1e-3 / n
            out = {"launches": n, "window_s": wall, "us_per_launch": per * 1e6, "GBps": nbytes / per / 1e9, "frac_of_hbm_peak": nbytes / per / HBM_PEAK}
            if pw:
                out.update({"board_w_mean": pw["board_w_mean"], "board_w_max": pw["board_w_max"], "cap_w": pw["cap_w"], "sclk_mhz_mean": pw["sclk_mhz_mean"],
                            "power_samples": pw["samples"], "frac_of_cap": (pw["board_w_mean"] / pw["cap_w"]) if pw["cap_w"] else None})
            return out
        b1, b2 = 2 * N * 8 * nb * L, 2 * N * 8 * nb2 * L
        res = {"seconds_per_window": seconds,
               "copy_1GiB": window(lambda: ev.device_copy(y2, x2), b2),
               "ntt_fwd_configs1_in_place": window(lambda: ev.ntt_forward_(x), b1),
               "ntt_inv_configs1_in_place": window(lambda: ev.ntt_inverse_(x), b1),
               "ntt_fwd_1GiB_out_of_place": window(lambda: ev.ntt_forward(x2, out=y2), b2),
               "ntt_inv_1GiB_out_of_place": window(lambda: ev.ntt_inverse(x2, out=y2), b2),
               # the headline kernel itself: >= `seconds` of back-to-back dpfhe_ct_mul launches on this rank's shard (algorithmic bytes 7 L N 8 per pair)
               "ct_mul_shard": window(lambda: ev.multiply(a, b, out=outs[0]), 7 * L * N * 8 * a.data.shape[0]),
               "note": "each entry: back-to-back launches for >= seconds_per_window, one host sync at the end; power / cap / sclk are hwmon samples of this GPU over "
                       "that window.  frac_of_copy = the entry's algorithmic GB/s over copy_1GiB's (the same run, the same thermal state); configs[1]'s 128 MiB live in the "
                       "Infinity Cache, the 1 GiB entries stream from HBM"}
        for k_ in list(res):
            if k_.startswith("ntt_"):
                res[k_]["frac_of_copy"] = res[k_]["GBps"] / res["copy_1GiB"]["GBps"]
        return res

    def measure_packed_kernels(alu_peak):
        """N3 roofline: the stages of one packed GPT-2 layer application (QKV 768 -> 2304: 64 baby x 16 giant steps, one output
        ciphertext) at N=8192, 5 data limbs + special prime, 8 tokens, on synthetic operands, each stage timed alone (median of 5) against its
        ALGORITHMIC bytes (keys + operands + results, every buffer counted once) and, where it transforms, its butterflies."""
        from deeppowers_amd.evaluator import Plaintext
        pe = FheParams.n8192_l6()
        cx = Context(pe, local_rank)
        evx = Evaluator(cx)
        Lx, Ld, Nx, T, n1, n2 = pe.n_limbs, pe.n_limbs - 1, pe.n, 8, 64, 16   # PackedLinear's split of the 1024 diagonals (baby-heavy: fhe_api.cpp kBabyShiftDefault)
        qx = torch.tensor(pe.moduli, dtype=torch.int64, device=dev)

        def rnd(*shape, limbs):
            return torch.randint(0, 2**62, shape + (limbs, Nx), generator=g, dtype=torch.int64, device=dev) % qx[:limbs].view(*([1] * len(shape)), limbs, 1)

        def timed(fn, reps=5):
            fn(); fn()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for s_, e_ in evs:
                s_.record(); fn(); e_.record()
            torch.cuda.synchronize()
            ts = sorted(s_.elapsed_time(e_) * 1e-3 for s_, e_ in evs)
            return ts[len(ts) // 2]
        W = 8 * Nx                                               # bytes of one residue polynomial
        bfly = (Nx // 2) * pe.log2_n                             # butterflies of one transform
        keys = rnd(n1 - 1, Ld, 2, limbs=Lx)
        xin = Ciphertext(rnd(T, 2, limbs=Ld))
        elts = [pow(3, j + 1, 2 * Nx) for j in range(n1 - 1)]
        diag = Plaintext(rnd(n2, n1, limbs=Lx), True)
        stages = []

        def stage(name, kernels, t, nbytes, transforms, note):
            e = {"stage": name, "kernels": kernels, "median_us": t * 1e6, "us_per_token": t * 1e6 / T, "algorithmic_bytes": nbytes, "GBps": nbytes / t / 1e9,
                 "frac_of_hbm_peak": nbytes / t / HBM_PEAK, "transforms": transforms, "note": note}
            if transforms and alu_peak:
                e["butterflies_per_s"] = transforms * bfly / t
                e["frac_of_butterfly_ceiling"] = transforms * bfly / t / alu_peak
            stages.append(e)
        babies = evx.rotate_hoisted_qp(xin, elts, keys)
        t = timed(lambda: evx.rotate_hoisted_qp(xin, elts, keys))
        stage("baby steps (dpfhe_rotate_hoisted_qp)", "ntt_fwd (inputs, digits) + lift_digits + lift_qp + hoisted_qp_kernel", t,
              ((n1 - 1) * Ld * 2 * Lx + T * 2 * Ld + T * Ld * Lx + n1 * T * 2 * Lx) * W, T * (2 * Ld + Ld * Lx),
              f"keys [{n1 - 1}][5][2][6] + inputs + lifted digits (permuted {n1 - 1} times out of L2, counted once) + {n1} x 8 results over Q P; no inverse transform, no division by P")
        inner = evx.matvec_plain_multi(diag, babies, T)
        t = timed(lambda: evx.matvec_plain_multi(diag, babies, T))
        stage("plaintext products (dpfhe_matvec_plain_multi)", "matvec_fold_kernel<4,4,1>", t, (n2 * n1 * Lx + n1 * T * 2 * Lx + n2 * T * 2 * Lx) * W, 0,
              "1024 diagonals over Q P (W, streamed once) + baby steps + inner sums; 4 multiply-adds per term (operands split at bit 30)")
        ielts = [1] + [pow(3, n1 * i, 2 * Nx) for i in range(1, n2)]
        t = timed(lambda: evx.ntt_inverse_galois(inner, ielts, out=inner))
        stage("inverse transform + giant-step automorphism (dpfhe_ntt_inv_galois)", "ntt_inv_galois_kernel", t, 2 * n2 * T * 2 * Lx * W, n2 * T * 2 * Lx,
              "in place; the automorphism is the gather pattern of the loads")
        rot = evx.rescale_words(inner)
        t = timed(lambda: evx.rescale_words(inner))
        stage("division by P of the inner sums (dpfhe_rescale)", "rescale_kernel", t, (n2 * T * 2 * Lx + n2 * T * 2 * Ld) * W, 0, "the ONE division by P of baby steps and plaintext products")
        gkeys = rnd(n2 - 1, Ld, 2, limbs=Lx)
        gin = Ciphertext(rot[1:].reshape((n2 - 1) * T, 2, Ld, Nx))
        terms = evx.switch_key_qp(gin, gkeys, T)
        t = timed(lambda: evx.switch_key_qp(gin, gkeys, T))
        stage("giant-step key inner products (dpfhe_switch_key_qp)", "relin_kernel<MODE 4>", t, ((n2 - 1) * Ld * 2 * Lx + (n2 - 1) * T * Ld + (n2 - 1) * T * 2 * Lx) * W,
              (n2 - 1) * T * Ld * Lx, f"keys [{n2 - 1}][5][2][6] + the c1 digits + {n2 - 1} x 8 terms over Q P; Ld transforms per (item, limb) instead of Ld + 2")
        ksum = torch.empty((T, 2, Lx, Nx), dtype=torch.int64, device=dev)

        def tail():
            from deeppowers_amd import _cabi
            _cabi.check(cx._lib.dpfhe_reduce_sum(cx.handle, ksum.data_ptr(), terms.data_ptr(), n2 - 1, T * 2, evx._sp(None)), "dpfhe_reduce_sum")
            evx.ntt_inverse_(ksum)
            return evx.rescale_bsgs(ksum, rot)
        t = timed(tail)
        stage("sum of the terms + ONE inverse transform + ONE division by P (dpfhe_reduce_sum, dpfhe_ntt_inv, dpfhe_rescale_bsgs)",
              "reduce_partial/final + ntt_inv_kernel + rescale_bsgs_kernel", t, ((n2 - 1) * T * 2 * Lx + 3 * T * 2 * Lx + n2 * T * Ld + T * Ld + T * 2 * Ld) * W, T * 2 * Lx,
              f"reads the {n2 - 1} x 8 terms once, adds the c0 parts of the {n2} rotated inner sums")
        total = sum(e["median_us"] for e in stages)
        worst = min((e for e in stages if e["median_us"] > 0.05 * total), key=lambda e: max(e["frac_of_hbm_peak"], e.get("frac_of_butterfly_ceiling") or 0))
        out = {"workload": "one application of the packed QKV layer (768 -> 2304) to 8 tokens, stage by stage, synthetic operands; N=8192, 5 x 60-bit data limbs + special prime",
               "stages": stages, "sum_of_stages_us": total, "sum_of_stages_ms_per_token": total * 1e-3 / T,
               "furthest_from_its_bound": {"stage": worst["stage"], "frac_of_hbm_peak": worst["frac_of_hbm_peak"], "frac_of_butterfly_ceiling": worst.get("frac_of_butterfly_ceiling")}}
        del keys, gkeys, diag, babies, inner, rot, terms, ksum
        cx.close()
        return out

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
        # PER-LIMB ARITHMETIC CLASSES (round 6) at the headline shape: N=4096, L=4, configs[1]'s 1024 RNS polynomials and configs[3]'s 8192-pair shard on
        # parameter sets whose primes are NOT the pinned 2^60 - d ones.  Until round 5 one such limb sent the whole context to the generic (Harvey / Shoup)
        # kernels; now every limb runs on the fastest policy its prime admits (include/dpfhe.h dpfhe_ctx_limb_class).
        #   shoup_n4096_l4  : the round-5 line's primes, 59 / 50 / 40 / 33 bits (now fold_scaled, fold_scaled, f64, f64) - name kept for continuity
        #   p31_n4096_l4    : four 31-bit primes (f64: residues as doubles inside a transform) - the reference's widest integer is INT32 (hal.hpp:27-33)
        #   wide49_n4096_l4 : four 49-bit primes (f64_wide: doubles, with reductions inside the transforms)
        #   shoup55_n4096_l4: four 55-bit primes too far below 2^55 for the scaled fold - what still runs on the generic kernels
        def class_line(pg, what):
            ctxg = Context(pg, local_rank)
            evg = Evaluator(ctxg)
            try:
                qg = torch.tensor(pg.moduli, dtype=torch.int64, device=dev)
                xg = torch.randint(0, 2**62, (1024, L, N), generator=g, dtype=torch.int64, device=dev) % qg.view(1, L, 1)
                yg = torch.empty_like(xg)
                sg = {"workload": what + ": N=4096, L=4; NTT on 1024 RNS polys (configs[1]), fused multiply on 8192 pairs (configs[3]'s shard)",
                      "moduli_bits": [int(m).bit_length() for m in pg.moduli], "uses_fold": ctxg.uses_fold, "limb_classes": list(ctxg.limb_classes)}
                for name, fn in (("ntt_fwd", evg.ntt_forward), ("ntt_inv", evg.ntt_inverse)):
                    t = timed(lambda: fn(xg, out=yg), 10)
                    nbytes = 2 * N * 8 * 1024 * L
                    sg[name] = {"median_us": t * 1e6, "GBps": nbytes / t / 1e9, "frac_of_hbm_peak": nbytes / t / HBM_PEAK}
                del xg, yg
                ag = Ciphertext(torch.randint(0, 2**62, (B, 2, L, N), generator=g, dtype=torch.int64, device=dev) % qg.view(1, 1, L, 1))
                bg = Ciphertext(torch.randint(0, 2**62, (B, 2, L, N), generator=g, dtype=torch.int64, device=dev) % qg.view(1, 1, L, 1))
                og = outs[0].view(-1)[: B * 3 * L * N].view(B, 3, L, N)
                t = timed(lambda: evg.multiply(ag, bg, out=og), 5)
                sg["ct_mul"] = {"pairs": B, "median_us": t * 1e6, "per_s": B / t, "frac_of_hbm_peak": 7 * L * N * 8 * B / t / HBM_PEAK}
                t = timed(lambda: ev.multiply(a, b, out=og), 5)     # the fold arm, same launch shape, same moment
                sg["fold_ct_mul_per_s_same_moment"] = B / t
                sg["fold_over_shoup_ct_mul"] = sg["fold_ct_mul_per_s_same_moment"] / sg["ct_mul"]["per_s"]
                del ag, bg
                # the key switch that follows the metric op, on this context's classes (round 6: no longer on the generic kernels): 2048 three-component ciphertexts
                nbr = 2048
                c3g = Ciphertext(torch.randint(0, 2**62, (nbr, 3, L, N), generator=g, dtype=torch.int64, device=dev) % qg.view(1, 1, L, 1))
                evkg = torch.randint(0, 2**62, (L, 2, L, N), generator=g, dtype=torch.int64, device=dev) % qg.view(1, 1, L, 1)
                o2g = ctxg.empty(nbr, components=2)
                t = timed(lambda: evg.relinearize(c3g, evkg, out=o2g), 6)
                sg["relinearize"] = {"cts": nbr, "median_us": t * 1e6, "per_s": nbr / t}
                del c3g, evkg, o2g
                return sg
            finally:
                ctxg.close()
        from deeppowers_amd.params import is_prime, min_primitive_2n_root, ntt_primes

        def shoup55():   # too wide for the doubles (>= 2^50), too far below 2^55 for the scaled fold ((2^55 - q) 2^5 >= 2^24)
            qs, qq = [], (1 << 55) - ((1 << 55) - 1) % 8192
            while len(qs) < 4:
                if is_prime(qq) and (((1 << 55) - qq) << 5) >= (1 << 24):
                    qs.append(qq)
                qq -= 8192
            return FheParams(12, tuple(qs), tuple(min_primitive_2n_root(4096, v) for v in qs))
        for key, mk, what in (("shoup_n4096_l4", FheParams.generic_n4096_l4, "primes of 59/50/40/33 bits, none 2^60 - d (classes: scaled fold x2, f64 x2)"),
                              ("p31_n4096_l4", lambda: ntt_primes(12, 4, 31), "four 31-bit primes (f64 class: error-free FMA products on doubles)"),
                              ("wide49_n4096_l4", lambda: ntt_primes(12, 4, 49), "four 49-bit primes (f64_wide class: doubles with reductions inside the transforms)"),
                              ("shoup55_n4096_l4", shoup55, "four 55-bit primes no fast class takes (generic class: Harvey/Shoup butterflies + 128-bit Barrett products)")):
            try:
                other[key] = class_line(mk(), what)
            except Exception as e:
                other[key] = {"error": repr(e)[:200]}
        # N3: one token through the reference's dense-layer shapes under encryption, slot-packed, through the C++
        # operator API (examples/encrypted_gpt2_linear.cpp: PackedLinear at N=8192, 5 data limbs + special prime); the program
        # decrypts every result and compares it with W x mod t
        try:
            import subprocess
            lib = os.path.join(ROOT, "deeppowers_amd")

            def example(name):   # built by __graft_entry__.build() (examples/Makefile): nothing is compiled inside a bench run
                exe = os.path.join(ROOT, "examples", name)
                if not os.path.exists(exe):
                    raise FileNotFoundError(f"{exe}: run `python -c 'import __graft_entry__ as g; g.build()'` first")
                return exe
            exe = example("encrypted_gpt2_linear")
            torch.cuda.synchronize()
            layers, ok = [], True
            for tokens, reps in ((1, 5), (8, 3)):   # latency of one token, and throughput with 8 tokens per application
                run = subprocess.run([exe, "all", str(reps), "json", str(tokens)], capture_output=True, text=True, timeout=300)
                got = [json.loads(l) for l in run.stdout.splitlines() if l.startswith("{")]
                ok = ok and bool(got) and run.returncode == 0
                layers += got
            other["packed_linear"] = {
                "workload": "GPT-2-small's dense layers (gpt_model.cpp:793 QKV, :848 FFN up/down, attention output) on encrypted, slot-packed "
                            "hidden states, N=8192, 5 x 60-bit data limbs + special prime, t=65537; 1 and 8 tokens per application "
                            "(ms_per_token = enqueue + one sync over `reps` applications / tokens)",
                "layers": layers, "all_correct": ok and all(l["correct"] for l in layers)}
            # two layers chained on the device: x + W_down (W_up x), the linear path of the FFN block with its residual (examples/encrypted_gpt2_ffn.cpp)
            run = subprocess.run([example("encrypted_gpt2_ffn"), "8", "3", "json"], capture_output=True, text=True, timeout=300)
            got = [json.loads(l) for l in run.stdout.splitlines() if l.startswith("{")]
            other["packed_linear"]["ffn_block"] = got[0] if got and run.returncode == 0 else {"error": (run.stdout + run.stderr)[-300:]}
            # configs[4] as one object: a whole transformer block's linear skeleton (QKV -> attention output -> FFN up -> FFN down, residuals) on
            # encrypted hidden states, decrypted and compared with the plaintext result (examples/encrypted_gpt2_block.cpp)
            run = subprocess.run([example("encrypted_gpt2_block"), "8", "2", "json"], capture_output=True, text=True, timeout=600)
            got = [json.loads(l) for l in run.stdout.splitlines() if l.startswith("{")]
            other["packed_linear"]["transformer_block"] = got[0] if got and run.returncode == 0 else {"error": (run.stdout + run.stderr)[-300:]}
            # the FFN block WITH its non-linearity: x + W_down (W_up x)^2, W_up on five limbs, modulus switch to two, the activation as an exact
            # ciphertext x ciphertext multiply (the metric op inside the end-to-end example) + relinearisation, W_down on two limbs
            # (examples/encrypted_gpt2_ffn_act.cpp; gpt_model.cpp:842-859 with the square standing in for GELU)
            run = subprocess.run([example("encrypted_gpt2_ffn_act"), "8", "2", "json"], capture_output=True, text=True, timeout=600)
            got = [json.loads(l) for l in run.stdout.splitlines() if l.startswith("{")]
            other["packed_linear"]["activated_ffn"] = got[0] if got and run.returncode == 0 else {"error": (run.stdout + run.stderr)[-300:]}
            # ... and the whole block with that activation (attention half + W_up on five limbs, square + W_down on two): examples/encrypted_gpt2_block_act.cpp
            run = subprocess.run([example("encrypted_gpt2_block_act"), "8", "2", "json", "ladder"], capture_output=True, text=True, timeout=600)   # modulus 5 / 4 / 3 / 2 limbs inside the block
            got = [json.loads(l) for l in run.stdout.splitlines() if l.startswith("{")]
            other["packed_linear"]["activated_block"] = got[0] if got and run.returncode == 0 else {"error": (run.stdout + run.stderr)[-300:]}
            # ... and the same block on a ring with a SECURITY MARGIN (round 5): N = 16384, six primes = 1 mod 2^15 - 360 bits under key switching against the
            # 438 bits of 128-bit security there; no fused kernel above N = 8192: every key switch and the multiply composed from the batched transforms
            # round 6: two tokens per ciphertext (the slot rows carry two tokens: PackedLinear tokens_per_ciphertext = 2), 16 tokens in 8 ciphertexts, both rings
            for key2, ring in (("activated_block_two_tokens_per_ct", "13"), ("activated_block_n16384_two_tokens_per_ct", "14")):
                try:
                    run2 = subprocess.run([example("encrypted_gpt2_block_act"), "16", "2", "json", "ladder", ring, "2"], capture_output=True, text=True, timeout=600)
                    d2 = json.loads([l for l in run2.stdout.splitlines() if l.startswith("{")][0])
                    other["packed_linear"][key2] = {"ms_per_token": d2["ms_per_token"], "correct": d2["correct"], "tokens": d2["tokens"],
                                                    "tokens_per_ciphertext": d2["tokens_per_ciphertext"], "log2_n": d2["log2_n"], "budget_bits": d2["budget_bits"]}
                except Exception as e:
                    other["packed_linear"][key2] = {"error": repr(e)[:160]}
            run = subprocess.run([example("encrypted_gpt2_block_act"), "8", "2", "json", "ladder", "14"], capture_output=True, text=True, timeout=600)
            got = [json.loads(l) for l in run.stdout.splitlines() if l.startswith("{")]
            other["packed_linear"]["activated_block_n16384"] = got[0] if got and run.returncode == 0 else {"error": (run.stdout + run.stderr)[-300:]}
            # ... and THREE such blocks in a row on ten data limbs (600 bits), the limb count of every level planned from a budget model and falling
            # 10 -> 2 over the 18 levels, the activations as exact multiplies at eight-, five- and two-limb levels; every block's output decrypted
            # and compared (examples/encrypted_gpt2_stack.cpp; gpt_model.cpp:626-672, the layer loop)
            try:   # round 6: the same three blocks with two tokens per ciphertext (16 tokens in 8 ciphertexts)
                run2 = subprocess.run([example("encrypted_gpt2_stack"), "16", "1", "json", "10", "0", "2"], capture_output=True, text=True, timeout=600)
                d2 = json.loads([l for l in run2.stdout.splitlines() if l.startswith("{")][0])
                other["packed_linear"]["activated_stack_two_tokens_per_ct"] = {k_: d2.get(k_) for k_ in ("ms_per_token", "ms_per_token_per_block", "correct", "blocks", "correct_blocks", "data_limbs",
                                                                                                           "tokens", "tokens_per_ciphertext", "log2_n")}
            except Exception as e:
                other["packed_linear"]["activated_stack_two_tokens_per_ct"] = {"error": repr(e)[:160]}
            run = subprocess.run([example("encrypted_gpt2_stack"), "8", "1", "json", "10"], capture_output=True, text=True, timeout=600)
            got = [json.loads(l) for l in run.stdout.splitlines() if l.startswith("{")]
            other["packed_linear"]["activated_stack"] = got[0] if got and run.returncode == 0 else {"error": (run.stdout + run.stderr)[-300:]}
        except Exception as e:   # a missing example binary must not take the headline metric down with it
            other.setdefault("packed_linear", {})["error"] = repr(e)[:300]
        return other

    def alu_ceiling():
        import re as _re
        import subprocess

        def parse(txt):
            m = _re.findall(r"butterflies fused12\s+8 blk/CU:.*?clock\s+([0-9.]+) MHz.*?([0-9.]+) T bfly/s", txt)
            return (float(m[-1][1]) * 1e12, float(m[-1][0])) if m else (None, None)
        exe = os.path.join(ROOT, "tools", "bin", "ubench2")
        if rank == 0 and os.path.exists(exe):
            try:
                torch.cuda.synchronize()
                run = subprocess.run([exe, "bfly"], capture_output=True, text=True, timeout=120)
                peak, clock = parse(run.stdout)
                if peak:
                    return peak, clock, "tools/bin/ubench2 bfly, run by this bench.py before the timed region", True
            except Exception:
                pass
        for nm in ("r04_ubench2.log", "r03_ubench2.log", "r02_ubench2.log"):
            path = os.path.join(ROOT, "profiles", nm)
            if os.path.exists(path):
                peak, clock = parse(open(path).read())
                if peak:
                    return peak, clock, "profiles/" + nm + " (committed profile, NOT measured in this run)", False
        return None, None, None, False

    alu_result = alu_ceiling()
    regime_result = box_regime(local_rank) if rank == 0 else None
    # before the long multiply loop heats the chip into lower clocks; every rank measures (same thermal history on every GPU),
    # rank 0 reports
    ntt_result = measure_ntt()
    other_result = measure_other_configs() if (world == 1 and not args.skip_other) else None
    if other_result is not None:
        try:
            other_result.setdefault("packed_linear", {})["kernels"] = measure_packed_kernels(alu_result[0])
        except Exception as e:   # a secondary block must not take the headline metric down with it
            other_result.setdefault("packed_linear", {})["kernels"] = {"error": repr(e)[:300]}
    # every rank runs the windows whatever N is (no collective inside; rank 0 reports): the timed region below is shorter than the power
    # manager's memory, so what ran before it decides which regime it sees (MEASUREMENTS.md section 5) - the same history for every N
    sustained_result = measure_sustained(args.sustained_seconds) if args.sustained_seconds > 0 else None

    def tune_form():
        """Which form of the fused multiply this run times.  (1) the library's own measurement repeated on scratch of the timed batch's
        size (dpfhe_ctx_autotune over outs[1]: ~3500 synthetic pairs, 3 launches per form, twice); (2) THREE STEPS per form of the very
        step that is timed (multiply || reduce + gather), twice in opposite orders, best pass per form - the step is what `value`
        measures, and a form that wins alone can lose next to the reduce.  The library's pick stays unless another form's step is
        >= 3 % faster.  Every rank measures; rank 0's pick is broadcast so that all ranks run the same code."""
        names = ctx.variants()
        if not names:
            return
        if args.ct_mul_form != "auto":
            ctx.set_ct_mul_variant(args.ct_mul_form)
            autotune["chosen"], autotune["how"] = args.ct_mul_form, "--ct-mul-form"
            return
        torch.cuda.synchronize()
        autotune["dpfhe_ctx_autotune"] = ctx.autotune(outs[1].view(-1), reps=3, stream=main)
        lib_pick = autotune["dpfhe_ctx_autotune"]["chosen"]
        best = {}
        for order in (names, names[::-1]):
            for nm in order:
                ctx.set_ct_mul_variant(nm)
                step()                                   # untimed: the previous form's reduce drains behind it
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0_) / 3 * 1e3
                best[nm] = min(best.get(nm, ms), ms)
        pick = lib_pick
        fastest = min(best, key=best.get)
        if best[fastest] < 0.97 * best[lib_pick]:
            pick = fastest
        if dist.is_initialized():
            idx = torch.tensor([names.index(pick)], device=dev)
            dist.broadcast(idx, 0)
            pick = names[int(idx.item())]
        ctx.set_ct_mul_variant(pick)
        autotune["step_probe_ms"] = {k_: round(v, 4) for k_, v in best.items()}
        autotune["chosen"] = pick
        autotune["how"] = ("library pick (dpfhe_ctx_autotune), confirmed by three timed steps per form" if pick == lib_pick
                           else "three timed steps per form (>= 3 % faster than the library's pick inside the overlapped step)")

    tune_form()
    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    power = PowerSampler(dev)   # board power / cap / shader clock from the GPU's hwmon files while the timed steps run
    power.start()
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = step(i)
    fence()
    elapsed = time.perf_counter() - t0
    power_result = power.finish()
    out = outs[last]
    per_rank_elapsed = [elapsed]
    if dist.is_initialized():
        # every rank's own clock travels to rank 0 (per-rank rates in the line); the step time is the MAX over ranks
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        every = torch.empty(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(every, t)
        per_rank_elapsed = [float(v) for v in every.tolist()]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # SURVEY.md section 8(e)'s alternative exchange, measured next to the timed one whenever there is more than one rank (after the timed region: `value`
    # is the all-gather step's): the SAME step with a 64-bit sum all-reduce of the partials + one mod-q pass instead of all-gather + local sum.
    collective_ab = None
    if world > 1:
        try:   # (every rank walks the same path: an exception here is raised on all of them, so the ranks stay in step)
            pipe_ar = ShardedMultiplyReduce(ev, B, comm=comm, main=main, collective="allreduce", outs=pipe.outs)   # same output buffers: no second 6 GiB
            for _ in range(max(1, args.warmup)):
                pipe_ar.step(a, b)
            fence()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                k_ar = pipe_ar.step(a, b)
            fence()
            t_ar = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
            dist.all_reduce(t_ar, op=dist.ReduceOp.MAX)
            same = torch.tensor([int(torch.equal(pipe_ar.totals[k_ar], totals[last]))], device=dev)
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
            collective_ab = {"allgather_ms_per_step": elapsed / args.steps * 1e3, "allreduce_ms_per_step": float(t_ar.item()) / args.steps * 1e3,
                             "allreduce_total_equals_allgather_total": bool(same.item()),
                             "what": "the timed step with ncclAllReduce(u64, sum) of one partial per rank + one mod-q pass, against all-gather + local sum (the timed `value`)"}
        except Exception as e:
            collective_ab = {"error": repr(e)[:200]}

    def workgroup_timeline():
        """median microseconds per segment of a quad-form workgroup's life over a 2048-pair launch (steady-state workgroups only): the share
        spent waiting for the first operand word is what two waves per SIMD cannot hide, and what differs between the boxes of the pool"""
        from deeppowers_amd import _cabi
        nb = min(B, 2048)
        if not ctx.variants() or nb < 256:
            return None
        tr = torch.zeros(nb * L * 12, dtype=torch.int64, device=dev)
        for _ in range(2):
            _cabi.check(ctx._lib.dpfhe_debug_ct_mul_trace(ctx.handle, outs[0].data_ptr(), a.data.data_ptr(), b.data.data_ptr(), nb, tr.data_ptr(), main.cuda_stream), "dpfhe_debug_ct_mul_trace")
        torch.cuda.synchronize()
        t = tr.cpu().numpy().view(np.uint64).reshape(nb * L, 12).astype(np.int64)
        rel = (t - t[:, 0].min()) / 100.0
        span = rel[:, 6].max()
        keep = (rel[:, 0] > 0.15 * span) & (rel[:, 6] < 0.85 * span)
        if keep.sum() < 64:
            keep[:] = True
        med = lambda v: round(float(np.median(v[keep])), 2)
        life = rel[:, 6] - rel[:, 0]
        return {"lifetime": med(life), "wait_first_operand": med(rel[:, 1] - rel[:, 0]), "prologue": med(rel[:, 8] - rel[:, 0]), "loads_issued": med(rel[:, 10] - rel[:, 0]),
                "first_operand_complete": med(rel[:, 11] - rel[:, 0]), "forward_x4": med(rel[:, 2] - rel[:, 1]), "tensor_product": med(rel[:, 3] - rel[:, 2]),
                "inverse_x3": med(rel[:, 4] - rel[:, 3]), "stores": med(rel[:, 6] - rel[:, 4]),
                "wait_share_of_lifetime": round(float(np.mean((rel[:, 1] - rel[:, 0])[keep]) / np.mean(life[keep])), 4),
                "launch": f"{nb} pairs, traced quad form (timestamps in scalar registers; same words as the timed kernel)"}

    def first_profile(*names):
        for nm in names:
            path = os.path.join(ROOT, "profiles", nm)
            if os.path.exists(path):
                return path
        return None

    # HBM traffic of the dominant kernel from the committed PMC passes (collected with rocprofv3 --pmc in their own
    # runs, corrected as MI355X_MICROARCH.md prescribes); scaled per ct-mul because traffic is linear in the batch.
    traffic, traffic_src = None, first_profile("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")
    try:
        with open(traffic_src) as f:
            tj = json.load(f)
            traffic = next(tj[k] for k in ("ct_mul_quad_kernel<FoldArith,12,4>", "ct_mul_dual_kernel<FoldArith,12,4>", "ct_mul_kernel<FoldArith,12,4>") if k in tj)["hbm_bytes_per_ct_mul"] * B
    except Exception:
        pass
    # ... and, when rocprofv3 is on this box, MEASURED in this run: tools/live_traffic.py profiles 2048-pair launches of the same kernel form with
    # FETCH_SIZE and WRITE_SIZE in two separate --pmc passes (after the timed region; ~20 s), scaled per ct-mul
    live_traffic = None
    if rank == 0 and world == 1 and not args.no_live_traffic:
        try:
            import subprocess
            torch.cuda.synchronize()
            run = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "live_traffic.py"), autotune.get("chosen") or ""], capture_output=True, text=True, timeout=170)
            live_traffic = json.loads([l for l in run.stdout.splitlines() if l.startswith("{")][-1])
            if "hbm_bytes_per_ct_mul" in live_traffic:
                traffic = live_traffic["hbm_bytes_per_ct_mul"] * B
                traffic_src = None
        except Exception as e:
            live_traffic = {"error": repr(e)[:160]}
    # The bound of this kernel is VALU issue (integer multiply-adds), not HBM: its ceiling is the register-only butterfly loop
    # of tools/ubench2 (same 12-instruction butterfly, no memory traffic, shader clock measured inside the kernel) - measured
    # LIVE by alu_ceiling() before the timed region when tools/bin/ubench2 exists (built by build()), else quoted from the
    # committed profile and labelled so.
    alu_peak, alu_clock, alu_src, alu_live = alu_result
    kernel_ms = [s.elapsed_time(e) for s, e in zip(ev_start, ev_end)]
    k_avg = sum(kernel_ms) / len(kernel_ms) * 1e-3
    gather_us = sorted(s.elapsed_time(e) * 1e3 for s, e in ev_gather)
    alg_bytes = 7 * L * N * 8 * B
    achieved = alg_bytes / k_avg
    bfly_per_s = 7 * L * (N // 2) * 12 * B / k_avg

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
            "collective": ("dpfhe_comm_allgather (RCCL behind the C ABI)" if comm is not None else "torch.distributed all_gather_into_tensor (RCCL)") if (world > 1 or comm is not None or dist.is_initialized()) else "none (one rank)",
            "arith": "fold(2^60-d)" if ctx.uses_fold else "shoup",
            "autotune": autotune,
        },
        # `bound` / `peak` / `frac` keep the contract's meaning (the roofline priced against: algorithmic bytes / launch time / HBM peak); what limits
        # the kernel in practice is VALU issue (`limited_by`), so both fractions are first-class: frac_hbm (= frac) and frac_alu (butterflies/s over the register-only ceiling).
        "roofline": {
            "kernel": {"quad": "ct_mul_quad_kernel<FoldArith,12,4>", "dual": "ct_mul_dual_kernel<FoldArith,12,4,false>"}.get(autotune.get("chosen"), "ct_mul_quad_kernel<FoldArith,12,4>"),
            "bound": "hbm", "limited_by": "valu issue at the board's power cap (see frac_alu, power)", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": achieved / HBM_PEAK, "frac_hbm": achieved / HBM_PEAK, "frac_alu": (bfly_per_s / alu_peak) if alu_peak else None,
            "traffic": traffic,
            "traffic_source": ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes of tools/live_traffic.py after the timed region: 2048-pair launches of the same kernel form), (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per ct-mul x batch"
                               if (live_traffic and "hbm_bytes_per_ct_mul" in live_traffic) else
                               (os.path.relpath(traffic_src, ROOT) + " (committed profile, NOT measured in this run: rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE in their own passes, bytes per launch)") if (traffic and traffic_src) else None),
            "traffic_live": live_traffic,
            "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None,
            "measured_in_this_run": ["achieved", "frac", "frac_hbm", "avg_launch_ms", "power"] + (["frac_alu", "alu.achieved", "alu.peak"] if alu_live else ["alu.achieved"])
                                    + (["traffic"] if (live_traffic and "hbm_bytes_per_ct_mul" in live_traffic) else []),
            "quoted_from_committed_profiles": ([] if (live_traffic and "hbm_bytes_per_ct_mul" in live_traffic) else ["traffic"]) + ([] if alu_live else ["alu.peak", "frac_alu (its denominator)"]),
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": k_avg * 1e3,
            "power": power_result,
            "alu": {"unit": "butterflies/s", "achieved": bfly_per_s, "peak": alu_peak, "peak_clock_mhz": alu_clock,
                    "frac": (bfly_per_s / alu_peak) if alu_peak else None,
                    "peak_source": (alu_src + ": register-only radix-2 butterflies (the kernels' 12-instruction fused butterfly), 8 workgroups per CU, shader clock measured inside the kernel") if alu_src else None,
                    "note": "7 transforms x L limbs x (N/2) log2 N butterflies per ct-mul.  Not in the peak: the dyadic products, canonicalisation, addressing (~15 % of the kernel's VALU instructions) and the clock the chip sustains under HBM load (1.9-2.1 GHz against the loop's 2.3 GHz) - see MEASUREMENTS.md section 5"},
        },
        # "report compute-only and end-to-end": `value` is end-to-end (multiply + shard-local reduce +
        # all-gather + final sum); this is the multiply kernel alone, timed inside the same overlapped steps
        "compute_only_ct_mul_per_s": world * B / k_avg,
        # the collective alone (side stream, HIP events around it), per step: latency-bound (one 384 KiB partial per rank)
        "allgather_us": {"median": gather_us[len(gather_us) // 2], "min": gather_us[0], "max": gather_us[-1]} if (world > 1 or comm is not None or dist.is_initialized()) else None,
    }

    result["roofline"]["autotune_chosen"] = autotune.get("chosen") or autotune["at_ctx_create"].get("chosen")
    if collective_ab is not None:
        result["collective_ab"] = collective_ab
    rates = [B * args.steps / e for e in per_rank_elapsed]
    result["per_rank_ct_mul_per_s"] = {"min": min(rates), "max": max(rates), "ranks": len(rates)}
    detail = {}
    if ntt_result is not None:
        # BASELINE.json metric, second half ("NTT HBM GB/s vs peak"), where the driver's record keeps it: configs[1] in place, the
        # region-bracketed medians (30 samples of 32 back-to-back launches) and, when the windows ran, >= 2 s of launches with the board's
        # power / cap / shader clock over that window and the copy kernel's rate in the same run
        nv = {"workload": "BASELINE configs[1]: 1024 RNS polys x 4 limbs, N=4096, in place (268 435 456 algorithmic bytes per launch)",
              "fwd_frac": ntt_result["fwd"]["frac_of_hbm_peak"], "inv_frac": ntt_result["inv"]["frac_of_hbm_peak"],
              "fwd_us": ntt_result["fwd"]["median_us"], "inv_us": ntt_result["inv"]["median_us"],
              "fwd_GBps": ntt_result["fwd"]["GBps"], "inv_GBps": ntt_result["inv"]["GBps"],
              "fwd_best_us": ntt_result["fwd"]["min_us"], "inv_best_us": ntt_result["inv"]["min_us"],
              "out_of_place_fwd_frac": ntt_result["out_of_place"]["fwd"]["frac_of_hbm_peak"], "out_of_place_inv_frac": ntt_result["out_of_place"]["inv"]["frac_of_hbm_peak"],
              "device_copy_frac": ntt_result["device_copy"]["frac_of_hbm_peak"], "round_trip_exact": ntt_result["round_trip_exact"],
              "method": "HIP events over regions of 32 back-to-back launches, median of 30 regions"}
        if sustained_result is not None:
            sf, si, sc = (sustained_result[k_] for k_ in ("ntt_fwd_configs1_in_place", "ntt_inv_configs1_in_place", "copy_1GiB"))
            nv["sustained_2s"] = {"fwd_frac": sf["frac_of_hbm_peak"], "inv_frac": si["frac_of_hbm_peak"], "fwd_us": sf["us_per_launch"], "inv_us": si["us_per_launch"],
                                  "fwd_frac_of_copy": sf["frac_of_copy"], "inv_frac_of_copy": si["frac_of_copy"], "copy_frac": sc["frac_of_hbm_peak"],
                                  "board_w": sf.get("board_w_mean"), "cap_w": sf.get("cap_w"), "sclk_mhz": sf.get("sclk_mhz_mean"),
                                  "inv_board_w": si.get("board_w_mean"), "inv_sclk_mhz": si.get("sclk_mhz_mean"), "copy_board_w": sc.get("board_w_mean"), "copy_sclk_mhz": sc.get("sclk_mhz_mean"),
                                  "fwd_1GiB_out_of_place_frac": sustained_result["ntt_fwd_1GiB_out_of_place"]["frac_of_hbm_peak"],
                                  "inv_1GiB_out_of_place_frac": sustained_result["ntt_inv_1GiB_out_of_place"]["frac_of_hbm_peak"]}
        result["roofline"]["ntt"] = nv
        # ... and as SCALARS of `roofline` (the driver's parsed record drops nested objects)
        rf = result["roofline"]
        rf.update({"ntt_fwd_frac": nv["fwd_frac"], "ntt_inv_frac": nv["inv_frac"], "ntt_fwd_us": nv["fwd_us"], "ntt_inv_us": nv["inv_us"], "copy_frac": nv["device_copy_frac"]})
        if "sustained_2s" in nv:
            s2 = nv["sustained_2s"]
            rf.update({"ntt_sustained_fwd_frac": s2["fwd_frac"], "ntt_sustained_inv_frac": s2["inv_frac"], "copy_sustained_frac": s2["copy_frac"],
                       "board_w": s2["board_w"], "cap_w": s2["cap_w"], "sclk_mhz": s2["sclk_mhz"]})
        detail["ntt"] = ntt_result
    if sustained_result is not None:
        detail["sustained"] = sustained_result
        cs = sustained_result.get("ct_mul_shard")
        if cs:   # the sustained figure of the headline kernel, next to the ntt_sustained_* scalars
            result["roofline"].update({"ct_mul_sustained_frac": cs["frac_of_hbm_peak"], "ct_mul_sustained_per_s": a.data.shape[0] / (cs["us_per_launch"] * 1e-6),
                                       "ct_mul_sustained_window_s": cs["window_s"], "ct_mul_sustained_board_w": cs.get("board_w_mean"),
                                       "ct_mul_sustained_cap_w": cs.get("cap_w"), "ct_mul_sustained_sclk_mhz": cs.get("sclk_mhz_mean")})
    if rank == 0:
        try:   # where a workgroup of the multiply spends its life ON THIS BOX (diagnostic launch after the timed region; include/dpfhe.h dpfhe_debug_ct_mul_trace)
            regime_result["workgroup_timeline_us"] = workgroup_timeline()
        except Exception as e:
            regime_result["workgroup_timeline_us"] = {"error": repr(e)[:160]}
        result["roofline"]["regime"] = regime_result
    if other_result is not None:
        detail["other_configs"] = other_result
        result["other_configs"] = digest_other(other_result)
        c5 = other_result.get("n8192_l6")
        if c5:
            result["roofline"].update({"n8192_ntt_fwd_frac": c5["ntt_fwd"]["frac_of_hbm_peak"], "n8192_ntt_inv_frac": c5["ntt_inv"]["frac_of_hbm_peak"],
                                       "n8192_ct_mul_frac": c5["ct_mul"]["frac_of_hbm_peak"]})
        sg = other_result.get("shoup_n4096_l4") or {}
        if "ct_mul" in sg:
            result["roofline"].update({"shoup_ntt_fwd_frac": sg["ntt_fwd"]["frac_of_hbm_peak"], "shoup_ntt_inv_frac": sg["ntt_inv"]["frac_of_hbm_peak"],
                                       "shoup_ct_mul_per_s": sg["ct_mul"]["per_s"], "shoup_ct_mul_frac": sg["ct_mul"]["frac_of_hbm_peak"],
                                       "fold_over_shoup_ct_mul": sg["fold_over_shoup_ct_mul"]})
        for key, pre in (("p31_n4096_l4", "p31"), ("wide49_n4096_l4", "wide49"), ("shoup55_n4096_l4", "shoup55")):
            sg = other_result.get(key) or {}
            if "ct_mul" in sg:
                result["roofline"].update({pre + "_ntt_fwd_frac": sg["ntt_fwd"]["frac_of_hbm_peak"], pre + "_ntt_inv_frac": sg["ntt_inv"]["frac_of_hbm_peak"],
                                           pre + "_ct_mul_per_s": sg["ct_mul"]["per_s"], pre + "_ct_mul_frac": sg["ct_mul"]["frac_of_hbm_peak"]})
    # the reduced result of the last step equals a recomputation of the same sequence on the main stream (every rank checks,
    # rank 0 reports; at N>1 the recomputation repeats the all-gather, so all ranks must take part)
    chk = ev.reduce_sum(Ciphertext(out), stream=main)
    if world == 1 and comm is None and not dist.is_initialized():
        torch.cuda.synchronize()
        result["reduce_consistent"] = bool(torch.equal(chk.data, totals[last]))
    else:
        from deeppowers_amd.sharding import allgather_partials
        gathered = allgather_partials(chk.data, comm=comm)
        tot = ev.reduce_sum(Ciphertext(gathered), stream=main)
        torch.cuda.synchronize()
        ok = torch.tensor([int(torch.equal(tot.data, totals[last]))], device=dev)
        if dist.is_initialized():
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        result["reduce_consistent"] = bool(ok.item())

    if world == 1:
        if not args.no_cpu_baseline:
            from oracle.cbind import Oracle
            orc = Oracle.from_params(params)
            # CPUs this process may actually use: the scheduler affinity AND the cgroup CPU quota (the GPU boxes of this pool are
            # 2 x 64-core hosts whose containers get cpu.max = 16 CPUs: 128 threads there measure CFS throttling, not arithmetic)
            hw = len(os.sched_getaffinity(0))
            quota = None
            try:
                q_us, period_us = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
                if q_us != "max":
                    quota = int(q_us) / int(period_us)
            except Exception:
                pass
            cores = max(1, min(orc.max_threads(), hw, int(quota) if quota else hw))
            # single-thread rate first (2 passes over a few pairs), then a SUSTAINED all-thread sample sized from it: three timed
            # passes of ~cpu_seconds/3 each (many scheduler periods long), best pass reported
            n_1 = min(B, 24)
            _, t_one = orc.ct_mul_timed(to_host(a.data[:n_1]), to_host(b.data[:n_1]), threads=1, reps=2)
            single = n_1 / t_one
            n_s = int(min(B, max(cores, single * cores * args.cpu_seconds / 3.0)))
            n_s -= n_s % cores if n_s > cores else 0
            ah, bh = to_host(a.data[:n_s]), to_host(b.data[:n_s])
            cpu_out, t_all = orc.ct_mul_timed(ah, bh, threads=cores, reps=3)
            # the CPU baseline leg doubles as the checker of the timed GPU output: every word of the sample must match
            result["bit_exact_sample"] = bool(np.array_equal(to_host(out[:n_s]), cpu_out))
            try:
                cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
            except Exception:
                cpu_model = "unknown"
            result["cpu_baseline"] = {
                "value": n_s / t_all, "unit": "ct-mul/s", "cores": cores, "kind": "port",
                "sample": f"first {n_s} ciphertext pairs of the same batch, oracle/oracle.c Harvey-NTT evaluator, OpenMP static schedule over "
                          f"{cores} threads on NUMA-local, pre-touched buffers, best of 3 passes of {t_all:.2f} s "
                          f"(reference has no CPU evaluator: build CPU evaluator)",
                "host": f"{cpu_model}; {hw} hardware threads visible, cgroup cpu.max = " + (f"{quota:g} CPUs" if quota else "unlimited")
                        + f" -> {cores} threads used (more threads than the quota only measure scheduler throttling)",
                "isa": orc.isa(),
                "single_thread_value": single,
                "threads_over_single": (n_s / t_all) / single,
                "parallel_efficiency": (n_s / t_all) / single / cores,
            }
            result["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]
            # the quota-limited 16-thread figure flatters the ratio to "the CPU": the per-core ratio sits next to it in the record the driver keeps
            result["cpu_baseline"]["gpu_over_cpu_all_threads"] = result["gpu_over_cpu"]
            result["cpu_baseline"]["gpu_over_one_core"] = result["value"] / single

    if world == 1 and not args.skip_other and "other_configs" in result:
        # For a caller who needs ONLY the sum of the shard's products: the inverse transform is linear, so the products stay in the NTT domain
        # (dpfhe_ct_mul with DPFHE_OUT_NTT), are summed there, and the total alone is transformed back.  NOT the metric (its products are
        # coefficient-domain ciphertexts; here 3 of the 7 transforms per pair never run) - reported next to it because it is what the library offers
        # for BASELINE configs[3]'s job when the individual products are not wanted.  After every check of the timed outputs: it reuses their buffers.
        try:
            pipe_ns = ShardedMultiplyReduce(ev, B, comm=comm, main=main, outs=pipe.outs, sum_in_ntt_domain=True)
            for _ in range(2):
                pipe_ns.step(a, b)
            torch.cuda.synchronize()
            t_ns = time.perf_counter()
            for _ in range(args.steps):
                k_ns = pipe_ns.step(a, b)
            torch.cuda.synchronize()
            t_ns = (time.perf_counter() - t_ns) / args.steps
            result["other_configs"]["sum_in_ntt_domain"] = {
                "ms_per_step": t_ns * 1e3, "pairs_per_s": B / t_ns, "total_equals_the_timed_steps_total": bool(torch.equal(pipe_ns.totals[k_ns], totals[last])),
                "what": "multiply (NTT-domain out) || reduce + ONE inverse transform of the total: same total as the timed step, not the metric op"}
        except Exception as e:
            result["other_configs"]["sum_in_ntt_domain"] = {"error": repr(e)[:200]}
        if isinstance(detail.get("other_configs"), dict):
            detail["other_configs"]["sum_in_ntt_domain"] = result["other_configs"]["sum_in_ntt_domain"]

    if world > 1 and not args.skip_other:
        torch.cuda.synchronize()
        progs = on_rank0_while_others_wait(rank, lambda: multi_gpu_programs(world))
        if rank == 0:
            result["multi_gpu_programs"] = progs
    if rank == 0:
        try:   # the full blocks the line only digests (gpurun merges gpurun_out/ back; elsewhere this is a scratch file)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_detail.json"), "w") as f:
                json.dump(dict(result, **detail), f)
            result["detail_file"] = "gpurun_out/bench_detail.json (ntt, sustained, other_configs in full)"
        except Exception:
            pass
        print(json.dumps(result))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
