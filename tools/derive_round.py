"""Derive <tag>_valu_busy.txt and <tag>_pmc_traffic.json from the PMC pass summaries collect_round.sh wrote (tool).

usage: python tools/derive_round.py gpurun_out/r02 r02      (reads <dir>/pmc_*.txt, writes <dir>/valu_busy.txt, <dir>/pmc_traffic.json)

  clock     = SQ_BUSY_CYCLES / 32 shader engines / dispatch duration           (counters and durations of the same launches)
  VALU busy = SQ_ACTIVE_INST_VALU * 4 cycles / 1024 SIMDs / (SQ_BUSY_CYCLES / 32)
  HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024    (gfx950: FETCH_SIZE counts 128-B requests at 64 B; MI355X_MICROARCH.md, HBM section)
"""
import json, os, re, sys


def parse_pass(path):
    """{(kernel, grid): {counter: avg}}"""
    out, key = {}, None
    if not os.path.exists(path):
        return out
    for line in open(path):
        m = re.match(r"^(\S.*?)\s+\[grid (\d+)\]", line)
        if m:
            key = (m.group(1).strip(), int(m.group(2)))
            out[key] = {}
            continue
        m = re.match(r"^\s+(\w+)\s+avg/dispatch\s+([\d.]+)", line)
        if m and key:
            out[key][m.group(1)] = float(m.group(2))
    return out


def parse_durations(path):
    """{short kernel name: avg_us} from a prof_summary.py table"""
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        if line.startswith("#") or "|" not in line:
            continue
        cols = [c.strip() for c in line.split("|")]
        m = re.search(r"dpfhe::(\w+)<dpfhe::(\w+), (\d+), (\d+)", cols[0])
        if m:
            out[f"{m.group(1)}<{m.group(2)}, {m.group(3)}, {m.group(4)}"] = float(cols[3])
    return out


def main():
    d, tag = sys.argv[1], sys.argv[2]
    sq = parse_pass(os.path.join(d, "pmc_ntt_pass1.txt"))
    dur = parse_durations(os.path.join(d, "pmc_ntt_pass1_durations.txt"))
    lines = [f"# derived by tools/derive_round.py from {tag}_pmc_ntt_pass1.txt + {tag}_pmc_ntt_pass1_durations.txt (one rocprofv3 --pmc SQ_* --kernel-trace run of tools/ntt_bench.py 8192 2048):",
             "# clock = SQ_BUSY_CYCLES / 32 shader engines / dispatch duration;  VALU busy = SQ_ACTIVE_INST_VALU * 4 cycles / 1024 SIMDs / (SQ_BUSY_CYCLES / 32)",
             "# kernel | grid | avg dispatch us | shader clock GHz | VALU busy | VALU inst per wave | cycles per VALU inst while active | wave time parked (s_waitcnt/barrier)"]
    for (k, grid), c in sorted(sq.items()):
        short = next((s for s in dur if k.startswith(s)), None)
        if short is None or "SQ_BUSY_CYCLES" not in c:
            continue
        us = dur[short]
        busy = c["SQ_BUSY_CYCLES"] / 32.0
        clock = busy / us / 1e3
        valu_busy = c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / busy
        per_wave = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
        cyc = c["SQ_ACTIVE_INST_VALU"] * 4 / c["SQ_INSTS_VALU"]
        parked = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        lines.append(f"{k} | {grid} | {us:.1f} | {clock:.2f} | {valu_busy:.3f} | {per_wave:.0f} | {cyc:.2f} | {parked:.3f}")
    open(os.path.join(d, "valu_busy.txt"), "w").write("\n".join(lines) + "\n")

    fetch = parse_pass(os.path.join(d, "pmc_bench_pass1.txt"))
    write = parse_pass(os.path.join(d, "pmc_bench_pass2.txt"))
    traffic = {
        "source": f"profiles/{tag}_pmc_bench_pass1.txt (FETCH_SIZE), {tag}_pmc_bench_pass2.txt (WRITE_SIZE): rocprofv3 --pmc in two separate passes over `python tools/ntt_bench.py 1024 8192` (configs[1]'s 4096 residue polynomials, configs[3]'s 8192-pair shard); per-dispatch averages, grouped by grid size",
        "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts 128-B requests at 64 B, MI355X_MICROARCH.md section HBM)",
    }
    want = {"ct_mul_quad_kernel<FoldArith,12,4>": ("ct_mul_quad_kernel<FoldArith, 12, 4>", 8192 * 4 * 256),
            "ntt_fwd_kernel<FoldArith,12,4>": ("ntt_fwd_kernel<FoldArith, 12, 4>", 4096 * 256),
            "ntt_inv_kernel<FoldArith,12,4>": ("ntt_inv_kernel<FoldArith, 12, 4>", 4096 * 256)}
    def lookup(table, k, grid, counter):   # kernel names carry further template arguments (non-temporal / trace / prefetch flags): prefix match, plain form first
        stem = k[:-1]
        hits = sorted((kk for (kk, g) in table if g == grid and kk.startswith(stem) and "true" not in kk[len(stem):]), key=len)
        return table[(hits[0], grid)].get(counter) if hits else None
    for name, (k, grid) in want.items():
        f = lookup(fetch, k, grid, "FETCH_SIZE")
        w = lookup(write, k, grid, "WRITE_SIZE")
        if f is None or w is None:
            continue
        total = (2 * f + w) * 1024
        e = {"fetch_size_kib": f, "write_size_kib": w, "hbm_bytes_per_launch": total}
        if name.startswith("ct_mul"):
            e.update(batch=8192, hbm_bytes_per_ct_mul=total / 8192, algorithmic_bytes_per_ct_mul=7 * 4 * 4096 * 8)
        else:
            e.update(residue_polys=4096, algorithmic_bytes_per_launch=2 * 4096 * 8 * 4096)
        traffic[name] = e
    json.dump(traffic, open(os.path.join(d, "pmc_traffic.json"), "w"), indent=1)
    print(open(os.path.join(d, "valu_busy.txt")).read())
    print(json.dumps({k: v for k, v in traffic.items() if isinstance(v, dict)}, indent=1))


if __name__ == "__main__":
    main()
