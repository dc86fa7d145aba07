"""Static instruction statistics of the gfx950 kernels in one translation unit (tool).
    python tools/isa_stats.py deeppowers_amd/csrc/k_ctmul_fold.hip [name filter] [-D...]
Compiles device-only to assembly and prints, per kernel matching the filter: instructions by class per basic block (label),
so that loop bodies can be weighted by hand, and the code size against the 64 KiB instruction cache."""
import re, subprocess, sys, tempfile, os

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
out = os.path.join(tempfile.mkdtemp(), "k.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", out, src] + extra)
lines = open(out).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if l.startswith("_ZN5dpfhe") and ":" in l and not l.startswith("\t")]
for idx, (i, name) in enumerate(starts):
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in dem:
        continue
    j = starts[idx + 1][0] if idx + 1 < len(starts) else len(lines)
    body = [l.strip() for l in lines[i + 1:j]]
    end = max((k for k, l in enumerate(body) if l.startswith("s_endpgm")), default=len(body) - 1)
    code = next((l for l in body if "codeLenInByte" in l), "")
    print(dem.split("(")[0], code)
    blocks, cur, curname = [], [], "entry"
    for l in body[:end + 1]:
        if not l or l.startswith((";", "//")) or (l.startswith(".") and not l.endswith(":")):
            continue
        if l.endswith(":") or re.match(r"^\.LBB\w+:", l):
            blocks.append((curname, cur)); cur, curname = [], l.split(":")[0]
            continue
        cur.append(l.split(";")[0].strip())
    blocks.append((curname, cur))
    tot = {"valu": 0, "salu": 0, "ds": 0, "vmem": 0, "all": 0}
    for bn, ins in blocks:
        if not ins:
            continue
        c = lambda p: sum(1 for x in ins if x.startswith(p))
        row = {"valu": c("v_"), "salu": c("s_"), "ds": c("ds_"), "vmem": c("global_") + c("buffer_") + c("scratch_") + c("flat_"), "all": len(ins)}
        for k in tot:
            tot[k] += row[k]
        br = [x for x in ins if x.startswith(("s_cbranch", "s_branch"))]
        if row["all"] > 50 or br:
            print(f"   {bn:14s} insts {row['all']:6d}  valu {row['valu']:6d}  salu {row['salu']:5d}  ds {row['ds']:4d}  vmem {row['vmem']:4d}  {' '.join(b.replace('s_cbranch_', 'br_').replace('s_branch', 'jmp') for b in br)}")
    print(f"   {'TOTAL (static)':14s} insts {tot['all']:6d}  valu {tot['valu']:6d}  salu {tot['salu']:5d}  ds {tot['ds']:4d}  vmem {tot['vmem']:4d}")
