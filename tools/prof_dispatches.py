"""rocprofv3 --kernel-trace output (rocpd sqlite db) -> per (kernel, grid size) dispatch durations: count, mean, median, min (tool).
The --stats table averages over every launch size of a kernel; bench.py launches the NTT kernels at three batch sizes, so the
figure that has to agree with bench.py's `ntt` block is the one of the 4096-workgroup launches only.
usage: prof_dispatches.py <results.db> <out.txt> "<header>" [kernel-name regex]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
cand = [v for v in views if v.lower() in ("kernels", "kernel_dispatch", "rocpd_kernel_dispatch")] + [v for v in views if "kernel" in v.lower()]
rows, used = [], None
for v in cand:
    cols = [d[0] for d in cur.execute(f"select * from {v} limit 1").description]
    name = next((c for c in ("name", "kernel_name") if c in cols), None)
    if name and "start" in cols and "end" in cols:
        grid = next((c for c in ("grid_size", "grid_size_x", "grid_x", "grid") if c in cols), None)
        wg = next((c for c in ("workgroup_size", "workgroup_size_x", "workgroup_x") if c in cols), None)
        rows = list(cur.execute(f"select {name}, start, end, {grid or 0}, {wg or 1} from {v}"))
        used = v
        break
with open(sys.argv[2], "w") as f:
    f.write("# " + sys.argv[3] + "\n")
    if not rows:
        f.write(f"# no kernel dispatch view with start/end found; views: {views}\n")
        sys.exit(0)
    f.write(f"# view {used}; durations in us (end - start of each dispatch)\n# kernel | grid (work-items) | workgroups | calls | mean_us | median_us | min_us\n")
    pat = sys.argv[4] if len(sys.argv) > 4 else "."
    acc = collections.defaultdict(list)
    for n, s, e, g, wg in rows:
        k = re.sub(r"\(.*", "", n).replace("void ", "").replace("dpfhe::", "")
        if re.search(pat, k):
            acc[(k, int(g or 0), int(wg or 1))].append((e - s) / 1e3)
    for (k, g, wg), ds in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        ds.sort()
        f.write(f"{k[:110]} | {g} | {g // wg if wg else 0} | {len(ds)} | {sum(ds) / len(ds):.2f} | {ds[len(ds) // 2]:.2f} | {ds[0]:.2f}\n")
print(open(sys.argv[2]).read()[:3000])
