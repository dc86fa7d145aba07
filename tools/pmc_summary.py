"""Summarise rocprofv3 --pmc output (rocpd sqlite db): per kernel name (and launch size, so that priming launches do not dilute the
averages), average counter value per dispatch.  usage: pmc_summary.py <results.db> [kernel-name regex]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if view is None:
    print("tables:", tabs)
    sys.exit(0)
cols = [d[0] for d in cur.execute(f"select * from {view} limit 1").description]
name_col = "kernel_name" if "kernel_name" in cols else "name"
grid_col = next((c for c in ("grid_size", "grid_size_x", "grid_x") if c in cols), None)
sel = f"select {name_col}, counter_name, value, dispatch_id" + (f", {grid_col}" if grid_col else ", 0") + f" from {view}"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in cur.execute(sel):
    k = re.sub(r"\(.*", "", row[0]).replace("void ", "").replace("dpfhe::", "")
    acc[(k, row[4])][row[1]].append((row[3], row[2]))
pat = sys.argv[2] if len(sys.argv) > 2 else "."
for (k, grid), ctrs in sorted(acc.items(), key=lambda kv: (kv[0][0], -int(kv[0][1] or 0))):
    if not re.search(pat, k):
        continue
    print(f"{k[:100]}  [grid {grid}]" if grid_col else k[:100])
    for c, vals in sorted(ctrs.items()):
        per = collections.defaultdict(float)
        for d, v in vals:
            per[d] += v
        v = list(per.values())
        print(f"    {c:28s} avg/dispatch {sum(v) / len(v):16.1f}   (n={len(v)})")
