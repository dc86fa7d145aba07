#!/bin/bash
# round 5, GPU call I: is the single-token packed layer launch-bound?  kernel time per application against wall time per application
OUT=gpurun_out/r05i; mkdir -p $OUT; export TMPDIR=/tmp
./examples/encrypted_gpt2_linear qkv 50 text 1 2>&1 | grep -v amdgpu | tee $OUT/wall.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o pl -- ./examples/encrypted_gpt2_linear qkv 50 text 1 > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*.db" | head -1)
python - "$f" <<'P'
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel" in t.lower()][0]
cols = [d[0] for d in cur.execute(f"select * from {view} limit 1").description]
print(view, cols[:20])
rows = list(cur.execute(f"select name, start, end from {view} order by start"))
# the last 50 applications: find the repeating pattern by taking launches after the last lift_qp... simply the last 51*len pattern
names = [r[0].split("(")[0].replace("void dpfhe::", "")[:40] for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("lift_digits")]
starts = idx[-51:]
per = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a - 1:b - 1]   # one application: from the input transform before lift_digits to the next one
    busy = sum(r[2] - r[1] for r in seg); span = seg[-1][2] - seg[0][1]
    per.append((len(seg), busy / 1e3, span / 1e3))
import statistics
print("launches per application:", per[0][0], " kernel-busy us (median):", round(statistics.median(p[1] for p in per), 1), " first-start to last-end us (median):", round(statistics.median(p[2] for p in per), 1))
a, b = starts[-2], starts[-1]
for r in rows[a - 1:b - 1]:
    print(f"  {r[0].split('(')[0].replace('void dpfhe::','')[:60]:60s} {(r[2]-r[1])/1e3:8.1f} us   gap before: {(r[1]-prev)/1e3 if 'prev' in dir() else 0:6.1f} us") ; prev = r[2]
P
rm -rf $OUT/prof
