"""Summarise rocprofv3 --pmc output (rocpd sqlite db): per kernel name, average counter value per dispatch."""
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
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in cur.execute(f"select {name_col}, counter_name, value, dispatch_id from {view}"):
    k = re.sub(r"\(.*", "", row[0]).replace("void ", "").replace("dpfhe::", "")
    acc[k][row[1]].append((row[3], row[2]))
for k, ctrs in acc.items():
    if not re.search(sys.argv[2] if len(sys.argv) > 2 else ".", k):
        continue
    print(k[:90])
    for c, vals in sorted(ctrs.items()):
        per = collections.defaultdict(float)
        for d, v in vals:
            per[d] += v
        v = list(per.values())
        print(f"    {c:28s} avg/dispatch {sum(v) / len(v):16.1f}   (n={len(v)})")
