"""rocprofv3 (rocpd sqlite) -> per-dispatch timeline (start/end in us relative to the first dispatch) to see whether
kernels on different streams really overlap.  usage: prof_timeline.py <results.db> [name-regex] [max rows]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 60
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = next((t for t in tables if t == "kernels"), None)
if view is None:
    print("no `kernels` view; tables:", tables[:40]); sys.exit(1)
cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
rows = list(db.execute(f"select name, start, end, queue_id from {view} order by start"))
t0 = rows[0][1]
n = 0
for name, s, e, qid in rows:
    if pat and not pat.search(name):
        continue
    print(f"{(s - t0) / 1e3:12.1f} {(e - t0) / 1e3:12.1f} {(e - s) / 1e3:10.1f} us  q{qid}  {name[:70]}")
    n += 1
    if n >= limit:
        break
