"""rocprofv3 (rocpd sqlite output) -> small text summary for profiles/.  usage: prof_summary.py <results.db> <out.txt> "<header>" """
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
with open(sys.argv[2], "w") as f:
    f.write("# " + sys.argv[3] + "\n# kernel | calls | total_us | avg_us | pct_of_gpu_time\n")
    for n, calls, tot, avg, pct in rows:
        n = n if len(n) <= 150 else n[:147] + "..."
        f.write(f"{n} | {calls} | {tot:.1f} | {avg:.2f} | {pct:.2f}\n")
print(open(sys.argv[2]).read()[:1200])
