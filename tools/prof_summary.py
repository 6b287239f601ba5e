"""rocprofv3 results.db (rocpd) -> per-kernel csv summary (+ printed table).   python tools/prof_summary.py <db> <out.csv> [launch_divisor]"""
import re
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
div = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
c = sqlite3.connect(db).cursor()
rows = list(c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
with open(out, "w") as f:
    f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
    for r in rows:
        f.write('"%s",%d,%d,%.1f,%.2f,%d,%d\n' % (r[0], r[1], r[2], r[3], 100 * r[2] / tot, r[4], r[5]))
for r in rows[:int(sys.argv[4]) if len(sys.argv) > 4 else 30]:
    n = re.sub(r"^void ", "", r[0])
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*", "", n)
    print("%-62s %5d %8.3f ms %8.1f us %5.1f%%" % (n[:62], r[1], r[2] / 1e6 / div, r[3] / 1e3, 100 * r[2] / tot))
print("total kernel time / divisor: %.3f ms" % (tot / 1e6 / div))
