"""Per-kernel durations from a rocprofv3 results database (rocpd sqlite): totals by kernel and, with --timeline N,
the kernels of the last N dispatches in start order.  usage: python tools/rocprof_db.py <results.db> [--timeline N]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]


def short(name):
    import re
    m = re.search(r"N\d+_GLOBAL__N_1(\d+)", name)
    if m:
        k = int(m.group(1))
        i = m.end()
        return name[i:i + k] + name[i + k:i + k + 12].split("E")[0]
    return name[:60]


q = (f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), sum(d.end-d.start) "
     f"from {disp} d join {sym} s on d.kernel_id=s.id group by s.kernel_name order by 6 desc")
for r in cur.execute(q):
    print(f"{short(r[0]):40s} n={r[1]:5d} avg={r[2] / 1e3:9.1f}us min={r[3] / 1e3:9.1f} max={r[4] / 1e3:9.1f} total={r[5] / 1e6:9.3f}ms")
if "--timeline" in sys.argv:
    n = int(sys.argv[sys.argv.index("--timeline") + 1])
    rows = list(cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from {disp} d join {sym} s "
                            f"on d.kernel_id=s.id order by d.start"))[-n:]
    t0 = rows[0][1]
    for r in rows:
        print(f"{short(r[0]):40s} start {(r[1] - t0) / 1e3:9.1f} dur {(r[2] - r[1]) / 1e3:8.1f} us  grid {r[3] // max(1, r[4])} x {r[4]}")
