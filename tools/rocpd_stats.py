"""Kernel statistics (name, calls, total / average duration) from a rocprofv3 rocpd database, restricted to the
LAST `--steps` replays when --per-step N is given (total / N).  Usage: rocpd_stats.py results.db [steps] [top]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"void |at::native::|\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)
    return name[:90]


def main():
    db = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = c.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0])
        a[0] += 1
        a[1] += e - s
    tot = sum(a[1] for a in agg.values())
    print("%-92s %8s %10s %9s %6s" % ("kernel", "calls", "total_ms", "avg_us", "%"))
    for n, (k, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%-92s %8d %10.3f %9.2f %6.2f" % (n, k, t / 1e6, t / k / 1e3, 100.0 * t / tot))
    print("total kernel time %.3f ms over %d launches; per step (/%d): %.3f ms, %.1f launches" %
          (tot / 1e6, len(rows), steps, tot / 1e6 / steps, len(rows) / steps))


if __name__ == "__main__":
    main()
