"""Kernel statistics (name, calls, total / average duration) from a rocprofv3 rocpd database.
    rocpd_stats.py results.db [passes] [top]     whole run, totals divided by `passes`
    rocpd_stats.py results.db last [top]         only the LAST replayed step (the kernels between the last two
                                                 optimizer launches): what one HIP-graph replay of the step costs"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"void |at::native::|\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)
    return name[:90]


def main():
    db = sys.argv[1]
    last = len(sys.argv) > 2 and sys.argv[2] == "last"
    steps = 1 if last or len(sys.argv) <= 2 else int(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    if cols:
        name_col = "name" if "name" in cols else "kernel_name"
        rows = c.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
    else:
        # no `kernels` view in this database: join the dispatch and symbol tables (rocpd_kernel_dispatch_<guid>,
        # rocpd_info_kernel_symbol_<guid>) directly
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
        disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")]
        sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")]
        if not disp or not sym:
            raise SystemExit("no kernel tables in %s: %s" % (db, tabs))
        dcols = [r[1] for r in c.execute("pragma table_info(%s)" % disp[0])]
        scols = [r[1] for r in c.execute("pragma table_info(%s)" % sym[0])]
        nm = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else "name")
        rows = c.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start"
                         % (nm, disp[0], sym[0])).fetchall()
    if last:
        idx = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
        g = idx[-1]
        while g - 1 in idx:
            g -= 1
        prev = [i for i in idx if i < g][-1]
        rows = rows[prev + 1:g]
        print("last replayed step: %d launches, span %.3f ms" % (len(rows), (rows[-1][2] - rows[0][1]) / 1e6))
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
