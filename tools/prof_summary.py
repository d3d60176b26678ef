"""Prints a per-step summary of a rocprofv3 kernel_stats.csv (development tool)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total GPU ms/step %.3f over %d kernels" % (tot / steps / 1e6, len(rows)))
for r in rows[:top]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0][:90]
    print("%8.1f us/step %7.1f calls/step %8.1f us avg %6.2f%%  %s" % (
        int(r["TotalDurationNs"]) / steps / 1e3, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3,
        float(r["Percentage"]), n))
