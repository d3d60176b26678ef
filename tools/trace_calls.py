"""Prints per-call durations of kernels whose name contains a substring, from a rocprofv3
kernel_trace.csv (development tool):  trace_calls.py trace.csv substring [last_n]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
for r in rows[-n:]:
    print("%9.1f us  grid %s wg %s lds %s  %s" % (
        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", "?") + "x" + r.get("Grid_Size_Y", "?"),
        r.get("Workgroup_Size_X", "?"), r.get("LDS_Block_Size", "?"), r["Kernel_Name"].split("(")[0][-60:]))
