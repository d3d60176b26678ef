"""Sums rocprofv3 --pmc counter_collection.csv per kernel (development tool)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in rows:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r.get("Dispatch_Id"))
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print("%-60s launches %5d  " % (k, cnt[k]) + "  ".join("%s=%.4g/launch" % (c, v / max(cnt[k], 1)) for c, v in acc[k].items()))
