"""Per-launch kNN / FPS kernel durations of the LAST replayed step in a rocprofv3 rocpd database."""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "knn|fps"
rows = c.execute("select name,start,end,grid_x,grid_y,grid_z,workgroup_x,lds_size,vgpr_count from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
g = idx[-1]
while g - 1 in idx:
    g -= 1
prev = [i for i in idx if i < g][-1]
seg = rows[prev + 1:g]
t0 = seg[0][1]
tot = 0.0
for r in seg:
    if re.search(pat, r[0]):
        n = re.sub(r"void |\(anonymous namespace\)::", "", r[0])
        n = re.sub(r"\(.*", "", n)
        print("%8.1f %-38s %8.2f us grid %dx%dx%d wg %d lds %d" % ((r[1] - t0) / 1e3, n, (r[2] - r[1]) / 1e3, r[3] // r[6], r[4], r[5], r[6], r[7]))
        tot += (r[2] - r[1]) / 1e3
print("step span %.1f us, matched %.1f us" % ((seg[-1][2] - t0) / 1e3, tot))
