#!/usr/bin/env python3
"""Tooling: one V-cycle of rank 0 as a kernel timeline (start, duration, gap to the previous kernel) from the rocpd
database that `rocprofv3 --kernel-trace --stats -d <dir> -o m -- python3 tools/mr8_budget.py --ranks 8 --agg 16` leaves.
usage: mr8_timeline.py <dir with *_results.db> <commit> [out.txt]"""
import glob
import os
import re
import sqlite3
import sys

d, commit = sys.argv[1:3]
out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
db = sqlite3.connect(glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)[0])
rows = list(db.execute("select name, start, end, duration from kernels order by start"))


def short(n):
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n).replace("te::", "")


names = [short(r[0]) for r in rows]
# a cycle starts with the level-0 pre-sweep; rank 0's unprofiled back-to-back section is cycles 45..85 of its run
# (the level-0 pre-sweep: the pre-sweep that follows a post-sweep -- coarser levels may run the same symbols)
starts = [i for i, n in enumerate(names)
          if n.startswith("k_rbgs_zero_resid3d<32, false") and (i == 0 or names[i - 1].startswith("k_rbgs_resweep_prolong3d"))]
i0, i1 = starts[60], starts[61]
t0 = rows[i0][1]
print(f"# commit {commit}; rocprofv3 --kernel-trace -- python3 tools/mr8_budget.py --ranks 8 --agg 16: one cycle of rank 0 of 8 (512^3, RB-GS,", file=out)
print("# native RCCL back-end in loop-back mode, the rank alone on one MI355X); us", file=out)
print(f"# {'start':>8s} {'duration':>9s} {'gap':>7s}  kernel", file=out)
rccl = 0.0
for i in range(i0, i1):
    n, s, e, du = rows[i]
    gap = (s - rows[i - 1][2]) / 1e3
    if "rccl" in names[i].lower():
        rccl += du / 1e3
    print(f"  {(s - t0) / 1e3:8.1f} {du / 1e3:9.1f} {gap:7.1f}  {names[i][:100]}", file=out)
print(f"# cycle: {(rows[i1][1] - t0) / 1e3:.1f} us, {i1 - i0} kernels, RCCL kernels {rccl:.1f} us", file=out)
