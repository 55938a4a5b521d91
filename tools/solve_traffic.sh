#!/bin/bash
# (through gpurun) HBM traffic of the kernels te_bicgstab itself launches on 512^3, with and without the cycle's compact x-face
# columns feeding the operator applications (TE_NO_BICG_XF): FETCH_SIZE and WRITE_SIZE in separate passes, per-launch means.
#   bash tools/solve_traffic.sh <outdir under gpurun_out>
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1
mkdir -p $O
for mode in xf noxf; do
  arg=""; [ $mode = noxf ] && arg="TE_NO_BICG_XF"
  rocprofv3 --pmc FETCH_SIZE -d $O/$mode/fetch -o c -- python3 tools/solve_once.py $arg > $O/$mode.fetch.log 2>&1 || { tail -3 $O/$mode.fetch.log; exit 1; }
  echo "$mode fetch done"
  rocprofv3 --pmc WRITE_SIZE -d $O/$mode/write -o c -- python3 tools/solve_once.py $arg > $O/$mode.write.log 2>&1 || { tail -3 $O/$mode.write.log; exit 1; }
  echo "$mode write done"
done
python3 - $O <<'PY'
import os, sys
sys.path.insert(0, "tools")
from prof_summary import counters
O = sys.argv[1]
sites = 512 ** 3
with open(os.path.join(O, "solve_traffic.txt"), "w") as out:
    for mode in ("xf", "noxf"):
        fe, wr = counters(os.path.join(O, mode, "fetch")), counters(os.path.join(O, mode, "write"))
        print(f"== {mode}: per-launch HBM bytes of the solve's level-0 kernels (read = 2 x FETCH_SIZE KiB, write = WRITE_SIZE KiB), B/site at 512^3", file=out)
        for k in sorted(set(fe) | set(wr)):
            a, b = fe.get(k, {}).get("FETCH_SIZE", [0, 1]), wr.get(k, {}).get("WRITE_SIZE", [0, 1])
            rb, wb = 2048.0 * a[0] / max(a[1], 1), 1024.0 * b[0] / max(b[1], 1)
            if rb + wb < 0.5e9:
                continue  # level 0 only
            print(f"{k[:110]:110s} launches {max(a[1], b[1]):4d}  read {rb / 1e9:6.3f} GB  write {wb / 1e9:6.3f} GB  = {(rb + wb) / sites:6.2f} B/site", file=out)
print(open(os.path.join(O, "solve_traffic.txt")).read())
PY
