#!/bin/bash
# (through gpurun) kernel-trace stats and two SQ counter passes over tools/bcgs_time.py (V-cycles at 4096^2 with each 2D smoother;
# the rows of interest are k_patch_bcgs2d's):  bash tools/bcgs_profile.sh <outdir under gpurun_out>
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python3 tools/bcgs_time.py > $O/stats.json 2> $O/stats.log || { tail -3 $O/stats.log; exit 1; }
echo "stats done"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY -d $O/sq1 -o c -- python3 tools/bcgs_time.py > $O/sq1.json 2> $O/sq1.log || { tail -3 $O/sq1.log; exit 1; }
echo "sq1 done"
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES -d $O/sq2 -o c -- python3 tools/bcgs_time.py > $O/sq2.json 2> $O/sq2.log || { tail -3 $O/sq2.log; exit 1; }
echo "sq2 done"
python3 - $O <<'PY'
import os, sys
sys.path.insert(0, "tools")
from prof_summary import counters, kernel_stats
O = sys.argv[1]
with open(os.path.join(O, "bcgs_profile.txt"), "w") as out:
    print("== rocprofv3 --kernel-trace --stats: tools/bcgs_time.py (name, calls, total ms, average us)", file=out)
    for r in kernel_stats(os.path.join(O, "stats"))[:8]:
        print(f"{r[0][:100]:100s} {r[1]:6d} {r[2] / 1e6:10.3f} {r[3] / 1e3:10.2f}", file=out)
    c = {}
    for d in ("sq1", "sq2"):
        for k, v in counters(os.path.join(O, d)).items():
            if "bcgs" in k:
                c.setdefault(k, {}).update({n: a[0] / max(a[1], 1) for n, a in v.items()})
    for k, m in c.items():
        print(f"== {k}: per-launch means of the SQ counters (summed over the chip)", file=out)
        for n in sorted(m):
            print(f"   {n:28s} {m[n]:16.0f}", file=out)
        if m.get("SQ_WAVES") and m.get("SQ_INSTS_VALU"):
            print(f"   VALU instructions per wave {m['SQ_INSTS_VALU'] / m['SQ_WAVES']:.0f}, LDS {m.get('SQ_INSTS_LDS', 0) / m['SQ_WAVES']:.0f}, SALU {m.get('SQ_INSTS_SALU', 0) / m['SQ_WAVES']:.0f}", file=out)
        if m.get("SQ_WAVE_CYCLES"):
            print(f"   of a wave's cycles: VALU active {m.get('SQ_ACTIVE_INST_VALU', 0) / m['SQ_WAVE_CYCLES']:.3f}, LDS active {m.get('SQ_ACTIVE_INST_LDS', 0) / m['SQ_WAVE_CYCLES']:.3f}, "
                  f"waiting to issue {m.get('SQ_WAIT_INST_ANY', 0) / m['SQ_WAVE_CYCLES']:.3f}, waiting for LDS {m.get('SQ_WAIT_INST_LDS', 0) / m['SQ_WAVE_CYCLES']:.3f}", file=out)
print(open(os.path.join(O, "bcgs_profile.txt")).read())
PY
