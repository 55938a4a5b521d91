#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag> <commit> [bench args...]
# kernel-trace stats of the default bench command, then counter passes (own runs, --pmc only, as the pool requires):
# two SQ passes, FETCH_SIZE, WRITE_SIZE. Summaries -> gpurun_out/<tag>/summary_*.csv (copy into profiles/).
set -o pipefail
TAG=$1; COMMIT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary "$@" > $OUT/stats.json 2> $OUT/stats.log || { tail -5 $OUT/stats.log; exit 1; }
echo "stats pass done"
pass() { # name, counters
  rocprofv3 --pmc $2 -d $OUT/$1 -o c -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary "${@:3}" > $OUT/$1.json 2> $OUT/$1.log || { echo "pass $1 failed"; tail -3 $OUT/$1.log; return 1; }
  echo "pass $1 done"
}
pass sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "$@" &&
pass sq2 "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "$@"
# the matrix pipe and the LDS next to each other (which unit is saturated, if any: k_ps_sym); a pass of its own that may fail
# (an unknown counter name fails the whole pass) without taking the others with it
pass sq3 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA" "$@" || true
pass fetch "FETCH_SIZE" "$@" && pass write "WRITE_SIZE" "$@"
python3 tools/prof_summary.py $OUT $OUT/summary $COMMIT "bench.py --steps 20 --warmup 5 $*" || true
ls -la $OUT/summary_*
