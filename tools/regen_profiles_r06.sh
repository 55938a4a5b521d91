#!/bin/bash
# (run through gpurun) round 6's judged files at one commit: tools/regen_profiles_r06.sh <commit>
# = tools/regen_profiles.sh (rocprofv3 stats + counters of the RB-GS headline and the reference smoother, traffic.json, the 2D stats, the
# bench lines of the five configurations) + counters of C4 (2refine --divide 3) and of C5 under the reference smoother (SQ counters of
# k_patch_solve2d_sym), the loop-back budgets of the sharded cycle, the tail stamps. Progress lines on stdout.
set -o pipefail
cd $GRAFT_REPO_ROOT
C=$1
PART=${2:-all}   # base | extra | all (one gpurun call is at most 20 minutes: base and extra fit one each)
T=r06
W=gpurun_out/${T}f
if [ "$PART" != "extra" ]; then
bash tools/regen_profiles.sh $C $T || exit 1
echo "base regen done"
fi
[ "$PART" = "base" ] && exit 0
bash tools/profile_round.sh ${T}f_c4 $C --mesh tests/golden/2refine.bin --divide 3 > ${W}_c4_profile.log 2>&1 || { tail -5 ${W}_c4_profile.log; exit 1; }
python3 tools/prof_summary.py ${W}_c4 ${W}_c4/summary $C "bench.py --steps 20 --warmup 5 --mesh tests/golden/2refine.bin --divide 3" "3d:2refine.bin+3:p32:rbgs" 1 > /dev/null
echo "c4 counters done"
bash tools/profile_round.sh ${T}f_2dps $C --dim 2 --size 4096 --patch 64 --smoother patch_solve > ${W}_2dps_profile.log 2>&1 || { tail -5 ${W}_2dps_profile.log; exit 1; }
echo "2d patch_solve counters done"
cp profiles/traffic.json ${W}_traffic.json
python3 tools/mr8_budget.py --size 512 --smoother rbgs --agg 64 --out ${W}_mr8_budget_rccl.txt > /dev/null 2> ${W}_mr8.err && echo "mr8 rccl ok"
python3 tools/mr8_budget.py --size 512 --smoother rbgs --agg 64 --push --ranks 2,4,8 --out ${W}_mr8_budget_push.txt > /dev/null 2>> ${W}_mr8.err && echo "mr8 push ok"
python3 tools/mr8_budget.py --size 512 --smoother patch_solve --agg 64 --ranks 1,8 --out ${W}_mr8_budget_ps.txt > /dev/null 2>> ${W}_mr8.err && echo "mr8 ps ok"
python3 tools/mr8_budget.py --dim 2 --size 4096 --agg 64 --ranks 1,8 --out ${W}_mr8_budget_2d.txt > /dev/null 2>> ${W}_mr8.err && echo "mr8 2d ok"
python3 bench.py --steps 10 --warmup 3 --size 1024 --no-cpu-baseline > ${W}_bench_1024.json 2>> ${W}_bench_512.err && echo "1024 ok"
TE_BENCH_BACKEND=gloo timeout -k 10 600 python3 bench.py --gpus 4 --size 512 > ${W}_rehearsal_512_n4.json 2> ${W}_rehearsal.err && echo "rehearsal 512 ok"
TE_BENCH_BACKEND=gloo timeout -k 10 600 python3 bench.py --gpus 4 --size 1024 --no-cpu-baseline > ${W}_rehearsal_1024_n4.json 2>> ${W}_rehearsal.err && echo "rehearsal 1024 ok"
echo "all done"
