#!/bin/bash
# (run through gpurun) every judged file of a round at one commit: tools/regen_profiles.sh <commit> [tag, default r03]
# rocprofv3 passes (profile_round.sh) for the RB-GS headline and the reference smoother, traffic.json from them, then
# the bench lines of the five configurations. Progress lines go to stdout (a long run must not look hung).
set -o pipefail
cd $GRAFT_REPO_ROOT
C=$1
T=${2:-r03}
W=gpurun_out/${T}f
bash tools/profile_round.sh ${T}f $C > ${W}_profile.log 2>&1 || { tail -5 ${W}_profile.log; exit 1; }
tail -3 ${W}_profile.log
python3 tools/prof_summary.py $W $W/summary $C "bench.py --steps 20 --warmup 5" 3d:u512:p32:rbgs 1 > /dev/null
# the reference smoother: kernel stats + SQ + traffic of its own command
bash tools/profile_round.sh ${T}f_ps $C --smoother patch_solve > ${W}_ps_profile.log 2>&1 || { tail -5 ${W}_ps_profile.log; exit 1; }
tail -2 ${W}_ps_profile.log
python3 tools/prof_summary.py ${W}_ps ${W}_ps/summary $C "bench.py --steps 20 --warmup 5 --smoother patch_solve" 3d:u512:p32:patch_solve 1 > /dev/null
cp profiles/traffic.json ${W}_traffic.json
# kernel-trace stats of C5 in 2D: <dir>/stats is what prof_summary.py reads
mkdir -p ${W}_2d
A="--dim 2 --size 4096 --patch 64"
rocprofv3 --kernel-trace --stats -d ${W}_2d/stats -o s -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $A > /dev/null 2> ${W}_2d/stats.log || { tail -3 ${W}_2d/stats.log; exit 1; }
python3 tools/prof_summary.py ${W}_2d ${W}_2d/summary $C "bench.py --steps 20 --warmup 5 --no-cpu-baseline $A" > /dev/null
echo "2d stats ok"
python3 bench.py --steps 20 --warmup 5 > ${W}_bench_512.json 2> ${W}_bench_512.err && echo bench512 ok &&
python3 bench.py --steps 20 --warmup 5 --smoother patch_solve --no-cpu-baseline > ${W}_bench_512_ps.json 2>> ${W}_bench_512.err && echo ps ok &&
python3 bench.py --steps 20 --warmup 5 --size 256 --no-cpu-baseline > ${W}_bench_256.json 2>> ${W}_bench_512.err && echo 256 ok &&
python3 bench.py --steps 20 --warmup 5 --dim 2 --size 4096 --patch 64 --no-cpu-baseline > ${W}_bench_2d.json 2>> ${W}_bench_512.err && echo 2d ok &&
python3 bench.py --steps 20 --warmup 5 --dim 2 --size 4096 --patch 64 --smoother patch_solve --no-cpu-baseline > ${W}_bench_2d_ps.json 2>> ${W}_bench_512.err && echo 2d ps ok &&
python3 bench.py --steps 20 --warmup 5 --mesh tests/golden/2refine.bin --divide 3 --no-cpu-baseline > ${W}_bench_c4.json 2>> ${W}_bench_512.err && echo c4 ok
