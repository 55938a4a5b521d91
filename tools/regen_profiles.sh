#!/bin/bash
# (run through gpurun) every judged file of a round at one commit: tools/regen_profiles.sh <commit>
# rocprofv3 passes (profile_round.sh), traffic.json from them, then the bench lines of the five configurations
set -o pipefail
cd $GRAFT_REPO_ROOT
C=$1
bash tools/profile_round.sh r02f $C > gpurun_out/r02f_profile.log 2>&1 || { tail -5 gpurun_out/r02f_profile.log; exit 1; }
tail -3 gpurun_out/r02f_profile.log
python3 tools/prof_summary.py gpurun_out/r02f gpurun_out/r02f/summary $C "bench.py --steps 20 --warmup 5" 512 1 > /dev/null && cp profiles/traffic.json gpurun_out/r02f_traffic.json
# kernel-trace stats of the secondary benches (reference smoother; C5 in 2D): <dir>/stats is what prof_summary.py reads
for v in ps 2d; do
  if [ $v = ps ]; then A="--smoother patch_solve"; else A="--dim 2 --size 4096 --patch 64"; fi
  mkdir -p gpurun_out/r02f_$v
  rocprofv3 --kernel-trace --stats -d gpurun_out/r02f_$v/stats -o s -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $A > /dev/null 2> gpurun_out/r02f_$v/stats.log || { tail -3 gpurun_out/r02f_$v/stats.log; exit 1; }
  python3 tools/prof_summary.py gpurun_out/r02f_$v gpurun_out/r02f_$v/summary $C "bench.py --steps 20 --warmup 5 --no-cpu-baseline $A" > /dev/null
done
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r02f_bench_512.json 2> gpurun_out/r02f_bench_512.err && echo bench512 ok &&
python3 bench.py --steps 20 --warmup 5 --smoother patch_solve --no-cpu-baseline > gpurun_out/r02f_bench_512_ps.json 2>> gpurun_out/r02f_bench_512.err && echo ps ok &&
python3 bench.py --steps 20 --warmup 5 --size 256 --no-cpu-baseline > gpurun_out/r02f_bench_256.json 2>> gpurun_out/r02f_bench_512.err && echo 256 ok &&
python3 bench.py --steps 20 --warmup 5 --dim 2 --size 4096 --patch 64 --no-cpu-baseline > gpurun_out/r02f_bench_2d.json 2>> gpurun_out/r02f_bench_512.err && echo 2d ok &&
python3 bench.py --steps 20 --warmup 5 --mesh tests/golden/2refine.bin --divide 3 --no-cpu-baseline > gpurun_out/r02f_bench_c4.json 2>> gpurun_out/r02f_bench_512.err && echo c4 ok
