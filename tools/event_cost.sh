pj() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); r=d.get('roofline') or {}; print('$1', round(d['ms_per_step'],4), 'median', round(d.get('ms_per_step_median',0),4), 'dominant avg ms', round(r.get('avg_launch_ms',0),4), 'timed launches', r.get('launches'), 'stride', r.get('event_stride'))"; }
for i in 1 2 3; do
python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | pj "512^3 stride4 "
TE_BENCH_EVENT_STRIDE=1 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | pj "512^3 stride1 "
TE_BENCH_NOPROFILE=1 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | pj "512^3 noevents"
done
for i in 1 2; do
python bench.py --no-secondary --no-cpu-baseline --dim 2 --size 4096 --patch 64 2>/dev/null | pj "4096^2 stride4(5)"
TE_BENCH_EVENT_STRIDE=1 python bench.py --no-secondary --no-cpu-baseline --dim 2 --size 4096 --patch 64 2>/dev/null | pj "4096^2 stride1  "
TE_BENCH_NOPROFILE=1 python bench.py --no-secondary --no-cpu-baseline --dim 2 --size 4096 --patch 64 2>/dev/null | pj "4096^2 noevents "
done
python bench.py --no-secondary --no-cpu-baseline --mesh tests/golden/2refine.bin --divide 3 2>/dev/null | pj "c4 stride4"
TE_BENCH_EVENT_STRIDE=1 python bench.py --no-secondary --no-cpu-baseline --mesh tests/golden/2refine.bin --divide 3 2>/dev/null | pj "c4 stride1"
