#!/usr/bin/env python3
"""Tooling (build container, after tools/regen_profiles_r06.sh ran on a GPU box and gpurun merged gpurun_out/r06f*): copies the judged
summaries into profiles/r06_* (bench lines stamped with the commit), rebuilds profiles/traffic.json from the counter databases.
usage: collect_profiles_r06.py <commit>"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = sys.argv[1]
W = os.path.join(ROOT, "gpurun_out", "r06f")
P = os.path.join(ROOT, "profiles")
for run, label, key in (("", "bench.py --steps 20 --warmup 5", "3d:u512:p32:rbgs"),
                        ("_ps", "bench.py --steps 20 --warmup 5 --smoother patch_solve", "3d:u512:p32:patch_solve"),
                        ("_c4", "bench.py --steps 20 --warmup 5 --mesh tests/golden/2refine.bin --divide 3", "3d:2refine.bin+3:p32:rbgs"),
                        ("_2dps", "bench.py --steps 20 --warmup 5 --dim 2 --size 4096 --patch 64 --smoother patch_solve", "2d:u4096:p64:patch_solve")):
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"), W + run, W + run + "/summary", C, label, key, "1"], check=True,
                   stdout=subprocess.DEVNULL)
for src, dst in (("_bench_512.json", "r06_bench_512_n1.json"), ("_bench_512_ps.json", "r06_bench_512_n1_patch_solve.json"),
                 ("_bench_256.json", "r06_bench_256_n1.json"), ("_bench_2d.json", "r06_bench_2d_4096_n1.json"),
                 ("_bench_2d_ps.json", "r06_bench_2d_4096_n1_patch_solve.json"), ("_bench_c4.json", "r06_bench_c4_2refine_div3_n1.json"),
                 ("_bench_1024.json", "r06_bench_1024_n1.json"), ("_rehearsal_512_n4.json", "r06_rehearsal_512_n4.json"),
                 ("_rehearsal_1024_n4.json", "r06_rehearsal_1024_n4.json")):
    d = json.loads(open(W + src).read().strip().split("\n")[-1])
    json.dump({"measured_at_commit": C, **d}, open(os.path.join(P, dst), "w"), indent=1)
    r = d.get("roofline") or {}
    print(f"{dst:44s} {d['ms_per_step']:8.4f} ms  median {d.get('ms_per_step_median', 0):8.4f}  {d['value'] / 1e9:6.1f} G/s  {d['u_checksum_after_timed_region']}  "
          f"{r.get('kernel')} {r.get('frac', 0):.3f} traffic {r.get('traffic')}")
for a, b in (("/summary_kernel_stats.csv", "r06_kernel_stats_512_rbgs.csv"), ("/summary_pmc_fetch_write.csv", "r06_pmc_fetch_write_512_rbgs.csv"),
             ("/summary_pmc_sq.csv", "r06_pmc_sq_512_rbgs.csv"), ("_ps/summary_kernel_stats.csv", "r06_kernel_stats_512_patch_solve.csv"),
             ("_ps/summary_pmc_fetch_write.csv", "r06_pmc_fetch_write_512_patch_solve.csv"), ("_ps/summary_pmc_sq.csv", "r06_pmc_sq_512_patch_solve.csv"),
             ("_c4/summary_kernel_stats.csv", "r06_kernel_stats_c4_2refine_div3.csv"), ("_c4/summary_pmc_fetch_write.csv", "r06_pmc_fetch_write_c4_2refine_div3.csv"),
             ("_c4/summary_pmc_sq.csv", "r06_pmc_sq_c4_2refine_div3.csv"), ("_2dps/summary_kernel_stats.csv", "r06_kernel_stats_2d_4096_patch_solve.csv"),
             ("_2dps/summary_pmc_sq.csv", "r06_pmc_sq_2d_4096_patch_solve.csv"), ("_2dps/summary_pmc_fetch_write.csv", "r06_pmc_fetch_write_2d_4096_patch_solve.csv"),
             ("_2d/summary_kernel_stats.csv", "r06_kernel_stats_2d_4096.csv")):
    shutil.copy(W + a, os.path.join(P, b))


def cat(out, header, parts):
    with open(os.path.join(P, out), "w") as f:
        f.write(header + "\n")
        for title, path in parts:
            if title:
                f.write(title + "\n")
            f.write(open(path).read() + "\n")


cat("r06_mr8_budget.txt", f"# measured at commit {C}: tools/mr8_budget.py --agg 64, loop-back, one rank of N alone on one MI355X (round 5: "
    "profiles/r05_mr8_budget.txt -- N = 8: 355 us RCCL / 312 us direct stores)",
    (("# ---- native RCCL back-end", W + "_mr8_budget_rccl.txt"), ("# ---- direct-store transport", W + "_mr8_budget_push.txt")))
cat("r06_mr8_budget_patch_solve.txt", f"# measured at commit {C}: tools/mr8_budget.py --smoother patch_solve --agg 64 --ranks 1,8", (("", W + "_mr8_budget_ps.txt"),))
cat("r06_mr8_budget_2d.txt", f"# measured at commit {C}: tools/mr8_budget.py --dim 2 --size 4096 --agg 64 --ranks 1,8", (("", W + "_mr8_budget_2d.txt"),))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
t = json.load(open(os.path.join(P, "traffic.json")))
print("traffic.json:", t["kernel_sources_sha"], "current sources:", bench.kernel_sources_sha())
