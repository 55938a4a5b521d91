#!/usr/bin/env python3
"""Copies what tools/regen_profiles.sh left under gpurun_out/ into profiles/ under the round's names, each bench line
stamped with the commit it measured.   usage: install_profiles.py <round tag, e.g. r03> <commit>"""
import json
import os
import shutil
import sys

tag, commit = sys.argv[1:3]
W = f"gpurun_out/{tag}f"
names = {"_bench_512.json": "bench_512_n1.json", "_bench_512_ps.json": "bench_512_n1_patch_solve.json",
         "_bench_256.json": "bench_256_n1.json", "_bench_2d.json": "bench_2d_4096_n1.json",
         "_bench_2d_ps.json": "bench_2d_4096_n1_patch_solve.json",
         "_bench_c4.json": "bench_c4_2refine_div3_n1.json"}
for src, dst in names.items():
    d = json.loads(open(W + src).read().strip().splitlines()[-1])
    d = {"measured_at_commit": commit, **d}
    json.dump(d, open(f"profiles/{tag}_{dst}", "w"), indent=1)
    r = d.get("roofline", {})
    print(dst, round(d["ms_per_step"], 4), d.get("ms_per_step_median") and round(d["ms_per_step_median"], 4), round(d["value"] / 1e9, 1),
          r.get("kernel"), round(r.get("frac", 0), 3), r.get("traffic"))
for src, dst in {"summary_kernel_stats.csv": "kernel_stats_512_rbgs.csv", "summary_pmc_sq.csv": "pmc_sq_512_rbgs.csv",
                 "summary_pmc_fetch_write.csv": "pmc_fetch_write_512_rbgs.csv"}.items():
    shutil.copy(f"{W}/" + src, f"profiles/{tag}_{dst}")
for src, dst in {"summary_kernel_stats.csv": "kernel_stats_512_patch_solve.csv", "summary_pmc_sq.csv": "pmc_sq_512_patch_solve.csv",
                 "summary_pmc_fetch_write.csv": "pmc_fetch_write_512_patch_solve.csv"}.items():
    if os.path.exists(f"{W}_ps/" + src):
        shutil.copy(f"{W}_ps/" + src, f"profiles/{tag}_{dst}")
shutil.copy(f"{W}_2d/summary_kernel_stats.csv", f"profiles/{tag}_kernel_stats_2d_4096.csv")
shutil.copy(f"{W}_traffic.json", "profiles/traffic.json")
