#!/usr/bin/env python3
"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs (two separate passes, as
MI355X_MICROARCH.md prescribes: the TCC block cannot hold both in one pass) into
profiles/traffic.json: average HBM bytes per launch per kernel class.

gfx950 corrections applied (MI355X_MICROARCH.md, HBM): counters are in KiB; FETCH_SIZE reports
exactly half of the bytes of a wide (16 B/lane) coalesced read stream, so it is doubled;
WRITE_SIZE is exact for 16 B/lane streaming stores. All kernels priced here use 16 B/lane.

usage: pmc_traffic.py <fetch_dir> <write_dir> <workload-key> <world> [out.json]   (bench.workload_key, e.g. 3d:u512:p32:rbgs)
"""
import collections
import csv
import glob
import json
import os
import sys

import re

CLASS = [(r"k_rbgs_zero_resid3d<\d+, false, \w+, true", "rbgs_zero_resid_restrict_faces_fcorr"),
         (r"k_rbgs_zero_resid3d<\d+, false", "rbgs_zero_resid_restrict_faces"),
         (r"k_rbgs_resweep_prolong3d<\d+, \d+, true", "rbgs_resweep_prolong_fcorr"), (r"k_rbgs_resweep_prolong3d", "rbgs_resweep_prolong"),
         (r"k_rbgs_zero_resid3d", "rbgs_zero_resid_restrict"), (r"k_restrict_fixup3d", "restrict_fixup"), (r"k_fcorr_gather3d", "fcorr_gather"),
         (r"k_rbgs3d<\d+, \w+, \w+, [248][,>]", "stencil_rbgs_slabs"), (r"k_stencil3d<\d+, \d, [248][,>]", "stencil_slabs"),
         (r"k_rbgs3d<\d+, true", "stencil_rbgs_zero"), (r"k_rbgs3d<\d+, false, true", "stencil_rbgs_prolong"),
         (r"k_rbgs3d<", "stencil_rbgs"), (r"k_stencil3d<\d+, 0,", "stencil_apply"), (r"k_stencil3d<\d+, 1,", "stencil_resid"),
         (r"k_stencil3d<\d+, 2,", "stencil_jacobi"), (r"k_stencil3d<\d+, 3,", "resid_restrict"),
         (r"k_restrict3d", "restrict"), (r"k_prolong3d", "prolong_add"), (r"k_vecop", "vecop"),
         (r"k_dst_axis3d", "dst_axis"), (r"k_patch_rhs3d|k_face_corr3d", "patch_rhs"), (r"k_ps_sym<false, true", "patch_solve_mfma_faces"),
         (r"k_ps_sym|k_ps_fused", "patch_solve_mfma"),
         (r"k_ps_xy|k_ps_z", "patch_solve_3pass")]


def klass(name):
    for pat, c in CLASS:
        if re.search(pat, name):
            return c
    return None


def collect(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            c = klass(r["Kernel_Name"])
            if c:
                agg[c][0] += float(r["Counter_Value"])
                agg[c][1] += 1
    return agg


def main():
    fetch_dir, write_dir, size, world = sys.argv[1:5]
    out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    fe, wr = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    res = json.load(open(out)) if os.path.exists(out) else {}
    for c in sorted(set(fe) | set(wr)):
        fb = 2.0 * 1024 * fe[c][0] / max(fe[c][1], 1)
        wb = 1024.0 * wr[c][0] / max(wr[c][1], 1)
        res[f"{c}:{size}:{world}"] = fb + wb
        print(f"{c:18s} launches {fe[c][1]:4d}/{wr[c][1]:4d}  read {fb/1e6:10.2f} MB  write {wb/1e6:10.2f} MB per launch")
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
