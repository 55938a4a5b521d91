#!/usr/bin/env python3
"""Summarises the rocprofv3 runs of tools/profile_round.sh into small CSVs for profiles/ (each stamped with the
commit it measured): kernel-trace stats, SQ counters (two passes of 8), FETCH_SIZE / WRITE_SIZE (separate passes;
FETCH_SIZE doubled for 16-B/lane reads as MI355X_MICROARCH.md prescribes for gfx950).

usage: prof_summary.py <run dir> <out prefix> <commit> <label> [workload-key world]
With the workload key (bench.py workload_key, e.g. 3d:u512:p32:rbgs) and world it also rewrites profiles/traffic.json (bytes per launch per kernel class, what bench.py quotes as
roofline.traffic) stamped with the commit and a hash of the kernel sources it was measured on.
"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("te::", "")


def counters(d):
    """{kernel: {counter: [sum, launches]}} from rocprofv3's CSV or rocpd (sqlite) output"""
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        import sqlite3
        db = sqlite3.connect(f)
        # one row per (dispatch, counter, dimension instance): sum the instances of a dispatch first
        q = "select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection group by 1, 2, 3"
        for name, cn, _, val in db.execute(q):
            a = agg[short(name)][cn]
            a[0] += float(val)
            a[1] += 1
    return agg


def kernel_stats(d):
    """rows (name, calls, total ns, average ns, percentage, min ns, max ns) from --stats CSV or the rocpd database"""
    st = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if st:
        rows = list(csv.reader(open(st[0])))[1:]
        return [(r[0], int(r[1]), float(r[2]), float(r[3]), float(r[4]), float(r[5]), float(r[6])) for r in rows]
    out = []
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        import sqlite3
        db = sqlite3.connect(f)
        rows = list(db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name"))
        tot = sum(r[2] for r in rows) or 1
        out = [(r[0], r[1], float(r[2]), float(r[3]), 100.0 * r[2] / tot, float(r[4]), float(r[5])) for r in rows]
        out.sort(key=lambda r: -r[2])
    return out


def main():
    run, out, commit, label = sys.argv[1:5]
    head = f"# commit {commit}; {label}\n"
    # 1. kernel stats
    rows = kernel_stats(os.path.join(run, "stats"))
    if rows:
        with open(out + "_kernel_stats.csv", "w") as f:
            f.write(head)
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                if r[4] >= 0.05:
                    w.writerow([short(r[0]), r[1], f"{r[2]:.0f}", f"{r[3]:.1f}", f"{r[4]:.2f}", f"{r[5]:.0f}", f"{r[6]:.0f}"])
    # 2. SQ counters
    sq = counters(os.path.join(run, "sq1"))
    for extra in ("sq2", "sq3"):
        for k, v in counters(os.path.join(run, extra)).items():
            sq[k].update(v)
    if sq:
        names = sorted({c for v in sq.values() for c in v})
        with open(out + "_pmc_sq.csv", "w") as f:
            f.write(head)
            f.write("# per-launch means; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (MI355X_MICROARCH.md, PMC slots)\n")
            w = csv.writer(f)
            f.write("# mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): share of the time a matrix pipe of a busy CU works\n")
            w.writerow(["kernel", "launches"] + names + ["wait_any_frac", "wait_inst_frac", "active_frac", "lds_conflict_frac", "mfma_busy_frac"])
            for k, v in sorted(sq.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", [0, 1])[0]):
                if not k.startswith("k_"):
                    continue
                m = {c: v[c][0] / max(v[c][1], 1) for c in v}
                wc = m.get("SQ_WAVE_CYCLES", 0) or 1
                lds = m.get("SQ_LDS_IDX_ACTIVE", 0) or 1
                w.writerow([k, max(x[1] for x in v.values())] + [f"{m.get(c, 0):.0f}" for c in names]
                           + [f"{m.get('SQ_WAIT_ANY', 0) / wc:.3f}", f"{m.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}",
                              f"{m.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f}", f"{m.get('SQ_LDS_BANK_CONFLICT', 0) / lds:.3f}",
                              f"{m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4.0 * m['SQ_BUSY_CU_CYCLES']):.3f}" if m.get("SQ_BUSY_CU_CYCLES") else ""])
    # 3. traffic
    fe, wr = counters(os.path.join(run, "fetch")), counters(os.path.join(run, "write"))
    if fe or wr:
        with open(out + "_pmc_fetch_write.csv", "w") as f:
            f.write(head)
            f.write("# per-launch means in bytes: read = 2 x FETCH_SIZE KiB x 1024 (gfx950: wide reads are tallied at half), write = WRITE_SIZE KiB x 1024\n")
            w = csv.writer(f)
            w.writerow(["kernel", "launches", "read_bytes", "write_bytes", "total_bytes"])
            for k in sorted(set(fe) | set(wr)):
                if not k.startswith("k_"):
                    continue
                a, b = fe.get(k, {}).get("FETCH_SIZE", [0, 1]), wr.get(k, {}).get("WRITE_SIZE", [0, 1])
                rb, wb = 2048.0 * a[0] / max(a[1], 1), 1024.0 * b[0] / max(b[1], 1)
                w.writerow([k, max(a[1], b[1]), f"{rb:.0f}", f"{wb:.0f}", f"{rb + wb:.0f}"])
        if len(sys.argv) > 6:
            import json
            sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from pmc_traffic import klass
            import bench
            size, world = sys.argv[5], sys.argv[6]  # size: bench.workload_key() of the profiled command
            if re.fullmatch(r"\d+", size):
                size = f"3d:u{size}:p32:rbgs"
            tot = collections.defaultdict(lambda: [0.0, 0.0, 0, 0])
            for k in set(fe) | set(wr):
                c = klass("te::" + k + "(")
                if not c:
                    continue
                a, b = fe.get(k, {}).get("FETCH_SIZE", [0, 0]), wr.get(k, {}).get("WRITE_SIZE", [0, 0])
                t = tot[c]
                t[0] += 2048.0 * a[0]
                t[1] += 1024.0 * b[0]
                t[2] += a[1]
                t[3] += b[1]
            tf = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
            res = {}
            if os.path.exists(tf):  # figures of other workloads measured on the same kernel sources stay
                old = json.load(open(tf))
                if old.get("kernel_sources_sha") == bench.kernel_sources_sha():
                    res = old
            res.update({"commit": commit, "kernel_sources_sha": bench.kernel_sources_sha(),
                   "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py; bytes per launch = "
                          "2 x FETCH_SIZE KiB x 1024 + WRITE_SIZE KiB x 1024 (gfx950: wide reads are tallied at half); "
                          "keys: <kernel class>:<bench.workload_key>:<ranks>"})
            for c, t in sorted(tot.items()):
                res[f"{c}:{size}:{world}"] = t[0] / max(t[2], 1) + t[1] / max(t[3], 1)
            json.dump(res, open(tf, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
