#!/usr/bin/env python3
"""gpurun_out/$ROUND/ (scratch, written by scripts/collect_profiles.sh on the GPU box) -> gpurun_out/$ROUND/summary/ (small:
travels back; copied to profiles/$ROUND/, tracked):
per-kernel summaries of every kernel trace, the PMC passes of the headline bench per kernel, pmc_traffic.json
(what bench.py's roofline.traffic quotes) and the bench lines printed under the profiler."""
import csv
import glob
import json
import os
import re
import shutil
import sqlite3
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("ROUND", "r03")
SRC = os.path.join(ROOT, "gpurun_out", ROUND)
DST = os.path.join(SRC, "summary")


def short(name):
    name = re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", ""))
    return re.sub(r"^void ", "", name)


def kernel_stats(db_path, out_csv):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0])
        a[0] += 1
        a[1] += e - s
    total = sum(v[1] for v in agg.values()) or 1
    with open(out_csv, "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
        for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write('"%s",%d,%d,%.1f,%.2f\n' % (name, n, t, t / n, 100.0 * t / total))


def group(name):
    """bench.py's roofline kernel names"""
    if "k_dense" in name:
        return "k_dense"
    if "k_hash_tiles" in name or "k_bm_tiles" in name:
        return "k_bm_tiles|k_hash_tiles2"
    if "k_direct_tiles" in name:
        return "k_direct_tiles"
    if "k_hash<" in name and "true" in name.split(",")[3]:
        return "k_hash(window cells)"
    if "k_hash<" in name:
        return "k_hash(rows)"
    if "k_light" in name:
        return "k_light"
    return None


def main():
    os.makedirs(DST, exist_ok=True)
    for d in sorted(glob.glob(os.path.join(SRC, "trace_*"))):
        name = os.path.basename(d)[len("trace_"):]
        dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
        if dbs:
            kernel_stats(dbs[0], os.path.join(DST, "%s_kernel_stats.csv" % name))
        b = os.path.join(SRC, "bench_%s.json" % name)
        if os.path.exists(b):
            shutil.copy(b, os.path.join(DST, "%s_bench_under_rocprof.json" % name))
    blk = glob.glob(os.path.join(SRC, "trace_block", "**", "*_results.db"), recursive=True)
    if blk:
        with open(os.path.join(DST, "block_step_timeline.txt"), "w") as f:
            subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "step_timeline.py"), blk[0]], stdout=f)
        lg = os.path.join(SRC, "block.log")
        if os.path.exists(lg):
            shutil.copy(lg, os.path.join(DST, "block_rehearsal.log"))
    # PMC passes: per-kernel sums per launch
    per = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(os.path.join(SRC, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    counters = sorted({c for v in per.values() for c in v})
    with open(os.path.join(DST, "cfg2_pmc_per_kernel.csv"), "w") as f:
        f.write("Kernel,Launches," + ",".join(c + "_per_launch" for c in counters) + "\n")
        for k in sorted(per, key=lambda k: -per[k].get("SQ_WAVE_CYCLES", 0.0)):
            n = max(cnt[k].values())
            f.write('"%s",%d,' % (k, n) + ",".join("%.6g" % (per[k][c] / cnt[k][c]) if cnt[k].get(c) else "" for c in counters) + "\n")
    grp = defaultdict(lambda: defaultdict(float))
    for k, v in per.items():
        g = group(k)
        if g:
            for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"):
                if cnt[k].get(c):
                    grp[g][c] += v[c] / cnt[k][c]            # every kernel of a group runs once per bench step
    try:
        commit = os.environ.get("COMMIT") or subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        commit = "?"
    j = {"command": "python3 bench.py --steps 2 --warmup 1 --no-other-configs --no-cpu-baseline", "commit": commit,
         "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum, separate passes (scripts/collect_profiles.sh); "
                   "per-kernel values in cfg2_pmc_per_kernel.csv",
         "correction": "bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE): FETCH_SIZE is in KB and on gfx950 reports half of a coalesced "
                       "stream's bytes (MI355X_MICROARCH.md, HBM section); the factor 2 is calibrated for 16 B/lane streams -- the "
                       "dense-cell loop reads 16-byte pieces, the hash cells gather 12 B tuples, for which it is an upper bound",
         "kernels": {}}
    for g, v in grp.items():
        hit, miss = v.get("TCC_HIT_sum", 0.0), v.get("TCC_MISS_sum", 0.0)
        j["kernels"][g] = {"fetch_size_kb_per_launch": v.get("FETCH_SIZE", 0.0), "write_size_kb_per_launch": v.get("WRITE_SIZE", 0.0),
                           "l2_hit_rate": hit / (hit + miss) if hit + miss else None,
                           "traffic_bytes_per_launch": int(1024 * (2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)))}
    with open(os.path.join(DST, "pmc_traffic.json"), "w") as f:
        json.dump(j, f, indent=1)
    print("wrote", sorted(os.listdir(DST)))


if __name__ == "__main__":
    main()
