"""Turn rocprofv3 --pmc passes of `bench.py` into profiles/<round>/pmc_traffic.json (dev tool).

usage: make_pmc_json.py <fetch_dir> <write_dir> <hit_dir> <out.json>
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def load(d):
    tot = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            tot[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Kernel_Name"]][r["Counter_Name"]] += 1
    return tot, cnt


def group(name):
    if "k_dense" in name:
        return "k_dense"
    if "k_hash_tiles" in name or ("k_hash<" in name and "true>" in name):
        return "k_hash(window cells)"
    if "k_hash<" in name:
        return "k_hash(rows)"
    if "k_light" in name:
        return "k_light"
    return None


def main():
    fd, wd, hd, out = sys.argv[1:5]
    res = defaultdict(lambda: defaultdict(float))
    launches = {}
    for d, keys in ((fd, ["FETCH_SIZE"]), (wd, ["WRITE_SIZE"]), (hd, ["TCC_HIT_sum", "TCC_MISS_sum"])):
        tot, cnt = load(d)
        for k, vals in tot.items():
            g = group(k)
            if not g:
                continue
            for key in keys:
                if key in vals:
                    # per bench step: every kernel of the group is launched once per step
                    res[g][key] += vals[key] / cnt[k][key]
    j = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum (separate passes) -- python3 bench.py "
                   "--steps 2 --warmup 1 --no-cpu-baseline; raw csv beside this file",
         "correction": "bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE): FETCH_SIZE is in KB and on gfx950 reports half of a coalesced "
                       "stream's bytes (MI355X_MICROARCH.md, HBM section); the factor 2 is calibrated for 16 B/lane streams and is an "
                       "upper bound for the 12 B/lane segment gathers of these kernels",
         "kernels": {}}
    for g, v in res.items():
        hit = v.get("TCC_HIT_sum", 0.0)
        miss = v.get("TCC_MISS_sum", 0.0)
        j["kernels"][g] = {
            "fetch_size_kb_per_launch": v.get("FETCH_SIZE", 0.0),
            "write_size_kb_per_launch": v.get("WRITE_SIZE", 0.0),
            "l2_hit_rate": hit / (hit + miss) if hit + miss else None,
            "traffic_bytes_per_launch": int(1024 * (2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0))),
        }
    json.dump(j, open(out, "w"), indent=1)
    print(json.dumps(j["kernels"], indent=1))


if __name__ == "__main__":
    main()
