#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace run (its rocpd .db output): calls, total, average,
share -- the table the --stats CSV holds, written as CSV to stdout.

    python scripts/kernel_stats.py gpurun_out/prof_x/name_results.db > profiles/r02/name_kernel_stats.csv
"""
import re
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for name, s, e in rows:
        short = re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", ""))
        short = re.sub(r"^void ", "", short)
        a = agg.setdefault(short, [0, 0])
        a[0] += 1
        a[1] += e - s
    total = sum(v[1] for v in agg.values())
    print("Name,Calls,TotalDurationNs,AverageNs,Percentage")
    for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('"%s",%d,%d,%.1f,%.2f' % (name, n, t, t / n, 100.0 * t / total))


if __name__ == "__main__":
    main(sys.argv[1])
