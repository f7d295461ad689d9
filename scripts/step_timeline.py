#!/usr/bin/env python3
"""Timeline of the LAST step in a rocprofv3 --kernel-trace run (its rocpd .db): every kernel between the last two
`k_digest_reduce` launches with its start offset, duration, stream and the gap since the previous kernel ended -- where a
step's time goes between kernels (host synchronisations, launch latency) and which kernels overlap on the side streams.

    python scripts/step_timeline.py gpurun_out/x/name_results.db [marker-kernel]
"""
import re
import sqlite3
import sys


def main(path, marker="k_digest_reduce"):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    stream_col = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
    q = "select %s, start, end%s from kernels order by start" % (name_col, (", " + stream_col) if stream_col else "")
    rows = cur.execute(q).fetchall()
    ends = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(ends) < 2:
        raise SystemExit("fewer than two %s launches" % marker)
    seg = rows[ends[-2] + 1:ends[-1] + 1]
    t0 = seg[0][1]
    last_end = t0
    busy = 0
    print("%9s %9s %8s %4s  %s" % ("start_us", "dur_us", "gap_us", "strm", "kernel"))
    for r in seg:
        name = re.sub(r"\(.*$", "", r[0].replace("(anonymous namespace)::", "")).replace("void ", "").replace("spsamd::", "")
        gap = (r[1] - last_end) / 1e3
        print("%9.1f %9.1f %8.1f %4s  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, gap, r[3] if stream_col else "-", name[:90]))
        if r[2] > last_end:
            busy += r[2] - max(r[1], last_end)
            last_end = r[2]
    span = (seg[-1][2] - t0) / 1e3
    print("# span %.1f us, device busy %.1f us, idle %.1f us, %d launches" % (span, busy / 1e3, span - busy / 1e3, len(seg)))


if __name__ == "__main__":
    main(*sys.argv[1:3])
