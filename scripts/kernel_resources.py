#!/usr/bin/env python3
"""Register / LDS / occupancy figures of every kernel of one csrc file, from hipcc's kernel-resource-usage remarks.

    python scripts/kernel_resources.py k_dense.hip [extra hipcc flags]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spsparse_amd import build as b  # noqa: E402


def main():
    src = os.path.join(b.CSRC, sys.argv[1])
    cmd = [b.hipcc()] + b.FLAGS + sys.argv[2:] + ["-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = {}
    rows = []
    for ln in err.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?)\s*\[-Rpass", ln) or re.search(r"remark:\s+(.*?)\s*\[-Rpass", ln)
        if not m:
            continue
        t = m.group(1)
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void spsamd::", "")
        print("%-60s vgpr %4s agpr %3s sgpr %4s lds %7s scratch %4s occ %s" % (name[:60], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"),
              r.get("LDS Size [bytes/block]"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]")))


if __name__ == "__main__":
    main()
