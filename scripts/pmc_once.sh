#!/bin/bash
# Developer tool, runs on the GPU box: one PMC pass (counters in "$1") over one bench.py invocation, per-kernel sums of the two big kernels printed.
#   bash scripts/pmc_once.sh "TCC_HIT_sum TCC_MISS_sum" <bench.py args...>
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
ctr=$1; shift
D=$R/gpurun_out/x/pmc_$$
mkdir -p $D
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctr -d $D -o p --output-format csv -- python3 $R/bench.py "$@" > $D/bench.json 2> $D/err.txt || exit 1
python3 $R/scripts/pmc_summary.py $D | grep -A2 "k_dense<\|k_bm_tiles<2" | grep -v "^--"
rm -rf $D
