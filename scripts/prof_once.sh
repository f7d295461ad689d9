#!/bin/bash
# Developer tool, runs on the GPU box: kernel-trace statistics of one bench.py invocation as gpurun_out/x/<name>_kernel_stats.csv.
#   bash scripts/prof_once.sh <name> <bench.py args...>
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
name=$1; shift
mkdir -p $R/gpurun_out/x
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/x/tr_$name -o $name -- python3 $R/bench.py "$@" > $R/gpurun_out/x/bench_$name.json 2> $R/gpurun_out/x/err_$name.txt || exit 1
find $R/gpurun_out/x/tr_$name -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/x/${name}_kernel_stats.csv \;
rm -rf $R/gpurun_out/x/tr_$name
python3 - $R/gpurun_out/x/${name}_kernel_stats.csv <<'PY'
import csv, re, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    n = re.sub(r'\(.*', '', r['Name']).replace('void spsamd::', '').replace('spsamd::', '')
    print('%-45s calls %3s avg %9.3f ms  %5s%%' % (n, r['Calls'], float(r['AverageNs']) / 1e6, r['Percentage']))
PY
