#!/bin/bash
# Runs on the GPU box (gpurun): kernel traces of every BASELINE config and the PMC passes of the headline bench,
# all into gpurun_out/$ROUND/ (scratch; ROUND defaults to r03).  scripts/make_profiles.py then writes the summaries -- run it on the
# box too (the traces are too large to travel): ROUND=r03 python3 scripts/make_profiles.py writes gpurun_out/$ROUND/summary/, which is copied to profiles/$ROUND/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
ROUND=${ROUND:-r03}
OUT=$R/gpurun_out/$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
trace() {   # name, bench args...
	local name=$1; shift
	rocprofv3 --kernel-trace --stats -d $OUT/trace_$name -o $name -- $B "$@" > $OUT/bench_$name.json 2> $OUT/err_$name.txt || return 1
	echo "trace $name done"
}
trace cfg2 --steps 5 --warmup 1 --no-other-configs --no-cpu-baseline &&
trace cfg2_coo --sink coo --steps 2 --warmup 1 --no-cpu-baseline &&
trace cfg3 --workload poisson --steps 5 --warmup 1 &&
trace cfg3_coo --workload poisson --sink coo --steps 5 --warmup 1 &&
trace cfg5 --workload galerkin --steps 5 --warmup 1 &&
trace cfg4 --scale 23 --steps 2 --warmup 1 --no-cpu-baseline &&
PYTHONPATH=$R rocprofv3 --kernel-trace --stats -d $OUT/trace_block -o block -- python3 $R/scripts/rehearse_dist.py 20 8 --trace=4 > $OUT/block.log 2> $OUT/err_block.txt &&
echo "trace block done" &&
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY"; do
	tag=$(echo $pass | cut -d' ' -f1)
	rocprofv3 --pmc $pass -d $OUT/pmc_$tag -o cfg2 --output-format csv -- $B --steps 2 --warmup 1 --no-other-configs --no-cpu-baseline > $OUT/bench_pmc_$tag.json 2> $OUT/err_pmc_$tag.txt || exit 1
	echo "pmc $tag done"
done
