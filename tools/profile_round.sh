#!/bin/bash
# Runs on the GPU box (gpurun): kernel statistics and HBM counters of the default bench.py workload.
#   tools/profile_round.sh TAG     -> gpurun_out/TAG_{stats,fetch,write}/..., gpurun_out/TAG_bench.json
# The three rocprofv3 passes are separate on purpose (counters are never combined with traces).
set -e
TAG=${1:-r01}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
ARGS="bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o run -- python3 $ARGS > $OUT/${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -o run -- python3 $ARGS > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -o run -- python3 $ARGS > $OUT/${TAG}_write.log 2>&1
python3 tools/pmc_summary.py $TAG
