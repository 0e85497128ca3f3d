#!/bin/bash
# Runs on the GPU box (gpurun): kernel statistics and HBM counters of a bench.py workload.
#   tools/profile_round.sh TAG [CONFIG]  -> gpurun_out/TAG_CONFIG_{stats,fetch,write}/..., gpurun_out/profiles/TAG_*_CONFIG.*
# The three rocprofv3 passes are separate on purpose (counters are never combined with traces).
set -e
TAG=${1:-r02}
CFG=${2:-C4}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
python3 bench.py --config $CFG > $OUT/${TAG}_${CFG}_bench.json 2> $OUT/${TAG}_${CFG}_bench.err
ARGS="bench.py --config $CFG --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_${CFG}_stats -o run -- python3 $ARGS > $OUT/${TAG}_${CFG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_${CFG}_fetch -o run -- python3 $ARGS > $OUT/${TAG}_${CFG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_${CFG}_write -o run -- python3 $ARGS > $OUT/${TAG}_${CFG}_write.log 2>&1
python3 tools/pmc_summary.py $TAG $CFG
