#!/bin/bash
# Runs on the GPU box (gpurun): kernel statistics and HBM counters of a bench.py workload.
#   tools/profile_round.sh TAG [CONFIG] [u8]  -> gpurun_out/TAG_CONFIG_{stats,fetch,write}/..., gpurun_out/profiles/TAG_*_CONFIG.*
# (third argument u8: the same workload in compact storage, summaries named CONFIGu8)
# The three rocprofv3 passes are separate on purpose (counters are never combined with traces).
set -e
TAG=${1:-r04}
CFG=${2:-C4}
STO=${3:-f32}
SFX=""
if [ "$STO" = "u8" ]; then SFX="u8"; fi
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
python3 bench.py --config $CFG --storage $STO > $OUT/${TAG}_${CFG}${SFX}_bench.json 2> $OUT/${TAG}_${CFG}${SFX}_bench.err
ARGS="bench.py --config $CFG --storage $STO --steps 10 --warmup 2 --no-cpu-baseline --no-compact --chains-per-pass 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_${CFG}${SFX}_stats -o run -- python3 $ARGS > $OUT/${TAG}_${CFG}${SFX}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_${CFG}${SFX}_fetch -o run -- python3 $ARGS > $OUT/${TAG}_${CFG}${SFX}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_${CFG}${SFX}_write -o run -- python3 $ARGS > $OUT/${TAG}_${CFG}${SFX}_write.log 2>&1
python3 tools/pmc_summary.py $TAG ${CFG}${SFX} $CFG $STO
