#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r04y; mkdir -p $O
export NGP_TOOL_METHODS=R4
rocprofv3 --kernel-trace --stats --output-format csv -d $O/R_stats -o run -- python3 tools/method_time.py 10000 100000 10 > $O/R_stats.log 2>&1
head -8 $(find $O/R_stats -name '*kernel_stats.csv' | head -1) | cut -c1-160
