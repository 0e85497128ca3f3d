mkdir -p gpurun_out/r03x
timeout -k 10 300 python -m pytest tests/test_gpu_chains_per_pass.py -m gpu -x -q > gpurun_out/r03x/t.txt 2>&1; tail -3 gpurun_out/r03x/t.txt
for cfg in "8 0" "8 8192" "8 0" "8 8192" "4 0" "4 8192" "2 0" "2 8192"; do set -- $cfg; NGP_TOOL_KNOB=$2 timeout -k 10 120 python tools/chains_per_pass.py 10000 100000 $1 100 8 | sed "s/^/knob=$2 /" ; done > gpurun_out/r03x/streams.txt 2>&1
cut -c1-175 gpurun_out/r03x/streams.txt
