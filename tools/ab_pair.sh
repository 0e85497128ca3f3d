# chains per pass at 10k x 100k, K = 2..8 (one line each) -> gpurun_out/r03x/pass_table.txt
mkdir -p gpurun_out/r03x
for K in 2 3 4 5 6 7 8; do timeout -k 10 120 python tools/chains_per_pass.py 10000 100000 $K 100 8; done > gpurun_out/r03x/pass_table.txt 2>&1
NGP_TOOL_METHOD=B timeout -k 10 120 python tools/chains_per_pass.py 10000 100000 8 100 8 >> gpurun_out/r03x/pass_table.txt 2>&1
cut -c1-220 gpurun_out/r03x/pass_table.txt
