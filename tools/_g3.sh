mkdir -p gpurun_out/r02c
timeout -k 10 200 python tools/fine.py 5 50000 600000 2 > gpurun_out/r02c/c4_fine_rows.txt 2>&1
timeout -k 10 200 python tools/stamps.py 5 50000 600000 2 > gpurun_out/r02c/c4_stamps_rows.txt 2>&1
for m in 1 3 4; do NGP_TOOL_DEBUG_MODE=$m timeout -k 10 120 python tools/shape_sweep.py 50000 600000 5 10 1 2 > gpurun_out/r02c/c4_rows_mode$m.txt 2>&1; done
timeout -k 10 120 python tools/shape_sweep.py 50000 600000 6 10 1 2 > gpurun_out/r02c/c4_rows_lag6.txt 2>&1
cat gpurun_out/r02c/*.txt
