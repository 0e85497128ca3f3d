mkdir -p gpurun_out/r02al
for rep in 1 2; do
timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 30 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02al/c.txt
timeout -k 10 120 python tools/shape_sweep.py 10000 100000 8 50 1 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02al/c.txt
done
NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 30 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02al/c.txt
NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 30 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02al/c.txt
timeout -k 10 200 python tools/shape_sweep.py 50000 600000 5 30 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02al/c.txt
timeout -k 10 200 python tools/shape_sweep.py 100000 100000 5 30 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02al/c.txt
