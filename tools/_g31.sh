mkdir -p gpurun_out/r02ak
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r02ak/all.txt 2>&1 || { tail -25 gpurun_out/r02ak/all.txt; exit 1; }
tail -2 gpurun_out/r02ak/all.txt
timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 30 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02ak/c2.txt
NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 30 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02ak/c2.txt
for N in 30000 40000 50000; do
timeout -k 10 200 python tools/shape_sweep.py $N 200000 6 30 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02ak/c2.txt
done
NGP_TOOL_KNOB=256 timeout -k 10 200 python tools/shape_sweep.py 40000 200000 6 30 1 2>&1 | grep -v invariant | sed "s/^/late /" | tee -a gpurun_out/r02ak/c2.txt
