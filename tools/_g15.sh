mkdir -p gpurun_out/r02ab
for m in 1 2 3 4 6; do
NGP_TOOL_DEBUG_MODE=$m timeout -k 10 120 python tools/shape_sweep.py 10000 100000 8 50 1 2 2>&1 | grep -v invariant | tee -a gpurun_out/r02ab/c2m.txt
NGP_TOOL_DEBUG_MODE=$m timeout -k 10 120 python tools/shape_sweep.py 10000 100000 8 50 1 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02ab/c2m.txt
done
