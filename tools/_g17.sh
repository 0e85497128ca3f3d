mkdir -p gpurun_out/r02ab
for rep in 1 2; do for k in 0 64 128 192; do
NGP_TOOL_KNOB=$k timeout -k 10 120 python tools/shape_sweep.py 10000 100000 8 50 1 1 2>&1 | grep -v invariant | sed "s/^/knob=$k /" | tee -a gpurun_out/r02ab/c2s.txt
NGP_TOOL_DEBUG_MODE=2 NGP_TOOL_KNOB=$k timeout -k 10 120 python tools/shape_sweep.py 10000 100000 8 50 1 1 2>&1 | grep -v invariant | sed "s/^/knob=$k /" | tee -a gpurun_out/r02ab/c2s.txt
done; done
