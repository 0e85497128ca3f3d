mkdir -p gpurun_out/r02z
for lag in 4 5 6; do for k in 0 1; do
NGP_TOOL_KNOB=$k timeout -k 10 120 python tools/shape_sweep.py 10000 100000 $lag 30 1 2 2>&1 | grep -v invariant | sed "s/^/rows pace$k /" | tee -a gpurun_out/r02z/c3.txt
done; done
timeout -k 10 120 python tools/shape_sweep.py 10000 100000 8 30 1 1 2>&1 | grep -v invariant | sed "s/^/phase /" | tee -a gpurun_out/r02z/c3.txt
NGP_TOOL_KNOB=0 timeout -k 10 120 python tools/shape_sweep.py 20000 300000 6 20 1 2 2>&1 | grep -v invariant | sed "s/^/rows /" | tee -a gpurun_out/r02z/c3.txt
timeout -k 10 120 python tools/shape_sweep.py 20000 300000 8 20 1 1 2>&1 | grep -v invariant | sed "s/^/phase /" | tee -a gpurun_out/r02z/c3.txt
NGP_TOOL_KNOB=0 timeout -k 10 120 python tools/shape_sweep.py 30000 300000 6 20 1 2 2>&1 | grep -v invariant | sed "s/^/rows /" | tee -a gpurun_out/r02z/c3.txt
timeout -k 10 120 python tools/shape_sweep.py 30000 300000 8 20 1 1 2>&1 | grep -v invariant | sed "s/^/phase /" | tee -a gpurun_out/r02z/c3.txt
