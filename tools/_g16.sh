mkdir -p gpurun_out/r02ab
for k in 0 256 512 768 1024 1280 1536; do
NGP_TOOL_KNOB=$k timeout -k 10 120 python tools/shape_sweep.py 10000 100000 8 50 1 2 2>&1 | grep -v invariant | sed "s/^/knob=$k /" | tee -a gpurun_out/r02ab/c2k.txt
done
NGP_TOOL_KNOB=1280 timeout -k 10 120 python tools/shape_sweep.py 10000 100000 12 50 1 2 2>&1 | grep -v invariant | sed "s/^/knob=1280 /" | tee -a gpurun_out/r02ab/c2k.txt
timeout -k 10 120 python tools/shape_sweep.py 10000 100000 8 50 1 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02ab/c2k.txt
for k in 0 256 1024 1280 512; do
NGP_TOOL_KNOB=$k timeout -k 10 120 python tools/shape_sweep.py 50000 600000 6 30 1 2 2>&1 | grep -v invariant | sed "s/^/knob=$k /" | tee -a gpurun_out/r02ab/c4k.txt
done
