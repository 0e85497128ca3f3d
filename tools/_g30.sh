mkdir -p gpurun_out/r02ak
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compact.py -m gpu -x -q > gpurun_out/r02ak/t.txt 2>&1 || { tail -25 gpurun_out/r02ak/t.txt; exit 1; }
tail -2 gpurun_out/r02ak/t.txt
for rep in 1 2; do for k in 0 256; do
NGP_TOOL_KNOB=$k timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 30 1 2>&1 | grep -v invariant | sed "s/^/knob=$k /" | tee -a gpurun_out/r02ak/c.txt
done; done
for k in 0 256; do
NGP_TOOL_KNOB=$k NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 30 1 2>&1 | grep -v invariant | sed "s/^/knob=$k /" | tee -a gpurun_out/r02ak/c.txt
NGP_TOOL_KNOB=$k NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 30 1 2>&1 | grep -v invariant | sed "s/^/knob=$k /" | tee -a gpurun_out/r02ak/c.txt
done
