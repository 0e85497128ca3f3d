mkdir -p gpurun_out/r02v
for lag in 5 6; do for nr in 3 4; do for k in 1 2 3; do
NGP_TOOL_KNOB=$k NGP_TOOL_NEAR=$nr timeout -k 10 120 python tools/shape_sweep.py 50000 600000 $lag 10 1 2 2>&1 | grep -v invariant | sed "s/^/near=$nr pace$k /" | tee -a gpurun_out/r02v/c.txt
done; done; done
NGP_TOOL_KNOB=2 NGP_TOOL_NEAR=3 timeout -k 10 200 python tools/stamps.py 6 50000 600000 2 2>&1 | head -24 > gpurun_out/r02v/c4_stamps6_near3.txt
NGP_TOOL_KNOB=2 NGP_TOOL_NEAR=3 timeout -k 10 200 python tools/fine.py 6 50000 600000 2 2>&1 > gpurun_out/r02v/c4_fine6_near3.txt
