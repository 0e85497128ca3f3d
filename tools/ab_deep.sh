mkdir -p gpurun_out/r03x
for cfg in "50000 600000 6" "20000 100000 6" "30000 200000 6"; do for k in 0 32 0 32; do NGP_TOOL_KNOB=$k timeout -k 10 200 python tools/shape_sweep.py $cfg 40 | sed "s/^/knob=$k /"; done; done > gpurun_out/r03x/ab_deep.txt 2>&1
grep -o "^knob=[0-9]* N=[0-9]* P=[0-9]*\|[0-9.]* ms/iter\|varE [0-9.]*" gpurun_out/r03x/ab_deep.txt | paste - - -
