# interleaved A/B at 50k x 600k: the previous library (build_ab/head.so), the current lean kernel, the current full kernel (knob bit 14)
for rep in $(seq 1 ${1:-5}); do
  echo -n "lean: "; timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 60 | grep -o "[0-9.]* ms/iter"
  echo -n "head: "; NGP_HIP_LIB=$PWD/build_ab/head.so timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 60 | grep -o "[0-9.]* ms/iter"
  echo -n "full: "; NGP_TOOL_KNOB=16384 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 60 | grep -o "[0-9.]* ms/iter"
done
