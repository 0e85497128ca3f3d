# head vs current (k_sweep<false>) vs current (k_sweep_tup), interleaved, at 50k x 600k
for rep in 1 2 3 4; do
  echo -n "head: "; NGP_HIP_LIB=$PWD/build_ab/head.so timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 60 | grep -o "[0-9.]* ms/iter"
  echo -n "new k_sweep: "; timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 60 | grep -o "[0-9.]* ms/iter"
  echo -n "new k_sweep_tup: "; NGP_TOOL_KNOB=16384 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 60 | grep -o "[0-9.]* ms/iter"
done
