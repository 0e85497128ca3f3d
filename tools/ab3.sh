# interleaved A/B of library builds at 50k x 600k: tools/ab3.sh "lib1 lib2 ..." [reps]
libs=${1:-"build_ab/head.so nextgp.jl_amd/libnextgp_hip.so"}
for rep in $(seq 1 ${2:-4}); do
  for l in $libs; do echo -n "$l: "; NGP_HIP_LIB=$PWD/$l timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 60 | grep -o "[0-9.]* ms/iter"; done
done
