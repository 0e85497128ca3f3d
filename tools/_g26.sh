mkdir -p gpurun_out/r02af
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "share_a_device or allreduce" > gpurun_out/r02af/t.txt 2>&1; rc=$?
tail -15 gpurun_out/r02af/t.txt
exit $rc
