mkdir -p gpurun_out/r02ac
timeout -k 10 900 python -m pytest tests/test_gpu_compact.py -m gpu -x -q > gpurun_out/r02ac/t.txt 2>&1; rc=$?
tail -25 gpurun_out/r02ac/t.txt
exit $rc
