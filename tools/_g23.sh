mkdir -p gpurun_out/r02ae
timeout -k 10 1100 python -m pytest tests/test_gpu_fullsize.py tests/test_host_api.py -m gpu -x -q -k "compact or panel" > gpurun_out/r02ae/t.txt 2>&1; rc=$?
tail -25 gpurun_out/r02ae/t.txt
exit $rc
