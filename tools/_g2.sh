set -e
mkdir -p gpurun_out/r02b
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rows or shard_height or draws or math" > gpurun_out/r02b/pytest_rows.log 2>&1 || { tail -30 gpurun_out/r02b/pytest_rows.log; exit 1; }
tail -3 gpurun_out/r02b/pytest_rows.log
timeout -k 10 200 python tools/shape_sweep.py 50000 600000 5 20 > gpurun_out/r02b/c4_rows.txt 2>&1 || true
cat gpurun_out/r02b/c4_rows.txt
