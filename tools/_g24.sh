mkdir -p gpurun_out/r02ae
timeout -k 10 600 python -m pytest tests/test_bench_contract.py -m gpu -x -q > gpurun_out/r02ae/tb.txt 2>&1 || { tail -25 gpurun_out/r02ae/tb.txt; exit 1; }
tail -2 gpurun_out/r02ae/tb.txt
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02ae/bench_C4.json 2> gpurun_out/r02ae/bench_C4.err || { tail -20 gpurun_out/r02ae/bench_C4.err; exit 1; }
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r02ae/bench_C4.json") if l.startswith("{")][0])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["compact_storage"], d["cpu_baseline"]["value"])
PY
