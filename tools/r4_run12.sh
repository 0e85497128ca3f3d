#!/bin/bash
O=gpurun_out/r04m; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_bench_contract.py tests/test_host_api.py tests/test_gpu_chains_per_pass.py -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt; tail -8 $O/pytest.txt
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04m/bench_default.json") if l.startswith("{")][0])
print("value", d["value"], "frac", d["roofline"]["frac"], "setup", d["setup_s"], "retries", d["census_retries"])
print("compact", d["compact_storage"]["value"], d["compact_storage"]["roofline"]["frac"], d["compact_storage"]["layout"])
print("pass", {k: d["chains_per_pass"][k] for k in ("chains","value","lag","fused","census_retries","panel_stream_frac_of_peak")})
print("cpu", d["cpu_baseline"]["value"], d["speedup_vs_cpu_baseline"])
PY
for k in 0 1 2; do echo "== knob pace $k lag6"; NGP_TOOL_KNOB=$((32768+k)) NGP_TOOL_CHAIN_FORM=0 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant; done
echo "== lag5"; NGP_TOOL_KNOB=32768 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 5 40 | grep -v invariant
