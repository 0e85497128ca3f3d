#!/bin/bash
# round 4: parity suite, ESS figures (C1 device vs oracle; C4 after a 1000-iteration burn-in), set-up timing, profiles of C4 / C2 / C4 u8
O=gpurun_out/r04r; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt; tail -4 $O/pytest.txt
timeout -k 10 600 python tests/run_config.py C1 1000 1000 > $O/ess_C1.json 2> $O/ess_C1.err; tail -c 900 $O/ess_C1.json
timeout -k 10 900 python bench.py --steps 1000 --warmup 1000 --no-compact --chains-per-pass 0 --no-cpu-baseline > $O/bench_C4_1000.json 2> $O/bench_C4_1000.err; echo "bench1000 rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04r/bench_C4_1000.json") if l.startswith("{")][0])
print("C4 1000+1000:", d["value"], d["roofline"]["frac"], d["effective_samples"]["ess"], d["effective_samples"]["ess_min_per_sec"], d["setup_s"], d["setup_parts_ms"])
PY
