#!/bin/bash
# Runs on the GPU box: kernel statistics and HBM counters of a 10k x 100k model with a Tuple set (k = 2, inverse form) and with a
# BayesR set (four classes) -- the two methods that had no rocprof evidence.  tools/profile_methods.sh TAG
# (three separate rocprofv3 passes per workload: counters are never combined with traces)
set -e
TAG=${1:-r04}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT/profiles
# TK: eight chains with a Tuple set in one fused launch per iteration (k_sweep_multi_tup); UK: two chains over compact storage
# (k_sweep_multi, byte tiles); RK: eight chains with a BayesR set (k_sweep_multi_r) -- kernel statistics only
for W in TK UK RK; do
  unset NGP_TOOL_CHAIN_FORM NGP_TOOL_STORAGE NGP_TOOL_METHOD
  if [ "$W" = "TK" ]; then CMD="tools/tuple_chains_time.py 10000 100000 2 8 10"; elif [ "$W" = "RK" ]; then export NGP_TOOL_METHOD=R; CMD="tools/chains_per_pass.py 10000 100000 8 20"; else export NGP_TOOL_STORAGE=u8; CMD="tools/chains_per_pass.py 10000 100000 2 30"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_${W}_stats -o run -- python3 $CMD > $OUT/${TAG}_${W}_stats.log 2>&1
  cp "$(find $OUT/${TAG}_${W}_stats -name '*kernel_stats.csv' | head -1)" $OUT/profiles/${TAG}_kernel_stats_${W}.csv
  tail -3 $OUT/${TAG}_${W}_stats.log
done
unset NGP_TOOL_STORAGE NGP_TOOL_METHOD
for W in T R; do
  if [ "$W" = "T" ]; then export NGP_TOOL_CHAIN_FORM=1; CMD="tools/tuple_time.py 10000 100000 2 10"; else unset NGP_TOOL_CHAIN_FORM; export NGP_TOOL_METHODS=R4; CMD="tools/method_time.py 10000 100000 10"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_${W}_stats -o run -- python3 $CMD > $OUT/${TAG}_${W}_stats.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_${W}_fetch -o run -- python3 $CMD > $OUT/${TAG}_${W}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_${W}_write -o run -- python3 $CMD > $OUT/${TAG}_${W}_write.log 2>&1
  python3 - "$TAG" "$W" <<'PY'
import csv, glob, json, os, shutil, sys
tag, w = sys.argv[1], sys.argv[2]
out = os.path.join(os.getcwd(), "gpurun_out")
def counter(kind, name):
    vals = {}
    for f in glob.glob(os.path.join(out, f"{tag}_{w}_{kind}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_sweep" in row["Kernel_Name"] and row["Counter_Name"] == name:
                vals.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return vals
fe, wr = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
summ = {"workload": {"T": "10k x 100k, one Tuple set of k = 2 (inverse form), tools/tuple_time.py", "R": "10k x 100k, one BayesR set of four classes, tools/method_time.py"}[w],
        "algorithmic_bytes_per_launch": 4.0 * 10000 * 100000, "kernels": {}}
for k in fe:
    f = sum(fe[k]) / len(fe[k]) * 1024.0 * 2.0
    wv = sum(wr.get(k, [0.0])) / max(len(wr.get(k, [0.0])), 1) * 1024.0
    summ["kernels"][k] = {"launches": len(fe[k]), "fetch_bytes_corrected": f, "write_bytes": wv, "hbm_bytes_per_launch": f + wv}
json.dump(summ, open(os.path.join(out, "profiles", f"{tag}_pmc_k_sweep_{w}.json"), "w"), indent=1)
st = glob.glob(os.path.join(out, f"{tag}_{w}_stats", "**", "*kernel_stats.csv"), recursive=True)
if st: shutil.copy(st[0], os.path.join(out, "profiles", f"{tag}_kernel_stats_{w}.csv"))
print(json.dumps(summ, indent=1))
PY
done
