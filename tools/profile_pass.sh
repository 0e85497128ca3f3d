#!/bin/bash
# Runs on the GPU box (gpurun): the fused launch of K chains per pass (k_sweep_multi) at 10k x 100k -- kernel statistics and fetched bytes.
#   tools/profile_pass.sh TAG K  -> gpurun_out/profiles/TAG_pass_K{K}_{stats.csv,fetch.json,run.txt}
# One panel read must serve K iterations' worth of sampling: FETCH_SIZE per launch stays about 4 N P = 4 GB whatever K.
set -e
TAG=${1:-r03}
K=${2:-6}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT/profiles
python3 tools/chains_per_pass.py 10000 100000 $K 200 > $OUT/profiles/${TAG}_pass_K${K}_run.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_pass${K}_stats -o run -- python3 tools/chains_per_pass.py 10000 100000 $K 30 > $OUT/${TAG}_pass${K}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pass${K}_fetch -o run -- python3 tools/chains_per_pass.py 10000 100000 $K 30 > $OUT/${TAG}_pass${K}_fetch.log 2>&1
python3 - <<PY
import csv, glob, json, os, shutil
out = "$OUT"; tag = "$TAG"; K = "$K"
st = glob.glob(os.path.join(out, f"{tag}_pass{K}_stats", "**", "*kernel_stats.csv"), recursive=True)
if st: shutil.copy(st[0], os.path.join(out, "profiles", f"{tag}_pass_K{K}_kernel_stats.csv"))
vals = []
for f in glob.glob(os.path.join(out, f"{tag}_pass{K}_fetch", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_sweep_multi" in row["Kernel_Name"] and row["Counter_Name"] == "FETCH_SIZE": vals.append(float(row["Counter_Value"]))
s = {"kernel": "ngp::k_sweep_multi", "chains_per_pass": int(K), "launches": len(vals), "FETCH_SIZE_KB_per_launch_raw": sum(vals) / max(len(vals), 1)}
s["fetch_bytes_corrected"] = s["FETCH_SIZE_KB_per_launch_raw"] * 1024.0 * 2.0   # gfx950: FETCH_SIZE counts 64 B per 128-B request
s["algorithmic_bytes_per_launch"] = 4.0 * 10000 * 100000
json.dump(s, open(os.path.join(out, "profiles", f"{tag}_pass_K{K}_fetch.json"), "w"), indent=1)
print(json.dumps(s))
PY
