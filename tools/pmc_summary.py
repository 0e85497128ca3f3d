"""Folds the rocprofv3 outputs of tools/profile_round.sh into the summaries kept under profiles/.

FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B per the guide's HBM section; on gfx950 FETCH_SIZE counts
64 B per 128-B request, hence the factor 2 (MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
cfg = sys.argv[2] if len(sys.argv) > 2 else "C4"
base_cfg = sys.argv[3] if len(sys.argv) > 3 else cfg
storage = sys.argv[4] if len(sys.argv) > 4 else "f32"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
# the persistent sweep kernel in its instantiations (k_sweep_tup: Tuple sets and tall fp32 shards of the row-owning streamer;
# k_sweep_tall: fp32 panels with several shards per streamer workgroup)
KERNELS = ("ngp::k_sweep(ngp::SweepArgs)", "void ngp::k_sweep<false>(ngp::SweepArgs)", "void ngp::k_sweep<true>(ngp::SweepArgs)",
           "ngp::k_sweep_tup(ngp::SweepArgs)", "ngp::k_sweep_tall(ngp::SweepArgs)", "ngp::k_sweep_r(ngp::SweepArgs)")


def counter(kind, name):
    files = glob.glob(os.path.join(out, f"{tag}_{cfg}_{kind}", "**", "*counter_collection.csv"), recursive=True)
    vals = []
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Kernel_Name"] in KERNELS and row["Counter_Name"] == name:
                vals.append(float(row["Counter_Value"]))
    return vals


fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
found = sorted({row["Kernel_Name"] for f in glob.glob(os.path.join(out, f"{tag}_{cfg}_fetch", "**", "*counter_collection.csv"), recursive=True)
                for row in csv.DictReader(open(f)) if row["Kernel_Name"] in KERNELS})
bench = json.loads(open(os.path.join(out, f"{tag}_{cfg}_bench.json")).read().strip().splitlines()[-1])
summary = {
    "command": f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --config {base_cfg} --storage {storage} --steps 10 --warmup 2 --no-cpu-baseline --no-compact --chains-per-pass 0 (two separate passes)",
    "workload": bench["config"]["workload"],
    "kernel": ", ".join(found) if found else "ngp::k_sweep",
    "FETCH_SIZE_KB_per_launch_raw": sum(fetch) / max(len(fetch), 1),
    "FETCH_SIZE_launches": len(fetch),
    "WRITE_SIZE_KB_per_launch_raw": sum(write) / max(len(write), 1),
    "WRITE_SIZE_launches": len(write),
}
summary["fetch_bytes_corrected"] = summary["FETCH_SIZE_KB_per_launch_raw"] * 1024.0 * 2.0
summary["write_bytes"] = summary["WRITE_SIZE_KB_per_launch_raw"] * 1024.0
summary["hbm_bytes_per_launch"] = summary["fetch_bytes_corrected"] + summary["write_bytes"]
summary["algorithmic_bytes_per_launch"] = bench["roofline"].get("algorithmic_bytes_per_launch")
os.makedirs(os.path.join(out, "profiles"), exist_ok=True)
json.dump(summary, open(os.path.join(out, "profiles", f"{tag}_pmc_k_sweep_{cfg}.json"), "w"), indent=1)
stats = glob.glob(os.path.join(out, f"{tag}_{cfg}_stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(out, "profiles", f"{tag}_kernel_stats_{cfg}.csv"))
shutil.copy(os.path.join(out, f"{tag}_{cfg}_bench.json"), os.path.join(out, "profiles", f"{tag}_bench_{cfg}.json"))
print(json.dumps(summary, indent=1))
