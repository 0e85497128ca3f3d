"""Register / spill / scratch / LDS figures of every kernel in the built library, from the code-object notes:
python tools/resource_usage.py [lib.so] > profiles/rNN_resource_usage.txt"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "nextgp.jl_amd", "libnextgp_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
rows = []
with tempfile.TemporaryDirectory() as td:
    # every translation unit left one gfx950 code object in the fat binary section
    import shutil
    lib = shutil.copy(lib, os.path.join(td, "lib.so"))  # (the bundles are written next to the file)
    out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", lib], cwd=td, capture_output=True, text=True)
    cos = [f for f in os.listdir(td) if "gfx950" in f]
    if not cos:
        raise SystemExit("no code objects extracted: " + out.stderr[:400])
    for co in sorted(cos):
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(td, co)], capture_output=True, text=True).stdout
        for blk in txt.split("- .agpr_count:")[1:]:
            def g(key, default="?"):
                m = re.search(r"\." + key + r":\s+(\S+)", blk)
                return m.group(1) if m else default
            name = g("name")
            try:
                name = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt"), name], capture_output=True, text=True).stdout.strip().split("(")[0]
            except Exception:
                pass
            rows.append((name, g("vgpr_count"), blk.split()[0], g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"),
                         g("private_segment_fixed_size"), g("group_segment_fixed_size")))
print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'Vspill':>6s} {'SGPR':>5s} {'Sspill':>6s} {'scratch':>7s} {'LDS':>6s}")
for r in sorted(set(rows)):
    print(f"{r[0][:58]:58s} {r[1]:>5s} {r[2]:>5s} {r[3]:>6s} {r[4]:>5s} {r[5]:>6s} {r[6]:>7s} {r[7]:>6s}")
