"""A/B builds: python tools/build_variant.py NAME [-DFOO=1 ...] -> build_ab/NAME.so (the library with extra defines; select it at run
time with NGP_HIP_LIB=$PWD/build_ab/NAME.so, e.g. through tools/ab.sh)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
name, defs = sys.argv[1], sys.argv[2:]
obj = os.path.join(ROOT, "build_ab", "obj_" + name)
os.makedirs(obj, exist_ok=True)
procs = []
for o, src, d, _ in g.UNITS:
    cmd = [g.HIPCC] + g.HIP_FLAGS + d + defs + ["-c", "-o", os.path.join(obj, o), os.path.join(g.CSRC, src)]
    procs.append((cmd, subprocess.Popen(cmd, cwd=ROOT)))
for cmd, p in procs:
    if p.wait() != 0:
        raise SystemExit("failed: " + " ".join(cmd))
out = os.path.join(ROOT, "build_ab", name + ".so")
subprocess.check_call([g.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + [os.path.join(obj, u[0]) for u in g.UNITS] + ["-ldl", "-lpthread"], cwd=ROOT)
print(out)
