"""Set-up time of a generated panel with the Gram window by the fp64 VALU kernel (default) and on the matrix cores (knob bit 10):
python tools/gram_time.py N P"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pkg import load_pkg
ngp = load_pkg()
N, P = int(sys.argv[1]), int(sys.argv[2])
ref = None
for name, knob in (("valu", 0), ("mfma", 1024), ("valu", 0), ("mfma", 1024)):
    s = ngp.Sampler(device=0, seed=1, chain=0)
    s.debug_set_knob(knob)
    t0 = time.perf_counter(); s.generate_panel(N, P); dt = time.perf_counter() - t0
    g = s.gram(min(7, s.layout()[2] - 1)); m = s.mpm()[:4096].copy()
    if ref is None: ref = (g, m)
    print(f"N={N} P={P} layout {s.layout()} lag {s.config()[1]} gram engine {name}: generate_panel (tiles + Gram window) {dt:.3f} s; identical to the first: {np.array_equal(g, ref[0]) and np.array_equal(m, ref[1])}", flush=True)
    s.close()
