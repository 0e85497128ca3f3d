"""Times the host-panel upload paths (ngp_set_panel_f64 / _f32 / _u8: staging + device-side centring and tiling + Gram window):
python tools/panel_upload_time.py N P"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pkg import load_pkg
ngp = load_pkg()
N, P = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(1)
G8 = np.asfortranarray(rng.integers(0, 3, size=(N, P), dtype=np.uint8))
for name, M in (("u8", G8), ("f32", np.asfortranarray(G8, dtype=np.float32)), ("f64", np.asfortranarray(G8, dtype=np.float64))):
    s = ngp.Sampler(device=0, seed=1, chain=0)
    t = time.perf_counter(); s.set_panel(M, centre=True); dt = time.perf_counter() - t
    if name != "u8":
        t = time.perf_counter(); s.begin_panel(N, P); s.panel_columns(0, M, centre=True); t1 = time.perf_counter() - t
        t = time.perf_counter(); s.end_panel(); t2 = time.perf_counter() - t
    else:
        t1 = t2 = float("nan")
    print(f"{name}: N={N} P={P} host {M.nbytes / 1e9:.2f} GB: set_panel {dt:.2f} s ({M.nbytes / 1e9 / dt:.1f} GB/s of host data); columns {t1:.2f} s + Gram {t2:.2f} s", flush=True)
    del s, M
