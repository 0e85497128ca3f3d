"""Barrier-arrival timeline of one streamer (diagnostic): per wave, time of each stamp relative to the iteration start."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pkg import load_pkg
ngp = load_pkg()
lag = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
P = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
s = ngp.Sampler(device=0, seed=1001, chain=0, mode=1, lag=lag, streamer=int(sys.argv[4]) if len(sys.argv) > 4 else None,
                storage=os.environ.get("NGP_TOOL_STORAGE"))
if "NGP_TOOL_CHAIN_FORM" in os.environ: s.set_chain_form(int(os.environ["NGP_TOOL_CHAIN_FORM"]))
if "NGP_TOOL_KNOB" in os.environ: s.debug_set_knob(int(os.environ["NGP_TOOL_KNOB"]))
if "NGP_TOOL_NEAR" in os.environ: s.set_near(int(os.environ["NGP_TOOL_NEAR"]))
s.generate_panel(N, P)
rng = np.random.default_rng(1); bt = np.zeros(P); idx = rng.choice(P, P // 100, replace=False); bt[idx] = rng.normal(size=P // 100)
g = s.xbeta(bt); y = 10 + g + np.random.default_rng(2).normal(size=N) * np.sqrt(g.var())
v = 0.5 * y.var() / (s.mpm().sum() / N)
s.add_marker_set(0, P, 0, 4.0, v * 0.5, [(0, P)], [v]); s.set_y(y); s.set_residual_prior(4.0, 0.25 * y.var())
s.run(3)
dbgmode = int(os.environ.get("NGP_TOOL_DEBUG_MODE", "0"))  # timing modes of the diagnostic kernel (results invalid): 1 stream only, 3 never wait for dlt
if dbgmode: s.debug_set_mode(dbgmode)
s.debug_stamps(True)
try: s.run(1)
except ngp.NextGPHipError as e:
    if "diagnostic" not in str(e): raise
base = (7 << 17) + 8192
d = s.debug_stamps(True, n=base + 1024).astype(np.int64)
F = d[base:base + 1024].reshape(16, 8, 8)
t0 = F[:, :, 0].min(axis=1)  # earliest wave start of each iteration
names = ["start", "A done", "B1 (pp written)", "B2 (ys updated)", "C gemv done", "C poll done", "iter end", "A: tile drained (w4-6)"]
if s.streamer()[0] >= 2:  # row-owning waves (waves 0-6) + loader (wave 7: start, requests issued, counted wait done)
    names = ["start", "update done | w7 issued", "gemv done | w7 landed", "keep filled", "dlt fetched", "past barrier", "published", "-"]
print("stamp (us after the first wave's start of the iteration), median over 16 iterations; rows = waves 0..7")
for k in range(8):
    rel = (F[:, :, k] - t0[:, None]) / 100.0
    print(f"{names[k]:24s}", " ".join(f"{x:6.2f}" if abs(x) < 1e6 else "     -" for x in np.median(rel, axis=0)))
print("iteration period us:", np.median(np.diff(t0)) / 100.0)
