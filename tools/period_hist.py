"""Distribution of the sampler's per-block period over one sweep (diagnostic): python tools/period_hist.py lag N P [streamer]"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pkg import load_pkg
ngp = load_pkg()
lag, N, P = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
s = ngp.Sampler(device=0, seed=1001, chain=0, mode=1, lag=lag, streamer=int(sys.argv[4]) if len(sys.argv) > 4 else None)
if "NGP_TOOL_KNOB" in os.environ: s.debug_set_knob(int(os.environ["NGP_TOOL_KNOB"]))
if "NGP_TOOL_NEAR" in os.environ: s.set_near(int(os.environ["NGP_TOOL_NEAR"]))
s.generate_panel(N, P)
rng = np.random.default_rng(1); bt = np.zeros(P); idx = rng.choice(P, P // 100, replace=False); bt[idx] = rng.normal(size=P // 100)
g = s.xbeta(bt); y = 10 + g + np.random.default_rng(2).normal(size=N) * np.sqrt(g.var())
v = 0.5 * y.var() / (s.mpm().sum() / N)
s.add_marker_set(0, P, 0, 4.0, v * 0.5, [(0, P)], [v]); s.set_y(y); s.set_residual_prior(4.0, 0.25 * y.var())
s.run(3)
s.debug_stamps(True); s.run(1)
nb = s.layout()[2]
d = s.debug_stamps(True, n=4 * nb + 16).astype(np.int64)
S = d[:4 * nb].reshape(nb, 4)
per = np.diff(S[:, 0]) / 100.0
print("streamer", s.streamer(), "lag", s.config()[1], "blocks", nb, "sweep us", (S[-1, 0] - S[0, 0]) / 100.0)
print("period us: mean %.3f median %.3f p10 %.3f p90 %.3f p99 %.3f max %.3f" % (per.mean(), np.median(per), *np.percentile(per, [10, 90, 99]), per.max()))
big = np.where(per > 2 * np.median(per))[0]
print("blocks with period > 2 x median:", len(big), "total excess us", (per[big] - np.median(per)).sum(), "first", big[:20])
k = 256
seg = per[: (len(per) // k) * k].reshape(-1, k).mean(axis=1)
print("mean period per %d-block segment:" % k, np.round(seg, 2))
