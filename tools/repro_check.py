"""Runs the same chain twice and compares the states bit for bit (hand-off races show up as differences): N P lag iters [storage]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pkg import load_pkg
ngp = load_pkg()
N, P, lag, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
storage = sys.argv[5] if len(sys.argv) > 5 else None
outs = []
for rep in range(int(os.environ.get("NGP_TOOL_REPS", "3"))):
    s = ngp.Sampler(device=0, seed=1001, chain=0, mode=1, lag=lag, storage=storage)
    s.generate_panel(N, P)
    rng = np.random.default_rng(1); bt = np.zeros(P); idx = rng.choice(P, max(10, P // 100), replace=False); bt[idx] = rng.normal(size=len(idx))
    g = s.xbeta(bt); y = 10 + g + np.random.default_rng(2).normal(size=N) * np.sqrt(g.var())
    v = 0.5 * y.var() / (s.mpm().sum() / N)
    nsets = int(os.environ.get("NGP_TOOL_SETS", "1"))
    for c in range(nsets):
        w = P // nsets
        s.add_marker_set(c * w, w if c < nsets - 1 else P - c * w, 0, 4.0, v * 0.5, [(0, w if c < nsets - 1 else P - c * w)], [v])
    s.set_y(y); s.set_residual_prior(4.0, 0.25 * y.var())
    first = None
    chunk = int(os.environ.get("NGP_TOOL_CHUNK", "1"))
    for it in range(iters):
        s.run(chunk)
        st = s.get_state()
        if outs and first is None and not np.array_equal(st["beta"], outs[0][it]):
            first = it
            bad = np.nonzero(st["beta"] != outs[0][it])[0]
            print(f"rep {rep}: first difference at iteration {it}, first column {bad[0]} (block {bad[0] // 64}), {len(bad)} columns differ", flush=True)
        if not outs or True:
            pass
        if rep == 0:
            if it == 0: hist = []
            hist.append(st["beta"].copy())
    if rep == 0: outs.append(hist)
    print(f"rep {rep}: layout {s.layout()} lag {s.config()[1]} streamer {s.streamer()} {'identical' if first is None else 'DIFFERENT'}", flush=True)
    s.close()
