"""Times K chains with a correlated (Tuple BayesPR) marker set over one copy of the panel, one fused launch per iteration
(k_sweep_multi_tup), beside one chain alone: python tools/tuple_chains_time.py N P k K [iters]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pkg import load_pkg
ngp = load_pkg()
N, P, k, K = (int(a) for a in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
form = int(os.environ.get("NGP_TOOL_CHAIN_FORM", "1"))
for nch in (1, K):
    chains = []
    for c in range(nch):
        s = ngp.Sampler(device=0, seed=1001 + c, chain=c)
        s.set_chain_form(form)
        if c == 0:
            if nch > 1: s.set_max_shards(s.shards_for_pass(nch))
            s.generate_panel(N, P)
            rng = np.random.default_rng(1); bt = np.zeros(P); idx = rng.choice(P, max(10, P // 100), replace=False); bt[idx] = rng.normal(size=len(idx))
            g = s.xbeta(bt); y = 10 + g + np.random.default_rng(2).normal(size=N) * np.sqrt(g.var())
            v = 0.5 * y.var() / (s.mpm().sum() / N)
        else:
            s.share_panel(chains[0])
        nloc = (P // 64) * (64 // k)
        V = v * (0.7 * np.eye(k) + 0.3)
        s.add_marker_set_tuple(0, nloc, k, 3.0 + k, V * 0.5, [(0, nloc)], V)
        s.set_y(y); s.set_residual_prior(4.0, 0.25 * y.var())
        chains.append(s)
    ngp.Sampler.run_many(chains, 3) if nch > 1 else chains[0].run(3)
    t = time.perf_counter()
    ngp.Sampler.run_many(chains, iters) if nch > 1 else chains[0].run(iters)
    dt = (time.perf_counter() - t) / iters
    R, S, nb = chains[0].layout()
    print(f"tuple k={k} chains={nch} N={N} P={P} layout R={R} S={S} lag={chains[0].config()[1]} grid={chains[0].census()['grid']}: "
          f"{dt * 1e3:.3f} ms/pass, {nch / dt:.1f} chain-iterations/s, {dt / nb * 1e6:.2f} us/block", flush=True)
    for s in chains[::-1]: s.close()
