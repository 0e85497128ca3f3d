"""K independent chains on ONE GPU, each in its own host thread with its own handle, panel copy and persistent kernel on a share
of the CUs (ngp_set_max_shards): python tools/chains_per_gpu.py N P K iters [storage].  Prints the aggregate Gibbs iterations/s.
The chains are the path's own parallelism (independent chains, pooled once at the end); where one chain is
bound by the serial chain of its sampler workgroup and not by HBM, chains on disjoint CUs add up."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pkg import load_pkg
ngp = load_pkg()
N, P, K, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
storage = sys.argv[5] if len(sys.argv) > 5 else None
lag = int(os.environ.get("NGP_TOOL_LAG", "0"))
chains = []
shards = None
for k in range(K):
    s = ngp.Sampler(device=0, seed=1001 + k, chain=k, storage=storage, **({"mode": 1, "lag": lag} if lag else {}))
    if K > 1:
        shards = s.shards_for_chains(K)   # a chain's grid = 1 sampler + ceil(S / 32) reducers + S streamers; K grids co-resident
        s.set_max_shards(shards)
    s.generate_panel(N, P)
    rng = np.random.default_rng(1); bt = np.zeros(P); idx = rng.choice(P, max(10, P // 100), replace=False); bt[idx] = rng.normal(size=len(idx))
    g = s.xbeta(bt); y = 10 + g + np.random.default_rng(2).normal(size=N) * np.sqrt(g.var())
    v = 0.5 * y.var() / (s.mpm().sum() / N)
    s.add_marker_set(0, P, 0, 4.0, v * 0.5, [(0, P)], [v]); s.set_y(y); s.set_residual_prior(4.0, 0.25 * y.var())
    s.run(2)
    chains.append(s)
print(f"N={N} P={P} chains={K} storage={storage or 'f32'} layout of a chain {chains[0].layout()} lag {chains[0].config()[1]} streamer {chains[0].streamer()}", flush=True)
bar = threading.Barrier(K + 1)
def work(s):
    bar.wait()
    s.run(iters)
    bar.wait()
ths = [threading.Thread(target=work, args=(s,)) for s in chains]
for t in ths: t.start()
bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0
for t in ths: t.join()
print(f"   {K} x {iters} iterations in {dt:.3f} s: {K * iters / dt:.1f} it/s aggregate, {dt / iters * 1e3:.3f} ms per iteration of a chain")
for k, s in enumerate(chains):
    st = s.get_state()
    assert np.isfinite(st["beta"]).all() and st["varE"] > 0
    c = s.census()   # placement of the chain's last launch: workgroups per XCD, and per shader engine inside each XCD (HW_ID bits 15:13)
    per = [int((c["xcc"] == x).sum()) for x in range(8)]
    se = [[int(((c["xcc"] == x) & (((c["hw_id"] >> 13) & 7) == e)).sum()) for e in range(4)] for x in range(8)]
    print(f"   chain {k}: grid {c['grid']}, launches ended at the census and run again alone: {c['retries']}, whole-device lease now: {c['exclusive']}; "
          f"workgroups per XCD {per}; per shader engine {se}", flush=True)
