#!/usr/bin/env python3
"""bench.py -- Gibbs iterations/s of the marker-effect sampler on a synthetic N x P SNP panel.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one full Gibbs iteration (varE draw, intercept draw, sweep of all P SNPs with residual
update, variance draws, posterior accumulation) of BASELINE.json configs[1]: BayesPR, single trait,
10,000 individuals x 100,000 SNPs, fp32 panel resident in HBM.  With N > 1 every rank runs its own
independent chain on its own GPU (seeds 1001+rank, weak scaling, no data-path collective); the
posterior sums are all-reduced once over RCCL after the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_chain(ngp, device, seed, N, P, method, panel_seed=20250509):
    s = ngp.Sampler(device=device, seed=seed, chain=seed - 1001)
    t0 = time.time()
    s.generate_panel(N, P, 0.05, 0.5, panel_seed)
    setup_s = time.time() - t0
    # phenotype: y = 10 + X beta + e, 1 % causal SNPs ~ N(0,1), h2 = 0.5
    rng = np.random.default_rng(1)
    bt = np.zeros(P)
    idx = rng.choice(P, max(10, P // 100), replace=False)
    bt[idx] = rng.normal(size=len(idx))
    g = s.xbeta(bt)
    e = np.random.default_rng(2).normal(size=N) * np.sqrt(g.var())
    y = 10.0 + g + e
    sum2pq = s.mpm().sum() / N
    v = 0.5 * y.var() / sum2pq
    df = 4.0
    if method == "BayesB":
        s.add_marker_set(0, P, 1, df, v * (df - 2) / df, [(j, j + 1) for j in range(P)], np.full(P, v), pi0=0.01, estPi=True)
    else:
        s.add_marker_set(0, P, 0, df, v * (df - 2) / df, [(0, P)], [v])
    s.set_y(y)
    ve = 0.5 * y.var()
    s.set_residual_prior(4.0, ve * 2.0 / 4.0)
    return s, setup_s


def cpu_baseline(N, P, sample_cols, sample_iters):
    """Reference-order CPU oracle (fp64, daxpy + ddot + daxpy per SNP, second copy of the panel) on a bounded sample."""
    from oracle import oracle as O
    X, mu = O.generate_panel(N, sample_cols)
    rng = np.random.default_rng(1)
    bt = np.zeros(sample_cols)
    idx = rng.choice(sample_cols, max(10, sample_cols // 100), replace=False)
    bt[idx] = rng.normal(size=len(idx))
    g = X.astype(np.float64) @ bt
    y = 10.0 + g + np.random.default_rng(2).normal(size=N) * np.sqrt(g.var())
    v = 0.5 * y.var() / float((mu * (1 - mu / 2)).sum())
    def timed(threads, iters, budget_s, cols=None):
        """it/s of the reference-order oracle on the first `cols` columns, in chunks so that a slow setting stops at the budget."""
        cols = sample_cols if cols is None else cols
        O.set_threads(threads)
        o = O.Oracle(order=0, seed=1001, chain=0)
        o.set_panel_f32(X[:, :cols])
        o.add_marker_set(0, cols, 0, 4.0, v * 0.5, [(0, cols)], [v])
        o.set_y(y)
        o.set_residual_prior(4.0, 0.25 * y.var())
        o.run(1)  # warm-up
        t0 = time.time()
        done = 0
        while done < iters and time.time() - t0 < budget_s:
            k = min(max(1, iters // 20), iters - done)
            o.run(k)
            done += k
        dt = time.time() - t0
        O.set_threads(1)
        return done / dt * cols / sample_cols, dt, done

    # (i) one thread, (ii) the cores of the box (threaded daxpy / ddot, as OpenBLAS would run them).
    # A GPU box shows far more CPUs than its share (16 per GPU): the thread count is capped, and the threaded setting is
    # first tried on a few hundred columns -- with oversubscribed cores every one of its 3 P parallel regions per iteration
    # costs milliseconds, and it is then left out.
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    ncores = max(1, min(avail, 16))
    cal_cols = min(sample_cols, 400)
    cal1, _, _ = timed(1, 2, 5.0, cal_cols)
    caln = cal1 if ncores == 1 else timed(ncores, 2, 5.0, cal_cols)[0]
    threads = ncores if caln > 1.15 * cal1 else 1
    its, dt, done = timed(threads, sample_iters, 30.0)
    return {
        "value": its * sample_cols / P,
        "unit": "it/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{done} iterations of N={N} x P={sample_cols} columns of the workload ({dt:.1f} s), it/s scaled by {sample_cols}/{P} "
                  f"(per-SNP cost is independent of P); reference-order C restatement, 24*N bytes of DRAM traffic per SNP; "
                  f"{threads} thread(s) chosen by a trial on {cal_cols} columns: 1 thread {cal1 * sample_cols / P:.3f} it/s, "
                  f"{ncores} threads {caln * sample_cols / P:.3f} it/s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--N", type=int, default=10000)
    ap.add_argument("--P", type=int, default=100000)
    ap.add_argument("--method", default="BayesPR", choices=["BayesPR", "BayesB"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N > 1 on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--cpu-cols", type=int, default=16000)
    ap.add_argument("--cpu-iters", type=int, default=100)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ngp_pkg import load_pkg
    ngp = load_pkg()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libnextgp_hip has no CPU fallback")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    s, setup_s = build_chain(ngp, local_rank, 1001 + rank, args.N, args.P, args.method)
    K, W = args.steps, args.warmup
    s.set_schedule(W + K, W, 1)
    s.run(W)
    s.get_timing()
    barrier()
    t0 = time.perf_counter()
    s.run(K)
    barrier()
    dt = time.perf_counter() - t0
    tm = s.get_timing()
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # dominant-kernel launch duration, HIP events on the library's own stream (one extra iteration)
    prof = s.profile_iteration()
    achieved = prof["bytes_per_launch"] / (prof["avg_ms"] * 1e-3) / 1e9
    bytes_iter = 4.0 * args.N * args.P
    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the committed
    # rocprofv3 summary of the same workload is quoted when the configuration matches (else null)
    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_k_sweep.json")
    if os.path.exists(pmc) and (args.N, args.P, args.method) == (10000, 100000, "BayesPR") and prof["launches"] == 1:
        pj = json.load(open(pmc))
        traffic, traffic_src = pj["hbm_bytes_per_launch"], "profiles/r01_pmc_k_sweep.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
    # posterior means across chains: ONE all-reduce of the packed sums over RCCL / xGMI
    allreduce_ms = None
    n = s.posterior_len()
    buf = torch.zeros(n, device="cuda", dtype=torch.float64)
    s.export_posterior_device(buf.data_ptr(), n)
    if world > 1:
        torch.cuda.synchronize()
        ta = time.perf_counter()
        ngp.multichain.allreduce_posterior(buf)
        torch.cuda.synchronize()
        allreduce_ms = (time.perf_counter() - ta) * 1e3
    pooled = ngp.multichain.unpack_means(buf.cpu().numpy(), args.P, s.nvb, s.nsets)
    nkept, post_mean_varE = pooled["nKept"], pooled["varE"]

    out = None
    if rank == 0:
        its = world * K / dt
        out = {
            "metric": "gibbs_iterations_per_sec",
            "value": its,
            "unit": "it/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": dt / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.method} single-trait Gibbs sweep, N={args.N} individuals x P={args.P} SNPs, fp32 panel in HBM "
                            "(BASELINE.json configs[1])",
                "N": args.N, "P": args.P, "method": args.method, "chains": world,
                "parallelism": "independent chains, one per GPU; one RCCL all-reduce of posterior sums at the end",
                "panel_dtype": "f32", "accumulate_dtype": "f64",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "ngp::k_sweep (persistent sweep: one launch streams the whole N x P fp32 panel once)" if prof["launches"] == 1 else "ngp::k_step (one 64-SNP column block per launch)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": prof["bytes_per_launch"],
                "launch_avg_ms": prof["avg_ms"],
                "launches_per_iteration": prof["launches"],
                "iteration_achieved": bytes_iter * (K / dt) / 1e9,
                "iteration_frac": bytes_iter * (K / dt) / 1e9 / HBM_PEAK_GBS,
            },
            "device_iter_ms": tm["iter_ms"] / max(tm["iters"], 1),
            "setup_s": setup_s,
            "allreduce_ms": allreduce_ms,
            "posterior_mean_varE": post_mean_varE,
            "pooled_kept_samples": nkept,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.N, args.P, min(args.cpu_cols, args.P), args.cpu_iters)
            out["speedup_vs_cpu_baseline"] = its / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
