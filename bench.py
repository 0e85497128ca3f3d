#!/usr/bin/env python3
"""bench.py -- Gibbs iterations/s (and effective samples/s) of the marker-effect sampler on a synthetic N x P SNP panel.

    python bench.py --gpus N --steps K --warmup W [--config C4|C2|C3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one full Gibbs iteration (varE draw, intercept draw, sweep of all P SNPs with residual update, variance
draws, posterior accumulation).  Default workload = the configuration BASELINE.json's north-star target is stated on and
the largest that fits one GPU: configs[3], N = 50,000 individuals x P = 600,000 SNPs as three BayesPR marker sets of
200,000 columns (the reachable multi-breed random regression, SURVEY.md section 8 d), fp32 panel resident in HBM (120 GB,
generated on the device).  --config C2 / C3 select configs[1] / configs[2] (10k x 100k BayesPR / BayesB).

With N > 1 every rank runs its own independent chain on its own GPU (seeds 1001+rank, weak scaling, no data-path
collective); the posterior sums are all-reduced once over RCCL after the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

# before anything can initialise HIP / HSA (the host driver only supports dmabuf IPC; RCCL needs this for N > 1)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("OMP_WAIT_POLICY", "ACTIVE")  # CPU baseline: threads spin at the per-SNP barrier instead of sleeping

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

CONFIGS = {
    # name: (N, P, [(method, ncol), ...], BASELINE.json entry)
    "C4": (50000, 600000, [("BayesPR", 200000)] * 3, "configs[3]: multi-breed random regression = three BayesPR marker sets of 200,000 SNPs"),
    "C2": (10000, 100000, [("BayesPR", 100000)], "configs[1]: BayesPR single-trait"),
    "C3": (10000, 100000, [("BayesB", 100000)], "configs[2]: BayesB (variable-selection indicators)"),
}
N_TRACE_LOCI = 128  # ESS of varE, of each varBeta / pi, and the minimum over a fixed set of 128 effects


def split_sets(P, sets):
    """Scale the set sizes of a named configuration to an overridden P (sizes keep their proportions, last set takes the rest)."""
    tot = sum(n for _, n in sets)
    out, c0 = [], 0
    for i, (m, n) in enumerate(sets):
        ncol = P - c0 if i == len(sets) - 1 else max(1, int(round(n * P / tot)))
        out.append((m, c0, ncol))
        c0 += ncol
    return out


def simulate_y(xbeta, N, P):
    """y = 10 + X beta + e, 1 % causal SNPs ~ N(0,1), h2 = 0.5."""
    rng = np.random.default_rng(1)
    bt = np.zeros(P)
    idx = rng.choice(P, max(10, P // 100), replace=False)
    bt[idx] = rng.normal(size=len(idx))
    g = xbeta(bt)
    e = np.random.default_rng(2).normal(size=N) * np.sqrt(g.var())
    return 10.0 + g + e


def build_chain(ngp, device, seed, N, P, sets, panel_seed=20250509, storage=None, share=0, per_pass=0, owner=None, y=None):
    # chains per pass on tall shards (row-owning streamer): two chains at lag 4 -- its register delay line leaves room for the second
    # chain's arithmetic there (measured at 50k x 600k: 58.6 it/s aggregate at lag 4, 55.2 at lag 5, 47.5 at lag 6; one chain: 41.9 at lag 6)
    eng = dict(mode=1, lag=4) if (per_pass > 1 or owner is not None) and N >= 64 * 247 and storage is None else {}
    s = ngp.Sampler(device=device, seed=seed, chain=seed - 1001, storage=storage, **eng)
    if share > 1:  # this chain is one of `share` that run side by side on the device
        s.set_max_shards(s.shards_for_chains(share))
    if per_pass > 1 and owner is None:  # the first of `per_pass` chains that share ONE fused sweep launch (and one panel)
        s.set_max_shards(s.shards_for_pass(per_pass))
    t0 = time.time()
    if owner is not None:
        s.share_panel(owner)  # no second copy of the panel
    else:
        s.generate_panel(N, P, 0.05, 0.5, panel_seed)
    setup_s = time.time() - t0
    if y is None:
        y = simulate_y(s.xbeta, N, P)
    v = 0.5 * y.var() / (s.mpm().sum() / N)
    df = 4.0
    for method, col0, ncol in sets:
        if method == "BayesB":
            s.add_marker_set(col0, ncol, 1, df, v * (df - 2) / df, [(j, j + 1) for j in range(ncol)], np.full(ncol, v), pi0=0.01, estPi=True)
        else:
            s.add_marker_set(col0, ncol, 0, df, v * (df - 2) / df, [(0, ncol)], [v])
    s.set_y(y)
    s.set_residual_prior(4.0, 0.5 * y.var() * 2.0 / 4.0)
    return s, setup_s


def ess_geyer(x):
    """Effective sample size by Geyer's initial positive sequence estimator (SURVEY.md section 8 d)."""
    x = np.asarray(x, float)
    x = x - x.mean()
    n = len(x)
    if n < 4 or not np.any(x):
        return float(n)
    f = np.fft.rfft(x, 2 * n)
    acf = np.fft.irfft(f * np.conj(f))[:n] / (x @ x)
    tau, k = -1.0, 0
    while k + 1 < n:
        pair = acf[k] + acf[k + 1]
        if pair <= 0:
            break
        tau += 2.0 * pair
        k += 2
    return float(min(n, n / max(tau, 1e-12)))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(N, P, sets, sample_cols, budget_s):
    """Reference-order CPU oracle (fp64; add-back daxpy, ddot over the Mp copy, update daxpy per SNP: 24 N bytes of DRAM traffic,
    src/functions.jl:128-133) on a bounded sample: N rows x sample_cols columns of the same synthetic panel, it/s scaled by
    sample_cols / P (the per-SNP cost does not depend on P).  Two settings, both reported: one thread, and all cores of this
    box's share with ONE parallel region per sweep (oracle/ngp_oracle.c: ora_run_pr_threaded); the faster one is `value`."""
    from oracle import oracle as O
    X, mu = O.generate_panel(N, sample_cols)
    y = simulate_y(lambda b: X.astype(np.float64) @ b, N, sample_cols)
    v = 0.5 * y.var() / float((mu * (1 - mu / 2)).sum())
    threaded_ok = all(m == "BayesPR" for m, _, _ in sets)

    def make():
        o = O.Oracle(order=0, seed=1001, chain=0)
        o.set_panel_f32(X)
        for m, c0, n in split_sets(sample_cols, [(m, n) for m, _, n in sets]):
            if m == "BayesB":
                o.add_marker_set(c0, n, 1, 4.0, v * 0.5, [(j, j + 1) for j in range(n)], np.full(n, v), pi0=0.01, estPi=True)
            else:
                o.add_marker_set(c0, n, 0, 4.0, v * 0.5, [(0, n)], [v])
        o.set_y(y)
        o.set_residual_prior(4.0, 0.25 * y.var())
        return o

    def timed(run, budget):
        run(1)  # warm-up (page-in, thread pool)
        t0, done = time.time(), 0
        while time.time() - t0 < budget or done < 2:
            run(1)
            done += 1
        return done / (time.time() - t0), done

    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    ncores = max(1, min(avail, 16))  # a GPU box shows far more CPUs than its share: 16 per GPU
    o1 = make()
    its1, n1 = timed(o1.run, budget_s / 2)
    itsn, nn = (None, 0)
    if threaded_ok and ncores > 1:
        on = make()
        itsn, nn = timed(lambda k: on.run_pr_threaded(k, ncores), budget_s / 2)
    scale = sample_cols / P
    best, cores = (itsn, ncores) if (itsn is not None and itsn > its1) else (its1, 1)
    return {
        "value": best * scale,
        "unit": "it/s",
        "cores": cores,
        "kind": "port",
        "cpu_model": cpu_model(),
        "cpus_available": avail,
        "one_thread_it_per_s": its1 * scale,
        "all_cores_it_per_s": None if itsn is None else itsn * scale,
        "all_cores_threads": ncores if itsn is not None else None,
        "sample": f"N={N} rows x {sample_cols} columns of the workload's panel, {n1} iterations on 1 thread and {nn} on {ncores} threads "
                  f"(about {budget_s:.0f} s in all), it/s scaled by {sample_cols}/{P}: per-SNP cost is independent of P; reference-order C "
                  f"restatement with the reference's second copy of the panel (24*N bytes of DRAM traffic per SNP); the threaded "
                  f"setting keeps one parallel region per sweep, one barrier per SNP",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="C4", choices=sorted(CONFIGS))
    ap.add_argument("--N", type=int, default=None, help="override the number of individuals (parity / contract tests)")
    ap.add_argument("--P", type=int, default=None, help="override the number of SNPs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-compact", action="store_true", help="skip the extra leg in compact (one byte per genotype) storage")
    ap.add_argument("--chains-per-gpu", type=int, default=0,
                    help="extra leg: that many independent chains side by side on the GPU (aggregate it/s; pays where one chain is not bandwidth-bound)")
    ap.add_argument("--chains-per-pass", type=int, default=-1,
                    help="extra leg: that many independent chains in ONE fused sweep launch per iteration, the panel streamed once for all of them "
                         "(aggregate it/s beside the single-chain value; 2..8; default: 8 on shards of at most 64 rows, 2 on taller ones; 0 or 1: skip)")
    ap.add_argument("--storage", default="f32", choices=["f32", "u8"],
                    help="panel storage of the MAIN measurement (default f32 = the headline; u8 = compact storage, for profiling that mode)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N > 1 on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses GPU 0 (the persistent kernels then run one after the other)")
    ap.add_argument("--cpu-cols", type=int, default=None, help="columns of the CPU baseline sample (default: about 1.6e8 / N)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--seed-offset", type=int, default=0, help="chain seed = 1001 + rank + this (tests: a rank's chain run alone)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ngp_pkg import load_pkg
    ngp = load_pkg()

    cN, cP, csets, cdesc = CONFIGS[args.config]
    N, P = args.N or cN, args.P or cP
    sets = split_sets(P, csets)
    overridden = (N, P) != (cN, cP)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libnextgp_hip has no CPU fallback")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    compact_main = args.storage == "u8"
    s, setup_s = build_chain(ngp, local_rank, 1001 + rank + args.seed_offset, N, P, sets, storage="u8" if compact_main else None)
    K, W = args.steps, args.warmup
    loci = np.unique(np.linspace(0, P - 1, N_TRACE_LOCI).astype(np.int64))
    ntvb = min(s.nvb, 8)
    s.set_trace_loci(loci, ntvb)
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
    # effective sample sizes over the K timed iterations (rank 0's chain), Geyer's initial positive sequence
    # (only when at least 200 kept iterations back the estimate: below that the figure is noise)
    ess, ess_min = None, None
    if K >= 200:
        tr, trx = s.get_trace(K), s.get_trace_ext(K)
        ess = {"varE": ess_geyer(tr["varE"]),
               "varBeta": [ess_geyer(trx["varBeta"][:, i]) for i in range(ntvb)],
               "beta_min_of_%d" % len(loci): min(ess_geyer(trx["beta"][:, i]) for i in range(len(loci)))}
        if any(m == "BayesB" for m, _, _ in sets):
            ess["pi"] = [ess_geyer(trx["pi"][:, i]) for i in range(s.nsets)]
        ess_min = min([ess["varE"], ess["beta_min_of_%d" % len(loci)]] + ess["varBeta"] + ess.get("pi", []))
    # dominant-kernel launch duration, HIP events on the library's own stream (five extra iterations, averaged)
    profs = [s.profile_iteration() for _ in range(5)]
    prof = dict(profs[0], avg_ms=float(np.mean([p["avg_ms"] for p in profs])))
    achieved = prof["bytes_per_launch"] / (prof["avg_ms"] * 1e-3) / 1e9
    bytes_iter = (1.0 if compact_main else 4.0) * N * P
    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the committed rocprofv3
    # summary of the SAME workload (tools/profile_round.sh) is quoted when the configuration matches, else null
    traffic, traffic_src = None, None
    pmc = next((q for q in (os.path.join(ROOT, "profiles", f"{tag}_pmc_k_sweep_{args.config}{'u8' if compact_main else ''}.json") for tag in ("r04", "r03", "r02"))
                if os.path.exists(q)), "")
    if pmc and not overridden and prof["launches"] == 1:
        pj = json.load(open(pmc))
        traffic, traffic_src = pj["hbm_bytes_per_launch"], f"profiles/{os.path.basename(pmc)} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
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
    pooled = ngp.multichain.unpack_means(buf.cpu().numpy(), P, s.nvb, s.nsets)
    nkept, post_mean_varE = pooled["nKept"], pooled["varE"]
    R, S, nblk = s.layout()
    mode, lag = s.config()
    variant, nchain = s.streamer()
    main_census = s.census() if mode == 1 else {"retries": 0, "exclusive": 0}
    setup_parts = s.setup_timing()

    if rank == 0:
        its = world * K / dt
        setdesc = " + ".join(f"{m}({ncol})" for m, _, ncol in sets)
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
                "workload": f"single-trait Gibbs sweep, N={N} individuals x P={P} SNPs, marker sets {setdesc}, {'u8 genotype codes + f64 column means (compact storage)' if compact_main else 'fp32 panel'} in HBM "
                            + (f"(BASELINE.json {cdesc})" if not overridden else f"(shape overridden from {args.config})"),
                "name": args.config if not overridden else f"{args.config}-override", "N": N, "P": P,
                "sets": [{"method": m, "col0": c0, "ncol": n} for m, c0, n in sets], "chains": world,
                "parallelism": "independent chains, one per GPU; one RCCL all-reduce of posterior sums at the end",
                "panel_dtype": "u8" if compact_main else "f32", "accumulate_dtype": "f64",
                "layout": {"rows_per_shard": R, "shards": S, "blocks": nblk, "engine": mode, "lag": lag, "near_lags": s.near(),
                           "streamer": variant, "gemv_chains": nchain},
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "ngp::k_sweep<false> (the persistent sweep kernel: one launch streams the whole N x P panel once; models with a Tuple / BayesR set run the k_sweep_tup / k_sweep_r instantiations)" if prof["launches"] == 1 else "ngp::k_step (one 64-SNP column block per launch)",
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
            "effective_samples": {
                "estimator": "Geyer initial positive sequence over the timed iterations of rank 0's chain (every iteration kept)",
                "ess": ess,
                "ess_min": ess_min,
                "ess_min_per_sec": None if ess_min is None else ess_min / dt,
                "ess_varE_per_sec": None if ess is None else ess["varE"] / dt,
                "note": "single-chain figures; with n_gpus chains the pooled rate is n_gpus times these"
                        + ("" if ess is not None else "; not estimated: fewer than 200 timed iterations (run with --steps 1000 for an ESS figure)"),
            },
            "device_iter_ms": tm["iter_ms"] / max(tm["iters"], 1),
            "census_retries": main_census["retries"], "exclusive": bool(main_census["exclusive"]),   # launches run again alone on the device
            "setup_s": setup_s, "setup_parts_ms": setup_parts,   # allocation (+ zeroing) | tile generation | Gram window
            "allreduce_ms": allreduce_ms,
            "posterior_mean_varE": post_mean_varE,
            "pooled_kept_samples": nkept,
        }
        if world == 1 and not args.no_compact and not compact_main:
            # the same workload with the panel kept one byte per genotype (ngp_set_storage: analytic centring, no fp32 rounding of
            # the panel) -- reported BESIDE the fp32 headline above, never instead of it
            s.close()
            del buf
            torch.cuda.empty_cache()
            c, csetup = build_chain(ngp, local_rank, 1001, N, P, sets, storage="u8")
            c.set_schedule(W + K, W, 1)
            c.run(W)
            torch.cuda.synchronize()
            tc = time.perf_counter()
            c.run(K)
            torch.cuda.synchronize()
            cdt = time.perf_counter() - tc
            cprofs = [c.profile_iteration() for _ in range(5)]
            cms = float(np.mean([p["avg_ms"] for p in cprofs]))
            cR, cS, _ = c.layout()
            out["compact_storage"] = {
                "value": K / cdt, "unit": "it/s", "ms_per_step": cdt / K * 1e3, "panel_dtype": "u8 codes + f64 column means",
                "panel_bytes": float(N) * float(P), "launch_avg_ms": cms,
                "roofline": {"bound": "hbm", "achieved": cprofs[0]["bytes_per_launch"] / (cms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": cprofs[0]["bytes_per_launch"] / (cms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "note": "not bandwidth-bound: at a quarter of the bytes the sweep is bound by the serial chain of the "
                                     "sampler workgroup and the streamers' arithmetic (DESIGN.md)"},
                "layout": {"rows_per_shard": cR, "shards": cS, "lag": c.config()[1], "near_lags": c.near(), "streamer": c.streamer()[0]},
                "speedup_vs_fp32_storage": (K / cdt) / (K / dt), "setup_s": csetup,
            }
            c.close()
            # ... and three chains per pass over the byte tiles (every byte converted once for all of them)
            KB = 3
            first, _ = build_chain(ngp, local_rank, 1001, N, P, sets, storage="u8", per_pass=KB)
            ysh = simulate_y(first.xbeta, N, P)
            cs = [first] + [build_chain(ngp, local_rank, 1001 + i, N, P, sets, storage="u8", owner=first, y=ysh)[0] for i in range(1, KB)]
            for c in cs:
                c.set_schedule(W + K, W, 1)
            ngp.Sampler.run_many(cs, W)
            cs[0].get_timing()
            torch.cuda.synchronize()
            tk = time.perf_counter()
            ngp.Sampler.run_many(cs, K)
            torch.cuda.synchronize()
            kdt = time.perf_counter() - tk
            ptm, cen = cs[0].get_timing(), cs[0].census()
            out["compact_storage"]["chains_per_pass"] = {
                "chains": KB, "value": KB * K / kdt, "unit": "it/s (aggregate over the chains of ONE fused launch per iteration)",
                "ms_per_pass": kdt / K * 1e3, "rows_per_shard": cs[0].layout()[0], "shards": cs[0].layout()[1], "lag": cs[0].config()[1],
                "census_retries": cen["retries"], "exclusive": bool(cen["exclusive"]), "fused": bool(ptm["sweep_launches"] == K),
                "speedup_vs_single_chain": (KB * K / kdt) / (K / cdt)}
            for c in cs:
                c.close()
        if world == 1 and args.chains_per_gpu > 1 and not compact_main:
            # independent chains side by side on ONE GPU (disjoint CU shares, one thread each inside the library): the aggregate
            # rate of the same metric where a single chain is bound by its sampler workgroup -- beside the single-chain value
            try:
                s.close()
            except Exception:  # noqa: BLE001
                pass
            torch.cuda.empty_cache()
            kc = args.chains_per_gpu
            cs = [build_chain(ngp, local_rank, 1001 + i, N, P, sets, share=kc)[0] for i in range(kc)]
            for c in cs:
                c.set_schedule(W + K, W, 1)
            ngp.Sampler.run_many(cs, W)
            torch.cuda.synchronize()
            tk = time.perf_counter()
            ngp.Sampler.run_many(cs, K)
            torch.cuda.synchronize()
            kdt = time.perf_counter() - tk
            gcen = [c.census() for c in cs]
            out["chains_per_gpu"] = {"chains": kc, "value": kc * K / kdt, "unit": "it/s (aggregate over the chains of this GPU)",
                                     "census_retries": int(sum(g["retries"] for g in gcen)), "exclusive": bool(any(g["exclusive"] for g in gcen)),
                                     "ms_per_step_of_a_chain": kdt / K * 1e3, "rows_per_shard": cs[0].layout()[0], "shards_per_chain": cs[0].layout()[1],
                                     "speedup_vs_single_chain": (kc * K / kdt) / its}
            for c in cs:
                c.close()
        if args.chains_per_pass < 0:  # default: what the fused kernel serves at this shape (phase streamer: 8 chains, row-owning streamer: 2)
            args.chains_per_pass = (2 if N >= 64 * 247 else 8) if (world == 1 and not compact_main and N <= 224 * 240) else 0
        if world == 1 and args.chains_per_pass > 1 and not compact_main:
            # K independent chains in ONE fused sweep launch per iteration: every streamer forms X_t'[y_1 .. y_K] from each tile it
            # reads, so one pass over the panel (4 N P algorithmic bytes) serves K iterations' worth of sampling -- the aggregate
            # rate BESIDE the single-chain value above, never instead of it; each chain is bit for bit the chain it is alone
            try:
                s.close()
            except Exception:  # noqa: BLE001
                pass
            torch.cuda.empty_cache()
            kp = args.chains_per_pass
            first, _ = build_chain(ngp, local_rank, 1001, N, P, sets, per_pass=kp)
            ysh = simulate_y(first.xbeta, N, P)
            cs = [first] + [build_chain(ngp, local_rank, 1001 + i, N, P, sets, owner=first, y=ysh)[0] for i in range(1, kp)]
            for c in cs:
                c.set_schedule(W + K, W, 1)
            ngp.Sampler.run_many(cs, W)
            cs[0].get_timing()
            torch.cuda.synchronize()
            tk = time.perf_counter()
            ngp.Sampler.run_many(cs, K)
            torch.cuda.synchronize()
            kdt = time.perf_counter() - tk
            ptm = cs[0].get_timing()
            cen = cs[0].census()
            pR, pS, _ = cs[0].layout()
            out["chains_per_pass"] = {
                "chains": kp, "value": kp * K / kdt, "unit": "it/s (aggregate over the chains of ONE fused launch per iteration)",
                "ms_per_pass": kdt / K * 1e3, "device_ms_per_pass": ptm["iter_ms"] / max(ptm["iters"], 1), "sweep_launches": ptm["sweep_launches"],
                "algorithmic_bytes_per_pass": 4.0 * N * P, "panel_stream_GBps": 4.0 * N * P * (K / kdt) / 1e9,
                "panel_stream_frac_of_peak": 4.0 * N * P * (K / kdt) / 1e9 / HBM_PEAK_GBS,
                "grid_workgroups": cen["grid"], "rows_per_shard": pR, "shards": pS, "lag": cs[0].config()[1],
                "census_retries": cen["retries"], "exclusive": bool(cen["exclusive"]),
                "fused": bool(ptm["sweep_launches"] == K),   # ONE launch per iteration served all the chains (else they ran side by side)
                "speedup_vs_single_chain": (kp * K / kdt) / its if its else None,
                "note": "independent chains (own seeds, own draws); every chain bit-identical to the chain it is alone with this layout "
                        "(tests/test_gpu_chains_per_pass.py)",
            }
            for c in cs:
                c.close()
        if world == 1 and not args.no_cpu_baseline:
            cols = args.cpu_cols or int(max(256, min(P, 1.6e8 // N)))
            out["cpu_baseline"] = cpu_baseline(N, P, sets, min(cols, P), args.cpu_seconds)
            out["speedup_vs_cpu_baseline"] = its / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
