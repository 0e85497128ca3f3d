"""N > 1 path on CPU: two ranks over gloo, each running its own chain (the oracle stands in for the GPU
sampler, which needs a device), packed posterior sums all-reduced once, as bench.py does over RCCL."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from conftest import add_sets, make_problem
    from ngp_pkg import load_pkg
    from oracle import oracle as O
    ngp = load_pkg()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, P = 60, 40
    X, y, bt, v = make_problem(O, N, P, seed=1)
    o = O.Oracle(0, seed=1001 + rank, chain=rank)           # seeds 1001+rank as in bench.py
    o.set_panel_f32(X); add_sets(o, [(0, 25, "PR"), (25, 15, "B")], v); o.set_y(y); o.set_schedule(30, 10, 2); o.run(30)
    ps = o.get_posterior_sums()
    nvb, nsets = len(ps["sum_varBeta"]), 2
    local = ngp.multichain.pack_posterior(ps, P, nvb, nsets)
    t = torch.from_numpy(local.copy())
    dt = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.barrier()
    ngp.multichain.allreduce_posterior(t)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)                # the timing reduction of bench.py
    np.save(os.path.join(out_dir, f"local{rank}.npy"), local)
    np.save(os.path.join(out_dir, f"pooled{rank}.npy"), t.numpy())
    assert dt.item() == float(world)
    dist.destroy_process_group()


def test_two_chains_pool_posterior_sums(tmp_path, ngp):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    loc = [np.load(tmp_path / f"local{r}.npy") for r in range(world)]
    pooled = [np.load(tmp_path / f"pooled{r}.npy") for r in range(world)]
    assert not np.array_equal(loc[0], loc[1])                    # different seeds, different chains
    assert np.array_equal(pooled[0], pooled[1]) and np.allclose(pooled[0], loc[0] + loc[1], rtol=0, atol=0)
    P, nsets = 40, 2
    nvb = len(loc[0]) - 3 * P - 2 * nsets - 3
    assert len(loc[0]) == ngp.multichain.posterior_len(P, nvb, nsets)
    m = ngp.multichain.unpack_means(pooled[0], P, nvb, nsets)
    assert m["nKept"] == 20 and m["beta"].shape == (P,) and m["varE"] > 0 and np.all(m["beta_sd"] >= 0)
    assert np.allclose(m["beta"], (loc[0][:P] + loc[1][:P]) / 20.0)
