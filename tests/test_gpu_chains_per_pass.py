"""K chains per pass over the panel (ngp_share_panel + ngp_run_many -> k_sweep_multi): independent chains that share one panel take
ONE sweep launch per iteration, every streamer forming X_t'[y_1 .. y_K] from each tile it reads.  Each chain must stay, bit for bit,
the chain it is alone (and so the blocked oracle's chain) -- the reference's loop being replaced is still
/root/reference/src/functions.jl:124-136, once per chain (src/samplers.jl:23: one chain per task)."""
import numpy as np
import pytest

from conftest import add_sets, make_problem

pytestmark = pytest.mark.gpu


def build_chains(ngp, X, y, v, spec, K, lag=None, shards=None, seeds=None):
    chains = []
    for c in range(K):
        s = ngp.Sampler(device=0, seed=(seeds or [1001 + i for i in range(K)])[c], chain=c, **({"mode": 1, "lag": lag} if lag else {}))
        if c == 0:
            s.set_max_shards(shards if shards else s.shards_for_pass(K))
            s.set_panel(X)
        else:
            s.share_panel(chains[0])
        add_sets(s, spec, v)
        s.set_y(y + 0.01 * c)                  # chains need not even share y
        s.set_residual_prior(4.0, 1.0)
        s.set_schedule(20, 4, 2)
        chains.append(s)
    return chains


@pytest.mark.parametrize("K,lag", [(2, None), (3, 6), (4, 6), (5, None), (6, None), (7, 6), (8, None)])
def test_fused_chains_equal_the_chains_alone_and_the_oracle(ngp, O, K, lag):
    N, P = 500, 640
    X, y, bt, v = make_problem(O, N, P, seed=6)
    spec = [(0, 300, "PR"), (300, 200, "B"), (500, 140, "R")]
    fused = build_chains(ngp, X, y, v, spec, K, lag=lag)
    R, S, _ = fused[0].layout()
    assert R <= 64 and fused[0].streamer()[0] == 1
    ngp.Sampler.run_many(fused, 20)
    # ONE launch served all chains: K samplers, S streamers, and reducers per group of 32 shards -- one workgroup per chain, or
    # (from four chains on) one per pair of chains
    assert fused[0].census()["grid"] == K + (((K + 1) // 2) if K >= 4 else K) * ((S + 31) // 32) + S
    for c in (0, K - 1):
        alone = ngp.Sampler(device=0, seed=1001 + c, chain=c, **({"mode": 1, "lag": lag} if lag else {}))
        alone.set_max_shards(S); alone.set_panel(X)
        assert alone.layout()[:2] == (R, S)
        o = O.Oracle(order=1, seed=1001 + c, chain=c)
        o.set_panel_f32(X, R=R, S=S, D=alone.config()[1], near=alone.near(), nchain=alone.streamer()[1], tform=alone.chain_form())
        for m in (alone, o):
            add_sets(m, spec, v); m.set_y(y + 0.01 * c); m.set_residual_prior(4.0, 1.0); m.set_schedule(20, 4, 2); m.run(20)
        a, b, f = alone.get_state(), o.get_state(), fused[c].get_state()
        for key in ("ycorr", "beta", "delta", "varBeta", "piHat"):
            assert np.array_equal(f[key], a[key]), (c, key)
            assert np.array_equal(f[key], b[key][:len(f[key])]), (c, key)
        assert f["varE"] == a["varE"] == b["varE"] and f["iter"] == 20
        pf, pa = fused[c].get_posterior_sums(), alone.get_posterior_sums()
        assert pf["nKept"] == pa["nKept"] == 8 and np.array_equal(pf["sum_beta"], pa["sum_beta"]) and pf["sum_varE"] == pa["sum_varE"]
        assert np.array_equal(fused[c].get_class_state(2)["piHat"], alone.get_class_state(2)["piHat"])
    # the chains are different chains (own seeds), and a second fused call continues them
    assert not np.array_equal(fused[0].get_state()["beta"], fused[1].get_state()["beta"])
    ngp.Sampler.run_many(fused, 5)
    assert fused[0].get_state()["iter"] == 25


@pytest.mark.parametrize("K,lag", [(2, 6), (2, 4)])
def test_fused_chains_on_the_row_owning_streamer(ngp, O, K, lag):
    """The same over the row-owning streamer (shards of 64+ rows, lag 6: the layout of the 50k x 600k shape): two or three chains in
    one launch, each bit for bit the chain it is alone and the blocked oracle's."""
    N, P = 700, 520
    X, y, bt, v = make_problem(O, N, P, seed=8)
    spec = [(0, 260, "PR"), (260, 260, "B")]
    fused = build_chains(ngp, X, y, v, spec, K, lag=lag, shards=6)
    R, S, _ = fused[0].layout()
    assert fused[0].streamer() == (2, 7) and R >= 64 and fused[0].config() == (1, lag)
    ngp.Sampler.run_many(fused, 20)
    assert fused[0].census()["grid"] == K * (1 + (S + 31) // 32) + S
    for c in range(K):
        o = O.Oracle(order=1, seed=1001 + c, chain=c)
        o.set_panel_f32(X, R=R, S=S, D=lag, near=fused[0].near(), nchain=7, tform=fused[0].chain_form())
        add_sets(o, spec, v); o.set_y(y + 0.01 * c); o.set_residual_prior(4.0, 1.0); o.set_schedule(20, 4, 2); o.run(20)
        f, b = fused[c].get_state(), o.get_state()
        for key in ("ycorr", "beta", "delta", "varBeta", "piHat"):
            assert np.array_equal(f[key], b[key][:len(f[key])]), (c, key)
        assert f["varE"] == b["varE"]


@pytest.mark.parametrize("K,lag,N,shards,tasks", [(2, 8, 500, 0, 1), (2, 6, 500, 0, 1), (2, 4, 500, 0, 1), (2, 8, 1500, 6, 2), (2, 4, 1500, 6, 2), (2, 6, 2900, 6, 4),
                                                   (3, 8, 500, 0, 1), (3, 6, 500, 0, 1), (3, 4, 500, 0, 1), (3, 8, 4000, 12, 2), (3, 4, 4000, 12, 2)],
                         ids=["lag8", "lag6", "lag4", "two_tasks_lag8", "two_tasks_lag4", "four_tasks",
                              "three_chains", "three_chains_lag6", "three_chains_lag4", "three_chains_two_tasks", "three_chains_two_tasks_lag4"])
def test_fused_chains_over_compact_storage(ngp, O, K, lag, N, shards, tasks):
    """Two or three chains per pass over byte tiles (role_streamer_rows_multi<.., K, ST>): every byte converted once for all chains; each
    chain bit for bit the blocked oracle's compact chain (oracle/ngp_oracle.c ora_set_panel_u8) and the chain it is alone."""
    from test_gpu_compact import make_codes
    P = 450
    G, y, v = make_codes(O, N, P)
    spec = [(0, 150, "PR"), (150, 170, "B"), (320, 130, "R")]
    fused = []
    for c in range(K):
        s = ngp.Sampler(device=0, seed=1001 + c, chain=c, mode=1, lag=lag, storage="u8")
        if c == 0:
            s.set_max_shards(shards if shards else s.shards_for_pass(K))
            s.set_panel(G, centre=True)
        else:
            s.share_panel(fused[0])
        add_sets(s, spec, v); s.set_y(y + 0.01 * c); s.set_residual_prior(4.0, 1.0); s.set_schedule(20, 4, 2)
        fused.append(s)
    R, S, _ = fused[0].layout()
    D = fused[0].config()[1]
    assert fused[0].streamer() == (3, 7) and fused[1].storage() == 1
    nuw = -(-(R // 16) // 7)                                 # units of 16 rows per row-owning wave -> update tasks per lane
    assert {1: 1, 2: 2, 3: 4, 4: 4}[(4 * nuw + 7) // 8] == tasks
    ngp.Sampler.run_many(fused, 20)
    assert fused[0].census()["grid"] == K * (1 + (S + 31) // 32) + S          # one launch for both chains
    for c in range(K):
        o = O.Oracle(order=1, seed=1001 + c, chain=c)
        o.set_panel_u8(G, R=R, S=S, D=D, near=fused[0].near(), tform=fused[0].chain_form())
        add_sets(o, spec, v); o.set_y(y + 0.01 * c); o.set_residual_prior(4.0, 1.0); o.set_schedule(20, 4, 2); o.run(20)
        f, b = fused[c].get_state(), o.get_state()
        for k in ("ycorr", "beta", "delta", "varBeta", "piHat"):
            assert np.array_equal(f[k], b[k]), (c, k)
        assert f["varE"] == b["varE"] and f["b"] == b["b"]
        pf, pb = fused[c].get_posterior_sums(), o.get_posterior_sums()
        for k in ("sum_beta", "sum_beta2", "sum_delta", "sum_varBeta"):
            assert np.array_equal(pf[k], pb[k]), (c, k)


def test_shared_panel_lifetime_and_fallback(ngp, O):
    """The panel arrays live as long as any handle refers to them; handles whose engine the fused kernel does not serve still run
    through ngp_run_many (side by side, one thread each), with the same results."""
    N, P = 400, 256
    X, y, bt, v = make_problem(O, N, P, seed=3)
    spec = [(0, 256, "PR")]
    a, b = build_chains(ngp, X, y, v, spec, 2)
    mp = a.mpm().copy()
    a.close()                                   # the owner goes first: the sharer keeps the panel alive
    b.run(5)
    assert np.array_equal(b.mpm(), mp) and np.isfinite(b.get_state()["beta"]).all()
    # row-owning streamer at a lag the fused kernel does not serve: not fused, still correct
    c = build_chains(ngp, X, y, v, spec, 2, lag=5, shards=5)
    assert c[0].streamer()[0] == 2
    ngp.Sampler.run_many(c, 6)
    solo = ngp.Sampler(device=0, seed=1002, chain=1, mode=1, lag=5)
    solo.set_max_shards(5); solo.set_panel(X); add_sets(solo, spec, v); solo.set_y(y + 0.01); solo.set_residual_prior(4.0, 1.0); solo.set_schedule(20, 4, 2)
    solo.run(6)
    assert np.array_equal(solo.get_state()["beta"], c[1].get_state()["beta"])
    with pytest.raises(ngp.NextGPHipError, match="no panel"):
        ngp.Sampler(device=0).share_panel(ngp.Sampler(device=0))
