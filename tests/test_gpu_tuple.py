"""Correlated (Tuple) BayesPR on the device -- sampleBayesPR!(::Tuple), /root/reference/src/functions.jl:140-154, sampleVarCovBetaPR
:513-516, set-up src/mme.jl:448-489 -- through the C ABI (ngp_add_marker_set_tuple) against the oracle: bit for bit against the
blocked order, 1e-9 against the reference order; k = 1 is the Symbol method, bit for bit."""
import numpy as np
import pytest

from conftest import make_problem
from test_tuple_main_oracle import add_tuple, tuple_problem

pytestmark = pytest.mark.gpu

ENGINES = {  # name -> Sampler kwargs, set_max_shards
    "persistent": (dict(), 0),
    "blocklaunch": (dict(mode=0, lag=1), 0),
    "persist_lag3": (dict(mode=1, lag=3), 0),
    "rows_lag4": (dict(mode=1, lag=4, streamer=2), 3),
}


def oracle_like(O, s, X, order=1, seed=21, chain=0):
    o = O.Oracle(order=order, seed=seed, chain=chain)
    if order == 1:
        R, S, _ = s.layout()
        o.set_panel_f32(X, R=R, S=S, D=s.config()[1], near=s.near(), nchain=s.streamer()[1], tform=s.chain_form())
    else:
        o.set_panel_f32(X)
    return o


@pytest.mark.parametrize("form", [0, 1], ids=["steps", "inverse_form"])
@pytest.mark.parametrize("engine", list(ENGINES))
@pytest.mark.parametrize("k,nloc", [(1, 90), (2, 75), (3, 50), (4, 40)])
def test_tuple_chain_bit_exact_vs_blocked_oracle(ngp, O, k, nloc, engine, form):
    N = 300
    Xp, y, vm, v, span, off = tuple_problem(O, ngp, N, nloc, k, extra=40)
    regions = [(0, nloc // 3), (nloc // 3, nloc)]
    kw, shards = ENGINES[engine]
    s = ngp.Sampler(device=0, seed=21, chain=0, **kw)
    s.set_chain_form(form)   # 1: the Tuple blocks (and the BayesPR ones) as dlt = T e0, T by k_tinv
    if shards:
        s.set_max_shards(shards)
    s.set_panel(Xp)
    o = oracle_like(O, s, Xp)
    for m in (s, o):
        add_tuple(m, nloc, k, vm, regions)
        m.add_marker_set(off, 40, 1, 4.0, v * 0.5, [(j, j + 1) for j in range(40)], [v] * 40, pi0=0.2, estPi=True)
        m.set_y(y); m.set_residual_prior(4.0, 0.5); m.set_schedule(12, 2, 2); m.run(12)
    a, b = s.get_state(), o.get_state()
    for key in ("ycorr", "beta", "delta", "varBeta", "piHat"):
        assert np.array_equal(a[key], b[key][:len(a[key])]), (key, np.abs(a[key] - b[key][:len(a[key])]).max())
    assert a["varE"] == b["varE"] and a["b"] == b["b"]
    pa, pb = s.get_posterior_sums(), o.get_posterior_sums()
    for key in ("sum_beta", "sum_beta2", "sum_varBeta"):
        assert np.array_equal(pa[key], pb[key]), key
    vb = a["varBeta"][:2 * k * k].reshape(2, k, k)
    assert np.isfinite(vb).all() and all((np.linalg.eigvalsh((m + m.T) / 2) > 0).all() for m in vb)


@pytest.mark.parametrize("k,nloc", [(2, 75), (3, 50)])
def test_tuple_chain_vs_reference_order_oracle(ngp, O, k, nloc):
    N = 300
    Xp, y, vm, v, span, off = tuple_problem(O, ngp, N, nloc, k, extra=40)
    regions = [(0, nloc)]
    s = ngp.Sampler(device=0, seed=5, chain=1)
    s.set_panel(Xp)
    o = oracle_like(O, s, Xp, order=0, seed=5, chain=1)
    for m in (s, o):
        add_tuple(m, nloc, k, vm, regions)
        m.add_marker_set(off, 40, 0, 4.0, v * 0.5, [(0, 40)], [v])
        m.set_y(y); m.set_residual_prior(4.0, 0.5); m.run(15)
    a, b = s.get_state(), o.get_state()
    assert np.abs(a["beta"] - b["beta"]).max() < 1e-9 * np.abs(b["beta"]).max()
    assert np.allclose(a["varBeta"], b["varBeta"], rtol=1e-8) and abs(a["varE"] / b["varE"] - 1) < 1e-10


@pytest.mark.parametrize("form", [0, 1], ids=["steps", "inverse_form"])
def test_one_set_tuple_is_the_symbol_method_on_the_device(ngp, O, form):
    """k = 1 (a 1 x 1 variance 'matrix', InverseWishart = scaled inverse chi-square) draws, bit for bit, the chain of a plain BayesPR
    set with the same regions -- the Tuple method's arithmetic contains the Symbol method's."""
    N, nloc = 250, 100
    X, y, bt, v = make_problem(O, N, 128 + 30, seed=8)
    regions = [(0, 37), (37, 100)]
    sv = v * 0.5
    res = []
    for tup in (True, False):
        s = ngp.Sampler(device=0, seed=9, chain=2)
        s.set_chain_form(form)
        s.set_panel(X)
        if tup:
            s.add_marker_set_tuple(0, nloc, 1, 4.0, [[sv * 4.0]], regions, [[v]])
        else:
            s.add_marker_set(0, nloc, 0, 4.0, sv, regions, [v, v])
        s.add_marker_set(128, 30, 0, 4.0, sv, [(0, 30)], [v])
        s.set_y(y); s.set_residual_prior(4.0, 0.5); s.run(15)
        res.append(s.get_state())
    for key in ("ycorr", "beta", "delta", "varBeta"):
        assert np.array_equal(res[0][key], res[1][key]), key
    assert res[0]["varE"] == res[1]["varE"]


def test_tuple_compact_storage_snapshot_and_fine_seam(ngp, O, tmp_path):
    """The same path over genotype codes (compact storage), across a snapshot / resume, and through the fine seam (ngp_sweep_set with
    the k x k variance matrices as the set's varBeta)."""
    N, nloc, k = 320, 60, 2
    rng = np.random.default_rng(11)
    p = rng.uniform(0.1, 0.5, size=nloc * k)
    G = rng.binomial(2, p, size=(N, nloc * k)).astype(np.uint8)
    Gt = ngp.tuple_panel([np.asfortranarray(G[:, :nloc]), np.asfortranarray(G[:, nloc:])])
    y = 3.0 + (G[:, 5] - G[:, 5].mean()) * 0.8 - (G[:, nloc + 5] - G[:, nloc + 5].mean()) * 0.5 + rng.normal(size=N)
    vm = 0.01 * (0.6 * np.eye(k) + 0.4)
    regions = [(0, 25), (25, 60)]
    s = ngp.Sampler(device=0, seed=3, chain=0, storage="u8")
    s.set_panel(Gt, centre=True)
    R, S, _ = s.layout()
    o = O.Oracle(order=1, seed=3, chain=0)
    o.set_panel_u8(Gt, R=R, S=S, D=s.config()[1], near=s.near(), tform=s.chain_form())
    for m in (s, o):
        add_tuple(m, nloc, k, vm, regions)
        m.set_y(y); m.set_residual_prior(4.0, 0.5); m.set_schedule(14, 2, 2); m.run(6)
    path = str(tmp_path / "t.ngpsnap")
    s.save_snapshot(path)
    s2 = ngp.Sampler(device=0, seed=99, chain=7, storage="u8")
    s2.set_panel(Gt, centre=True); add_tuple(s2, nloc, k, vm, regions); s2.set_y(y); s2.set_residual_prior(4.0, 0.5); s2.set_schedule(14, 2, 2)
    s2.load_snapshot(path)
    for m in (s2, o):
        m.run(8)
    a, b = s2.get_state(), o.get_state()
    for key in ("ycorr", "beta", "varBeta"):
        assert np.array_equal(a[key], b[key][:len(a[key])]), key
    assert np.array_equal(s2.get_posterior_sums()["sum_varBeta"], o.get_posterior_sums()["sum_varBeta"])
    # fine seam: one call of the set's callback with the caller's arrays
    f = ngp.Sampler(device=0, seed=3, chain=0, storage="u8")
    f.set_panel(Gt, centre=True); add_tuple(f, nloc, k, vm, regions); f.set_y(y)
    yc = (y - y.mean()).copy(); be = np.zeros(Gt.shape[1]); vb = np.tile(vm.ravel(), 2)
    f.sweep_set(0, 1.3, yc, be, vb)
    assert np.isfinite(be).all() and np.abs(be).max() > 0 and not np.array_equal(vb, np.tile(vm.ravel(), 2))
    assert np.allclose(vb.reshape(2, k, k), vb.reshape(2, k, k).transpose(0, 2, 1), rtol=1e-9)


def test_tuple_argument_checks(ngp, O):
    X, y, bt, v = make_problem(O, 100, 200, seed=2)
    s = ngp.Sampler(device=0, seed=1, chain=0)
    s.set_panel(X)
    vm = v * np.eye(2)
    with pytest.raises(ngp.NextGPHipError, match="block boundary"):
        s.add_marker_set_tuple(10, 20, 2, 5.0, vm * 2, [(0, 20)], vm)
    with pytest.raises(ngp.NextGPHipError, match="1..4"):
        s.add_marker_set_tuple(0, 10, 5, 8.0, np.eye(5), [(0, 10)], np.eye(5))
    with pytest.raises(ngp.NextGPHipError, match="outside the panel"):
        s.add_marker_set_tuple(128, 64, 2, 5.0, vm * 2, [(0, 64)], vm)
    s.add_marker_set_tuple(0, 40, 2, 5.0, vm * 2, [(0, 40)], vm)        # columns 0..79: owns blocks 0 and 1
    with pytest.raises(ngp.NextGPHipError, match="overlap"):
        s.add_marker_set(100, 50, 0, 4.0, v * 0.5, [(0, 50)], [v])      # column 100 is inside the tuple set's last block
    s.add_marker_set(128, 50, 0, 4.0, v * 0.5, [(0, 50)], [v])


@pytest.mark.parametrize("form", [0, 1], ids=["steps", "inverse_form"])
def test_tuple_sets_through_run_many_and_on_tall_panels(ngp, form):
    """Chains with a Tuple set share ONE fused launch per iteration (k_sweep_multi_tup: the fused kernel with the Tuple chain in its
    samplers), each bit for bit the chain it is alone.  And a Tuple set on an fp32 panel too tall for one shard per
    streamer workgroup (k_sweep_tall) draws the chain of the per-block engine (1e-9: another layout, another summation order)."""
    rng = np.random.default_rng(1)
    N, P, k = 3000, 6400, 2
    v = 0.01; V = v * (0.7 * np.eye(k) + 0.3); nloc = (P // 64) * (64 // k)

    def mk(seed, chain, owner=None):
        s = ngp.Sampler(device=0, seed=seed, chain=chain, mode=1, lag=8)
        s.set_chain_form(form)
        if owner is None:
            s.set_max_shards(100); s.generate_panel(N, P)
        else:
            s.share_panel(owner)
        return s

    def model(s, yy):
        s.add_marker_set_tuple(0, nloc, k, 3.0 + k, V * 0.5, [(0, nloc)], V); s.set_y(yy); s.set_residual_prior(4.0, 0.5)

    first = mk(1001, 0)
    bt = np.zeros(P); bt[rng.choice(P, 30, replace=False)] = rng.normal(size=30)
    y = 3 + first.xbeta(bt) + np.random.default_rng(2).normal(size=N)
    second = mk(1002, 1, owner=first)
    model(first, y); model(second, y + 0.01)
    ngp.Sampler.run_many([first, second], 15)
    assert first.census()["grid"] == 2 * (1 + (first.layout()[1] + 31) // 32) + first.layout()[1]     # one grid for both chains
    for f, (seed, chain, yy) in ((first, (1001, 0, y)), (second, (1002, 1, y + 0.01))):
        a = mk(seed, chain); model(a, yy); a.run(15)
        sf, sa = f.get_state(), a.get_state()
        assert all(np.array_equal(sf[q], sa[q]) for q in ("beta", "ycorr", "varBeta")) and sf["varE"] == sa["varE"]
    N2, P2 = 70000, 1280
    nl2 = (P2 // 64) * 32
    out = []
    for kw in ({}, dict(mode=0, lag=1)):
        t = ngp.Sampler(device=0, seed=5, chain=0, **kw); t.generate_panel(N2, P2)
        if not out:
            bt = np.zeros(P2); bt[rng.choice(P2, 10, replace=False)] = rng.normal(size=10)
            y2 = 3 + t.xbeta(bt) + np.random.default_rng(3).normal(size=N2)
            assert t.config() == (1, 3) and t.layout()[1] % 2 == 0
        t.add_marker_set_tuple(0, nl2, 2, 5.0, V * 0.5, [(0, nl2)], V); t.set_y(y2); t.set_residual_prior(4.0, 0.5); t.run(6)
        out.append(t.get_state())
    st, sr = out
    assert np.abs(st["beta"] - sr["beta"]).max() <= 1e-9 * max(1e-3, np.abs(sr["beta"]).max()) and abs(st["varE"] - sr["varE"]) <= 1e-9 * sr["varE"]
