"""Compact storage (one byte per genotype, analytic centring; include/nextgp_hip.h ngp_set_storage) against the blocked oracle's
compact arithmetic (oracle/ngp_oracle.c ora_set_panel_u8): bit-exact chains through the C ABI, every instantiated lag and
task count, plus the distance to the reference's own Float64 arithmetic (src/prepMatVec.jl:129, src/functions.jl:118-137)."""
import numpy as np
import pytest

from conftest import add_sets, make_problem

pytestmark = pytest.mark.gpu


def make_codes(O, N, P, seed=5):
    X, y, bt, v = make_problem(O, N, P, seed=seed)
    _, mu = O.generate_panel(N, P)
    G = np.rint(X.astype(np.float64) + mu[None, :]).astype(np.uint8)
    assert G.max() <= 2
    return G, y, v


def _pair(ngp, O, G, lag, near=None, max_shards=0, seed=1001, chain=0):
    s = ngp.Sampler(device=0, seed=seed, chain=chain, mode=1, lag=lag, storage="u8")
    if near is not None:
        s.set_near(near)
    if max_shards:
        s.set_max_shards(max_shards)
    s.set_panel(G, centre=True)
    R, S, nblk = s.layout()
    mode, D = s.config()
    assert mode == 1 and R % 16 == 0 and s.streamer() == (3, 7) and s.storage() == 1
    o = O.Oracle(order=1, seed=seed, chain=chain)
    o.set_panel_u8(G, R=R, S=S, D=D, near=s.near(), tform=s.chain_form())
    return s, o


def _same_chain(s, o, niter):
    a, b = s.get_state(), o.get_state()
    assert np.array_equal(a["delta"], b["delta"])
    for k in ("ycorr", "beta", "varBeta", "piHat"):
        assert np.array_equal(a[k], b[k]), k
    assert a["varE"] == b["varE"] and a["b"] == b["b"] and a["iter"] == b["iter"] == niter
    pa, pb = s.get_posterior_sums(), o.get_posterior_sums()
    for k in ("sum_beta", "sum_beta2", "sum_delta", "sum_varBeta", "sum_pi"):
        assert np.array_equal(pa[k], pb[k]), k
    assert pa["sum_varE"] == pb["sum_varE"] and pa["sum_b"] == pb["sum_b"]


def test_means_gram_and_mpm(ngp, O):
    G, y, v = make_codes(O, 500, 300)
    s, o = _pair(ngp, O, G, lag=6)
    assert np.array_equal(s.means(), o.means())
    assert np.array_equal(s.means(), G.sum(axis=0, dtype=np.int64) / 500.0)
    mpm = s.mpm()
    for t in (0, 2, 4):
        g = o.get_gram(t)
        assert np.array_equal(s.gram(t), g)
        n = min(64, 300 - 64 * t)
        assert np.array_equal(mpm[64 * t:64 * t + n], np.diag(g)[:n])
    # the centred sums of squares the reference forms (src/mme.jl:305-307), to rounding
    Xc = G.astype(np.float64) - s.means()[None, :]
    assert np.allclose(s.mpm(), (Xc * Xc).sum(axis=0), rtol=1e-12, atol=0)


CASES = [
    ("pr_single", 500, 1000, [(0, 1000, "PR")]),
    ("pr_ragged", 257, 130, [(0, 130, "PR")]),
    ("pr_regions", 300, 200, [(0, 200, ("PRw", 37))]),
    ("b_single", 500, 600, [(0, 600, "B")]),
    ("multi", 400, 450, [(0, 150, "PR"), (150, 170, "B"), (320, 130, "PR")]),
    ("c_single", 400, 520, [(0, 520, "C")]),
    ("tiny", 7, 3, [(0, 3, "PR")]),
    ("r_multi", 300, 330, [(0, 100, "Rfix"), (100, 90, "B"), (190, 100, "R2"), (290, 40, "PR")]),
]
# (lag requested, near lags, streamer workgroups): one update task per lane unless the shards are made tall
ENGINES = [(3, None, 0), (4, None, 0), (6, None, 0), (8, None, 0), (12, None, 0), (8, 2, 0), (12, 1, 0), (12, 4, 0),
           (8, None, 2), (4, None, 2), (6, None, 1), (8, 2, 3)]
ENGINE_IDS = ["lag3", "lag4", "lag6", "lag8", "lag12", "lag8_near2", "lag12_near1", "lag12_near4", "tall2_lag8", "tall2_lag4",
              "one_shard", "three_shards_near2"]


@pytest.mark.parametrize("engine", ENGINES, ids=ENGINE_IDS)
@pytest.mark.parametrize("name,N,P,spec", CASES, ids=[c[0] for c in CASES])
def test_compact_chain_bit_exact_vs_blocked_oracle(ngp, O, name, N, P, spec, engine):
    G, y, v = make_codes(O, N, P)
    s, o = _pair(ngp, O, G, lag=engine[0], near=engine[1], max_shards=engine[2])
    niter = 12
    for m in (s, o):
        add_sets(m, spec, v)
        m.set_y(y)
        m.set_residual_prior(4.0, 0.5 * y.var() * 0.5)
        m.set_schedule(niter, 4, 2)
        m.run(niter)
    _same_chain(s, o, niter)


@pytest.mark.parametrize("N,max_shards,tasks", [(900, 4, 2), (1800, 4, 4), (1400, 3, 4), (1300, 3, 2), (2000, 12, 1)])
def test_compact_tall_shards(ngp, O, N, max_shards, tasks):
    """Two and four update tasks per lane (shards of 225..448 and 449..896 rows) on a small panel."""
    G, y, v = make_codes(O, N, 200)
    s, o = _pair(ngp, O, G, lag=8, max_shards=max_shards)
    R, S, _ = s.layout()
    nuw = (R // 16 + 6) // 7
    assert {1: 1, 2: 2, 3: 4, 4: 4}[(4 * nuw + 7) // 8] == tasks and S <= max_shards
    assert s.config()[1] == {1: 8, 2: 8, 4: 4}[tasks]
    for m in (s, o):
        add_sets(m, [(0, 120, "PR"), (120, 80, "B")], v)
        m.set_y(y)
        m.set_residual_prior(4.0, 0.25 * y.var())
        m.set_schedule(8, 2, 2)
        m.run(8)
    _same_chain(s, o, 8)


def test_compact_agrees_with_the_reference_float64_arithmetic(ngp, O):
    """What the compact storage buys besides bandwidth: no fp32 rounding of the panel.  Against the reference-order oracle on
    the Float64 panel the reference would hold (g - mean), the chain differs by rounding of the sums only; the fp32 tiles of
    the default storage are ~1e-7 away on the same problem."""
    N, P = 600, 800
    G, y, v = make_codes(O, N, P)
    niter = 10
    ref = O.Oracle(order=0, seed=77, chain=0)
    ref.set_panel_u8(G)
    s8 = ngp.Sampler(device=0, seed=77, chain=0, mode=1, lag=8, storage="u8")
    s8.set_panel(G, centre=True)
    s4 = ngp.Sampler(device=0, seed=77, chain=0, mode=1, lag=8)
    s4.set_panel(G, centre=True)
    for m in (ref, s8, s4):
        add_sets(m, [(0, P, "PR")], v)
        m.set_y(y)
        m.set_residual_prior(4.0, 0.25 * y.var())
        m.set_schedule(niter, 2, 2)
        m.run(niter)
    r, a, b = ref.get_state(), s8.get_state(), s4.get_state()
    scale = np.abs(r["beta"][:P]).max()
    d8 = np.abs(a["beta"][:P] - r["beta"][:P]).max() / scale
    d4 = np.abs(b["beta"][:P] - r["beta"][:P]).max() / scale
    assert d8 < 1e-11, d8                      # tolerance: rounding of the fp64 sums over 10 iterations (measured ~1e-14)
    assert abs(a["varE"] - r["varE"]) < 1e-11 * r["varE"]
    assert d4 > 100 * d8                       # the fp32 panel is the larger deviation by orders of magnitude
    assert d4 < 1e-4


def test_compact_invariant_and_reproducibility(ngp, O):
    G, y, v = make_codes(O, 3000, 2000)
    outs = []
    for rep in range(2):
        s = ngp.Sampler(device=0, seed=9, chain=1, mode=1, lag=12, storage="u8")
        s.set_panel(G, centre=True)
        add_sets(s, [(0, 2000, "PR")], v)
        s.set_y(y)
        s.set_residual_prior(4.0, 0.25 * y.var())
        s.run(6)
        st = s.get_state()
        outs.append(st)
        resid = y - st["b"] - s.xbeta(st["beta"][:2000])
        assert np.abs(st["ycorr"][:3000] - resid).max() < 1e-9 * np.abs(y).max()
    assert np.array_equal(outs[0]["beta"], outs[1]["beta"]) and outs[0]["varE"] == outs[1]["varE"]


def test_compact_refuses_centred_input_and_mode0(ngp, O):
    s = ngp.Sampler(device=0, seed=1, chain=0, storage="u8")
    with pytest.raises(ngp.NextGPHipError, match="genotype codes"):
        s.set_panel(np.zeros((10, 10)))
    s = ngp.Sampler(device=0, seed=1, chain=0, mode=0, lag=1, storage="u8")
    with pytest.raises(ngp.NextGPHipError, match="persistent sweep"):
        s.set_panel(np.zeros((10, 10), dtype=np.uint8), centre=True)


@pytest.mark.parametrize("bits", [8, 2])
@pytest.mark.parametrize("storage", ["f32", "u8"])
def test_panel_file_gives_the_tiles_of_set_panel_u8(ngp, O, tmp_path, bits, storage):
    """Binary panel file (include/nextgp_hip.h ngp_load_panel_file) in place of the text file of src/prepMatVec.jl:116-131."""
    N, P = 333, 150   # N not a multiple of 4: the last byte of a two-bit column is partly empty
    G, y, v = make_codes(O, N, P)
    path = tmp_path / f"panel{bits}.bin"
    ngp.write_panel_file(path, G, bits=bits)
    assert ngp.read_panel_header(path) == (N, P, bits)
    assert path.stat().st_size == 32 + P * (N if bits == 8 else (N + 3) // 4)
    outs = []
    for via_file in (True, False):
        s = ngp.Sampler(device=0, seed=3, chain=0, mode=1, lag=6, storage=storage)
        if via_file:
            s.load_panel_file(path, centre=True)
        else:
            s.set_panel(G, centre=True)
        add_sets(s, [(0, P, "PR")], v)
        s.set_y(y)
        s.set_residual_prior(4.0, 0.25 * y.var())
        s.run(5)
        outs.append((s.mpm(), s.get_state()))
    assert np.array_equal(outs[0][0], outs[1][0])
    for k in ("ycorr", "beta", "varBeta"):
        assert np.array_equal(outs[0][1][k], outs[1][1][k]), k


def test_panel_file_errors(ngp, O, tmp_path):
    G = np.full((8, 4), 3, dtype=np.uint8)
    with pytest.raises(ngp.NextGPHipError):
        ngp.write_panel_file(tmp_path / "x.bin", G, bits=2)        # code 3 does not fit two bits
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"not a panel")
    with pytest.raises(ngp.NextGPHipError):
        ngp.read_panel_header(bad)
    s = ngp.Sampler(device=0, seed=1, chain=0)
    with pytest.raises(ngp.NextGPHipError, match="not a panel file"):
        s._chk(s.L.ngp_load_panel_file(s.h, str(bad).encode(), 1))
    ok = tmp_path / "trunc.bin"
    ngp.write_panel_file(ok, np.ones((100, 70), dtype=np.uint8), bits=8)
    ok.write_bytes(ok.read_bytes()[:-50])
    with pytest.raises(ngp.NextGPHipError, match="truncated"):
        s._chk(s.L.ngp_load_panel_file(s.h, str(ok).encode(), 1))


def test_compact_snapshot_resume(ngp, O, tmp_path):
    G, y, v = make_codes(O, 500, 400)
    def build():
        s = ngp.Sampler(device=0, seed=21, chain=2, mode=1, lag=8, storage="u8")
        s.set_panel(G, centre=True)
        add_sets(s, [(0, 250, "PR"), (250, 150, "B")], v)
        s.set_y(y)
        s.set_residual_prior(4.0, 0.25 * y.var())
        s.set_schedule(12, 2, 2)
        return s
    a = build(); a.run(12)
    b = build(); b.run(5); b.save_snapshot(tmp_path / "c.snap")
    c = build(); c.load_snapshot(tmp_path / "c.snap"); c.run(7)
    sa, sc = a.get_state(), c.get_state()
    for k in ("ycorr", "beta", "varBeta", "piHat"):
        assert np.array_equal(sa[k], sc[k]), k
    pa, pc = a.get_posterior_sums(), c.get_posterior_sums()
    assert np.array_equal(pa["sum_beta"], pc["sum_beta"]) and pa["nKept"] == pc["nKept"]


@pytest.mark.parametrize("N", [16, 17, 3568, 3569, 6000])
def test_compact_row_count_edges(ngp, O, N):
    """Shards that are exactly full (16 x 223 rows), one row over (32-row shards, the last one nearly empty), a single unit."""
    P = 70
    G, y, v = make_codes(O, N, P)
    s, o = _pair(ngp, O, G, lag=6)
    R, S, _ = s.layout()
    assert R * S >= N and (N > 3568 or R == 16)
    for m in (s, o):
        add_sets(m, [(0, P, "PR")], v)
        m.set_y(y)
        m.set_residual_prior(4.0, max(0.25 * y.var(), 1e-3))
        m.set_schedule(6, 0, 1)
        m.run(6)
    _same_chain(s, o, 6)


def test_compact_without_centring(ngp, O):
    """centre = 0: the means are zero and the codes are the panel (the reference always centres, src/prepMatVec.jl:129; the flag
    exists for callers that centre themselves) -- same chain as the fp32 storage of the same uncentred values, to rounding."""
    G, y, v = make_codes(O, 300, 200)
    outs = []
    for storage in ("u8", None):
        s = ngp.Sampler(device=0, seed=4, chain=0, mode=1, lag=6, storage=storage)
        s.set_panel(G, centre=False)
        if storage == "u8":
            assert np.all(s.means() == 0.0)
        add_sets(s, [(0, 200, "PR")], v)
        s.set_y(y); s.set_residual_prior(4.0, 0.25 * y.var()); s.run(8)
        outs.append(s.get_state())
    assert np.abs(outs[0]["beta"] - outs[1]["beta"]).max() < 1e-9 * max(1.0, np.abs(outs[1]["beta"]).max())
