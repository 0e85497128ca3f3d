"""Host-side mirror of the reference interface: parsing, region building, output files (CPU) and an
end-to-end runLMEM on the GPU whose files must reproduce the oracle chain."""
import os

import numpy as np
import pytest

from conftest import make_problem


def test_parse_formula(ngp):
    lhs, ic, snps = ngp.parse_formula('y ~ 1 + SNP(M, "geno.txt")')
    assert (lhs, ic) == ("y", True) and snps[0].name == "M" and snps[0].path == "geno.txt" and snps[0].map == ""
    lhs, ic, snps = ngp.parse_formula('pheno ~ 0 + SNP(M1,"a b.txt","map.csv") + SNP(M2, "g2.txt")')
    assert (lhs, ic) == ("pheno", False) and [s.name for s in snps] == ["M1", "M2"] and snps[0].map == "map.csv"
    for bad in ("y ~ 1 + x1&x2 + SNP(M,\"g\")", "y ~ 1 + PED(ID) + SNP(M,\"g\")", "y ~ 1 + (1|herd) + SNP(M,\"g\")"):
        with pytest.raises(NotImplementedError, match="Julia path"):
            ngp.parse_formula(bad)
    first = ngp.parse_formula('y ~ 1 + age + herd + SNP(M,"g")')
    second = ngp.parse_formula('y ~ 1 + SNP(M,"g")')                  # a later parse does not change what an earlier one returned
    assert first.covariates == ["age", "herd"] and second.covariates == []   # covariates / factors: fixed-effect sets on the device
    assert not hasattr(ngp.parse_formula, "last_covariates")
    Xd, names = ngp.design_columns("herd", np.array(["a", "c", "b", "a"]))
    assert names == ["herd: b", "herd: c"] and np.array_equal(Xd, [[0, 0], [0, 1], [1, 0], [0, 0]])
    Xc, _ = ngp.design_columns("age", np.array([1.0, 2.0, 6.0]))
    assert np.allclose(Xc[:, 0], [-2.0, -1.0, 3.0])                   # Float columns are centred


def test_priors_have_reference_fields(ngp):
    p = ngp.BayesPR(9999, 0.001)
    assert (p.r, p.v, p.name) == (9999, 0.001, "BayesPR")          # src/runTime.jl:30-45
    b = ngp.BayesB(0.05, 0.01, estimatePi=True)
    assert (b.pi, b.v, b.name, b.estimatePi) == (0.05, 0.01, "BayesB", True)
    assert ngp.Random("I", 2.0).str == "I"


def test_read_genotypes_drops_missing_columns(ngp, tmp_path):
    f = tmp_path / "g.txt"
    f.write_text("0 1 2 1\n1 nan 0 2\n2 1 1 0\n")
    M = ngp.read_genotypes(str(f))
    assert M.shape == (3, 3) and np.array_equal(M[:, 0], [0, 1, 2]) and np.array_equal(M[:, 1], [2, 0, 1])


def test_prep2RegionData(ngp, tmp_path):
    m = tmp_path / "map.csv"
    rows = ["snpID,snpOrder,chrID"] + [f"s{i},{i},{1 if i <= 5 else 2}" for i in range(1, 13)]
    m.write_text("\n".join(rows) + "\n")
    assert ngp.prep2RegionData(str(tmp_path), "M", str(m), 9999) == [(0, 12)]
    assert ngp.prep2RegionData(str(tmp_path), "M", str(m), 99) == [(0, 5), (5, 12)]
    assert ngp.prep2RegionData(str(tmp_path), "M", str(m), 3) == [(0, 3), (3, 5), (5, 8), (8, 11), (11, 12)]
    gi = (tmp_path / "groupInfo_M.txt").read_text().splitlines()
    assert gi[0].split("\t") == ["snpID", "snpOrder", "chrID", "groupID"] and len(gi) == 13


def test_summaryMCMC_is_column_mean(ngp, tmp_path):
    (tmp_path / "betaMOut").write_text("M1\tM2\n1.0\t2.0\n3.0\t6.0\n")
    assert np.allclose(ngp.summaryMCMC("betaM", outFolder=str(tmp_path)), [[2.0, 4.0]])


@pytest.mark.gpu
def test_runLMEM_end_to_end_matches_oracle(ngp, O, tmp_path):
    N, P1, P2 = 120, 70, 90
    X, y, bt, v = make_problem(O, N, P1 + P2, seed=3)
    raw = np.rint(X.astype(np.float64) + X.astype(np.float64).mean(0))  # any integer-ish genotypes; centring is redone
    X1 = O.generate_panel(N, P1 + P2, seed=5)[0]
    mu = -X1.astype(np.float64).min(axis=0)
    G = np.rint(X1.astype(np.float64) + mu)                      # raw 0/1/2 genotypes
    g1, g2 = tmp_path / "g1.txt", tmp_path / "g2.txt"
    np.savetxt(g1, G[:, :P1], fmt="%d", delimiter=" ")
    np.savetxt(g2, G[:, P1:], fmt="%d", delimiter=" ")
    out = tmp_path / "outMCMC"
    VCV = {"M1": ngp.BayesPR(9999, v), "M2": ngp.BayesB(0.1, v, estimatePi=True), "e": ngp.Random("I", 0.5 * y.var())}
    res = ngp.runLMEM(f'y ~ 1 + SNP(M1,"{g1}") + SNP(M2,"{g2}")', {"y": y}, 20, 6, 2, outFolder=str(out), VCV=VCV, seed=9, engine=(1, 3))
    assert res["nKept"] == 7
    # the same model on the oracle (reference order)
    Gc = (G - G.mean(axis=0)).astype(np.float32)
    o = O.Oracle(0, seed=9, chain=0)
    o.set_panel_f32(Gc)
    o.add_marker_set(0, P1, 0, 4.0, v * 0.5, [(0, P1)], [v])
    o.add_marker_set(P1, P2, 1, 4.0, v * 0.5, [(j, j + 1) for j in range(P2)], [v] * P2, pi0=0.1, estPi=True)
    o.set_y(y); o.set_residual_prior(4.0, 0.5 * y.var() * 0.5); o.set_schedule(20, 6, 2); o.run(20)
    ps = o.get_posterior_sums()
    assert np.abs(res["sets"]["M1"]["beta"] - ps["sum_beta"][:P1] / 7).max() < 1e-9
    assert np.array_equal(res["sets"]["M2"]["delta"], ps["sum_delta"][P1:] / 7)
    assert abs(res["varE"] - ps["sum_varE"] / 7) < 1e-9 * res["varE"]
    # files: header + one row per kept iteration; posterior mean through summaryMCMC == device sums
    for name, ncol in (("b", 1), ("varE", 1), ("betaM1", P1), ("deltaM1", P1), ("betaM2", P2), ("deltaM2", P2), ("piM2", 2), ("varM1", 1),
                       ("varM2", P2)):
        lines = (out / f"{name}Out").read_text().splitlines()
        assert len(lines) == 8 and len(lines[0].split("\t")) == ncol, name
    assert np.allclose(ngp.summaryMCMC("betaM1", outFolder=str(out))[0], res["sets"]["M1"]["beta"], rtol=0, atol=1e-12)
    assert np.allclose(ngp.summaryMCMC("varE", outFolder=str(out))[0, 0], res["varE"])
    assert (out / "betaM2Out").read_text().splitlines()[0].split("\t")[:2] == ["M1", "M2"]
    with pytest.raises(FileExistsError):
        ngp.runLMEM(f'y ~ 1 + SNP(M1,"{g1}")', {"y": y}, 2, 0, 1, outFolder=str(out), VCV=VCV)
    # The files above came from the sample STREAM (samples="text": the kept samples leave the device while the chain runs, ONE ngp_run
    # for the whole chain).  The older stop-and-copy path (one ngp_run per kept iteration) writes the same files, byte for byte.
    out2 = tmp_path / "outSync"
    res2 = ngp.runLMEM(f'y ~ 1 + SNP(M1,"{g1}") + SNP(M2,"{g2}")', {"y": y}, 20, 6, 2, outFolder=str(out2), VCV=VCV, seed=9, engine=(1, 3),
                       samples="text-sync")
    assert sorted(p.name for p in out.iterdir()) == sorted(p.name for p in out2.iterdir())
    for pth in out.iterdir():
        assert pth.read_bytes() == (out2 / pth.name).read_bytes(), pth.name
    assert res2["nKept"] == 7 and np.array_equal(res["sets"]["M1"]["beta"], res2["sets"]["M1"]["beta"])
    # the binary file itself, and a census-retried launch in the middle of the stream: the dropped records are run again, none is lost
    out3 = tmp_path / "outBin"
    ngp.runLMEM(f'y ~ 1 + SNP(M1,"{g1}") + SNP(M2,"{g2}")', {"y": y}, 20, 6, 2, outFolder=str(out3), VCV=VCV, seed=9, engine=(1, 3), samples="binary")
    S = ngp.read_sample_file(str(out3 / "samples.ngpsmp"))
    assert S["iter"].tolist() == [8, 10, 12, 14, 16, 18, 20] and S["beta"].shape == (7, P1 + P2) and S["delta"].dtype == np.uint8
    assert np.array_equal(S["beta"][:, :P1].mean(axis=0), ngp.summaryMCMC("betaM1", outFolder=str(out))[0]) or \
        np.allclose(S["beta"][:, :P1].mean(axis=0), ngp.summaryMCMC("betaM1", outFolder=str(out))[0], rtol=0, atol=1e-15)


@pytest.mark.gpu
def test_runLMEM_bayesc_matches_oracle(ngp, O, tmp_path):
    """BayesC through the reference's interface (src/runTime.jl:64-77, src/mme.jl:362-373): one variance for the set, pi file."""
    N, P = 100, 150
    X, y, bt, v = make_problem(O, N, P, seed=4)
    X1 = O.generate_panel(N, P, seed=6)[0]
    G = np.rint(X1.astype(np.float64) - X1.astype(np.float64).min(axis=0))
    g = tmp_path / "g.txt"
    np.savetxt(g, G, fmt="%d", delimiter=" ")
    out = tmp_path / "outC"
    VCV = {"M": ngp.BayesC(0.2, v, estimatePi=True), "e": ngp.Random("I", 0.5 * y.var())}
    res = ngp.runLMEM(f'y ~ 1 + SNP(M,"{g}")', {"y": y}, 16, 4, 3, outFolder=str(out), VCV=VCV, seed=5, engine=(1, 4))
    assert res["nKept"] == 4
    Gc = (G - G.mean(axis=0)).astype(np.float32)
    o = O.Oracle(0, seed=5, chain=0)
    o.set_panel_f32(Gc)
    o.add_marker_set(0, P, 2, 4.0, v * 0.5, [(0, P)], [v], pi0=0.2, estPi=True)
    o.set_y(y); o.set_residual_prior(4.0, 0.5 * y.var() * 0.5); o.set_schedule(16, 4, 3); o.run(16)
    ps = o.get_posterior_sums()
    assert np.array_equal(res["sets"]["M"]["delta"], ps["sum_delta"] / 4)
    assert np.abs(res["sets"]["M"]["beta"] - ps["sum_beta"] / 4).max() < 1e-9
    assert abs(res["sets"]["M"]["var"][0] - ps["sum_varBeta"][0] / 4) < 1e-9 * res["sets"]["M"]["var"][0]
    assert np.abs(res["sets"]["M"]["pi"] - ps["sum_pi"] / 4).max() < 1e-12
    for name, ncol in (("betaM", P), ("deltaM", P), ("piM", 2), ("varM", 1)):
        lines = (out / f"{name}Out").read_text().splitlines()
        assert len(lines) == 5 and len(lines[0].split("\t")) == ncol, name


@pytest.mark.gpu
def test_runLMEM_bayesr_matches_oracle(ngp, O, tmp_path):
    """BayesR through the reference's interface (src/runTime.jl:78-93, src/mme.jl:374-383): class file per kept iteration with one
    column per class (header pi1..piK, src/mme.jl:569-570), delta = class of a locus, posterior means against the oracle."""
    N, P = 100, 150
    X, y, bt, v = make_problem(O, N, P, seed=4)
    X1 = O.generate_panel(N, P, seed=6)[0]
    G = np.rint(X1.astype(np.float64) - X1.astype(np.float64).min(axis=0))
    g = tmp_path / "g.txt"
    np.savetxt(g, G, fmt="%d", delimiter=" ")
    out = tmp_path / "outR"
    cls, pi0 = [0.0, 0.001, 0.01, 0.1], [0.7, 0.2, 0.07, 0.03]
    VCV = {"M": ngp.BayesR(pi0, cls, v, estimatePi=True), "e": ngp.Random("I", 0.5 * y.var())}
    res = ngp.runLMEM(f'y ~ 1 + SNP(M,"{g}")', {"y": y}, 16, 4, 3, outFolder=str(out), VCV=VCV, seed=5, engine=(1, 4))
    assert res["nKept"] == 4
    Gc = (G - G.mean(axis=0)).astype(np.float32)
    o = O.Oracle(0, seed=5, chain=0)
    o.set_panel_f32(Gc)
    o.add_marker_set_r(0, P, 4.0, v * 0.5, v, cls, pi0, estPi=True)
    o.set_y(y); o.set_residual_prior(4.0, 0.5 * y.var() * 0.5); o.set_schedule(16, 4, 3); o.run(16)
    ps = o.get_posterior_sums()
    assert np.array_equal(res["sets"]["M"]["delta"], ps["sum_delta"] / 4)
    assert np.abs(res["sets"]["M"]["beta"] - ps["sum_beta"] / 4).max() < 1e-9
    assert abs(res["sets"]["M"]["var"][0] - ps["sum_varBeta"][0] / 4) < 1e-9 * res["sets"]["M"]["var"][0]
    assert np.abs(res["sets"]["M"]["pi"] - o.get_class_state(0)["sum_pi"] / 4).max() < 1e-12 and abs(res["sets"]["M"]["pi"].sum() - 1.0) < 1e-12
    for name, ncol in (("betaM", P), ("deltaM", P), ("piM", 4), ("varM", 1)):
        lines = (out / f"{name}Out").read_text().splitlines()
        assert len(lines) == 5 and len(lines[0].split("\t")) == ncol, name
    assert (out / "piMOut").read_text().splitlines()[0].split("\t") == ["pi1", "pi2", "pi3", "pi4"]
    assert set(np.loadtxt(out / "deltaMOut", skiprows=1).ravel()) <= {1.0, 2.0, 3.0, 4.0}


@pytest.mark.gpu
def test_runLMEM_covariates_and_blocks(ngp, O, tmp_path):
    """Fixed effects through the reference's interface: a Float covariate (centred), a factor (dummy coded), two terms blocked with
    blockThese (sampleb!, src/functions.jl:22-36): bOut carries every level, posterior means against the oracle."""
    N, P = 120, 100
    X, y, bt, v = make_problem(O, N, P, seed=4)
    X1 = O.generate_panel(N, P, seed=6)[0]
    G = np.rint(X1.astype(np.float64) - X1.astype(np.float64).min(axis=0))
    rng = np.random.default_rng(2)
    age = rng.normal(40, 5, N); w = rng.normal(size=N); herd = rng.choice(["a", "b", "c"], N)
    y = y + 0.3 * (age - 40) + 1.2 * (herd == "c")
    g = tmp_path / "g.txt"
    np.savetxt(g, G, fmt="%d", delimiter=" ")
    out = tmp_path / "outF"
    VCV = {"M": ngp.BayesPR(9999, v), "e": ngp.Random("I", 0.5 * y.var())}
    res = ngp.runLMEM(f'y ~ 1 + age + herd + w + SNP(M,"{g}")', {"y": y, "age": age, "herd": herd, "w": w}, 30, 10, 2, outFolder=str(out), VCV=VCV,
                      seed=5, blockThese=[("age", "w")])
    assert res["fixed_names"] == ["(Intercept)", "age", "w", "herd: b", "herd: c"] and res["nKept"] == 10
    Gc = (G - G.mean(axis=0)).astype(np.float32)
    o = O.Oracle(0, seed=5, chain=0); o.set_panel_f32(Gc)
    o.add_fixed_set(np.column_stack([age - age.mean(), w - w.mean()]))
    o.add_fixed_set(np.column_stack([(herd == "b").astype(float), (herd == "c").astype(float)]))
    o.add_marker_set(0, P, 0, 4.0, v * 0.5, [(0, P)], [v])
    o.set_y(y); o.set_residual_prior(4.0, 0.5 * y.var() * 0.5); o.set_schedule(30, 10, 2); o.run(30)
    assert np.abs(res["fixed"] - o.get_fixed()["sum_b"] / 10).max() < 1e-9
    assert abs(res["fixed"][0] - 0.3) < 0.15                      # the covariate's effect is recovered (the factor's needs a longer chain)
    lines = (out / "bOut").read_text().splitlines()
    assert len(lines) == 11 and lines[0].split("\t") == res["fixed_names"] and len(lines[1].split("\t")) == 5


def test_panel_file_round_trip_on_the_host(ngp, tmp_path):
    """Binary panel files (include/nextgp_hip.h: ngp_write_panel_file / ngp_read_panel_header; api.read_panel_file) -- host only."""
    rng = np.random.default_rng(4)
    G = rng.integers(0, 3, size=(37, 11), dtype=np.uint8)      # 37: the last byte of a two-bit column is partly empty
    for bits in (8, 2):
        p = tmp_path / f"p{bits}.bin"
        ngp.write_panel_file(p, G, bits=bits)
        assert ngp.is_panel_file(p) and ngp.read_panel_header(p) == (37, 11, bits)
        assert np.array_equal(ngp.read_panel_file(p), G)
        assert np.array_equal(ngp.read_genotypes(str(p)), G) and ngp.read_genotypes(str(p)).dtype == np.uint8
    txt = tmp_path / "g.txt"
    np.savetxt(txt, G, fmt="%d", delimiter=" ")
    assert not ngp.is_panel_file(txt)
    G2 = G.copy(); G2[3, 2] = 200
    ngp.write_panel_file(tmp_path / "q.bin", G2, bits=8)       # any byte value with 8 bits
    assert np.array_equal(ngp.read_panel_file(tmp_path / "q.bin"), G2)
    with pytest.raises(ngp.NextGPHipError):
        ngp.write_panel_file(tmp_path / "r.bin", G2, bits=2)


@pytest.mark.gpu
def test_runLMEM_compact_storage(ngp, O, tmp_path):
    """storage="u8" through the reference's interface: text codes, a uint8 array and a two-bit panel file give the same chain, and
    that chain is the fp32-storage chain up to the fp32 rounding of the panel."""
    N, P = 150, 200
    X, y, bt, v = make_problem(O, N, P, seed=4)
    X1 = O.generate_panel(N, P, seed=6)[0]
    G = np.rint(X1.astype(np.float64) - X1.astype(np.float64).min(axis=0)).astype(np.uint8)
    txt, pf = tmp_path / "g.txt", tmp_path / "g.bin"
    np.savetxt(txt, G, fmt="%d", delimiter=" ")
    ngp.write_panel_file(pf, G, bits=2)
    VCV = {"M": ngp.BayesPR(9999, v), "e": ngp.Random("I", 0.5 * y.var())}
    res = {}
    for name, src, storage in (("txt8", f'"{txt}"', "u8"), ("bin8", f'"{pf}"', "u8"), ("txt32", f'"{txt}"', None)):
        res[name] = ngp.runLMEM(f"y ~ 1 + SNP(M,{src})", {"y": y}, 16, 4, 3, outFolder=str(tmp_path / name), VCV=VCV, seed=5,
                                storage=storage, samples="none")
    assert np.array_equal(res["txt8"]["sets"]["M"]["beta"], res["bin8"]["sets"]["M"]["beta"]) and res["txt8"]["varE"] == res["bin8"]["varE"]
    a, b = res["txt8"]["sets"]["M"]["beta"], res["txt32"]["sets"]["M"]["beta"]
    assert 0 < np.abs(a - b).max() < 1e-4 * np.abs(a).max()     # fp32 rounding of the panel, amplified over 16 iterations
    frac = tmp_path / "frac.txt"
    np.savetxt(frac, G + 0.25, fmt="%.2f", delimiter=" ")
    with pytest.raises(ValueError, match="integer genotype codes"):
        ngp.runLMEM(f'y ~ 1 + SNP(M,"{frac}")', {"y": y}, 4, 1, 1, outFolder=str(tmp_path / "bad"), VCV=VCV, storage="u8")


@pytest.mark.gpu
def test_runLMEM_correlated_marker_sets(ngp, O, tmp_path):
    """A VCV key that is a tuple of set names = correlated marker sets (src/mme.jl:448-489): sampleBayesPR!(::Tuple), src/functions.jl:140-154.
    The mirror interleaves the members' columns for the device and writes beta / delta files per member, one var file for the tuple."""
    N, nloc = 150, 70
    X1 = O.generate_panel(N, 2 * nloc + 40, seed=5)[0]
    G = np.rint(X1.astype(np.float64) - X1.astype(np.float64).min(axis=0))          # raw 0/1/2 genotypes
    rng = np.random.default_rng(4)
    y = 2.0 + (G[:, 3] - G[:, 3].mean()) * 0.7 - (G[:, nloc + 3] - G[:, nloc + 3].mean()) * 0.4 + rng.normal(size=N)
    files = []
    for i, sl in enumerate((slice(0, nloc), slice(nloc, 2 * nloc), slice(2 * nloc, 2 * nloc + 40))):
        f = tmp_path / f"g{i}.txt"; np.savetxt(f, G[:, sl], fmt="%d", delimiter=" "); files.append(f)
    vm = np.array([[0.02, 0.008], [0.008, 0.03]])
    VCV = {("A", "B"): ngp.BayesPR(9999, vm), "C": ngp.BayesPR(9999, 0.01), "e": ngp.Random("I", 0.5 * y.var())}
    out = tmp_path / "out"
    res = ngp.runLMEM(f'y ~ 1 + SNP(A,"{files[0]}") + SNP(B,"{files[1]}") + SNP(C,"{files[2]}")', {"y": y}, 16, 4, 2, outFolder=str(out), VCV=VCV, seed=6)
    assert res["nKept"] == 6 and res["sets"]["A"]["var"].shape == (1, 2, 2)
    # the same model on the reference-order oracle: members interleaved from column 0, the plain set behind the tuple's last block
    Gc = G - G.mean(axis=0)
    Xt = ngp.tuple_panel([np.asfortranarray(Gc[:, :nloc]), np.asfortranarray(Gc[:, nloc:2 * nloc])])
    off = -(-Xt.shape[1] // 64) * 64
    Xp = np.asfortranarray(np.hstack([Xt, np.zeros((N, off - Xt.shape[1])), Gc[:, 2 * nloc:]]).astype(np.float32))
    o = O.Oracle(0, seed=6, chain=0)
    o.set_panel_f32(Xp)
    o.add_marker_set_tuple(0, nloc, 2, 5.0, vm * 2.0, [(0, nloc)], vm)
    o.add_marker_set(off, 40, 0, 4.0, 0.01 * 0.5, [(0, 40)], [0.01])
    o.set_y(y); o.set_residual_prior(4.0, 0.5 * y.var() * 0.5); o.set_schedule(16, 4, 2); o.run(16)
    ps = o.get_posterior_sums()
    cols = ngp.tuple_columns(0, nloc, 2)
    for m, nm in enumerate(("A", "B")):
        assert np.abs(res["sets"][nm]["beta"] - ps["sum_beta"][cols[:, m]] / 6).max() < 1e-9
    assert np.abs(res["sets"]["C"]["beta"] - ps["sum_beta"][off:off + 40] / 6).max() < 1e-9
    assert np.allclose(res["sets"]["A"]["var"].ravel(), ps["sum_varBeta"][:4] / 6, rtol=1e-8)
    for name, ncol in (("betaA", nloc), ("betaB", nloc), ("deltaA", nloc), ("betaC", 40), ("varA_B", 4), ("varC", 1)):
        lines = (out / f"{name}Out").read_text().splitlines()
        assert len(lines) == 7 and len(lines[0].split("\t")) == ncol, name
    assert (out / "varA_BOut").read_text().splitlines()[0].split("\t") == ["reg_1_11", "reg_1_12", "reg_1_21", "reg_1_22"]
    assert np.allclose(ngp.summaryMCMC("betaB", outFolder=str(out))[0], res["sets"]["B"]["beta"], rtol=0, atol=1e-12)


@pytest.mark.gpu
def test_runLMEM_with_several_chains_over_one_panel(ngp, O, tmp_path):
    """runLMEM(..., chains=K): K independent chains (chain ids 0..K-1) over ONE copy of the panel on the device, fused into one sweep
    launch per iteration; every chain writes its own *Out files, bit for bit the files of the same chain run alone with that layout,
    and the returned means are the chains' pooled means."""
    N, P1, P2 = 900, 330, 310
    X1 = O.generate_panel(N, P1 + P2, seed=5)[0]
    mu = -X1.astype(np.float64).min(axis=0)
    G = np.rint(X1.astype(np.float64) + mu)
    rng = np.random.default_rng(3)
    bt = np.zeros(P1 + P2); bt[rng.choice(P1 + P2, 12, replace=False)] = rng.normal(size=12)
    y = 4.0 + (G - G.mean(0)) @ bt + rng.normal(size=N)
    v = 0.01
    g1, g2 = tmp_path / "g1.txt", tmp_path / "g2.txt"
    np.savetxt(g1, G[:, :P1], fmt="%d", delimiter=" ")
    np.savetxt(g2, G[:, P1:], fmt="%d", delimiter=" ")
    VCV = {"M1": ngp.BayesPR(9999, v), "M2": ngp.BayesB(0.1, v, estimatePi=True), "e": ngp.Random("I", 0.5 * y.var())}
    f = f'y ~ 1 + SNP(M1,"{g1}") + SNP(M2,"{g2}")'
    K = 3
    out = tmp_path / "outChains"
    res = ngp.runLMEM(f, {"y": y}, 24, 6, 3, outFolder=str(out), VCV=VCV, seed=9, chains=K)
    assert len(res["chains"]) == K and res["nKept"] == K * 6
    first = res["samplers"][0]
    R, S, _ = first.layout()
    assert first.census()["grid"] == K + K * ((S + 31) // 32) + S          # ONE fused launch per iteration served the three chains
    # pooled means = mean of the chains' means
    for nm in ("M1", "M2"):
        assert np.allclose(res["sets"][nm]["beta"], np.mean([c["sets"][nm]["beta"] for c in res["chains"]], axis=0), rtol=0, atol=1e-15)
    assert abs(res["varE"] - np.mean([c["varE"] for c in res["chains"]])) < 1e-12
    # every chain's files == the files of that chain run alone (same seed, its chain id, the layout of the fused run)
    for c in (0, K - 1):
        alone = tmp_path / f"alone{c}"
        r1 = ngp.runLMEM(f, {"y": y}, 24, 6, 3, outFolder=str(alone), VCV=VCV, seed=9, chain=c, max_shards=S)
        assert r1["sampler"].layout()[:2] == (R, S)
        cdir = out / f"chain{c}"
        assert sorted(p.name for p in cdir.iterdir()) == sorted(p.name for p in alone.iterdir())
        for pth in alone.iterdir():
            assert pth.read_bytes() == (cdir / pth.name).read_bytes(), (c, pth.name)
        assert np.array_equal(r1["sets"]["M2"]["delta"], res["chains"][c]["sets"]["M2"]["delta"])
    assert res["fused"] is True
    # the same model over genotype codes (compact storage), three chains: fused as well, each chain the chain alone
    outc = tmp_path / "outChainsU8"
    rc = ngp.runLMEM(f, {"y": y}, 12, 3, 3, outFolder=str(outc), VCV=VCV, seed=9, chains=3, storage="u8", samples="none")
    assert rc["fused"] is True and rc["samplers"][0].storage() == 1 and rc["samplers"][0].census()["retries"] == 0
    Sc = rc["samplers"][0].layout()[1]
    r2 = ngp.runLMEM(f, {"y": y}, 12, 3, 3, outFolder=str(tmp_path / "aloneU8"), VCV=VCV, seed=9, chain=2, max_shards=Sc, storage="u8", samples="none")
    assert np.array_equal(r2["sets"]["M1"]["beta"], rc["chains"][2]["sets"]["M1"]["beta"]) and r2["varE"] == rc["chains"][2]["varE"]
    with pytest.raises(ValueError):
        ngp.runLMEM(f, {"y": y}, 4, 0, 1, outFolder=str(tmp_path / "bad"), VCV=VCV, chains=2, samples="text-sync")
