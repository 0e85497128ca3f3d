"""Host-side mirror of the NextGP.jl user interface for the accelerated path.

Same names, argument meaning and output files as the reference, so a model script ports line by line:

    reference (Julia)                                   this module (Python stand-in for the Julia shim)
    ------------------------------------------------    ------------------------------------------------
    runLMEM(f, data, nChain, nBurn, nThin; VCV, ...)    runLMEM(f, data, nChain, nBurn, nThin, VCV=..., ...)   src/MCMC.jl:31-41
    BayesPR(r, v) / BayesB(pi, v; estimatePi)           BayesPR(r, v) / BayesB(pi, v, estimatePi=...)          src/runTime.jl:30-61
    Random("I", v)          (residual prior, key :e)    Random("I", v)                (key "e")                 src/runTime.jl:135-146
    SNP(M, "geno.txt"[, "map.txt"]) in the formula      the same text inside the formula string                src/runTime.jl:13-28
    summaryMCMC("betaM"; outFolder)                     summaryMCMC("betaM", outFolder=...)                    src/misc.jl:241-244

Interpreted here: the response, the intercept `1`, covariate / factor columns (optionally grouped by `blockThese`) and `SNP(...)`
terms.  Interactions, `PED(...)`, `(1|g)` terms, GBLUP priors, BayesRC/LV are outside the accelerated
path and raise NotImplementedError naming the reference code that handles them.
All arithmetic happens in libnextgp_hip.so; this file only parses, reshapes and writes files.
"""
import os
import re
from dataclasses import dataclass
from typing import Optional

import numpy as np

from ._lib import METHOD_BAYESB, METHOD_BAYESC, METHOD_BAYESPR, Sampler, tuple_columns, tuple_panel, tuple_span

__all__ = ["BayesPR", "BayesB", "BayesC", "BayesR", "Random", "SNP", "runLMEM", "summaryMCMC", "read_genotypes", "read_panel_file", "is_panel_file", "prep2RegionData", "parse_formula", "design_columns", "samples_to_out_files"]


# ----------------------------------------------------------------------------------------------
# prior / term types (src/runTime.jl)
# ----------------------------------------------------------------------------------------------
@dataclass
class BayesPRType:  # src/runTime.jl:30-45
    r: int
    v: object       # a variance, or -- for correlated marker sets, VCV key (:M1, :M2) -- their k x k covariance matrix (src/mme.jl:493-516)
    name: str = "BayesPR"


@dataclass
class BayesBType:  # src/runTime.jl:48-61
    pi: float
    v: float
    name: str = "BayesB"
    estimatePi: bool = False


@dataclass
class BayesCType:  # src/runTime.jl:64-77
    pi: float
    v: float
    name: str = "BayesC"
    estimatePi: bool = False


@dataclass
class BayesRType:  # src/runTime.jl:78-93
    pi: object      # class probabilities (one per class)
    class_: object  # variance-class multipliers (the reference's field is `class`, a Python keyword)
    v: float
    name: str = "BayesR"
    estimatePi: bool = False


@dataclass
class RandomEffectType:  # src/runTime.jl:135-146
    str: object
    v: float
    type: int = 1


@dataclass
class GenomicTerm:  # src/runTime.jl:13-28
    name: str
    path: object
    map: str = ""


def BayesPR(r, v, name="BayesPR"):
    return BayesPRType(int(r), float(v) if np.ndim(v) == 0 else np.asarray(v, dtype=np.float64), name)


def BayesB(pi, v, name="BayesB", estimatePi=False):
    return BayesBType(float(pi), float(v), name, bool(estimatePi))


def BayesC(pi, v, name="BayesC", estimatePi=False):
    return BayesCType(float(pi), float(v), name, bool(estimatePi))


def BayesR(pi, class_, v, name="BayesR", estimatePi=False):
    """BayesR(pi, class, v; estimatePi) of src/runTime.jl:87-93: `class_` = variance-class multipliers, `pi` their probabilities."""
    return BayesRType([float(x) for x in pi], [float(x) for x in class_], float(v), name, bool(estimatePi))


def Random(str, v, type=1):
    return RandomEffectType(str, float(v), type)


def SNP(name, path, map=""):
    return GenomicTerm(name, path, map)


# ----------------------------------------------------------------------------------------------
# formula (the subset of the StatsModels DSL the marker path uses, src/prepMatVec.jl:112-169)
# ----------------------------------------------------------------------------------------------
class ParsedFormula(tuple):
    """(lhs, intercept, [GenomicTerm, ...]) with the plain covariate / factor terms as `.covariates` -- the parse result carries
    everything itself (no state is left on the function: two models parsed in turn, or from threads, cannot pick up each other's
    covariates)."""

    def __new__(cls, lhs, intercept, snps, covariates):
        t = super().__new__(cls, (lhs, intercept, snps))
        t.covariates = list(covariates)
        return t


def parse_formula(formula):
    """'y ~ 1 + x + SNP(M, "geno.txt", "map.txt")' -> ParsedFormula (lhs, intercept, [GenomicTerm, ...]; .covariates = ['x'])."""
    if "~" not in formula:
        raise ValueError("formula needs a '~'")
    lhs, rhs = [t.strip() for t in formula.split("~", 1)]
    terms, depth, cur = [], 0, ""
    for ch in rhs:
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == "+" and depth == 0:
            terms.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        terms.append(cur.strip())
    intercept, snps, covs = False, [], []
    for t in terms:
        if t == "1":
            intercept = True
        elif t == "0":
            intercept = False
        elif t.startswith("SNP(") and t.endswith(")"):
            args = [a.strip() for a in re.split(r",(?=(?:[^\"]*\"[^\"]*\")*[^\"]*$)", t[4:-1])]
            if len(args) < 2:
                raise ValueError(f"SNP term needs a name and a path: {t}")
            unq = [a.strip("\"'") for a in args]
            snps.append(GenomicTerm(unq[0], unq[1], unq[2] if len(unq) > 2 else ""))
        elif t.startswith("PED(") or "|" in t:
            raise NotImplementedError(f"term '{t}': pedigree / (1|g) random effects stay on the reference's Julia path "
                                      "(src/functions.jl:57-110, src/mme.jl:165-272); they are outside the accelerated sweep")
        elif re.fullmatch(r"[A-Za-z_][A-Za-z_0-9]*", t):
            covs.append(t)  # a column of the data: covariate or factor (src/prepMatVec.jl:150-165)
        else:
            raise NotImplementedError(f"term '{t}': interactions / function terms stay on the reference's Julia path (StatsModels, "
                                      "src/prepMatVec.jl:150-165); use the fine seam (ngp_sweep_set) to combine them with the GPU sweep")
    return ParsedFormula(lhs, intercept, snps, covs)


def design_columns(name, col):
    """One model term -> (N x k design, level names): Float columns are centred, Int / String columns dummy coded with the first
    level as the base (the reference's rules, src/prepMatVec.jl docstring :33-37 and :46-60, through StatsModels)."""
    col = np.asarray(col)
    if col.dtype.kind == "f":
        x = col.astype(np.float64)
        return (x - x.mean())[:, None], [name]
    levels = sorted(set(col.tolist()))
    if len(levels) < 2:
        raise ValueError(f"factor {name} has a single level")
    X = np.column_stack([(col == lv).astype(np.float64) for lv in levels[1:]])
    return X, [f"{name}: {lv}" for lv in levels[1:]]


# ----------------------------------------------------------------------------------------------
# marker-matrix builders (src/prepMatVec.jl:113-134, src/misc.jl:163-215)
# ----------------------------------------------------------------------------------------------
def read_genotypes(path):
    """Space-delimited text, one row per individual, no header (prepMatVec.jl:116); columns holding a missing
    value are dropped (prepMatVec.jl:118).  Returns a float64 (N, P) Fortran-ordered matrix, NOT centred.
    Beyond the reference: an (N, P) array, or a `.npy` file (memory-mapped), is taken as it is -- uint8 allele counts stay
    uint8 (one byte per genotype; at 50k x 600k the text parse alone would take hours)."""
    if isinstance(path, (str, os.PathLike)) and str(path).endswith(".npy"):
        path = np.load(path, mmap_mode="r")
    elif isinstance(path, (str, os.PathLike)) and is_panel_file(path):
        return read_panel_file(path)
    if isinstance(path, np.ndarray):
        if path.dtype == np.uint8:
            return np.asfortranarray(path)
        M = np.asarray(path, dtype=np.float64)
    else:
        M = np.genfromtxt(path, delimiter=" ", dtype=np.float64)
        if M.ndim == 1:
            M = M[:, None]
    keep = ~np.isnan(M).any(axis=0)
    return np.asfortranarray(M[:, keep])


def is_panel_file(path):
    """True for the binary panel format of include/nextgp_hip.h (magic NGPPNL01)."""
    try:
        with open(path, "rb") as f:
            return f.read(8) == b"NGPPNL01"
    except OSError:
        return False


def read_panel_file(path):
    """Binary panel file -> (N, P) uint8 Fortran-ordered codes (host side; Sampler.load_panel_file streams the same file
    straight to the device)."""
    with open(path, "rb") as f:
        hd = f.read(32)
        if hd[:8] != b"NGPPNL01":
            raise ValueError(f"not a panel file: {path}")
        N, P = np.frombuffer(hd, dtype="<i8", count=2, offset=8)
        bits = int(np.frombuffer(hd, dtype="<i4", count=1, offset=24)[0])
        N, P = int(N), int(P)
        if bits == 8:
            G = np.fromfile(f, dtype=np.uint8, count=N * P).reshape((N, P), order="F")
        elif bits == 2:
            nb = (N + 3) // 4
            raw = np.fromfile(f, dtype=np.uint8, count=nb * P).reshape((nb, P), order="F")
            G = np.empty((4 * nb, P), dtype=np.uint8, order="F")
            for k in range(4):
                G[k::4, :] = (raw >> (2 * k)) & 3
            G = np.asfortranarray(G[:N, :])
            if (G == 3).any():
                raise ValueError("panel file holds a missing genotype (code 3): impute before loading")
        else:
            raise ValueError(f"panel file with {bits} bits per genotype")
    return G


def prep2RegionData(outPutFolder, markerSet, mapFile, fixedRegSize):
    """Region ranges from a map file with header snpID,snpOrder,chrID (src/misc.jl:163-215).
    99 = one region per chromosome, 9999 = whole genome, anything else = windows of that many SNPs inside each
    chromosome.  Writes groupInfo_<set>.txt like the reference and returns 0-based [start, stop) pairs."""
    import csv
    with open(mapFile, newline="") as f:
        rows = list(csv.DictReader(f))
    chrs = [r["chrID"] for r in rows]
    groups, g = [], 0
    if fixedRegSize == 9999:
        groups = [1] * len(rows)
    else:
        order = []
        for c in chrs:
            if c not in order:
                order.append(c)
        gid = {}
        for c in order:
            idx = [i for i, x in enumerate(chrs) if x == c]
            if fixedRegSize == 99:
                g += 1
                for i in idx:
                    gid[i] = g
            else:
                for k, i in enumerate(idx):
                    gid[i] = g + 1 + k // int(fixedRegSize)
                g += (len(idx) + int(fixedRegSize) - 1) // int(fixedRegSize)
        groups = [gid[i] for i in range(len(rows))]
    if outPutFolder is not None:
        with open(os.path.join(outPutFolder, f"groupInfo_{markerSet}.txt"), "w") as f:
            f.write("snpID\tsnpOrder\tchrID\tgroupID\n")
            for r, gg in zip(rows, groups):
                f.write(f"{r['snpID']}\t{r['snpOrder']}\t{r['chrID']}\t{gg}\n")
    regions, start = [], 0
    for i in range(1, len(groups) + 1):
        if i == len(groups) or groups[i] != groups[start]:
            regions.append((start, i))
            start = i
    return regions


def _regions_for(prior, P, map_path, out_folder, set_name):
    """M[set][:regionArray] (src/mme.jl:324-358)."""
    if isinstance(prior, BayesBType):
        return [(j, j + 1) for j in range(P)]
    if isinstance(prior, (BayesCType, BayesRType)):  # one variance for the set (nVarCov = 1, src/mme.jl:370, :381)
        return [(0, P)]
    if not map_path:
        if prior.r == 1:
            return [(j, j + 1) for j in range(P)]
        if prior.r == 9999:
            return [(0, P)]
        raise ValueError("Please enter a valid region size (1 or 9999)")  # src/mme.jl:343
    return prep2RegionData(out_folder, set_name, map_path, prior.r)


# ----------------------------------------------------------------------------------------------
# output files (src/outFiles.jl:17-21, headers src/mme.jl:545-595, rows src/samplers.jl:57-103)
# ----------------------------------------------------------------------------------------------
def _out(folder, name, row):
    with open(os.path.join(folder, f"{name}Out"), "a") as f:
        f.write("\t".join(row) + "\n")


def _var_names(s):
    """Header of var<set>Out: reg_r (src/mme.jl:593-595); for correlated sets one column per entry of the region's k x k matrix."""
    if s["k"] > 1:
        return [f"reg_{r + 1}_{a + 1}{b + 1}" for r in range(s["nreg"]) for a in range(s["k"]) for b in range(s["k"])]
    return [f"reg_{r + 1}" for r in range(s["nvb"])]


def _fmt(x):
    return [repr(float(v)) for v in np.atleast_1d(x)]


def summaryMCMC(param, outFolder=None):
    """Posterior mean = column mean of <outFolder>/<param>Out, first row is the header (src/misc.jl:241-244)."""
    outFolder = outFolder or os.path.join(os.getcwd(), "outMCMC")
    return np.loadtxt(os.path.join(outFolder, f"{param}Out"), delimiter="\t", skiprows=1, ndmin=2).mean(axis=0, keepdims=True)


def samples_to_out_files(sample_path, outFolder, sets, intercept, has_fixed):
    """Binary sample stream (ngp_set_sample_file) -> the rows of the reference's *Out text files (src/samplers.jl:56-104), appended
    behind the header rows: the same text, number for number, as writing them at every kept iteration.  Record by record -- one
    kept iteration in memory at a time, every *Out file open for appending -- so a long chain of a large model converts in
    constant memory and an interrupted conversion leaves the rows written so far (the caller keeps the binary file until the
    conversion has succeeded)."""
    from ._lib import iter_sample_file
    handles = {}

    def row(name, fields):
        f = handles.get(name)
        if f is None:
            f = handles[name] = open(os.path.join(outFolder, f"{name}Out"), "a")
        f.write("\t".join(fields) + "\n")

    n = 0
    try:
        for S in iter_sample_file(sample_path):
            n += 1
            row("b", (_fmt(S["b"]) if intercept else []) + (_fmt(S["b_fixed"]) if has_fixed else []))
            row("varE", _fmt(S["varE"]))
            vb_off, cls_off = 0, 0
            for k, s in enumerate(sets):
                K = len(s["prior"].pi) if isinstance(s["prior"], BayesRType) else 0
                for m, nm in enumerate(s["members"]):
                    row(f"beta{nm}", _fmt(S["beta"][s["cols"][:, m]]))
                    row(f"delta{nm}", [str(int(v)) for v in S["delta"][s["cols"][:, m]]])
                if isinstance(s["prior"], (BayesBType, BayesCType)):
                    row(f"pi{s['name']}", _fmt(S["piHat"][2 * k:2 * k + 2]))
                if K:
                    row(f"pi{s['name']}", _fmt(S["class_pi"][cls_off:cls_off + K]))
                row(f"var{s['name']}", _fmt(S["varBeta"][vb_off:vb_off + s["nvb"]]))
                vb_off += s["nvb"]
                cls_off += K
    finally:
        for f in handles.values():
            f.close()
    return n


# ----------------------------------------------------------------------------------------------
# runLMEM (src/MCMC.jl:31-41)
# ----------------------------------------------------------------------------------------------
def runLMEM(formula, userData, nChain, nBurn, nThin, myHints=None, blockThese=None, outFolder="outMCMC", VCV=None, userPedData=None,
            summaryStat=None, seed=1, chain=0, device=0, samples="text", overwrite=False, engine=None, storage=None, chains=1, max_shards=None):
    """Runs the chain on the GPU and writes the reference's *Out files.

    Differences from the reference, all deliberate: (1) `seed`/`chain` key the random streams (the reference never
    seeds); (2) an existing non-empty outFolder is refused unless overwrite=True (the reference deletes it,
    src/misc.jl:221-227); (3) samples="none" skips the per-iteration text rows and only returns posterior means;
    (4) storage="u8" keeps the panel one byte per genotype on the device, centred analytically (ngp_set_storage: a quarter of
    the memory, no fp32 rounding of the panel) -- every SNP set must then hold integer codes 0..255 (uint8 arrays, binary panel
    files, or text files whose values are such integers); (5) chains=K runs K independent chains (chain ids chain .. chain+K-1, the
    same seed) over ONE copy of the panel on the device -- one fused sweep launch per iteration where the engine serves it (the result's
    "fused" says whether it did; a warning otherwise)
    (ngp_share_panel + ngp_run_many), side by side otherwise; every chain is bit for bit the chain it is alone with that layout,
    writes its own *Out files to outFolder/chain<c>/, and the returned means are pooled over the chains (res["chains"] holds each).
    Returns a dict of posterior means taken from the on-device sums."""
    if userPedData is not None and len(userPedData):
        raise NotImplementedError("userPedData: pedigree effects stay on the Julia path (src/mme.jl:26-46)")
    VCV = dict(VCV or {})
    summaryStat = dict(summaryStat or {})
    parsed = parse_formula(formula)
    lhs, intercept, snps = parsed
    if not snps:
        raise ValueError("the accelerated path needs at least one SNP(...) term")
    y = np.asarray(userData[lhs], dtype=np.float64)
    # folderHandler (src/misc.jl:221-232) -- without the silent rm -r
    if os.path.isdir(outFolder) and os.listdir(outFolder):
        if not overwrite:
            raise FileExistsError(f"output folder {outFolder} exists and is not empty (pass overwrite=True to clear it)")
        for fn in os.listdir(outFolder):
            os.remove(os.path.join(outFolder, fn))
    os.makedirs(outFolder, exist_ok=True)
    # prep: SNP branch (src/prepMatVec.jl:113-134); marker sets become consecutive column ranges of ONE panel
    mats = [read_genotypes(t.path) for t in snps]
    for M in mats:
        if M.shape[0] != len(y):
            raise ValueError("genotype rows must match the phenotype records (marker files are ordered as the data, runTime.jl:23)")
    if storage in ("u8", 1):  # codes stay codes: integer-valued float input is converted, anything else refused
        conv = []
        for M in mats:
            if M.dtype != np.uint8:
                if not (np.all(M == np.rint(M)) and M.min() >= 0 and M.max() <= 255):
                    raise ValueError('storage="u8" needs integer genotype codes in 0..255 in every SNP set')
                M = np.asfortranarray(M.astype(np.uint8))
            conv.append(M)
        mats = conv
    if not all(M.dtype == np.uint8 for M in mats):  # one byte per genotype only when every set comes that way
        mats = [np.asarray(M, dtype=np.float64) for M in mats]
    # Correlated marker sets: a VCV key that is a TUPLE of set names (src/mme.jl:448-489) joins those sets into one unit whose loci
    # draw their k effects together (src/functions.jl:140-154).  On the device the k columns of a locus sit side by side, from a
    # 64-column boundary on (ngp_add_marker_set_tuple): the panel is assembled accordingly, everything else in formula order.
    by_name = {t.name: i for i, t in enumerate(snps)}
    tuples = [key for key in VCV if isinstance(key, tuple)]
    in_tuple = {}
    for key in tuples:
        if not all(nm in by_name for nm in key):
            raise ValueError(f"correlated marker sets {key}: every member needs its SNP(...) term")
        if len({mats[by_name[nm]].shape[1] for nm in key}) != 1:
            raise ValueError("correlated marker sets must have the same loci (src/mme.jl:453)")
        if len({snps[by_name[nm]].map for nm in key}) != 1:
            raise ValueError("correlated marker sets must have the same map file!")          # src/mme.jl:453
        if any(nm in summaryStat for nm in key):
            raise ValueError("Not available to use summary statistics in correlated effects")  # src/mme.jl:469
        for nm in key:
            in_tuple[nm] = key
    units, pieces, ncols = [], [], 0          # units: ("set", name) | ("tuple", key), in formula order of their first member
    done_t = set()
    for t in snps:
        if t.name in in_tuple:
            key = in_tuple[t.name]
            if key in done_t:
                continue
            done_t.add(key)
            pad = (-ncols) % 64
            if pad:
                pieces.append(np.zeros((len(y), pad), dtype=mats[0].dtype)); ncols += pad
            blk = tuple_panel([np.asfortranarray(mats[by_name[nm]]) for nm in key])
            nloc = mats[by_name[key[0]]].shape[1]
            units.append(("tuple", key, ncols, nloc))
            pieces.append(blk); ncols += blk.shape[1]
            pad = (-ncols) % 64                  # the set owns its blocks to the end of the last one
            if pad:
                pieces.append(np.zeros((len(y), pad), dtype=mats[0].dtype)); ncols += pad
        else:
            units.append(("set", t.name, ncols, mats[by_name[t.name]].shape[1]))
            pieces.append(mats[by_name[t.name]]); ncols += mats[by_name[t.name]].shape[1]
    K = int(chains)
    if K < 1 or K > 8:
        raise ValueError("chains: 1..8")
    if K > 1 and samples == "text-sync":
        raise ValueError('chains > 1: samples "text", "binary" or "none"')
    skw = dict(mode=engine[0], lag=engine[1]) if engine else {}
    smp = Sampler(device=device, seed=seed, chain=chain, storage=storage, **skw)
    if max_shards:  # an explicit shard count (ngp_set_max_shards): e.g. the layout of a fused run, to repeat one of its chains alone
        smp.set_max_shards(int(max_shards))
    elif K > 1:  # the layout with which K chains share one fused sweep launch (fp32 tiles), or the device side by side
        # (compact storage: the fused kernel serves two or three chains; more run side by side on disjoint CU shares)
        smp.set_max_shards(smp.shards_for_pass(K) if (storage is None or K <= 3) else smp.shards_for_chains(K))
    kinds = {np.asarray(pc).dtype == np.uint8 for pc in pieces if pc.shape[1]}
    if (storage is None and kinds == {False}) or kinds == {True}:
        # the sets go to the device one after another (ngp_begin_panel / ngp_panel_columns_* / ngp_end_panel): no concatenated host copy
        smp.begin_panel(len(y), ncols)
        c0 = 0
        for pc in pieces:
            if pc.shape[1] and np.any(pc):       # (padding columns stay the zero columns they are born as)
                smp.panel_columns(c0, pc, centre=True)  # centring: src/prepMatVec.jl:129
            c0 += pc.shape[1]
        smp.end_panel()
    else:
        smp.set_panel(np.asfortranarray(np.concatenate(pieces, axis=1)), centre=True)  # centring: src/prepMatVec.jl:129
    samplers = [smp]
    for c in range(1, K):  # the other chains share the first one's panel by reference (no copy, no second upload)
        sc = Sampler(device=device, seed=seed, chain=chain + c, storage=storage, **skw)
        sc.share_panel(smp)
        samplers.append(sc)
    region_cache = {}

    def regions_of(prior, P, map_path, name):
        if name not in region_cache:
            region_cache[name] = _regions_for(prior, P, map_path, outFolder, name)
        return region_cache[name]

    built = [_build_model(sc, VCV, summaryStat, parsed, userData, blockThese, intercept, units, snps, by_name, regions_of, y, nChain, nBurn, nThin)
             for sc in samplers]
    sets, fixed_names = built[0]
    folders = [outFolder] if K == 1 else [os.path.join(outFolder, f"chain{chain + c}") for c in range(K)]
    for f in folders:
        os.makedirs(f, exist_ok=True)
    return _run_model(samplers, folders, sets, fixed_names, intercept, nChain, nBurn, nThin, samples)


def _build_model(smp, VCV, summaryStat, parsed, userData, blockThese, intercept, units, snps, by_name, regions_of, y, nChain, nBurn, nThin):
    """Priors, fixed-effect sets, marker sets, y and the schedule of ONE chain's handle (its panel is set); returns (sets, fixed_names)."""
    # residual prior (src/mme.jl:63-94)
    e_prior = VCV.get("e", Random("I", 100.0))
    if not (e_prior.str in ("I", "", None) or (isinstance(e_prior.str, (list, tuple)) and len(e_prior.str) == 0)):
        raise NotImplementedError("weighted residuals (E.str == \"D\") stay on the Julia path (src/mme.jl:71-75)")
    e_df = 4.0
    e_scale = 0.0005 if e_prior.v == 0.0 else e_prior.v * (e_df - 2.0) / e_df
    smp.set_residual_prior(e_df, e_scale)
    smp.set_intercept(intercept)
    # fixed effects beyond the intercept (src/prepMatVec.jl:150-165, blocks src/mme.jl:96-108): every term its own set, the terms of a
    # blockThese group one multi-column set (sampleb!, src/functions.jl:22-36); blocks first, in the user's order, then the rest in
    # model order (the reference walks a Julia Dict, whose order is not defined: documented difference)
    covs = list(parsed.covariates)
    fixed_names = ["(Intercept)"] if intercept else []
    designs = {c: design_columns(c, userData[c]) for c in covs}
    used = set()
    for blk in (blockThese or []):
        blk = [c for c in covs if c in set(blk)]          # columns inside a block keep the model's order
        if not blk:
            continue
        smp.add_fixed_set(np.column_stack([designs[c][0] for c in blk]))
        for c in blk:
            fixed_names += designs[c][1]
            used.add(c)
    for c in covs:
        if c in used:
            continue
        Xc, names = designs[c]
        if c in summaryStat and Xc.shape[1] == 1:         # src/mme.jl:140-147
            m, vv = float(np.atleast_1d(summaryStat[c][0])[0]), float(np.atleast_1d(summaryStat[c][1])[0])
            smp.add_fixed_set(Xc, lhs0=[1.0 / vv], rhs0=[m / vv])
        else:
            smp.add_fixed_set(Xc)
        fixed_names += names
    # marker sets (src/mme.jl:287-347, 492-520; correlated sets :448-489)
    sets = []
    for unit in units:
        if unit[0] == "tuple":
            _, key, col0, nloc = unit
            prior = VCV[key]
            k = len(key)
            if not isinstance(prior, BayesPRType) or np.shape(prior.v) != (k, k):
                raise NotImplementedError(f"correlated marker sets {key}: BayesPR(r, v) with v the {k} x {k} covariance matrix (src/mme.jl:448-489)")
            df = 3.0 + k                                                   # src/mme.jl:493
            vm = np.asarray(prior.v, dtype=np.float64)
            scale = vm * (df - k - 1.0) if k > 1 else vm * (df - 2.0) / df  # src/mme.jl:501
            t0 = snps[by_name[key[0]]]
            regions = regions_of(prior, nloc, t0.map, "_".join(key))
            sid = smp.add_marker_set_tuple(col0, nloc, k, df, scale, regions, vm)
            smp.set_chain_form(1)   # a model with correlated sets: block chains in the inverse form (its Tuple blocks 3.8 -> 3.0 us)
            cols = tuple_columns(col0, nloc, k)
            sets.append(dict(id=sid, name="_".join(key), members=list(key), cols=cols, P=nloc, prior=prior, nreg=len(regions), nvb=len(regions) * k * k, k=k))
            continue
        _, name, col0, P = unit
        t = snps[by_name[name]]
        prior = VCV.get(t.name)
        if prior is None:  # src/mme.jl:324-329, 504, 518
            prior = BayesPR(9999, 0.05)
        if not isinstance(prior, (BayesPRType, BayesBType, BayesCType, BayesRType)):
            raise NotImplementedError(f"prior {type(prior).__name__} for {t.name}: only BayesPR, BayesB, BayesC and BayesR are on the accelerated path")
        df = 4.0                                  # 3 + size(v,1), src/mme.jl:493
        scale = prior.v * (df - 2.0) / df         # src/mme.jl:501
        regions = regions_of(prior, P, t.map, t.name)
        lhs0 = rhs0 = None
        if t.name in summaryStat:                 # src/mme.jl:316-322
            m, v = np.asarray(summaryStat[t.name][0], float), np.asarray(summaryStat[t.name][1], float)
            lhs0 = np.where(np.isinf(1.0 / v), 0.0, 1.0 / v)
            rhs0 = np.nan_to_num(lhs0 * m)
        if isinstance(prior, BayesBType):
            sid = smp.add_marker_set(col0, P, METHOD_BAYESB, df, scale, regions, [prior.v] * P, pi0=prior.pi, estPi=prior.estimatePi,
                                     lhs0=lhs0, rhs0=rhs0)
        elif isinstance(prior, BayesCType):
            sid = smp.add_marker_set(col0, P, METHOD_BAYESC, df, scale, regions, [prior.v], pi0=prior.pi, estPi=prior.estimatePi,
                                     lhs0=lhs0, rhs0=rhs0)
        elif isinstance(prior, BayesRType):  # src/mme.jl:374-383
            sid = smp.add_marker_set_r(col0, P, df, scale, prior.v, prior.class_, prior.pi, estPi=prior.estimatePi, lhs0=lhs0, rhs0=rhs0)
        else:
            sid = smp.add_marker_set(col0, P, METHOD_BAYESPR, df, scale, regions, [prior.v] * len(regions), lhs0=lhs0, rhs0=rhs0)
        sets.append(dict(id=sid, name=t.name, members=[t.name], cols=np.arange(col0, col0 + P)[:, None], P=P, prior=prior, nreg=len(regions),
                         nvb=P if isinstance(prior, BayesBType) else len(regions), k=1))
    smp.set_y(y)
    smp.set_schedule(nChain, nBurn, nThin)
    return sets, fixed_names


def _run_model(samplers, folders, sets, fixed_names, intercept, nChain, nBurn, nThin, samples):
    """Header rows, the chain(s), the *Out rows and the posterior means (pooled over the chains when there are several)."""
    if samples not in ("text", "text-sync", "binary", "none"):
        raise ValueError('samples: "text", "text-sync", "binary" or "none"')
    if len(samplers) > 1:
        paths = [os.path.join(f, "samples.ngpsmp") for f in folders]
        for sc, f, pth in zip(samplers, folders, paths):
            if samples == "text":
                _write_headers(f, sets, fixed_names)
            if samples in ("text", "binary"):
                sc.set_sample_file(pth)
        samplers[0].get_timing()
        Sampler.run_many(samplers, nChain)  # ONE fused sweep launch per iteration for all chains where the engine serves it
        fused = samplers[0].get_timing()["sweep_launches"] == nChain
        if not fused:  # (the layout was chosen for a fused launch: side by side each chain's grid takes most of the device, so they take turns)
            import warnings
            warnings.warn(f"runLMEM(chains={K}): this layout / engine is not served by the fused sweep kernel (fp32 tiles: shards of at most "
                          "64 rows with lag 6 or 8, or two chains on 64-224-row shards with lag 4-6; compact storage: two or three chains); the chains "
                          "ran one launch each per iteration. Results are the same; pass max_shards=Sampler.shards_for_chains(K) for side-by-side runs.")
        results = []
        for sc, f, pth in zip(samplers, folders, paths):
            if samples in ("text", "binary"):
                sc.set_sample_file(None)
            if samples == "text":
                samples_to_out_files(pth, f, sets, intercept, len(fixed_names) > int(intercept))
                os.remove(pth)
            results.append(_posterior_means(sc, sets, fixed_names, intercept))
        pooled = _pool_results(results)
        pooled["fused"] = bool(fused)
        return pooled
    smp, outFolder = samplers[0], folders[0]
    # header rows (src/mme.jl:543-595)
    if samples in ("text", "text-sync"):
        _write_headers(outFolder, sets, fixed_names)
    return _run_one(smp, outFolder, sets, fixed_names, intercept, nChain, nBurn, nThin, samples)


def _write_headers(outFolder, sets, fixed_names):
    """Header rows of the *Out files (src/mme.jl:543-595)."""
    _out(outFolder, "b", fixed_names)
    _out(outFolder, "varE", ["e"])
    for s in sets:
        names = [f"M{i + 1}" for i in range(s["P"])]  # src/prepMatVec.jl:131
        for nm in s["members"]:
            _out(outFolder, f"beta{nm}", names)
            _out(outFolder, f"delta{nm}", names)
        if isinstance(s["prior"], (BayesBType, BayesCType)):  # src/samplers.jl:80-82
            _out(outFolder, f"pi{s['name']}", ["pi1", "pi2"])
        if isinstance(s["prior"], BayesRType):               # one column per class (src/mme.jl:589-591)
            _out(outFolder, f"pi{s['name']}", [f"pi{v + 1}" for v in range(len(s["prior"].pi))])
        _out(outFolder, f"var{s['name']}", _var_names(s))


def _run_one(smp, outFolder, sets, fixed_names, intercept, nChain, nBurn, nThin, samples):
    # the chain (src/samplers.jl:29-105): kept iterations = burnIn+thin : thin : chainLength
    done = 0
    smp_path = os.path.join(outFolder, "samples.ngpsmp")
    if samples in ("text", "binary"):
        smp.set_sample_file(smp_path)       # kept samples stream out while the chain runs: ONE ngp_run for the whole chain
    if samples == "text-sync":
        for it in range(nBurn + nThin, nChain + 1, nThin):
            smp.run(it - done)
            done = it
            st = smp.get_state()
            _out(outFolder, "b", (_fmt(st["b"]) if intercept else []) + (_fmt(smp.get_fixed()["b"]) if len(fixed_names) > int(intercept) else []))
            _out(outFolder, "varE", _fmt(st["varE"]))
            vb_off = 0
            for k, s in enumerate(sets):
                for m, nm in enumerate(s["members"]):
                    _out(outFolder, f"beta{nm}", _fmt(st["beta"][s["cols"][:, m]]))
                    _out(outFolder, f"delta{nm}", [str(int(v)) for v in st["delta"][s["cols"][:, m]]])
                if isinstance(s["prior"], (BayesBType, BayesCType)):
                    _out(outFolder, f"pi{s['name']}", _fmt(st["piHat"][2 * k:2 * k + 2]))
                if isinstance(s["prior"], BayesRType):
                    _out(outFolder, f"pi{s['name']}", _fmt(smp.get_class_state(s["id"])["piHat"]))
                _out(outFolder, f"var{s['name']}", _fmt(st["varBeta"][vb_off:vb_off + s["nvb"]]))
                vb_off += s["nvb"]
    smp.run(nChain - done)
    if samples in ("text", "binary"):
        smp.set_sample_file(None)
    if samples == "text":
        samples_to_out_files(smp_path, outFolder, sets, intercept, len(fixed_names) > int(intercept))
        os.remove(smp_path)
    return _posterior_means(smp, sets, fixed_names, intercept)


def _posterior_means(smp, sets, fixed_names, intercept):
    ps = smp.get_posterior_sums()
    n = max(ps["nKept"], 1)
    res = dict(nKept=ps["nKept"], b=ps["sum_b"] / n, varE=ps["sum_varE"] / n, sets={}, fixed_names=fixed_names,
               fixed=smp.get_fixed()["sum_b"] / n if len(fixed_names) > int(intercept) else np.zeros(0))
    vb_off = 0
    for k, s in enumerate(sets):
        for m, nm in enumerate(s["members"]):
            cm = s["cols"][:, m]
            res["sets"][nm] = dict(beta=ps["sum_beta"][cm] / n, delta=ps["sum_delta"][cm] / n,
                                   var=ps["sum_varBeta"][vb_off:vb_off + s["nvb"]] / n, pi=ps["sum_pi"][2 * k:2 * k + 2] / n)
            if s["k"] > 1:   # correlated sets: the posterior mean of every region's k x k covariance matrix
                res["sets"][nm]["var"] = (ps["sum_varBeta"][vb_off:vb_off + s["nvb"]] / n).reshape(s["nreg"], s["k"], s["k"])
        if isinstance(s["prior"], BayesRType):
            res["sets"][s["name"]]["pi"] = smp.get_class_state(s["id"])["sum_pi"] / n
        vb_off += s["nvb"]
    res["sampler"] = smp
    return res


def _pool_results(results):
    """Posterior means pooled over chains of equal length (the mean of the chains' means); res["chains"] keeps every chain's own."""
    w = np.array([r["nKept"] for r in results], dtype=np.float64)
    w = w / w.sum() if w.sum() > 0 else np.full(len(results), 1.0 / len(results))
    def avg(get):
        return sum(wi * np.asarray(get(r), dtype=np.float64) for wi, r in zip(w, results))
    pooled = dict(nKept=int(sum(r["nKept"] for r in results)), b=float(avg(lambda r: r["b"])), varE=float(avg(lambda r: r["varE"])),
                  fixed_names=results[0]["fixed_names"], fixed=avg(lambda r: r["fixed"]), sets={}, chains=results,
                  sampler=results[0]["sampler"], samplers=[r["sampler"] for r in results])
    for nm in results[0]["sets"]:
        pooled["sets"][nm] = {k: avg(lambda r, k=k: r["sets"][nm][k]) for k in results[0]["sets"][nm]}
    return pooled
