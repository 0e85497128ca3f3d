"""ctypes binding of libnextgp_hip.so (include/nextgp_hip.h).

This is the stand-in for the Julia `ccall` shim (julia/NextGPHIP.jl): one thin method per C entry
point, no arithmetic on the Python side.  There is no CPU fallback: if the shared library is
missing, or no gfx950 device is usable, construction raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NGP_HIP_LIB") or os.path.join(_HERE, "libnextgp_hip.so")  # override: a library built elsewhere

METHOD_BAYESPR, METHOD_BAYESB, METHOD_BAYESC, METHOD_BAYESR, METHOD_TUPLE = 0, 1, 2, 3, 4


def read_sample_file(path):
    """Binary sample file of ngp_set_sample_file -> dict(iter[n], varE[n], b[n], b_fixed[n, nfix], beta[n, P], varBeta[n, nvb], piHat[n, 2 nsets],
    class_pi[n, nclass], delta[n, P] (uint8), sets=[dict(method, K, col0, ncol, nvb, tk)])."""
    with open(path, "rb") as f:
        if f.read(8) != b"NGPSMP01":
            raise NextGPHipError(f"not a sample file: {path}")
        P, nvb, nsets, nfix, ncls, rec = np.frombuffer(f.read(48), dtype=np.int64)
        sets = [dict(zip(("method", "K", "col0", "ncol", "nvb", "tk"), np.frombuffer(f.read(48), dtype=np.int64).tolist())) for _ in range(nsets)]
        raw = np.frombuffer(f.read(), dtype=np.uint8)
    n = len(raw) // rec
    raw = raw[:n * rec].reshape(n, rec)
    nd = 3 + nfix + P + nvb + 2 * nsets + ncls
    d = raw[:, :nd * 8].copy().view(np.float64)
    o = 3
    out = dict(iter=raw[:, :8].copy().view(np.int64)[:, 0], varE=d[:, 1], b=d[:, 2], sets=sets)
    out["b_fixed"] = d[:, o:o + nfix]; o += nfix
    out["beta"] = d[:, o:o + P]; o += P
    out["varBeta"] = d[:, o:o + nvb]; o += nvb
    out["piHat"] = d[:, o:o + 2 * nsets]; o += 2 * nsets
    out["class_pi"] = d[:, o:o + ncls]
    out["delta"] = raw[:, nd * 8:nd * 8 + P]
    return out


def iter_sample_file(path):
    """The records of a sample file ONE AT A TIME (a record of a 600k-SNP model is 5 MB; a whole file of 1000 kept samples would be
    5 GB): yields dicts with the fields of read_sample_file for a single kept iteration.  Memory-mapped, nothing is copied but
    the record being looked at."""
    with open(path, "rb") as f:
        if f.read(8) != b"NGPSMP01":
            raise NextGPHipError(f"not a sample file: {path}")
        P, nvb, nsets, nfix, ncls, rec = (int(v) for v in np.frombuffer(f.read(48), dtype=np.int64))
        sets = [dict(zip(("method", "K", "col0", "ncol", "nvb", "tk"), np.frombuffer(f.read(48), dtype=np.int64).tolist())) for _ in range(nsets)]
        off = f.tell()
        size = os.fstat(f.fileno()).st_size
    n = (size - off) // rec
    if n <= 0:
        return
    nd = 3 + nfix + P + nvb + 2 * nsets + ncls
    mm = np.memmap(path, dtype=np.uint8, mode="r", offset=off, shape=(n, rec))
    for i in range(n):
        d = np.frombuffer(mm[i, :nd * 8].tobytes(), dtype=np.float64)
        o = 3
        out = dict(iter=int(np.frombuffer(mm[i, :8].tobytes(), dtype=np.int64)[0]), varE=d[1], b=d[2], sets=sets)
        out["b_fixed"] = d[o:o + nfix]; o += nfix
        out["beta"] = d[o:o + P]; o += P
        out["varBeta"] = d[o:o + nvb]; o += nvb
        out["piHat"] = d[o:o + 2 * nsets]; o += 2 * nsets
        out["class_pi"] = d[o:o + ncls]
        out["delta"] = np.asarray(mm[i, nd * 8:nd * 8 + P])
        yield out
    del mm


def tuple_columns(col0, nloc, k):
    """Panel columns of a Tuple (correlated BayesPR) set: array [nloc, k], component m of locus l at col0 + 64 (l // Lb) + k (l % Lb) + m
    with Lb = 64 // k loci per 64-column block (include/nextgp_hip.h, ngp_add_marker_set_tuple)."""
    if col0 % 64 or not 1 <= k <= 4:
        raise ValueError("tuple set: col0 on a 64-column boundary, k in 1..4")
    l = np.arange(nloc, dtype=np.int64)[:, None]
    Lb = 64 // k
    return col0 + 64 * (l // Lb) + k * (l % Lb) + np.arange(k, dtype=np.int64)[None, :]


def tuple_span(nloc, k):
    """Panel columns a tuple set occupies from its first column on (holes included)."""
    Lb = 64 // k
    nblk = (nloc + Lb - 1) // Lb
    return 64 * (nblk - 1) + k * (nloc - Lb * (nblk - 1))


def tuple_panel(sets):
    """k matrices (N x nloc, one per correlated set, same loci) -> the interleaved N x span block a tuple set occupies (unused
    columns zero), same dtype."""
    k, (N, nloc) = len(sets), sets[0].shape
    out = np.zeros((N, tuple_span(nloc, k)), dtype=sets[0].dtype, order="F")
    cols = tuple_columns(0, nloc, k)
    for m, X in enumerate(sets):
        out[:, cols[:, m]] = X
    return out

# every symbol include/nextgp_hip.h declares
SYMBOLS = [
    "ngp_abi_version", "ngp_create", "ngp_destroy", "ngp_last_error", "ngp_set_panel_f64", "ngp_set_panel_f32", "ngp_set_panel_u8",
    "ngp_begin_panel", "ngp_panel_columns_f64", "ngp_panel_columns_f32", "ngp_panel_columns_u8", "ngp_end_panel",
    "ngp_generate_panel", "ngp_get_layout", "ngp_get_mpm", "ngp_get_gram", "ngp_xbeta", "ngp_add_marker_set", "ngp_set_y",
    "ngp_set_residual_prior", "ngp_set_intercept", "ngp_set_schedule", "ngp_run", "ngp_get_state", "ngp_set_state",
    "ngp_get_trace", "ngp_get_posterior_sums", "ngp_posterior_len", "ngp_export_posterior_device", "ngp_sweep_set", "ngp_sweep_set_dev",
    "ngp_get_timing", "ngp_profile_iteration", "ngp_draws_indexed", "ngp_eval_math", "ngp_configure", "ngp_get_config", "ngp_debug_stamps", "ngp_set_near_lags", "ngp_get_near_lags",
    "ngp_set_streamer", "ngp_get_streamer", "ngp_set_storage", "ngp_get_storage", "ngp_set_max_shards", "ngp_shards_for_chains", "ngp_run_many", "ngp_write_panel_file", "ngp_read_panel_header", "ngp_load_panel_file", "ngp_debug_set_mode", "ngp_debug_set_knob", "ngp_set_posterior_sums", "ngp_save_snapshot", "ngp_load_snapshot",
    "ngp_set_trace_loci", "ngp_get_trace_ext", "ngp_allreduce_posterior", "ngp_add_marker_set_r", "ngp_get_class_state", "ngp_set_class_state", "ngp_add_fixed_set", "ngp_get_fixed", "ngp_set_fixed", "ngp_debug_throw", "ngp_get_census", "ngp_debug_set_virtual_device", "ngp_debug_fail_census", "ngp_add_marker_set_tuple", "ngp_share_panel", "ngp_shards_for_pass", "ngp_set_sample_file",
    "ngp_set_chain_form", "ngp_get_chain_form", "ngp_get_setup_timing",
]

_lib = None


class NextGPHipError(RuntimeError):
    pass


def load():
    """dlopen the HIP library; raises if it has not been built (see __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NextGPHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        L.ngp_last_error.restype = C.c_char_p
        L.ngp_last_error.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def write_panel_file(path, G, bits=8):
    """Genotype codes (N x P, uint8) as a binary panel file: 8 bits per genotype, or 2 (allele counts 0/1/2 only)."""
    G = np.asfortranarray(G, dtype=np.uint8)
    rc = load().ngp_write_panel_file(str(path).encode(), _p(G, C.c_uint8), C.c_int64(G.shape[0]), C.c_int64(G.shape[1]),
                                     C.c_int64(G.shape[0]), C.c_int32(bits))
    if rc != 0:
        raise NextGPHipError(f"ngp_write_panel_file failed ({rc}): path not writable, or codes above 2 with bits=2")


def read_panel_header(path):
    n, p, b = C.c_int64(), C.c_int64(), C.c_int32()
    rc = load().ngp_read_panel_header(str(path).encode(), C.byref(n), C.byref(p), C.byref(b))
    if rc != 0:
        raise NextGPHipError(f"not a panel file: {path}")
    return n.value, p.value, b.value


class Sampler:
    """One chain on one device == one `ngp_handle` (reference: one Julia task running runSampler!)."""

    def __init__(self, device=0, seed=1, chain=0, mode=None, lag=None, streamer=None, storage=None):
        self.L = load()
        self.h = C.c_void_p()
        rc = self.L.ngp_create(C.c_int32(device), C.c_uint64(seed), C.c_uint32(chain), C.byref(self.h))
        if rc != 0:
            raise NextGPHipError(f"ngp_create failed ({rc}): " + (self.L.ngp_last_error(None) or b"").decode())
        self.nsets = 0
        self.set_shapes = []  # (ncol, nreg) per set
        self.ntl = 0
        self.ntvb = 0
        if mode is not None or lag is not None:
            self.configure(1 if mode is None else mode, 8 if lag is None else lag)
        if streamer is not None:
            self.set_streamer(streamer)
        if storage is not None:
            self.set_storage(storage)

    def configure(self, mode, lag):
        self._chk(self.L.ngp_configure(self.h, C.c_int32(mode), C.c_int32(lag)))

    def set_near(self, near):
        self._chk(self.L.ngp_set_near_lags(self.h, C.c_int32(near)))

    def near(self):
        n = C.c_int32()
        self._chk(self.L.ngp_get_near_lags(self.h, C.byref(n)))
        return n.value

    def setup_timing(self):
        """Parts of the last generate_panel in ms: dict(alloc_ms, tiles_ms, gram_ms)."""
        a, t, g = C.c_double(), C.c_double(), C.c_double()
        self._chk(self.L.ngp_get_setup_timing(self.h, C.byref(a), C.byref(t), C.byref(g)))
        return dict(alloc_ms=a.value, tiles_ms=t.value, gram_ms=g.value)

    def set_chain_form(self, form):
        """1 (default): BayesPR blocks as dlt = T e0 (k_tinv); 0: the 64 serial steps per block."""
        self._chk(self.L.ngp_set_chain_form(self.h, C.c_int32(int(form))))

    def chain_form(self):
        n = C.c_int32()
        self._chk(self.L.ngp_get_chain_form(self.h, C.byref(n)))
        return n.value

    def set_streamer(self, variant):
        self._chk(self.L.ngp_set_streamer(self.h, C.c_int32(variant)))

    def set_max_shards(self, n):
        self._chk(self.L.ngp_set_max_shards(self.h, C.c_int32(int(n))))

    def shards_for_chains(self, chains):
        """The largest max_shards with which `chains` chains share this device side by side."""
        v = C.c_int32()
        self._chk(self.L.ngp_shards_for_chains(self.h, C.c_int32(int(chains)), C.byref(v)))
        return v.value

    def shards_for_pass(self, chains):
        """The largest max_shards with which `chains` chains share one fused sweep launch (K chains per pass)."""
        v = C.c_int32()
        self._chk(self.L.ngp_shards_for_pass(self.h, C.c_int32(int(chains)), C.byref(v)))
        return v.value

    def share_panel(self, owner):
        """Take `owner`'s panel by reference (K chains per pass): same tiles, Gram window and layout; chain state is this handle's."""
        self._chk(self.L.ngp_share_panel(self.h, owner.h))
        self.N, self.P = owner.N, owner.P
        self._panel_owner = owner   # keeps the owner's Python object alive as long as this one

    def set_storage(self, storage):
        """0 / "f32": centred fp32 tiles; 1 / "u8": compact storage (bytes + Float64 column means, analytic centring)."""
        code = {"f32": 0, "u8": 1}.get(storage, storage)
        self._chk(self.L.ngp_set_storage(self.h, C.c_int32(int(code))))

    def storage(self):
        v = C.c_int32()
        self._chk(self.L.ngp_get_storage(self.h, C.byref(v), None, C.c_int64(0)))
        return v.value

    def means(self):
        """Compact storage: the P column means the centring uses."""
        out = np.zeros(self.P)
        v = C.c_int32()
        self._chk(self.L.ngp_get_storage(self.h, C.byref(v), _p(out, C.c_double), C.c_int64(self.P)))
        return out

    def streamer(self):
        """(variant in force, GEMV chains per shard partial)"""
        v, n = C.c_int32(), C.c_int32()
        self._chk(self.L.ngp_get_streamer(self.h, C.byref(v), C.byref(n)))
        return v.value, n.value

    def debug_set_knob(self, knob):
        self._chk(self.L.ngp_debug_set_knob(self.h, C.c_int32(knob)))

    def debug_set_mode(self, mode):
        self._chk(self.L.ngp_debug_set_mode(self.h, C.c_int32(mode)))

    def config(self):
        m, l = C.c_int32(), C.c_int32()
        self._chk(self.L.ngp_get_config(self.h, C.byref(m), C.byref(l)))
        return m.value, l.value

    def debug_stamps(self, enable=True, n=0):
        out = np.zeros(max(n, 1), dtype=np.uint64)
        self._chk(self.L.ngp_debug_stamps(self.h, C.c_int32(int(enable)), _p(out, C.c_uint64) if n else None, C.c_int64(n)))
        return out

    def _chk(self, rc):
        if rc != 0:
            raise NextGPHipError(f"libnextgp_hip error {rc}: " + (self.L.ngp_last_error(self.h) or b"").decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.ngp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- panel -------------------------------------------------------------------------
    def load_panel_file(self, path, centre=True):
        """Binary panel file (write_panel_file) straight into the tiles of this handle's storage."""
        n, p, _ = read_panel_header(path)
        self._chk(self.L.ngp_load_panel_file(self.h, str(path).encode(), C.c_int32(1 if centre else 0)))
        self.N, self.P = n, p

    def set_panel(self, M, centre=False):
        M = np.asarray(M)
        if M.dtype == np.uint8:  # one byte per genotype: converted and centred on the device
            M = np.asfortranarray(M)
            f, t = self.L.ngp_set_panel_u8, C.c_uint8
        elif M.dtype == np.float32:
            M = np.asfortranarray(M)
            f, t = self.L.ngp_set_panel_f32, C.c_float
        else:
            M = np.asfortranarray(M, dtype=np.float64)
            f, t = self.L.ngp_set_panel_f64, C.c_double
        self.N, self.P = M.shape
        self._chk(f(self.h, _p(M, t), C.c_int64(self.N), C.c_int64(self.P), C.c_int64(self.N), C.c_int32(int(centre))))

    def begin_panel(self, N, P):
        """A panel handed over in column ranges (one marker set after another, no concatenated host copy): begin_panel,
        panel_columns(col0, M) for every range, end_panel."""
        self._chk(self.L.ngp_begin_panel(self.h, C.c_int64(N), C.c_int64(P)))
        self.N, self.P = N, P

    def panel_columns(self, col0, M, centre=False):
        M = np.asarray(M)
        if M.dtype == np.uint8:     # genotype codes: the only form the compact storage takes
            M = np.asfortranarray(M)
            f, t = self.L.ngp_panel_columns_u8, C.c_uint8
        elif M.dtype == np.float32:
            M = np.asfortranarray(M)
            f, t = self.L.ngp_panel_columns_f32, C.c_float
        else:
            M = np.asfortranarray(M, dtype=np.float64)
            f, t = self.L.ngp_panel_columns_f64, C.c_double
        self._chk(f(self.h, C.c_int64(col0), _p(M, t), C.c_int64(M.shape[1]), C.c_int64(M.shape[0]), C.c_int32(int(centre))))

    def end_panel(self):
        self._chk(self.L.ngp_end_panel(self.h))

    def generate_panel(self, N, P, maf_lo=0.05, maf_hi=0.5, seed=20250509):
        self._chk(self.L.ngp_generate_panel(self.h, C.c_int64(N), C.c_int64(P), C.c_double(maf_lo), C.c_double(maf_hi),
                                            C.c_uint64(seed)))
        self.N, self.P = N, P

    def layout(self):
        R, S, nb = C.c_int64(), C.c_int64(), C.c_int64()
        self._chk(self.L.ngp_get_layout(self.h, C.byref(R), C.byref(S), C.byref(nb)))
        return R.value, S.value, nb.value

    def mpm(self):
        out = np.empty(self.P)
        self._chk(self.L.ngp_get_mpm(self.h, _p(out, C.c_double), C.c_int64(self.P)))
        return out

    def gram(self, t):
        out = np.empty((64, 64))
        self._chk(self.L.ngp_get_gram(self.h, C.c_int64(t), _p(out, C.c_double)))
        return out

    def xbeta(self, beta):
        beta = np.ascontiguousarray(beta, dtype=np.float64)
        out = np.empty(self.N)
        self._chk(self.L.ngp_xbeta(self.h, _p(beta, C.c_double), C.c_int64(len(beta)), _p(out, C.c_double), C.c_int64(self.N)))
        return out

    # ---- model -------------------------------------------------------------------------
    def add_marker_set(self, col0, ncol, method, df, scale, regions, varBeta0, pi0=0.0, estPi=False, lhs0=None, rhs0=None):
        rs = np.ascontiguousarray([r[0] for r in regions], dtype=np.int64)
        re = np.ascontiguousarray([r[1] for r in regions], dtype=np.int64)
        vb = np.ascontiguousarray(varBeta0, dtype=np.float64)
        if len(vb) != len(rs):
            raise ValueError("varBeta0 needs one entry per region")
        l0 = None if lhs0 is None else np.ascontiguousarray(lhs0, dtype=np.float64)
        r0 = None if rhs0 is None else np.ascontiguousarray(rhs0, dtype=np.float64)
        sid = C.c_int32()
        self._chk(self.L.ngp_add_marker_set(self.h, C.c_int64(col0), C.c_int64(ncol), C.c_int32(method), C.c_double(df),
                                            C.c_double(scale), _p(rs, C.c_int64), _p(re, C.c_int64), C.c_int64(len(rs)),
                                            _p(vb, C.c_double), C.c_double(pi0), C.c_int32(int(estPi)), _p(l0, C.c_double),
                                            _p(r0, C.c_double), C.byref(sid)))
        self.nsets += 1
        self.set_shapes.append((ncol, len(rs)))
        return sid.value

    def add_fixed_set(self, X, lhs0=None, rhs0=None):
        """Columns of one fixed-effect term / block (src/functions.jl:22-53), sampled after the intercept in the order added."""
        X = np.asfortranarray(X, dtype=np.float64)
        if X.ndim == 1:
            X = np.asfortranarray(X[:, None])
        l0 = None if lhs0 is None else np.ascontiguousarray(lhs0, dtype=np.float64)
        r0 = None if rhs0 is None else np.ascontiguousarray(rhs0, dtype=np.float64)
        sid = C.c_int32()
        self._chk(self.L.ngp_add_fixed_set(self.h, _p(X, C.c_double), C.c_int64(X.shape[0]), C.c_int64(X.shape[1]), C.c_int64(X.shape[0]),
                                           _p(l0, C.c_double), _p(r0, C.c_double), C.byref(sid)))
        self.nfixcol = getattr(self, "nfixcol", 0) + X.shape[1]
        return sid.value

    def get_fixed(self):
        n = getattr(self, "nfixcol", 0)
        b = np.empty(max(n, 1)); sb = np.empty(max(n, 1)); nn = C.c_int64()
        self._chk(self.L.ngp_get_fixed(self.h, _p(b, C.c_double), _p(sb, C.c_double), C.byref(nn)))
        return dict(b=b[:nn.value].copy(), sum_b=sb[:nn.value].copy())

    def set_fixed(self, b=None, sum_b=None):
        a = None if b is None else np.ascontiguousarray(b, dtype=np.float64)
        c = None if sum_b is None else np.ascontiguousarray(sum_b, dtype=np.float64)
        self._chk(self.L.ngp_set_fixed(self.h, _p(a, C.c_double), _p(c, C.c_double), C.c_int64(getattr(self, "nfixcol", 0))))

    def add_marker_set_r(self, col0, ncol, df, scale, varBeta0, vClass, pi, estPi=False, lhs0=None, rhs0=None):
        """BayesR set: class multipliers vClass of the set's single variance, class probabilities pi (src/mme.jl:374-383)."""
        vc = np.ascontiguousarray(vClass, dtype=np.float64); pp = np.ascontiguousarray(pi, dtype=np.float64)
        if len(vc) != len(pp):
            raise ValueError("vClass and pi need one entry per class")
        l0 = None if lhs0 is None else np.ascontiguousarray(lhs0, dtype=np.float64)
        r0 = None if rhs0 is None else np.ascontiguousarray(rhs0, dtype=np.float64)
        sid = C.c_int32()
        self._chk(self.L.ngp_add_marker_set_r(self.h, C.c_int64(col0), C.c_int64(ncol), C.c_double(df), C.c_double(scale), C.c_double(varBeta0),
                                              _p(vc, C.c_double), _p(pp, C.c_double), C.c_int32(len(vc)), C.c_int32(int(estPi)), _p(l0, C.c_double),
                                              _p(r0, C.c_double), C.byref(sid)))
        self.nsets += 1
        self.set_shapes.append((ncol, 1))
        self.nclasses = getattr(self, "nclasses", 0) + len(vc)
        return sid.value

    def add_marker_set_tuple(self, col0, nloc, k, df, scale, regions, varBeta0):
        """Correlated sets (BayesPR's Tuple method, src/functions.jl:140-154): k sets, nloc loci, columns as tuple_columns(col0, nloc, k);
        scale and varBeta0 k x k, regions ranges of loci."""
        rs = np.ascontiguousarray([r[0] for r in regions], dtype=np.int64)
        re = np.ascontiguousarray([r[1] for r in regions], dtype=np.int64)
        sc = np.ascontiguousarray(np.asarray(scale, dtype=np.float64).reshape(k, k)); vb = np.ascontiguousarray(np.asarray(varBeta0, dtype=np.float64).reshape(k, k))
        sid = C.c_int32()
        self._chk(self.L.ngp_add_marker_set_tuple(self.h, C.c_int64(col0), C.c_int64(nloc), C.c_int32(k), C.c_double(df), _p(sc, C.c_double),
                                                  _p(rs, C.c_int64), _p(re, C.c_int64), C.c_int64(len(rs)), _p(vb, C.c_double), C.byref(sid)))
        self.nsets += 1
        self.set_shapes.append((tuple_span(nloc, k), len(rs) * k * k))
        return sid.value

    def get_class_state(self, set_id):
        pi = np.empty(16); sp = np.empty(16); K = C.c_int64()
        self._chk(self.L.ngp_get_class_state(self.h, C.c_int32(set_id), _p(pi, C.c_double), _p(sp, C.c_double), C.byref(K)))
        return dict(piHat=pi[:K.value].copy(), sum_pi=sp[:K.value].copy())

    def set_class_state(self, set_id, piHat=None, sum_pi=None):
        a = None if piHat is None else np.ascontiguousarray(piHat, dtype=np.float64)
        b = None if sum_pi is None else np.ascontiguousarray(sum_pi, dtype=np.float64)
        K = len(a) if a is not None else len(b)
        self._chk(self.L.ngp_set_class_state(self.h, C.c_int32(set_id), _p(a, C.c_double), _p(b, C.c_double), C.c_int64(K)))

    def set_y(self, y):
        y = np.ascontiguousarray(y, dtype=np.float64)
        self._chk(self.L.ngp_set_y(self.h, _p(y, C.c_double), C.c_int64(len(y))))

    def set_residual_prior(self, df, scale):
        self._chk(self.L.ngp_set_residual_prior(self.h, C.c_double(df), C.c_double(scale)))

    def set_intercept(self, on):
        self._chk(self.L.ngp_set_intercept(self.h, C.c_int32(int(on))))

    def set_schedule(self, chainLength, burnIn, thin):
        self._chk(self.L.ngp_set_schedule(self.h, C.c_int64(chainLength), C.c_int64(burnIn), C.c_int64(thin)))

    # ---- run / read back ---------------------------------------------------------------
    def run(self, niter):
        self._chk(self.L.ngp_run(self.h, C.c_int64(niter)))

    @property
    def nvb(self):
        return sum(s[1] for s in self.set_shapes)

    def get_state(self):
        yc = np.empty(self.N); beta = np.empty(self.P); delta = np.empty(self.P, dtype=np.int64)
        vb = np.empty(max(self.nvb, 1)); pi = np.empty(2 * max(self.nsets, 1))
        varE, b, it = C.c_double(), C.c_double(), C.c_int64()
        self._chk(self.L.ngp_get_state(self.h, _p(yc, C.c_double), _p(beta, C.c_double), _p(delta, C.c_int64), _p(vb, C.c_double),
                                       _p(pi, C.c_double), C.byref(varE), C.byref(b), C.byref(it)))
        return dict(ycorr=yc, beta=beta, delta=delta, varBeta=vb[:self.nvb], piHat=pi[:2 * self.nsets], varE=varE.value,
                    b=b.value, iter=it.value)

    def set_state(self, st):
        yc = np.ascontiguousarray(st["ycorr"], dtype=np.float64); beta = np.ascontiguousarray(st["beta"], dtype=np.float64)
        delta = np.ascontiguousarray(st["delta"], dtype=np.int64); vb = np.ascontiguousarray(st["varBeta"], dtype=np.float64)
        pi = np.ascontiguousarray(st["piHat"], dtype=np.float64)
        self._chk(self.L.ngp_set_state(self.h, _p(yc, C.c_double), _p(beta, C.c_double), _p(delta, C.c_int64), _p(vb, C.c_double),
                                       _p(pi, C.c_double), C.c_double(st["varE"]), C.c_double(st["b"]), C.c_int64(st["iter"])))

    def get_trace(self, n):
        v = np.empty(n); b = np.empty(n)
        self._chk(self.L.ngp_get_trace(self.h, _p(v, C.c_double), _p(b, C.c_double), C.c_int64(n)))
        return dict(varE=v, b=b)

    def get_posterior_sums(self):
        sb = np.empty(self.P); sb2 = np.empty(self.P); sd = np.empty(self.P); sv = np.empty(max(self.nvb, 1))
        sp = np.empty(2 * max(self.nsets, 1)); se, sbb, nk = C.c_double(), C.c_double(), C.c_int64()
        self._chk(self.L.ngp_get_posterior_sums(self.h, _p(sb, C.c_double), _p(sb2, C.c_double), _p(sd, C.c_double),
                                                _p(sv, C.c_double), _p(sp, C.c_double), C.byref(se), C.byref(sbb), C.byref(nk)))
        return dict(sum_beta=sb, sum_beta2=sb2, sum_delta=sd, sum_varBeta=sv[:self.nvb], sum_pi=sp[:2 * self.nsets],
                    sum_varE=se.value, sum_b=sbb.value, nKept=nk.value)

    def posterior_len(self):
        n = C.c_int64()
        self._chk(self.L.ngp_posterior_len(self.h, C.byref(n)))
        return n.value

    def export_posterior_device(self, device_ptr, length):
        self._chk(self.L.ngp_export_posterior_device(self.h, C.c_void_p(device_ptr), C.c_int64(length)))

    def sweep_set(self, set_id, varE, ycorr, beta, varBeta, piHat=None):
        """Fine seam (M[set].funct): arrays are updated in place; returns delta."""
        assert ycorr.dtype == np.float64 and beta.dtype == np.float64 and varBeta.dtype == np.float64
        delta = np.empty(len(beta), dtype=np.int64)
        self._chk(self.L.ngp_sweep_set(self.h, C.c_int32(set_id), C.c_double(varE), _p(ycorr, C.c_double), _p(beta, C.c_double),
                                       _p(delta, C.c_int64), _p(varBeta, C.c_double), _p(piHat, C.c_double)))
        return delta

    def sweep_set_dev(self, set_id, varE, d_ycorr, d_beta, d_varBeta, d_delta=0, d_piHat=0):
        """The fine seam over DEVICE arrays (addresses, e.g. torch.Tensor.data_ptr(): ycorr N, beta ncol, varBeta nreg float64; delta
        ncol int64 or 0; piHat 2 float64 or 0), updated in place on the device."""
        self._chk(self.L.ngp_sweep_set_dev(self.h, C.c_int32(set_id), C.c_double(varE), C.c_void_p(d_ycorr), C.c_void_p(d_beta),
                                           C.c_void_p(d_delta or None), C.c_void_p(d_varBeta), C.c_void_p(d_piHat or None)))

    def get_timing(self):
        it = C.c_double()
        sl, ni = C.c_int64(), C.c_int64()
        self._chk(self.L.ngp_get_timing(self.h, C.byref(sl), C.byref(it), C.byref(ni)))
        return dict(sweep_launches=sl.value, iter_ms=it.value, iters=ni.value)

    def set_posterior_sums(self, ps):
        a = {k: np.ascontiguousarray(ps[k], dtype=np.float64) for k in ("sum_beta", "sum_beta2", "sum_delta", "sum_varBeta", "sum_pi")}
        self._chk(self.L.ngp_set_posterior_sums(self.h, _p(a["sum_beta"], C.c_double), _p(a["sum_beta2"], C.c_double),
                                                _p(a["sum_delta"], C.c_double), _p(a["sum_varBeta"], C.c_double), _p(a["sum_pi"], C.c_double),
                                                C.c_double(ps["sum_varE"]), C.c_double(ps["sum_b"]), C.c_int64(ps["nKept"])))

    def set_sample_file(self, path):
        """Every kept iteration of the following runs leaves a binary record in `path` without stopping the chain; None closes the file."""
        self._chk(self.L.ngp_set_sample_file(self.h, None if path is None else os.fsencode(path)))

    def save_snapshot(self, path):
        self._chk(self.L.ngp_save_snapshot(self.h, os.fsencode(path)))

    def load_snapshot(self, path):
        self._chk(self.L.ngp_load_snapshot(self.h, os.fsencode(path)))

    def set_trace_loci(self, loci, n_varBeta=0):
        loci = np.ascontiguousarray(loci, dtype=np.int64)
        self._chk(self.L.ngp_set_trace_loci(self.h, _p(loci, C.c_int64) if len(loci) else None, C.c_int64(len(loci)), C.c_int64(n_varBeta)))
        self.ntl, self.ntvb = len(loci), n_varBeta

    def get_trace_ext(self, n):
        bt = np.empty((n, max(self.ntl, 1))); vt = np.empty((n, max(self.ntvb, 1))); pt = np.empty((n, max(self.nsets, 1)))
        self._chk(self.L.ngp_get_trace_ext(self.h, _p(bt, C.c_double), _p(vt, C.c_double), _p(pt, C.c_double), C.c_int64(n)))
        return dict(beta=bt[:, :self.ntl], varBeta=vt[:, :self.ntvb], pi=pt[:, :self.nsets])

    def census(self):
        """Placement of the workgroups of the last persistent-sweep launch (ngp_get_census): dict(grid, retries, exclusive,
        xcc[grid] (0-7, -1 = never resident), hw_id[grid])."""
        g, r, e = C.c_int64(), C.c_int64(), C.c_int32()
        self._chk(self.L.ngp_get_census(self.h, None, C.c_int64(0), C.byref(g), C.byref(r), C.byref(e)))
        tb = np.zeros(g.value, dtype=np.uint64)
        self._chk(self.L.ngp_get_census(self.h, _p(tb, C.c_uint64), C.c_int64(g.value), None, None, None))
        return dict(grid=g.value, retries=r.value, exclusive=bool(e.value), xcc=(tb >> np.uint64(32)).astype(np.int64) - 1,
                    hw_id=(tb & np.uint64(0xFFFFFFFF)).astype(np.int64))

    def debug_fail_census(self, iteration):
        self._chk(self.L.ngp_debug_fail_census(self.h, C.c_int64(iteration)))

    def debug_set_virtual_device(self, vdev):
        self._chk(self.L.ngp_debug_set_virtual_device(self.h, C.c_int32(vdev)))

    @staticmethod
    def allreduce_posterior(samplers):
        """Pooled posterior sums over the chains of `samplers` (ngp_allreduce_posterior): every sampler then holds them."""
        L = samplers[0].L
        arr = (C.c_void_p * len(samplers))(*[s.h for s in samplers])
        samplers[0]._chk(L.ngp_allreduce_posterior(arr, C.c_int32(len(samplers))))

    @staticmethod
    def run_many(samplers, niter):
        """niter iterations of every chain of `samplers` at once, one thread per chain inside the library (ngp_run_many)."""
        L = samplers[0].L
        arr = (C.c_void_p * len(samplers))(*[s.h for s in samplers])
        rc = L.ngp_run_many(arr, C.c_int32(len(samplers)), C.c_int64(niter))
        if rc != 0:
            msgs = [(s.L.ngp_last_error(s.h) or b"").decode() for s in samplers]
            raise NextGPHipError(f"libnextgp_hip error {rc}: " + " | ".join(m for m in msgs if m))

    def profile_iteration(self):
        ms, by = C.c_double(), C.c_double()
        n = C.c_int64()
        self._chk(self.L.ngp_profile_iteration(self.h, C.byref(ms), C.byref(n), C.byref(by)))
        return dict(avg_ms=ms.value, launches=n.value, bytes_per_launch=by.value)

    # ---- probes ------------------------------------------------------------------------
    def draws_indexed(self, it, kind, index0, what, n, p1=0.0, p2=0.0):
        out = np.empty(n)
        self._chk(self.L.ngp_draws_indexed(self.h, C.c_uint64(it), C.c_uint64(kind), C.c_uint64(index0), C.c_int32(what),
                                           C.c_double(p1), C.c_double(p2), C.c_int64(n), _p(out, C.c_double)))
        return out

    def eval_math(self, which, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty(len(x))
        self._chk(self.L.ngp_eval_math(self.h, C.c_int32(which), _p(x, C.c_double), C.c_int64(len(x)), _p(out, C.c_double)))
        return out
