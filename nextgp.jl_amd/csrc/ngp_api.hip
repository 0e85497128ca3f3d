// ngp_api.hip -- C ABI of libnextgp_hip.so (include/nextgp_hip.h): handle lifecycle, panel
// upload / generation with re-tiling, model set-up, the per-iteration launch sequence of the
// blocked Gibbs sweep, state and posterior read-back.  gfx950 only; there is NO CPU fallback:
// without a usable device every entry point fails with NGP_ERR_NODEVICE / NGP_ERR_HIP.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/nextgp_hip.h"
#include "ngp_kernels.h"
#include "ngp_sweep_args.h"

using namespace ngp;

namespace {

struct HFix {  // one fixed-effect set beyond the intercept
    int64_t ncol, off;
    double *d_X = nullptr, *d_xpx0 = nullptr, *d_xpxR = nullptr, *d_lhs0 = nullptr, *d_rhs0 = nullptr;
};

struct PanelMem {  // d_tiles / d_mean / d_gramx / d_mpm of one uploaded panel; handles that share it hold a reference each
    void *tiles = nullptr, *mean = nullptr, *gramx = nullptr, *mpm = nullptr;
    std::atomic<int> refs{1};
};

// Kept samples streamed to a binary file while the chain runs (ngp_set_sample_file): a ring of NSLOT records on the device, copied to
// pinned host memory on a second stream, written by a thread of its own.  The chain's stream never waits for the file -- only for a
// ring slot, when the writer is NSLOT samples behind.
struct SampleStream {
    static constexpr int NSLOT = 4;
    FILE *f = nullptr;
    std::string path;
    size_t rec_bytes = 0;
    bool header_written = false;
    int device = 0;
    unsigned char *d_slot[NSLOT] = {nullptr, nullptr, nullptr, nullptr}, *h_slot[NSLOT] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_packed[NSLOT] = {nullptr, nullptr, nullptr, nullptr}, ev_copied[NSLOT] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
    std::thread writer;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<int> queue;          // slots whose copy has been enqueued, in order
    bool busy[NSLOT] = {false, false, false, false};
    bool stop = false, io_error = false;
    int64_t nenq = 0, nwritten = 0, ndropped = 0;
};

struct HSet {
    int64_t col0, ncol;
    int method;
    double df, scale;
    int64_t nreg;
    int64_t vb_off;
    int estPi;
    uint64_t fine_calls;
    double pi0;               // prior inclusion probability (src/mme.jl:351,360): what ngp_set_y goes back to
    std::vector<double> vb0;  // initial variances (src/mme.jl:516)
    int K = 0;                // BayesR: classes, their multipliers and prior probabilities (src/mme.jl:374-383)
    std::vector<double> vcls, rpi;
    int tk = 0;               // Tuple (correlated BayesPR) set: number of correlated sets; nreg k x k variance matrices in varBeta
    int64_t nloc = 0;
};

std::string g_create_err;

}  // namespace

struct ngp_handle {
    int device = 0;
    uint64_t seed = 0;
    uint32_t chain = 0;
    hipStream_t stream = nullptr;
    int64_t N = 0, P = 0, R = 0, S = 0, NBLK = 0, Ppad = 0, L = 0;
    size_t lds_step = 0, lds_sweep = 0, lds_rows = 0;
    size_t lds_sweep_lean = 0;  // the same without the sampler's staging area of Tuple coefficients: what k_sweep<false> is launched with
    size_t lds_sweep_r = 0;     // with the BayesR staging area as well: what k_sweep_r is launched with (0: does not fit -> k_sweep_tup)
    int mode = 1;      // 1: persistent sweep kernel, 0: one streaming + one recursion launch per block
    int lag = 8;       // look-ahead D of the persistent sweep (blocks); shards taller than 128 rows are capped at 5
    bool lag_auto = true;  // lag not chosen by the caller (ngp_configure): tall shards then take the measured best
    int near_req = 0;  // near lags requested (0 = automatic)
    int near = 3;      // look-ahead lags 1..near corrected by the sampler itself, farther ones by the reducers
    int max_shards_req = 0;  // streamer workgroups the persistent sweep may use (0 = all CUs but the sampler's and the reducers')
    int storage = 0;       // 0: centred fp32 tiles; 1: compact -- byte tiles + Float64 column means (ngp_set_storage)
    double *d_mean = nullptr;  // compact storage: column means, Ppad
    int streamer_req = 0;  // streamer variant requested: 0 automatic, 1 phase streamer, 2 row-owning waves + loader wave, 4 / 6 = 2 with
                           // two / three shards per workgroup at any N (automatic only above one resident wave of 256-row shards)
    bool panel_open = false;  // between ngp_begin_panel and ngp_end_panel: columns may still arrive, the Gram window does not exist yet
    int V = 1;             // shards per streamer workgroup (role_streamer_rows_tall: 2, 3); the sweep's grid has S / V streamers
    int streamer = 1;      // variant in force (persistent sweep only)
    int nchain = 8;        // GEMV chains per shard partial: 8 (phase streamer, per-block engine) or 7 (row-owning waves)
    int D = 1;         // Gram planes stored per block (= lag in mode 1, 1 in mode 0)
    int NG = 1;        // reducer groups = ceil(S/32)
    int cu_count = 256;
    double *d_cdlt = nullptr;
    unsigned long long *d_cacc = nullptr;   // fixed-point accumulators of X_t'ycorr (inside d_ccnt: zeroed with the counters by k_prep)
    double mpm_max = 0.0;                   // max_j x_j'x_j of the panel (scale of the accumulators, k_head)
    double setup_ms[3] = {0.0, 0.0, 0.0};   // wall time of the last panel set-up: device allocation (+ zeroing) | tiles (generation / upload) | Gram window
    unsigned long long *d_cdltg = nullptr;  // dlt as tagged granules
    unsigned launch_seq = 0;                // launch nonce of the granule tags
    unsigned *d_ccnt = nullptr, *d_abort = nullptr;
    unsigned long long *d_dbg = nullptr;
    size_t ccnt_words = 0;
    float *d_tiles = nullptr;
    double *d_gramx = nullptr, *d_mpm = nullptr, *d_lhs0 = nullptr, *d_rhs0 = nullptr, *d_beta = nullptr;
    double *d_c = nullptr, *d_w = nullptr, *d_q = nullptr, *d_T = nullptr, *d_chi = nullptr;
    // inverse form of the linear blocks' chain (k_tinv): T per block, written before every sweep
    double *d_tinv = nullptr;
    int64_t tinv_blocks = 0;  // blocks d_tinv was allocated for
    unsigned *d_blin = nullptr;   // [NBLK] 1 = linear block (static for a model and an active set: written by sync_linear_blocks)
    int blin_for = -2;            // active set d_blin was written for (-1: the whole model, -2: stale)
    int lin_all = 0, lin_any = 0; // every / any block of d_blin is linear
    int chain_form = 0;       // ngp_set_chain_form: 0 = every block by the 64-step chain (default), 1 = linear blocks as dlt = T e0
    int8_t *d_setof = nullptr;
    int32_t *d_loc = nullptr, *d_vbidx = nullptr;
    uint8_t *d_delta = nullptr;
    double *d_sum_beta = nullptr, *d_sum_beta2 = nullptr, *d_sum_delta = nullptr;
    double *d_ycorr = nullptr, *d_part = nullptr, *d_dlt = nullptr;
    DSet *d_sets = nullptr;
    DScal *d_scal = nullptr;
    double *d_varBeta = nullptr, *d_sum_varBeta = nullptr;
    int64_t vb_cap = 0;
    std::vector<HFix> fix;           // fixed-effect sets beyond the intercept (src/functions.jl:22-53), in sampling order
    int64_t nfixcol = 0;
    double *d_bfix = nullptr, *d_sum_bfix = nullptr;
    double *d_rcls = nullptr;        // BayesR per-locus class coefficients [4][NGP_RMAX][Ppad] (allocated with the first BayesR set)
    int32_t *d_seg_set = nullptr;    // set of every variance segment
    std::vector<int32_t> h_seg_set;
    int64_t nclass_total = 0;        // sum of K over the BayesR sets (entries of the packed posterior)
    // Tuple sets (src/functions.jl:140-154): per-set constants, coefficient rows, region tables of the inverse-Wishart draws
    DTup *d_tup = nullptr;
    double *d_tupc = nullptr, *d_tupg = nullptr, *d_tsegpart = nullptr;
    DTReg *d_tregs = nullptr;
    long long *d_tseg_l0 = nullptr;
    int32_t *d_tseg_len = nullptr, *d_tseg_set = nullptr;
    std::vector<DTReg> h_tregs;
    std::vector<long long> h_tseg_l0;
    std::vector<int32_t> h_tseg_len, h_tseg_set;
    int ntuple = 0;
    // PR region tables
    DReg *d_regs = nullptr;
    long long *d_seg_k0 = nullptr;
    int32_t *d_seg_len = nullptr;
    double *d_segpart = nullptr, *d_regchi = nullptr;
    std::vector<DReg> h_regs;
    std::vector<long long> h_seg_k0;
    std::vector<int32_t> h_seg_len;
    bool tables_dirty = false;
    // model (host mirror)
    std::vector<HSet> sets;
    std::vector<int8_t> h_setof;
    std::vector<int32_t> h_loc, h_vbidx;
    int64_t nvb = 0;
    double e_df = 4.0, e_scale = 0.0005;
    int intercept = 1;
    int64_t chainLength = 0, burnIn = 0, thin = 1;
    int64_t iter = 0;
    bool have_y = false;
    // traces
    double *d_tr_varE = nullptr, *d_tr_b = nullptr;
    int64_t ntrace = 0, trace_cap = 0;
    // timing
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double iter_ms = 0.0;
    int64_t iters_timed = 0;
    int64_t sweep_launches = 0;
    // diagnostics (ngp_debug_set_mode): != 0 makes every chain invalid, ngp_run / ngp_sweep_set then return NGP_ERR_DEBUG
    int dbg_mode = 0;
    int gram_engine = 0;  // 0: Gram window by the fp64 VALU kernel (k_gram_part), 1: on the matrix cores (k_gram_part_mfma, bit-identical); the knob's bit 10
    int knob = 0;  // pacing of the loader wave of the row-owning streamer: s_sleep units after every four requests (ngp_debug_set_knob)
    bool adding_r = false;  // ngp_add_marker_set is being called by ngp_add_marker_set_r
    bool poisoned = false;  // a sweep gave up half-way (abort word): the chain state is unusable until ngp_set_y / ngp_set_state
    bool exclusive = false;  // a grid of this handle was once not co-resident beside other chains' grids: its calls now lease the whole device
    int64_t last_grid = 0;       // workgroups of the last sweep launch this handle led (fused launches: K (1 + NG) + S)
    int64_t census_retries = 0;  // launches that ended at the census and were run again with the device to themselves
    int64_t dbg_census_fail_iter = 0;  // ngp_debug_fail_census: the sweep of this iteration ends at its census (once)
    SampleStream *smp = nullptr;        // ngp_set_sample_file
    struct PanelMem *pm = nullptr;      // the panel's device arrays (tiles, Gram window, x'x, means), shared by reference count: ngp_share_panel
    int vdev = -1;           // ngp_debug_set_virtual_device: the device ngp_allreduce_posterior groups this handle under (-1: the real one)
    unsigned long long *d_census_tbl = nullptr;  // placement of the workgroups of the last sweep launch (inside d_ccnt)
    size_t census_off = 0;   // word offset of the census counters inside d_ccnt
    // optional per-iteration traces of selected effects, variances and pi (ngp_set_trace_loci)
    int64_t *d_trace_loci = nullptr;
    int64_t ntl = 0, ntvb = 0;
    double *d_tr_beta = nullptr, *d_tr_vb = nullptr, *d_tr_pi = nullptr;
    int64_t trace_ext_cap = 0;
    std::string err;
};

namespace {

int fail(ngp_handle *h, int code, const std::string &msg) {
    if (h) h->err = msg; else g_create_err = msg;
    return code;
}
// the same for the exception barrier of the C ABI: storing the message must not throw a second time
int fail_nothrow(ngp_handle *h, int code, const char *what) noexcept {
    try {
        std::string &dst = h ? h->err : g_create_err;
        dst.assign(what ? what : "?");
    } catch (...) {  // out of memory while storing the message: keep the code, drop the text
        try { (h ? h->err : g_create_err).clear(); } catch (...) {}
    }
    return code;
}

// Exception barrier (include/nextgp_hip.h: "no C++ exception crosses this boundary"; SURVEY.md section 8b, error conventions):
// every extern "C" body runs inside NGP_TRY ... NGP_CATCH(handle).  std::vector / std::string / std::thread can throw
// (bad_alloc, length_error, system_error); unwinding into the caller's ccall frame would abort the Julia process, so they become
// a negative status and a message for ngp_last_error, like every other failure (the reference's error(...) style, src/mme.jl:77,343).
#define NGP_TRY try {
#define NGP_CATCH(H)                                                                                                 \
    }                                                                                                                \
    catch (const std::bad_alloc &) { return fail_nothrow((H), NGP_ERR_NOMEM, "out of host memory (std::bad_alloc)"); } \
    catch (const std::exception &e_) {                                                                               \
        char b_[256];                                                                                                \
        std::snprintf(b_, sizeof b_, "internal error (C++ exception stopped at the C ABI): %s", e_.what());          \
        return fail_nothrow((H), NGP_ERR_HIP, b_);                                                                   \
    }                                                                                                                \
    catch (...) { return fail_nothrow((H), NGP_ERR_HIP, "internal error (unknown C++ exception stopped at the C ABI)"); }

#define HCHK(call)                                                                                          \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return fail(h, NGP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));                 \
    } while (0)

#define REQUIRE(cond, code, msg)                \
    do {                                        \
        if (!(cond)) return fail(h, code, msg); \
    } while (0)

// Persistent sweeps of SEVERAL handles of this process on one device (chains in their own host threads, ngp_set_max_shards):
// every workgroup of a sweep waits for others of its grid, so two sweeps may only run together if both grids fit the device at
// once.  A call that launches sweeps leases its grid's CUs for its duration; a lease that does not fit waits for the running
// calls to return (the chains then take turns instead of giving up on their spins).  Other processes cannot be seen from here:
// against them the bounded spins and the abort word remain.
struct CuLease {
    static std::mutex &mu() { static std::mutex m; return m; }
    static std::condition_variable &cv() { static std::condition_variable c; return c; }
    static int *in_use() { static int u[64] = {0}; return u; }
    int dev = -1, n = 0, cap = 0, want = 0;
    bool excl = false;
    explicit CuLease(ngp_handle *h, int64_t grid_override = 0) {
        if (h->mode != 1) return;
        // workgroups are handed to the 8 XCDs in turn, so a grid occupies ceil(grid / 8) CUs of EVERY XCD: the unit of the lease
        // (three grids of 85 workgroups -- 255 of 256 CUs -- do not fit: 3 x 11 > 32 per XCD; measured, they wait for each other)
        dev = h->device & 63; want = (int)(((grid_override > 0 ? grid_override : 1 + h->NG + h->S / h->V) + 7) / 8);
        cap = std::max(1, h->cu_count / 8);
        acquire(h->exclusive);
    }
    void acquire(bool exclusive) {
        if (dev < 0) return;
        excl = exclusive; n = exclusive ? cap : want;  // exclusive: the whole device, i.e. nobody else's sweep beside this call's
        std::unique_lock<std::mutex> lk(mu());
        cv().wait(lk, [&] { return in_use()[dev] == 0 || in_use()[dev] + n <= cap; });
        in_use()[dev] += n;
    }
    void release() {
        if (dev < 0 || n == 0) return;
        { std::lock_guard<std::mutex> lk(mu()); in_use()[dev] -= n; }
        n = 0;
        cv().notify_all();
    }
    // the count fitted and the grids still were not all resident (census): give the share back and wait for the device to be free
    void make_exclusive() { release(); acquire(true); }
    ~CuLease() { release(); }
};

int enter(ngp_handle *h) {
    if (!h) return fail(nullptr, NGP_ERR_ARG, "null handle");
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) return fail(h, NGP_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    return NGP_OK;
}

template <typename T>
int dalloc(ngp_handle *h, T **p, size_t n) {
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    hipError_t e = hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T));
    if (e != hipSuccess) return fail(h, NGP_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    e = hipMemsetAsync(*p, 0, std::max<size_t>(n, 1) * sizeof(T), h->stream);
    if (e != hipSuccess) return fail(h, NGP_ERR_HIP, std::string("hipMemset: ") + hipGetErrorString(e));
    return NGP_OK;
}
template <typename T>
void dfree(T *&p) {
    if (p) { (void)hipFree(p); p = nullptr; }
}

// rows per shard R: a multiple of 4 -- 4*odd where that fits under the cap when the shard count is left to the library (the layouts
// of the first versions, kept so that results stay comparable; with quad-major tiles any multiple of 4 reads conflict-free), the
// smallest multiple of 4 that serves an explicit ngp_set_max_shards (there every workgroup counts: 10k rows on 209 shards are 48 rows
// each, 52 would leave 16 CUs idle); S = ceil(N/R)
void choose_layout(int64_t N, int64_t max_shards, int64_t r_cap, int64_t *R, int64_t *S, bool prefer_odd = true) {
    int64_t r0 = (N + max_shards - 1) / max_shards;
    int64_t m = (r0 + 3) / 4;
    if (m < 1) m = 1;
    if (prefer_odd && (m & 1) == 0 && 4 * (m + 1) <= r_cap) m += 1;
    int64_t r = 4 * m;
    if (r > r_cap) r = r_cap;
    *R = r;
    *S = (N + r - 1) / r;
}

// drop this handle's reference to its panel arrays; the last reference frees them
void release_panel(ngp_handle *h) {
    if (h->pm) {
        if (h->pm->refs.fetch_sub(1) == 1) {
            (void)hipFree(h->pm->tiles); (void)hipFree(h->pm->mean); (void)hipFree(h->pm->gramx); (void)hipFree(h->pm->mpm);
            delete h->pm;
        }
        h->pm = nullptr;
    } else {  // arrays of an allocation that failed half-way (no store yet)
        if (h->d_tiles) (void)hipFree(h->d_tiles);
        if (h->d_mean) (void)hipFree(h->d_mean);
        if (h->d_gramx) (void)hipFree(h->d_gramx);
        if (h->d_mpm) (void)hipFree(h->d_mpm);
    }
    h->d_tiles = nullptr; h->d_mean = nullptr; h->d_gramx = nullptr; h->d_mpm = nullptr;
}

int alloc_panel(ngp_handle *h, int64_t N, int64_t P, ngp_handle *owner = nullptr) {
    REQUIRE(N > 0 && P > 0, NGP_ERR_ARG, "panel dimensions must be positive");
    REQUIRE(N <= (int64_t)508 * 1024, NGP_ERR_ARG, "N too large for this build (max 520192)");
    release_panel(h);  // (handles that share the old panel keep it alive)
    h->N = N; h->P = P;
    if (owner) {  // the owner's layout, engine and storage, as they are
        h->mode = owner->mode; h->lag = owner->lag; h->lag_auto = owner->lag_auto; h->near_req = owner->near_req; h->near = owner->near;
        h->max_shards_req = owner->max_shards_req; h->storage = owner->storage; h->streamer_req = owner->streamer_req;
        h->lds_rows = owner->lds_rows;
        h->streamer = owner->streamer; h->nchain = owner->nchain; h->D = owner->D; h->NG = owner->NG; h->R = owner->R; h->S = owner->S;
        h->V = owner->V;
    } else {
    // persistent mode: sampler + reducers + S streamers must all be resident, one workgroup per CU
    int64_t max_shards = 256;
    h->V = 1;
    if (h->storage == 1) {
        // compact storage: byte tiles, units of 16 rows, the row-owning roles only (persistent sweep)
        REQUIRE(h->mode == 1, NGP_ERR_ARG, "compact storage runs in the persistent sweep (ngp_configure mode 1) only");
        max_shards = h->cu_count - 1 - (h->cu_count + NGP_GRP - 1) / NGP_GRP;
        if (h->max_shards_req > 0) max_shards = std::min<int64_t>(max_shards, h->max_shards_req);
        const int64_t r0 = (N + max_shards - 1) / max_shards;
        h->R = 16 * std::max<int64_t>(1, (r0 + 15) / 16);
        REQUIRE(h->R <= NGP_U8_MAX_R, NGP_ERR_ARG, "N too large for one resident wave of streamers in compact storage");
        h->S = (N + h->R - 1) / h->R;
        h->NG = (int)((h->S + NGP_GRP - 1) / NGP_GRP);
        h->streamer = 3;
        h->nchain = NGP_ROWS_NW;
        // delay line: 8 VGPRs per lag and update task of a lane (1, 2 or 4 tasks: ngp_u8_tasks)
        const int nt = ngp_u8_tasks((int)h->R);
        const int want = h->lag_auto ? 8 : h->lag;
        int D;  // the instantiated lags (ngp_sweep.h, variant 3): the largest one not above the request
        if (nt == 1) D = want >= 12 ? 12 : (want >= 8 ? 8 : (want >= 6 ? 6 : (want >= 4 ? 4 : 3)));
        else if (nt == 2) D = want >= 8 ? 8 : 4;
        else D = 4;
        h->D = D;
        h->near = h->near_req ? h->near_req : ((h->R > 128) ? 2 : 3);
    } else {
    if (h->mode == 1) {
        max_shards = h->cu_count - 1 - (h->cu_count + NGP_GRP - 1) / NGP_GRP;
        if (h->max_shards_req > 0) max_shards = std::min<int64_t>(max_shards, h->max_shards_req);
        // Taller than one resident wave of 256-row shards: every streamer workgroup owns V = 2 (lag 3) or 3 (lag 2) shards of at
        // most 224 rows (role_streamer_rows_tall) -- S = V W shards, W workgroups, 1 + ceil(V W / 32) + W <= CUs.
        int tallV = 0;
        int64_t w_max = 0;
        for (int v = 2; v <= 3 && !tallV; v++) {
            if (h->streamer_req != 0 && h->streamer_req != 2 * v) continue;
            if (h->streamer_req == 0 && N <= max_shards * 256) continue;
            int64_t w = h->cu_count - 1;
            while (w > 1 && 1 + (v * w + NGP_GRP - 1) / NGP_GRP + w > h->cu_count) w--;
            if (h->max_shards_req > 0) w = std::max<int64_t>(1, std::min<int64_t>(w, h->max_shards_req / v));
            if (N <= v * w * NGP_ROWS_MAX_R && h->lag >= 3) { tallV = v; w_max = w; }
        }
        if (tallV) {
            choose_layout(N, tallV * w_max, NGP_ROWS_MAX_R, &h->R, &h->S, h->max_shards_req <= 0);
            h->S = (h->S + tallV - 1) / tallV * tallV;  // (all-padding shards at the end if need be: zero tiles, zero rows of ycorr)
            h->V = tallV;
        } else if (N > max_shards * 256) h->mode = 0;  // too many rows for one resident wave of streamers (2 LDS tile slots + partials)
        else choose_layout(N, max_shards, 256, &h->R, &h->S, h->max_shards_req <= 0);  // 8 R / 4 update tasks <= 512 threads, two 1040 R / 4 byte LDS slots
    }
    if (h->mode == 0) choose_layout(N, 256, 508, &h->R, &h->S);  // LDS bound of k_step: R*264 + 4096 <= 160 KiB
    h->NG = (int)((h->S + NGP_GRP - 1) / NGP_GRP);
    h->D = (h->mode == 1) ? std::min(h->lag, 8) : 1;
    // streamer variant (ngp_sweep.h): the row-owning waves serve shards of up to NGP_ROWS_MAX_R rows at lags 3..6 and are the
    // default from 64-row shards on
    h->streamer = 1;
    // (from 64-row shards on since the publisher stopped waiting for the block's barrier: 20k x 100k 3.66 -> 3.26 ms, 28k x 100k
    // 3.97 -> 3.45, 16k x 100k 3.38 -> 3.16, equal at 52-60 rows, the phase streamer ahead at 44 rows: 1.86 against 1.99 us per block)
    if (h->mode == 1 && h->R <= NGP_ROWS_MAX_R && h->lag >= 3 && (h->streamer_req == 2 || (h->streamer_req == 0 && h->R >= 64))) h->streamer = 2;
    if (h->V > 1) h->streamer = 2;
    h->nchain = (h->streamer == 2) ? NGP_ROWS_NW : 8;
    if (h->streamer == 2) {
        if (h->D > 6) h->D = 6;  // register delay line: 32 VGPRs per lag
        if (h->V > 1) h->D = (h->V == 2) ? 3 : 2;  // ... and per shard of the workgroup
    } else if (h->mode == 1 && h->R > 128 && h->D > 5) h->D = 5;  // tall shards: the register delay line holds 5 tiles at most
    // short shards (phase streamer), lag left to the library: 6.  Lag 8 was the better one while the shard partials crossed two hops (rounds 1-3);
    // with the one-hop fixed-point sums: 10k x 100k 2.66-2.71 ms at lag 6 against 2.77-2.81 at lag 8 (7: 2.82-2.91, 5: 3.16-3.21, 4: 3.02-3.16),
    // 8k x 100k 2.64 / 2.70, 14k x 100k 2.83 / 2.87, eight chains per pass 1839 / 1775 it/s (tools/r4_run41.sh)
    else if (h->mode == 1 && h->lag_auto && h->D > 6) h->D = 6;
    // a fourth near lag overloads the sampler CU at short shards (+17 % time at 10k x 100k); the phase streamer of tall shards,
    // where with lag 5 nothing is left for the reducers then, saves 8 % with it; with the row-owning streamer (lag 6) the sampler
    // CU is again the busier end (its Gram traffic: 32 KB per near lag and block) and three near lags measure better
    // (row-owning streamer on tall shards: two near lags measured 1.5 % better still -- the far path is one hop since dlt travels as granules)
    h->near = h->near_req ? h->near_req : ((h->mode == 1 && h->streamer == 2 && h->R >= 64) ? 2 : ((h->mode == 1 && h->R > 128) ? 4 : 3));
    }
    }
    // the sampler adds more than 8 group sums only where it fetches them one block ahead (lags 2-3: fetch_group_sums)
    if (h->mode == 1 && !owner) REQUIRE(h->NG <= 8 || h->D <= 3, NGP_ERR_STATE, "internal: more shard groups than the sampler adds");
    h->NBLK = (P + NGP_BLK - 1) / NGP_BLK;
    h->Ppad = h->NBLK * NGP_BLK;
    h->L = h->R * h->S;
    h->lds_step = (size_t)h->R * 264 + 4096;
    int rc;
    size_t tile_elems = (size_t)h->R * NGP_BLK;
    const size_t pp = (size_t)h->Ppad;
    if (owner) {
        h->d_tiles = owner->d_tiles; h->d_mean = owner->d_mean; h->d_gramx = owner->d_gramx; h->d_mpm = owner->d_mpm;
        h->mpm_max = owner->mpm_max;
        h->pm = owner->pm; h->pm->refs.fetch_add(1);
    } else {
    // column means: what the analytic centring of the compact storage uses; kept for the fp32 tiles too (ngp_get_storage: a host
    // can then rebuild any centred row of the panel from the genotype codes)
    if ((rc = dalloc(h, &h->d_mean, (size_t)h->Ppad))) return rc;
    if (h->storage == 1) {  // one byte per element (R is a multiple of 16), held behind the same pointer
        if ((rc = dalloc(h, &h->d_tiles, (size_t)h->NBLK * h->S * tile_elems / 4))) return rc;
    } else if ((rc = dalloc(h, &h->d_tiles, (size_t)h->NBLK * h->S * tile_elems))) return rc;
    if ((rc = dalloc(h, &h->d_gramx, (size_t)h->NBLK * h->D * NGP_BLK * NGP_BLK))) return rc;
    if ((rc = dalloc(h, &h->d_mpm, pp))) return rc;
    h->pm = new PanelMem();
    h->pm->tiles = h->d_tiles; h->pm->mean = h->d_mean; h->pm->gramx = h->d_gramx; h->pm->mpm = h->d_mpm;
    }
    if ((rc = dalloc(h, &h->d_lhs0, pp))) return rc;
    if ((rc = dalloc(h, &h->d_rhs0, pp))) return rc;
    if ((rc = dalloc(h, &h->d_beta, pp))) return rc;
    if ((rc = dalloc(h, &h->d_c, pp))) return rc;
    if ((rc = dalloc(h, &h->d_w, pp))) return rc;
    if ((rc = dalloc(h, &h->d_q, pp))) return rc;
    if ((rc = dalloc(h, &h->d_T, pp))) return rc;
    if ((rc = dalloc(h, &h->d_chi, pp))) return rc;
    if ((rc = dalloc(h, &h->d_setof, pp))) return rc;
    if ((rc = dalloc(h, &h->d_loc, pp))) return rc;
    if ((rc = dalloc(h, &h->d_vbidx, pp))) return rc;
    if ((rc = dalloc(h, &h->d_delta, pp))) return rc;
    if ((rc = dalloc(h, &h->d_sum_beta, pp))) return rc;
    if ((rc = dalloc(h, &h->d_sum_beta2, pp))) return rc;
    if ((rc = dalloc(h, &h->d_sum_delta, pp))) return rc;
    if ((rc = dalloc(h, &h->d_ycorr, (size_t)h->L))) return rc;
    if ((rc = dalloc(h, &h->d_part, (size_t)h->S * NGP_BLK))) return rc;
    if ((rc = dalloc(h, &h->d_dlt, NGP_BLK))) return rc;
    if ((rc = dalloc(h, &h->d_sets, 16))) return rc;
    if ((rc = dalloc(h, &h->d_scal, 1))) return rc;
    HCHK(hipMemsetAsync(h->d_setof, 0xFF, pp, h->stream));
    HCHK(hipMemsetAsync(h->d_delta, 1, pp, h->stream));
    h->h_setof.assign(pp, -1);
    h->h_loc.assign(pp, 0);
    h->h_vbidx.assign(pp, 0);
    h->sets.clear(); h->nvb = 0; h->h_regs.clear(); h->h_seg_k0.clear(); h->h_seg_len.clear(); h->h_seg_set.clear(); h->nclass_total = 0;
    dfree(h->d_rcls);
    dfree(h->d_tup); dfree(h->d_tupc); dfree(h->d_tupg); h->h_tregs.clear(); h->h_tseg_l0.clear(); h->h_tseg_len.clear(); h->h_tseg_set.clear(); h->ntuple = 0;
    dfree(h->d_varBeta); dfree(h->d_sum_varBeta); h->vb_cap = 0;
    // a new panel is a new model: the fixed-effect sets (N rows of the OLD panel) and the trace selection (loci of the old P) go
    // with the marker sets -- k_fixed would read d_X of the old N, k_post beta[loci[k]] beyond the new P
    for (auto &fx : h->fix) { dfree(fx.d_X); dfree(fx.d_xpx0); dfree(fx.d_xpxR); dfree(fx.d_lhs0); dfree(fx.d_rhs0); }
    h->fix.clear(); h->nfixcol = 0; dfree(h->d_bfix); dfree(h->d_sum_bfix);
    dfree(h->d_trace_loci); dfree(h->d_tr_beta); dfree(h->d_tr_vb); dfree(h->d_tr_pi);
    h->ntl = 0; h->ntvb = 0; h->trace_ext_cap = 0;
    h->have_y = false; h->iter = 0; h->poisoned = false; h->panel_open = false;
    if (h->storage == 0) HCHK(hipFuncSetAttribute((const void *)k_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_step));
    if (h->mode == 1) {
        const size_t lds_sampler = (size_t)(3 * 4096 + 2 * NGP_RING * NGP_BLK + 6 * NGP_BLK) * sizeof(double) + 2 * NGP_BLK * sizeof(int) + 320 + NGP_SAMPLER_TUPLE_LDS;
        const size_t lds_max = 160 * 1024;
        const size_t misc = (size_t)h->R * 16 + 4096 + 2 * 512 + 128 + 3072 + (size_t)h->R * 64;
        const size_t TB = (size_t)(h->R / 4) * NGP_QS;  // LDS footprint of one tile (quads NGP_QS bytes apart)
        h->lds_sweep = std::max(2 * TB + misc, lds_sampler);
        if (2 * TB + misc + 8192 <= lds_max) h->lds_sweep = std::max(h->lds_sweep, 2 * TB + misc + 8192);  // room for the diagnostic timeline
        if (h->streamer >= 2) {  // ring of 2 NQ + H slots | shard | 2 x 7 x 64 chain partials | 2 x 72 dlt | 2 x 8 row sums | flags | 1 KiB sink
            const size_t nq = (h->streamer == 3) ? (size_t)h->R / 16 : (size_t)h->R / 4, hq = std::min<size_t>(NGP_ROWS_HMAX, (nq + 1) / 2);
            const size_t need = (2 * nq + hq) * NGP_QS + (size_t)h->V * ((h->R + 7) & ~7) * 8 + 2 * NGP_ROWS_NW * NGP_BLK * 8 + 2 * NGP_DLS * 8 + 16 * 8 + 64 + 1024;
            h->lds_rows = need;
            h->lds_sweep = std::max(need, lds_sampler);
            if (need + 8192 <= lds_max) h->lds_sweep = std::max(h->lds_sweep, need + 8192);
        }
        {   // the lean kernel (models of BayesPR / BayesB / BayesC sets) has no use for the sampler's BayesR / Tuple staging area
            const size_t lean_sampler = lds_sampler - NGP_SAMPLER_TUPLE_LDS;
            const size_t streamer_need = (h->streamer >= 2) ? h->lds_rows : 2 * TB + misc;
            h->lds_sweep_lean = std::max(streamer_need, lean_sampler);
            if (streamer_need + 8192 <= lds_max) h->lds_sweep_lean = std::max(h->lds_sweep_lean, streamer_need + 8192);
        }
        {   // k_sweep_r: the sampler's BayesR staging area on top
            const size_t streamer_need = (h->streamer >= 2) ? h->lds_rows : 2 * TB + misc;
            size_t r = std::max(streamer_need, lds_sampler + (size_t)NGP_SAMPLER_R_LDS);
            if (streamer_need + 8192 <= lds_max) r = std::max(r, streamer_need + 8192);
            h->lds_sweep_r = (r <= lds_max) ? r : 0;
            if (h->lds_sweep_r) HCHK(sweep_r_set_max_lds((int)h->lds_sweep_r));
        }
        if (h->lds_sweep > lds_max) return fail(h, NGP_ERR_ARG, "panel too tall for the persistent sweep (LDS)");
        HCHK(sweep_set_max_lds_0((int)h->lds_sweep));
        HCHK(sweep_set_max_lds_1((int)h->lds_sweep));
        HCHK(sweep_tup_set_max_lds((int)h->lds_sweep));
        if (h->V > 1) HCHK(sweep_tall_set_max_lds((int)h->lds_sweep));
        // every workgroup of the persistent kernel waits for others: the whole grid must be resident at once, one workgroup
        // per CU.  Checked here, not assumed (a grid that does not fit would only show up as a spin timeout).
        int wg_per_cu = 0;
        if (h->V > 1) HCHK(sweep_tall_occupancy(&wg_per_cu, h->lds_sweep));
        else HCHK(sweep_occupancy_0(&wg_per_cu, h->lds_sweep));
        if (wg_per_cu < 1 || 1 + h->NG + h->S / h->V > (int64_t)wg_per_cu * h->cu_count)
            return fail(h, NGP_ERR_STATE, "persistent sweep: grid of " + std::to_string(1 + h->NG + h->S / h->V) + " workgroups cannot be co-resident (" +
                                              std::to_string(wg_per_cu) + " per CU x " + std::to_string(h->cu_count) + " CUs); use ngp_configure(mode 0)");
        if ((rc = dalloc(h, &h->d_cdlt, (size_t)NGP_RING * NGP_BLK))) return rc;
        if ((rc = dalloc(h, &h->d_cdltg, (size_t)NGP_RING * NGP_BLK * 2))) return rc;
        // fixed-point accumulators (RING x 8 copies x 64 x 8 bytes) | dlt flag (one line) | census counters (one line) | census table
        // (placement of each workgroup, 2 words each); all zeroed by k_prep
        h->census_off = (size_t)NGP_RING * NGP_FX_COPIES * NGP_BLK * 2 + 32;
        h->ccnt_words = h->census_off + 32 + 2 * (size_t)320;  // (320 >= any grid, also the fused grid of K chains per pass)
        if ((rc = dalloc(h, &h->d_ccnt, h->ccnt_words))) return rc;
        h->d_census_tbl = (unsigned long long *)(h->d_ccnt + h->census_off + 32);
    }
    if ((rc = dalloc(h, &h->d_abort, 32))) return rc;
    HCHK(hipStreamSynchronize(h->stream));
    return NGP_OK;
}

// max_j x_j'x_j: with ycorr'ycorr it bounds every X_t'ycorr (the scale of the fixed-point accumulators, k_head)
int refresh_mpm_max(ngp_handle *h) {
    std::vector<double> m((size_t)h->Ppad);
    HCHK(hipMemcpy(m.data(), h->d_mpm, (size_t)h->Ppad * sizeof(double), hipMemcpyDeviceToHost));
    double mx = 0.0;
    for (double v : m) if (v > mx) mx = v;
    h->mpm_max = mx;
    return NGP_OK;
}

int build_gram8(ngp_handle *h) {  // compact storage: exact integer dot products, then G = dot - N (m_k m_j)
    const size_t per_block = (size_t)h->S * NGP_BLK * NGP_BLK * sizeof(uint32_t);
    int nb_max = (int)std::max<size_t>(1, std::min<size_t>((size_t)h->NBLK, ((size_t)1 << 30) / per_block));
    nb_max = std::min(nb_max, 32768);
    uint32_t *d_gpart = nullptr;
    int rc;
    if ((rc = dalloc(h, &d_gpart, (size_t)nb_max * h->S * NGP_BLK * NGP_BLK))) return rc;
    for (int d = 0; d < h->D; d++)
        for (int64_t t0 = 0; t0 < h->NBLK; t0 += nb_max) {
            int nb = (int)std::min<int64_t>(nb_max, h->NBLK - t0);
            hipLaunchKernelGGL(k_gram8_part, dim3((unsigned)h->S, (unsigned)nb), dim3(256), 0, h->stream, (const uint8_t *)h->d_tiles, d_gpart,
                               (int)h->R, (int)h->S, (int)t0, d);
            long long ne = (long long)nb * NGP_BLK * NGP_BLK;
            hipLaunchKernelGGL(k_gram8_reduce, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, h->stream, d_gpart, h->d_gramx, h->d_mpm,
                               h->d_mean, (long long)h->N, (int)h->S, (int)t0, nb, d, h->D);
        }
    hipError_t e = hipStreamSynchronize(h->stream);
    dfree(d_gpart);
    if (e != hipSuccess) return fail(h, NGP_ERR_HIP, std::string("gram: ") + hipGetErrorString(e));
    e = hipGetLastError();
    if (e != hipSuccess) return fail(h, NGP_ERR_HIP, std::string("gram launch: ") + hipGetErrorString(e));
    return refresh_mpm_max(h);
}

int build_gram(ngp_handle *h) {
    if (h->storage == 1) return build_gram8(h);
    // batches of blocks so the shard-partial scratch stays <= ~1 GiB
    const size_t per_block = (size_t)h->S * NGP_BLK * NGP_BLK * sizeof(double);
    int nb_max = (int)std::max<size_t>(1, std::min<size_t>((size_t)h->NBLK, ((size_t)1 << 30) / per_block));
    nb_max = std::min(nb_max, 32768);
    double *d_gpart = nullptr;
    int rc;
    if ((rc = dalloc(h, &d_gpart, (size_t)nb_max * h->S * NGP_BLK * NGP_BLK))) return rc;
    for (int d = 0; d < h->D; d++)
        for (int64_t t0 = 0; t0 < h->NBLK; t0 += nb_max) {
            int nb = (int)std::min<int64_t>(nb_max, h->NBLK - t0);
            if (h->gram_engine == 0)  // fp64 VALU contraction: the default (1.63 ms per launch at 50k x 600k against 1.84 on the matrix cores)
                hipLaunchKernelGGL(k_gram_part, dim3((unsigned)h->S, (unsigned)nb), dim3(256), 0, h->stream, h->d_tiles, d_gpart, (int)h->R,
                                   (int)h->S, (int)t0, d);
            else                      // matrix cores (ngp_debug_set_knob bit 10 before the panel is set): v_mfma_f64_16x16x4_f64, the same sums in the same order
                hipLaunchKernelGGL(k_gram_part_mfma, dim3((unsigned)h->S, (unsigned)nb), dim3(256), 0, h->stream, h->d_tiles, d_gpart, (int)h->R,
                                   (int)h->S, (int)t0, d);
            long long ne = (long long)nb * NGP_BLK * NGP_BLK;
            hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, h->stream, d_gpart, h->d_gramx, h->d_mpm,
                               (int)h->S, (int)t0, nb, d, h->D);
        }
    hipError_t e = hipStreamSynchronize(h->stream);
    dfree(d_gpart);
    if (e != hipSuccess) return fail(h, NGP_ERR_HIP, std::string("gram: ") + hipGetErrorString(e));
    e = hipGetLastError();
    if (e != hipSuccess) return fail(h, NGP_ERR_HIP, std::string("gram launch: ") + hipGetErrorString(e));
    return refresh_mpm_max(h);
}

// ---- host panels in Float64 / Float32, whole or in column ranges (ngp_begin_panel / ngp_panel_columns_* / ngp_end_panel) ----
int begin_panel(ngp_handle *h, int64_t N, int64_t P) {
    int rc;
    if ((rc = enter(h))) return rc;
    if ((rc = alloc_panel(h, N, P))) return rc;  // (tiles and means are born zero: columns never uploaded stay zero columns)
    h->panel_open = true;
    return NGP_OK;
}

// columns [col0, col0 + ncol) from a column-major host matrix: staged through the device in chunks of whole columns (256 MiB)
template <typename TIn>
int panel_columns(ngp_handle *h, int64_t col0, const TIn *M, int64_t ncol, int64_t ld, int centre) {
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->panel_open && h->d_tiles != nullptr, NGP_ERR_STATE, "ngp_panel_columns_* needs an open panel (ngp_begin_panel)");
    REQUIRE(M != nullptr, NGP_ERR_ARG, "null panel pointer");
    REQUIRE(ld >= h->N, NGP_ERR_ARG, "leading dimension smaller than N");
    REQUIRE(col0 >= 0 && ncol > 0 && col0 + ncol <= h->P, NGP_ERR_ARG, "column range outside the panel");
    REQUIRE(h->storage == 0 || sizeof(TIn) == 1, NGP_ERR_ARG,
            "compact storage takes genotype codes: ngp_panel_columns_u8, ngp_set_panel_u8, ngp_load_panel_file or ngp_generate_panel");
    const int64_t N = h->N;
    const int64_t cchunk = std::max<int64_t>(1, std::min<int64_t>(ncol, ((int64_t)256 << 20) / (int64_t)(ld * sizeof(TIn))));
    TIn *d_g = nullptr;
    unsigned *d_bad = nullptr;
    if (hipMalloc((void **)&d_g, (size_t)cchunk * ld * sizeof(TIn)) != hipSuccess) return fail(h, NGP_ERR_NOMEM, "staging buffer");
    if ((rc = dalloc(h, &d_bad, 1))) { (void)hipFree(d_g); return rc; }
    hipError_t e = hipSuccess;
    for (int64_t c0 = 0; c0 < ncol && e == hipSuccess; c0 += cchunk) {
        const int64_t nc = std::min<int64_t>(cchunk, ncol - c0);
        // the last column may be shorter than ld in the caller's buffer: nc - 1 full columns + N elements
        e = hipMemcpyAsync(d_g, M + (size_t)c0 * ld, ((size_t)(nc - 1) * ld + (size_t)N) * sizeof(TIn), hipMemcpyHostToDevice, h->stream);
        if (e != hipSuccess) break;
        double *d_mu = h->d_mean + col0 + c0;
        hipLaunchKernelGGL(k_cols_mean<TIn>, dim3((unsigned)((nc + 63) / 64)), dim3(64), 0, h->stream, (const TIn *)d_g, (long long)N, (long long)ld,
                           (long long)nc, centre, d_mu, d_bad);
        if constexpr (sizeof(TIn) == 1) {
            if (h->storage == 1)  // the codes stay codes (the means are what the analytic centring uses)
                hipLaunchKernelGGL(k_cols_fill8, dim3((unsigned)((h->L / 16 + 255) / 256), (unsigned)nc), dim3(256), 0, h->stream, (uint8_t *)h->d_tiles,
                                   (const uint8_t *)d_g, (long long)N, (long long)ld, (long long)(col0 + c0), (int)h->R, (int)h->S);
        }
        if (h->storage == 0)
        hipLaunchKernelGGL(k_cols_fill<TIn>, dim3((unsigned)((h->L / 4 + 255) / 256), (unsigned)nc), dim3(256), 0, h->stream, h->d_tiles, (const TIn *)d_g,
                           (long long)N, (long long)ld, (long long)(col0 + c0), (int)h->R, (int)h->S, (const double *)d_mu);
        e = hipStreamSynchronize(h->stream);  // the staging buffer is reused by the next chunk
    }
    unsigned bad = 0;
    if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, sizeof(unsigned), hipMemcpyDeviceToHost);
    (void)hipFree(d_g);
    dfree(d_bad);
    if (e != hipSuccess) return fail(h, NGP_ERR_HIP, std::string("panel columns: ") + hipGetErrorString(e));
    REQUIRE(bad == 0u, NGP_ERR_ARG, "non-finite genotype value in panel");
    return NGP_OK;
}

int end_panel(ngp_handle *h) {
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->panel_open && h->d_tiles != nullptr, NGP_ERR_STATE, "no open panel (ngp_begin_panel)");
    h->panel_open = false;
    return build_gram(h);
}

template <typename TIn>
int set_panel_host(ngp_handle *h, const TIn *M, int64_t N, int64_t P, int64_t ld, int centre) {
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(M != nullptr, NGP_ERR_ARG, "null panel pointer");
    REQUIRE(ld >= N, NGP_ERR_ARG, "leading dimension smaller than N");
    if ((rc = begin_panel(h, N, P))) return rc;
    if ((rc = panel_columns<TIn>(h, 0, M, P, ld, centre))) return rc;
    return end_panel(h);
}

// does any block of this model take the inverse form?  (a BayesPR set must exist: blocks without an owner are all zeros either way)
bool wants_tinv(const ngp_handle *h) {
    if (h->chain_form != 1) return false;
    for (const HSet &st : h->sets)
        if (st.method == NGP_METHOD_BAYESPR || st.method == NGP_METHOD_TUPLE) return true;
    return false;
}

// Which blocks are linear (every lane BayesPR, unowned or -- in a fine-seam call, active_set >= 0 -- of another set than the sampled
// one: those lanes are inactive): their chain takes the inverse form (k_tinv).  Static for a model; the table goes to the device when
// the model or the active set changed.  A Tuple set owns its blocks to the end of the last one (ncol = its span).
int sync_linear_blocks(ngp_handle *h, int active_set) {
    if (!wants_tinv(h) || h->tinv_blocks != h->NBLK) { h->lin_all = 0; h->lin_any = 0; return NGP_OK; }
    if (h->blin_for == active_set) return NGP_OK;
    // 1 = BayesPR / unowned / inactive lanes only; 1 + k = a block of a k-set Tuple (its chain is linear too: one step per locus);
    // 0 = a lane of BayesB / BayesC / BayesR: the step chains
    std::vector<unsigned> bl((size_t)h->NBLK, 1u);
    for (size_t si = 0; si < h->sets.size(); si++) {
        const HSet &st = h->sets[si];
        if (st.method == NGP_METHOD_BAYESPR || (active_set >= 0 && (int)si != active_set)) continue;
        const unsigned code = (st.method == NGP_METHOD_TUPLE) ? 1u + (unsigned)st.tk : 0u;
        for (int64_t t = st.col0 / NGP_BLK; t <= (st.col0 + st.ncol - 1) / NGP_BLK && t < h->NBLK; t++) bl[(size_t)t] = code;
    }
    int64_t n1 = 0, nany = 0;
    for (unsigned v : bl) { n1 += (v == 1u); nany += (v != 0u); }
    h->lin_all = (n1 == h->NBLK) ? 1 : 0;
    h->lin_any = (nany > 0) ? 1 : 0;
    HCHK(hipMemcpyAsync(h->d_blin, bl.data(), (size_t)h->NBLK * sizeof(unsigned), hipMemcpyHostToDevice, h->stream));
    HCHK(hipStreamSynchronize(h->stream));  // (bl is a local)
    h->blin_for = active_set;
    return NGP_OK;
}

int sync_tables(ngp_handle *h) {
    int rc;
    if (wants_tinv(h) && h->tinv_blocks != h->NBLK) {
        if ((rc = dalloc(h, &h->d_tinv, (size_t)h->NBLK * NGP_BLK * NGP_BLK))) return rc;
        if ((rc = dalloc(h, &h->d_blin, (size_t)h->NBLK))) return rc;
        h->tinv_blocks = h->NBLK;
        h->blin_for = -2;
    }
    if (h->tables_dirty) h->blin_for = -2;
    if (!h->tables_dirty) return NGP_OK;
    const size_t pp = (size_t)h->Ppad;
    HCHK(hipMemcpy(h->d_setof, h->h_setof.data(), pp, hipMemcpyHostToDevice));
    HCHK(hipMemcpy(h->d_loc, h->h_loc.data(), pp * sizeof(int32_t), hipMemcpyHostToDevice));
    HCHK(hipMemcpy(h->d_vbidx, h->h_vbidx.data(), pp * sizeof(int32_t), hipMemcpyHostToDevice));
    if ((rc = dalloc(h, &h->d_regs, h->h_regs.size()))) return rc;
    if ((rc = dalloc(h, &h->d_seg_k0, h->h_seg_k0.size()))) return rc;
    if ((rc = dalloc(h, &h->d_seg_len, h->h_seg_len.size()))) return rc;
    if ((rc = dalloc(h, &h->d_segpart, h->h_seg_k0.size()))) return rc;
    if ((rc = dalloc(h, &h->d_seg_set, h->h_seg_set.size()))) return rc;
    if ((rc = dalloc(h, &h->d_regchi, h->h_regs.size()))) return rc;
    if (!h->h_regs.empty()) {
        HCHK(hipMemcpy(h->d_regs, h->h_regs.data(), h->h_regs.size() * sizeof(DReg), hipMemcpyHostToDevice));
        HCHK(hipMemcpy(h->d_seg_k0, h->h_seg_k0.data(), h->h_seg_k0.size() * sizeof(long long), hipMemcpyHostToDevice));
        HCHK(hipMemcpy(h->d_seg_len, h->h_seg_len.data(), h->h_seg_len.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HCHK(hipMemcpy(h->d_seg_set, h->h_seg_set.data(), h->h_seg_set.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    if (!h->h_tregs.empty()) {
        if ((rc = dalloc(h, &h->d_tregs, h->h_tregs.size()))) return rc;
        if ((rc = dalloc(h, &h->d_tseg_l0, h->h_tseg_l0.size()))) return rc;
        if ((rc = dalloc(h, &h->d_tseg_len, h->h_tseg_len.size()))) return rc;
        if ((rc = dalloc(h, &h->d_tseg_set, h->h_tseg_set.size()))) return rc;
        if ((rc = dalloc(h, &h->d_tsegpart, h->h_tseg_l0.size() * NGP_TPAIRS))) return rc;
        HCHK(hipMemcpy(h->d_tregs, h->h_tregs.data(), h->h_tregs.size() * sizeof(DTReg), hipMemcpyHostToDevice));
        HCHK(hipMemcpy(h->d_tseg_l0, h->h_tseg_l0.data(), h->h_tseg_l0.size() * sizeof(long long), hipMemcpyHostToDevice));
        HCHK(hipMemcpy(h->d_tseg_len, h->h_tseg_len.data(), h->h_tseg_len.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HCHK(hipMemcpy(h->d_tseg_set, h->h_tseg_set.data(), h->h_tseg_set.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    h->tables_dirty = false;
    return NGP_OK;
}

bool is_kept(const ngp_handle *h, int64_t it) {  // src/samplers.jl:26
    if (it < h->burnIn + h->thin || it > h->chainLength) return false;
    return ((it - h->burnIn) % h->thin) == 0;
}

// one sweep over blocks [tb0, tb1): persistent kernel (mode 1) or two launches per block (mode 0)
// launch arguments of the persistent sweep over blocks [tb0, tb1) of this handle's chain (advances the launch nonce)
void fill_sweep_args(ngp_handle *h, int64_t tb0, int64_t tb1, SweepArgs &A) {
    const int R = (int)h->R, S = (int)h->S;
    A.tiles = h->d_tiles; A.ycorr = h->d_ycorr; A.gramx = h->d_gramx;
    const bool tf = wants_tinv(h) && h->tinv_blocks == h->NBLK;
    A.tinv = (tf && h->lin_any) ? h->d_tinv : nullptr;
    A.blin = (tf && h->lin_any) ? h->d_blin : nullptr;
    A.lin_all = (tf && h->lin_any) ? h->lin_all : 0;
    A.V = h->V; A.D = h->D; A.R = R; A.S = S; A.NG = h->NG; A.near = h->near; A.fine_ok = 0; A.t0 = (int)tb0; A.t1 = (int)tb1;
    A.beta = h->d_beta; A.delta = h->d_delta; A.c = h->d_c; A.w = h->d_w; A.q = h->d_q; A.mpm = h->d_mpm; A.chi = h->d_chi;
    A.setof = h->d_setof; A.vbidx = h->d_vbidx; A.sets = h->d_sets; A.varBeta = h->d_varBeta;
    A.rcls = h->d_rcls; A.rhs0 = h->d_rhs0; A.scal = h->d_scal; A.Ppad = h->Ppad;
    A.tup = h->ntuple ? h->d_tup : nullptr; A.tupc = h->d_tupc; A.tupg = h->d_tupg;
    A.acc = (unsigned long long *)h->d_ccnt; A.dlt = h->d_cdlt; A.dltg = h->d_cdltg;
    h->launch_seq = (h->launch_seq % 4095u) + 1u;  // 1..4095: never the zero the ring is born with
    A.nonce = h->launch_seq;
    A.flag_dlt = h->d_ccnt + (size_t)NGP_RING * NGP_FX_COPIES * NGP_BLK * 2; A.abort_w = h->d_abort; A.xcc_w = h->d_abort + 16;
    A.census = (h->dbg_mode == 0) ? h->d_ccnt + h->census_off : nullptr;  // timing modes leave roles out: no census there
    A.census_tbl = h->d_census_tbl; A.iter_tag = (unsigned)(h->iter + 1);
    A.census_fail = (h->dbg_census_fail_iter > 0 && !h->exclusive) ? (unsigned)h->dbg_census_fail_iter : 0u;
    A.dbg = h->d_dbg;
    A.fine_ok = ((size_t)2 * (R / 4) * NGP_QS + (size_t)R * 80 + 8320 + 8192 <= h->lds_sweep) ? 1 : 0;  // diagnostic timeline fits in LDS
    A.variant = h->streamer; A.knob = h->knob;
    if (h->streamer >= 2) A.fine_ok = (h->lds_rows + 8192 <= h->lds_sweep) ? 1 : 0;
    A.mean = h->d_mean; A.N = h->N;
    A.dbg_mode = h->dbg_mode;
}

// one sweep over blocks [tb0, tb1): persistent kernel (mode 1) or two launches per block (mode 0)
void launch_sweep(ngp_handle *h, int64_t tb0, int64_t tb1, hipEvent_t *evs) {
    const int R = (int)h->R, S = (int)h->S;
    if (h->mode == 1) {
        // (the hand-off counters were zeroed by k_prep, which precedes every sweep in the stream)
        SweepArgs A;
        fill_sweep_args(h, tb0, tb1, A);
        h->last_grid = 1 + h->NG + S / h->V;
        if (evs) (void)hipEventRecord(evs[0], h->stream);
        if (h->V > 1)  // several shards per streamer workgroup: a kernel of its own (no diagnostics there)
            sweep_tall_launch((unsigned)h->last_grid, h->lds_sweep, h->stream, A);
        else if (h->d_dbg || h->dbg_mode)  // diagnostic instantiation: stamps and timing modes exist only there
            sweep_launch_1((unsigned)h->last_grid, h->lds_sweep, h->stream, A);
        else if (h->nclass_total > 0 && h->lds_sweep_r && !(h->knob & 65536))  // a BayesR set: the flavour that fetches its coefficients ahead
            sweep_r_launch((unsigned)h->last_grid, h->lds_sweep_r, h->stream, A);
        // Models with a Tuple or a BayesR set run the kernel that carries those chains (k_sweep<false>, the kernel of BayesPR / BayesB /
        // BayesC, does not: ngp_sweep.h, role_sampler).  Every other model takes the lean kernel at every shape (in round 3 tall fp32
        // shards ran 1.7-2 % faster in the full kernel -- register allocation, not design; since the round-4 hand-off the lean
        // kernel is level or ahead there too: 23.4-23.6 against 23.3-23.9 ms per iteration at 50k x 600k).  Knob bit 14 forces the
        // full kernel.
        else if (h->ntuple > 0 || h->nclass_total > 0 || (h->knob & 16384))
            sweep_tup_launch((unsigned)h->last_grid, h->lds_sweep, h->stream, A);
        else
            sweep_launch_0((unsigned)h->last_grid, h->lds_sweep_lean, h->stream, A);
        if (evs) (void)hipEventRecord(evs[1], h->stream);
        h->sweep_launches += 1;
        return;
    }
    int e = 0;
    for (int64_t t = tb0; t <= tb1; t++) {
        const int do_upd = t > tb0, do_gemv = t < tb1;
        if (evs && do_gemv) (void)hipEventRecord(evs[e++], h->stream);
        hipLaunchKernelGGL(k_step, dim3((unsigned)S), dim3(256), h->lds_step, h->stream, h->d_tiles, h->d_ycorr, h->d_dlt, h->d_part, R,
                           S, (int)t, do_upd, do_gemv);
        if (evs && do_gemv) (void)hipEventRecord(evs[e++], h->stream);
        if (do_gemv)
            hipLaunchKernelGGL(k_recur, dim3(1), dim3(256), 0, h->stream, h->d_part, h->d_gramx, h->D, S, (int)t, h->d_beta, h->d_delta,
                               h->d_c, h->d_w, h->d_q, h->d_mpm, h->d_chi, h->d_setof, h->d_vbidx, h->d_sets, h->d_varBeta, h->d_dlt, h->d_rcls,
                               (long long)h->Ppad, h->d_rhs0, h->d_scal, h->d_tup, h->d_tupc, h->d_tupg,
                               (const double *)((wants_tinv(h) && h->tinv_blocks == h->NBLK && h->lin_any) ? h->d_tinv : nullptr), (const unsigned *)h->d_blin);
    }
    h->sweep_launches += 2 * (tb1 - tb0) + 1;
}

// placement census of the last sweep launch (SweepArgs.census_tbl): who arrived, and where
std::string census_report(ngp_handle *h) {
    const size_t grid = (size_t)(h->last_grid > 0 ? h->last_grid : 1 + h->NG + h->S / h->V);
    std::vector<unsigned long long> tb(grid, 0ull);
    if (!h->d_census_tbl || hipMemcpy(tb.data(), h->d_census_tbl, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return "(no census)";
    int per_xcc[16] = {0}, per_se[16][8] = {{0}};
    size_t arrived = 0;
    std::string missing;
    for (size_t b = 0; b < grid; b++) {
        if (tb[b] == 0ull) { if (missing.size() < 120) missing += (missing.empty() ? "" : ",") + std::to_string(b); continue; }
        arrived++;
        const unsigned x = ((unsigned)(tb[b] >> 32) - 1u) & 15u, hw = (unsigned)tb[b];
        per_xcc[x]++; per_se[x][(hw >> 13) & 7u]++;  // HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    }
    std::string r = std::to_string(arrived) + " of " + std::to_string(grid) + " workgroups resident; per XCD";
    for (int x = 0; x < 8; x++) {
        r += " " + std::to_string(per_xcc[x]) + "(";
        for (int e = 0; e < 4; e++) r += (e ? "/" : "") + std::to_string(per_se[x][e]);
        r += ")";
    }
    if (!missing.empty()) r += "; missing blocks " + missing;
    return r;
}

// status of the sweeps launched so far: NGP_OK, NGP_RETRY_CENSUS (a launch ended at its census with nothing changed: *iter_failed
// says which iteration; the abort words are cleared, the caller runs it again with the device to itself) or an error (poisoned)
#define NGP_RETRY_CENSUS 1
int check_abort(ngp_handle *h, int64_t *iter_failed = nullptr) {
    if (h->mode != 1) return NGP_OK;
    unsigned w[2] = {0, 0};
    HCHK(hipMemcpy(w, h->d_abort, sizeof(w), hipMemcpyDeviceToHost));
    if (w[0] == 0) return NGP_OK;
    (void)hipMemset(h->d_abort, 0, sizeof(w));
    if (w[0] == NGP_ABORT_CENSUS && iter_failed && !h->exclusive) {
        // iter_tag holds the low 32 bits of the iteration: the launches in flight are at most 16 iterations ahead of it
        const int64_t base = (h->iter + 1) & ~(int64_t)0xFFFFFFFF;
        int64_t it = base | (int64_t)w[1];
        if (it > h->iter + 1) it -= ((int64_t)1 << 32);
        *iter_failed = it;
        return NGP_RETRY_CENSUS;
    }
    h->poisoned = true;  // ycorr / beta were left half-way through a sweep (or a grid was not resident even alone on the device)
    if (w[0] == NGP_ABORT_CENSUS)
        return fail(h, NGP_ERR_HIP, "persistent sweep: the grid did not become resident although this call had leased the whole device (" +
                                        census_report(h) + "): another process holding CUs?  The chain state is invalid until ngp_set_y / ngp_set_state");
    if (w[0] == 7u)  // NGP_ABORT_FX
        return fail(h, NGP_ERR_HIP, "persistent sweep: a partial dot product left the fixed-point range of its accumulator (non-finite residual or effects, "
                                    "or the residual grew more than 32-fold within one sweep); the chain state is invalid until ngp_set_y / ngp_set_state");
    return fail(h, NGP_ERR_HIP, "persistent sweep kernel gave up waiting (role code " + std::to_string(w[0]) +
                                    "): workgroups not co-resident (another kernel holding CUs?) or a hand-off was lost; the chain "
                                    "state is invalid until ngp_set_y / ngp_set_state");
}

void launch_variance(ngp_handle *h, int active_set, uint64_t it) {
    const long long nseg = (long long)h->h_seg_k0.size(), nreg = (long long)h->h_regs.size();
    if (nseg > 0) {
        hipLaunchKernelGGL(k_regssq, dim3((unsigned)((nseg + 3) / 4)), dim3(256), 0, h->stream, nseg, h->d_seg_k0, h->d_seg_len,
                           h->d_beta, h->d_segpart, h->d_abort);
        if (h->nclass_total > 0)
            hipLaunchKernelGGL(k_rssq, dim3((unsigned)((nseg + 3) / 4)), dim3(256), 0, h->stream, nseg, h->d_seg_k0, h->d_seg_len, h->d_seg_set,
                               h->d_sets, h->d_beta, h->d_delta, h->d_segpart, h->d_abort);
        hipLaunchKernelGGL(k_regdraw, dim3((unsigned)((nreg + 63) / 64)), dim3(64), 0, h->stream, nreg, h->d_regs, h->d_segpart,
                           h->d_sets, h->d_varBeta, active_set, h->d_regchi, h->seed, (uint64_t)h->chain, it, h->d_abort);
    }
    if (!h->h_tregs.empty()) {  // Tuple sets: Sb = B_r'B_r per region, then the inverse-Wishart draw of its variance matrix
        const long long ntseg = (long long)h->h_tseg_l0.size(), ntreg = (long long)h->h_tregs.size();
        hipLaunchKernelGGL(k_tuple_ssq, dim3((unsigned)((ntseg + 3) / 4)), dim3(256), 0, h->stream, ntseg, h->d_tseg_l0, h->d_tseg_len, h->d_tseg_set,
                           h->d_tup, h->d_beta, h->d_tsegpart, h->d_abort);
        hipLaunchKernelGGL(k_tuple_draw, dim3((unsigned)((ntreg + 63) / 64)), dim3(64), 0, h->stream, ntreg, h->d_tregs, h->d_tsegpart, h->d_tup,
                           h->d_varBeta, active_set, h->seed, (uint64_t)h->chain, it, h->d_abort);
    }
    hipLaunchKernelGGL(k_pidraw, dim3(1), dim3(64), 0, h->stream, (int)h->sets.size(), h->d_sets, active_set, h->seed,
                       (uint64_t)h->chain, it, h->d_abort);
}

// resume_mid: the head of this iteration (varE, intercept, fixed-effect sets) has run already -- its sweep ended at the census
// with nothing changed and is launched again, k_prep first (it redraws the same keyed numbers and clears the hand-off counters)
int sample_enqueue(ngp_handle *h);  // (below)

// T = inv(L) of every linear block from this iteration's coefficients (k_tinv; behind k_prep in the stream, in front of the sweep)
void launch_tinv(ngp_handle *h) {  // (sync_linear_blocks has run for this call's active set)
    if (!wants_tinv(h) || h->tinv_blocks != h->NBLK || !h->lin_any) return;
    hipLaunchKernelGGL(k_tinv, dim3((unsigned)h->NBLK), dim3(64), 0, h->stream, (const double *)h->d_gramx, h->D, (const double *)h->d_c,
                       (const unsigned *)h->d_blin, h->d_tinv, (const unsigned *)h->d_abort, (const double *)h->d_tupc, (long long)h->Ppad);
}

void iteration_pre(ngp_handle *h, int64_t trace_idx, bool resume_mid) {  // everything in front of the sweep
    const uint64_t it = (uint64_t)(h->iter + 1);
    if (!resume_mid) {
    hipLaunchKernelGGL(k_head, dim3(1), dim3(1024), 0, h->stream, h->d_ycorr, (long long)h->L, (long long)h->N, h->d_scal, h->e_df,
                       h->e_scale, h->intercept, 1, h->seed, (uint64_t)h->chain, it, h->d_tr_varE, h->d_tr_b, (long long)trace_idx, h->d_abort, h->mpm_max);
    for (size_t f = 0; f < h->fix.size(); f++)  // the other fixed-effect sets, in the order they were added (src/samplers.jl:39-41)
        hipLaunchKernelGGL(k_fixed, dim3(1), dim3(1024), 0, h->stream, h->d_ycorr, (long long)h->N, h->fix[f].d_X, (int)h->fix[f].ncol, h->fix[f].d_xpx0,
                           h->fix[f].d_xpxR, h->fix[f].d_lhs0, h->fix[f].d_rhs0, h->d_bfix + h->fix[f].off, h->d_scal, (int)f, h->seed,
                           (uint64_t)h->chain, it, h->d_abort);
    }
    hipLaunchKernelGGL(k_prep, dim3((unsigned)(h->Ppad / 256 + 1)), dim3(256), 0, h->stream, (long long)h->Ppad, h->d_setof, h->d_loc,
                       h->d_vbidx, h->d_sets, h->d_scal, h->d_varBeta, h->d_mpm, h->d_lhs0, h->d_rhs0, h->d_beta, h->d_c, h->d_w,
                       h->d_q, h->d_T, h->d_chi, -1, h->seed, (uint64_t)h->chain, it, (long long)h->h_regs.size(), h->d_regs, h->d_regchi, h->d_rcls,
                       h->d_ccnt, (long long)(h->mode == 1 ? h->ccnt_words : 0), h->d_abort, h->d_tup, h->d_tupc, h->d_tupg);
    launch_tinv(h);
}

int iteration_post(ngp_handle *h, int64_t trace_idx) {  // variance / pi draws, traces and posterior sums; advances h->iter
    const uint64_t it = (uint64_t)(h->iter + 1);
    launch_variance(h, -1, it);
    h->iter += 1;
    const bool do_trace = h->d_trace_loci && trace_idx < h->trace_ext_cap, do_accum = is_kept(h, h->iter);
    if (do_trace || do_accum) {
        long long n = 16;
        if (do_trace) n = std::max<long long>(n, std::max<long long>(std::max<long long>(h->ntl, h->ntvb), (long long)h->sets.size()));
        if (do_accum) n = std::max<long long>(n, std::max<long long>(h->P, h->nvb));
        hipLaunchKernelGGL(k_post, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, (int)do_accum, (long long)h->P, (long long)h->nvb,
                           (int)h->sets.size(), h->d_beta, h->d_delta, h->d_varBeta, h->d_sum_beta, h->d_sum_beta2, h->d_sum_delta,
                           h->d_sum_varBeta, h->d_sets, h->d_scal, (int)do_trace, (long long)h->ntl, (const long long *)h->d_trace_loci,
                           (long long)h->ntvb, h->d_tr_beta, h->d_tr_vb, h->d_tr_pi, (long long)trace_idx, h->d_abort);
    }
    if (do_accum) {
        if (h->nfixcol > 0)
            hipLaunchKernelGGL(k_accum_fixed, dim3((unsigned)((h->nfixcol + 255) / 256)), dim3(256), 0, h->stream, (long long)h->nfixcol, h->d_bfix,
                               h->d_sum_bfix, h->d_abort);
        if (h->smp) return sample_enqueue(h);  // the kept sample goes to the file without stopping the chain (src/samplers.jl:56-104)
    }
    return NGP_OK;
}

// resume_mid: the head of this iteration (varE, intercept, fixed-effect sets) has run already -- its sweep ended at the census
// with nothing changed and is launched again, k_prep first (it redraws the same keyed numbers and clears the hand-off counters)
int one_iteration(ngp_handle *h, int64_t trace_idx, hipEvent_t *evs, bool resume_mid = false) {
    iteration_pre(h, trace_idx, resume_mid);
    launch_sweep(h, 0, h->NBLK, evs);
    return iteration_post(h, trace_idx);
}

// niter iterations from the handle's current state under `lease`, the launch queue bounded to 16 iterations; a launch that ends at
// its census (grid not co-resident beside another chain's, nothing changed) is run again once the device is this call's alone
int run_iterations(ngp_handle *h, int64_t niter, CuLease &lease, hipEvent_t *evs_first) {
    int rc;
    const int64_t iter0 = h->iter;
    bool resume_mid = false;
    int64_t n = 0;
    while (n < niter) {
        if ((rc = one_iteration(h, n, (n == 0) ? evs_first : nullptr, resume_mid))) return rc;
        resume_mid = false;
        ++n;
        if ((n & 15) == 0 || n == niter) {  // bound the launch queue
            HCHK(hipStreamSynchronize(h->stream));
            int64_t itf = 0;
            rc = check_abort(h, &itf);
            if (rc == NGP_RETRY_CENSUS) {
                // kernels behind the failing launch returned at once (abort word): the chain stands at iteration itf, head done
                h->exclusive = true; h->census_retries += 1;
                lease.make_exclusive();
                h->iter = itf - 1;
                n = h->iter - iter0;
                resume_mid = true;
                continue;
            }
            if (rc) return rc;
        }
    }
    return NGP_OK;
}

// class probabilities / their posterior sums of a BayesR set (either may be null); also clears the class counters
int set_class_state_dev(ngp_handle *h, int si, const double *pi, const double *sum_pi) {
    const int K = h->sets[(size_t)si].K;
    double *d = nullptr;
    int rc;
    if ((rc = dalloc(h, &d, (size_t)2 * NGP_RMAX))) return rc;
    if (pi) HCHK(hipMemcpyAsync(d, pi, (size_t)K * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (sum_pi) HCHK(hipMemcpyAsync(d + NGP_RMAX, sum_pi, (size_t)K * sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_set_class_state, dim3(1), dim3(1), 0, h->stream, h->d_sets, si, K, pi ? d : nullptr, sum_pi ? d + NGP_RMAX : nullptr);
    hipError_t e = hipStreamSynchronize(h->stream);
    dfree(d);
    if (e != hipSuccess) return fail(h, NGP_ERR_HIP, std::string("class state: ") + hipGetErrorString(e));
    return NGP_OK;
}

// doubles of the packed posterior (ngp_export_posterior_device): 3P + nvb + 2 nsets + sum K + fixed-effect columns + 3
int64_t posterior_words(const ngp_handle *h) {
    return 3 * h->P + h->nvb + 2 * (int64_t)h->sets.size() + h->nclass_total + h->nfixcol + 3;
}

// ---- sample stream (ngp_set_sample_file) ----
size_t sample_rec_bytes(const ngp_handle *h) {
    const size_t nd = 3 + (size_t)h->nfixcol + (size_t)h->P + (size_t)h->nvb + 2 * h->sets.size() + (size_t)h->nclass_total;
    return nd * 8 + (((size_t)h->P + 7) & ~(size_t)7);
}
void sample_writer_loop(SampleStream *S) {
    (void)hipSetDevice(S->device);
    for (;;) {
        int slot;
        {
            std::unique_lock<std::mutex> lk(S->mu);
            S->cv.wait(lk, [&] { return S->stop || !S->queue.empty(); });
            if (S->queue.empty()) return;  // stop, and nothing left
            slot = S->queue.front();
        }
        const bool ok = hipEventSynchronize(S->ev_copied[slot]) == hipSuccess;
        const long long it = *(const long long *)S->h_slot[slot];
        bool wrote = false, bad = !ok;
        if (ok && it >= 0) { wrote = true; bad = std::fwrite(S->h_slot[slot], 1, S->rec_bytes, S->f) != S->rec_bytes; }
        {
            std::lock_guard<std::mutex> lk(S->mu);
            S->queue.pop_front();
            S->busy[slot] = false;
            if (bad) S->io_error = true;
            if (wrote && !bad) S->nwritten++; else if (!wrote) S->ndropped++;
        }
        S->cv.notify_all();
    }
}
void sample_close(ngp_handle *h) {
    SampleStream *S = h->smp;
    if (!S) return;
    { std::lock_guard<std::mutex> lk(S->mu); S->stop = true; }
    S->cv.notify_all();
    if (S->writer.joinable()) S->writer.join();
    if (S->f) std::fclose(S->f);
    for (int i = 0; i < SampleStream::NSLOT; i++) {
        if (S->d_slot[i]) (void)hipFree(S->d_slot[i]);
        if (S->h_slot[i]) (void)hipHostFree(S->h_slot[i]);
        if (S->ev_packed[i]) (void)hipEventDestroy(S->ev_packed[i]);
        if (S->ev_copied[i]) (void)hipEventDestroy(S->ev_copied[i]);
    }
    if (S->copy_stream) (void)hipStreamDestroy(S->copy_stream);
    delete S;
    h->smp = nullptr;
}
// the kept sample of the iteration just enqueued on h->stream goes into the next ring slot, from there to the host on the copy stream
int sample_enqueue(ngp_handle *h) {
    SampleStream *S = h->smp;
    if (!S->header_written) {  // the model is final now: sizes and the file header
        S->rec_bytes = sample_rec_bytes(h);
        for (int i = 0; i < SampleStream::NSLOT; i++) {
            if (hipMalloc((void **)&S->d_slot[i], S->rec_bytes) != hipSuccess || hipHostMalloc((void **)&S->h_slot[i], S->rec_bytes, hipHostMallocDefault) != hipSuccess)
                return fail(h, NGP_ERR_NOMEM, "sample ring");
        }
        const int64_t hd[6] = {h->P, h->nvb, (int64_t)h->sets.size(), h->nfixcol, h->nclass_total, (int64_t)S->rec_bytes};
        bool ok = std::fwrite("NGPSMP01", 1, 8, S->f) == 8 && std::fwrite(hd, sizeof(hd), 1, S->f) == 1;
        for (auto &hs : h->sets) { const int64_t sg[6] = {hs.method, hs.K, hs.col0, hs.ncol, (int64_t)hs.vb0.size(), hs.tk}; ok = ok && std::fwrite(sg, sizeof(sg), 1, S->f) == 1; }
        if (!ok) return fail(h, NGP_ERR_ARG, "cannot write the sample file header: " + S->path);
        S->header_written = true;
    } else if (S->rec_bytes != sample_rec_bytes(h)) {
        return fail(h, NGP_ERR_STATE, "the model changed while a sample file is open (ngp_set_sample_file again)");
    }
    const int slot = (int)(S->nenq % SampleStream::NSLOT);
    {
        std::unique_lock<std::mutex> lk(S->mu);
        S->cv.wait(lk, [&] { return !S->busy[slot]; });  // only when the writer is NSLOT samples behind
        if (S->io_error) return fail(h, NGP_ERR_ARG, "writing the sample file failed: " + S->path);
        S->busy[slot] = true;
    }
    const long long n = std::max<long long>(std::max<long long>(h->P, h->nvb), std::max<long long>(h->nfixcol, 1));
    hipLaunchKernelGGL(k_sample_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, S->d_slot[slot], (long long)h->P, (long long)h->nvb,
                       (int)h->sets.size(), (long long)h->nfixcol, (long long)h->nclass_total, (long long)h->iter, h->d_beta, h->d_delta, h->d_varBeta,
                       h->d_sets, h->d_scal, h->d_bfix, h->d_abort);
    // (a HIP call that fails here gives the slot back: the next enqueue would otherwise wait for it forever instead of reporting)
    hipError_t e = hipEventRecord(S->ev_packed[slot], h->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(S->copy_stream, S->ev_packed[slot], 0);
    if (e == hipSuccess) e = hipMemcpyAsync(S->h_slot[slot], S->d_slot[slot], S->rec_bytes, hipMemcpyDeviceToHost, S->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(S->ev_copied[slot], S->copy_stream);
    if (e != hipSuccess) {
        { std::lock_guard<std::mutex> lk(S->mu); S->busy[slot] = false; }
        S->cv.notify_all();
        return fail(h, NGP_ERR_HIP, std::string("sample stream: ") + hipGetErrorString(e));
    }
    { std::lock_guard<std::mutex> lk(S->mu); S->queue.push_back(slot); }
    S->cv.notify_all();
    S->nenq++;
    return NGP_OK;
}
// end of a run: every enqueued sample is in the file when the call returns
int sample_flush(ngp_handle *h) {
    SampleStream *S = h->smp;
    if (!S) return NGP_OK;
    std::unique_lock<std::mutex> lk(S->mu);
    S->cv.wait(lk, [&] { return S->queue.empty(); });
    if (S->f) std::fflush(S->f);
    if (S->io_error) return fail(h, NGP_ERR_ARG, "writing the sample file failed: " + S->path);
    return NGP_OK;
}

int ready(ngp_handle *h) {
    REQUIRE(h->d_tiles != nullptr, NGP_ERR_STATE, "panel not set");
    REQUIRE(!h->panel_open, NGP_ERR_STATE, "the panel is still open: ngp_end_panel builds the Gram window the sweep needs");
    REQUIRE(h->have_y, NGP_ERR_STATE, "y not set");
    REQUIRE(!h->sets.empty(), NGP_ERR_STATE, "no marker set added");
    int rc;
    if ((rc = sync_tables(h))) return rc;
    return sync_linear_blocks(h, -1);
}

}  // namespace

extern "C" {

int32_t ngp_abi_version(void) { return NGP_ABI_VERSION; }

int32_t ngp_create(int32_t device, uint64_t seed, uint32_t chain_id, ngp_handle **out) {
    NGP_TRY
    if (!out) return fail(nullptr, NGP_ERR_ARG, "null out pointer");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, NGP_ERR_NODEVICE, std::string("no HIP device available (") + hipGetErrorString(e) +
                                                   "); libnextgp_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(nullptr, NGP_ERR_ARG, "device index out of range");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return fail(nullptr, NGP_ERR_HIP, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(nullptr, NGP_ERR_NODEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    ngp_handle *h = new (std::nothrow) ngp_handle();
    if (!h) return fail(nullptr, NGP_ERR_NOMEM, "out of host memory");
    h->device = device; h->seed = seed; h->chain = chain_id; h->cu_count = prop.multiProcessorCount;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreate(&h->stream)) != hipSuccess ||
        (e = hipEventCreate(&h->ev0)) != hipSuccess || (e = hipEventCreate(&h->ev1)) != hipSuccess) {
        std::string m = std::string("ngp_create: ") + hipGetErrorString(e);
        delete h;
        return fail(nullptr, NGP_ERR_HIP, m);
    }
    *out = h;
    return NGP_OK;
    NGP_CATCH(nullptr)
}

int32_t ngp_destroy(ngp_handle *h) {
    NGP_TRY
    if (!h) return NGP_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    sample_close(h);
    release_panel(h);
     dfree(h->d_cdlt); dfree(h->d_cdltg); dfree(h->d_rcls); dfree(h->d_seg_set); dfree(h->d_ccnt); dfree(h->d_abort); dfree(h->d_dbg); dfree(h->d_mpm); dfree(h->d_lhs0); dfree(h->d_rhs0); dfree(h->d_beta);
    dfree(h->d_c); dfree(h->d_w); dfree(h->d_q); dfree(h->d_T); dfree(h->d_chi); dfree(h->d_setof); dfree(h->d_loc);
    dfree(h->d_tinv); dfree(h->d_blin);
    dfree(h->d_vbidx); dfree(h->d_delta); dfree(h->d_sum_beta); dfree(h->d_sum_beta2); dfree(h->d_sum_delta);
    dfree(h->d_ycorr); dfree(h->d_part); dfree(h->d_dlt); dfree(h->d_sets); dfree(h->d_scal); dfree(h->d_varBeta);
    dfree(h->d_sum_varBeta); dfree(h->d_regs); dfree(h->d_seg_k0); dfree(h->d_seg_len); dfree(h->d_segpart); dfree(h->d_regchi);
    for (auto &fx : h->fix) { dfree(fx.d_X); dfree(fx.d_xpx0); dfree(fx.d_xpxR); dfree(fx.d_lhs0); dfree(fx.d_rhs0); }
    dfree(h->d_bfix); dfree(h->d_sum_bfix);
    dfree(h->d_tup); dfree(h->d_tupc); dfree(h->d_tupg); dfree(h->d_tsegpart); dfree(h->d_tregs); dfree(h->d_tseg_l0); dfree(h->d_tseg_len); dfree(h->d_tseg_set);
    dfree(h->d_tr_varE); dfree(h->d_tr_b); dfree(h->d_trace_loci); dfree(h->d_tr_beta); dfree(h->d_tr_vb); dfree(h->d_tr_pi);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return NGP_OK;
    NGP_CATCH(h)
}

const char *ngp_last_error(ngp_handle *h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int32_t ngp_set_panel_f64(ngp_handle *h, const double *M, int64_t N, int64_t P, int64_t ld, int32_t centre) {
    NGP_TRY
    return set_panel_host<double>(h, M, N, P, ld, centre);
    NGP_CATCH(h)
}
int32_t ngp_set_panel_f32(ngp_handle *h, const float *M, int64_t N, int64_t P, int64_t ld, int32_t centre) {
    NGP_TRY
    return set_panel_host<float>(h, M, N, P, ld, centre);
    NGP_CATCH(h)
}
int32_t ngp_begin_panel(ngp_handle *h, int64_t N, int64_t P) {
    NGP_TRY
    return begin_panel(h, N, P);
    NGP_CATCH(h)
}
int32_t ngp_panel_columns_f64(ngp_handle *h, int64_t col0, const double *M, int64_t ncol, int64_t ld, int32_t centre) {
    NGP_TRY
    return panel_columns<double>(h, col0, M, ncol, ld, centre);
    NGP_CATCH(h)
}
int32_t ngp_panel_columns_f32(ngp_handle *h, int64_t col0, const float *M, int64_t ncol, int64_t ld, int32_t centre) {
    NGP_TRY
    return panel_columns<float>(h, col0, M, ncol, ld, centre);
    NGP_CATCH(h)
}
int32_t ngp_panel_columns_u8(ngp_handle *h, int64_t col0, const uint8_t *G, int64_t ncol, int64_t ld, int32_t centre) {
    NGP_TRY
    return panel_columns<uint8_t>(h, col0, G, ncol, ld, centre);
    NGP_CATCH(h)
}
int32_t ngp_end_panel(ngp_handle *h) {
    NGP_TRY
    return end_panel(h);
    NGP_CATCH(h)
}

}  // extern "C" (helpers below are C++)

// one staged chunk of genotype bytes (whole 64-column blocks, column-major with leading dimension ld, already on the device)
// into the tiles: fp32 centred tiles, or the bytes as they are plus the column means (compact storage)
static void ingest_u8_chunk(ngp_handle *h, const uint8_t *d_g, int64_t N, int64_t ld, int64_t t0, int64_t nb, int64_t ncols, int centre,
                            double *d_mu) {
    const int64_t c0 = t0 * NGP_BLK;
    if (h->storage == 1) {  // the bytes stay bytes; the means go to the handle
        hipLaunchKernelGGL(k_u8_colmean, dim3((unsigned)ncols), dim3(256), 0, h->stream, d_g, (long long)N, (long long)ld, (int)centre,
                           h->d_mean + c0);
        hipLaunchKernelGGL(k_u8_fill8, dim3((unsigned)h->S, (unsigned)nb), dim3(256), 0, h->stream, (uint8_t *)h->d_tiles, d_g, (long long)N,
                           (long long)ld, (long long)ncols, (int)h->R, (int)h->S, (long long)t0);
    } else {
        hipLaunchKernelGGL(k_u8_colmean, dim3((unsigned)ncols), dim3(256), 0, h->stream, d_g, (long long)N, (long long)ld, (int)centre, d_mu);
        (void)hipMemcpyAsync(h->d_mean + c0, d_mu, (size_t)ncols * sizeof(double), hipMemcpyDeviceToDevice, h->stream);
        hipLaunchKernelGGL(k_u8_fill, dim3((unsigned)h->S, (unsigned)nb), dim3(256), 0, h->stream, h->d_tiles, d_g, (long long)N,
                           (long long)ld, (long long)ncols, (int)h->R, (int)h->S, (long long)t0, d_mu);
    }
}

extern "C" {

int32_t ngp_set_panel_u8(ngp_handle *h, const uint8_t *G, int64_t N, int64_t P, int64_t ld, int32_t centre) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(G != nullptr, NGP_ERR_ARG, "null panel pointer");
    REQUIRE(ld >= N, NGP_ERR_ARG, "leading dimension smaller than N");
    if ((rc = alloc_panel(h, N, P))) return rc;
    // staged through the device in chunks of whole 64-column blocks (about 64 MiB of genotypes at a time)
    const int64_t blk_bytes = (int64_t)NGP_BLK * ld;
    const int64_t nb_chunk = std::max<int64_t>(1, std::min<int64_t>(h->NBLK, ((int64_t)64 << 20) / blk_bytes));
    uint8_t *d_g = nullptr;
    double *d_mu = nullptr;
    if (hipMalloc((void **)&d_g, (size_t)nb_chunk * blk_bytes) != hipSuccess) return fail(h, NGP_ERR_NOMEM, "staging buffer");
    if ((rc = dalloc(h, &d_mu, (size_t)nb_chunk * NGP_BLK))) { (void)hipFree(d_g); return rc; }
    hipError_t e = hipSuccess;
    for (int64_t t0 = 0; t0 < h->NBLK && e == hipSuccess; t0 += nb_chunk) {
        const int64_t nb = std::min<int64_t>(nb_chunk, h->NBLK - t0);
        const int64_t c0 = t0 * NGP_BLK, ncols = std::min<int64_t>(nb * NGP_BLK, P - c0);
        // the last column may be shorter than ld in the caller's buffer: copy ncols-1 full columns + N bytes
        const size_t bytes = (size_t)(ncols - 1) * ld + (size_t)N;
        e = hipMemcpyAsync(d_g, G + (size_t)c0 * ld, bytes, hipMemcpyHostToDevice, h->stream);
        if (e != hipSuccess) break;
        ingest_u8_chunk(h, d_g, N, ld, t0, nb, ncols, centre, d_mu);
        e = hipStreamSynchronize(h->stream);  // the staging buffer is reused by the next chunk
    }
    (void)hipFree(d_g);
    dfree(d_mu);
    if (e != hipSuccess) return fail(h, NGP_ERR_HIP, std::string("set_panel_u8: ") + hipGetErrorString(e));
    return build_gram(h);
    NGP_CATCH(h)
}

// ---- binary panel file (replaces the text genotype file of src/prepMatVec.jl:116-131 for large panels) ----
// header (32 bytes): magic "NGPPNL01", int64 N, int64 P, int32 bits (8 or 2), int32 0.  Then P columns: N bytes (bits 8), or
// ceil(N / 4) bytes with four genotypes per byte, individual i in bits 2 (i mod 4) .. 2 (i mod 4) + 1 (bits 2; code 3 is refused).
namespace {
struct PanelHeader { char magic[8]; int64_t N, P; int32_t bits, zero; };
}

int32_t ngp_write_panel_file(const char *path, const uint8_t *G, int64_t N, int64_t P, int64_t ld, int32_t bits) {
    NGP_TRY
    if (!path || !G || N <= 0 || P <= 0 || ld < N || (bits != 8 && bits != 2)) return NGP_ERR_ARG;
    std::vector<uint8_t> packed(bits == 2 ? (size_t)(N + 3) / 4 : 0);  // before the file is opened: an allocation failure leaves nothing behind
    FILE *f = std::fopen(path, "wb");
    if (!f) return NGP_ERR_ARG;
    PanelHeader hd;
    std::memcpy(hd.magic, "NGPPNL01", 8);
    hd.N = N; hd.P = P; hd.bits = bits; hd.zero = 0;
    bool ok = std::fwrite(&hd, sizeof hd, 1, f) == 1;
    for (int64_t j = 0; j < P && ok; j++) {
        const uint8_t *col = G + (size_t)j * ld;
        if (bits == 8) {
            ok = std::fwrite(col, 1, (size_t)N, f) == (size_t)N;
        } else {
            std::fill(packed.begin(), packed.end(), 0);
            for (int64_t i = 0; i < N; i++) {
                if (col[i] > 2) { ok = false; break; }  // two bits hold the allele counts 0, 1, 2
                packed[(size_t)i >> 2] |= (uint8_t)(col[i] << (2 * (i & 3)));
            }
            if (ok) ok = std::fwrite(packed.data(), 1, packed.size(), f) == packed.size();
        }
    }
    ok = (std::fclose(f) == 0) && ok;
    return ok ? NGP_OK : NGP_ERR_ARG;
    NGP_CATCH(nullptr)
}

int32_t ngp_read_panel_header(const char *path, int64_t *N, int64_t *P, int32_t *bits) {
    NGP_TRY
    if (!path) return NGP_ERR_ARG;
    FILE *f = std::fopen(path, "rb");
    if (!f) return NGP_ERR_ARG;
    PanelHeader hd;
    const bool ok = std::fread(&hd, sizeof hd, 1, f) == 1 && std::memcmp(hd.magic, "NGPPNL01", 8) == 0 && hd.N > 0 && hd.P > 0 &&
                    (hd.bits == 8 || hd.bits == 2);
    std::fclose(f);
    if (!ok) return NGP_ERR_ARG;
    if (N) *N = hd.N;
    if (P) *P = hd.P;
    if (bits) *bits = hd.bits;
    return NGP_OK;
    NGP_CATCH(nullptr)
}

int32_t ngp_load_panel_file(ngp_handle *h, const char *path, int32_t centre) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(path != nullptr, NGP_ERR_ARG, "null path");
    FILE *f = std::fopen(path, "rb");
    REQUIRE(f != nullptr, NGP_ERR_ARG, std::string("cannot open panel file ") + path);
    PanelHeader hd;
    if (std::fread(&hd, sizeof hd, 1, f) != 1 || std::memcmp(hd.magic, "NGPPNL01", 8) != 0 || hd.N <= 0 || hd.P <= 0 || (hd.bits != 8 && hd.bits != 2)) {
        std::fclose(f);
        return fail(h, NGP_ERR_ARG, std::string("not a panel file (magic NGPPNL01, bits 8 or 2): ") + path);
    }
    const int64_t N = hd.N, P = hd.P;
    if ((rc = alloc_panel(h, N, P))) { std::fclose(f); return rc; }
    // the file streams through a pinned host buffer in chunks of whole 64-column blocks; two-bit columns are unpacked on the host
    const int64_t colbytes = (hd.bits == 8) ? N : (N + 3) / 4;
    const int64_t blk_bytes = (int64_t)NGP_BLK * N;
    const int64_t nb_chunk = std::max<int64_t>(1, std::min<int64_t>(h->NBLK, ((int64_t)64 << 20) / blk_bytes));
    uint8_t *d_g = nullptr, *h_g = nullptr;
    double *d_mu = nullptr;
    std::vector<uint8_t> packed((hd.bits == 2) ? (size_t)colbytes : 0);
    hipError_t e = hipMalloc((void **)&d_g, (size_t)nb_chunk * blk_bytes);
    if (e == hipSuccess) e = hipHostMalloc((void **)&h_g, (size_t)nb_chunk * blk_bytes, hipHostMallocDefault);
    if (e != hipSuccess) { std::fclose(f); if (d_g) (void)hipFree(d_g); return fail(h, NGP_ERR_NOMEM, "staging buffers"); }
    if ((rc = dalloc(h, &d_mu, (size_t)nb_chunk * NGP_BLK))) { std::fclose(f); (void)hipFree(d_g); (void)hipHostFree(h_g); return rc; }
    std::string why;
    for (int64_t t0 = 0; t0 < h->NBLK && e == hipSuccess && why.empty(); t0 += nb_chunk) {
        const int64_t nb = std::min<int64_t>(nb_chunk, h->NBLK - t0);
        const int64_t c0 = t0 * NGP_BLK, ncols = std::min<int64_t>(nb * NGP_BLK, P - c0);
        for (int64_t jc = 0; jc < ncols && why.empty(); jc++) {
            uint8_t *dst = h_g + (size_t)jc * N;
            if (hd.bits == 8) {
                if (std::fread(dst, 1, (size_t)N, f) != (size_t)N) why = "panel file truncated";
            } else {
                if (std::fread(packed.data(), 1, packed.size(), f) != packed.size()) { why = "panel file truncated"; break; }
                for (int64_t i = 0; i < N; i++) {
                    const uint8_t g = (uint8_t)((packed[(size_t)i >> 2] >> (2 * (i & 3))) & 3u);
                    if (g == 3) { why = "panel file holds a missing genotype (code 3): impute before loading"; break; }
                    dst[i] = g;
                }
            }
        }
        if (!why.empty()) break;
        e = hipMemcpyAsync(d_g, h_g, (size_t)ncols * N, hipMemcpyHostToDevice, h->stream);
        if (e != hipSuccess) break;
        ingest_u8_chunk(h, d_g, N, N, t0, nb, ncols, centre, d_mu);
        e = hipStreamSynchronize(h->stream);  // both staging buffers are reused by the next chunk
    }
    std::fclose(f);
    (void)hipFree(d_g);
    (void)hipHostFree(h_g);
    dfree(d_mu);
    if (!why.empty()) { release_panel(h); return fail(h, NGP_ERR_ARG, why); }
    if (e != hipSuccess) { release_panel(h); return fail(h, NGP_ERR_HIP, std::string("load_panel_file: ") + hipGetErrorString(e)); }
    return build_gram(h);
    NGP_CATCH(h)
}

int32_t ngp_generate_panel(ngp_handle *h, int64_t N, int64_t P, double maf_lo, double maf_hi, uint64_t panel_seed) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(maf_lo > 0.0 && maf_hi < 1.0 && maf_lo <= maf_hi, NGP_ERR_ARG, "maf range must satisfy 0 < lo <= hi < 1");
    const auto ts0 = std::chrono::steady_clock::now();
    if ((rc = alloc_panel(h, N, P))) return rc;
    (void)hipStreamSynchronize(h->stream);  // (the allocations' zeroing: timed with them)
    const auto ts1 = std::chrono::steady_clock::now();
    double *d_mu = nullptr;
    uint32_t *d_thr = nullptr;
    if ((rc = dalloc(h, &d_mu, (size_t)P))) return rc;
    if ((rc = dalloc(h, &d_thr, (size_t)P))) { dfree(d_mu); return rc; }
    hipLaunchKernelGGL(k_gen_colmean, dim3((unsigned)P), dim3(256), 0, h->stream, (long long)N, (long long)P, maf_lo, maf_hi, panel_seed,
                       d_mu, d_thr);
    (void)hipMemcpyAsync(h->d_mean, d_mu, (size_t)P * sizeof(double), hipMemcpyDeviceToDevice, h->stream);
    if (h->storage == 1) {
        hipLaunchKernelGGL(k_gen_fill8, dim3((unsigned)h->S, (unsigned)h->NBLK), dim3(256), 0, h->stream, (uint8_t *)h->d_tiles, (long long)N,
                           (long long)P, (int)h->R, (int)h->S, panel_seed, d_thr);
    } else
    hipLaunchKernelGGL(k_gen_fill, dim3((unsigned)h->S, (unsigned)h->NBLK), dim3(256), 0, h->stream, h->d_tiles, (long long)N,
                       (long long)P, (int)h->R, (int)h->S, panel_seed, d_mu, d_thr);
    hipError_t e = hipStreamSynchronize(h->stream);
    dfree(d_mu); dfree(d_thr);
    if (e != hipSuccess) return fail(h, NGP_ERR_HIP, std::string("generate_panel: ") + hipGetErrorString(e));
    const auto ts2 = std::chrono::steady_clock::now();
    rc = build_gram(h);
    const auto ts3 = std::chrono::steady_clock::now();
    h->setup_ms[0] = std::chrono::duration<double, std::milli>(ts1 - ts0).count();
    h->setup_ms[1] = std::chrono::duration<double, std::milli>(ts2 - ts1).count();
    h->setup_ms[2] = std::chrono::duration<double, std::milli>(ts3 - ts2).count();
    return rc;
    NGP_CATCH(h)
}

int32_t ngp_get_setup_timing(ngp_handle *h, double *alloc_ms, double *tiles_ms, double *gram_ms) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    if (alloc_ms) *alloc_ms = h->setup_ms[0];
    if (tiles_ms) *tiles_ms = h->setup_ms[1];
    if (gram_ms) *gram_ms = h->setup_ms[2];
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_get_layout(ngp_handle *h, int64_t *R, int64_t *S, int64_t *nblk) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_tiles != nullptr, NGP_ERR_STATE, "panel not set");
    if (R) *R = h->R;
    if (S) *S = h->S;
    if (nblk) *nblk = h->NBLK;
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_get_mpm(ngp_handle *h, double *out, int64_t P) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_tiles != nullptr, NGP_ERR_STATE, "panel not set");
    REQUIRE(!h->panel_open, NGP_ERR_STATE, "the panel is still open: x'x and the Gram window exist after ngp_end_panel");
    REQUIRE(out && P == h->P, NGP_ERR_ARG, "mpm buffer must hold P entries");
    HCHK(hipMemcpy(out, h->d_mpm, (size_t)P * sizeof(double), hipMemcpyDeviceToHost));
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_get_gram(ngp_handle *h, int64_t t, double *out) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_tiles != nullptr, NGP_ERR_STATE, "panel not set");
    REQUIRE(!h->panel_open, NGP_ERR_STATE, "the panel is still open: x'x and the Gram window exist after ngp_end_panel");
    REQUIRE(out && t >= 0 && t < h->NBLK, NGP_ERR_ARG, "block index out of range");
    HCHK(hipMemcpy(out, h->d_gramx + (size_t)t * h->D * NGP_BLK * NGP_BLK, NGP_BLK * NGP_BLK * sizeof(double), hipMemcpyDeviceToHost));
    // the device keeps entry [k][j] for j > k only (plus x'x in mpm); hand back the symmetric block
    double diag[NGP_BLK];
    HCHK(hipMemcpy(diag, h->d_mpm + (size_t)t * NGP_BLK, sizeof(diag), hipMemcpyDeviceToHost));
    for (int k = 0; k < NGP_BLK; k++) {
        out[k * NGP_BLK + k] = diag[k];
        for (int j = k + 1; j < NGP_BLK; j++) out[j * NGP_BLK + k] = out[k * NGP_BLK + j];
    }
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_xbeta(ngp_handle *h, const double *beta, int64_t P, double *out, int64_t N) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_tiles != nullptr, NGP_ERR_STATE, "panel not set");
    REQUIRE(!h->panel_open, NGP_ERR_STATE, "the panel is still open: x'x and the Gram window exist after ngp_end_panel");
    REQUIRE(beta && out && P == h->P && N == h->N, NGP_ERR_ARG, "xbeta: size mismatch");
    double *d_b = nullptr, *d_o = nullptr;
    if ((rc = dalloc(h, &d_b, (size_t)h->Ppad))) return rc;
    if ((rc = dalloc(h, &d_o, (size_t)h->L))) { dfree(d_b); return rc; }
    hipError_t e = hipMemcpyAsync(d_b, beta, (size_t)P * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess && (size_t)h->Ppad > (size_t)P) e = hipMemsetAsync(d_b + P, 0, ((size_t)h->Ppad - (size_t)P) * sizeof(double), h->stream);
    if (h->storage == 1)
        hipLaunchKernelGGL(k_xbeta8, dim3((unsigned)h->S), dim3(256), 0, h->stream, (const uint8_t *)h->d_tiles, h->d_mean, d_b, d_o, (int)h->R,
                           (int)h->S, (long long)h->NBLK, (long long)h->N);
    else
    hipLaunchKernelGGL(k_xbeta, dim3((unsigned)h->S), dim3(256), 0, h->stream, h->d_tiles, d_b, d_o, (int)h->R, (int)h->S,
                       (long long)h->NBLK);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_o, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    dfree(d_b); dfree(d_o);
    if (e != hipSuccess) return fail(h, NGP_ERR_HIP, std::string("xbeta: ") + hipGetErrorString(e));
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_add_marker_set(ngp_handle *h, int64_t col0, int64_t ncol, int32_t method, double df, double scale,
                           const int64_t *reg_start, const int64_t *reg_stop, int64_t nreg, const double *varBeta0, double pi0,
                           int32_t estPi, const double *lhs0, const double *rhs0, int32_t *set_id) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_tiles != nullptr, NGP_ERR_STATE, "panel not set");
    REQUIRE(!h->panel_open, NGP_ERR_STATE, "the panel is still open: x'x and the Gram window exist after ngp_end_panel");
    REQUIRE(h->sets.size() < 16, NGP_ERR_ARG, "at most 16 marker sets");
    REQUIRE(col0 >= 0 && ncol > 0 && col0 + ncol <= h->P, NGP_ERR_ARG, "marker set outside the panel");
    REQUIRE(method == NGP_METHOD_BAYESPR || method == NGP_METHOD_BAYESB || method == NGP_METHOD_BAYESC || method == NGP_METHOD_BAYESR, NGP_ERR_ARG,
            "unknown method");
    if (method == NGP_METHOD_BAYESR) REQUIRE(h->adding_r, NGP_ERR_ARG, "BayesR sets are added with ngp_add_marker_set_r (classes and their probabilities)");
    REQUIRE(reg_start && reg_stop && varBeta0 && nreg > 0, NGP_ERR_ARG, "regions / varBeta0 missing");
    REQUIRE(std::isfinite(df) && std::isfinite(scale) && df > 0, NGP_ERR_ARG, "df/scale must be finite, df > 0");
    for (int64_t k = col0; k < col0 + ncol; k++) REQUIRE(h->h_setof[k] == -1, NGP_ERR_ARG, "marker sets overlap");
    if (method == NGP_METHOD_BAYESB) {
        REQUIRE(nreg == ncol, NGP_ERR_ARG, "BayesB needs one region per locus (src/mme.jl:356)");
        REQUIRE(pi0 > 0.0 && pi0 < 1.0, NGP_ERR_ARG, "BayesB pi must be in (0,1)");
    }
    if (method == NGP_METHOD_BAYESC) {
        REQUIRE(nreg == 1, NGP_ERR_ARG, "BayesC has one variance for the whole set (src/functions.jl:205)");
        REQUIRE(pi0 > 0.0 && pi0 < 1.0, NGP_ERR_ARG, "BayesC pi must be in (0,1)");
        REQUIRE(varBeta0[0] > 0.0, NGP_ERR_ARG, "BayesC varBeta0 must be positive");
    }
    // regions must tile [0,ncol) in order (regionArray of UnitRanges, src/mme.jl:335-347)
    int64_t expect = 0;
    for (int64_t r = 0; r < nreg; r++) {
        REQUIRE(reg_start[r] == expect && reg_stop[r] > reg_start[r], NGP_ERR_ARG, "regions must be consecutive and non-empty");
        REQUIRE(std::isfinite(varBeta0[r]) && varBeta0[r] >= 0.0, NGP_ERR_ARG, "varBeta0 must be finite and >= 0");
        expect = reg_stop[r];
    }
    REQUIRE(expect == ncol, NGP_ERR_ARG, "regions must cover the whole set");
    const int si = (int)h->sets.size();
    HSet hs{col0, ncol, method, df, scale, nreg, h->nvb, estPi, 0, pi0, std::vector<double>(varBeta0, varBeta0 + nreg)};
    // grow varBeta storage
    const int64_t new_nvb = h->nvb + nreg;
    if (new_nvb > h->vb_cap) {
        int64_t cap = std::max<int64_t>(new_nvb, 2 * h->vb_cap);
        double *nv = nullptr, *ns = nullptr;
        if ((rc = dalloc(h, &nv, (size_t)cap))) return rc;
        if ((rc = dalloc(h, &ns, (size_t)cap))) return rc;
        if (h->nvb > 0) {
            HCHK(hipMemcpyAsync(nv, h->d_varBeta, (size_t)h->nvb * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            HCHK(hipMemcpyAsync(ns, h->d_sum_varBeta, (size_t)h->nvb * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        }
        HCHK(hipStreamSynchronize(h->stream));
        dfree(h->d_varBeta); dfree(h->d_sum_varBeta);
        h->d_varBeta = nv; h->d_sum_varBeta = ns; h->vb_cap = cap;
    }
    HCHK(hipMemcpy(h->d_varBeta + h->nvb, varBeta0, (size_t)nreg * sizeof(double), hipMemcpyHostToDevice));
    for (int64_t r = 0; r < nreg; r++)
        for (int64_t l = reg_start[r]; l < reg_stop[r]; l++) {
            int64_t k = col0 + l;
            h->h_setof[k] = (int8_t)si;
            h->h_loc[k] = (int32_t)l;
            h->h_vbidx[k] = (int32_t)(h->nvb + (method == NGP_METHOD_BAYESB ? l : r));
        }
    if (method == NGP_METHOD_BAYESPR || method == NGP_METHOD_BAYESC || method == NGP_METHOD_BAYESR) {  // sets whose variance comes from a sum of squares
        for (int64_t r = 0; r < nreg; r++) {
            DReg dr;
            dr.seg0 = (long long)h->h_seg_k0.size();
            dr.set = si; dr.rg = (int)r; dr.vb = (int)(h->nvb + r); dr.n = reg_stop[r] - reg_start[r];
            int ns = 0;
            for (int64_t l0 = reg_start[r]; l0 < reg_stop[r]; l0 += NGP_SEG) {
                h->h_seg_k0.push_back(col0 + l0);
                h->h_seg_len.push_back((int32_t)std::min<int64_t>(NGP_SEG, reg_stop[r] - l0));
                h->h_seg_set.push_back((int32_t)si);
                ns++;
            }
            dr.nseg = ns;
            h->h_regs.push_back(dr);
        }
    }
    std::vector<double> z((size_t)ncol, 0.0);
    HCHK(hipMemcpy(h->d_lhs0 + col0, lhs0 ? lhs0 : z.data(), (size_t)ncol * sizeof(double), hipMemcpyHostToDevice));
    HCHK(hipMemcpy(h->d_rhs0 + col0, rhs0 ? rhs0 : z.data(), (size_t)ncol * sizeof(double), hipMemcpyHostToDevice));
    DSet ds;
    memset(&ds, 0, sizeof(ds));
    ds.method = method; ds.estPi = estPi; ds.df = df; ds.scale = scale; ds.sdf = scale * df; ds.col0 = col0; ds.ncol = ncol;
    HCHK(hipMemcpy(h->d_sets + si, &ds, sizeof(DSet), hipMemcpyHostToDevice));
    const double p1 = pi0, p0 = 1.0 - pi0;
    hipLaunchKernelGGL(k_set_pi, dim3(1), dim3(1), 0, h->stream, h->d_sets, si, p0, p1);  // src/mme.jl:351,360
    HCHK(hipStreamSynchronize(h->stream));
    h->nvb = new_nvb;
    h->sets.push_back(hs);
    h->tables_dirty = true;
    h->trace_ext_cap = 0;  // d_tr_pi holds one column per set: sized again by the next traced ngp_run
    if (set_id) *set_id = si;
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_set_y(ngp_handle *h, const double *y, int64_t N) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_tiles != nullptr, NGP_ERR_STATE, "panel not set");
    REQUIRE(y && N == h->N, NGP_ERR_ARG, "y must have N entries");
    for (int64_t i = 0; i < N; i++) REQUIRE(std::isfinite(y[i]), NGP_ERR_ARG, "non-finite phenotype");
    HCHK(hipMemsetAsync(h->d_ycorr, 0, (size_t)h->L * sizeof(double), h->stream));
    HCHK(hipMemcpyAsync(h->d_ycorr, y, (size_t)N * sizeof(double), hipMemcpyHostToDevice, h->stream));  // src/mme.jl:57
    HCHK(hipMemsetAsync(h->d_beta, 0, (size_t)h->Ppad * sizeof(double), h->stream));                  // src/mme.jl:443
    HCHK(hipMemsetAsync(h->d_delta, 1, (size_t)h->Ppad, h->stream));                                  // src/mme.jl:444
    HCHK(hipMemsetAsync(h->d_scal, 0, sizeof(DScal), h->stream));
    HCHK(hipMemsetAsync(h->d_sum_beta, 0, (size_t)h->Ppad * sizeof(double), h->stream));
    HCHK(hipMemsetAsync(h->d_sum_beta2, 0, (size_t)h->Ppad * sizeof(double), h->stream));
    HCHK(hipMemsetAsync(h->d_sum_delta, 0, (size_t)h->Ppad * sizeof(double), h->stream));
    // a second chain on the same handle starts from the priors, with empty posterior sums (src/mme.jl:351-360, 516)
    if (h->nvb > 0) HCHK(hipMemsetAsync(h->d_sum_varBeta, 0, (size_t)h->nvb * sizeof(double), h->stream));
    for (size_t si = 0; si < h->sets.size(); si++) {
        const HSet &hs = h->sets[si];
        HCHK(hipMemcpyAsync(h->d_varBeta + hs.vb_off, hs.vb0.data(), hs.vb0.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_set_pi, dim3(1), dim3(1), 0, h->stream, h->d_sets, (int)si, 1.0 - hs.pi0, hs.pi0);
        hipLaunchKernelGGL(k_set_sum_pi, dim3(1), dim3(1), 0, h->stream, h->d_sets, (int)si, 0.0, 0.0);
        if (hs.K > 0) {
            int rc2 = set_class_state_dev(h, (int)si, hs.rpi.data(), std::vector<double>((size_t)hs.K, 0.0).data());
            if (rc2) return rc2;
        }
    }
    if (h->nfixcol > 0) {
        HCHK(hipMemsetAsync(h->d_bfix, 0, (size_t)h->nfixcol * sizeof(double), h->stream));
        HCHK(hipMemsetAsync(h->d_sum_bfix, 0, (size_t)h->nfixcol * sizeof(double), h->stream));
    }
    HCHK(hipStreamSynchronize(h->stream));
    h->iter = 0; h->have_y = true; h->poisoned = false; h->ntrace = 0;
    for (auto &hs : h->sets) hs.fine_calls = 0;
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_set_residual_prior(ngp_handle *h, double df, double scale) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(std::isfinite(df) && std::isfinite(scale) && df > 0 && scale >= 0, NGP_ERR_ARG, "bad residual prior");
    h->e_df = df; h->e_scale = scale;
    return NGP_OK;
    NGP_CATCH(h)
}
int32_t ngp_set_intercept(ngp_handle *h, int32_t on) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    h->intercept = on ? 1 : 0;
    return NGP_OK;
    NGP_CATCH(h)
}
int32_t ngp_set_schedule(ngp_handle *h, int64_t chainLength, int64_t burnIn, int64_t thin) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(chainLength >= 0 && burnIn >= 0 && thin >= 1, NGP_ERR_ARG, "bad schedule");
    h->chainLength = chainLength; h->burnIn = burnIn; h->thin = thin;
    return NGP_OK;
    NGP_CATCH(h)
}

}  // extern "C"

namespace {
// checks and trace buffers in front of a run of niter iterations
int prepare_run(ngp_handle *h, int64_t niter) {
    int rc;
    if ((rc = enter(h))) return rc;
    if ((rc = ready(h))) return rc;
    REQUIRE(niter >= 0, NGP_ERR_ARG, "niter must be >= 0");
    REQUIRE(!h->poisoned, NGP_ERR_STATE, "an earlier sweep was abandoned half-way: set the state again (ngp_set_y / ngp_set_state)");
    REQUIRE(h->ntvb <= h->nvb, NGP_ERR_STATE, "more variance traces requested than the model has variance components");
    if (h->ntl + h->ntvb + (int64_t)h->sets.size() > 0 && h->d_trace_loci && niter > h->trace_ext_cap) {
        if ((rc = dalloc(h, &h->d_tr_beta, (size_t)niter * std::max<int64_t>(h->ntl, 1)))) return rc;
        if ((rc = dalloc(h, &h->d_tr_vb, (size_t)niter * std::max<int64_t>(h->ntvb, 1)))) return rc;
        if ((rc = dalloc(h, &h->d_tr_pi, (size_t)niter * std::max<size_t>(h->sets.size(), 1)))) return rc;
        h->trace_ext_cap = niter;
    }
    if (niter > h->trace_cap) {
        if ((rc = dalloc(h, &h->d_tr_varE, (size_t)niter))) return rc;
        if ((rc = dalloc(h, &h->d_tr_b, (size_t)niter))) return rc;
        h->trace_cap = niter;
    }
    h->ntrace = niter;
    return NGP_OK;
}

// K chains per pass over the panel (k_sweep_multi, ngp_sweep.h): can these handles' chains share ONE sweep launch?  They must
// share one panel (ngp_share_panel) and run the engine the fused kernel is built for: persistent sweep, fp32 tiles, phase streamer
// on shards of at most 64 rows, lag 6 or 8, no diagnostics -- and the fused grid must fit the device.
// reducer workgroups of a fused launch serve two chains each from NGP_PAIR_FROM chains on (phase streamer only; knob bits 11 / 12
// force it on / off for timing)
int fused_pair(const ngp_handle *h0, int n) {
    if (h0->streamer != 1 || n < 2) return 0;
    if (h0->knob & 2048) return 1;
    if (h0->knob & 4096) return 0;
    return n >= NGP_PAIR_FROM ? 1 : 0;
}

bool fusable(ngp_handle **hs, int n) {
    if (n < 2 || n > NGP_MAXC) return false;
    ngp_handle *h0 = hs[0];
    if (!h0->pm || h0->mode != 1 || h0->V != 1) return false;
    const bool phase = h0->storage == 0 && h0->streamer == 1 && h0->R <= 64 && (h0->D == 6 || h0->D == 8);   // role_streamer_multi
    const bool rows = h0->storage == 0 && h0->streamer == 2 && (h0->D >= 4 && h0->D <= 6) && n == 2 &&        // role_streamer_rows_multi
                      ngp_rows_multi_lds_bytes((int)h0->R, n) <= (size_t)160 * 1024;
    bool bytes = false;                                                                                     // ... over byte tiles
    if (h0->storage == 1 && h0->streamer == 3 && n <= 3 && ngp_rows_multi_lds_bytes((int)h0->R, n, true) <= (size_t)160 * 1024) {
        const int nt = ngp_u8_tasks((int)h0->R);  // update tasks per lane: what the delay line leaves for more chains' arithmetic
        bytes = (nt == 1 && (h0->D == 4 || h0->D == 6 || h0->D == 8)) || (nt == 2 && (h0->D == 4 || h0->D == 8)) || (nt == 4 && n == 2 && h0->D == 4);
    }
    if (!phase && !rows && !bytes) return false;
    for (int i = 0; i < n; i++) {
        ngp_handle *h = hs[i];
        if (h->pm != h0->pm || h->device != h0->device || h->dbg_mode != 0 || h->d_dbg || h->dbg_census_fail_iter > 0) return false;
    }
    // (the samplers sit at blocks 0, 8, .., 8 (n - 1) of the grid -- one XCD under round-robin placement: the grid must reach the last)
    const int64_t grid = (int64_t)n + ngp_multi_reducers(n, h0->NG, fused_pair(h0, n)) + h0->S;
    return grid <= h0->cu_count && grid > (int64_t)8 * (n - 1);
}

// niter iterations of n chains, every iteration ONE fused sweep launch on the first handle's stream; each chain's small kernels
// (head, coefficients, variance draws, posterior sums) run there too, chain after chain.  Bit for bit what each chain draws alone.
int run_fused(ngp_handle **hs, int n, int64_t niter) {
    ngp_handle *h = hs[0];  // errors are reported on the leader (and copied to the others by the caller)
    int rc;
    for (int i = 0; i < n; i++)
        if ((rc = prepare_run(hs[i], niter))) { if (i) h->err = hs[i]->err; return rc; }
    const int pair = fused_pair(h, n);
    const int64_t grid = (int64_t)n + ngp_multi_reducers(n, h->NG, pair) + h->S;
    const size_t lds_sampler = (size_t)(3 * 4096 + 2 * NGP_RING * NGP_BLK + 6 * NGP_BLK) * sizeof(double) + 2 * NGP_BLK * sizeof(int) + 320 + NGP_SAMPLER_TUPLE_LDS;
    const size_t lds = std::max(h->streamer == 3 ? ngp_rows_multi_lds_bytes((int)h->R, n, true)
                                                 : (h->streamer == 2 ? ngp_rows_multi_lds_bytes((int)h->R, n) : ngp_multi_lds_bytes((int)h->R, n)), lds_sampler);
    REQUIRE(lds <= 160 * 1024, NGP_ERR_STATE, "fused sweep: LDS of a streamer with this many chains exceeds 160 KiB");
    bool tup = false, rset = false;  // a chain with a Tuple / BayesR set: the fused kernel whose samplers hold that chain
    for (int i = 0; i < n; i++) { tup = tup || hs[i]->ntuple > 0; rset = rset || hs[i]->nclass_total > 0; }
    // (BayesR: the samplers of k_sweep_r, their class coefficients staged in LDS -- where that fits beside the sampler's own; else the chain of k_sweep_multi(_tup))
    const size_t lds_r = std::max(lds, lds_sampler + (size_t)NGP_SAMPLER_R_LDS);
    const bool use_r = rset && lds_r <= (size_t)160 * 1024 && !(h->knob & 65536);
    const size_t lds_launch = use_r ? lds_r : lds;
    HCHK(use_r ? sweep_multi_r_set_max_lds((int)lds_launch) : (tup ? sweep_multi_tup_set_max_lds((int)lds_launch) : sweep_multi_set_max_lds((int)lds_launch)));
    // One abort word: the leader's.  The sweep runs on the leader's stream; every chain's small kernels (head, coefficients, variance
    // draws, posterior sums: six launches of a few microseconds each) stay on the chain's OWN stream, tied to the sweep by events --
    // the K chains' small kernels then run side by side instead of one chain after the other (eight chains: 0.34 ms of a 4.35-ms pass).
    // knob bit 13: everything on the leader's stream, as the first version did.
    const bool serial = (h->knob & 8192) != 0;
    std::vector<hipStream_t> st((size_t)n);
    std::vector<unsigned *> ab((size_t)n);
    std::vector<hipEvent_t> evp((size_t)n, nullptr);
    hipEvent_t evs = nullptr;
    for (int i = 0; i < n; i++) HCHK(hipStreamSynchronize(hs[i]->stream));
    for (int i = 0; i < n; i++) {
        st[i] = hs[i]->stream; ab[i] = hs[i]->d_abort; hs[i]->d_abort = h->d_abort;
        if (serial) hs[i]->stream = h->stream;
    }
    auto restore = [&]() {
        for (int i = 0; i < n; i++) { hs[i]->stream = st[i]; hs[i]->d_abort = ab[i]; if (evp[i]) (void)hipEventDestroy(evp[i]); }
        if (evs) (void)hipEventDestroy(evs);
    };
    // (a scope guard: an exception on the way -- a std::string or std::vector that cannot allocate -- unwinds to the ABI's barrier
    // with every handle's own abort word and stream back in place; ngp_destroy would otherwise free the leader's buffer once per handle)
    struct Guard {
        decltype(restore) &f;
        bool armed = true;
        ~Guard() { if (armed) f(); }
        void now() { if (armed) { armed = false; f(); } }
    } guard{restore};
    hipError_t e = hipSuccess;
    if (!serial) {
        for (int i = 1; i < n && e == hipSuccess; i++) e = hipEventCreateWithFlags(&evp[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&evs, hipEventDisableTiming);
        if (e != hipSuccess) { guard.now(); return fail(h, NGP_ERR_HIP, std::string("fused run: ") + hipGetErrorString(e)); }
    }
    auto sync_all = [&]() {
        hipError_t r = hipStreamSynchronize(h->stream);
        if (!serial) for (int i = 1; i < n; i++) { hipError_t q = hipStreamSynchronize(hs[i]->stream); if (r == hipSuccess) r = q; }
        return r;
    };
    CuLease lease(h, grid);
    e = hipEventRecord(h->ev0, h->stream);
    rc = NGP_OK;
    for (int64_t it = 0; it < niter && rc == NGP_OK && e == hipSuccess; ++it) {
        MultiArgs M;
        M.K = n; M.pair = pair;
        for (int i = 0; i < n; i++) {
            iteration_pre(hs[i], it, false);
            fill_sweep_args(hs[i], 0, hs[i]->NBLK, M.a[i]);
            if (!serial && i > 0) {  // the sweep waits for this chain's coefficients (and cleared hand-off counters)
                (void)hipEventRecord(evp[i], hs[i]->stream);
                (void)hipStreamWaitEvent(h->stream, evp[i], 0);
            }
        }
        for (int i = 1; i < n; i++) { M.a[i].census = nullptr; M.a[i].xcc_w = M.a[0].xcc_w; }
        if (use_r) sweep_multi_r_launch((unsigned)grid, lds_launch, h->stream, M);
        else if (tup) sweep_multi_tup_launch((unsigned)grid, lds_launch, h->stream, M);
        else sweep_multi_launch((unsigned)grid, lds_launch, h->stream, M);
        h->sweep_launches += 1; h->last_grid = grid;
        if (!serial) {
            (void)hipEventRecord(evs, h->stream);
            for (int i = 1; i < n; i++) (void)hipStreamWaitEvent(hs[i]->stream, evs, 0);
        }
        for (int i = 0; i < n && rc == NGP_OK; i++) { rc = iteration_post(hs[i], it); if (rc && i) h->err = hs[i]->err; }
        if (rc) break;
        if ((it & 15) == 15 || it + 1 == niter) {  // bound the launch queue
            e = sync_all();
            if (e == hipSuccess) {
                rc = check_abort(h);  // (no retry: there is one grid, and the lease covers it)
                if (rc) for (int i = 1; i < n; i++) { hs[i]->poisoned = true; hs[i]->err = h->err; }
            }
        }
    }
    if (e == hipSuccess) e = sync_all();
    else (void)sync_all();
    if (e == hipSuccess) e = hipEventRecord(h->ev1, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    guard.now();
    if (e != hipSuccess) return fail(h, NGP_ERR_HIP, std::string("fused run: ") + hipGetErrorString(e));
    if (rc) return rc;
    float ms = 0.f;
    HCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    for (int i = 0; i < n; i++) { hs[i]->iter_ms += ms; hs[i]->iters_timed += niter; }
    for (int i = 0; i < n; i++)
        if ((rc = sample_flush(hs[i]))) { if (i) h->err = hs[i]->err; return rc; }
    return NGP_OK;
}
}  // namespace

extern "C" {

int32_t ngp_run(ngp_handle *h, int64_t niter) {
    NGP_TRY
    int rc;
    if ((rc = prepare_run(h, niter))) return rc;
    CuLease lease(h);
    HCHK(hipEventRecord(h->ev0, h->stream));
    if ((rc = run_iterations(h, niter, lease, nullptr))) return rc;
    HCHK(hipEventRecord(h->ev1, h->stream));
    HCHK(hipStreamSynchronize(h->stream));
    HCHK(hipGetLastError());
    float ms = 0.f;
    HCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->iter_ms += ms; h->iters_timed += niter;
    if ((rc = sample_flush(h))) return rc;
    if (h->dbg_mode != 0) return fail(h, NGP_ERR_DEBUG, "diagnostic timing mode " + std::to_string(h->dbg_mode) + " is active: the chain is invalid (ngp_debug_set_mode(h, 0) ends it)");
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_get_state(ngp_handle *h, double *ycorr, double *beta, int64_t *delta, double *varBeta, double *piHat, double *varE,
                      double *b, int64_t *iter) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_tiles != nullptr && h->have_y, NGP_ERR_STATE, "panel / y not set");
    HCHK(hipStreamSynchronize(h->stream));
    if (ycorr) HCHK(hipMemcpy(ycorr, h->d_ycorr, (size_t)h->N * sizeof(double), hipMemcpyDeviceToHost));
    if (beta) HCHK(hipMemcpy(beta, h->d_beta, (size_t)h->P * sizeof(double), hipMemcpyDeviceToHost));
    if (delta) {
        std::vector<uint8_t> d((size_t)h->P);
        HCHK(hipMemcpy(d.data(), h->d_delta, (size_t)h->P, hipMemcpyDeviceToHost));
        for (int64_t k = 0; k < h->P; k++) delta[k] = d[k];
    }
    if (varBeta && h->nvb) HCHK(hipMemcpy(varBeta, h->d_varBeta, (size_t)h->nvb * sizeof(double), hipMemcpyDeviceToHost));
    if (piHat && !h->sets.empty()) {
        std::vector<DSet> ds(h->sets.size());
        HCHK(hipMemcpy(ds.data(), h->d_sets, ds.size() * sizeof(DSet), hipMemcpyDeviceToHost));
        for (size_t s = 0; s < ds.size(); s++) { piHat[2 * s] = ds[s].piHat0; piHat[2 * s + 1] = ds[s].piHat1; }
    }
    DScal sc;
    HCHK(hipMemcpy(&sc, h->d_scal, sizeof(DScal), hipMemcpyDeviceToHost));
    if (varE) *varE = sc.varE;
    if (b) *b = sc.b;
    if (iter) *iter = h->iter;
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_set_state(ngp_handle *h, const double *ycorr, const double *beta, const int64_t *delta, const double *varBeta,
                      const double *piHat, double varE, double b, int64_t iter) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_tiles != nullptr && h->have_y, NGP_ERR_STATE, "panel / y not set");
    // (varE is 0 before the first iteration -- ngp_set_y -- and drawn before it is used; any later state has varE > 0)
    REQUIRE(std::isfinite(varE) && (varE > 0.0 || (varE == 0.0 && iter == 0)) && std::isfinite(b) && iter >= 0, NGP_ERR_ARG,
            "bad scalar state (varE must be finite and positive)");
    if (ycorr) HCHK(hipMemcpy(h->d_ycorr, ycorr, (size_t)h->N * sizeof(double), hipMemcpyHostToDevice));
    if (beta) HCHK(hipMemcpy(h->d_beta, beta, (size_t)h->P * sizeof(double), hipMemcpyHostToDevice));
    if (delta) {
        std::vector<uint8_t> d((size_t)h->P);
        for (int64_t k = 0; k < h->P; k++) d[k] = (uint8_t)(delta[k] != 0);
        HCHK(hipMemcpy(h->d_delta, d.data(), (size_t)h->P, hipMemcpyHostToDevice));
    }
    if (varBeta && h->nvb) HCHK(hipMemcpy(h->d_varBeta, varBeta, (size_t)h->nvb * sizeof(double), hipMemcpyHostToDevice));
    if (piHat)
        for (size_t s = 0; s < h->sets.size(); s++)
            hipLaunchKernelGGL(k_set_pi, dim3(1), dim3(1), 0, h->stream, h->d_sets, (int)s, piHat[2 * s], piHat[2 * s + 1]);
    DScal sc;
    HCHK(hipMemcpy(&sc, h->d_scal, sizeof(DScal), hipMemcpyDeviceToHost));
    sc.varE = varE; sc.iVarE = 1.0 / varE; sc.b = b;
    HCHK(hipMemcpy(h->d_scal, &sc, sizeof(DScal), hipMemcpyHostToDevice));
    HCHK(hipStreamSynchronize(h->stream));
    h->iter = iter;
    if (ycorr && beta) h->poisoned = false;
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_get_trace(ngp_handle *h, double *varE, double *b, int64_t n) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    n = std::min(n, h->ntrace);
    if (n <= 0) return NGP_OK;
    if (varE) HCHK(hipMemcpy(varE, h->d_tr_varE, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    if (b) HCHK(hipMemcpy(b, h->d_tr_b, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_get_posterior_sums(ngp_handle *h, double *sum_beta, double *sum_beta2, double *sum_delta, double *sum_varBeta,
                               double *sum_pi, double *sum_varE, double *sum_b, int64_t *nKept) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_tiles != nullptr, NGP_ERR_STATE, "panel not set");
    HCHK(hipStreamSynchronize(h->stream));
    const size_t pb = (size_t)h->P * sizeof(double);
    if (sum_beta) HCHK(hipMemcpy(sum_beta, h->d_sum_beta, pb, hipMemcpyDeviceToHost));
    if (sum_beta2) HCHK(hipMemcpy(sum_beta2, h->d_sum_beta2, pb, hipMemcpyDeviceToHost));
    if (sum_delta) HCHK(hipMemcpy(sum_delta, h->d_sum_delta, pb, hipMemcpyDeviceToHost));
    if (sum_varBeta && h->nvb) HCHK(hipMemcpy(sum_varBeta, h->d_sum_varBeta, (size_t)h->nvb * sizeof(double), hipMemcpyDeviceToHost));
    if (sum_pi && !h->sets.empty()) {
        std::vector<DSet> ds(h->sets.size());
        HCHK(hipMemcpy(ds.data(), h->d_sets, ds.size() * sizeof(DSet), hipMemcpyDeviceToHost));
        for (size_t s = 0; s < ds.size(); s++) { sum_pi[2 * s] = ds[s].sum_pi0; sum_pi[2 * s + 1] = ds[s].sum_pi1; }
    }
    DScal sc;
    HCHK(hipMemcpy(&sc, h->d_scal, sizeof(DScal), hipMemcpyDeviceToHost));
    if (sum_varE) *sum_varE = sc.sum_varE;
    if (sum_b) *sum_b = sc.sum_b;
    if (nKept) *nKept = sc.nKept;
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_posterior_len(ngp_handle *h, int64_t *len) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(len != nullptr, NGP_ERR_ARG, "null len");
    *len = posterior_words(h);
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_export_posterior_device(ngp_handle *h, void *device_ptr, int64_t len) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_tiles != nullptr, NGP_ERR_STATE, "panel not set");
    const int64_t need = posterior_words(h);
    REQUIRE(device_ptr && len == need, NGP_ERR_ARG, "export buffer length mismatch (see ngp_posterior_len)");
    double *o = (double *)device_ptr;
    const size_t pb = (size_t)h->P * sizeof(double);
    HCHK(hipMemcpyAsync(o, h->d_sum_beta, pb, hipMemcpyDeviceToDevice, h->stream));
    HCHK(hipMemcpyAsync(o + h->P, h->d_sum_beta2, pb, hipMemcpyDeviceToDevice, h->stream));
    HCHK(hipMemcpyAsync(o + 2 * h->P, h->d_sum_delta, pb, hipMemcpyDeviceToDevice, h->stream));
    if (h->nvb) HCHK(hipMemcpyAsync(o + 3 * h->P, h->d_sum_varBeta, (size_t)h->nvb * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    std::vector<DSet> ds(h->sets.size());
    HCHK(hipStreamSynchronize(h->stream));
    if (!ds.empty()) HCHK(hipMemcpy(ds.data(), h->d_sets, ds.size() * sizeof(DSet), hipMemcpyDeviceToHost));
    DScal sc;
    HCHK(hipMemcpy(&sc, h->d_scal, sizeof(DScal), hipMemcpyDeviceToHost));
    std::vector<double> tail;
    for (auto &s : ds) { tail.push_back(s.sum_pi0); tail.push_back(s.sum_pi1); }
    for (auto &s : ds) for (int v = 0; v < s.K; v++) tail.push_back(s.sum_pic[v]);  // BayesR class probabilities, set by set
    const size_t nfix_at = tail.size();
    tail.resize(nfix_at + (size_t)h->nfixcol);  // fixed-effect sums beyond the intercept (all columns of all sets, in order)
    if (h->nfixcol > 0) HCHK(hipMemcpy(tail.data() + nfix_at, h->d_sum_bfix, (size_t)h->nfixcol * sizeof(double), hipMemcpyDeviceToHost));
    tail.push_back(sc.sum_varE); tail.push_back(sc.sum_b); tail.push_back((double)sc.nKept);
    HCHK(hipMemcpy(o + 3 * h->P + h->nvb, tail.data(), tail.size() * sizeof(double), hipMemcpyHostToDevice));
    return NGP_OK;
    NGP_CATCH(h)
}

namespace {
// one fine-seam call; dev: the caller's arrays (and piHat) are DEVICE memory of this handle's device, delta is int64 there too
int sweep_set_impl(ngp_handle *h, int32_t set_id, double varE, double *ycorr, double *beta, int64_t *delta, double *varBeta, double *piHat, bool dev) {
    int rc;
    REQUIRE(h->d_tiles != nullptr, NGP_ERR_STATE, "panel not set");
    REQUIRE(set_id >= 0 && set_id < (int)h->sets.size(), NGP_ERR_ARG, "unknown set id");
    REQUIRE(ycorr && beta && varBeta, NGP_ERR_ARG, "null state pointer");
    REQUIRE(std::isfinite(varE) && varE > 0.0, NGP_ERR_ARG, "varE must be finite and positive");
    if ((rc = sync_tables(h))) return rc;
    if ((rc = sync_linear_blocks(h, (int)set_id))) return rc;
    HSet &hs = h->sets[set_id];
    const int64_t nvbs = (int64_t)hs.vb0.size();  // variance entries of the set: regions (loci for BayesB), k x k per region for a tuple set
    // (device arrays are not read back to be looked at: a variance that is not finite poisons the chain visibly -- k_prep -- as in ngp_run)
    if (!dev) for (int64_t r = 0; r < nvbs; r++) REQUIRE(std::isfinite(varBeta[r]) && (varBeta[r] >= 0.0 || hs.tk > 1), NGP_ERR_ARG, "varBeta must be finite, >= 0");
    const bool has_pi = hs.method != NGP_METHOD_BAYESPR && hs.method != NGP_METHOD_TUPLE;
    if (has_pi) REQUIRE(piHat != nullptr, NGP_ERR_ARG, "BayesB / BayesC need piHat");
    const hipMemcpyKind in = dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, out = dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    const uint64_t it = ++hs.fine_calls;
    const int64_t tb0 = hs.col0 / NGP_BLK, tb1 = (hs.col0 + hs.ncol - 1) / NGP_BLK + 1;
    CuLease lease(h);
    for (int attempt = 0;; ++attempt) {
        HCHK(hipMemsetAsync(h->d_ycorr, 0, (size_t)h->L * sizeof(double), h->stream));
        HCHK(hipMemcpyAsync(h->d_ycorr, ycorr, (size_t)h->N * sizeof(double), in, h->stream));
        HCHK(hipMemcpyAsync(h->d_beta + hs.col0, beta, (size_t)hs.ncol * sizeof(double), in, h->stream));
        HCHK(hipMemcpyAsync(h->d_varBeta + hs.vb_off, varBeta, (size_t)nvbs * sizeof(double), in, h->stream));
        if (has_pi) {
            if (dev) hipLaunchKernelGGL(k_set_pi_dev, dim3(1), dim3(1), 0, h->stream, h->d_sets, (int)set_id, (const double *)piHat);
            else hipLaunchKernelGGL(k_set_pi, dim3(1), dim3(1), 0, h->stream, h->d_sets, (int)set_id, piHat[0], piHat[1]);
        }
        hipLaunchKernelGGL(k_set_varE, dim3(1), dim3(1), 0, h->stream, h->d_scal, varE);
        // (no draws, no intercept: ycorr'ycorr of the caller's residual sets the scale of the fixed-point accumulators)
        hipLaunchKernelGGL(k_head, dim3(1), dim3(1024), 0, h->stream, h->d_ycorr, (long long)h->L, (long long)h->N, h->d_scal, h->e_df,
                           h->e_scale, 0, 0, h->seed, (uint64_t)h->chain, it, (double *)nullptr, (double *)nullptr, (long long)0, h->d_abort, h->mpm_max);
        hipLaunchKernelGGL(k_prep, dim3((unsigned)(h->Ppad / 256 + 1)), dim3(256), 0, h->stream, (long long)h->Ppad, h->d_setof, h->d_loc,
                           h->d_vbidx, h->d_sets, h->d_scal, h->d_varBeta, h->d_mpm, h->d_lhs0, h->d_rhs0, h->d_beta, h->d_c, h->d_w,
                           h->d_q, h->d_T, h->d_chi, (int)set_id, h->seed, (uint64_t)h->chain, it, (long long)h->h_regs.size(), h->d_regs, h->d_regchi, h->d_rcls,
                           h->d_ccnt, (long long)(h->mode == 1 ? h->ccnt_words : 0), h->d_abort, h->d_tup, h->d_tupc, h->d_tupg);
        launch_tinv(h);
        launch_sweep(h, tb0, tb1, nullptr);
        launch_variance(h, (int)set_id, it);
        HCHK(hipStreamSynchronize(h->stream));
        HCHK(hipGetLastError());
        int64_t itf = 0;
        rc = check_abort(h, &itf);
        if (rc == NGP_RETRY_CENSUS && attempt == 0) {  // nothing was changed (the caller's arrays are the state): once more, alone on the device
            h->exclusive = true; h->census_retries += 1;
            lease.make_exclusive();
            continue;
        }
        if (rc) return rc;
        break;
    }
    HCHK(hipMemcpyAsync(ycorr, h->d_ycorr, (size_t)h->N * sizeof(double), out, h->stream));
    HCHK(hipMemcpyAsync(beta, h->d_beta + hs.col0, (size_t)hs.ncol * sizeof(double), out, h->stream));
    HCHK(hipMemcpyAsync(varBeta, h->d_varBeta + hs.vb_off, (size_t)nvbs * sizeof(double), out, h->stream));
    if (dev) {
        if (delta) hipLaunchKernelGGL(k_delta_widen, dim3((unsigned)((hs.ncol + 255) / 256)), dim3(256), 0, h->stream, (const uint8_t *)(h->d_delta + hs.col0), (long long *)delta, (long long)hs.ncol);
        if (piHat) hipLaunchKernelGGL(k_get_pi_dev, dim3(1), dim3(1), 0, h->stream, (const DSet *)h->d_sets, (int)set_id, piHat);
    }
    HCHK(hipStreamSynchronize(h->stream));
    HCHK(hipGetLastError());
    if (h->dbg_mode != 0) return fail(h, NGP_ERR_DEBUG, "diagnostic timing mode is active: the sweep is invalid");
    if (dev) return NGP_OK;
    if (delta) {
        std::vector<uint8_t> d((size_t)hs.ncol);
        HCHK(hipMemcpy(d.data(), h->d_delta + hs.col0, (size_t)hs.ncol, hipMemcpyDeviceToHost));
        for (int64_t k = 0; k < hs.ncol; k++) delta[k] = d[k];
    }
    if (piHat) {
        DSet ds;
        HCHK(hipMemcpy(&ds, h->d_sets + set_id, sizeof(DSet), hipMemcpyDeviceToHost));
        piHat[0] = ds.piHat0; piHat[1] = ds.piHat1;
    }
    return NGP_OK;
}
}  // namespace

int32_t ngp_sweep_set(ngp_handle *h, int32_t set_id, double varE, double *ycorr, double *beta, int64_t *delta, double *varBeta,
                      double *piHat) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    return sweep_set_impl(h, set_id, varE, ycorr, beta, delta, varBeta, piHat, false);
    NGP_CATCH(h)
}

/* The fine seam with the caller's state in DEVICE memory (a host that keeps ycorr, beta, varBeta on the GPU between the calls of
 * src/samplers.jl:52 -- ROCArrays -- pays no PCIe round trip per set and iteration): device-to-device copies on the handle's stream. */
int32_t ngp_sweep_set_dev(ngp_handle *h, int32_t set_id, double varE, void *d_ycorr, void *d_beta, void *d_delta, void *d_varBeta, void *d_piHat) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    return sweep_set_impl(h, set_id, varE, (double *)d_ycorr, (double *)d_beta, (int64_t *)d_delta, (double *)d_varBeta, (double *)d_piHat, true);
    NGP_CATCH(h)
}

int32_t ngp_get_timing(ngp_handle *h, int64_t *sweep_launches, double *iter_ms, int64_t *iters) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    if (sweep_launches) *sweep_launches = h->sweep_launches;
    if (iter_ms) *iter_ms = h->iter_ms;
    if (iters) *iters = h->iters_timed;
    h->sweep_launches = 0; h->iter_ms = 0; h->iters_timed = 0;
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_profile_iteration(ngp_handle *h, double *avg_ms, int64_t *launches, double *bytes_per_launch) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    if ((rc = ready(h))) return rc;
    const int64_t n = (h->mode == 1) ? 1 : h->NBLK;
    std::vector<hipEvent_t> evs((size_t)(2 * n));
    for (auto &e : evs) HCHK(hipEventCreate(&e));
    if (h->trace_cap < 1) {
        if ((rc = dalloc(h, &h->d_tr_varE, 1))) return rc;
        if ((rc = dalloc(h, &h->d_tr_b, 1))) return rc;
        h->trace_cap = 1;
    }
    h->ntrace = 1;
    CuLease lease(h);
    rc = run_iterations(h, 1, lease, evs.data());  // (checks the abort word; a launch that ends at its census is run again)
    hipError_t e = hipStreamSynchronize(h->stream);
    double tot = 0.0;
    if (rc == NGP_OK && e == hipSuccess)
        for (int64_t i = 0; i < n; i++) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, evs[2 * i], evs[2 * i + 1]);
            tot += ms;
        }
    for (auto &ev : evs) (void)hipEventDestroy(ev);
    if (e != hipSuccess) return fail(h, NGP_ERR_HIP, std::string("profile_iteration: ") + hipGetErrorString(e));
    if (avg_ms) *avg_ms = tot / (double)n;
    if (launches) *launches = n;
    const double bpe = (h->storage == 1) ? 1.0 : 4.0;  // algorithmic bytes per genotype: the panel is read once per iteration
    if (bytes_per_launch) *bytes_per_launch = (h->mode == 1) ? (double)h->N * (double)h->P * bpe : (double)h->N * NGP_BLK * bpe;
    return rc;
    NGP_CATCH(h)
}

int32_t ngp_debug_stamps(ngp_handle *h, int32_t enable, uint64_t *out, int64_t n) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    const size_t words = (size_t)2 << 20;
    if (enable && !h->d_dbg) { if ((rc = dalloc(h, &h->d_dbg, words))) return rc; HCHK(hipStreamSynchronize(h->stream)); }
    if (out && h->d_dbg) HCHK(hipMemcpy(out, h->d_dbg, std::min<size_t>((size_t)n, words) * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (!enable && h->d_dbg) { HCHK(hipStreamSynchronize(h->stream)); dfree(h->d_dbg); }
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_configure(ngp_handle *h, int32_t mode, int32_t lag) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_tiles == nullptr, NGP_ERR_STATE, "ngp_configure must precede the panel upload");
    REQUIRE(mode == 0 || mode == 1, NGP_ERR_ARG, "mode must be 0 (per-block launches) or 1 (persistent sweep)");
    REQUIRE(lag >= 1 && lag <= NGP_MAX_LAG, NGP_ERR_ARG, "lag must be in 1..12 (above 8: compact storage)");
    h->mode = mode; h->lag = lag; h->lag_auto = false;
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_set_near_lags(ngp_handle *h, int32_t near) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_tiles == nullptr, NGP_ERR_STATE, "ngp_set_near_lags must precede the panel upload");
    REQUIRE(near >= 0 && near <= 4, NGP_ERR_ARG, "near lags: 0 (automatic) or 1..4");
    h->near_req = near;
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_get_near_lags(ngp_handle *h, int32_t *near) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    if (near) *near = h->near;
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_set_chain_form(ngp_handle *h, int32_t form) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(form == 0 || form == 1, NGP_ERR_ARG, "chain form: 0 (64 steps per block) or 1 (linear blocks by the inverse form)");
    h->chain_form = form;
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_get_chain_form(ngp_handle *h, int32_t *form) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    if (form) *form = h->chain_form;
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_get_config(ngp_handle *h, int32_t *mode, int32_t *lag) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    if (mode) *mode = h->mode;
    if (lag) *lag = h->D;
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_draws_indexed(ngp_handle *h, uint64_t iter, uint64_t kind, uint64_t index0, int32_t what, double p1, double p2, int64_t n,
                          double *out) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(out && n > 0, NGP_ERR_ARG, "bad output buffer");
    double *d = nullptr;
    if ((rc = dalloc(h, &d, (size_t)n))) return rc;
    hipLaunchKernelGGL(k_draws_indexed, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->seed, (uint64_t)h->chain, iter, kind,
                       index0, what, p1, p2, (long long)n, d);
    hipError_t e = hipMemcpyAsync(out, d, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    dfree(d);
    if (e != hipSuccess) return fail(h, NGP_ERR_HIP, std::string("draws: ") + hipGetErrorString(e));
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_eval_math(ngp_handle *h, int32_t which, const double *in, int64_t n, double *out) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(in && out && n > 0, NGP_ERR_ARG, "bad buffers");
    double *di = nullptr, *dout = nullptr;
    if ((rc = dalloc(h, &di, (size_t)n))) return rc;
    if ((rc = dalloc(h, &dout, (size_t)n))) { dfree(di); return rc; }
    hipError_t e = hipMemcpyAsync(di, in, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream);
    hipLaunchKernelGGL(k_eval_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, which, di, (long long)n, dout);
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    dfree(di); dfree(dout);
    if (e != hipSuccess) return fail(h, NGP_ERR_HIP, std::string("eval_math: ") + hipGetErrorString(e));
    return NGP_OK;
    NGP_CATCH(h)
}


/* ------------------------------------------------------------------------------------------------
 * round 2 additions: diagnostics out of the environment, posterior-sum restore, snapshots, traces,
 * streamer variants, pooled posterior sums
 * ---------------------------------------------------------------------------------------------- */
int32_t ngp_debug_set_mode(ngp_handle *h, int32_t mode) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(mode >= 0 && mode <= 6, NGP_ERR_ARG, "diagnostic mode must be in 0..6");
    h->dbg_mode = mode;
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_debug_set_knob(ngp_handle *h, int32_t knob) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    h->knob = knob;
    h->gram_engine = (knob & 1024) ? 1 : 0;  // bit 10: build the Gram window on the matrix cores instead of the fp64 VALU kernel (same bits)
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_set_streamer(ngp_handle *h, int32_t variant) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_tiles == nullptr, NGP_ERR_STATE, "ngp_set_streamer must precede the panel upload");
    REQUIRE((variant >= 0 && variant <= 2) || variant == 4 || variant == 6, NGP_ERR_ARG,
            "streamer variant: 0 (automatic), 1 (phase streamer), 2 (row-owning waves) or 4 / 6 (row-owning waves, two / three shards per workgroup)");
    h->streamer_req = variant;
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_set_max_shards(ngp_handle *h, int32_t max_shards) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_tiles == nullptr, NGP_ERR_STATE, "ngp_set_max_shards must precede the panel upload");
    REQUIRE(max_shards >= 0, NGP_ERR_ARG, "max_shards: 0 (automatic) or a positive number of streamer workgroups");
    h->max_shards_req = max_shards;
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_shards_for_chains(ngp_handle *h, int32_t chains, int32_t *max_shards) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(chains >= 1 && max_shards, NGP_ERR_ARG, "chains must be >= 1");
    // workgroups go to the 8 XCDs in turn: `chains` grids are co-resident when each takes at most (CUs / 8) / chains CUs per XCD;
    // a grid is 1 sampler + ceil(S / 32) reducers + S streamers
    const int per = 8 * ((h->cu_count / 8) / chains);
    const int s = per - 1 - (per + NGP_GRP - 1) / NGP_GRP;
    REQUIRE(s >= 1, NGP_ERR_ARG, "too many chains for this device");
    *max_shards = s;
    return NGP_OK;
    NGP_CATCH(h)
}

/* K chains per pass, set-up: h takes `owner`'s panel (tiles, Gram window, x'x, column means) by reference -- no copy, no second
 * 120 GB -- together with its engine, layout and storage; everything that belongs to a chain (effects, residuals, variances, draws,
 * hand-off rings, posterior sums) is h's own.  The arrays live as long as any handle refers to them. */
int32_t ngp_share_panel(ngp_handle *h, ngp_handle *owner) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(owner && owner != h && owner->d_tiles != nullptr && owner->pm != nullptr, NGP_ERR_ARG, "ngp_share_panel: the owner has no panel");
    REQUIRE(owner->device == h->device, NGP_ERR_ARG, "ngp_share_panel: both handles must be on one device");
    REQUIRE(!owner->panel_open, NGP_ERR_STATE, "ngp_share_panel: the owner's panel is still open (ngp_end_panel)");
    HCHK(hipStreamSynchronize(owner->stream));
    h->cu_count = owner->cu_count;
    return alloc_panel(h, owner->N, owner->P, owner);
    NGP_CATCH(h)
}

/* The largest max_shards (ngp_set_max_shards, before the panel is set) with which `chains` chains share ONE fused sweep launch:
 * chains x (1 sampler + ceil(S / 32) reducers) + S streamers <= CUs. */
int32_t ngp_shards_for_pass(ngp_handle *h, int32_t chains, int32_t *max_shards) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(chains >= 1 && chains <= NGP_MAXC && max_shards, NGP_ERR_ARG, "chains per pass: 1..8");
    int s = h->cu_count;
    const int pair = (chains >= NGP_PAIR_FROM && !(h->knob & 4096)) || (h->knob & 2048) ? 1 : 0;  // (fused_pair, for the phase streamer this serves)
    while (s >= 1 && (int64_t)chains + ngp_multi_reducers(chains, (s + NGP_GRP - 1) / NGP_GRP, pair) + s > h->cu_count) --s;
    REQUIRE(s >= 1, NGP_ERR_ARG, "too many chains for this device");
    *max_shards = s;
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_set_storage(ngp_handle *h, int32_t storage) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_tiles == nullptr, NGP_ERR_STATE, "ngp_set_storage must precede the panel upload");
    REQUIRE(storage == NGP_STORAGE_F32 || storage == NGP_STORAGE_U8, NGP_ERR_ARG, "storage: 0 (fp32 tiles) or 1 (compact: bytes + column means)");
    h->storage = storage;
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_get_storage(ngp_handle *h, int32_t *storage, double *means, int64_t P) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    if (storage) *storage = h->storage;
    if (means) {
        REQUIRE(h->d_tiles != nullptr && h->d_mean != nullptr, NGP_ERR_STATE, "column means exist after the panel is set");
        REQUIRE(P == h->P, NGP_ERR_ARG, "means buffer must hold P entries");
        HCHK(hipStreamSynchronize(h->stream));
        HCHK(hipMemcpy(means, h->d_mean, (size_t)P * sizeof(double), hipMemcpyDeviceToHost));
    }
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_get_streamer(ngp_handle *h, int32_t *variant, int32_t *gemv_chains) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    if (variant) *variant = (h->mode == 1) ? h->streamer : 0;
    if (gemv_chains) *gemv_chains = h->nchain;
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_set_posterior_sums(ngp_handle *h, const double *sum_beta, const double *sum_beta2, const double *sum_delta,
                               const double *sum_varBeta, const double *sum_pi, double sum_varE, double sum_b, int64_t nKept) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_tiles != nullptr && h->have_y, NGP_ERR_STATE, "panel / y not set");
    REQUIRE(sum_beta && sum_beta2 && sum_delta && nKept >= 0 && std::isfinite(sum_varE) && std::isfinite(sum_b), NGP_ERR_ARG, "bad posterior sums");
    REQUIRE(h->nvb == 0 || sum_varBeta, NGP_ERR_ARG, "sum_varBeta missing");
    REQUIRE(h->sets.empty() || sum_pi, NGP_ERR_ARG, "sum_pi missing");
    HCHK(hipStreamSynchronize(h->stream));
    const size_t pb = (size_t)h->P * sizeof(double);
    HCHK(hipMemcpy(h->d_sum_beta, sum_beta, pb, hipMemcpyHostToDevice));
    HCHK(hipMemcpy(h->d_sum_beta2, sum_beta2, pb, hipMemcpyHostToDevice));
    HCHK(hipMemcpy(h->d_sum_delta, sum_delta, pb, hipMemcpyHostToDevice));
    if (h->nvb) HCHK(hipMemcpy(h->d_sum_varBeta, sum_varBeta, (size_t)h->nvb * sizeof(double), hipMemcpyHostToDevice));
    for (size_t si = 0; si < h->sets.size(); si++)
        hipLaunchKernelGGL(k_set_sum_pi, dim3(1), dim3(1), 0, h->stream, h->d_sets, (int)si, sum_pi[2 * si], sum_pi[2 * si + 1]);
    HCHK(hipStreamSynchronize(h->stream));
    DScal sc;
    HCHK(hipMemcpy(&sc, h->d_scal, sizeof(DScal), hipMemcpyDeviceToHost));
    sc.sum_varE = sum_varE; sc.sum_b = sum_b; sc.nKept = nKept;
    HCHK(hipMemcpy(h->d_scal, &sc, sizeof(DScal), hipMemcpyHostToDevice));
    return NGP_OK;
    NGP_CATCH(h)
}

/* Snapshot file: the chain state and the posterior sums, little-endian, no padding:
 *   char[8] "NGPSNAP2" | int64 N, P, nvb, nsets, iter, nKept | uint64 seed | uint64 chain |
 *   model signature: per marker set int64 {method, K, nreg, col0, ncol} | int64 nfixsets | per fixed-effect set int64 ncol |
 *   double varE, b, sum_varE, sum_b | ycorr[N] | beta[P] | delta[P] (uint8) | varBeta[nvb] | piHat[2 nsets] |
 *   sum_beta[P] | sum_beta2[P] | sum_delta[P] | sum_varBeta[nvb] | sum_pi[2 nsets] | fine_calls[nsets] (uint64) |
 *   int64 nfix | b_fixed[nfix] | sum_b_fixed[nfix] | per BayesR set: piHat[K] | sum_pi[K]
 * It plays the role of the reference's append-only *Out files for a resumed run (src/outFiles.jl:17-21): what was kept
 * before the interruption is not lost. */
int32_t ngp_save_snapshot(ngp_handle *h, const char *path) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_tiles != nullptr && h->have_y, NGP_ERR_STATE, "panel / y not set");
    REQUIRE(path != nullptr, NGP_ERR_ARG, "null path");
    REQUIRE(!h->poisoned, NGP_ERR_STATE, "the chain state is invalid (abandoned sweep)");
    const size_t N = (size_t)h->N, P = (size_t)h->P, nvb = (size_t)h->nvb, ns = h->sets.size();
    std::vector<double> yc(N), be(P), vb(std::max<size_t>(nvb, 1)), pi(2 * std::max<size_t>(ns, 1)), sb(P), sb2(P), sd(P), sv(std::max<size_t>(nvb, 1)),
        sp(2 * std::max<size_t>(ns, 1));
    std::vector<int64_t> de(P);
    double varE = 0, b = 0, svE = 0, sbb = 0;
    int64_t iter = 0, nk = 0;
    if ((rc = ngp_get_state(h, yc.data(), be.data(), de.data(), vb.data(), pi.data(), &varE, &b, &iter))) return rc;
    if ((rc = ngp_get_posterior_sums(h, sb.data(), sb2.data(), sd.data(), sv.data(), sp.data(), &svE, &sbb, &nk))) return rc;
    std::vector<uint8_t> d8(P);
    for (size_t k = 0; k < P; k++) d8[k] = (uint8_t)(de[k] != 0);
    const std::string tmp = std::string(path) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return fail(h, NGP_ERR_ARG, "cannot open " + tmp + " for writing");
    bool ok = true;
    auto W = [&](const void *p, size_t n) { if (n && fwrite(p, 1, n, f) != n) ok = false; };
    const int64_t hdr[6] = {h->N, h->P, h->nvb, (int64_t)ns, iter, nk};
    const uint64_t ids[2] = {h->seed, (uint64_t)h->chain};
    const double scal[4] = {varE, b, svE, sbb};
    W("NGPSNAP2", 8); W(hdr, sizeof(hdr)); W(ids, sizeof(ids));
    {   // model signature: a snapshot only loads into the model it was taken from (equal counts are not enough)
        for (auto &hs : h->sets) { const int64_t sg[5] = {hs.method, hs.K + 16 * hs.tk, hs.nreg, hs.col0, hs.ncol}; W(sg, sizeof(sg)); }
        const int64_t nfs = (int64_t)h->fix.size();
        W(&nfs, 8);
        for (auto &fx : h->fix) W(&fx.ncol, 8);
    }
    W(scal, sizeof(scal));
    W(yc.data(), N * 8); W(be.data(), P * 8); W(d8.data(), P); W(vb.data(), nvb * 8); W(pi.data(), 2 * ns * 8);
    W(sb.data(), P * 8); W(sb2.data(), P * 8); W(sd.data(), P * 8); W(sv.data(), nvb * 8); W(sp.data(), 2 * ns * 8);
    for (auto &hs : h->sets) W(&hs.fine_calls, 8);
    {   // fixed-effect sets beyond the intercept: effects and their posterior sums
        std::vector<double> fb((size_t)std::max<int64_t>(h->nfixcol, 1)), fs((size_t)std::max<int64_t>(h->nfixcol, 1));
        int64_t nf = 0;
        if ((rc = ngp_get_fixed(h, fb.data(), fs.data(), &nf))) { fclose(f); remove(tmp.c_str()); return rc; }
        W(&nf, 8); W(fb.data(), (size_t)nf * 8); W(fs.data(), (size_t)nf * 8);
    }
    for (size_t si = 0; si < ns; si++)  // BayesR sets: class probabilities and their posterior sums (K each)
        if (h->sets[si].K > 0) {
            double cp[NGP_RMAX], cs[NGP_RMAX];
            int64_t K = 0;
            if ((rc = ngp_get_class_state(h, (int32_t)si, cp, cs, &K))) { fclose(f); remove(tmp.c_str()); return rc; }
            W(cp, (size_t)K * 8); W(cs, (size_t)K * 8);
        }
    if (fclose(f) != 0) ok = false;
    if (!ok || rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); return fail(h, NGP_ERR_ARG, std::string("writing the snapshot failed: ") + path); }
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_load_snapshot(ngp_handle *h, const char *path) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_tiles != nullptr && h->have_y, NGP_ERR_STATE, "panel / y not set (build the model first, then load the snapshot)");
    REQUIRE(path != nullptr, NGP_ERR_ARG, "null path");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(h, NGP_ERR_ARG, std::string("cannot open snapshot ") + path);
    bool ok = true;
    auto Rd = [&](void *p, size_t n) { if (n && fread(p, 1, n, f) != n) ok = false; };
    char magic[8]; int64_t hdr[6] = {0, 0, 0, 0, 0, 0}; uint64_t ids[2] = {0, 0}; double scal[4] = {0, 0, 0, 0};
    Rd(magic, 8); Rd(hdr, sizeof(hdr)); Rd(ids, sizeof(ids));
    if (!ok || memcmp(magic, "NGPSNAP2", 8) != 0) { fclose(f); return fail(h, NGP_ERR_ARG, "not a snapshot file (bad magic or truncated header)"); }
    if (hdr[0] != h->N || hdr[1] != h->P || hdr[2] != h->nvb || hdr[3] != (int64_t)h->sets.size() || hdr[4] < 0 || hdr[5] < 0) {
        fclose(f);
        return fail(h, NGP_ERR_ARG, "snapshot does not match the model of this handle (N, P, variance components or marker sets differ)");
    }
    {   // model signature
        bool same = true;
        for (auto &hs : h->sets) {
            int64_t sg[5] = {-1, -1, -1, -1, -1};
            Rd(sg, sizeof(sg));
            same = same && sg[0] == hs.method && sg[1] == hs.K + 16 * hs.tk && sg[2] == hs.nreg && sg[3] == hs.col0 && sg[4] == hs.ncol;
        }
        int64_t nfs = -1;
        Rd(&nfs, 8);
        same = same && ok && nfs == (int64_t)h->fix.size();
        if (same)
            for (auto &fx : h->fix) { int64_t nc = -1; Rd(&nc, 8); same = same && nc == fx.ncol; }
        if (!ok) { fclose(f); return fail(h, NGP_ERR_ARG, "snapshot file is truncated (model signature)"); }
        if (!same) {
            fclose(f);
            return fail(h, NGP_ERR_ARG, "snapshot does not match the model of this handle (methods, classes, regions or fixed-effect sets differ)");
        }
    }
    Rd(scal, sizeof(scal));
    if (ok && !(std::isfinite(scal[0]) && (scal[0] > 0.0 || (scal[0] == 0.0 && hdr[4] == 0)))) { fclose(f); return fail(h, NGP_ERR_ARG, "snapshot holds an invalid residual variance"); }
    const size_t N = (size_t)h->N, P = (size_t)h->P, nvb = (size_t)h->nvb, ns = h->sets.size();
    std::vector<double> yc(N), be(P), vb(std::max<size_t>(nvb, 1)), pi(2 * std::max<size_t>(ns, 1)), sb(P), sb2(P), sd(P), sv(std::max<size_t>(nvb, 1)),
        sp(2 * std::max<size_t>(ns, 1));
    std::vector<uint8_t> d8(P);
    std::vector<uint64_t> fc(std::max<size_t>(ns, 1));
    Rd(yc.data(), N * 8); Rd(be.data(), P * 8); Rd(d8.data(), P); Rd(vb.data(), nvb * 8); Rd(pi.data(), 2 * ns * 8);
    Rd(sb.data(), P * 8); Rd(sb2.data(), P * 8); Rd(sd.data(), P * 8); Rd(sv.data(), nvb * 8); Rd(sp.data(), 2 * ns * 8); Rd(fc.data(), ns * 8);
    int64_t nf = -1;
    Rd(&nf, 8);
    if (ok && nf != h->nfixcol) { fclose(f); return fail(h, NGP_ERR_ARG, "snapshot does not match the model of this handle (fixed-effect columns differ)"); }
    std::vector<double> fb((size_t)std::max<int64_t>(h->nfixcol, 1)), fs((size_t)std::max<int64_t>(h->nfixcol, 1));
    Rd(fb.data(), (size_t)h->nfixcol * 8); Rd(fs.data(), (size_t)h->nfixcol * 8);
    std::vector<double> cls((size_t)2 * std::max<int64_t>(h->nclass_total, 1));
    Rd(cls.data(), (size_t)2 * h->nclass_total * 8);
    char extra;
    const bool at_end = fread(&extra, 1, 1, f) == 0;
    fclose(f);
    if (!ok || !at_end) return fail(h, NGP_ERR_ARG, "snapshot file is truncated or has trailing bytes");
    std::vector<int64_t> de(P);
    for (size_t k = 0; k < P; k++) de[k] = d8[k];
    h->poisoned = true;  // until the whole restore has gone through: a failure half-way must not leave a mixed state behind as valid
    if ((rc = ngp_set_state(h, yc.data(), be.data(), de.data(), vb.data(), pi.data(), scal[0], scal[1], hdr[4]))) { h->poisoned = true; return rc; }
    h->poisoned = true;  // (ngp_set_state has just declared the state valid: not before the sums, fixed effects and classes are in)
    if ((rc = ngp_set_posterior_sums(h, sb.data(), sb2.data(), sd.data(), sv.data(), sp.data(), scal[2], scal[3], hdr[5]))) return rc;
    for (size_t si = 0; si < ns; si++) h->sets[si].fine_calls = fc[si];
    if (h->nfixcol > 0 && (rc = ngp_set_fixed(h, fb.data(), fs.data(), h->nfixcol))) return rc;
    {
        size_t off = 0;
        for (size_t si = 0; si < ns; si++)
            if (h->sets[si].K > 0) {
                const size_t K = (size_t)h->sets[si].K;
                if ((rc = set_class_state_dev(h, (int)si, cls.data() + off, cls.data() + off + K))) return rc;
                off += 2 * K;
            }
    }
    h->seed = ids[0]; h->chain = (uint32_t)ids[1];  // the draws continue the interrupted chain's streams
    h->poisoned = false;
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_set_trace_loci(ngp_handle *h, const int64_t *loci, int64_t n, int64_t n_varBeta) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_tiles != nullptr, NGP_ERR_STATE, "panel not set");
    REQUIRE(n >= 0 && n <= 4096 && (n == 0 || loci) && n_varBeta >= 0, NGP_ERR_ARG, "at most 4096 traced loci");
    for (int64_t i = 0; i < n; i++) REQUIRE(loci[i] >= 0 && loci[i] < h->P, NGP_ERR_ARG, "traced locus outside the panel");
    HCHK(hipStreamSynchronize(h->stream));
    dfree(h->d_trace_loci); dfree(h->d_tr_beta); dfree(h->d_tr_vb); dfree(h->d_tr_pi);
    h->trace_ext_cap = 0; h->ntl = n; h->ntvb = n_varBeta;
    if (n == 0 && n_varBeta == 0) return NGP_OK;
    if ((rc = dalloc(h, &h->d_trace_loci, (size_t)std::max<int64_t>(n, 1)))) return rc;
    if (n) HCHK(hipMemcpy(h->d_trace_loci, loci, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice));
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_get_trace_ext(ngp_handle *h, double *beta_tr, double *varBeta_tr, double *pi_tr, int64_t n) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_trace_loci != nullptr, NGP_ERR_STATE, "no traces requested (ngp_set_trace_loci)");
    n = std::min(n, std::min(h->ntrace, h->trace_ext_cap));
    if (n <= 0) return NGP_OK;
    HCHK(hipStreamSynchronize(h->stream));
    const int64_t ntvb = std::min<int64_t>(h->ntvb, h->nvb);
    REQUIRE(ntvb == h->ntvb, NGP_ERR_STATE, "more variance traces requested than the model has variance components");
    if (beta_tr && h->ntl) HCHK(hipMemcpy(beta_tr, h->d_tr_beta, (size_t)(n * h->ntl) * sizeof(double), hipMemcpyDeviceToHost));
    if (varBeta_tr && h->ntvb) HCHK(hipMemcpy(varBeta_tr, h->d_tr_vb, (size_t)(n * h->ntvb) * sizeof(double), hipMemcpyDeviceToHost));
    if (pi_tr && !h->sets.empty()) HCHK(hipMemcpy(pi_tr, h->d_tr_pi, (size_t)n * h->sets.size() * sizeof(double), hipMemcpyDeviceToHost));
    return NGP_OK;
    NGP_CATCH(h)
}


/* Pooled posterior sums of n independent chains (one handle each): afterwards every handle holds the sums over all chains
 * (nKept included), so posterior means come from any of them.  Handles on DIFFERENT devices are reduced by ONE RCCL
 * all-reduce (ncclSum, fp64) over xGMI -- RCCL is loaded on first use (dlopen), the library has no link-time dependency on
 * it; handles that SHARE a device are added on that device first.  Single-process form of the end-of-run exchange
 * (SURVEY.md section 8 e); bench.py's multi-process form uses torch.distributed on the same packed buffer. */
namespace {
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
    bool load() {
        if (lib) return true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names) { lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
        if (!lib) { err = std::string("cannot load RCCL: ") + (dlerror() ? dlerror() : "?"); return false; }
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !AllReduce || !GroupStart || !GroupEnd || !CommDestroy) { err = "RCCL symbols missing"; dlclose(lib); lib = nullptr; return false; }
        return true;
    }
} g_rccl;

int import_posterior_device(ngp_handle *h, const double *o) {  // inverse of ngp_export_posterior_device
    const size_t pb = (size_t)h->P * sizeof(double);
    HCHK(hipMemcpyAsync(h->d_sum_beta, o, pb, hipMemcpyDeviceToDevice, h->stream));
    HCHK(hipMemcpyAsync(h->d_sum_beta2, o + h->P, pb, hipMemcpyDeviceToDevice, h->stream));
    HCHK(hipMemcpyAsync(h->d_sum_delta, o + 2 * h->P, pb, hipMemcpyDeviceToDevice, h->stream));
    if (h->nvb) HCHK(hipMemcpyAsync(h->d_sum_varBeta, o + 3 * h->P, (size_t)h->nvb * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    std::vector<double> tail(2 * h->sets.size() + (size_t)h->nclass_total + (size_t)h->nfixcol + 3);
    HCHK(hipMemcpyAsync(tail.data(), o + 3 * h->P + h->nvb, tail.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HCHK(hipStreamSynchronize(h->stream));
    for (size_t si = 0; si < h->sets.size(); si++)
        hipLaunchKernelGGL(k_set_sum_pi, dim3(1), dim3(1), 0, h->stream, h->d_sets, (int)si, tail[2 * si], tail[2 * si + 1]);
    {
        size_t off = 2 * h->sets.size();
        for (size_t si = 0; si < h->sets.size(); si++)
            if (h->sets[si].K > 0) {
                int rc2 = set_class_state_dev(h, (int)si, nullptr, tail.data() + off);
                if (rc2) return rc2;
                off += (size_t)h->sets[si].K;
            }
    }
    const size_t tf = 2 * h->sets.size() + (size_t)h->nclass_total;
    if (h->nfixcol > 0) HCHK(hipMemcpy(h->d_sum_bfix, tail.data() + tf, (size_t)h->nfixcol * sizeof(double), hipMemcpyHostToDevice));
    DScal sc;
    HCHK(hipMemcpy(&sc, h->d_scal, sizeof(DScal), hipMemcpyDeviceToHost));
    const size_t t0 = tf + (size_t)h->nfixcol;
    sc.sum_varE = tail[t0]; sc.sum_b = tail[t0 + 1]; sc.nKept = (long long)std::llround(tail[t0 + 2]);
    HCHK(hipMemcpy(h->d_scal, &sc, sizeof(DScal), hipMemcpyHostToDevice));
    HCHK(hipStreamSynchronize(h->stream));
    return NGP_OK;
}
}  // namespace

int32_t ngp_run_many(ngp_handle **hs, int32_t n, int64_t niter) {
    NGP_TRY
    if (!hs || n < 1) return fail(nullptr, NGP_ERR_ARG, "ngp_run_many: no handles");
    for (int i = 0; i < n; i++) {
        if (!hs[i]) return fail(nullptr, NGP_ERR_ARG, "ngp_run_many: null handle");
        for (int k = 0; k < i; k++)
            if (hs[k] == hs[i]) return fail(hs[0], NGP_ERR_ARG, "ngp_run_many: the same handle twice");
    }
    // chains that share one panel (ngp_share_panel) and run the engine the fused kernel serves take ONE sweep launch per iteration:
    // the panel is streamed once for all of them (K chains per pass)
    if (fusable(hs, n)) return run_fused(hs, n, niter);
    // one host thread per chain, as a caller would do it (src/samplers.jl:23: one chain per Julia task); the chains of a device run
    // side by side when their grids fit it together (ngp_set_max_shards), in turns otherwise (CuLease)
    std::vector<int32_t> rcs((size_t)n, NGP_OK);
    struct Joiner {  // a std::thread constructor that throws (system_error) must not unwind past joinable threads
        std::vector<std::thread> th;
        ~Joiner() { for (auto &t : th) if (t.joinable()) t.join(); }
    } jn;
    jn.th.reserve((size_t)n);
    // ngp_run is itself an entry point behind the exception barrier: nothing can leave a worker's lambda
    for (int i = 1; i < n; i++) jn.th.emplace_back([&rcs, hs, niter, i]() noexcept { rcs[(size_t)i] = ngp_run(hs[i], niter); });
    rcs[0] = ngp_run(hs[0], niter);
    for (auto &t : jn.th) t.join();
    for (int i = 0; i < n; i++)
        if (rcs[(size_t)i] != NGP_OK) return rcs[(size_t)i];  // the message is on that handle (ngp_last_error)
    return NGP_OK;
    NGP_CATCH((hs ? hs[0] : nullptr))
}

int32_t ngp_allreduce_posterior(ngp_handle **hs, int32_t n) {
    NGP_TRY
    if (!hs || n < 1) return fail(nullptr, NGP_ERR_ARG, "ngp_allreduce_posterior: no handles");
    for (int i = 0; i < n; i++)
        if (!hs[i]) return fail(nullptr, NGP_ERR_ARG, "ngp_allreduce_posterior: null handle");
    ngp_handle *h = hs[0];  // errors are reported on the first handle
    int rc;
    int64_t len = 0;
    if ((rc = ngp_posterior_len(h, &len))) return rc;
    for (int i = 0; i < n; i++) {
        REQUIRE(hs[i]->d_tiles != nullptr, NGP_ERR_STATE, "ngp_allreduce_posterior: a handle has no panel");
        REQUIRE(hs[i]->P == h->P && hs[i]->nvb == h->nvb && hs[i]->sets.size() == h->sets.size() && hs[i]->nfixcol == h->nfixcol &&
                    hs[i]->nclass_total == h->nclass_total,
                NGP_ERR_ARG, "ngp_allreduce_posterior: the chains do not share one model");
        for (int k = 0; k < i; k++) REQUIRE(hs[k] != hs[i], NGP_ERR_ARG, "ngp_allreduce_posterior: a handle is listed twice");
    }
    std::vector<double *> buf((size_t)n, nullptr);
    auto cleanup = [&]() { for (int i = 0; i < n; i++) if (buf[i]) { (void)hipSetDevice(hs[i]->device); (void)hipFree(buf[i]); } };
    for (int i = 0; i < n; i++) {
        if ((rc = enter(hs[i]))) { cleanup(); return rc; }
        if (hipMalloc((void **)&buf[i], (size_t)len * sizeof(double)) != hipSuccess) { cleanup(); return fail(h, NGP_ERR_NOMEM, "posterior buffer"); }
        if ((rc = ngp_export_posterior_device(hs[i], buf[i], len))) { if (hs[i] != h) h->err = hs[i]->err; cleanup(); return rc; }
    }
    // leaders: the first handle of every device; the others are added into their leader on the device
    auto devof = [](const ngp_handle *x) { return x->vdev >= 0 ? 1000 + x->vdev : x->device; };  // (virtual devices: test hook)
    std::vector<int> leader((size_t)n);
    std::vector<int> leaders;
    for (int i = 0; i < n; i++) {
        leader[i] = i;
        for (int k = 0; k < i; k++) if (devof(hs[k]) == devof(hs[i])) { leader[i] = leader[k]; break; }
        if (leader[i] == i) leaders.push_back(i);
    }
    hipError_t e = hipSuccess;
    for (int i = 0; i < n && e == hipSuccess; i++)
        if (leader[i] != i) {
            ngp_handle *L = hs[leader[i]];
            (void)hipSetDevice(L->device);
            hipLaunchKernelGGL(k_add_inplace, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, L->stream, buf[leader[i]], buf[i], (long long)len);
            e = hipStreamSynchronize(L->stream);
        }
    if (e != hipSuccess) { cleanup(); return fail(h, NGP_ERR_HIP, std::string("pooling on one device: ") + hipGetErrorString(e)); }
    if (leaders.size() > 1) {
        const int nl = (int)leaders.size();
        bool one_physical = true;
        for (int i = 1; i < nl; i++) one_physical = one_physical && hs[leaders[i]]->device == hs[leaders[0]]->device;
        if (one_physical) {
            // every leader on one GPU (virtual devices, ngp_debug_set_virtual_device): the collective is a sum on that GPU, leader
            // order -- the same packing, grouping and unpacking as across devices, everything but the ncclAllReduce call
            ngp_handle *L0 = hs[leaders[0]];
            (void)hipSetDevice(L0->device);
            double *tot = nullptr;
            if (hipMalloc((void **)&tot, (size_t)len * sizeof(double)) != hipSuccess) { cleanup(); return fail(h, NGP_ERR_NOMEM, "posterior buffer"); }
            e = hipMemcpyAsync(tot, buf[leaders[0]], (size_t)len * sizeof(double), hipMemcpyDeviceToDevice, L0->stream);
            for (int i = 1; i < nl && e == hipSuccess; i++)
                hipLaunchKernelGGL(k_add_inplace, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, L0->stream, tot, buf[leaders[i]], (long long)len);
            for (int i = 0; i < nl && e == hipSuccess; i++)
                e = hipMemcpyAsync(buf[leaders[i]], tot, (size_t)len * sizeof(double), hipMemcpyDeviceToDevice, L0->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(L0->stream);
            (void)hipFree(tot);
            if (e != hipSuccess) { cleanup(); return fail(h, NGP_ERR_HIP, std::string("pooling leaders on one device: ") + hipGetErrorString(e)); }
        } else {
        if (!g_rccl.load()) { cleanup(); return fail(h, NGP_ERR_HIP, g_rccl.err); }
        std::vector<void *> comms((size_t)nl, nullptr);
        std::vector<int> devs((size_t)nl);
        for (int i = 0; i < nl; i++) devs[i] = hs[leaders[i]]->device;
        int r = g_rccl.CommInitAll(comms.data(), nl, devs.data());
        if (r == 0) {
            g_rccl.GroupStart();
            for (int i = 0; i < nl && r == 0; i++) {
                ngp_handle *L = hs[leaders[i]];
                (void)hipSetDevice(L->device);
                r = g_rccl.AllReduce(buf[leaders[i]], buf[leaders[i]], (size_t)len, /*ncclDouble*/ 8, /*ncclSum*/ 0, comms[i], L->stream);
            }
            const int r2 = g_rccl.GroupEnd();
            if (r == 0) r = r2;
            for (int i = 0; i < nl; i++) { (void)hipSetDevice(devs[i]); (void)hipStreamSynchronize(hs[leaders[i]]->stream); }
        }
        for (int i = 0; i < nl; i++) if (comms[i]) g_rccl.CommDestroy(comms[i]);
        if (r != 0) { cleanup(); return fail(h, NGP_ERR_HIP, std::string("RCCL all-reduce failed: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?")); }
        }
    }
    for (int i = 0; i < n; i++) {
        (void)hipSetDevice(hs[i]->device);
        if ((rc = import_posterior_device(hs[i], buf[leader[i]]))) { if (hs[i] != h) h->err = hs[i]->err; cleanup(); return rc; }
    }
    cleanup();
    return NGP_OK;
    NGP_CATCH((hs ? hs[0] : nullptr))
}


/* BayesR marker set (src/runTime.jl:78-93, set-up src/mme.jl:374-383, sampler src/functions.jl:238-289): ONE variance for the set
 * (varBeta0), K <= 4 variance classes with multipliers vClass[v] of that variance (a class with multiplier 0 = effect exactly
 * 0) and class probabilities pi[v]; estPi: pi ~ Dirichlet(nLoci + 1) after every sweep.  delta holds the class of a locus,
 * counted from 1 as the reference writes it. */
int32_t ngp_add_marker_set_r(ngp_handle *h, int64_t col0, int64_t ncol, double df, double scale, double varBeta0, const double *vClass,
                             const double *pi, int32_t K, int32_t estPi, const double *lhs0, const double *rhs0, int32_t *set_id) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_tiles != nullptr, NGP_ERR_STATE, "panel not set");
    REQUIRE(!h->panel_open, NGP_ERR_STATE, "the panel is still open: x'x and the Gram window exist after ngp_end_panel");
    REQUIRE(vClass && pi && K >= 2 && K <= NGP_RMAX, NGP_ERR_ARG, "BayesR needs 2..16 variance classes with their probabilities");
    double ps = 0.0;
    for (int v = 0; v < K; v++) {
        REQUIRE(std::isfinite(vClass[v]) && vClass[v] >= 0.0 && std::isfinite(pi[v]) && pi[v] > 0.0, NGP_ERR_ARG,
                "BayesR: class multipliers must be >= 0 and class probabilities > 0");
        ps += pi[v];
    }
    REQUIRE(std::fabs(ps - 1.0) < 1e-8, NGP_ERR_ARG, "BayesR: class probabilities must sum to 1");
    REQUIRE(std::isfinite(varBeta0) && varBeta0 > 0.0, NGP_ERR_ARG, "BayesR varBeta0 must be positive");
    if (!h->d_rcls) {
        if ((rc = dalloc(h, &h->d_rcls, (size_t)4 * NGP_RMAX * (size_t)h->Ppad))) return rc;
    }
    const int64_t rs = 0, re = ncol;
    int32_t sid = -1;
    h->adding_r = true;
    rc = ngp_add_marker_set(h, col0, ncol, NGP_METHOD_BAYESR, df, scale, &rs, &re, 1, &varBeta0, 0.5, estPi, lhs0, rhs0, &sid);
    h->adding_r = false;
    if (rc) return rc;
    HSet &hs = h->sets[(size_t)sid];
    hs.K = K; hs.vcls.assign(vClass, vClass + K); hs.rpi.assign(pi, pi + K);
    DSet ds;
    HCHK(hipMemcpy(&ds, h->d_sets + sid, sizeof(DSet), hipMemcpyDeviceToHost));
    ds.K = K;
    for (int v = 0; v < NGP_RMAX; v++) { ds.vcls[v] = v < K ? vClass[v] : 0.0; ds.pic[v] = 0.0; ds.logpic[v] = 0.0; ds.sum_pic[v] = 0.0; ds.ncls[v] = 0; }
    HCHK(hipMemcpy(h->d_sets + sid, &ds, sizeof(DSet), hipMemcpyHostToDevice));
    if ((rc = set_class_state_dev(h, sid, pi, std::vector<double>((size_t)K, 0.0).data()))) return rc;
    h->nclass_total += K;
    if (set_id) *set_id = sid;
    return NGP_OK;
    NGP_CATCH(h)
}

/* Correlated marker sets -- the Tuple method of BayesPR (src/functions.jl:140-154, sampleVarCovBetaPR :513-516, set-up
 * src/mme.jl:448-489): k sets (breeds) share nloc loci; per locus a k x k conditional, per region an inverse-Wishart draw of the
 * k x k variance matrix.  The panel holds the k columns of a locus side by side: component m of locus l is panel column
 * col0 + 64 (l / Lb) + k (l % Lb) + m with Lb = floor(64 / k) loci per 64-column block and col0 on a block boundary (a locus never
 * straddles two blocks; for k = 3 column 63 of every block of the set is unused: fill it with zeros). */
int32_t ngp_add_marker_set_tuple(ngp_handle *h, int64_t col0, int64_t nloc, int32_t k, double df, const double *scale,
                                 const int64_t *reg_start, const int64_t *reg_stop, int64_t nreg, const double *varBeta0, int32_t *set_id) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_tiles != nullptr, NGP_ERR_STATE, "panel not set");
    REQUIRE(!h->panel_open, NGP_ERR_STATE, "the panel is still open: x'x and the Gram window exist after ngp_end_panel");
    REQUIRE(h->sets.size() < 16, NGP_ERR_ARG, "at most 16 marker sets");
    REQUIRE(k >= 1 && k <= NGP_KMAX, NGP_ERR_ARG, "a tuple holds 1..4 correlated sets");
    REQUIRE(nloc >= 1 && col0 >= 0 && col0 % NGP_BLK == 0, NGP_ERR_ARG, "tuple set: first column on a 64-column block boundary");
    REQUIRE(scale && varBeta0 && reg_start && reg_stop && nreg > 0, NGP_ERR_ARG, "scale / varBeta0 / regions missing");
    REQUIRE(std::isfinite(df) && df > 0, NGP_ERR_ARG, "df must be finite and positive");
    const int64_t Lb = NGP_BLK / k, nblk = (nloc + Lb - 1) / Lb, span = NGP_BLK * (nblk - 1) + (int64_t)k * (nloc - Lb * (nblk - 1));
    REQUIRE(col0 + span <= h->P, NGP_ERR_ARG, "tuple set outside the panel");
    REQUIRE((int64_t)nloc * k < ((int64_t)1 << 31), NGP_ERR_ARG, "tuple set too large");  // (every check in front of the first change to the host tables)
    // the set owns its 64-column blocks to the end of the last one (its block chain draws whole loci; no other set's column may sit there)
    for (int64_t c = col0; c < std::min<int64_t>(col0 + NGP_BLK * nblk, h->Ppad); c++) REQUIRE(h->h_setof[c] == -1, NGP_ERR_ARG, "marker sets overlap");
    for (int a = 0; a < k * k; a++) REQUIRE(std::isfinite(scale[a]) && std::isfinite(varBeta0[a]), NGP_ERR_ARG, "scale / varBeta0 must be finite");
    for (int a = 0; a < k; a++) REQUIRE(varBeta0[a * k + a] > 0.0, NGP_ERR_ARG, "varBeta0 must be positive definite");
    int64_t expect = 0;
    for (int64_t r = 0; r < nreg; r++) {
        REQUIRE(reg_start[r] == expect && reg_stop[r] > reg_start[r], NGP_ERR_ARG, "regions must be consecutive and non-empty");
        expect = reg_stop[r];
    }
    REQUIRE(expect == nloc, NGP_ERR_ARG, "regions must cover all loci of the set");
    const int si = (int)h->sets.size();
    const int64_t nv = nreg * k * k;
    std::vector<double> vb0((size_t)nv);
    for (int64_t r = 0; r < nreg; r++) for (int a = 0; a < k * k; a++) vb0[(size_t)(r * k * k + a)] = varBeta0[a];  // src/mme.jl:516
    HSet hs{col0, span, NGP_METHOD_TUPLE, df, 0.0, nreg, h->nvb, 0, 0, 0.5, vb0};
    hs.tk = k; hs.nloc = nloc;
    const int64_t new_nvb = h->nvb + nv;
    if (new_nvb > h->vb_cap) {
        int64_t cap = std::max<int64_t>(new_nvb, 2 * h->vb_cap);
        double *nvp = nullptr, *nsp = nullptr;
        if ((rc = dalloc(h, &nvp, (size_t)cap))) return rc;
        if ((rc = dalloc(h, &nsp, (size_t)cap))) return rc;
        if (h->nvb > 0) {
            HCHK(hipMemcpyAsync(nvp, h->d_varBeta, (size_t)h->nvb * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            HCHK(hipMemcpyAsync(nsp, h->d_sum_varBeta, (size_t)h->nvb * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        }
        HCHK(hipStreamSynchronize(h->stream));
        dfree(h->d_varBeta); dfree(h->d_sum_varBeta);
        h->d_varBeta = nvp; h->d_sum_varBeta = nsp; h->vb_cap = cap;
    }
    HCHK(hipMemcpy(h->d_varBeta + h->nvb, vb0.data(), (size_t)nv * sizeof(double), hipMemcpyHostToDevice));
    if (!h->d_tup) {
        if ((rc = dalloc(h, &h->d_tup, 16))) return rc;
        if ((rc = dalloc(h, &h->d_tupc, (size_t)NGP_KMAX * (size_t)h->Ppad))) return rc;
        if ((rc = dalloc(h, &h->d_tupg, (size_t)NGP_KMAX * (size_t)h->Ppad))) return rc;
    }
    for (int64_t c = col0; c < std::min<int64_t>(col0 + NGP_BLK * nblk, h->Ppad); c++) h->h_setof[c] = (int8_t)-2;  // owned, no locus (unused lanes)
    for (int64_t r = 0; r < nreg; r++) {
        for (int64_t l = reg_start[r]; l < reg_stop[r]; l++)
            for (int m = 0; m < k; m++) {
                const int64_t c = tuple_col(col0, k, l, m);
                h->h_setof[c] = (int8_t)si;
                h->h_loc[c] = (int32_t)(l * k + m);   // also the key of the component's normal draw
                h->h_vbidx[c] = (int32_t)(h->nvb + r * k * k);
            }
        DTReg tr;
        tr.seg0 = (long long)h->h_tseg_l0.size(); tr.rg = r; tr.n = reg_stop[r] - reg_start[r]; tr.set = si;
        int ns = 0;
        for (int64_t l0 = reg_start[r]; l0 < reg_stop[r]; l0 += NGP_SEG) {
            h->h_tseg_l0.push_back(l0);
            h->h_tseg_len.push_back((int32_t)std::min<int64_t>(NGP_SEG, reg_stop[r] - l0));
            h->h_tseg_set.push_back((int32_t)si);
            ns++;
        }
        tr.nseg = ns;
        h->h_tregs.push_back(tr);
    }
    DTup tp;
    memset(&tp, 0, sizeof(tp));
    tp.k = k; tp.col0 = col0; tp.nloc = nloc; tp.vb_off = h->nvb; tp.df = df;
    for (int a = 0; a < k * k; a++) tp.scale[a] = scale[a];
    HCHK(hipMemcpy(h->d_tup + si, &tp, sizeof(DTup), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_tuple_gkk, dim3((unsigned)((nloc * k + 255) / 256)), dim3(256), 0, h->stream, h->d_gramx, h->D, h->d_mpm, tp, h->d_tupg,
                       (long long)h->Ppad);
    DSet ds;
    memset(&ds, 0, sizeof(ds));
    ds.method = NGP_METHOD_TUPLE; ds.df = df; ds.col0 = col0; ds.ncol = span;
    HCHK(hipMemcpy(h->d_sets + si, &ds, sizeof(DSet), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_set_pi, dim3(1), dim3(1), 0, h->stream, h->d_sets, si, 0.5, 0.5);
    HCHK(hipStreamSynchronize(h->stream));
    h->nvb = new_nvb;
    h->sets.push_back(hs);
    h->ntuple += 1;
    h->tables_dirty = true;
    h->trace_ext_cap = 0;
    if (set_id) *set_id = si;
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_get_class_state(ngp_handle *h, int32_t set_id, double *piHat, double *sum_pi, int64_t *K) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(set_id >= 0 && set_id < (int)h->sets.size(), NGP_ERR_ARG, "unknown set id");
    HCHK(hipStreamSynchronize(h->stream));
    DSet ds;
    HCHK(hipMemcpy(&ds, h->d_sets + set_id, sizeof(DSet), hipMemcpyDeviceToHost));
    if (K) *K = h->sets[(size_t)set_id].K;
    for (int v = 0; v < h->sets[(size_t)set_id].K; v++) {
        if (piHat) piHat[v] = ds.pic[v];
        if (sum_pi) sum_pi[v] = ds.sum_pic[v];
    }
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_set_class_state(ngp_handle *h, int32_t set_id, const double *piHat, const double *sum_pi, int64_t K) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(set_id >= 0 && set_id < (int)h->sets.size(), NGP_ERR_ARG, "unknown set id");
    REQUIRE(h->sets[(size_t)set_id].K > 0 && K == h->sets[(size_t)set_id].K, NGP_ERR_ARG, "not a BayesR set, or another number of classes");
    if (piHat) for (int64_t v = 0; v < K; v++) REQUIRE(std::isfinite(piHat[v]) && piHat[v] > 0.0, NGP_ERR_ARG, "class probabilities must be > 0");
    return set_class_state_dev(h, set_id, piHat, sum_pi);
    NGP_CATCH(h)
}


/* A fixed-effect set beyond the intercept: the columns of one model term, or of one `blockThese` group (X[xSet].data, N x ncol,
 * column-major; src/prepMatVec.jl:150-165).  Sets are sampled after the intercept in the order they are added -- the order of
 * `keys(X)` at src/samplers.jl:39 (a Julia Dict: the shim passes that order; to put the intercept elsewhere, switch it off and add
 * a column of ones).  One column: sampleX! (src/functions.jl:41-47) with the summary-statistics terms lhs0 / rhs0 (src/mme.jl:140-147);
 * several: sampleb! (src/functions.jl:22-36), Gauss-Seidel over X'X + min|diag| / 10000 (src/mme.jl:149-152). */
int32_t ngp_add_fixed_set(ngp_handle *h, const double *X, int64_t N, int64_t ncol, int64_t ld, const double *lhs0, const double *rhs0,
                          int32_t *set_id) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_tiles != nullptr, NGP_ERR_STATE, "panel not set");
    REQUIRE(!h->panel_open, NGP_ERR_STATE, "the panel is still open: x'x and the Gram window exist after ngp_end_panel");
    REQUIRE(X && N == h->N && ncol >= 1 && ncol <= 64 && ld >= N, NGP_ERR_ARG, "fixed-effect set: N rows, 1..64 columns");
    REQUIRE(h->fix.size() < 16, NGP_ERR_ARG, "at most 16 fixed-effect sets");
    std::vector<double> xc((size_t)N * ncol), x0((size_t)ncol * ncol), xr;
    for (int64_t a = 0; a < ncol; a++)
        for (int64_t i = 0; i < N; i++) {
            const double v = X[(size_t)a * ld + i];
            REQUIRE(std::isfinite(v), NGP_ERR_ARG, "non-finite value in a fixed-effect column");
            xc[(size_t)a * N + i] = v;
        }
    for (int64_t a = 0; a < ncol; a++)
        for (int64_t b = 0; b <= a; b++) {
            double acc = 0.0;
            for (int64_t i = 0; i < N; i++) acc = std::fma(xc[(size_t)a * N + i], xc[(size_t)b * N + i], acc);
            x0[(size_t)a * ncol + b] = acc; x0[(size_t)b * ncol + a] = acc;
        }
    xr = x0;
    if (ncol > 1) {  // src/mme.jl:149-152
        double mn = std::fabs(x0[0]);
        for (int64_t a = 1; a < ncol; a++) mn = std::min(mn, std::fabs(x0[(size_t)a * ncol + a]));
        for (int64_t a = 0; a < ncol; a++) xr[(size_t)a * ncol + a] += mn / 10000.0;
    }
    REQUIRE(x0[0] > 0.0 || ncol > 1, NGP_ERR_ARG, "fixed-effect column is identically zero");
    HFix fx;
    fx.ncol = ncol; fx.off = h->nfixcol;
    std::vector<double> z((size_t)ncol, 0.0);
    if ((rc = dalloc(h, &fx.d_X, xc.size()))) return rc;
    if ((rc = dalloc(h, &fx.d_xpx0, x0.size()))) return rc;
    if ((rc = dalloc(h, &fx.d_xpxR, xr.size()))) return rc;
    if ((rc = dalloc(h, &fx.d_lhs0, (size_t)ncol))) return rc;
    if ((rc = dalloc(h, &fx.d_rhs0, (size_t)ncol))) return rc;
    HCHK(hipMemcpy(fx.d_X, xc.data(), xc.size() * sizeof(double), hipMemcpyHostToDevice));
    HCHK(hipMemcpy(fx.d_xpx0, x0.data(), x0.size() * sizeof(double), hipMemcpyHostToDevice));
    HCHK(hipMemcpy(fx.d_xpxR, xr.data(), xr.size() * sizeof(double), hipMemcpyHostToDevice));
    HCHK(hipMemcpy(fx.d_lhs0, lhs0 ? lhs0 : z.data(), (size_t)ncol * sizeof(double), hipMemcpyHostToDevice));
    HCHK(hipMemcpy(fx.d_rhs0, rhs0 ? rhs0 : z.data(), (size_t)ncol * sizeof(double), hipMemcpyHostToDevice));
    const int64_t nn = h->nfixcol + ncol;
    double *nb = nullptr, *nsb = nullptr;
    if ((rc = dalloc(h, &nb, (size_t)nn))) return rc;
    if ((rc = dalloc(h, &nsb, (size_t)nn))) return rc;
    if (h->nfixcol > 0) {
        HCHK(hipMemcpyAsync(nb, h->d_bfix, (size_t)h->nfixcol * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        HCHK(hipMemcpyAsync(nsb, h->d_sum_bfix, (size_t)h->nfixcol * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    }
    HCHK(hipStreamSynchronize(h->stream));
    dfree(h->d_bfix); dfree(h->d_sum_bfix);
    h->d_bfix = nb; h->d_sum_bfix = nsb; h->nfixcol = nn;
    if (set_id) *set_id = (int32_t)h->fix.size();
    h->fix.push_back(fx);
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_get_fixed(ngp_handle *h, double *b, double *sum_b, int64_t *ncols_total) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    if (ncols_total) *ncols_total = h->nfixcol;
    if (h->nfixcol == 0) return NGP_OK;
    HCHK(hipStreamSynchronize(h->stream));
    if (b) HCHK(hipMemcpy(b, h->d_bfix, (size_t)h->nfixcol * sizeof(double), hipMemcpyDeviceToHost));
    if (sum_b) HCHK(hipMemcpy(sum_b, h->d_sum_bfix, (size_t)h->nfixcol * sizeof(double), hipMemcpyDeviceToHost));
    return NGP_OK;
    NGP_CATCH(h)
}

int32_t ngp_set_fixed(ngp_handle *h, const double *b, const double *sum_b, int64_t ncols_total) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(ncols_total == h->nfixcol, NGP_ERR_ARG, "fixed-effect column count mismatch");
    if (h->nfixcol == 0) return NGP_OK;
    HCHK(hipStreamSynchronize(h->stream));
    if (b) HCHK(hipMemcpy(h->d_bfix, b, (size_t)h->nfixcol * sizeof(double), hipMemcpyHostToDevice));
    if (sum_b) HCHK(hipMemcpy(h->d_sum_bfix, sum_b, (size_t)h->nfixcol * sizeof(double), hipMemcpyHostToDevice));
    return NGP_OK;
    NGP_CATCH(h)
}

/* Kept samples to a binary file WITHOUT stopping the chain (the reference appends text rows at every kept iteration,
 * src/samplers.jl:56-104, src/outFiles.jl:17-21: 10 MB of text per sample at P = 600,000).  From the next ngp_run on, every kept
 * iteration (ngp_set_schedule) leaves one record: packed on the device into a ring of four slots, copied to pinned host memory on a
 * second stream and written by a thread of the library's own; ngp_run returns when its last record is in the file.  path == NULL
 * closes the file.  File: "NGPSMP01" | int64 P, nvb, nsets, nfix, nclass, record bytes | per set int64 {method, K, col0, ncol,
 * variance entries, tuple k} | records: int64 iteration | varE | b | b_fixed[nfix] | beta[P] | varBeta[nvb] | piHat[2 nsets] | class
 * probabilities[nclass] | delta[P] as bytes, padded to 8.  nextgp.jl_amd/api.py (samples_to_out_files) turns it into the
 * reference's *Out text files. */
int32_t ngp_set_sample_file(ngp_handle *h, const char *path) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    HCHK(hipStreamSynchronize(h->stream));
    sample_close(h);
    if (!path) return NGP_OK;
    REQUIRE(h->d_tiles != nullptr, NGP_ERR_STATE, "panel not set");
    SampleStream *S = new SampleStream();
    S->path = path; S->device = h->device;
    S->f = std::fopen(path, "wb");
    if (!S->f) { delete S; return fail(h, NGP_ERR_ARG, std::string("cannot open the sample file for writing: ") + path); }
    hipError_t e = hipStreamCreate(&S->copy_stream);
    for (int i = 0; i < SampleStream::NSLOT && e == hipSuccess; i++) {
        e = hipEventCreateWithFlags(&S->ev_packed[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&S->ev_copied[i], hipEventDisableTiming);
    }
    h->smp = S;
    if (e != hipSuccess) { sample_close(h); return fail(h, NGP_ERR_HIP, std::string("sample stream: ") + hipGetErrorString(e)); }
    S->writer = std::thread(sample_writer_loop, S);
    return NGP_OK;
    NGP_CATCH(h)
}

/* Placement census of the last persistent-sweep launch of this handle: out[b] = (XCC id + 1) << 32 | HW_REG_HW_ID of workgroup b,
 * 0 for a workgroup that never became resident (n >= grid entries; *grid = 1 sampler + reducers + streamers); *retries = launches of
 * this handle that ended at the census (grid not co-resident beside another chain's) and were run again with the device to
 * themselves; *exclusive = whether the handle's calls now lease the whole device. */
int32_t ngp_get_census(ngp_handle *h, uint64_t *out, int64_t n, int64_t *grid, int64_t *retries, int32_t *exclusive) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(h->d_tiles != nullptr && h->mode == 1, NGP_ERR_STATE, "no persistent sweep on this handle");
    const int64_t g = h->last_grid > 0 ? h->last_grid : 1 + h->NG + h->S / h->V;
    if (grid) *grid = g;
    if (retries) *retries = h->census_retries;
    if (exclusive) *exclusive = h->exclusive ? 1 : 0;
    if (out) {
        REQUIRE(n >= g, NGP_ERR_ARG, "census buffer smaller than the grid");
        HCHK(hipStreamSynchronize(h->stream));
        HCHK(hipMemcpy(out, h->d_census_tbl, (size_t)g * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    return NGP_OK;
    NGP_CATCH(h)
}

/* Test hook of the census fallback: the sweep of iteration `iteration` (1-based, as ngp_get_state counts) closes its own census as
 * "timed out" -- what a grid that is not co-resident does after 20 ms -- so that the retry path (abort before any role has run,
 * kernels behind it skipped, whole-device lease, the iteration resumed from k_prep) can be exercised on one chain.  0 = off. */
int32_t ngp_debug_fail_census(ngp_handle *h, int64_t iteration) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(iteration >= 0, NGP_ERR_ARG, "iteration must be >= 0");
    h->dbg_census_fail_iter = iteration;
    if (iteration > 0) h->exclusive = false;
    return NGP_OK;
    NGP_CATCH(h)
}

/* Test hook of ngp_allreduce_posterior's grouping: the handle is treated as living on device `vdev` (-1: its real device) when
 * leaders are chosen and buffers packed / unpacked -- with several handles of ONE GPU given different virtual devices the
 * multi-device branch runs up to the collective itself, which is then a sum on that one device instead of ncclAllReduce. */
int32_t ngp_debug_set_virtual_device(ngp_handle *h, int32_t vdev) {
    NGP_TRY
    int rc;
    if ((rc = enter(h))) return rc;
    REQUIRE(vdev >= -1 && vdev < 64, NGP_ERR_ARG, "virtual device: -1 (off) or 0..63");
    h->vdev = vdev;
    return NGP_OK;
    NGP_CATCH(h)
}

/* Test hook of the exception barrier (tests/test_abi.py): throws inside an entry point, past the same NGP_TRY / NGP_CATCH every
 * other one runs in.  kind 0: std::bad_alloc, 1: std::length_error (a std::vector asked for more than max_size), 2: a
 * non-standard exception.  h may be NULL (the message then goes where ngp_create's go).  Never returns NGP_OK. */
int32_t ngp_debug_throw(ngp_handle *h, int32_t kind) {
    NGP_TRY
    if (kind == 0) throw std::bad_alloc();
    if (kind == 1) { std::vector<double> v; v.resize(v.max_size() + 1); }
    if (kind == 2) throw 42;
    return fail(h, NGP_ERR_ARG, "ngp_debug_throw: kind must be 0, 1 or 2");
    NGP_CATCH(h)
}

}  // extern "C"
