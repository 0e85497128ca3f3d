// ngp_sweep_args.h -- launch arguments and layout constants of the persistent sweep kernel (ngp_sweep.h), shared with the host side
#pragma once
#include "ngp_common.h"

#define NGP_RING 16        // slots of every communication ring (>= lag D)
#define NGP_MAX_LAG 12     // lags above 8: compact storage only (8 VGPRs of delay line per lag and task)
#define NGP_SPIN_LIMIT (1u << 21)
#define NGP_WG 512          // threads per workgroup of the persistent kernel
#define NGP_DBG_WAVES (1u << 19)   // sampler: 8 words per block, end-of-work stamp of every wave
#define NGP_DBG_RED (3u << 18)     // reducer 0: 2 words per block (counter complete, group sum published)
#define NGP_DBG_W5 (5u << 17)      // sampler: 4 words per block (lag-1 wave: its Gram rows have arrived, its product is done; chain wave: has its total, has the diagonal block)
#define NGP_DBG_ALL (7u << 17)     // every streamer: publish time of local block 800 and its XCC id
#define NGP_SPARSE_MAX 16  // BayesB / BayesC blocks with at most this many active lanes take the sparse chain
// LDS distance of two quads of a tile: 1 KiB of data + 16 B, so that the update tasks (lanes = consecutive quads, same
// columns) read conflict-free
#ifndef NGP_LAZY_LAG
#define NGP_LAZY_LAG 7  // streamers count their partials lazily from this lag on
#endif
#define NGP_QS 1040
#define NGP_SAMPLER_TUPLE_LDS (2 * 9 * 64 * 8 + 64)  // sampler workgroup: lane coefficients of a Tuple block, two parities, + (k, used lanes)
// ... and behind them, in k_sweep_r only: lane coefficients of a BayesR block (17 x 64 doubles, two parities), (method, classes) per lane,
// flags, classes 5..8 (16 x 64 doubles, two parities)
#define NGP_SAMPLER_R_LDS (2 * 17 * 64 * 8 + 2 * 128 * 4 + 64 + 2 * 16 * 64 * 8)
#define NGP_ABORT_CENSUS 9u        // abort code: the grid was not co-resident within NGP_CENSUS_TICKS (no state was modified)
#define NGP_CENSUS_TICKS 2000000ull  // 20 ms of the 100 MHz wall clock
#define NGP_DBG_STREAM (1u << 20)  // offset of streamer 0's stamps in the debug buffer

namespace ngp {

struct SweepArgs {
    const float *tiles;
    double *ycorr;
    const double *gramx;  // [block][lag d < D][64][64]: d = 0 one-sided diagonal block (natural order), d >= 1 cross blocks with
                          // row pairs interleaved (gram_pair_index)
    const double *tinv;   // [block][64][64], written by k_tinv before the sweep: for a LINEAR block (blin[t] != 0) the transposed inverse T
                          // of its chain's unit lower-triangular system, in the layout of the diagonal Gram block (element (i, j) =
                          // T[j][i]); the sampler stages it in place of that block and takes the chain as dlt = T e0.  Null: 64 steps.
    const unsigned *blin; // [block] 1 = linear: every lane BayesPR, unowned or (fine seam) of another set than the sampled one; 1 + k: a
                          // block of a k-set Tuple (linear too).  Static for a model: written by the host (ngp_api.hip,
                          // sync_linear_blocks), read-only on the device
    int lin_all;          // every block is linear (no per-block look needed)
    int knob;     // tuning knob of the streamers (ngp_debug_set_knob), see role_streamer_rows
    int variant;  // streamer variant: 1 = phase streamer (role_streamer), 2 = row-owning waves + loader wave (role_streamer_rows)
    int V;        // shards per streamer workgroup (1; 2: role_streamer_rows_tall, the grid then has S / V streamers)
    int D, R, S, NG, near, fine_ok, t0, t1;  // near: look-ahead lags 1..near are corrected by the sampler, farther ones by the reducers
     // fine_ok: the streamers' LDS has room for the diagnostic timeline
    double *beta;
    uint8_t *delta;
    const double *c, *w, *q, *mpm, *chi;
    const int8_t *setof;
    const int32_t *vbidx;
    DSet *sets;
    double *varBeta;
    // BayesR (null / unused when the model has no BayesR set): per-class coefficients of k_prep, M.rhs, chain scalars
    const double *rcls;
    const double *rhs0;
    const DScal *scal;
    long long Ppad;
    // Tuple (correlated BayesPR) sets; null when the model has none: per-set constants, rows of C (k_prep) and of X_l'X_l per column
    const DTup *tup;
    const double *tupc, *tupg;
    // compact storage (variant 3): tiles are bytes (genotype codes), centred analytically with the Float64 column means
    const double *mean;  // [Ppad]
    long long N;         // rows of the panel (the last shards carry padding rows, which must stay zero)
    // communication (zeroed before every launch)
    unsigned long long *acc;  // [RING][NGP_FX_COPIES][64] fixed-point accumulators of X_t'ycorr (ngp_common.h): shard s adds its term to copy
                              // s mod 8, the far-lag correctors theirs to copy lag mod 8; the sampler reads, checks the count and zeroes
    double *dlt;         // [RING][64]
    unsigned long long *dltg;  // [RING][64][2] the same values as self-validating 8-byte granules {32 data bits, 32-bit tag}
    unsigned nonce;      // 12-bit launch number inside every tag (the granule ring is not cleared between launches)
    unsigned *flag_dlt;  // number of blocks the sampler has finished
    unsigned *abort_w;   // [0] != 0: the sweep gave up (code = role whose spin timed out; NGP_ABORT_CENSUS: not every workgroup became
                         // resident, NOTHING was changed); [1] = iter_tag of the launch that gave up.  Every kernel of the iteration
                         // sequence returns at once while [0] != 0, so the chain stays where the failing launch found it.
    unsigned *census;    // [0] arrivals of this launch, [1] verdict (0 open, 1 all resident, 2 timed out); zeroed by k_prep; null: no census
    unsigned long long *census_tbl;  // [grid] placement of every workgroup that arrived: (XCC id + 1) << 32 | HW_REG_HW_ID
    unsigned iter_tag;   // low 32 bits of the iteration this launch belongs to
    unsigned census_fail;  // test hook (ngp_debug_fail_census): the launch with this iter_tag closes its own census as "timed out"
    unsigned *xcc_w;     // sampler's XCC id + 1 (speed only: same-XCD streamers warm the L2 with Gram blocks)
    unsigned long long *dbg;  // optional time stamps (diagnostic runs only), else nullptr
    int dbg_mode;             // diagnostic timing runs, results invalid: 1 = streamers only move tiles, 2 = sampler alone
                              // (never waits), 3 = streamers + reducers alone (never wait for dlt), 4 = as 3 without the tile DMA,
                              // 5 = whole pipeline, BayesPR blocks without the recursion, 6 = whole pipeline, reducers never wait for dlt
};

#define NGP_ROWS_MAX_R 224
#define NGP_ROWS_NW 7       // row-owning waves
#ifndef NGP_ROWS_HMAX
#define NGP_ROWS_HMAX 32
#endif
//   // quads of tile u+2 requested during block u (the counted wait keeps them in flight)
#define NGP_ROWS_PUBW 2
#define NGP_ROWS_POLLW 6
#define NGP_DLS 72            // doubles per parity of the dlt buffer: 64 + the scalar of the compact update

// update tasks (4 rows x 8 columns) a lane of a row-owning wave carries in compact storage: 4 per unit, ceil(NU / 7) units per
// wave, 8 lane slots -> 1, 2 or 4 (the template parameter; 3 runs as 4)
__host__ __device__ inline int ngp_u8_tasks(int R) {
    const int nuw = ((R >> 4) + NGP_ROWS_NW - 1) / NGP_ROWS_NW;
    const int nt = (4 * nuw + 7) / 8;
    return nt <= 1 ? 1 : (nt == 2 ? 2 : 4);
}
#define NGP_U8_MAX_R 896  // 7 waves x 8 units x 16 rows

// ---- K chains per pass over the panel (k_sweep_multi, ngp_sweep.h) ----
#ifndef NGP_MAXC
#define NGP_MAXC 8  // chains per pass (per-chain registers of the streamer: 2 VGPRs each of shard, GEMV sum, dlt)
#endif
#ifndef NGP_PAIR_FROM
#define NGP_PAIR_FROM 4  // from this many chains on a reducer workgroup serves TWO chains (four waves each): K NG / 2 CUs go back to the streamers
#endif
// reducer workgroups of a fused launch
__host__ __device__ inline int ngp_multi_reducers(int K, int NG, int pair) { return pair ? ((K + 1) / 2) * NG : K * NG; }
struct MultiArgs {
    int K, pair;  // pair != 0: reducer workgroup (p, g) serves chains 2 p (waves 0-3) and 2 p + 1 (waves 4-7) of group g
    SweepArgs a[NGP_MAXC];  // same tiles / gramx / layout in every entry; a[c].abort_w, census of chain 0 are the launch's
};
struct ChainPtrs {  // what a streamer needs of one chain (LDS copy: lanes index it by chain)
    double *ycorr;
    unsigned long long *acc;
    const double *dlt;
    const unsigned *flag_dlt;
};
// doubles of per-chain LDS state of a multi-chain streamer: shard R | 8 x 64 chain sums | 2 x 64 dlt | 8 x R update partials
__host__ __device__ inline size_t ngp_multi_chain_doubles(int R) { return (size_t)R + 512 + 128 + (size_t)8 * R; }
__host__ __device__ inline size_t ngp_multi_lds_bytes(int R, int K) {
    return (size_t)2 * (R >> 2) * NGP_QS + (size_t)K * ngp_multi_chain_doubles(R) * 8 + 64 + 3072;
}

// the same for the row-owning streamer (fp32 tiles, shards of 64..NGP_ROWS_MAX_R rows, lag 6; 2 or 3 chains)
// (per chain: shard | 2 x 7 x 64 chain sums | 2 x 72 dlt | 2 x 8 row sums of the compact update; u8: ring slots of 16 rows x 64 bytes)
__host__ __device__ inline size_t ngp_rows_multi_chain_doubles(int R) { return (size_t)((R + 7) & ~7) + 2 * NGP_ROWS_NW * NGP_BLK + 2 * NGP_DLS + 16; }
__host__ __device__ inline size_t ngp_rows_multi_lds_bytes(int R, int K, bool u8 = false) {
    const size_t nq = (size_t)R / (u8 ? 16 : 4), hq = nq + 1 < 2 * (size_t)NGP_ROWS_HMAX ? (nq + 1) / 2 : (size_t)NGP_ROWS_HMAX;
    return (2 * nq + hq) * NGP_QS + (size_t)K * ngp_rows_multi_chain_doubles(R) * 8 + 64 + 64 + 1024;
}
hipError_t sweep_multi_set_max_lds(int bytes);
void sweep_multi_launch(unsigned grid, size_t lds_bytes, hipStream_t stream, const MultiArgs &M);
hipError_t sweep_multi_tup_set_max_lds(int bytes);  // (chains with a Tuple set: k_sweep_multi_tup, the Tuple translation unit)
void sweep_multi_tup_launch(unsigned grid, size_t lds_bytes, hipStream_t stream, const MultiArgs &M);
hipError_t sweep_multi_r_set_max_lds(int bytes);  // (chains with a BayesR set: k_sweep_multi_r, the BayesR translation unit)
void sweep_multi_r_launch(unsigned grid, size_t lds_bytes, hipStream_t stream, const MultiArgs &M);

// Host entry points of the two instantiations of the persistent kernel (ngp_sweep_inst.hip is compiled twice, once per value of
// NGP_INST_DBG, so that the production kernel and the diagnostic one build in parallel and apart from the API's own kernels):
// _0 = k_sweep<false>, _1 = k_sweep<true> (time stamps and timing modes exist only there).
hipError_t sweep_set_max_lds_0(int bytes);
hipError_t sweep_set_max_lds_1(int bytes);
hipError_t sweep_occupancy_0(int *wg_per_cu, size_t lds_bytes);
void sweep_launch_0(unsigned grid, size_t lds_bytes, hipStream_t stream, const SweepArgs &A);
// k_sweep_r (models with a BayesR set; fourth translation unit, -DNGP_INST_DBG=3)
hipError_t sweep_r_set_max_lds(int bytes);
void sweep_r_launch(unsigned grid, size_t lds_bytes, hipStream_t stream, const SweepArgs &A);
// k_sweep_tup (models with a Tuple set; third translation unit, -DNGP_INST_DBG=2)
hipError_t sweep_tup_set_max_lds(int bytes);
void sweep_tup_launch(unsigned grid, size_t lds_bytes, hipStream_t stream, const SweepArgs &A);
// k_sweep_tall (several shards per streamer workgroup; lives in the second translation unit)
hipError_t sweep_tall_set_max_lds(int bytes);
hipError_t sweep_tall_occupancy(int *wg_per_cu, size_t lds_bytes);
void sweep_tall_launch(unsigned grid, size_t lds_bytes, hipStream_t stream, const SweepArgs &A);
void sweep_launch_1(unsigned grid, size_t lds_bytes, hipStream_t stream, const SweepArgs &A);

}  // namespace ngp
