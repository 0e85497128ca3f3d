// ngp_sweep.h -- the persistent sweep kernel ("stage B"): ONE launch per Gibbs iteration walks all
// 64-SNP blocks.  Workgroups take fixed roles, one workgroup per CU:
//
//   sampler  (1)   wave 0 runs the serial recursion of every block; the other waves publish the previous block, fetch
//                  group sums and Gram blocks ahead, and compute the near look-ahead corrections G[t,a] * dlt_a
//                  (see role_sampler).
//   reducers (NG)  reducer g sums the partial X_t'y of shards 32g..32g+31 (fixed order) and folds in the far
//                  look-ahead corrections.
//   streamers (S)  streamer s owns rows [sR,(s+1)R) of ycorr (resident in LDS for the whole sweep):
//                  for every block t it applies the update of block t-D, then streams tile (t,s)
//                  through LDS and publishes its 64 partial dot products.
//
// Because the streamers run D blocks ahead of the recursion, the 5-8 us hand-off round trip
// (streamer -> reducer -> sampler -> streamer) is hidden; the updates a block has not seen yet
// are added back in 64-SNP space through the precomputed cross Gram blocks (DESIGN.md "Blocked
// sweep arithmetic", lag D).
//
// Hand-offs follow /opt/skills/guides/cdna_hip_programming.md Guideline 16, the sc1 form:
// payload stored with 8-byte agent-scope (sc1, write-through) stores by ONE wave, that wave's
// s_waitcnt vmcnt(0), then one lane's agent-scope atomic add / flag store; the consumer polls
// that word with a relaxed agent-scope load and reads the payload with sc1 loads only.
// Every spin is bounded; a timeout raises a device-wide abort word that every poll observes.
//
// Reference being replaced: the per-SNP loop of /root/reference/src/functions.jl:124-136 and
// :163-189 (three BLAS-1 passes per SNP over the panel column and its copy).
#pragma once
#include <cstddef>
#include <type_traits>
#include "ngp_sweep_args.h"

#ifndef NGP_ROWS_DPP
#define NGP_ROWS_DPP 1  // row-owning streamer: y of a quad by DPP broadcast (see fmac4_bcast)
#endif

#pragma clang fp contract(off)

namespace ngp {


// Diagnostics (time stamps, timing modes) are run-time state of the launch (SweepArgs.dbg / dbg_mode, set per handle through
// ngp_debug_* -- never from the environment): predicted-not-taken scalar branches in the production launch.  Compiling them
// out (a k_sweep<false> instantiation with both folded to constants) was built and A/B-timed on one box: the phase streamer
// became 20 % SLOWER (10k x 100k: 1.82 -> 2.24 us per block), reducers 3 %, sampler 4 % -- without the branches the compiler
// places the waits of the look-ahead loads differently.  So the template parameter only selects folding for the row-owning
// streamer, where it was measured to be neutral.
#define NGP_DBG_FOLD                                                     \
    unsigned long long *const dbg = DBG ? A.dbg : nullptr;               \
    const int dbg_mode = DBG ? A.dbg_mode : 0;                           \
    (void)dbg; (void)dbg_mode;
#define NGP_DBG_LOCALS                                                   \
    unsigned long long *const dbg = A.dbg;                               \
    const int dbg_mode = A.dbg_mode;                                     \
    (void)dbg; (void)dbg_mode;
#define NGP_DBG_ROWS NGP_DBG_FOLD

__device__ inline unsigned ld_u32(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_u32(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline double ld_f64(const double *p) {
    unsigned long long u = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __longlong_as_double((long long)u);
}
__device__ inline void st_f64(double *p, double v) {
    __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline unsigned long long ld_u64(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_u64(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// dlt of a finished block as tagged granules (cdna_hip_programming.md Guideline 16, form R2): lane j owns two naturally aligned
// 8-byte words {low half of dlt_j | tag << 32} and {high half | tag << 32}, each written by ONE 8-byte agent-scope store.  A
// reader that finds the expected tag in a word has that word's data (single-copy atomicity of an aligned 8-byte access): no
// flag, no store acknowledgement, no second load that depends on a first.  tag = launch nonce << 20 | (local block + 1).
__device__ inline unsigned dlt_tag(const unsigned nonce, const int a) { return (nonce << 20) | (unsigned)(a + 1); }
__device__ inline void publish_dlt_granules(unsigned long long *dltg, const unsigned nonce, const int a, const int j, const double v) {
    const unsigned long long t = (unsigned long long)dlt_tag(nonce, a) << 32;
    unsigned long long *g = dltg + ((size_t)(a % NGP_RING) * NGP_BLK + j) * 2;
    st_u64(g, t | (unsigned)__double2loint(v));
    st_u64(g + 1, t | (unsigned)__double2hiint(v));
}
__device__ inline bool dlt_granules_valid(const unsigned long long g0, const unsigned long long g1, const unsigned tag) {
    return __ballot(((unsigned)(g0 >> 32) == tag) && ((unsigned)(g1 >> 32) == tag)) == ~0ull;
}
__device__ inline double dlt_granules_value(const unsigned long long g0, const unsigned long long g1) {
    return __hiloint2double((int)(unsigned)g1, (int)(unsigned)g0);
}
// whole wave: spin (bounded) until the 64 x 2 granules of local block a carry its tag; false = give up.  Every lane re-reads
// its own two granules: one round trip from "published" to "in registers".  (Used by the reducers and by the row-owning
// streamers, which ask a block early and so rarely come here; the phase streamers, 228-246 of them polling from inside a
// phase, keep the one-word flag: built with granules they ran 1.90 -> 3.43 us per block at 10k x 100k.)
__device__ inline bool wait_dlt_granules_all(const unsigned long long *dltg, const unsigned nonce, const int a, const int j, unsigned *abort_w,
                                             const unsigned code, unsigned long long &g0, unsigned long long &g1) {
    const unsigned tag = dlt_tag(nonce, a);
    const unsigned long long *g = dltg + ((size_t)(a % NGP_RING) * NGP_BLK + j) * 2;
    for (unsigned spins = 0;; ++spins) {
        if (dlt_granules_valid(g0, g1, tag)) return true;
        if ((spins & 7u) == 7u && ld_u32(abort_w) != 0u) return false;
        if (spins > (NGP_SPIN_LIMIT >> 3)) {  // every turn is a memory round trip
            st_u32(abort_w, code);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
        g0 = ld_u64(g);
        g1 = ld_u64(g + 1);
    }
}
// one term of the fixed-point accumulator of local block `slot` (ngp_common.h): fire and forget -- no acknowledgement is waited for,
// no counter follows (the accumulator counts its own terms)
#define NGP_ABORT_FX 7u  // abort code: a term outside the fixed-point range (a non-finite partial sum, or ycorr grew 32-fold within a sweep)
__device__ __attribute__((always_inline)) inline void acc_add(unsigned long long *acc, unsigned *abort_w, const int slot, const int copy, const int lane,
                                                              const double p, const double fxs) {
    const double x = p * fxs;
    if (!(__builtin_fabs(x) < 0x1p53)) { st_u32(abort_w, NGP_ABORT_FX); return; }
    unsigned long long *a = acc + ((size_t)(slot * NGP_FX_COPIES + (copy & (NGP_FX_COPIES - 1))) * NGP_BLK + lane);
    (void)__hip_atomic_fetch_add(a, ((unsigned long long)fx_from_f64(x) << NGP_FX_CNT_BITS) + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// XCC (XCD) id of the executing wave: HW_REG_XCC_ID (id 20), bits 3:0
__device__ inline unsigned xcc_id() { return (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xFu; }
__device__ inline void drain_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Flags in LDS, read and written as LDS (ds_read / ds_write).  Through a generic pointer the atomics of a flag become FLAT
// instructions, and every poll then waits for ALL global loads the wave has in flight (a flat access counts on vmcnt too).
typedef __attribute__((address_space(3))) int lds_int_t;
__device__ inline int lds_flag_ld(const int *p) { return *(volatile lds_int_t *)(lds_int_t *)p; }
__device__ inline void lds_flag_st(int *p, int v) { *(volatile lds_int_t *)(lds_int_t *)p = v; }
// workgroup barrier that leaves VMEM traffic in flight (__syncthreads() drains vmcnt, which would serialise every
// LDS-DMA issued just before it); LDS traffic of the wave is complete
__device__ inline void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// LDS-DMA of 64 x 16 B (lane l: global gsrc_lane -> LDS lds_base + 16 l), issued from inline asm so that the compiler's
// wait-count pass does not see an LDS write it would have to drain before the wave's next LDS access; the caller
// owns every wait (drain_vm / counted vmcnt) before the data is read
__device__ inline void dma16_lds(const void *gsrc_lane, const void *lds_base_uniform) {
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) const char *)lds_base_uniform);
    unsigned keep_m0;  // M0 is read when the instruction issues, so it can be handed back at once
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep_m0)
                 : "s"(m), "v"(gsrc_lane)
                 : "memory");
}

// cache policy of the loader wave's tile requests (the panel is read once per sweep): "" default, " nt" non-temporal
#ifndef NGP_DMA_POLICY
#define NGP_DMA_POLICY ""
#endif
// Lean form for a wave that does nothing but request tiles: LDS address and the 64-bit global base are wave-uniform (SGPRs),
// the per-lane part is one 32-bit VGPR offset (lane * 16), so a request costs a handful of scalar instructions.  With
// per-lane 64-bit addresses and a v_readfirstlane per request (dma16_lds) ONE wave issues about 18 KiB per us, below a CU's
// share of the stream; in this form it keeps up (tools/microbench/dma_bench.hip: 2.69 -> 2.16 us per 51 KiB tile, every CU
// streaming; a second issuing wave would give 2.04).
// (default cache policy: the L2-warming requests of the Gram blocks, which are meant to stay in L2)
__device__ __attribute__((always_inline)) inline void dma16_warm(unsigned lds_addr_uniform, const void *gbase_uniform, unsigned voff) {
    unsigned keep_m0;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep_m0)
                 : "s"(lds_addr_uniform), "v"(voff), "s"(gbase_uniform)
                 : "memory");
}
__device__ __attribute__((always_inline)) inline void dma16_s(unsigned lds_addr_uniform, const void *gbase_uniform, unsigned voff) {
    unsigned keep_m0;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3" NGP_DMA_POLICY "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep_m0)
                 : "s"(lds_addr_uniform), "v"(voff), "s"(gbase_uniform)
                 : "memory");
}

// Four requests for four consecutive quads of global memory into four consecutive ring slots: ONE 64-bit scalar base, the
// instruction's immediate offset walks the source (0, 1024, 2048, 3072).  The immediate is added to the LDS address too, so M0
// carries slot address minus offset: M0 advances by NGP_QS - 1024 per request.  About six scalar instructions per request
// instead of seventeen: the loader is then bound by memory, not by its own instruction stream (51 requests took 2.2 us).
__device__ __attribute__((always_inline)) inline void dma16_s4(unsigned lds_addr_uniform, const void *gbase_uniform, unsigned voff) {
    unsigned keep_m0;
    const unsigned m1 = lds_addr_uniform + (NGP_QS - 1024), m2 = lds_addr_uniform + 2 * (NGP_QS - 1024), m3 = lds_addr_uniform + 3 * (NGP_QS - 1024);
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %6" NGP_DMA_POLICY "\n\t"
        "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %6 offset:1024" NGP_DMA_POLICY "\n\t"
        "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %6 offset:2048" NGP_DMA_POLICY "\n\t"
        "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %6 offset:3072" NGP_DMA_POLICY "\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep_m0)
        : "s"(lds_addr_uniform), "s"(m1), "s"(m2), "s"(m3), "v"(voff), "s"(gbase_uniform)
        : "memory");
}

// ONE lane polls ONE word; bounded; false = give up (abort word set by us or by somebody else)
__device__ inline bool wait_ge(const unsigned *flag, unsigned target, unsigned *abort_w, unsigned code) {
    for (unsigned spins = 0;; ++spins) {
        if (ld_u32(flag) >= target) return true;
        if ((spins & 31u) == 31u && ld_u32(abort_w) != 0u) return false;
        if (spins > NGP_SPIN_LIMIT) {
            st_u32(abort_w, code);
            return false;
        }
        __builtin_amdgcn_s_sleep(4);
    }
}

// Cross Gram blocks (lag d >= 1) are stored with row pairs interleaved -- element (k, j) at ((k >> 1) * 64 + j) * 2 + (k & 1)
// -- so that a lane fetches its column of two rows with one 16-byte load: 32 loads per block instead of 64.  A wave can keep
// only 63 loads in flight (vmcnt has 6 bits): the 64th load of a block used to stall its wave for a full memory round trip,
// which made the lag-2 / lag-3 waves the last ones at the sampler's barrier.
__device__ inline void load_rows_pair(const double *blk, int j, double (&out)[NGP_BLK]) {
    const double2 *p = (const double2 *)blk + j;
#pragma unroll
    for (int k2 = 0; k2 < NGP_BLK / 2; k2++) {
        const double2 v = p[k2 * NGP_BLK];
        out[2 * k2] = v.x;
        out[2 * k2 + 1] = v.y;
    }
}

// scalar (SMEM) load that bypasses the scalar cache: lets a wave with LDS-DMA in flight look at a global word
// without touching vmcnt
__device__ inline unsigned sld_u32(const unsigned *p) {
    // (the address is wave-uniform by contract; say so, or a pointer read through a run-time index lands in vector registers)
    const unsigned long long a = (unsigned long long)p;
    p = (const unsigned *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a));
    unsigned v;
    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// two words in one scalar round trip: p0 past the scalar cache (a word others write), p1 through it (read-only during the launch)
__device__ inline void sld_u32x2(const unsigned *p0, const unsigned *p1, unsigned &v0, unsigned &v1) {
    const unsigned long long a0 = (unsigned long long)p0, a1 = (unsigned long long)p1;
    p0 = (const unsigned *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a0 >> 32)) << 32) |
                            (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a0));
    p1 = (const unsigned *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a1 >> 32)) << 32) |
                            (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a1));
    asm volatile("s_load_dword %0, %2, 0x0 glc\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v0), "=&s"(v1) : "s"(p0), "s"(p1) : "memory");
}
// the 32 KiB the sampler stages for the chain of block t: T of a linear block, else the one-sided diagonal Gram block
__device__ inline const double *chain_block_src(const SweepArgs &A, const long long t, const bool lin) {
    return lin ? A.tinv + (size_t)t * (NGP_BLK * NGP_BLK) : A.gramx + ((size_t)t * A.D + 0) * (NGP_BLK * NGP_BLK);
}
// L2 warming (speed only) of what the sampler CU reads for block t, as 1 KiB pieces at byte offset `off` of
//   [ T (models with linear blocks) | diagonal Gram block (unless every block is linear) | cross Gram planes 1, 2, .. ]
// (T not warmed: the sampler's 32 KiB per block then come from HBM and every other load of that CU queues behind them -- 10k x 100k
// 1.80 -> 2.05 us per block.  No look at the block's flag here: a streamer's loader would pay a memory round trip for it.)
__device__ __attribute__((always_inline)) inline size_t warm_bytes(const SweepArgs &A, const int planes) {
    return (size_t)(planes + ((A.tinv && !A.lin_all) ? 1 : 0)) * (NGP_BLK * NGP_BLK * sizeof(double));
}
// (per block: tb = T of the block, gb = its Gram planes; a piece at virtual offset off comes from tb + off below t_hi, else from
// gb + off - shift; t_hi and shift are fixed for the sweep -- one scalar compare per piece, nothing else in the loaders' issue loops)
__device__ __attribute__((always_inline)) inline size_t warm_t_hi(const SweepArgs &A) { return A.tinv ? (size_t)(NGP_BLK * NGP_BLK * sizeof(double)) : 0; }
__device__ __attribute__((always_inline)) inline size_t warm_shift(const SweepArgs &A) {
    return (A.tinv && !A.lin_all) ? (size_t)(NGP_BLK * NGP_BLK * sizeof(double)) : 0;
}

// gemv4: v_j = sum_k G[k][j] * d[k], four interleaved partial sums, ((s0+s1)+(s2+s3))
template <typename GLoad>
__device__ inline double gemv4(GLoad G, const double *d) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 16
    for (int k = 0; k < NGP_BLK; k += 4) {
        s0 = __builtin_fma(G(k + 0), d[k + 0], s0);
        s1 = __builtin_fma(G(k + 1), d[k + 1], s1);
        s2 = __builtin_fma(G(k + 2), d[k + 2], s2);
        s3 = __builtin_fma(G(k + 3), d[k + 3], s3);
    }
    return (s0 + s1) + (s2 + s3);
}

// ------------------------------------------------------------------------------------------
// 512-thread workgroups (8 waves).  Streamer s keeps its ycorr shard in LDS for the whole sweep.
// Every tile is read from HBM exactly ONCE: LDS-DMA (global_load_lds_dwordx4, tiles are contiguous)
// brings tile u+1 into one of two LDS slots while tile u is used for the GEMV; in the same phase each
// thread copies the elements of tile u that ITS update tasks will need into registers, where they
// wait D blocks (a D-deep register delay line, statically indexed through the unrolled inner loop)
// until dlt of block u arrives.  LDS is the staging / transposition buffer, registers are the delay.
//   waves 4-6  LDS-DMA of the next tile        wave 7  waits for dlt of block u-D and stages it in LDS
//   waves 0-3  GEMV (lane = column, strided row quads); wave 0 publishes the 64 partial sums
//   all waves  update tasks (c, i): 8-column chain c of row i
// TPT = update tasks per thread: 8*R <= TPT * NGP_WG (R <= 64 / 128 / 256 for TPT = 1 / 2 / 4)
template <bool DBG, int DT, int NGP_TPT>
__device__ inline void role_streamer(const SweepArgs &A, const int s, char *smem) {
    NGP_DBG_LOCALS
    const int R = A.R, S = A.S, tid = threadIdx.x;
    const int wv = tid >> 6, j = tid & 63;
    const size_t TB = (size_t)R * 256;               // tile bytes in HBM: R/4 quads of 1 KiB (quad-major, ngp_kernels.h)
    const size_t TBL = (size_t)(R >> 2) * NGP_QS;    // the same tile in LDS: quads NGP_QS bytes apart
    char *ring = smem;                               // 2 slots of TBL bytes
    // Short shards (R <= 64, one update task per thread): a shard fits into the lanes of one wave, so every wave forms the
    // updated shard itself at the start of phase C (lane i: tree of the 8 chain partials of row i, y_i - T) and reads its
    // quads' rows with v_readlane -- the separate pass over the shard and its barrier are gone.  The shard is double
    // buffered in LDS by iteration parity (wave 0 writes the new one while the others may still read the old one).
    constexpr bool FUSE1 = (NGP_TPT == 1);
    double *ys = (double *)(smem + 2 * TBL);  // FUSE1: 2 x R (parity), else R
    double *red = ys + 2 * R;                 // 8 x 64 chain partials
    double *dl = red + 512;                   // 2 x 64: dlt of the block being applied, double-buffered by iteration parity
    int *sflag = (int *)(dl + 128);
    char *scratch = (char *)(dl + 128) + 64;  // 3 KiB sink of the L2-warming DMA
    double *pp = (double *)(scratch + 3072);  // 8 x R partial sums of the update
    // diagnostic runs, short shards only (the workgroup's LDS is sized by the sampler then): barrier-arrival stamps of
    // every wave of streamer 1 for local blocks 800..815, staged in LDS and dumped at the end
    unsigned long long *fine = (unsigned long long *)(pp + 8 * (size_t)R);
    const bool fine_on = dbg && s == 1 && A.fine_ok == 1;
#define NGP_FINE(k)                                                                                          \
    do {                                                                                                     \
        if (fine_on && (unsigned)(u - 800) < 16u && j == 0) fine[(((u - 800) * 8 + wv) << 3) + (k)] = wall_clock64(); \
    } while (0)
    const size_t tile_elems = (size_t)R * NGP_BLK;
    const int nchunk = R >> 2;  // 1 KiB pieces per tile
    const int ntask = (8 * R + NGP_WG - 1) / NGP_WG;
    double *yg = A.ycorr + (size_t)s * R;
    const int g = s / NGP_GRP;
    const int nb = A.t1 - A.t0;
    auto dma_tile = [&](int ub) {  // waves 4..6 copy tile ub into slot ub&1, 1 KiB per wave-instruction
        if (dbg_mode == 4) return;  // timing experiment: the compute phases without the stream
        const char *src = (const char *)(A.tiles + ((size_t)(A.t0 + ub) * S + s) * tile_elems);
        char *dst = ring + (size_t)(ub & 1) * TBL;
        for (int c = wv - 4; c < nchunk; c += 3) dma16_lds(src + (size_t)c * 1024 + (size_t)j * 16, dst + (size_t)c * NGP_QS);
    };
    // wave 7: wait for dlt of local block (uu - DT) and stage it in dl[uu & 1]
    auto poll_dlt = [&](int uu) {
        const int aa = uu - DT;
        if (aa < 0 || uu >= nb + DT) return;
        int ok = 1;
        if (j == 0) {
            ok = (dbg_mode == 3 || dbg_mode == 4 || wait_ge(A.flag_dlt, (unsigned)(aa + 1), A.abort_w, 1u)) ? 1 : 0;
            if (dbg && s == 0) dbg[NGP_DBG_STREAM + 2 * (size_t)uu + 1] = wall_clock64();
            if (!ok) *sflag = 0;
        }
        ok = __builtin_amdgcn_readfirstlane(ok);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (ok) dl[(uu & 1) * 64 + j] = ld_f64(&A.dlt[(size_t)(aa % NGP_RING) * NGP_BLK + j]);
    };
    // the update task of this thread (fixed for the whole sweep): 8-column chain tcc of the NGP_TPT consecutive rows
    // [ti0, ti0 + NGP_TPT) -- one 4 / 8 / 16-byte LDS read per column instead of NGP_TPT 4-byte reads (tall shards are bound
    // by LDS instruction issue, not by arithmetic); surplus threads redo the last task
    const int gpc = R / NGP_TPT;  // row groups per chain (R is a multiple of 4)
    const int tq_ = min(tid, 8 * gpc - 1);
    const int tcc = tq_ / gpc, ti0 = (tq_ - tcc * gpc) * NGP_TPT;
    float keep[DT][NGP_TPT][8];
#pragma unroll
    for (int d = 0; d < DT; d++)
#pragma unroll
        for (int tp = 0; tp < NGP_TPT; tp++)
#pragma unroll
            for (int jj = 0; jj < 8; jj++) keep[d][tp][jj] = 0.0f;
    for (int i = tid; i < R; i += NGP_WG) ys[i] = yg[i];
    if (tid == 0) *sflag = 1;
    if (wv >= 4 && wv <= 6 && nb > 0) dma_tile(0);
    // speed only, but worth 25 % at 10k x 100k (3.35 -> 4.43 ms/iteration without it): streamers that share the sampler's
    // XCD pull the Gram blocks of the block after their own into that XCD's L2, so the sampler CU (one CU, latency-bound)
    // finds them there
    const unsigned my_xcc = xcc_id() + 1u;
    const int nslice = max(1, S / 8);
    const int slice = (s / 8) % nslice;
    const size_t gram_bytes = warm_bytes(A, DT), w_thi = warm_t_hi(A), w_shift = warm_shift(A);
    const size_t slice_bytes = ((gram_bytes / nslice + 1023) / 1024) * 1024;
    bool same_xcd = false;
    // (the partial sums leave as fire-and-forget atomic adds since round 4: the lazy counting of round 1-3 -- store, acknowledgement
    // watched through VM_CNT, counter -- is gone with the counters)
    const double fxs = A.scal->fx_scale;
    auto try_signal = [&](bool) {};
    unsigned long long accA = 0, accB = 0, accC = 0, accP = 0, tt0 = 0;
    __syncthreads();
    for (int u0 = 0; u0 < nb + DT; u0 += DT) {
#pragma unroll
        for (int d = 0; d < DT; d++) {
            const int u = u0 + d;
            if (u >= nb + DT) break;
            const int a = u - DT;  // block whose update is applied in this iteration (if >= 0); its tile sits in keep[d]
            if (dbg) tt0 = wall_clock64();
            NGP_FINE(0);
            // ---------------- phase A: everything that waits on memory ----------------
            if (wv >= 4 && wv <= 6) {
                drain_vm();  // tile u (issued one iteration ago) has landed
                NGP_FINE(7);
                if (u + 1 < nb) dma_tile(u + 1);
                if ((u & 7) == 0 && !same_xcd) same_xcd = (ld_u32(A.xcc_w) == my_xcc);
                if (same_xcd && u + 1 < nb) {  // fire-and-forget: the lines only have to reach this XCD's L2
                    const char *gb = (const char *)(A.gramx + (size_t)(A.t0 + u + 1) * DT * NGP_BLK * NGP_BLK) - w_shift;
                    const char *tb = (const char *)(A.tinv + (size_t)(A.t0 + u + 1) * NGP_BLK * NGP_BLK);
                    const size_t lo = (size_t)slice * slice_bytes, hi = min((size_t)(slice + 1) * slice_bytes, gram_bytes);
                    for (size_t off = lo + (size_t)(wv - 4) * 1024; off + 1024 <= hi; off += 3 * 1024)
                        dma16_lds((off < w_thi ? tb : gb) + off + (size_t)j * 16, scratch + (wv - 4) * 1024);
                }
            } else if (wv == 7 && dbg_mode != 1) {
                if (DT <= 2 || u == 0) poll_dlt(u);  // lags 1-2 cannot poll ahead: the flag would (transitively, through the
                                                     // sampler's own look-ahead fetch of the next group sums) need this block's partial
            }
            if (dbg && (tid == 448 || tid == 256)) accP += wall_clock64() - tt0;  // wave 7 poll / wave 4 DMA drain
            NGP_FINE(1);
            try_signal(false);
            wg_barrier();
            if (!*sflag) return;
            if (dbg && tid == 0) { unsigned long long n = wall_clock64(); accA += n - tt0; tt0 = n; }
            // ---------------- phase B: ycorr -= X_a dlt_a (tile a waits in keep[d]) ----------------
            if (a >= 0 && dbg_mode != 1) {
                {
                    const double *dq = dl + (u & 1) * 64 + 8 * tcc;
                    double dqv[8];
#pragma unroll
                    for (int jj = 0; jj < 8; jj++) dqv[jj] = dq[jj];
                    double pv[NGP_TPT];
#pragma unroll
                    for (int tp = 0; tp < NGP_TPT; tp++) {
                        double p = 0.0;
#pragma unroll
                        for (int jj = 0; jj < 8; jj++) p = __builtin_fma((double)keep[d][tp][jj], dqv[jj], p);
                        pv[tp] = p;
                    }
                    double *ppw = pp + (size_t)tcc * R + ti0;
#pragma unroll
                    for (int tp = 0; tp < NGP_TPT; tp++) ppw[tp] = pv[tp];
                }
                NGP_FINE(2);
                try_signal(false);
                wg_barrier();
                try_signal(false);
                if (!FUSE1) {
                    for (int i = tid; i < R; i += NGP_WG) {
                        const double T = ((pp[i] + pp[R + i]) + (pp[2 * R + i] + pp[3 * R + i])) +
                                         ((pp[4 * R + i] + pp[5 * R + i]) + (pp[6 * R + i] + pp[7 * R + i]));
                        ys[i] = ys[i] - T;
                    }
                }
            } else if (!FUSE1) {
                wg_barrier();
            }
            NGP_FINE(3);
            if (!FUSE1) wg_barrier();
            // FUSE1: lane i < R of EVERY wave holds the updated y_i of this iteration; wave 0 stores it for the next one
            double yn = 0.0;
            if (FUSE1) {
                const double *yc = ys + (size_t)(u & 1) * R;
                if (j < R) {
                    yn = yc[j];
                    if (a >= 0 && dbg_mode != 1) {
                        const double T = ((pp[j] + pp[R + j]) + (pp[2 * R + j] + pp[3 * R + j])) +
                                         ((pp[4 * R + j] + pp[5 * R + j]) + (pp[6 * R + j] + pp[7 * R + j]));
                        yn = yn - T;
                    }
                    if (wv == 0) ys[(size_t)((u & 1) ^ 1) * R + j] = yn;
                }
            }
            if (dbg && tid == 0) { unsigned long long n = wall_clock64(); accB += n - tt0; tt0 = n; }
            // ---------------- phase C: partial X_u' ycorr, and tile u into the delay line ----------------
            if (u < nb) {
                // wave 7, lag >= 3: dlt of the NEXT iteration in two asynchronous steps -- the flag is read while the wave
                // works on its chain, the 64 values travel while the workgroup crosses the barrier
                const int pa = u + 1 - DT;
                const bool pollw = (wv == 7) && (DT >= 3) && (dbg_mode != 1) && (pa >= 0) && (u + 1 < nb + DT);
                unsigned fl = 0;
                if (pollw) fl = ld_u32(A.flag_dlt);
                const float *slotp = (const float *)(ring + (size_t)(u & 1) * TBL);
                {
                    // rows ti0.. of columns 8 tcc + jj: quad ti0 >> 2, NGP_QS / 4 floats per quad, 4 floats per column
                    const float *tq = slotp + (size_t)(ti0 >> 2) * (NGP_QS / 4) + 32 * tcc + (ti0 & 3);
#pragma unroll
                    for (int jj = 0; jj < 8; jj++) {
                        if (NGP_TPT == 4) {
                            const float4 v = *(const float4 *)(tq + 4 * jj);
                            keep[d][0][jj] = v.x; keep[d][1 % NGP_TPT][jj] = v.y; keep[d][2 % NGP_TPT][jj] = v.z; keep[d][3 % NGP_TPT][jj] = v.w;
                        } else if (NGP_TPT == 2) {
                            const float2 v = *(const float2 *)(tq + 4 * jj);
                            keep[d][0][jj] = v.x; keep[d][1 % NGP_TPT][jj] = v.y;
                        } else {
                            keep[d][0][jj] = tq[4 * jj];
                        }
                    }
                }
                {   // chain wv: row quads wv, wv+8, ... (lane = column)
                    const float *col = slotp + 4 * j;  // quad qd of column j: NGP_QS qd + 16 j bytes -- consecutive lanes, consecutive 16 B
                    double acc = 0.0;
                    for (int qd = __builtin_amdgcn_readfirstlane(wv); qd < (R >> 2); qd += 8) {  // (uniform: it indexes v_readlane)
                        float4 x = *(const float4 *)(col + (size_t)qd * (NGP_QS / 4));
                        double y0, y1, y2, y3;
                        if (FUSE1) {
                            y0 = readlane_d(yn, 4 * qd); y1 = readlane_d(yn, 4 * qd + 1);
                            y2 = readlane_d(yn, 4 * qd + 2); y3 = readlane_d(yn, 4 * qd + 3);
                        } else {
                            const double *yq = ys + 4 * qd;
                            y0 = yq[0]; y1 = yq[1]; y2 = yq[2]; y3 = yq[3];
                        }
                        acc = __builtin_fma((double)x.x, y0, acc);
                        acc = __builtin_fma((double)x.y, y1, acc);
                        acc = __builtin_fma((double)x.z, y2, acc);
                        acc = __builtin_fma((double)x.w, y3, acc);
                    }
                    red[wv * 64 + j] = acc;
                }
                NGP_FINE(4);
                double dnext = 0.0;
                bool have_dnext = false;
                if (pollw) {
                    int ok = 1;
                    if (__builtin_amdgcn_readfirstlane((int)fl) < pa + 1 && dbg_mode != 3 && dbg_mode != 4) {
                        if (j == 0) {
                            ok = wait_ge(A.flag_dlt, (unsigned)(pa + 1), A.abort_w, 1u) ? 1 : 0;
                            if (!ok) *sflag = 0;
                        }
                        ok = __builtin_amdgcn_readfirstlane(ok);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (ok) {
                        dnext = ld_f64(&A.dlt[(size_t)(pa % NGP_RING) * NGP_BLK + j]);
                        have_dnext = true;
                    }
                }
                NGP_FINE(5);
                try_signal(false);
                wg_barrier();
                if (have_dnext) dl[((u + 1) & 1) * 64 + j] = dnext;  // read in phase B of the next iteration, two barriers away
                if (wv == 1 && dbg_mode != 1) {
                    const int slot = u % NGP_RING;
                    double p = ((red[j] + red[64 + j]) + (red[128 + j] + red[192 + j])) + ((red[256 + j] + red[320 + j]) + (red[384 + j] + red[448 + j]));
                    acc_add(A.acc, A.abort_w, slot, s, j, p, fxs);
                    if (j == 0) {
                        if (dbg && s == 0) dbg[NGP_DBG_STREAM + 2 * (size_t)u] = wall_clock64();
                        if (dbg && (u == 800 || u == 1200)) {
                            dbg[NGP_DBG_ALL + 4 * (size_t)s + (u == 800 ? 0 : 2)] = wall_clock64();
                            dbg[NGP_DBG_ALL + 4 * (size_t)s + 1] = my_xcc;
                            dbg[NGP_DBG_ALL + 4 * (size_t)s + 3] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
                        }
                    }
                }
            }
            else if (wv == 7 && DT >= 3 && dbg_mode != 1) poll_dlt(u + 1);
            if (dbg && tid == 0) accC += wall_clock64() - tt0;
            NGP_FINE(6);
        }
    }
    try_signal(true);
    if (dbg && tid == 0) { dbg[NGP_DBG_ALL + 4096 + 8 * (size_t)s] = accA; dbg[NGP_DBG_ALL + 4096 + 8 * (size_t)s + 1] = accB; dbg[NGP_DBG_ALL + 4096 + 8 * (size_t)s + 2] = accC; }
    if (dbg && tid == 448) dbg[NGP_DBG_ALL + 4096 + 8 * (size_t)s + 3] = accP;
    if (dbg && tid == 256) dbg[NGP_DBG_ALL + 4096 + 8 * (size_t)s + 4] = accP;
    if (fine_on && nb > 816) {
        __syncthreads();
        for (int i = tid; i < 1024; i += NGP_WG) dbg[NGP_DBG_ALL + 8192 + i] = fine[i];
    }
#undef NGP_FINE
    __syncthreads();
    {   // FUSE1: the last iteration (index nb + DT - 1) wrote the buffer of parity nb + DT
        const double *yfin = FUSE1 ? ys + (size_t)((nb + DT) & 1) * R : ys;
        for (int i = tid; i < R; i += NGP_WG) yg[i] = yfin[i];
    }
}


// Wave-uniform operands without the LDS broadcast.  Every lane of a wave multiplies its own tile element by the SAME dlt_j (update)
// or y_i (GEMV); read as broadcast LDS reads that is 16 bytes per lane and operand pair -- 256 KB of LDS reads per block with eight
// chains, 2,000 of a block's 8,000 clocks on the LDS pipe alone, each.  Instead the 8 (4) values are read ONCE per wave, lane l
// holding value l mod 8 (l mod 4), and v_fmac_f64 takes its first factor through the DPP modifier row_newbcast:N -- lane N of the
// lane's row of 16, the DGEMM broadcast of the VALU -- fused exactly like __builtin_fma (tools/microbench/dpp_bcast.hip).
// One asm statement carries a whole group of chains, step by step, so that the chains' dependent fmac interleave; the leading
// s_nop covers the two wait states a DPP read needs after a VALU write of its source (the compiler does not see DPP in asm).
#define NGP_FB(P, D, X, N) "v_fmac_f64_dpp " P ", " D ", " X " row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t"
template <int GN>
__device__ __attribute__((always_inline)) inline void fmac8_bcast(double (&p)[GN], const double (&dv)[GN], const double (&x)[8]) {
    static_assert(GN >= 1 && GN <= 4, "group of 1..4 chains");
#define NGP_X8 "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7])
    if constexpr (GN == 1) {
#define NGP_ST(N, XO) NGP_FB("%0", "%1", XO, N)
        asm("s_nop 1\n\t" NGP_ST(0, "%2") NGP_ST(1, "%3") NGP_ST(2, "%4") NGP_ST(3, "%5") NGP_ST(4, "%6") NGP_ST(5, "%7") NGP_ST(6, "%8") NGP_ST(7, "%9")
            : "+v"(p[0]) : "v"(dv[0]), NGP_X8);
#undef NGP_ST
    } else if constexpr (GN == 2) {
#define NGP_ST(N, XO) NGP_FB("%0", "%2", XO, N) NGP_FB("%1", "%3", XO, N)
        asm("s_nop 1\n\t" NGP_ST(0, "%4") NGP_ST(1, "%5") NGP_ST(2, "%6") NGP_ST(3, "%7") NGP_ST(4, "%8") NGP_ST(5, "%9") NGP_ST(6, "%10") NGP_ST(7, "%11")
            : "+v"(p[0]), "+v"(p[1]) : "v"(dv[0]), "v"(dv[1]), NGP_X8);
#undef NGP_ST
    } else if constexpr (GN == 3) {
#define NGP_ST(N, XO) NGP_FB("%0", "%3", XO, N) NGP_FB("%1", "%4", XO, N) NGP_FB("%2", "%5", XO, N)
        asm("s_nop 1\n\t" NGP_ST(0, "%6") NGP_ST(1, "%7") NGP_ST(2, "%8") NGP_ST(3, "%9") NGP_ST(4, "%10") NGP_ST(5, "%11") NGP_ST(6, "%12") NGP_ST(7, "%13")
            : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]) : "v"(dv[0]), "v"(dv[1]), "v"(dv[2]), NGP_X8);
#undef NGP_ST
    } else {
#define NGP_ST(N, XO) NGP_FB("%0", "%4", XO, N) NGP_FB("%1", "%5", XO, N) NGP_FB("%2", "%6", XO, N) NGP_FB("%3", "%7", XO, N)
        asm("s_nop 1\n\t" NGP_ST(0, "%8") NGP_ST(1, "%9") NGP_ST(2, "%10") NGP_ST(3, "%11") NGP_ST(4, "%12") NGP_ST(5, "%13") NGP_ST(6, "%14") NGP_ST(7, "%15")
            : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(dv[0]), "v"(dv[1]), "v"(dv[2]), "v"(dv[3]), NGP_X8);
#undef NGP_ST
    }
#undef NGP_X8
}
// the same for a quad of rows: acc_c += x_e * y_c[e], e = 0..3 in order (yv: lane l holds y_c[l mod 4])
template <int GN>
__device__ __attribute__((always_inline)) inline void fmac4_bcast(double (&a)[GN], const double (&yv)[GN], const double (&x)[4]) {
    static_assert(GN >= 1 && GN <= 4, "group of 1..4 chains");
#define NGP_X4 "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3])
    if constexpr (GN == 1) {
#define NGP_ST(N, XO) NGP_FB("%0", "%1", XO, N)
        asm("s_nop 1\n\t" NGP_ST(0, "%2") NGP_ST(1, "%3") NGP_ST(2, "%4") NGP_ST(3, "%5") : "+v"(a[0]) : "v"(yv[0]), NGP_X4);
#undef NGP_ST
    } else if constexpr (GN == 2) {
#define NGP_ST(N, XO) NGP_FB("%0", "%2", XO, N) NGP_FB("%1", "%3", XO, N)
        asm("s_nop 1\n\t" NGP_ST(0, "%4") NGP_ST(1, "%5") NGP_ST(2, "%6") NGP_ST(3, "%7") : "+v"(a[0]), "+v"(a[1]) : "v"(yv[0]), "v"(yv[1]), NGP_X4);
#undef NGP_ST
    } else if constexpr (GN == 3) {
#define NGP_ST(N, XO) NGP_FB("%0", "%3", XO, N) NGP_FB("%1", "%4", XO, N) NGP_FB("%2", "%5", XO, N)
        asm("s_nop 1\n\t" NGP_ST(0, "%6") NGP_ST(1, "%7") NGP_ST(2, "%8") NGP_ST(3, "%9")
            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]) : "v"(yv[0]), "v"(yv[1]), "v"(yv[2]), NGP_X4);
#undef NGP_ST
    } else {
#define NGP_ST(N, XO) NGP_FB("%0", "%4", XO, N) NGP_FB("%1", "%5", XO, N) NGP_FB("%2", "%6", XO, N) NGP_FB("%3", "%7", XO, N)
        asm("s_nop 1\n\t" NGP_ST(0, "%8") NGP_ST(1, "%9") NGP_ST(2, "%10") NGP_ST(3, "%11")
            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(yv[0]), "v"(yv[1]), "v"(yv[2]), "v"(yv[3]), NGP_X4);
#undef NGP_ST
    }
#undef NGP_X4
}


// ------------------------------------------------------------------------------------------
// Streamer, variant 2 ("row-owning waves"), shards of up to NGP_ROWS_MAX_R rows, lags 3..6.
//
// What bounded the phase streamer above on tall shards (50k x 600k: 2.96 us per block against 2.20 us for the bare
// stream) was not arithmetic: four workgroup barriers per block, with the three DMA waves sitting 1.3 us in the issue of
// the next tile (the CU's memory pipeline admits new requests at the rate data returns) and the publisher wave 1.5 us in
// the acknowledgement of its store, while the other waves waited at the barrier behind them.  Here
//
//   wave 7     is the loader and nothing else: it keeps the LDS-DMA queue of the CU full (tile u+1 and the first H quads
//              of tile u+2 are requested during block u, into a ring of 2 NQ + H quad slots) and joins the ONE barrier of
//              the block after a counted vmcnt wait that leaves those H requests in flight;
//   waves 0-6  own rows: wave w holds quads w, w+7, w+14, ... of the shard (a quad = 4 rows).  It forms the GEMV chain of
//              exactly those quads (lane = column) and applies the update to exactly those rows (lane = (quad, 8-column
//              chain), the 8 chain partials of a row meet in a DPP tree inside the wave), so ycorr never crosses a wave:
//              no barrier between the update and the GEMV, no partial-sum array, one LDS read of the shard per quad.
//   wave 2     also publishes the 64 partial dot products of the previous block right after the barrier and counts them
//              when its own VM_CNT shows the store acknowledged (it has no other memory traffic);
//   wave 6     also fetches dlt of the block to be applied next (flag requested at the start of the block, read after its
//              arithmetic).
//
// Summation order (DESIGN.md section 2, step 3'): 7 GEMV chains instead of 8 -- chain w runs over quads w, w+7, ... (4
// sequential fma per quad), p = ((a0+a1)+(a2+a3))+((a4+a5)+a6); the update is unchanged (8 chains of 8 columns, pairwise
// tree).  The blocked oracle takes the chain count as a layout parameter (ngp_get_streamer).

__device__ inline double dpp_f64(double v, const int ctrl_sel) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    if (ctrl_sel == 0) {  // quad_perm [1,0,3,2]: lane ^ 1
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false);
    } else if (ctrl_sel == 1) {  // quad_perm [2,3,0,1]: lane ^ 2
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, false);
    } else if (ctrl_sel == 2) {  // row_half_mirror: lane i of each group of 8 reads lane 7 - i (the other group of four)
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xF, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xF, 0xF, false);
    } else if (ctrl_sel == 3) {  // row_mirror: lane i of each row of 16 reads lane 15 - i (the other group of eight)
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xF, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xF, 0xF, false);
    } else {  // row_ror:8: lane i of each row of 16 reads lane (i + 8) mod 16 = lane ^ 8
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x128, 0xF, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x128, 0xF, 0xF, false);
    }
    return __hiloint2double(hi, lo);
}

// quad_perm DPP of a double: lane l of every quad reads lane SEL[l mod 4] of its quad (CTRL = SEL0 | SEL1 << 2 | SEL2 << 4 | SEL3 << 6;
// 0x00 / 0x55 / 0xAA / 0xFF broadcast lane 0 / 1 / 2 / 3 of the quad)
template <int CTRL>
__device__ __attribute__((always_inline)) inline double dpp_quad_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n in 0..63 (the instruction takes an immediate; the counter has 6 bits)
__device__ inline void wait_vmcnt_le(int n) {
#define NGP_VMC(i) case i: asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); break;
#define NGP_VMC8(b) NGP_VMC(b) NGP_VMC(b + 1) NGP_VMC(b + 2) NGP_VMC(b + 3) NGP_VMC(b + 4) NGP_VMC(b + 5) NGP_VMC(b + 6) NGP_VMC(b + 7)
    switch (n) {
        NGP_VMC(1) NGP_VMC(2) NGP_VMC(3) NGP_VMC(4) NGP_VMC(5) NGP_VMC(6) NGP_VMC(7) NGP_VMC(8) NGP_VMC(9) NGP_VMC(10) NGP_VMC(11)
        NGP_VMC(12) NGP_VMC(13) NGP_VMC(14) NGP_VMC(15) NGP_VMC(16) NGP_VMC(17) NGP_VMC(18) NGP_VMC(19) NGP_VMC(20) NGP_VMC(21)
        NGP_VMC(22) NGP_VMC(23) NGP_VMC(24) NGP_VMC(25) NGP_VMC(26) NGP_VMC(27) NGP_VMC(28) NGP_VMC(29) NGP_VMC(30) NGP_VMC(31)
        NGP_VMC(32) NGP_VMC(33) NGP_VMC(34) NGP_VMC(35) NGP_VMC(36) NGP_VMC(37) NGP_VMC(38) NGP_VMC(39) NGP_VMC(40) NGP_VMC(41)
        NGP_VMC(42) NGP_VMC(43) NGP_VMC(44) NGP_VMC(45) NGP_VMC(46) NGP_VMC(47) NGP_VMC(48) NGP_VMC(49) NGP_VMC(50) NGP_VMC(51)
        NGP_VMC(52) NGP_VMC(53) NGP_VMC(54) NGP_VMC(55) NGP_VMC(56) NGP_VMC(57) NGP_VMC(58) NGP_VMC(59) NGP_VMC(60) NGP_VMC(61)
        NGP_VMC(62) NGP_VMC(63)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef NGP_VMC8
#undef NGP_VMC
}

// ST = 0: fp32 tiles, a ring slot holds a quad (4 rows x 64 columns x 4 bytes).  ST = 1, 2, 4: compact storage, a ring slot
// holds a unit of 16 rows x 64 columns x 1 byte, wave w owns units w, w+7, ..., and a lane carries ST update tasks of
// 4 rows x 8 columns (8 VGPRs per task and lag instead of 32: lags 3, 4, 6, 8, 12 with one task, 4 and 8 with two, 4 with four; shards of up to 896 rows).  Centring is
// analytic: the shard partial is sum_i g_ij y_i - m_j sum_i y_i, the update subtracts
// sum_j g_ij dlt_j - sum_j m_j dlt_j from the valid rows.
template <bool DBG, int DT, int ST>
__device__ __attribute__((always_inline)) inline void role_streamer_rows(const SweepArgs &A, const int s, char *smem) {
    NGP_DBG_ROWS
    constexpr bool U8 = (ST != 0);
    constexpr int NT = U8 ? ST : 1;
    const int R = A.R, S = A.S, tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int NQ = U8 ? (R >> 4) : (R >> 2);                 // 1 KiB ring slots per tile: quads (fp32) or units of 16 rows (bytes)
    const int H = min(NGP_ROWS_HMAX, (NQ + 1) >> 1);         // quads of a tile requested one block earlier than the rest
    const int RQ = 2 * NQ + H;                               // ring slots: tile u in use, tile u+1 complete, H quads of tile u+2
    char *ring = smem;
    double *ys = (double *)(smem + (size_t)RQ * NGP_QS);     // the shard of ycorr, resident for the whole sweep
    double *red = ys + ((R + 7) & ~7);                       // 2 (block parity) x 7 chains x 64 columns
    double *dl = red + 2 * NGP_ROWS_NW * NGP_BLK;            // 2 (block parity) x 72: dlt of the block being applied (+ sum_j m_j dlt_j)
    double *rsy = dl + 2 * NGP_DLS;                          // 2 (block parity) x 8: the waves' sums of their rows of ycorr (compact storage)
    int *sflag = (int *)(rsy + 16);
    char *scratch = (char *)(rsy + 16) + 64;                 // 1 KiB sink of the L2-warming DMA
    unsigned long long *fine = (unsigned long long *)(scratch + 1024);  // diagnostic timeline (DBG only, if it fits)
    const bool fine_on = DBG && dbg && s == 1 && A.fine_ok == 1;
#define NGP_FINE(k)                                                                                              \
    do {                                                                                                         \
        if (fine_on && (unsigned)(u - 800) < 16u && lane == 0) fine[(((u - 800) * 8 + wv) << 3) + (k)] = wall_clock64(); \
    } while (0)
    const size_t tile_bytes = (size_t)NQ * 1024;
    double *yg = A.ycorr + (size_t)s * R;
    const int g = s / NGP_GRP;
    const int nb = A.t1 - A.t0;
    for (int i = tid; i < R; i += NGP_WG) ys[i] = yg[i];
    // chains of the current block that have left their sum in `red` (two parities; the eighth slots of rsy are free)
    int *gcnt0 = (int *)(rsy + 7), *gcnt1 = (int *)(rsy + 15);
    // the publisher does not wait for the barrier (see below).  fp32 tiles, where the streamers wait for the loader: 50k x 600k
    // 2.66-2.69 -> 2.57 us per block (same box); byte tiles, where their own arithmetic is the bound: 2.05 -> 2.31, so not there
    const bool early = U8 ? ((A.knob & 256) != 0) : ((A.knob & 256) == 0);
    if (tid == 0) { *sflag = 1; *gcnt0 = 0; *gcnt1 = 0; }
    int base = 0;  // ring slot of quad 0 of tile u
    auto wrap = [&](int p) __attribute__((always_inline)) { return p >= RQ ? p - RQ : p; };
    if (wv == NGP_ROWS_NW) {
        // ------------------------------ loader ------------------------------
        const unsigned ring0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)ring;
        const unsigned scratch0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)scratch;
        const unsigned voff = (unsigned)lane * 16u;
        const bool no_dma = DBG && dbg_mode == 4;  // timing experiment: the arithmetic without the stream
        // Pacing: a short sleep after every four requests spreads them over the block.  Requested back to back (1.4 us per tile)
        // they stream faster in isolation (2.14 against 2.47 us per block) but every other memory access of this CU -- the partial
        // sums going out, dlt coming in -- then queues behind a full tile of requests, and the sweep is bound by that hand-off
        // loop, not by the stream.  Capping the outstanding requests with counted vmcnt waits instead was far worse (4.9 us).
        const int pace = A.knob & 7;
        // quads [q0, q1) of local tile `tile` into ring slots tbase + q (everything wave-uniform: scalar registers only)
        auto dma_quads = [&](int tile, int q0, int q1, int tbase) __attribute__((always_inline)) {
            if (no_dma || q0 >= q1) return 0;
            const char *g = (const char *)A.tiles + ((size_t)(A.t0 + tile) * S + s) * tile_bytes + (size_t)q0 * 1024;
            int p = wrap(tbase + q0);
            int q = q0;
            for (; q + 4 <= q1; q += 4) {
                if (p + 4 <= RQ) {
                    dma16_s4(ring0 + (unsigned)p * NGP_QS, g, voff);
                    p += 4;
                    if (p == RQ) p = 0;
                    switch (pace) {  // s_sleep takes an immediate (units of 64 clocks)
                        case 1: __builtin_amdgcn_s_sleep(1); break;
                        case 2: __builtin_amdgcn_s_sleep(2); break;
                        case 3: __builtin_amdgcn_s_sleep(3); break;
                        case 4: __builtin_amdgcn_s_sleep(4); break;
                        default: break;
                    }
                } else {
                    for (int k = 0; k < 4; k++) {
                        dma16_s(ring0 + (unsigned)p * NGP_QS, g + k * 1024, voff);
                        if (++p == RQ) p = 0;
                    }
                }
                g += 4096;
            }
            for (; q < q1; ++q) {
                dma16_s(ring0 + (unsigned)p * NGP_QS, g, voff);
                g += 1024;
                if (++p == RQ) p = 0;
            }
            return q1 - q0;
        };
        const unsigned my_xcc = xcc_id() + 1u;
        const int nslice = max(1, S / 8);
        const int slice = (s / 8) % nslice;
        // (only the planes the sampler itself reads: the diagonal block and the near lags; the far ones go to the reducers' CUs)
        const size_t gram_bytes = warm_bytes(A, min(DT, A.near + 1)), w_thi = warm_t_hi(A), w_shift = warm_shift(A);
        const size_t slice_bytes = ((gram_bytes / nslice + 1023) / 1024) * 1024;
        bool same_xcd = false, xcc_known = false;
        __builtin_amdgcn_s_setprio(3);  // the loader's few scalar instructions go first on its SIMD: a late request costs the whole CU
        if (nb > 0) dma_quads(0, 0, NQ, 0);
        if (nb > 1) dma_quads(1, 0, H, NQ);
        drain_vm();  // tile 0 has landed
        wg_barrier();
        for (int u = 0; u < nb + DT; ++u) {
            NGP_FINE(0);
            const int base1 = wrap(base + NQ), base2 = wrap(base1 + NQ);
            // speed only: streamers on the sampler's XCD pull the Gram blocks of the next block into that XCD's L2 (the
            // sampler is one latency-bound CU); requested first, so that the counted wait below covers them
            // (the sampler's XCC id is read until it is known -- a scalar load that bypasses the scalar cache costs the
            // loader about 1.5 us, during which it requests nothing: re-read every eighth block for the whole sweep, as the phase
            // streamer does, it put 511 blocks of 6 us and more into a 50k x 600k sweep)
            if (!xcc_known && (u & 7) == 0) {
                const unsigned x = sld_u32(A.xcc_w);
                xcc_known = (x != 0u);
                same_xcd = (x == my_xcc);
            }
            if (same_xcd && u + 1 < nb && !no_dma) {
                const char *gb = (const char *)(A.gramx + (size_t)(A.t0 + u + 1) * DT * NGP_BLK * NGP_BLK) - w_shift;
                const char *tb = (const char *)(A.tinv + (size_t)(A.t0 + u + 1) * NGP_BLK * NGP_BLK);
                const size_t lo = (size_t)slice * slice_bytes, hi = min((size_t)(slice + 1) * slice_bytes, gram_bytes);
                for (size_t off = lo; off + 1024 <= hi; off += 1024) dma16_warm(scratch0, (off < w_thi ? tb : gb) + off, voff);
            }
            // (Requesting more of tile u+2 here -- into the slots of tile u, which the row waves have left 1.7 us into the block,
            // so that more than H requests are in flight while the loader sits at the barrier -- was built and measured: 50k x 600k
            // 24.1 -> 25.1 ms per iteration.  The stream itself gains, but the partial sums going out and dlt coming in queue behind
            // the deeper request queue, and the hand-off loop is what bounds the sweep.)
            if (u + 1 < nb) dma_quads(u + 1, H, NQ, base1);
            int n2 = 0;
            if (u + 2 < nb) n2 = dma_quads(u + 2, 0, H, base2);
            NGP_FINE(1);
            wait_vmcnt_le(n2);  // everything up to the last quad of tile u+1 has landed
            NGP_FINE(2);
            wg_barrier();
            if (!*sflag) return;
            base = base1;
        }
        drain_vm();
    } else {
        // ------------------------------ row-owning waves ------------------------------
        const int c = lane & 7, ql = lane >> 3;
        const int nqw = (NQ - wv + NGP_ROWS_NW - 1) / NGP_ROWS_NW;  // ring slots (quads / units) of this wave (wave-uniform)
        // fp32: the update task of this lane is quad wv + 7 ql.  Compact: task i is tau = ql + 8 i -> unit wv + 7 (tau >> 2), row quad
        // tau & 3 of that unit.  Idle lanes shadow a valid slot (nothing is stored).
        int tslot[NT], trow[NT];   // ring slot (relative to the tile) and first row of task i
        bool thas[NT];
#pragma unroll
        for (int i = 0; i < NT; i++) {
            if constexpr (U8) {
                const int tau = ql + 8 * i;
                thas[i] = (tau >> 2) < nqw;
                tslot[i] = thas[i] ? wv + NGP_ROWS_NW * (tau >> 2) : wv;
                trow[i] = 16 * tslot[i] + 4 * (tau & 3);
            } else {
                const int qt = wv + NGP_ROWS_NW * ql;
                thas[i] = qt < NQ;
                tslot[i] = thas[i] ? qt : wv;
                trow[i] = 4 * qt;
            }
        }
        const int nvalid = U8 ? (int)max(0ll, min((long long)R, A.N - (long long)s * R)) : R;  // rows of this shard inside the panel
        // tile elements of the update tasks: fp32 rows 4 qt..4 qt+3 (x, y, z, w) of columns 8 c + jj; compact: the same four rows as
        // the bytes of one word
        float4 keep[U8 ? 1 : DT][8];
        unsigned keep8[U8 ? DT : 1][NT][8];
#pragma unroll
        for (int d = 0; d < DT; d++)
#pragma unroll
            for (int jj = 0; jj < 8; jj++) {
                if constexpr (U8) {
#pragma unroll
                    for (int i = 0; i < NT; i++) keep8[d][i][jj] = 0u;
                } else {
                    keep[d][jj] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        unsigned long long pg0 = 0, pg1 = 0;  // poller: granules of the next dlt as read just before the previous barrier
        double pm = 0.0;                      // poller, compact storage: column mean of (block pm_blk, column lane)
        int pm_blk = -1;
        double mpub = 0.0;                    // publisher, compact storage: column mean of the block published next
        if (U8 && wv == NGP_ROWS_PUBW && nb > 0) mpub = A.mean[(size_t)A.t0 * NGP_BLK + lane];
        // publisher: the stored, not yet counted partial (ring slot), counted once VM_CNT of this wave reads zero
        const double fxs = A.scal->fx_scale;
        auto try_signal = [&](bool) __attribute__((always_inline)) {};  // (nothing to count since the partial sums are atomic adds)
        // the 64 partial dot products of block u: the 7 chains in their fixed tree (compact storage: minus m_j x the shard's sum of y)
        auto publish = [&](const int u) __attribute__((always_inline)) {
            const int slot = u % NGP_RING;
            const double *rp = red + (u & 1) * NGP_ROWS_NW * NGP_BLK + lane;
            double p = ((rp[0] + rp[NGP_BLK]) + (rp[2 * NGP_BLK] + rp[3 * NGP_BLK])) + ((rp[4 * NGP_BLK] + rp[5 * NGP_BLK]) + rp[6 * NGP_BLK]);
            if constexpr (U8) {  // partial = sum_i g_ij y_i - m_j sum_i y_i
                const double *sp = rsy + (u & 1) * 8;
                const double sy = ((sp[0] + sp[1]) + (sp[2] + sp[3])) + ((sp[4] + sp[5]) + sp[6]);
                const double ms = mpub * sy;
                p = p - ms;
            }
            // (the next block's column means are requested BEFORE the atomic add leaves: memory operations of a wave return in order,
            // and the add's acknowledgement -- a read-modify-write at the memory side -- is the slowest of them)
            if (U8 && u + 1 < nb) mpub = A.mean[(size_t)(A.t0 + u + 1) * NGP_BLK + lane];
            acc_add(A.acc, A.abort_w, slot, s, lane, p, fxs);
            if (DBG && dbg && s == 0 && lane == 0) dbg[NGP_DBG_STREAM + 2 * (size_t)u] = wall_clock64();
        };
        wg_barrier();
        for (int u0 = 0; u0 < nb + DT; u0 += DT) {
#pragma unroll
            for (int d = 0; d < DT; d++) {
                const int u = u0 + d;
                if (u >= nb + DT) break;
                const int a = u - DT;  // block whose update is applied now (its tile waits in keep[d])
                NGP_FINE(0);
                // poller: dlt of the block applied in the NEXT iteration; the flag travels while this wave works
                const int pa = u + 1 - DT;
                const bool pollw = (wv == NGP_ROWS_POLLW) && (pa >= 0) && (u + 1 < nb + DT) && !(DBG && dbg_mode == 1);
                // (every load of a streaming CU queues behind a microsecond of tile requests: dlt comes as self-validating granules,
                // ONE load, requested before the previous barrier; only if that was too early is it requested again here)
                bool have_dnext = false;
                if (pollw) {
                    if (dlt_granules_valid(pg0, pg1, dlt_tag(A.nonce, pa))) {
                        have_dnext = true;
                    } else {  // asked for too early: ask again, the answer travels while this wave does its arithmetic
                        const unsigned long long *gp = A.dltg + ((size_t)(pa % NGP_RING) * NGP_BLK + lane) * 2;
                        pg0 = ld_u64(gp);
                        pg1 = ld_u64(gp + 1);
                    }
                    if (U8 && pm_blk != pa) { pm = A.mean[(size_t)(A.t0 + pa) * NGP_BLK + lane]; pm_blk = pa; }
                }
                // ---- ycorr -= X_a dlt_a for the rows of this wave ----
                if constexpr (!U8) {
                if (a >= 0 && !(DBG && dbg_mode == 1)) {
                    const double *dq = dl + (u & 1) * NGP_DLS + 8 * c;
                    double dqv[8];
#pragma unroll
                    for (int jj = 0; jj < 8; jj++) dqv[jj] = dq[jj];
                    double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll
                    for (int jj = 0; jj < 8; jj++) {
                        p0 = __builtin_fma((double)keep[d][jj].x, dqv[jj], p0);
                        p1 = __builtin_fma((double)keep[d][jj].y, dqv[jj], p1);
                        p2 = __builtin_fma((double)keep[d][jj].z, dqv[jj], p2);
                        p3 = __builtin_fma((double)keep[d][jj].w, dqv[jj], p3);
                    }
                    // ((p_0+p_1)+(p_2+p_3))+((p_4+p_5)+(p_6+p_7)) over the 8 chains = the 8 lanes of the group
                    p0 = p0 + dpp_f64(p0, 0); p1 = p1 + dpp_f64(p1, 0); p2 = p2 + dpp_f64(p2, 0); p3 = p3 + dpp_f64(p3, 0);
                    p0 = p0 + dpp_f64(p0, 1); p1 = p1 + dpp_f64(p1, 1); p2 = p2 + dpp_f64(p2, 1); p3 = p3 + dpp_f64(p3, 1);
                    p0 = p0 + dpp_f64(p0, 2); p1 = p1 + dpp_f64(p1, 2); p2 = p2 + dpp_f64(p2, 2); p3 = p3 + dpp_f64(p3, 2);
                    if (c == 0 && thas[0]) {
                        double *yq = ys + trow[0];
                        const double y0 = yq[0], y1 = yq[1], y2 = yq[2], y3 = yq[3];
                        yq[0] = y0 - p0; yq[1] = y1 - p1; yq[2] = y2 - p2; yq[3] = y3 - p3;
                    }
                }
                } else if (!(DBG && dbg_mode == 1)) {
                    // compact storage: y_i -= (sum_j g_ij dlt_j - sum_j m_j dlt_j) for the rows inside the panel, and the sum of
                    // this wave's rows of y as the GEMV below sees them (lane slots in task order, then a fixed tree over the slots)
                    const double *dq = dl + (u & 1) * NGP_DLS + 8 * c;
                    double dqv[8];
#pragma unroll
                    for (int jj = 0; jj < 8; jj++) dqv[jj] = (a >= 0) ? dq[jj] : 0.0;
                    const double cm = (a >= 0) ? dl[(u & 1) * NGP_DLS + NGP_BLK] : 0.0;
                    double vsum = 0.0;
#pragma unroll
                    for (int i = 0; i < NT; i++) {
                        double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
                        if (a >= 0) {
#pragma unroll
                            for (int jj = 0; jj < 8; jj++) {
                                const unsigned w = keep8[d][i][jj];
                                p0 = __builtin_fma((double)(float)(w & 0xffu), dqv[jj], p0);
                                p1 = __builtin_fma((double)(float)((w >> 8) & 0xffu), dqv[jj], p1);
                                p2 = __builtin_fma((double)(float)((w >> 16) & 0xffu), dqv[jj], p2);
                                p3 = __builtin_fma((double)(float)(w >> 24), dqv[jj], p3);
                            }
                            p0 = p0 + dpp_f64(p0, 0); p1 = p1 + dpp_f64(p1, 0); p2 = p2 + dpp_f64(p2, 0); p3 = p3 + dpp_f64(p3, 0);
                            p0 = p0 + dpp_f64(p0, 1); p1 = p1 + dpp_f64(p1, 1); p2 = p2 + dpp_f64(p2, 1); p3 = p3 + dpp_f64(p3, 1);
                            p0 = p0 + dpp_f64(p0, 2); p1 = p1 + dpp_f64(p1, 2); p2 = p2 + dpp_f64(p2, 2); p3 = p3 + dpp_f64(p3, 2);
                        }
                        if (c == 0 && thas[i]) {
                            double *yq = ys + trow[i];
                            double y0 = yq[0], y1 = yq[1], y2 = yq[2], y3 = yq[3];
                            if (a >= 0) {
                                const int r0 = trow[i];
                                const double t0 = p0 - cm, t1 = p1 - cm, t2 = p2 - cm, t3 = p3 - cm;
                                if (r0 + 0 < nvalid) y0 = y0 - t0;
                                if (r0 + 1 < nvalid) y1 = y1 - t1;
                                if (r0 + 2 < nvalid) y2 = y2 - t2;
                                if (r0 + 3 < nvalid) y3 = y3 - t3;
                                yq[0] = y0; yq[1] = y1; yq[2] = y2; yq[3] = y3;
                            }
                            const double q4 = (y0 + y1) + (y2 + y3);
                            vsum = (i == 0) ? q4 : vsum + q4;
                        }
                    }
                    if (u < nb) {  // ((v_0+v_1)+(v_2+v_3))+((v_4+v_5)+(v_6+v_7)) over the lane slots (lanes 0, 8, ..., 56)
                        vsum = vsum + dpp_f64(vsum, 4);  // slots 2r, 2r+1 live in lanes 0 and 8 of row r
                        const double r0 = readlane_d(vsum, 0), r1 = readlane_d(vsum, 16), r2 = readlane_d(vsum, 32), r3 = readlane_d(vsum, 48);
                        if (lane == 0) rsy[(u & 1) * 8 + wv] = (r0 + r1) + (r2 + r3);
                    }
                }
                NGP_FINE(1);
                try_signal(false);
                if (u < nb) {
                    // ---- GEMV chain of this wave: slots wv, wv+7, ... (lane = column); LDS serves a wave in order, so the
                    //      rows written above are read back without a barrier ----
                    double acc = 0.0;
                    for (int k = 0; k < nqw; k++) {
                        const int q = wv + NGP_ROWS_NW * k;
                        if constexpr (U8) {
                            const uint4 x = *(const uint4 *)(ring + (size_t)wrap(base + q) * NGP_QS + (size_t)lane * 16);
                            const double *yq = ys + 16 * q;
                            const unsigned xw[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                            for (int e4 = 0; e4 < 4; e4++) {
                                const unsigned w = xw[e4];
                                const double y0 = yq[4 * e4], y1 = yq[4 * e4 + 1], y2 = yq[4 * e4 + 2], y3 = yq[4 * e4 + 3];
                                acc = __builtin_fma((double)(float)(w & 0xffu), y0, acc);
                                acc = __builtin_fma((double)(float)((w >> 8) & 0xffu), y1, acc);
                                acc = __builtin_fma((double)(float)((w >> 16) & 0xffu), y2, acc);
                                acc = __builtin_fma((double)(float)(w >> 24), y3, acc);
                            }
                        } else {
                            const float4 x = *(const float4 *)(ring + (size_t)wrap(base + q) * NGP_QS + (size_t)lane * 16);
#if NGP_ROWS_DPP
                            // the quad's four y: ONE 8-byte read per lane (lane l: y[4 q + l mod 4]) and the DPP broadcast, instead of
                            // 32 bytes per lane of broadcast reads -- a quarter of this loop's LDS read cycles beside the tile DMA
                            const double xd[4] = {(double)x.x, (double)x.y, (double)x.z, (double)x.w};
                            double yv[1] = {ys[4 * q + (lane & 3)]}, ac[1] = {acc};
                            fmac4_bcast<1>(ac, yv, xd);
                            acc = ac[0];
#else
                            const double *yq = ys + 4 * q;
                            const double y0 = yq[0], y1 = yq[1], y2 = yq[2], y3 = yq[3];
                            acc = __builtin_fma((double)x.x, y0, acc);
                            acc = __builtin_fma((double)x.y, y1, acc);
                            acc = __builtin_fma((double)x.z, y2, acc);
                            acc = __builtin_fma((double)x.w, y3, acc);
#endif
                        }
                    }
                    red[((u & 1) * NGP_ROWS_NW + wv) * NGP_BLK + lane] = acc;
                    if (early) {
                        asm volatile("" ::: "memory");  // LDS serves a wave in order: the count follows the sum
                        if (lane == 0) __hip_atomic_fetch_add((lds_int_t *)((u & 1) ? gcnt1 : gcnt0), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    NGP_FINE(2);
                    try_signal(false);
                    // ---- tile u into the delay line (task view of the same slots) ----
                    if constexpr (U8) {
#pragma unroll
                        for (int i = 0; i < NT; i++) {
                            const char *tq = ring + (size_t)wrap(base + tslot[i]) * NGP_QS + c * 128 + (trow[i] & 15);
#pragma unroll
                            for (int jj = 0; jj < 8; jj++) keep8[d][i][jj] = *(const unsigned *)(tq + jj * 16);
                        }
                    } else {
                        const char *tq = ring + (size_t)wrap(base + tslot[0]) * NGP_QS + c * 128;
#pragma unroll
                        for (int jj = 0; jj < 8; jj++) keep[d][jj] = *(const float4 *)(tq + jj * 16);
                    }
                }
                NGP_FINE(3);
                // The publisher does not wait for the block's barrier (which the loader reaches a microsecond after the chains are
                // done): it waits for the seven chains through a counter in LDS and publishes at once -- the partial enters the
                // hand-off loop that much earlier, and the period of the sweep is that loop's latency / (lag - 1).
                if (early && wv == NGP_ROWS_PUBW && u < nb && !(DBG && dbg_mode == 1)) {
                    const int *gc = (u & 1) ? gcnt1 : gcnt0;
                    for (unsigned sp = 0; lds_flag_ld(gc) < NGP_ROWS_NW; ++sp) {
                        if ((sp & 255u) == 255u && (lds_flag_ld(sflag) == 0 || sp > (NGP_SPIN_LIMIT << 4))) {
                            if (lane == 0) *sflag = 0;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(0);
                    }
                    asm volatile("" ::: "memory");
                    if (lane == 0) lds_flag_st((u & 1) ? gcnt0 : gcnt1, 0);  // the other parity: counted in the next block, behind the barrier
                    publish(u);
                }
                if (pollw) {
                    int ok = 1;
                    if (!have_dnext && !(DBG && (dbg_mode == 3 || dbg_mode == 4))) {
                        ok = wait_dlt_granules_all(A.dltg, A.nonce, pa, lane, A.abort_w, 1u, pg0, pg1) ? 1 : 0;
                        if (!ok && lane == 0) *sflag = 0;
                    }
                    if (ok) {
                        const double dv = dlt_granules_value(pg0, pg1);
                        dl[((u + 1) & 1) * NGP_DLS + lane] = dv;
                        if constexpr (U8) {  // sum_j m_j dlt_j: butterfly over the 64 lanes, xor 1, 2, 4, 8 (DPP), then 16, 32 (the four rows)
                            double v = pm * dv;
                            v = v + dpp_f64(v, 0);
                            v = v + dpp_f64(v, 1);
                            v = v + dpp_f64(v, 2);  // quads are uniform by now: the mirror is lane ^ 4
                            v = v + dpp_f64(v, 3);  // groups of eight are uniform: lane ^ 8
                            const double r0 = readlane_d(v, 0), r1 = readlane_d(v, 16), r2 = readlane_d(v, 32), r3 = readlane_d(v, 48);
                            if (lane == 0) dl[((u + 1) & 1) * NGP_DLS + NGP_BLK] = (r0 + r1) + (r2 + r3);
                        }
                    }
                }
                if (wv == NGP_ROWS_POLLW && pa + 1 >= 0 && u + 2 < nb + DT && !(DBG && dbg_mode == 1)) {
                    // dlt of the block after that: looked at behind the barrier (if the sampler is that far, the next block pays nothing)
                    const unsigned long long *gp = A.dltg + ((size_t)((pa + 1) % NGP_RING) * NGP_BLK + lane) * 2;
                    pg0 = ld_u64(gp);
                    pg1 = ld_u64(gp + 1);
                    if (U8 && pa + 1 < nb) { pm = A.mean[(size_t)(A.t0 + pa + 1) * NGP_BLK + lane]; pm_blk = pa + 1; }
                }
                NGP_FINE(4);
                // Lag 3: dlt of block u-2, which the poller of this workgroup waits for before this barrier, needs the partial of
                // block u-1 of EVERY streamer counted (the sampler fetches the group sums of the next block before it lets a
                // block go) -- so the count must not slip behind the barrier.  From lag 4 on it may (needs: partials <= u-2).
                try_signal(DT < 4 || (A.knob & 16));
                wg_barrier();
                if (!*sflag) return;
                NGP_FINE(5);
                if (!early && wv == NGP_ROWS_PUBW && u < nb && !(DBG && dbg_mode == 1)) publish(u);
                base = wrap(base + NQ);
                NGP_FINE(6);
            }
        }
        try_signal(true);
    }
    if (fine_on && nb > 816) {
        __syncthreads();
        for (int i = tid; i < 1024; i += NGP_WG) dbg[NGP_DBG_ALL + 8192 + i] = fine[i];
    }
#undef NGP_FINE
    __syncthreads();
    for (int i = tid; i < R; i += NGP_WG) yg[i] = ys[i];
}

// ------------------------------------------------------------------------------------------
// Row-owning streamer with V shards per workgroup (fp32 tiles of panels too tall for one resident wave of 224-row shards:
// N above 63k).  The layout stays what it is everywhere else -- S shards of R <= 224 rows, one partial per shard and block, the
// reducers' groups of 32 shards -- so the chain is, bit for bit, the chain of that layout; only the assignment changes: workgroup
// s owns shards V s .. V s + V - 1 (the host makes S a multiple of V) and takes a block in V sub-steps, one tile each.  The LDS
// ring (2 tiles + H quads), the loader wave, the one barrier per tile, the publisher and the poller are those of
// role_streamer_rows; the delay line holds V tiles per lag (32 V VGPRs), so V = 2 runs at lag 3 and V = 3 at lag 2 -- enough
// there: a block takes V times as long on the stream side, and the hand-off loop's latency is hidden by (lag - 1) block periods.
template <int DT, int V>
__device__ __attribute__((always_inline)) inline void role_streamer_rows_tall(const SweepArgs &A, const int s, char *smem) {
    const int R = A.R, S = A.S, tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int NQ = R >> 2;
    const int H = min(NGP_ROWS_HMAX, (NQ + 1) >> 1);
    const int RQ = 2 * NQ + H;
    const int RP = (R + 7) & ~7;
    char *ring = smem;
    double *ys = (double *)(smem + (size_t)RQ * NGP_QS);     // the V shards of ycorr, resident for the whole sweep
    double *red = ys + V * RP;                               // 2 (sub-step parity) x 7 chains x 64 columns
    double *dl = red + 2 * NGP_ROWS_NW * NGP_BLK;            // 2 (block parity) x 72: dlt of the block being applied
    double *rsy = dl + 2 * NGP_DLS;                          // (only the chain counters live here)
    int *sflag = (int *)(rsy + 16);
    char *scratch = (char *)(rsy + 16) + 64;                 // 1 KiB sink of the L2-warming DMA
    const size_t tile_bytes = (size_t)NQ * 1024;
    const int sh0 = V * s;                                   // first shard of this workgroup
    double *yg = A.ycorr + (size_t)sh0 * R;
    const int nb = A.t1 - A.t0;
    const int nv = (nb + DT) * V;                            // sub-steps of the sweep
    for (int i = tid; i < V * R; i += NGP_WG) ys[(i / R) * RP + (i % R)] = yg[i];
    int *gcnt0 = (int *)(rsy + 7), *gcnt1 = (int *)(rsy + 15);
    if (tid == 0) { *sflag = 1; *gcnt0 = 0; *gcnt1 = 0; }
    int base = 0;  // ring slot of quad 0 of the tile of this sub-step
    auto wrap = [&](int p) __attribute__((always_inline)) { return p >= RQ ? p - RQ : p; };
    if (wv == NGP_ROWS_NW) {
        // ------------------------------ loader ------------------------------
        const unsigned ring0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)ring;
        const unsigned scratch0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)scratch;
        const unsigned voff = (unsigned)lane * 16u;
        const int pace = A.knob & 7;
        // quads [q0, q1) of the tile of sub-step w = (block w / V, shard sh0 + w % V) into ring slots tbase + q
        auto dma_quads = [&](int w, int q0, int q1, int tbase) __attribute__((always_inline)) {
            const int u = w / V, hh = w % V;
            if (u >= nb || q0 >= q1) return 0;
            const char *g = (const char *)A.tiles + ((size_t)(A.t0 + u) * S + sh0 + hh) * tile_bytes + (size_t)q0 * 1024;
            int p = wrap(tbase + q0);
            int q = q0;
            for (; q + 4 <= q1; q += 4) {
                if (p + 4 <= RQ) {
                    dma16_s4(ring0 + (unsigned)p * NGP_QS, g, voff);
                    p += 4;
                    if (p == RQ) p = 0;
                    switch (pace) {
                        case 1: __builtin_amdgcn_s_sleep(1); break;
                        case 2: __builtin_amdgcn_s_sleep(2); break;
                        case 3: __builtin_amdgcn_s_sleep(3); break;
                        case 4: __builtin_amdgcn_s_sleep(4); break;
                        default: break;
                    }
                } else {
                    for (int k = 0; k < 4; k++) {
                        dma16_s(ring0 + (unsigned)p * NGP_QS, g + k * 1024, voff);
                        if (++p == RQ) p = 0;
                    }
                }
                g += 4096;
            }
            for (; q < q1; ++q) {
                dma16_s(ring0 + (unsigned)p * NGP_QS, g, voff);
                g += 1024;
                if (++p == RQ) p = 0;
            }
            return q1 - q0;
        };
        const unsigned my_xcc = xcc_id() + 1u;
        const int W = S / V;
        const int nslice = max(1, W / 8);
        const int slice = (s / 8) % nslice;
        const size_t gram_bytes = warm_bytes(A, min(DT, A.near + 1)), w_thi = warm_t_hi(A), w_shift = warm_shift(A);
        const size_t slice_bytes = ((gram_bytes / nslice + 1023) / 1024) * 1024;
        bool same_xcd = false, xcc_known = false;
        __builtin_amdgcn_s_setprio(3);
        dma_quads(0, 0, NQ, 0);
        dma_quads(1, 0, H, NQ);
        drain_vm();  // the first tile has landed
        wg_barrier();
        for (int w = 0; w < nv; ++w) {
            const int base1 = wrap(base + NQ), base2 = wrap(base1 + NQ);
            if (w % V == 0) {  // speed only: Gram blocks of the next block into the sampler's L2 (see role_streamer_rows)
                const int u = w / V;
                if (!xcc_known && (u & 7) == 0) {
                    const unsigned x = sld_u32(A.xcc_w);
                    xcc_known = (x != 0u);
                    same_xcd = (x == my_xcc);
                }
                if (same_xcd && u + 1 < nb) {
                    const char *gb = (const char *)(A.gramx + (size_t)(A.t0 + u + 1) * DT * NGP_BLK * NGP_BLK) - w_shift;
                    const char *tb = (const char *)(A.tinv + (size_t)(A.t0 + u + 1) * NGP_BLK * NGP_BLK);
                    const size_t lo = (size_t)slice * slice_bytes, hi = min((size_t)(slice + 1) * slice_bytes, gram_bytes);
                    for (size_t off = lo; off + 1024 <= hi; off += 1024) dma16_warm(scratch0, (off < w_thi ? tb : gb) + off, voff);
                }
            }
            dma_quads(w + 1, H, NQ, base1);
            const int n2 = dma_quads(w + 2, 0, H, base2);
            wait_vmcnt_le(n2);  // everything up to the last quad of the next tile has landed
            wg_barrier();
            if (!*sflag) return;
            base = base1;
        }
        drain_vm();
    } else {
        // ------------------------------ row-owning waves ------------------------------
        const int c = lane & 7, ql = lane >> 3;
        const int nqw = (NQ - wv + NGP_ROWS_NW - 1) / NGP_ROWS_NW;
        const int qt = wv + NGP_ROWS_NW * ql;      // the quad of this lane's update task
        const bool thas = qt < NQ;
        const int tslot = thas ? qt : wv, trow = 4 * qt;
        float4 keep[DT][V][8];
#pragma unroll
        for (int d = 0; d < DT; d++)
#pragma unroll
            for (int hh = 0; hh < V; hh++)
#pragma unroll
                for (int jj = 0; jj < 8; jj++) keep[d][hh][jj] = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned long long pg0 = 0, pg1 = 0;
        const double fxs = A.scal->fx_scale;
        auto try_signal = [&](bool) __attribute__((always_inline)) {};
        auto publish = [&](const int u, const int hh, const int par) __attribute__((always_inline)) {
            const int slot = u % NGP_RING;
            const double *rp = red + par * NGP_ROWS_NW * NGP_BLK + lane;
            const double p = ((rp[0] + rp[NGP_BLK]) + (rp[2 * NGP_BLK] + rp[3 * NGP_BLK])) + ((rp[4 * NGP_BLK] + rp[5 * NGP_BLK]) + rp[6 * NGP_BLK]);
            acc_add(A.acc, A.abort_w, slot, sh0 + hh, lane, p, fxs);
        };
        wg_barrier();
        for (int u0 = 0; u0 < nb + DT; u0 += DT) {
#pragma unroll
            for (int d = 0; d < DT; d++) {
                const int u = u0 + d;
                if (u >= nb + DT) break;
                const int a = u - DT;  // block whose update is applied now (its tiles wait in keep[d])
                const int pa = u + 1 - DT;
#pragma unroll
                for (int hh = 0; hh < V; hh++) {
                    const int par = (V & 1) ? ((u * V + hh) & 1) : (hh & 1);
                    double *ysh = ys + hh * RP;
                    // poller (last sub-step of the block): dlt of the block applied in the next one
                    const bool pollw = (wv == NGP_ROWS_POLLW) && (hh == V - 1) && (pa >= 0) && (u + 1 < nb + DT);
                    bool have_dnext = false;
                    if (pollw) {
                        if (dlt_granules_valid(pg0, pg1, dlt_tag(A.nonce, pa))) {
                            have_dnext = true;
                        } else {
                            const unsigned long long *gp = A.dltg + ((size_t)(pa % NGP_RING) * NGP_BLK + lane) * 2;
                            pg0 = ld_u64(gp);
                            pg1 = ld_u64(gp + 1);
                        }
                    }
                    // ---- ycorr -= X_a dlt_a for the rows of this wave in shard hh ----
                    if (a >= 0) {
                        const double *dq = dl + (u & 1) * NGP_DLS + 8 * c;
                        double dqv[8];
#pragma unroll
                        for (int jj = 0; jj < 8; jj++) dqv[jj] = dq[jj];
                        double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll
                        for (int jj = 0; jj < 8; jj++) {
                            p0 = __builtin_fma((double)keep[d][hh][jj].x, dqv[jj], p0);
                            p1 = __builtin_fma((double)keep[d][hh][jj].y, dqv[jj], p1);
                            p2 = __builtin_fma((double)keep[d][hh][jj].z, dqv[jj], p2);
                            p3 = __builtin_fma((double)keep[d][hh][jj].w, dqv[jj], p3);
                        }
                        p0 = p0 + dpp_f64(p0, 0); p1 = p1 + dpp_f64(p1, 0); p2 = p2 + dpp_f64(p2, 0); p3 = p3 + dpp_f64(p3, 0);
                        p0 = p0 + dpp_f64(p0, 1); p1 = p1 + dpp_f64(p1, 1); p2 = p2 + dpp_f64(p2, 1); p3 = p3 + dpp_f64(p3, 1);
                        p0 = p0 + dpp_f64(p0, 2); p1 = p1 + dpp_f64(p1, 2); p2 = p2 + dpp_f64(p2, 2); p3 = p3 + dpp_f64(p3, 2);
                        if (c == 0 && thas) {
                            double *yq = ysh + trow;
                            const double y0 = yq[0], y1 = yq[1], y2 = yq[2], y3 = yq[3];
                            yq[0] = y0 - p0; yq[1] = y1 - p1; yq[2] = y2 - p2; yq[3] = y3 - p3;
                        }
                    }
                    try_signal(false);
                    if (u < nb) {
                        // ---- GEMV chain of this wave over the tile of (block u, shard hh) ----
                        double acc = 0.0;
                        for (int k = 0; k < nqw; k++) {
                            const int q = wv + NGP_ROWS_NW * k;
                            const float4 x = *(const float4 *)(ring + (size_t)wrap(base + q) * NGP_QS + (size_t)lane * 16);
                            const double *yq = ysh + 4 * q;
                            const double y0 = yq[0], y1 = yq[1], y2 = yq[2], y3 = yq[3];
                            acc = __builtin_fma((double)x.x, y0, acc);
                            acc = __builtin_fma((double)x.y, y1, acc);
                            acc = __builtin_fma((double)x.z, y2, acc);
                            acc = __builtin_fma((double)x.w, y3, acc);
                        }
                        red[(par * NGP_ROWS_NW + wv) * NGP_BLK + lane] = acc;
                        asm volatile("" ::: "memory");  // LDS serves a wave in order: the count follows the sum
                        if (lane == 0) __hip_atomic_fetch_add((lds_int_t *)(par ? gcnt1 : gcnt0), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        try_signal(false);
                        // ---- the tile into the delay line ----
                        const char *tq = ring + (size_t)wrap(base + tslot) * NGP_QS + c * 128;
#pragma unroll
                        for (int jj = 0; jj < 8; jj++) keep[d][hh][jj] = *(const float4 *)(tq + jj * 16);
                    }
                    // publisher: waits for the seven chains through the counter in LDS, not for the barrier (see role_streamer_rows)
                    if (wv == NGP_ROWS_PUBW && u < nb) {
                        const int *gc = par ? gcnt1 : gcnt0;
                        for (unsigned sp = 0; lds_flag_ld(gc) < NGP_ROWS_NW; ++sp) {
                            if ((sp & 255u) == 255u && (lds_flag_ld(sflag) == 0 || sp > (NGP_SPIN_LIMIT << 4))) {
                                if (lane == 0) *sflag = 0;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(0);
                        }
                        asm volatile("" ::: "memory");
                        if (lane == 0) lds_flag_st(par ? gcnt0 : gcnt1, 0);  // the other parity: counted in the next sub-step, behind the barrier
                        publish(u, hh, par);
                    }
                    if (pollw) {
                        int ok = 1;
                        if (!have_dnext) {
                            ok = wait_dlt_granules_all(A.dltg, A.nonce, pa, lane, A.abort_w, 1u, pg0, pg1) ? 1 : 0;
                            if (!ok && lane == 0) *sflag = 0;
                        }
                        if (ok) dl[((u + 1) & 1) * NGP_DLS + lane] = dlt_granules_value(pg0, pg1);
                    }
                    if (wv == NGP_ROWS_POLLW && hh == V - 1 && pa + 1 >= 0 && u + 2 < nb + DT) {
                        const unsigned long long *gp = A.dltg + ((size_t)((pa + 1) % NGP_RING) * NGP_BLK + lane) * 2;
                        pg0 = ld_u64(gp);
                        pg1 = ld_u64(gp + 1);
                    }
                    // the count of a stored partial must not slip behind the barrier at lag 3 (see role_streamer_rows)
                    try_signal(DT < 4 || (A.knob & 16));
                    wg_barrier();
                    if (!*sflag) return;
                    base = wrap(base + NQ);
                }
            }
        }
        try_signal(true);
    }
    __syncthreads();
    for (int i = tid; i < V * R; i += NGP_WG) yg[i] = ys[(i / R) * RP + (i % R)];
}

// ------------------------------------------------------------------------------------------
// "reducer" g (the name of rounds 1-3, when these workgroups also added the shard partials): the FAR look-ahead corrections.  Lag
// d = near + 1 + g, + NG, .. (< D) of every block u: -G[u, u-d]' dlt_{u-d} goes into block u's accumulator as one more fixed-point
// term (ngp_common.h) -- off the sweep's latency loop: dlt_{u-d} is more than `near` blocks old when the sampler needs block u.
// Every wave works on its own blocks (u = wave, wave + 8, ...), no workgroup barrier
// (waves wv0 .. wv0 + nwv - 1 of the workgroup serve this chain's group: all eight, or four when the workgroup serves two chains)
template <bool DBG>
__device__ __attribute__((always_inline)) inline void role_reducer(const SweepArgs &A, const int g, char *smem, const int wv0 = 0,
                                                                   const int nwv = NGP_WG / 64) {
    NGP_DBG_LOCALS
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int nb = A.t1 - A.t0;
    if (A.near + 1 + g >= A.D) return;  // no far lag falls to this workgroup
    const double fxs = A.scal->fx_scale;
    for (int u = wv - wv0; u < nb; u += nwv) {
        const int slot = u % NGP_RING;
        for (int d = A.near + 1 + g; d < A.D; d += A.NG) {
            const int a = u - d;
            if (a < 0) continue;
            // (the Gram rows are requested before dlt of block u-d is awaited)
            double gr[NGP_BLK];
            load_rows_pair(A.gramx + ((size_t)(A.t0 + u) * A.D + d) * (NGP_BLK * NGP_BLK), lane, gr);
            unsigned long long q0, q1;
            {
                const unsigned long long *gp = A.dltg + ((size_t)(a % NGP_RING) * NGP_BLK + lane) * 2;
                q0 = ld_u64(gp);
                q1 = ld_u64(gp + 1);
            }
            if (!(dbg_mode == 3 || dbg_mode == 4 || dbg_mode == 6) && !wait_dlt_granules_all(A.dltg, A.nonce, a, lane, A.abort_w, 4u, q0, q1)) return;
            const double dreg = dlt_granules_value(q0, q1);  // lane k holds dlt_k
            // dlt goes through this wave's 512 bytes of LDS and comes back as broadcast reads (every lane the same 16 bytes): 32 LDS
            // instructions instead of 128 v_readlane, each of which is a 20-clock trip through an SGPR, in front of the 64 fma
            double *ldw = (double *)smem + (size_t)wv * NGP_BLK;
            ldw[lane] = dreg;
            typedef const __attribute__((address_space(3))) double *lds_cdp;
            const lds_cdp dk = (lds_cdp)ldw;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
            for (int kk = 0; kk < NGP_BLK; kk += 16) {
                double dv[16];
#pragma unroll
                for (int i = 0; i < 16; i++) dv[i] = dk[kk + i];
#pragma unroll
                for (int i = 0; i < 16; i += 4) {
                    s0 = __builtin_fma(gr[kk + i + 0], dv[i + 0], s0);
                    s1 = __builtin_fma(gr[kk + i + 1], dv[i + 1], s1);
                    s2 = __builtin_fma(gr[kk + i + 2], dv[i + 2], s2);
                    s3 = __builtin_fma(gr[kk + i + 3], dv[i + 3], s3);
                }
                asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
            }
            const double v = (s0 + s1) + (s2 + s3);
            acc_add(A.acc, A.abort_w, slot, d, lane, -v, fxs);
            if (dbg && g == 0 && lane == 0) dbg[NGP_DBG_RED + 2 * (size_t)u + 1] = wall_clock64();
        }
    }
}

// ------------------------------------------------------------------------------------------
struct CoefRegs {
    double bo, cc, ww, st, gd;
    unsigned tf;  // the block is linear: its chain is dlt = T e0 (k_tinv), T staged in place of the diagonal Gram block
};
__device__ inline CoefRegs load_coef(const SweepArgs &A, long long k) {
    CoefRegs c;
    c.bo = A.beta[k];
    c.cc = A.c[k];
    c.ww = A.w[k];
    c.st = A.q[k];  // inclusion threshold on f = c r: included iff |f| > st (st < 0: always)
    c.gd = A.mpm[k];
    c.tf = A.lin_all ? 1u : (A.blin ? A.blin[k >> 6] : 0u);
    return c;
}
// terms of local block u's accumulator: one per shard, one per far lag whose block exists
__device__ __attribute__((always_inline)) inline unsigned acc_terms(const SweepArgs &A, const int u) {
    const int nfar = max(0, min(A.D - 1, u) - A.near);
    return (unsigned)(A.S + nfar);
}
// the eight copies of a block's accumulator, added: low NGP_FX_CNT_BITS bits = terms so far, the rest = their fixed-point sum
__device__ __attribute__((always_inline)) inline unsigned long long acc_load(const SweepArgs &A, const int u, const int j) {
    const unsigned long long *ap = A.acc + (size_t)(u % NGP_RING) * NGP_FX_COPIES * NGP_BLK + j;
    unsigned long long qv[NGP_FX_COPIES];
#pragma unroll
    for (int c = 0; c < NGP_FX_COPIES; c++) qv[c] = ld_u64(ap + (size_t)c * NGP_BLK);
    unsigned long long q = qv[0];
#pragma unroll
    for (int c = 1; c < NGP_FX_COPIES; c++) q += qv[c];
    return q;
}
__device__ __attribute__((always_inline)) inline bool acc_complete(const unsigned long long q, const unsigned terms) {
    return __ballot((unsigned)(q & ((1ull << NGP_FX_CNT_BITS) - 1ull)) == terms) == ~0ull;
}
// a complete sum -> X_t'ycorr of this lane's column; the accumulator goes back to zero for the block that takes the slot 16 blocks on
// (that block's terms are more than a lag behind this store: a streamer adds them only after it has seen dlt of a LATER block)
__device__ __attribute__((always_inline)) inline double acc_take(const SweepArgs &A, const int u, const int j, const unsigned long long q, const double fxi) {
    unsigned long long *ap = A.acc + (size_t)(u % NGP_RING) * NGP_FX_COPIES * NGP_BLK + j;
#pragma unroll
    for (int c = 0; c < NGP_FX_COPIES; c++) st_u64(ap + (size_t)c * NGP_BLK, 0ull);
    return fx_to_f64((long long)q >> NGP_FX_CNT_BITS) * fxi;
}
// blocking form (whole wave polls the accumulator itself: one round trip from "last term added" to "in registers"); false on abort
template <bool DBG, bool NGBIG = false>
__device__ __attribute__((always_inline)) inline bool fetch_group_sums(const SweepArgs &A, int u, int j, double *tot_out) {
    NGP_DBG_LOCALS
    const unsigned terms = acc_terms(A, u);
    const double fxi = A.scal->fx_inv;
    for (unsigned spins = 0;; ++spins) {
        const unsigned long long q = acc_load(A, u, j);
        if (dbg_mode == 2 || acc_complete(q, terms)) {
            *tot_out = acc_take(A, u, j, q, fxi);
            return true;
        }
        if ((spins & 7u) == 7u && ld_u32(A.abort_w) != 0u) return false;
        if (spins > (NGP_SPIN_LIMIT >> 3)) {  // every turn is a memory round trip
            st_u32(A.abort_w, 3u);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// results of a finished block: beta / delta / varBeta (dlt itself left as granules when the chain ended, see role_sampler)
template <bool DBG>
__device__ __attribute__((always_inline)) inline void publish_block(const SweepArgs &A, int up, int j, const double *hist, const double *outb, const int *outi,
                                     const int *smeth, const double *ssdf) {
    NGP_DBG_LOCALS
    const int pbuf = up & 1;
    (void)hist;
    const long long k = (long long)(A.t0 + up) * NGP_BLK + j;
    // what the variance bookkeeping of BayesB needs is requested first, so that it travels while the store is acknowledged
    // (three dependent round trips here made the publisher the slowest wave of BayesB sweeps)
    const int si = A.setof[k];
    const double chik = A.chi[k];
    const int vbi = A.vbidx[k];
    // plain copy + flag for the phase streamers (they poll ONE word and then read 512 bytes; polling the granules from their
    // phase C was built and measured: 10k x 100k 1.90 -> 3.43 us per block)
    st_f64(&A.dlt[(size_t)(up % NGP_RING) * NGP_BLK + j], hist[(up % NGP_RING) * NGP_BLK + j]);
    drain_vm();
    if (j == 0) st_u32(A.flag_dlt, (unsigned)(up + 1));
    if (dbg && j == 0) dbg[4 * (size_t)up + 2] = wall_clock64();
    const double bn = outb[pbuf * NGP_BLK + j];
    const int isave = outi[pbuf * NGP_BLK + j];
    A.beta[k] = bn;
    A.delta[k] = (uint8_t)isave;
    const int meth = (si >= 0) ? smeth[si] : -1;  // per-set constants live in LDS
    if (meth == 1) {
        double vb = 0.0;
        if (isave) {
            double tt = ssdf[si];
            double b2 = bn * bn;
            tt = tt + b2;
            vb = tt / chik;
            atomicAdd(&A.sets[si].nloci, 1);
        }
        A.varBeta[vbi] = vb;
    } else if (meth == 2) {
        if (isave) atomicAdd(&A.sets[si].nloci, 1);  // BayesC: one variance per set, drawn after the sweep
    } else if (meth == 3) {
        atomicAdd(&A.sets[si].ncls[isave - 1], 1);   // BayesR: loci per class (nLoci, src/functions.jl:263); isave = class, from 1
    }
}

// sampler (8 waves), one raw barrier per block:
//   wave 0   serial chain of block u (LDS + ALU only; coefficients prefetched one block ahead); sparse for BayesB / BayesC
//   wave 1   publishes block u-1 (dlt -> streamers, beta/delta/varBeta)
//   wave 2   fetches group sums up to two blocks ahead into the r0 ring
//   wave 3   LDS-DMA of the diagonal Gram block of block u+2 (3 slots, counted vmcnt)
//   wave 5   lag-1 correction (cross block in registers, loaded one block ahead), final total, LDS flag for wave 0
//   wave 4, 6, 7   lag-2, lag-3 and (near = 4) lag-4 corrections, Gram rows loaded one block ahead into registers
// LDS: Gd[3][4096] | hist[RING][64] | vacc[RING][64] | r0[4][64] | outb[2][64] | outi[2][64] | flags, per-set constants
// NGBIG: more than 8 groups of shards may exist (k_sweep_tall: several shards per streamer workgroup, lags 2-3)
// TUP / RCLS: the model may hold Tuple (correlated BayesPR) / BayesR sets.  The production kernel of every other model -- BayesPR, BayesB,
// BayesC: the methods of the benchmark configurations -- is compiled WITHOUT their chains (k_sweep<false>; models with such a set run
// k_sweep_tup, the full kernel): the four unrolled chains of the tuple path in the critical wave's loop cost the Symbol methods
// 3-7 % at 10k x 100k although none of it executes there (code layout / register allocation of a 240-VGPR kernel; measured against
// the previous library on the same box).
// RCLS: 0 = no BayesR chain, 1 = BayesR with the lane coefficients fetched inside the block, 2 = fetched one block ahead through LDS by
// wave 1 like the Tuple ones (10.4 -> 8.9 us per block; a flavour of its own, k_sweep_r: inside the full kernel this code cost the
// 50k x 600k sweep of the other methods 2 %)
// Class search of ONE BayesR locus with nine to sixteen classes, its classes spread over lanes 0 .. K-1 (role_sampler's lazy search,
// RCLS == 2): lanes 8 .. K-1 read their class from the coefficient arrays in memory -- the sampler's LDS holds eight -- and the
// steps of eval_rform run as loops over the classes.  Returns the class; dmine = this lane's dlt should its class be the one.
struct RWide { double dmine; int c; };
__device__ __attribute__((noinline)) RWide rform_search_wide(const double *rcls, const size_t Ppad, const size_t col, const int j, const int Kk,
                                                             const double qv, const double av, const double tv, const double uv,
                                                             const double hs_k, const double rhs_k, const double bo_k) {
    const bool on = j < Kk;
    double q2 = qv, a2 = av, t2 = tv, u2 = uv;
    if (j >= NGP_RLDS && on) {
        const double *gp = rcls + (size_t)j * Ppad + col;
        const size_t gstr = (size_t)NGP_RMAX * Ppad;
        q2 = gp[0]; a2 = gp[gstr]; t2 = gp[2 * gstr]; u2 = gp[3 * gstr];
    }
    const double Lv = (q2 == 0.0) ? a2 : __builtin_fma(hs_k, q2, a2);
    RWide out;
    out.dmine = (q2 != 0.0) ? __builtin_fma(rhs_k, q2, t2) - bo_k : -bo_k;
    double m = readlane_d(Lv, 0);
#pragma unroll 1
    for (int v = 1; v < Kk; v++) {
        const double Lu = readlane_d(Lv, v);
        m = (Lu > m) ? Lu : m;
    }
    const double ev = on ? det_exp(Lv - m) : 0.0;
    double S = 0.0;
#pragma unroll 1
    for (int v = 0; v < Kk; v++) S = S + readlane_d(ev, v);
    int c = Kk - 1;
    double cum = 0.0;
#pragma unroll 1
    for (int v = 0; v < Kk; v++) {
        cum = cum + readlane_d(ev, v);
        const double thr = readlane_d(u2, v) * S;
        if (cum >= thr) { c = v; break; }
    }
    out.c = c;
    return out;
}

template <bool DBG, bool NGBIG = false, bool TUP = true, int RCLS = 1>
__device__ __attribute__((always_inline)) inline void role_sampler(const SweepArgs &A, char *smem) {
    NGP_DBG_LOCALS
    const int D = A.D, tid = threadIdx.x, wv = tid >> 6, j = tid & 63;
    double *Gd = (double *)smem;                // 3 x 4096: diagonal Gram blocks of local blocks u, u+1, u+2
    double *hist = Gd + 3 * 4096;               // RING x 64
    double *vacc = hist + NGP_RING * NGP_BLK;   // RING x 64
    double *r0 = vacc + NGP_RING * NGP_BLK;     // 4 x 64 (ring over local blocks)
    double *outb = r0 + 4 * NGP_BLK;            // 2 x 64
    int *outi = (int *)(outb + 2 * NGP_BLK);    // 2 x 64
    int *sabort = outi + 2 * NGP_BLK;
    int *totflag = sabort + 1;  // local block index + 1 whose corrected total is ready in r0[buf]
    int *smeth = sabort + 4;                     // method of each of the (at most 16) marker sets
    double *ssdf = (double *)(sabort + 20);      // scale * df of each set
    const int nb = A.t1 - A.t0;
    const size_t bsz = NGP_BLK * NGP_BLK;
    int *sK = sabort + 52;  // classes of each set (BayesR), behind ssdf
    // Tuple sets: lane coefficients of the block (rows of C and of X_l'X_l, W: 9 x 64) and (k, used lanes), two parities -- wave 1
    // fetches them one block ahead, like the critical wave does its own Symbol coefficients (fetched inside the block, behind the
    // set lookup and the set's constants, they were three dependent round trips of the chain: 9.6 us per block)
    double *tl = (double *)(sabort + 80);        // 2 x 9 x 64
    int *tmeta = (int *)(tl + 2 * 9 * NGP_BLK);  // 2 x 4: k (0: not a tuple block), used lanes
    auto tuple_prefetch = [&](const int ub) __attribute__((always_inline)) {  // wave 1; ub: local block
        if (!TUP || !A.tup || ub >= nb) return;
        const long long tblk = (long long)(A.t0 + ub);
        const long long kcol = tblk * NGP_BLK + j;
        const int si = A.setof[kcol];
        const unsigned long long mask = __ballot(si >= 0 && smeth[si < 0 ? 0 : si] == NGP_METHOD_TUPLE_DEV);
        int *meta = tmeta + (ub & 1) * 4;
        if (mask == 0ull) {
            if (j == 0) meta[0] = 0;
            return;
        }
        const int sit = __builtin_amdgcn_readfirstlane(__shfl(si, __builtin_ctzll(mask)));
        const TupLane TL = load_tuplane(A.tupc, A.tupg, A.w, A.Ppad, kcol);
        const int k = A.tup[sit].k;
        const long long first_locus = (tblk - (A.tup[sit].col0 >> 6)) * tuple_loci_per_block(k);
        const int nvalid = tuple_nvalid(k, A.tup[sit].nloc, first_locus);
        double *dst = tl + (size_t)(ub & 1) * (9 * NGP_BLK) + j;
#pragma unroll
        for (int b = 0; b < NGP_KMAX; b++) { dst[b * NGP_BLK] = TL.crow[b]; dst[(NGP_KMAX + b) * NGP_BLK] = TL.grow[b]; }
        dst[2 * NGP_KMAX * NGP_BLK] = TL.ww;
        if (j == 0) { meta[0] = k; meta[1] = nvalid; }
    };
    // BayesR sets (RCLS == 2): the lane's method and class count, its class coefficients of the first four classes and M.rhs
    // (17 x 64), the classes 5..8 of the lanes that have them (16 x 64), two parities -- the same one-block-ahead fetch by wave 1
    double *rl = (double *)(tmeta + 16);               // 2 x 17 x 64
    int *rlm = (int *)(rl + 2 * 17 * NGP_BLK);         // 2 x (64 methods | 64 class counts)
    int *rmeta = rlm + 2 * 128;                        // 2 x 4: the block holds a BayesR locus
    double *rlx = (double *)(rmeta + 16);              // 2 x (4 arrays x 4 classes x 64)
    auto rcls_prefetch = [&](const int ub) __attribute__((always_inline)) {  // wave 1; ub: local block
        if (RCLS != 2 || !A.rcls || ub >= nb) return;
        const long long kcol = (long long)(A.t0 + ub) * NGP_BLK + j;
        const int si = A.setof[kcol];
        const int meth = (si >= 0) ? smeth[si] : -1;
        const int Kc = (si >= 0) ? sK[si] : 2;
        const unsigned long long mask = __ballot(meth == 3);
        if (j == 0) rmeta[(ub & 1) * 4] = (mask != 0ull) ? 1 : 0;
        if (mask == 0ull) return;
        RLane RL = empty_rlane();
        if (meth == 3) RL = load_rlane(A.rcls, A.Ppad, kcol, Kc, A.rhs0);
        double *dst = rl + (size_t)(ub & 1) * (17 * NGP_BLK) + j;
#pragma unroll
        for (int v = 0; v < NGP_RREG; v++) {
            dst[v * NGP_BLK] = RL.q[v]; dst[(4 + v) * NGP_BLK] = RL.a[v]; dst[(8 + v) * NGP_BLK] = RL.t[v]; dst[(12 + v) * NGP_BLK] = RL.u[v];
        }
        dst[16 * NGP_BLK] = RL.rhs0;
        rlm[(ub & 1) * 128 + j] = meth;
        rlm[(ub & 1) * 128 + 64 + j] = Kc;
        if (__ballot(meth == 3 && Kc > NGP_RREG) != 0ull) {  // classes 5..8: array arr of class v at rlx[(arr * 4 + v - 4) * 64 + lane]
            double *dx = rlx + (size_t)(ub & 1) * (16 * NGP_BLK) + j;
#pragma unroll
            for (int arr = 0; arr < 4; arr++)
#pragma unroll
                for (int v = NGP_RREG; v < NGP_RLDS; v++) {
                    const bool on = (meth == 3) && v < Kc;
                    dx[(arr * 4 + v - NGP_RREG) * NGP_BLK] = on ? A.rcls[((size_t)arr * NGP_RMAX + v) * (size_t)A.Ppad + (size_t)kcol] : 0.0;
                }
        }
    };
    if (tid < 16) {
        smeth[tid] = A.sets[tid].method;
        ssdf[tid] = A.sets[tid].sdf;
        sK[tid] = A.sets[tid].K;
    }
    if (tid == 0) {
        *sabort = 0;
        *totflag = 0;
        st_u32(A.xcc_w, xcc_id() + 1u);
        if (dbg) { dbg[NGP_DBG_ALL - 2] = xcc_id(); dbg[NGP_DBG_ALL - 1] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4); }
    }
    // prologue: diagonal Gram block of local block 0 (all waves) and its group sums (wave 2)
    {
        const double *gd = chain_block_src(A, A.t0, A.lin_all || (A.blin && A.blin[A.t0] != 0u));
        for (int idx = tid; idx < 4096; idx += NGP_WG) Gd[idx] = gd[idx];
        if (nb > 1) {
            const double *gd1 = chain_block_src(A, A.t0 + 1, A.lin_all || (A.blin && A.blin[A.t0 + 1] != 0u));
            for (int idx = tid; idx < 4096; idx += NGP_WG) Gd[4096 + idx] = gd1[idx];
        }
    }
    __syncthreads();
    if (wv == 2) {
        double tot;
        if (fetch_group_sums<DBG, NGBIG>(A, 0, j, &tot)) r0[j] = tot;
        else if (j == 0) *sabort = 1;
    }
    if (wv == 1) { tuple_prefetch(0); rcls_prefetch(0); }
    __syncthreads();
    if (*sabort) return;
#define NGP_END_OF_BLOCK()                                                                       \
    do {                                                                                         \
        if (dbg && j == 0) dbg[NGP_DBG_WAVES + 8 * (size_t)u + wv] = wall_clock64();           \
        wg_barrier();                                                                            \
        if (*sabort) return;                                                                     \
    } while (0)
    // every role runs its own block loop (own register budget); all meet at ONE raw barrier per block
    if (wv == 0) {
        // ---------------- critical wave: LDS + ALU only ----------------
        CoefRegs cur = load_coef(A, (long long)A.t0 * NGP_BLK + j), nxt = cur;
        double iVarE_sweep = 0.0;  // (RCLS == 2: 1 / varE of this iteration, constant during the sweep)
        if constexpr (RCLS == 2) {
            if (A.rcls) iVarE_sweep = A.scal->iVarE;
        }
        for (int u = 0; u < nb; ++u) {
            const int t = A.t0 + u, buf = u & 1, slot = u % NGP_RING, rs = u & 3;
            if (dbg && j == 0) dbg[4 * (size_t)u] = wall_clock64();
            if (u + 1 < nb) nxt = load_coef(A, (long long)(t + 1) * NGP_BLK + j);
            // Linear block (k_tinv): its chain is dlt = T e0, T staged in place of the diagonal Gram block.  (Row j of T is read chunk by
            // chunk inside the product, behind the total: loaded into registers ahead of the wait -- right after the block's barrier --
            // its 64 LDS reads compete with the lag-1 product of wave 5, which is what the chain waits for: measured slower.)
            constexpr bool PRELOAD = false;
            const bool tform = __builtin_amdgcn_readfirstlane((int)cur.tf) != 0;
            const double *gdb = Gd + (u % 3) * 4096 + j;
            double G[NGP_BLK];
            double tot;
            if (D == 1) {  // lag 1: nothing can be fetched or corrected ahead
                tot = r0[rs * NGP_BLK + j];
                bool okc = true;
                if (u >= 1) okc = fetch_group_sums<DBG, NGBIG>(A, u, j, &tot);
                if (!okc && j == 0) *sabort = 1;
            } else {       // wave 5 applies the look-ahead corrections and leaves the final total in r0[buf]
                // bounded like every other spin: an abort raised by another wave of this workgroup ends the wait
                for (unsigned sp = 0; lds_flag_ld(totflag) != u + 1; ++sp) {
                    if ((sp & 255u) == 255u && (lds_flag_ld(sabort) != 0 || sp > (NGP_SPIN_LIMIT << 4))) {
                        if (j == 0) *sabort = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(0);
                }
                asm volatile("" ::: "memory");  // r0 is read after the flag (LDS serves a wave in order)
                tot = r0[rs * NGP_BLK + j];
            }
            if (DBG && dbg && j == 0) dbg[NGP_DBG_W5 + 4 * (size_t)u + 2] = wall_clock64();  // the chain wave has its total
            const double bo = cur.bo, cc = cur.cc, ww = cur.ww, st = cur.st;
            const double r = __builtin_fma(cur.gd, bo, tot);
            // scaled recursion: e = c r + w is the candidate draw, f = c r feeds the
            // inclusion test |f| > thr.  The stored diagonal block is zero for lanes <= k, so a lane's e and f freeze at
            // its own step and nothing has to be captured.  H_k = -(c G[k][.]) is formed four steps ahead, in the
            // latency shadow of the serial path: v_readlane -> ONE fma per step (BayesPR), + compare / select (BayesB).
#define NGP_LOAD_G()                                                           \
    if constexpr (!PRELOAD) { _Pragma("unroll") for (int kk = 0; kk < NGP_BLK; kk++) G[kk] = gdb[kk * NGP_BLK]; } \
    double H0 = -(cc * G[0]), H1 = -(cc * G[1]), H2 = -(cc * G[2]), H3 = -(cc * G[3]);
            double e = __builtin_fma(r, cc, ww);
            double dsave;
            int isave = 1;
            // BayesR: the lane's set and method (one byte per lane, read only when the model has a BayesR set at all)
            int meth0 = -1, si0 = -1, rblk = 0;
            if constexpr (RCLS == 1) {
                if (A.rcls) {
                    si0 = A.setof[(long long)t * NGP_BLK + j];
                    meth0 = (si0 >= 0) ? smeth[si0] : -1;
                }
            } else if constexpr (RCLS == 2) {
                if (A.rcls) rblk = __builtin_amdgcn_readfirstlane(rmeta[buf * 4]);
                if (rblk) meth0 = rlm[buf * 128 + j];
            }
            int tk = 0;
            if constexpr (TUP) {
                if (A.tup) tk = __builtin_amdgcn_readfirstlane(tmeta[buf * 4]);
            }
            if (tform && dbg_mode != 5) {
                // dlt = T e0: e0 goes through the r0 slot of this block (free once the total has been
                // read) and comes back as broadcast reads, 16 values at a time; four accumulators over i mod 4, ((s0+s1)+(s2+s3))
                double e0 = e;
                if constexpr (TUP) {
                    if (tk != 0) {  // a Tuple block: e0 from the k x k conditional of the lane's locus (lane coefficients: LDS, wave 1)
                        const int nvalid = __builtin_amdgcn_readfirstlane(tmeta[buf * 4 + 1]);
                        const double *tsrc = tl + (size_t)buf * (9 * NGP_BLK) + j;
                        TupLane TL;
#pragma unroll
                        for (int b = 0; b < NGP_KMAX; b++) { TL.crow[b] = tsrc[b * NGP_BLK]; TL.grow[b] = tsrc[(NGP_KMAX + b) * NGP_BLK]; }
                        TL.ww = tsrc[2 * NGP_KMAX * NGP_BLK];
                        e0 = tuple_e0(tk, nvalid, j, tot, bo, TL);
                    }
                }
                r0[rs * NGP_BLK + j] = e0;
                typedef const __attribute__((address_space(3))) double *lds_cdp;
                const lds_cdp ek = (lds_cdp)(r0 + rs * NGP_BLK);
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
                for (int kk = 0; kk < NGP_BLK; kk += 16) {
                    double ev[16], tv[16];
#pragma unroll
                    for (int i = 0; i < 16; i++) ev[i] = ek[kk + i];
#pragma unroll
                    for (int i = 0; i < 16; i++) tv[i] = PRELOAD ? G[kk + i] : gdb[(kk + i) * NGP_BLK];
#pragma unroll
                    for (int i = 0; i < 16; i += 4) {
                        s0 = __builtin_fma(tv[i + 0], ev[i + 0], s0);
                        s1 = __builtin_fma(tv[i + 1], ev[i + 1], s1);
                        s2 = __builtin_fma(tv[i + 2], ev[i + 2], s2);
                        s3 = __builtin_fma(tv[i + 3], ev[i + 3], s3);
                    }
                    asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
                }
                dsave = (s0 + s1) + (s2 + s3);
                isave = 1;
            } else if (TUP && tk != 0) {
                // a block of a Tuple set: one step per locus, its k effects drawn together (tuple_chain, ngp_common.h); the lane
                // coefficients wait in LDS (wave 1, one block ahead)
                const int nvalid = __builtin_amdgcn_readfirstlane(tmeta[buf * 4 + 1]);
                const double *tsrc = tl + (size_t)buf * (9 * NGP_BLK) + j;
                TupLane TL;
#pragma unroll
                for (int b = 0; b < NGP_KMAX; b++) { TL.crow[b] = tsrc[b * NGP_BLK]; TL.grow[b] = tsrc[(NGP_KMAX + b) * NGP_BLK]; }
                TL.ww = tsrc[2 * NGP_KMAX * NGP_BLK];
                dsave = tuple_chain_nv(tk, nvalid, j, tot, bo, TL, [&](int sl, int cc2) { return Gd[(u % 3) * 4096 + sl * NGP_BLK + cc2]; });
                isave = 1;
            } else if (RCLS != 0 && A.rcls && (RCLS == 2 ? rblk != 0 : __ballot(meth0 == 3) != 0ull)) {
                // r-form chain (eval_rform, ngp_kernels.h): candidates of all lanes from the current r, the first non-zero one at
                // or behind the cursor takes its step.  With most loci in the zero class that is a few steps per block.
                double dfin = 0.0;
                int cfin = 1;
                if constexpr (RCLS == 2) {
                    // LAZY class search: a pass over the lanes costs a compare, not an evaluation -- a BayesR
                    // locus without an old effect whose hs = rhs^2 / 2 is at or below its threshold h* (k_prep) is certain to stay in
                    // the zero class and is passed over; the first lane at or behind the cursor that is NOT certain is evaluated
                    // ALONE, its classes spread over lanes 0 .. K-1 (one exponential deep instead of four wide), with the operations
                    // of eval_rform in their order: the same class, the same dlt, bit for bit.
                    const double *cq = rl + (size_t)buf * (17 * NGP_BLK);   // array arr of class v < 4 of lane l: cq[(4 arr + v) 64 + l]
                    const double *cx = rlx + (size_t)buf * (16 * NGP_BLK);  // classes 4..7: cx[(4 arr + v - 4) 64 + l]
                    const bool isr = (meth0 == 3);
                    const double q0j = cq[j], hstar = (isr && q0j == 0.0) ? cq[8 * NGP_BLK + j] : -1.0;
                    const double rhs0j = cq[16 * NGP_BLK + j];
                    const int Kj = rlm[buf * 128 + 64 + j];
                    const double iVarE = iVarE_sweep;
                    const int vj = (j < NGP_RLDS) ? j : 0;  // the class this lane holds when a locus is evaluated alone (classes 9..16: from memory, below)
                    const double *cls_base = (vj < NGP_RREG) ? cq + vj * NGP_BLK : cx + (vj - NGP_RREG) * NGP_BLK;
                    double rcur = r;
                    int kstart = 0;
                    for (int guard = 0; guard < NGP_BLK + 1; ++guard) {
                        double lazy_d, rhs = 0.0, hs = 0.0;
                        int lazy_c;
                        bool need;
                        if (isr) {
                            const double tt = rcur * iVarE;
                            rhs = tt + rhs0j;
                            const double s2 = rhs * rhs;
                            hs = 0.5 * s2;
                            need = (bo != 0.0) || !(hs <= hstar);
                            lazy_d = -bo;
                            lazy_c = 1;
                        } else {
                            eval_rform_other(rcur, bo, cc, ww, st, lazy_d, lazy_c);
                            need = (lazy_d != 0.0);
                        }
                        if (j >= kstart) { dfin = lazy_d; cfin = lazy_c; }
                        const unsigned long long todo = __ballot(need) & (~0ull << kstart);
                        if (!todo) break;
                        const int kk = __builtin_ctzll(todo);
                        const double Hk = -(gdb[kk * NGP_BLK]);  // row kk of the one-sided block (0 for lanes <= kk): asked for now, used after the evaluation
                        double dk;
                        if (__builtin_amdgcn_readlane(meth0, kk) == 3) {
                            const double rhs_k = readlane_d(rhs, kk), hs_k = readlane_d(hs, kk), bo_k = readlane_d(bo, kk);
                            const int Kk = __builtin_amdgcn_readlane(Kj, kk);
                            // lane v < K: class v of locus kk
                            const bool on = j < Kk;
                            const double *cp = cls_base + kk;  // (lanes beyond the locus' classes read what is there and do not use it)
                            const double qv = cp[0], av = cp[4 * NGP_BLK], tv = cp[8 * NGP_BLK], uv = cp[12 * NGP_BLK];
                            const double Lraw = (qv == 0.0) ? av : __builtin_fma(hs_k, qv, av);
                            // every lane's candidate should ITS class be the one (in the shadow of the exponential): dlt = rhs / lhs_c + sd_c z - beta
                            const double dcls = __builtin_fma(rhs_k, qv, tv);
                            double dmine = (qv != 0.0) ? dcls - bo_k : -bo_k;
                            int c;
                            if (Kk <= NGP_RREG) {
                                // up to four classes (the usual BayesR): no branch -- a lane beyond the set's classes carries log-weight -inf
                                // and comparison uniform +inf, as the padded register classes of eval_rform do
                                const double Lv = on ? Lraw : -__builtin_inf();
                                const double up = on ? uv : __builtin_inf();
                                // (lanes 0..3 are one DPP quad: the maximum by two exchanges -- exact, so the order does not matter --, the
                                // ordered sums from four quad broadcasts; the other quads of the wave compute along on nothing)
                                double m = __builtin_fmax(Lv, dpp_quad_f64<0xB1>(Lv));
                                m = __builtin_fmax(m, dpp_quad_f64<0x4E>(m));
                                const double ev = det_exp_le0(Lv - m);
                                const double c0 = dpp_quad_f64<0x00>(ev);  // 0.0 + e_0
                                const double c1 = c0 + dpp_quad_f64<0x55>(ev);
                                const double c2 = c1 + dpp_quad_f64<0xAA>(ev);
                                const double c3 = c2 + dpp_quad_f64<0xFF>(ev);  // = S
                                const int vq = j & 3;
                                const double mycum = (vq == 0) ? c0 : ((vq == 1) ? c1 : ((vq == 2) ? c2 : c3));
                                const double thr = up * c3;
                                const unsigned hitm = (unsigned)__ballot(mycum >= thr) & 0xFu;  // class v stops the search: cum_v >= u_v S
                                c = hitm ? __builtin_ctz(hitm) : Kk - 1;
                            } else if (Kk <= NGP_RLDS) {
                                const double Lv = Lraw;
                                // m = L_0, then (L_v > m) ? L_v : m in class order (eval_rform), on the gathered values
                                double m = readlane_d(Lv, 0);
#pragma unroll
                                for (int v = 1; v < NGP_RLDS; v++) {
                                    if (v < Kk) {
                                        const double Lu = readlane_d(Lv, v);
                                        m = (Lu > m) ? Lu : m;
                                    }
                                }
                                const double ev = on ? det_exp(Lv - m) : 0.0;
                                // S = ((e_0 + e_1) + e_2) + ..., then the comparisons cum_v >= u_v S in class order
                                double S = 0.0;
#pragma unroll
                                for (int v = 0; v < NGP_RLDS; v++)
                                    if (v < Kk) S = S + readlane_d(ev, v);
                                c = Kk - 1;
                                double cum = 0.0;
                                bool found = false;
#pragma unroll
                                for (int v = 0; v < NGP_RLDS; v++) {
                                    if (v < Kk && !found) {
                                        cum = cum + readlane_d(ev, v);
                                        const double thr = readlane_d(uv, v) * S;
                                        if (cum >= thr) { c = v; found = true; }
                                    }
                                }
                            } else {
                                // nine to sixteen classes (rare): a function of its own, out of the way of the two paths above
                                const RWide rw = rform_search_wide(A.rcls, (size_t)A.Ppad, (size_t)((long long)t * NGP_BLK + kk), j, Kk, qv, av, tv, uv, hs_k, rhs_k, bo_k);
                                c = rw.c;
                                dmine = rw.dmine;
                            }
                            dk = readlane_d(dmine, c);
                            if (j == kk) { dfin = dk; cfin = c + 1; }
                        } else {
                            dk = readlane_d(lazy_d, kk);
                        }
                        if (dk != 0.0) rcur = __builtin_fma(Hk, dk, rcur);
                        kstart = kk + 1;
                        if (kstart >= NGP_BLK) break;
                    }
                } else {
                RLane RL = empty_rlane();
                double iVarE;
                {
                    if (meth0 == 3) RL = load_rlane(A.rcls, A.Ppad, (long long)t * NGP_BLK + j, sK[si0], A.rhs0);
                    iVarE = A.scal->iVarE;
                }
                double rcur = r;
                int kstart = 0;
                for (int guard = 0; guard < NGP_BLK + 1; ++guard) {
                    double cand;
                    int cls;
                    eval_rform(meth0, rcur, bo, cc, ww, st, RL, iVarE, cand, cls);
                    if (j >= kstart) { dfin = cand; cfin = cls; }
                    const unsigned long long todo = __ballot(cand != 0.0) & (~0ull << kstart);
                    if (!todo) break;
                    const int kk = __builtin_ctzll(todo);
                    const double dk = readlane_d(cand, kk);
                    const double Hk = -(gdb[kk * NGP_BLK]);  // row kk of the one-sided block: 0 for lanes <= kk
                    rcur = __builtin_fma(Hk, dk, rcur);
                    kstart = kk + 1;
                    if (kstart >= NGP_BLK) break;
                }
                }
                dsave = dfin;
                isave = cfin;
            } else if (dbg_mode == 5 && __ballot(st >= 0.0) == 0ull) {
                dsave = e;  // timing experiment: BayesPR blocks without the 64-step recursion
            } else if (__ballot(st >= 0.0) == 0ull) {
                NGP_LOAD_G()
                if (DBG && dbg) {  // diagnostic kernel: the diagonal block is in the registers
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (j == 0) dbg[NGP_DBG_W5 + 4 * (size_t)u + 3] = wall_clock64();
                }
#pragma unroll
                for (int kk = 0; kk < NGP_BLK; kk += 4) {
                    double dk;
                    dk = readlane_d(e, kk + 0); e = __builtin_fma(H0, dk, e); H0 = -(cc * G[(kk + 4) & 63]);
                    dk = readlane_d(e, kk + 1); e = __builtin_fma(H1, dk, e); H1 = -(cc * G[(kk + 5) & 63]);
                    dk = readlane_d(e, kk + 2); e = __builtin_fma(H2, dk, e); H2 = -(cc * G[(kk + 6) & 63]);
                    dk = readlane_d(e, kk + 3); e = __builtin_fma(H3, dk, e); H3 = -(cc * G[(kk + 7) & 63]);
                }
                dsave = e;
            } else if (__popcll(__ballot((__builtin_fabs(r * cc) > st) || (bo != 0.0))) <= NGP_SPARSE_MAX) {
                // Sparse BayesB / BayesC block: a step k changes something only if locus k is included (|f_k| > thr_k) or
                // carries an old effect to take out (beta_k != 0); every other step multiplies by dlt_k = -0 and leaves all e, f
                // as they are, bit for bit.  So the chain visits only the lanes of that mask, which is rebuilt after every
                // visited step (f has changed for the later lanes).  With pi around 1 % that is a handful of steps, not 64.
                double f = r * cc;
                const double nbo = -bo;
                unsigned long long todo = __ballot((__builtin_fabs(f) > st) || (bo != 0.0));
                while (todo) {
                    const int k = __builtin_ctzll(todo);
                    const double Hk = -(cc * gdb[k * NGP_BLK]);  // row k of the one-sided block: 0 for lanes <= k
                    const unsigned long long inm = __ballot(__builtin_fabs(f) > st);
                    const double ek = readlane_d(e, k), nk = readlane_d(nbo, k);
                    const double dk = ((inm >> k) & 1ull) ? ek : nk;
                    e = __builtin_fma(Hk, dk, e);
                    f = __builtin_fma(Hk, dk, f);
                    const unsigned long long later = (k == 63) ? 0ull : (~0ull << (k + 1));
                    todo = __ballot((__builtin_fabs(f) > st) || (bo != 0.0)) & later;
                }
                isave = __builtin_fabs(f) > st;
                dsave = isave ? e : -bo;
            } else {
                NGP_LOAD_G()
                double f = r * cc;
#pragma unroll
                for (int kk = 0; kk < NGP_BLK; kk += 4) {
#define NGP_STEP(HX, KO)                                              \
    {                                                                 \
        int in = __builtin_fabs(f) > st;                              \
        double dlv = in ? e : -bo;                                    \
        double dk = readlane_d(dlv, kk + KO);                         \
        e = __builtin_fma(HX, dk, e);                                 \
        f = __builtin_fma(HX, dk, f);                                 \
        HX = -(cc * G[(kk + KO + 4) & 63]);                           \
    }
                    NGP_STEP(H0, 0) NGP_STEP(H1, 1) NGP_STEP(H2, 2) NGP_STEP(H3, 3)
#undef NGP_STEP
                }
                isave = __builtin_fabs(f) > st;
                dsave = isave ? e : -bo;
            }
#undef NGP_LOAD_G
            hist[slot * NGP_BLK + j] = dsave;
            // dlt leaves the chain wave at once as tagged granules (two fire-and-forget stores): streamers and reducers poll those.
            // (Handing this and the coefficient prefetch to wave 1 through an LDS flag was built and measured: wave 1's memory
            // operations queue behind the Gram blocks of five waves and it became the slowest wave -- 10k x 100k 1.71 -> 3.42 us
            // per block.)
            publish_dlt_granules(A.dltg, A.nonce, u, j, dsave);
            outb[buf * NGP_BLK + j] = bo + dsave;
            outi[buf * NGP_BLK + j] = isave;
            if (dbg && j == 0) dbg[4 * (size_t)u + 1] = wall_clock64();
            cur = nxt;
            NGP_END_OF_BLOCK();
        }
    } else if (wv == 1) {
        for (int u = 0; u < nb; ++u) {
            tuple_prefetch(u + 1);
            rcls_prefetch(u + 1);
            if (u >= 1) publish_block<DBG>(A, u - 1, j, hist, outb, outi, smeth, ssdf);
            NGP_END_OF_BLOCK();
        }
        if (nb >= 1) publish_block<DBG>(A, nb - 1, j, hist, outb, outi, smeth, ssdf);
    } else if (wv == 2) {
        // X_t'ycorr of the coming blocks -> r0 ring, from the blocks' fixed-point accumulators (ngp_common.h).  Lag >= 4: up to two blocks
        // ahead -- the eight copies of the next accumulator are requested at the end of a block and looked at at the start of the
        // following one (the load is the probe: a complete count means the sum is final), so the memory round trip of this busy CU
        // never sits on the block period.  Lags 2-3: one block ahead, blocking.
        // (A streamer with lag >= 3 polls dlt_{u+1-D} before it publishes partial u, so the sum of block u+2 exists during block u
        // only when D >= 4.)
        if (D >= 4) {
            const double fxi = A.scal->fx_inv;
            // One look per block, issued a WHOLE block period before it is examined: at the start of block u this wave first requests
            // the accumulator of the block after the one whose look is in flight (speculating that that one will turn out
            // complete -- it nearly always is: the sums stand two microseconds before they are needed), and only then examines the
            // look requested at the start of block u-1.  Requested and examined in one go -- the round 1-3 counter probe, and the first
            // version of this loop -- the wave sat one loaded memory round trip (1.9 us at 50k x 600k) in every block and was the last
            // at the barrier in four blocks of five.
            unsigned long long qa = 0ull;   // the look in flight ...
            int la = -1;                    // ... and the block it looked at (-1: none)
            int next_take = 1;              // first local block whose sum has not been taken yet
            if (nb > 1) {  // once, at the start of the sweep: block 1 by a blocking fetch, so that the loop below runs two blocks ahead
                double tot1;
                if (fetch_group_sums<DBG, NGBIG>(A, 1, j, &tot1)) { r0[NGP_BLK + j] = tot1; next_take = 2; }
                else if (j == 0) *sabort = 1;
            }
            if (nb > 2) { qa = acc_load(A, 2, j); la = 2; }
            for (int u = 0; u < nb; ++u) {
                // speculative request for the following block (examined a block from now: block ln must then be allowed in r0)
                unsigned long long qn = 0ull;
                int ln = -1;
                {
                    const int want = (la >= 0 ? la : next_take - 1) + 1;
                    if (want < nb && want <= u + 3) { qn = acc_load(A, want, j); ln = want; }
                }
                if (next_take < nb && next_take <= u + 2) {  // r0[(u+2) & 3] is free: block u-2 is done
                    const bool must = (next_take == u + 1);   // the next block needs this sum
                    const unsigned terms = acc_terms(A, next_take);
                    unsigned long long q = qa;
                    bool ready = false;
                    int ok = 1;
                    if (la == next_take) ready = dbg_mode == 2 || acc_complete(q, terms);
                    else if (must || la < 0) { q = acc_load(A, next_take, j); ready = dbg_mode == 2 || acc_complete(q, terms); }  // (start of the sweep, or after a look that came too early)
                    if (!ready && must) {
                        for (unsigned spins = 0;; ++spins) {
                            __builtin_amdgcn_s_sleep(2);
                            q = acc_load(A, next_take, j);
                            if (acc_complete(q, terms)) { ready = true; break; }
                            if ((spins & 7u) == 7u && ld_u32(A.abort_w) != 0u) { ok = 0; break; }
                            if (spins > (NGP_SPIN_LIMIT >> 3)) { st_u32(A.abort_w, 3u); ok = 0; break; }
                        }
                    }
                    if (!ok) {
                        if (j == 0) *sabort = 1;
                    } else if (ready) {
                        r0[(next_take & 3) * NGP_BLK + j] = acc_take(A, next_take, j, q, fxi);
                        if (dbg && j == 0) dbg[4 * (size_t)next_take + 3] = wall_clock64();
                        ++next_take;
                    }
                }
                // the speculative look becomes the look in flight if it is the right one for what comes next, else it is dropped
                if (ln == next_take) { qa = qn; la = ln; }
                else if (ln > next_take && ln >= 0) {  // its predecessor was not complete: look at that one again next time
                    la = -1;
                    if (next_take < nb && next_take <= u + 3) { qa = acc_load(A, next_take, j); la = next_take; }
                } else la = -1;
                NGP_END_OF_BLOCK();
            }
        } else {
            for (int u = 0; u < nb; ++u) {
                if (u + 1 < nb && D >= 2) {
                    double tot;
                    if (fetch_group_sums<DBG, NGBIG>(A, u + 1, j, &tot)) r0[((u + 1) & 3) * NGP_BLK + j] = tot;
                    else if (j == 0) *sabort = 1;
                }
                NGP_END_OF_BLOCK();
            }
        }
    } else if (wv == 3) {
        // LDS-DMA (global_load_lds_dwordx4) of the diagonal Gram block of local block u+2 into the 3-slot ring: 32 KiB
        // issued per block, and only the PREVIOUS block's 32 instructions have to be complete at the barrier
        // (counted vmcnt), so each transfer has a whole block period to land.  This wave never reads LDS (an LDS read
        // would make the compiler drain the DMA): it sees an abort through a scalar load of the abort word.
        // (is block u+2 linear: the word travels with the abort word of the previous block's end -- one scalar round trip for both,
        // outside vmcnt, which counts this wave's DMA)
        const bool per_block = !A.lin_all && A.blin != nullptr;
        unsigned lin2 = A.lin_all ? 1u : 0u;
        if (per_block && nb > 2) lin2 = sld_u32(A.blin + A.t0 + 2);
        for (int u = 0; u < nb; ++u) {
            if (u + 2 < nb) {
                const char *gsrc = (const char *)chain_block_src(A, A.t0 + u + 2, lin2 != 0u) + (size_t)j * 16;
                char *gdst = (char *)(Gd + ((u + 2) % 3) * 4096);
#pragma unroll
                for (int i = 0; i < 32; i++)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc + (size_t)i * 1024),
                                                     (__attribute__((address_space(3))) void *)(gdst + i * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            } else {
                drain_vm();
            }
            if (dbg && j == 0) dbg[NGP_DBG_WAVES + 8 * (size_t)u + wv] = wall_clock64();
            wg_barrier();
            unsigned ab;
            if (per_block && u + 3 < nb) sld_u32x2(A.abort_w, A.blin + A.t0 + u + 3, ab, lin2);
            else ab = sld_u32(A.abort_w);
            if (ab != 0u) return;
        }
    } else if (wv == 5) {
        // finishes r0 for the critical wave: total - ((lag-3 + lag-2 terms) + lag-1 term), then raises the LDS flag.
        // The lag-1 cross Gram block lives in this wave's registers, loaded one block ahead.
        double gxr[NGP_BLK];
#pragma unroll
        for (int kk = 0; kk < NGP_BLK; kk++) gxr[kk] = 0.0;
        for (int u = 0; u < nb; ++u) {
            if (D >= 2) {
                const int slot = u % NGP_RING, rs = u & 3;
                double tot = r0[rs * NGP_BLK + j];
                const bool have_far = (D >= 3) && (u >= 2) && (A.near >= 2);
                const bool have_one = (u >= 1);
                double cor = have_far ? vacc[slot * NGP_BLK + j] : 0.0;
                if (DBG && dbg) {  // diagnostic kernel: when are this block's lag-1 rows in the registers?
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (j == 0) dbg[NGP_DBG_W5 + 4 * (size_t)u] = wall_clock64();
                }
                if (have_one) {
                    // dlt of the previous block by broadcast reads from LDS (every lane reads the same 16 bytes): 32 LDS
                    // instructions instead of 128 v_readlane in front of the 64 fma -- this wave is what the chain waits for
                    typedef const __attribute__((address_space(3))) double *lds_cdp;
                    const lds_cdp dk = (lds_cdp)(hist + ((u - 1) % NGP_RING) * NGP_BLK);
                    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
                    for (int kk = 0; kk < NGP_BLK; kk += 16) {
                        double dv[16];
#pragma unroll
                        for (int i = 0; i < 16; i++) dv[i] = dk[kk + i];
#pragma unroll
                        for (int i = 0; i < 16; i += 4) {
                            s0 = __builtin_fma(gxr[kk + i + 0], dv[i + 0], s0);
                            s1 = __builtin_fma(gxr[kk + i + 1], dv[i + 1], s1);
                            s2 = __builtin_fma(gxr[kk + i + 2], dv[i + 2], s2);
                            s3 = __builtin_fma(gxr[kk + i + 3], dv[i + 3], s3);
                        }
                        // the sums are "used" here so that the 16 fma stay with their chunk (all 64 dlt values in registers
                        // next to the 64 Gram rows would not fit)
                        asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
                    }
                    const double v1 = (s0 + s1) + (s2 + s3);
                    cor = have_far ? cor + v1 : v1;
                }
                if (have_far || have_one) tot = tot - cor;
                if (DBG && dbg && j == 0) dbg[NGP_DBG_W5 + 4 * (size_t)u + 1] = wall_clock64();
                r0[rs * NGP_BLK + j] = tot;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (j == 0) lds_flag_st(totflag, u + 1);
                if (u + 1 < nb) {
                    load_rows_pair(A.gramx + ((size_t)(A.t0 + u + 1) * D + 1) * bsz, j, gxr);
                }
            }
            NGP_END_OF_BLOCK();
        }
    } else {
        // waves 4, 6 and 7: the lag-2, lag-3 and lag-4 corrections (farther lags are folded into the group sums by the
        // reducers -- a loop sampler -> reducer -> sampler of about 7 us that must fit into `near` block periods);
        // their Gram rows are loaded one block ahead.
        const int fx = (wv == 4 && A.near >= 2) ? 2 : ((wv == 6 && A.near >= 3) ? 3 : (wv == 7 && A.near >= 4 ? 4 : 99));
        const int top = min(A.near, D - 1);  // highest lag corrected here: its term opens the sum of a target
        double gr[NGP_BLK];
#pragma unroll
        for (int kk = 0; kk < NGP_BLK; kk++) gr[kk] = 0.0;
        bool have = false;
        for (int u = 0; u < nb; ++u) {
            if (fx < D) {
                if (u >= 1 && have) {  // dlt of local block a = u-1, target a + fx
                    const int a = u - 1, upb = a + fx;
                    const double *dp = hist + (a % NGP_RING) * NGP_BLK;
                    double v = gemv4([&](int kk) { return gr[kk]; }, dp);
                    double *va = vacc + (upb % NGP_RING) * NGP_BLK + j;
                    // higher lags arrive first, one block apart (all by this workgroup, separated by barriers): ((v_4 + v_3) + v_2)
                    const bool first = (fx == top) || (a == 0);
                    *va = first ? v : *va + v;
                }
                have = (u + fx < nb) && (u + 1 < nb);  // rows for the next block: a' = u, target u + fx
                if (have) {
                    load_rows_pair(A.gramx + ((size_t)(A.t0 + u + fx) * D + fx) * bsz, j, gr);
                }
            }
            NGP_END_OF_BLOCK();
        }
    }
#undef NGP_END_OF_BLOCK
}

// Census: every workgroup of this grid waits for others, so the whole grid must be resident at once.  The host checks that by
// count (occupancy query, CU lease), but what the dispatcher does with several grids of one process is not ours to know: each
// workgroup reports in, the last one opens the gate, and if the gate is still shut after 20 ms the launch ends BEFORE any role
// has touched the chain (abort code NGP_ABORT_CENSUS) -- the host then runs it again with the device to itself.
__device__ __attribute__((always_inline)) inline bool sweep_census(const SweepArgs &A, const int b, char *smem) {
    if (!A.census) return true;
    int *cflag = (int *)smem;
    if (threadIdx.x == 0) {
        A.census_tbl[b] = ((unsigned long long)(xcc_id() + 1u) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        if (A.census_fail != 0u && A.census_fail == A.iter_tag) atomicCAS(&A.census[1], 0u, 2u);  // test hook
        const unsigned old = atomicAdd(&A.census[0], 1u);
        if (old + 1u == gridDim.x) atomicCAS(&A.census[1], 0u, 1u);
        const unsigned long long t0 = wall_clock64();
        unsigned verdict;
        while ((verdict = ld_u32(&A.census[1])) == 0u) {
            if (wall_clock64() - t0 > NGP_CENSUS_TICKS) atomicCAS(&A.census[1], 0u, 2u);
            else __builtin_amdgcn_s_sleep(8);
        }
        if (verdict != 1u) {
            st_u32(A.abort_w + 1, A.iter_tag);
            st_u32(A.abort_w, NGP_ABORT_CENSUS);
        }
        *cflag = (verdict == 1u) ? 1 : 0;
    }
    __syncthreads();
    const int cok = *cflag;
    __syncthreads();  // the roles overwrite this word
    return cok != 0;
}

// ------------------------------------------------------------------------------------------
// k_sweep<false>: the production kernel of models without a Tuple set; k_sweep<true>: the diagnostic kernel (time stamps, timing
// modes; every method); k_sweep_tup: the production kernel of models with a Tuple set (third translation unit)
template <bool DBG>
__global__ __launch_bounds__(NGP_WG) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_sweep(SweepArgs A) {
    constexpr bool TUP = DBG;
    constexpr int RCLSV = DBG ? 1 : 0;
#include "ngp_sweep_body.inc"
}
#if defined(NGP_INST_DBG) && NGP_INST_DBG == 2
__global__ __launch_bounds__(NGP_WG) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_sweep_tup(SweepArgs A) {
    constexpr bool DBG = false, TUP = true;
    constexpr int RCLSV = 1;
#include "ngp_sweep_body.inc"
}
#endif
#if defined(NGP_INST_DBG) && NGP_INST_DBG == 3  // models with a BayesR set: its coefficients fetched one block ahead (fourth translation unit)
__global__ __launch_bounds__(NGP_WG) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_sweep_r(SweepArgs A) {
    constexpr bool DBG = false, TUP = true;
    constexpr int RCLSV = 2;
#include "ngp_sweep_body.inc"
}
#endif

#if defined(NGP_INST_DBG) && NGP_INST_DBG == 1  // one definition: the second translation unit (it is the shorter one to compile)
// The persistent sweep of fp32 panels too tall for one shard per workgroup (role_streamer_rows_tall; host: V = 2 at lag 3, V = 3 at
// lag 2, S a multiple of V, grid 1 + NG + S / V).  A kernel of its own so that k_sweep stays what it is: the sampler here adds up to
// 22 group sums (NGBIG), and the extra code of both cost k_sweep 7 % at 10k x 100k when they were branches of it (SGPR pressure).
__global__ __launch_bounds__(NGP_WG) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_sweep_tall(SweepArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    if (ld_u32(A.abort_w) != 0u) return;
    if (!sweep_census(A, b, smem)) return;
    if (b == 0)
        role_sampler<false, true>(A, smem);
    else if (b <= A.NG)
        role_reducer<false>(A, b - 1, smem);
    else if (A.V == 2)
        role_streamer_rows_tall<3, 2>(A, b - 1 - A.NG, smem);
    else
        role_streamer_rows_tall<2, 3>(A, b - 1 - A.NG, smem);
}
#endif

// ------------------------------------------------------------------------------------------
// K CHAINS PER PASS OVER THE PANEL (hard part 2).  Independent chains (one handle each, same panel, same
// layout) share ONE launch: every streamer holds K shards of ycorr and forms X_t'[y_1 .. y_K] from each tile it reads -- the
// panel is streamed once for K iterations' worth of sampling -- while every chain keeps its own sampler workgroup, reducers and
// hand-off rings.  Where the sweep is bound by the hand-off latency of its pipeline and not by HBM (short shards: 10k x 100k), the
// chains' round trips overlap and the aggregate rate grows almost K-fold; each chain stays, bit for bit, the chain it is alone
// with the same layout (R, S, lag, near lags): per chain nothing about the order of its arithmetic changes.
//   blocks 0, 8, .., 8 (K-1)   samplers (one XCD under round-robin placement: they read the same Gram blocks -- speed only)
//   the next K NG blocks       reducers, chain-major
//   the rest                   S streamers
// Streamer: the phase streamer of shards of at most 64 rows (role_streamer, one update task per thread, every wave forming the
// updated shard itself), with the per-chain state in LDS and every per-chain step looped over the chains.
// ------------------------------------------------------------------------------------------
// KC = chains of the launch, a compile-time constant.  What a chain adds to a block is a handful of short, latency-bound sequences
// (LDS round trip -> dependent fma chain -> LDS write); looped chain after chain they cost 0.4-0.5 us per chain and block (first
// version).  Here every phase is written "all loads of all chains, then all arithmetic, then all stores", so the chains' round
// trips overlap, and the publication of chain c's partial sums is wave c's job (the waves publish side by side).
// Arithmetic per chain: exactly the one-chain phase streamer's (role_streamer, one update task per thread).
#ifndef NGP_MULTI_G
#define NGP_MULTI_G 4  // chains whose loads / arithmetic / stores are interleaved at a time
#endif
template <int DT, int KC>
__device__ __attribute__((always_inline)) inline void role_streamer_multi(const MultiArgs &Mr, const int s, char *smem) {
    const MultiArgs *Mp = &Mr;     // (the kernel's by-value argument: every index below is a compile-time constant after unrolling)
    const SweepArgs &A = Mp->a[0];
    static_assert(KC >= 2 && KC <= 8, "publication: waves 0-3, two chains each at most");
    const int R = A.R, S = A.S, tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), j = tid & 63;
    const size_t TBL = (size_t)(R >> 2) * NGP_QS;
    char *ring = smem;  // 2 tile slots
    const size_t CH = ngp_multi_chain_doubles(R);
    double *cbase = (double *)(smem + 2 * TBL);
    int *sflag = (int *)(cbase + (size_t)KC * CH);
    char *scratch = (char *)sflag + 64;  // 3 KiB sink of the L2-warming DMA
#define NGP_YS(kc) (cbase + (size_t)(kc) * CH)   /* R: the chain's shard of ycorr */
#define NGP_RED(kc) (NGP_YS(kc) + R)             /* 8 x 64 GEMV chain sums */
#define NGP_DL(kc) (NGP_RED(kc) + 512)           /* 2 x 64: dlt of the block being applied, by block parity */
#define NGP_PP(kc) (NGP_DL(kc) + 128)            /* 8 x R: partial sums of the update */
    const size_t tile_elems = (size_t)R * NGP_BLK;
    const int nchunk = R >> 2;
    const int g = s / NGP_GRP;
    const int nb = A.t1 - A.t0;
    // (the lean scalar-base requests of the row-owning loader -- dma16_s4, runs of four quads per wave -- were tried here: three
    // chains unchanged, eight chains 1890 -> 1760 it/s: the requests of a block then leave in one burst, and the partial sums and
    // dlt of eight chains queue behind it)
    auto dma_tile = [&](int ub) {
        const char *src = (const char *)(A.tiles + ((size_t)(A.t0 + ub) * S + s) * tile_elems);
        char *dst = ring + (size_t)(ub & 1) * TBL;
        for (int c = wv - 4; c < nchunk; c += 3) dma16_lds(src + (size_t)c * 1024 + (size_t)j * 16, dst + (size_t)c * NGP_QS);
    };
    // per-chain hand-off words (static indices: registers)
    const double *dltp[KC];
    const unsigned *flagp[KC];
#pragma unroll
    for (int kc = 0; kc < KC; kc++) { dltp[kc] = Mp->a[kc].dlt; flagp[kc] = Mp->a[kc].flag_dlt; }
    // waves 0-3 publish: wave w the partial sums of chain w and of chain w + 4.  Not waves 4-6 -- they issue the tile DMA, and a
    // wave that waits for its DMA (s_waitcnt vmcnt) waits for the acknowledgement of its last store with it: with one publishing
    // wave per chain, the fifth chain put that microsecond into phase A of every block (10k x 100k: 2.06 -> 2.85 us per block).
    const bool pubw = wv < 4 && wv < KC;
    const bool pub2 = pubw && (wv + 4 < KC);
    unsigned long long *my_acc = nullptr, *my_acc2 = nullptr;
    double my_fxs = 0.0, my_fxs2 = 0.0;
#pragma unroll
    for (int kc = 0; kc < KC; kc++) {
        if (kc == wv) { my_acc = Mp->a[kc].acc; my_fxs = Mp->a[kc].scal->fx_scale; }
        if (kc == wv + 4) { my_acc2 = Mp->a[kc].acc; my_fxs2 = Mp->a[kc].scal->fx_scale; }
    }
#pragma unroll
    for (int kc = 0; kc < KC; kc++) {
        const double *yg = Mp->a[kc].ycorr + (size_t)s * R;
        for (int i = tid; i < R; i += NGP_WG) NGP_YS(kc)[i] = yg[i];
    }
    if (tid == 0) *sflag = 1;
    // wave 7: dlt of local block (uu - DT) of every chain into dl[uu & 1]
    auto wait_flags = [&](int target) -> int {  // whole wave; lane c < KC waits for chain c; 1 = every chain has finished `target` blocks
        int ok = 1;
#pragma unroll
        for (int kc = 0; kc < KC; kc++)
            if (j == kc) ok = wait_ge(flagp[kc], (unsigned)target, A.abort_w, 1u) ? 1 : 0;
        return __ballot(ok == 0) == 0ull ? 1 : 0;
    };
    auto poll_dlt = [&](int uu) {
        const int aa = uu - DT;
        if (aa < 0 || uu >= nb + DT) return;
        const int ok = wait_flags(aa + 1);
        if (!ok && j == 0) *sflag = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (ok) {
            double v[KC];
#pragma unroll
            for (int kc = 0; kc < KC; kc++) v[kc] = ld_f64(&dltp[kc][(size_t)(aa % NGP_RING) * NGP_BLK + j]);
#pragma unroll
            for (int kc = 0; kc < KC; kc++) NGP_DL(kc)[(uu & 1) * 64 + j] = v[kc];
        }
    };
    // the update task of this thread (fixed for the whole sweep): 8-column chain tcc = its wave, row ti0 = its lane -- so the eight
    // dlt of a task are wave-uniform (fmac8_bcast); lanes beyond the shard redo the last row (R <= 64: host)
    const int tcc = wv, ti0 = min(j, R - 1);
    float keep[DT][8];
#pragma unroll
    for (int d = 0; d < DT; d++)
#pragma unroll
        for (int jj = 0; jj < 8; jj++) keep[d][jj] = 0.0f;
    if (wv >= 4 && wv <= 6 && nb > 0) dma_tile(0);
    const unsigned my_xcc = xcc_id() + 1u;
    const int nslice = max(1, S / 8);
    const int slice = (s / 8) % nslice;
    const size_t gram_bytes = warm_bytes(A, DT), w_thi = warm_t_hi(A), w_shift = warm_shift(A);
    const size_t slice_bytes = ((gram_bytes / nslice + 1023) / 1024) * 1024;
    bool same_xcd = false;
    auto try_signal = [&](bool) {};
    __syncthreads();
    for (int u0 = 0; u0 < nb + DT; u0 += DT) {
#pragma unroll
        for (int d = 0; d < DT; d++) {
            const int u = u0 + d;
            if (u >= nb + DT) break;
            const int a = u - DT;
            // ---------------- phase A: everything that waits on memory ----------------
            if (wv >= 4 && wv <= 6) {
                drain_vm();
                if (u + 1 < nb) dma_tile(u + 1);
                if ((u & 7) == 0 && !same_xcd) same_xcd = (ld_u32(A.xcc_w) == my_xcc);
                if (same_xcd && u + 1 < nb) {
                    const char *gb = (const char *)(A.gramx + (size_t)(A.t0 + u + 1) * DT * NGP_BLK * NGP_BLK) - w_shift;
                    const char *tb = (const char *)(A.tinv + (size_t)(A.t0 + u + 1) * NGP_BLK * NGP_BLK);
                    const size_t lo = (size_t)slice * slice_bytes, hi = min((size_t)(slice + 1) * slice_bytes, gram_bytes);
                    for (size_t off = lo + (size_t)(wv - 4) * 1024; off + 1024 <= hi; off += 3 * 1024)
                        dma16_lds((off < w_thi ? tb : gb) + off + (size_t)j * 16, scratch + (wv - 4) * 1024);
                }
            }
            try_signal(false);
            wg_barrier();
            if (!*sflag) return;
            // ---------------- phase B: the 8-column partial sums of ycorr -= X_a dlt_a, every chain ----------------
            if (a >= 0) {
                double xk[8];
#pragma unroll
                for (int h = 0; h < 8; h++) xk[h] = (double)keep[d][h];
                // (per chain: the eight fma in column order, as the one-chain streamer; the chains of a group interleaved)
                auto upd_group = [&](auto gn, const int g0) __attribute__((always_inline)) {
                    constexpr int GN = decltype(gn)::value;
                    double dv[GN], p[GN];
#pragma unroll
                    for (int q = 0; q < GN; q++) { dv[q] = NGP_DL(g0 + q)[(u & 1) * 64 + 8 * tcc + (j & 7)]; p[q] = 0.0; }
                    fmac8_bcast<GN>(p, dv, xk);
#pragma unroll
                    for (int q = 0; q < GN; q++) NGP_PP(g0 + q)[(size_t)tcc * R + ti0] = p[q];
                };
#pragma unroll
                for (int g0 = 0; g0 + NGP_MULTI_G <= KC; g0 += NGP_MULTI_G) upd_group(std::integral_constant<int, NGP_MULTI_G>{}, g0);
                if constexpr (KC % NGP_MULTI_G != 0) upd_group(std::integral_constant<int, KC % NGP_MULTI_G>{}, KC - KC % NGP_MULTI_G);
                try_signal(false);
                wg_barrier();
                try_signal(false);
            }
            // the updated shard of chain c is formed ONCE, by wave c (lane = row: tree of the 8 partial sums), and read by every wave's
            // GEMV from LDS behind a barrier -- formed by every wave for itself (what the one-chain streamer does to save this
            // barrier) it cost 30 KB of LDS reads per chain and block, and its v_readlane-fed GEMV 32 readlanes per quad and chain
            if (a >= 0) {
                if (wv < KC && j < R) {
                    double *ys = NGP_YS(wv);
                    const double *pp = NGP_PP(wv);
                    const double T = ((pp[j] + pp[R + j]) + (pp[2 * R + j] + pp[3 * R + j])) +
                                     ((pp[4 * R + j] + pp[5 * R + j]) + (pp[6 * R + j] + pp[7 * R + j]));
                    ys[j] = ys[j] - T;
                }
                wg_barrier();
            }
            // ---------------- phase C: partial X_u' ycorr of every chain, and tile u into the delay line ----------------
            if (u < nb) {
                const int pa = u + 1 - DT;
                const bool pollw = (wv == 7) && (DT >= 3) && (pa >= 0) && (u + 1 < nb + DT);
                unsigned fl = 0xFFFFFFFFu;
                if (pollw) {
#pragma unroll
                    for (int kc = 0; kc < KC; kc++)
                        if (j == kc) fl = ld_u32(flagp[kc]);
                }
                const float *slotp = (const float *)(ring + (size_t)(u & 1) * TBL);
                {
                    const float *tq = slotp + (size_t)(ti0 >> 2) * (NGP_QS / 4) + 32 * tcc + (ti0 & 3);
#pragma unroll
                    for (int jj = 0; jj < 8; jj++) keep[d][jj] = tq[4 * jj];
                }
                {   // chain wv: row quads wv, wv+8, ... (lane = column); one read of the tile serves all chains
                    const float *col = slotp + 4 * j;
                    double acc[KC];
#pragma unroll
                    for (int kc = 0; kc < KC; kc++) acc[kc] = 0.0;
                    for (int qd = wv; qd < (R >> 2); qd += 8) {
                        const float4 x = *(const float4 *)(col + (size_t)qd * (NGP_QS / 4));
                        const double xd[4] = {(double)x.x, (double)x.y, (double)x.z, (double)x.w};
                        // (the quad's four y of a chain: one 8-byte read per lane, lane l holding y[4 qd + l mod 4]; fmac4_bcast)
                        auto gemv_group = [&](auto gn, const int g0) __attribute__((always_inline)) {
                            constexpr int GN = decltype(gn)::value;
                            double yv[GN], ac[GN];
#pragma unroll
                            for (int q = 0; q < GN; q++) { yv[q] = NGP_YS(g0 + q)[4 * qd + (j & 3)]; ac[q] = acc[g0 + q]; }
                            fmac4_bcast<GN>(ac, yv, xd);
#pragma unroll
                            for (int q = 0; q < GN; q++) acc[g0 + q] = ac[q];
                        };
#pragma unroll
                        for (int g0 = 0; g0 + NGP_MULTI_G <= KC; g0 += NGP_MULTI_G) gemv_group(std::integral_constant<int, NGP_MULTI_G>{}, g0);
                        if constexpr (KC % NGP_MULTI_G != 0) gemv_group(std::integral_constant<int, KC % NGP_MULTI_G>{}, KC - KC % NGP_MULTI_G);
                    }
#pragma unroll
                    for (int kc = 0; kc < KC; kc++) NGP_RED(kc)[wv * 64 + j] = acc[kc];
                }
                double dnext[KC];
                bool have_dnext = false;
                if (pollw) {
                    int ok = 1;
                    if (__ballot(fl < (unsigned)(pa + 1)) != 0ull) {  // some chain's sampler is not that far yet: wait (bounded)
                        ok = wait_flags(pa + 1);
                        if (!ok && j == 0) *sflag = 0;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (ok) {
#pragma unroll
                        for (int kc = 0; kc < KC; kc++) dnext[kc] = ld_f64(&dltp[kc][(size_t)(pa % NGP_RING) * NGP_BLK + j]);
                        have_dnext = true;
                    }
                }
                try_signal(false);
                wg_barrier();
                if (have_dnext) {
#pragma unroll
                    for (int kc = 0; kc < KC; kc++) NGP_DL(kc)[((u + 1) & 1) * 64 + j] = dnext[kc];
                }
                if (pubw) {  // waves 0-3 publish side by side, chains w and w + 4
                    const int slot = u % NGP_RING;
                    const double *red = NGP_RED(wv);
                    const double p = ((red[j] + red[64 + j]) + (red[128 + j] + red[192 + j])) + ((red[256 + j] + red[320 + j]) + (red[384 + j] + red[448 + j]));
                    double p2 = 0.0;
                    if (pub2) {
                        const double *red2 = NGP_RED(wv + 4);
                        p2 = ((red2[j] + red2[64 + j]) + (red2[128 + j] + red2[192 + j])) + ((red2[256 + j] + red2[320 + j]) + (red2[384 + j] + red2[448 + j]));
                    }
                    acc_add(my_acc, A.abort_w, slot, s, j, p, my_fxs);
                    if (pub2) acc_add(my_acc2, A.abort_w, slot, s, j, p2, my_fxs2);
                }
            } else if (wv == 7 && DT >= 3) poll_dlt(u + 1);
        }
    }
    try_signal(true);
    __syncthreads();
#pragma unroll
    for (int kc = 0; kc < KC; kc++) {
        const double *yfin = NGP_YS(kc);
        double *yg = Mp->a[kc].ycorr + (size_t)s * R;
        for (int i = tid; i < R; i += NGP_WG) yg[i] = yfin[i];
    }
#undef NGP_YS
#undef NGP_RED
#undef NGP_DL
#undef NGP_PP
}

// ------------------------------------------------------------------------------------------
// K chains per pass for the ROW-OWNING streamer (fp32 tiles, shards of 64..NGP_ROWS_MAX_R rows: the 50k x 600k shape, where one
// chain streams the panel at 0.62 of the HBM roofline): the loader wave and the tile ring are the one-chain streamer's
// (role_streamer_rows), every row wave applies the update and forms the GEMV chain of ITS rows for each chain in turn -- the tile
// element read from LDS serves all chains, the delay line holds the tile once -- wave 2 publishes the chains' partial sums, wave 6
// fetches their dlt granules.  Per chain: the arithmetic of role_streamer_rows<DT, 0>, operation for operation.
// LDS: ring | per chain: shard (R) | 2 x 7 x 64 chain sums | 2 x 72 dlt | then flags, counters and the 1 KiB sink.
// ------------------------------------------------------------------------------------------
template <int DT, int KC, int ST = 0>
__device__ __attribute__((always_inline)) inline void role_streamer_rows_multi(const MultiArgs &Mr, const int s, char *smem) {
    // ST = 0: fp32 tiles; ST = 1, 2, 4: compact storage (byte tiles, ST update tasks per lane), as role_streamer_rows<.., DT, ST>:
    // a byte is converted ONCE and multiplied into every chain's sums -- the conversions are what bounds the one-chain byte streamer
    constexpr bool U8 = (ST != 0);
    constexpr int NT = U8 ? ST : 1;
    constexpr bool early = !U8;  // (the publisher ahead of the block's barrier: a gain on fp32 tiles, a loss on byte tiles -- role_streamer_rows)
    const MultiArgs *Mp = &Mr;
    const SweepArgs &A = Mp->a[0];
    const int R = A.R, S = A.S, tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int NQ = U8 ? (R >> 4) : (R >> 2);
    const int H = min(NGP_ROWS_HMAX, (NQ + 1) >> 1);
    const int RQ = 2 * NQ + H;
    char *ring = smem;
    const size_t CH = ngp_rows_multi_chain_doubles(R);
    double *cbase = (double *)(smem + (size_t)RQ * NGP_QS);
#define NGP_YS(kc) (cbase + (size_t)(kc) * CH)
#define NGP_RED(kc) (NGP_YS(kc) + ((R + 7) & ~7))
#define NGP_DL(kc) (NGP_RED(kc) + 2 * NGP_ROWS_NW * NGP_BLK)
#define NGP_RSY(kc) (NGP_DL(kc) + 2 * NGP_DLS)
    int *sflag = (int *)(cbase + (size_t)KC * CH);
    int *gcnt0 = sflag + 4, *gcnt1 = sflag + 8;
    char *scratch = (char *)sflag + 64 + 64;
    const size_t tile_bytes = (size_t)NQ * 1024;
    const int nb = A.t1 - A.t0;
#pragma unroll
    for (int kc = 0; kc < KC; kc++) {
        const double *yg = Mp->a[kc].ycorr + (size_t)s * R;
        for (int i = tid; i < R; i += NGP_WG) NGP_YS(kc)[i] = yg[i];
    }
    if (tid == 0) { *sflag = 1; *gcnt0 = 0; *gcnt1 = 0; }
    int base = 0;
    auto wrap = [&](int p) __attribute__((always_inline)) { return p >= RQ ? p - RQ : p; };
    if (wv == NGP_ROWS_NW) {
        // ------------------------------ loader (as role_streamer_rows) ------------------------------
        const unsigned ring0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)ring;
        const unsigned scratch0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)scratch;
        const unsigned voff = (unsigned)lane * 16u;
        auto dma_quads = [&](int tile, int q0, int q1, int tbase) __attribute__((always_inline)) {
            if (q0 >= q1) return 0;
            const char *gp = (const char *)A.tiles + ((size_t)(A.t0 + tile) * S + s) * tile_bytes + (size_t)q0 * 1024;
            int p = wrap(tbase + q0);
            int q = q0;
            for (; q + 4 <= q1; q += 4) {
                if (p + 4 <= RQ) {
                    dma16_s4(ring0 + (unsigned)p * NGP_QS, gp, voff);
                    p += 4;
                    if (p == RQ) p = 0;
                } else {
                    for (int k = 0; k < 4; k++) {
                        dma16_s(ring0 + (unsigned)p * NGP_QS, gp + k * 1024, voff);
                        if (++p == RQ) p = 0;
                    }
                }
                gp += 4096;
            }
            for (; q < q1; ++q) {
                dma16_s(ring0 + (unsigned)p * NGP_QS, gp, voff);
                gp += 1024;
                if (++p == RQ) p = 0;
            }
            return q1 - q0;
        };
        const unsigned my_xcc = xcc_id() + 1u;
        const int nslice = max(1, S / 8);
        const int slice = (s / 8) % nslice;
        const size_t gram_bytes = warm_bytes(A, min(DT, A.near + 1)), w_thi = warm_t_hi(A), w_shift = warm_shift(A);
        const size_t slice_bytes = ((gram_bytes / nslice + 1023) / 1024) * 1024;
        bool same_xcd = false, xcc_known = false;
        __builtin_amdgcn_s_setprio(3);
        if (nb > 0) dma_quads(0, 0, NQ, 0);
        if (nb > 1) dma_quads(1, 0, H, NQ);
        drain_vm();
        wg_barrier();
        for (int u = 0; u < nb + DT; ++u) {
            const int base1 = wrap(base + NQ), base2 = wrap(base1 + NQ);
            if (!xcc_known && (u & 7) == 0) {
                const unsigned x = sld_u32(A.xcc_w);
                xcc_known = (x != 0u);
                same_xcd = (x == my_xcc);
            }
            if (same_xcd && u + 1 < nb) {
                const char *gb = (const char *)(A.gramx + (size_t)(A.t0 + u + 1) * DT * NGP_BLK * NGP_BLK) - w_shift;
                const char *tb = (const char *)(A.tinv + (size_t)(A.t0 + u + 1) * NGP_BLK * NGP_BLK);
                const size_t lo = (size_t)slice * slice_bytes, hi = min((size_t)(slice + 1) * slice_bytes, gram_bytes);
                for (size_t off = lo; off + 1024 <= hi; off += 1024) dma16_warm(scratch0, (off < w_thi ? tb : gb) + off, voff);
            }
            if (u + 1 < nb) dma_quads(u + 1, H, NQ, base1);
            int n2 = 0;
            if (u + 2 < nb) n2 = dma_quads(u + 2, 0, H, base2);
            wait_vmcnt_le(n2);
            wg_barrier();
            if (!*sflag) return;
            base = base1;
        }
        drain_vm();
    } else {
        // ------------------------------ row-owning waves ------------------------------
        const int c = lane & 7, ql = lane >> 3;
        const int nqw = (NQ - wv + NGP_ROWS_NW - 1) / NGP_ROWS_NW;
        int tslot[NT], trow[NT];  // update tasks of this lane (role_streamer_rows)
        bool thas[NT];
#pragma unroll
        for (int i = 0; i < NT; i++) {
            if constexpr (U8) {
                const int tau = ql + 8 * i;
                thas[i] = (tau >> 2) < nqw;
                tslot[i] = thas[i] ? wv + NGP_ROWS_NW * (tau >> 2) : wv;
                trow[i] = 16 * tslot[i] + 4 * (tau & 3);
            } else {
                const int qt = wv + NGP_ROWS_NW * ql;
                thas[i] = qt < NQ;
                tslot[i] = thas[i] ? qt : wv;
                trow[i] = 4 * qt;
            }
        }
        const int nvalid = U8 ? (int)max(0ll, min((long long)R, A.N - (long long)s * R)) : R;
        float4 keep[U8 ? 1 : DT][8];
        unsigned keep8[U8 ? DT : 1][NT][8];
#pragma unroll
        for (int d = 0; d < DT; d++)
#pragma unroll
            for (int jj = 0; jj < 8; jj++) {
                if constexpr (U8) {
#pragma unroll
                    for (int i = 0; i < NT; i++) keep8[d][i][jj] = 0u;
                } else {
                    keep[d][jj] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        unsigned long long pg0[KC], pg1[KC];  // poller: granules of the next dlt of every chain
#pragma unroll
        for (int kc = 0; kc < KC; kc++) { pg0[kc] = 0; pg1[kc] = 0; }
        double pm = 0.0;   // poller, compact storage: column mean of (block pm_blk, column lane) -- the panel's, one for all chains
        int pm_blk = -1;
        double mpub = 0.0;  // publisher, compact storage: column mean of the block published next
        if (U8 && wv == NGP_ROWS_PUBW && nb > 0) mpub = A.mean[(size_t)A.t0 * NGP_BLK + lane];
        auto try_signal = [&](bool) __attribute__((always_inline)) {};
        auto publish = [&](const int u) __attribute__((always_inline)) {
            const int slot = u % NGP_RING;
            double p[KC];
#pragma unroll
            for (int kc = 0; kc < KC; kc++) {
                const double *rp = NGP_RED(kc) + (u & 1) * NGP_ROWS_NW * NGP_BLK + lane;
                p[kc] = ((rp[0] + rp[NGP_BLK]) + (rp[2 * NGP_BLK] + rp[3 * NGP_BLK])) + ((rp[4 * NGP_BLK] + rp[5 * NGP_BLK]) + rp[6 * NGP_BLK]);
                if constexpr (U8) {  // partial = sum_i g_ij y_i - m_j sum_i y_i
                    const double *sp = NGP_RSY(kc) + (u & 1) * 8;
                    const double sy = ((sp[0] + sp[1]) + (sp[2] + sp[3])) + ((sp[4] + sp[5]) + sp[6]);
                    const double ms = mpub * sy;
                    p[kc] = p[kc] - ms;
                }
            }
            if (U8 && u + 1 < nb) mpub = A.mean[(size_t)(A.t0 + u + 1) * NGP_BLK + lane];  // (requested before the adds leave)
#pragma unroll
            for (int kc = 0; kc < KC; kc++) acc_add(Mp->a[kc].acc, A.abort_w, slot, s, lane, p[kc], Mp->a[kc].scal->fx_scale);
        };
        wg_barrier();
        for (int u0 = 0; u0 < nb + DT; u0 += DT) {
#pragma unroll
            for (int d = 0; d < DT; d++) {
                const int u = u0 + d;
                if (u >= nb + DT) break;
                const int a = u - DT;
                const int pa = u + 1 - DT;
                const bool pollw = (wv == NGP_ROWS_POLLW) && (pa >= 0) && (u + 1 < nb + DT);
                bool have_dnext[KC];
#pragma unroll
                for (int kc = 0; kc < KC; kc++) have_dnext[kc] = false;
                if (pollw) {
#pragma unroll
                    for (int kc = 0; kc < KC; kc++) {
                        if (dlt_granules_valid(pg0[kc], pg1[kc], dlt_tag(Mp->a[kc].nonce, pa))) {
                            have_dnext[kc] = true;
                        } else {
                            const unsigned long long *gp = Mp->a[kc].dltg + ((size_t)(pa % NGP_RING) * NGP_BLK + lane) * 2;
                            pg0[kc] = ld_u64(gp);
                            pg1[kc] = ld_u64(gp + 1);
                        }
                    }
                    if (U8 && pm_blk != pa) { pm = A.mean[(size_t)(A.t0 + pa) * NGP_BLK + lane]; pm_blk = pa; }
                }
                // ---- ycorr -= X_a dlt_a for the rows of this wave, chain after chain (the tile elements wait in keep[d]) ----
                if constexpr (!U8) {
                if (a >= 0) {
#pragma unroll
                    for (int kc = 0; kc < KC; kc++) {
                        const double *dq = NGP_DL(kc) + (u & 1) * NGP_DLS + 8 * c;
                        double dqv[8];
#pragma unroll
                        for (int jj = 0; jj < 8; jj++) dqv[jj] = dq[jj];
                        double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll
                        for (int jj = 0; jj < 8; jj++) {
                            p0 = __builtin_fma((double)keep[d][jj].x, dqv[jj], p0);
                            p1 = __builtin_fma((double)keep[d][jj].y, dqv[jj], p1);
                            p2 = __builtin_fma((double)keep[d][jj].z, dqv[jj], p2);
                            p3 = __builtin_fma((double)keep[d][jj].w, dqv[jj], p3);
                        }
                        p0 = p0 + dpp_f64(p0, 0); p1 = p1 + dpp_f64(p1, 0); p2 = p2 + dpp_f64(p2, 0); p3 = p3 + dpp_f64(p3, 0);
                        p0 = p0 + dpp_f64(p0, 1); p1 = p1 + dpp_f64(p1, 1); p2 = p2 + dpp_f64(p2, 1); p3 = p3 + dpp_f64(p3, 1);
                        p0 = p0 + dpp_f64(p0, 2); p1 = p1 + dpp_f64(p1, 2); p2 = p2 + dpp_f64(p2, 2); p3 = p3 + dpp_f64(p3, 2);
                        if (c == 0 && thas[0]) {
                            double *yq = NGP_YS(kc) + trow[0];
                            const double y0 = yq[0], y1 = yq[1], y2 = yq[2], y3 = yq[3];
                            yq[0] = y0 - p0; yq[1] = y1 - p1; yq[2] = y2 - p2; yq[3] = y3 - p3;
                        }
                    }
                }
                } else {
                    // compact storage (role_streamer_rows): y_i -= (sum_j g_ij dlt_j - sum_j m_j dlt_j) for the rows inside the panel,
                    // and the sum of this wave's rows of y; every byte of the delay line converted once for all chains
                    double dqv[KC][8], cm[KC], vsum[KC];
#pragma unroll
                    for (int kc = 0; kc < KC; kc++) {
                        const double *dq = NGP_DL(kc) + (u & 1) * NGP_DLS + 8 * c;
#pragma unroll
                        for (int jj = 0; jj < 8; jj++) dqv[kc][jj] = (a >= 0) ? dq[jj] : 0.0;
                        cm[kc] = (a >= 0) ? NGP_DL(kc)[(u & 1) * NGP_DLS + NGP_BLK] : 0.0;
                        vsum[kc] = 0.0;
                    }
#pragma unroll
                    for (int i = 0; i < NT; i++) {
                        double p0[KC], p1[KC], p2[KC], p3[KC];
#pragma unroll
                        for (int kc = 0; kc < KC; kc++) { p0[kc] = 0.0; p1[kc] = 0.0; p2[kc] = 0.0; p3[kc] = 0.0; }
                        if (a >= 0) {
#pragma unroll
                            for (int jj = 0; jj < 8; jj++) {
                                const unsigned w = keep8[d][i][jj];
                                const double g0 = (double)(float)(w & 0xffu), g1 = (double)(float)((w >> 8) & 0xffu);
                                const double g2 = (double)(float)((w >> 16) & 0xffu), g3 = (double)(float)(w >> 24);
#pragma unroll
                                for (int kc = 0; kc < KC; kc++) {
                                    p0[kc] = __builtin_fma(g0, dqv[kc][jj], p0[kc]);
                                    p1[kc] = __builtin_fma(g1, dqv[kc][jj], p1[kc]);
                                    p2[kc] = __builtin_fma(g2, dqv[kc][jj], p2[kc]);
                                    p3[kc] = __builtin_fma(g3, dqv[kc][jj], p3[kc]);
                                }
                            }
#pragma unroll
                            for (int kc = 0; kc < KC; kc++) {
                                p0[kc] = p0[kc] + dpp_f64(p0[kc], 0); p1[kc] = p1[kc] + dpp_f64(p1[kc], 0); p2[kc] = p2[kc] + dpp_f64(p2[kc], 0); p3[kc] = p3[kc] + dpp_f64(p3[kc], 0);
                                p0[kc] = p0[kc] + dpp_f64(p0[kc], 1); p1[kc] = p1[kc] + dpp_f64(p1[kc], 1); p2[kc] = p2[kc] + dpp_f64(p2[kc], 1); p3[kc] = p3[kc] + dpp_f64(p3[kc], 1);
                                p0[kc] = p0[kc] + dpp_f64(p0[kc], 2); p1[kc] = p1[kc] + dpp_f64(p1[kc], 2); p2[kc] = p2[kc] + dpp_f64(p2[kc], 2); p3[kc] = p3[kc] + dpp_f64(p3[kc], 2);
                            }
                        }
                        if (c == 0 && thas[i]) {
#pragma unroll
                            for (int kc = 0; kc < KC; kc++) {
                                double *yq = NGP_YS(kc) + trow[i];
                                double y0 = yq[0], y1 = yq[1], y2 = yq[2], y3 = yq[3];
                                if (a >= 0) {
                                    const int r0 = trow[i];
                                    const double t0 = p0[kc] - cm[kc], t1 = p1[kc] - cm[kc], t2 = p2[kc] - cm[kc], t3 = p3[kc] - cm[kc];
                                    if (r0 + 0 < nvalid) y0 = y0 - t0;
                                    if (r0 + 1 < nvalid) y1 = y1 - t1;
                                    if (r0 + 2 < nvalid) y2 = y2 - t2;
                                    if (r0 + 3 < nvalid) y3 = y3 - t3;
                                    yq[0] = y0; yq[1] = y1; yq[2] = y2; yq[3] = y3;
                                }
                                const double q4 = (y0 + y1) + (y2 + y3);
                                vsum[kc] = (i == 0) ? q4 : vsum[kc] + q4;
                            }
                        }
                    }
                    if (u < nb) {
#pragma unroll
                        for (int kc = 0; kc < KC; kc++) {
                            double vs = vsum[kc];
                            vs = vs + dpp_f64(vs, 4);
                            const double r0 = readlane_d(vs, 0), r1 = readlane_d(vs, 16), r2 = readlane_d(vs, 32), r3 = readlane_d(vs, 48);
                            if (lane == 0) NGP_RSY(kc)[(u & 1) * 8 + wv] = (r0 + r1) + (r2 + r3);
                        }
                    }
                }
                try_signal(false);
                if (u < nb) {
                    // ---- GEMV chains of this wave: slots wv, wv+7, ... (lane = column); one read of a tile quad serves all chains ----
                    double acc[KC];
#pragma unroll
                    for (int kc = 0; kc < KC; kc++) acc[kc] = 0.0;
                    for (int k = 0; k < nqw; k++) {
                        const int q = wv + NGP_ROWS_NW * k;
                        if constexpr (U8) {
                            const uint4 x = *(const uint4 *)(ring + (size_t)wrap(base + q) * NGP_QS + (size_t)lane * 16);
                            const unsigned xw[4] = {x.x, x.y, x.z, x.w};
                            // (every chain's y of a quad: ONE 8-byte read per lane and the DPP broadcast -- a quarter of the LDS read cycles of four
                            // broadcast reads: 50k x 600k 63.7 -> 66.9 it/s, 10k x 100k 475 -> 491.  The same change measured WORSE on fp32 tiles with two
                            // chains (67.8 -> 66.6) and on byte tiles with one (17.7 -> 17.9 ms): left as they are)
                            double y4[4][KC];
#pragma unroll
                            for (int e4 = 0; e4 < 4; e4++)
#pragma unroll
                                for (int kc = 0; kc < KC; kc++) y4[e4][kc] = NGP_YS(kc)[16 * q + 4 * e4 + (lane & 3)];
#pragma unroll
                            for (int e4 = 0; e4 < 4; e4++) {
                                const unsigned w = xw[e4];
                                const double xd[4] = {(double)(float)(w & 0xffu), (double)(float)((w >> 8) & 0xffu), (double)(float)((w >> 16) & 0xffu), (double)(float)(w >> 24)};
                                fmac4_bcast<KC>(acc, y4[e4], xd);
                            }
                        } else {
                            const float4 x = *(const float4 *)(ring + (size_t)wrap(base + q) * NGP_QS + (size_t)lane * 16);
#pragma unroll
                            for (int kc = 0; kc < KC; kc++) {
                                const double *yq = NGP_YS(kc) + 4 * q;
                                const double y0 = yq[0], y1 = yq[1], y2 = yq[2], y3 = yq[3];
                                acc[kc] = __builtin_fma((double)x.x, y0, acc[kc]);
                                acc[kc] = __builtin_fma((double)x.y, y1, acc[kc]);
                                acc[kc] = __builtin_fma((double)x.z, y2, acc[kc]);
                                acc[kc] = __builtin_fma((double)x.w, y3, acc[kc]);
                            }
                        }
                    }
#pragma unroll
                    for (int kc = 0; kc < KC; kc++) NGP_RED(kc)[((u & 1) * NGP_ROWS_NW + wv) * NGP_BLK + lane] = acc[kc];
                    if (early) {
                        asm volatile("" ::: "memory");  // LDS serves a wave in order: the count follows the sums
                        if (lane == 0) __hip_atomic_fetch_add((lds_int_t *)((u & 1) ? gcnt1 : gcnt0), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    try_signal(false);
                    // ---- tile u into the delay line ----
                    if constexpr (U8) {
#pragma unroll
                        for (int i = 0; i < NT; i++) {
                            const char *tq = ring + (size_t)wrap(base + tslot[i]) * NGP_QS + c * 128 + (trow[i] & 15);
#pragma unroll
                            for (int jj = 0; jj < 8; jj++) keep8[d][i][jj] = *(const unsigned *)(tq + jj * 16);
                        }
                    } else {
                        const char *tq = ring + (size_t)wrap(base + tslot[0]) * NGP_QS + c * 128;
#pragma unroll
                        for (int jj = 0; jj < 8; jj++) keep[d][jj] = *(const float4 *)(tq + jj * 16);
                    }
                }
                // the publisher waits for the seven chain waves through the LDS counter and publishes at once (before the barrier)
                if (early && wv == NGP_ROWS_PUBW && u < nb) {
                    const int *gc = (u & 1) ? gcnt1 : gcnt0;
                    for (unsigned sp = 0; lds_flag_ld(gc) < NGP_ROWS_NW; ++sp) {
                        if ((sp & 255u) == 255u && (lds_flag_ld(sflag) == 0 || sp > (NGP_SPIN_LIMIT << 4))) {
                            if (lane == 0) *sflag = 0;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(0);
                    }
                    asm volatile("" ::: "memory");
                    if (lane == 0) lds_flag_st((u & 1) ? gcnt0 : gcnt1, 0);
                    publish(u);
                }
                if (pollw) {
#pragma unroll
                    for (int kc = 0; kc < KC; kc++) {
                        int ok = 1;
                        if (!have_dnext[kc]) {
                            ok = wait_dlt_granules_all(Mp->a[kc].dltg, Mp->a[kc].nonce, pa, lane, A.abort_w, 1u, pg0[kc], pg1[kc]) ? 1 : 0;
                            if (!ok && lane == 0) *sflag = 0;
                        }
                        if (ok) {
                            const double dv = dlt_granules_value(pg0[kc], pg1[kc]);
                            NGP_DL(kc)[((u + 1) & 1) * NGP_DLS + lane] = dv;
                            if constexpr (U8) {  // sum_j m_j dlt_j (role_streamer_rows: the same butterfly)
                                double v = pm * dv;
                                v = v + dpp_f64(v, 0);
                                v = v + dpp_f64(v, 1);
                                v = v + dpp_f64(v, 2);
                                v = v + dpp_f64(v, 3);
                                const double r0 = readlane_d(v, 0), r1 = readlane_d(v, 16), r2 = readlane_d(v, 32), r3 = readlane_d(v, 48);
                                if (lane == 0) NGP_DL(kc)[((u + 1) & 1) * NGP_DLS + NGP_BLK] = (r0 + r1) + (r2 + r3);
                            }
                        }
                    }
                }
                if (wv == NGP_ROWS_POLLW && pa + 1 >= 0 && u + 2 < nb + DT) {
#pragma unroll
                    for (int kc = 0; kc < KC; kc++) {
                        const unsigned long long *gp = Mp->a[kc].dltg + ((size_t)((pa + 1) % NGP_RING) * NGP_BLK + lane) * 2;
                        pg0[kc] = ld_u64(gp);
                        pg1[kc] = ld_u64(gp + 1);
                    }
                    if (U8 && pa + 1 < nb) { pm = A.mean[(size_t)(A.t0 + pa + 1) * NGP_BLK + lane]; pm_blk = pa + 1; }
                }
                try_signal(DT < 4);
                wg_barrier();
                if (!*sflag) return;
                if (!early && wv == NGP_ROWS_PUBW && u < nb) publish(u);
                base = wrap(base + NQ);
            }
        }
        try_signal(true);
    }
    __syncthreads();
#pragma unroll
    for (int kc = 0; kc < KC; kc++) {
        double *yg = Mp->a[kc].ycorr + (size_t)s * R;
        for (int i = tid; i < R; i += NGP_WG) yg[i] = NGP_YS(kc)[i];
    }
#undef NGP_YS
#undef NGP_RED
#undef NGP_DL
#undef NGP_RSY
}

#if !defined(NGP_INST_DBG) || !NGP_INST_DBG || NGP_INST_DBG == 2 || NGP_INST_DBG == 3  // the production translation unit (ngp_sweep_inst.hip, -DNGP_INST_DBG=0), the Tuple and the BayesR one
// A chain's launch arguments by RUN-TIME chain index: indexing the by-value argument M.a[c] makes the compiler copy all of M to
// scratch (3 KB per lane) and read every field from there.  The arguments already sit in the kernarg segment -- constant address
// space, read with scalar loads -- so the entry is addressed there directly (M is the kernel's first argument: offset 0).
__device__ __attribute__((always_inline)) inline const SweepArgs &multi_chain_args(const int c) {
    typedef const __attribute__((address_space(4))) char *kptr;
    kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    return *(const SweepArgs *)(ka + offsetof(MultiArgs, a) + (size_t)c * sizeof(SweepArgs));
}
#endif
#if !defined(NGP_INST_DBG) || !NGP_INST_DBG  // one definition: the production translation unit
__global__ __launch_bounds__(NGP_WG) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_sweep_multi(MultiArgs M) {
    constexpr bool TUPM = false;
    constexpr int RCLSM = 1;
#include "ngp_sweep_multi_body.inc"
}
#endif
#if defined(NGP_INST_DBG) && NGP_INST_DBG == 2  // K chains with a Tuple set per pass: the Tuple translation unit
__global__ __launch_bounds__(NGP_WG) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_sweep_multi_tup(MultiArgs M) {
    constexpr bool TUPM = true;
    constexpr int RCLSM = 1;
#include "ngp_sweep_multi_body.inc"
}
#endif
#if defined(NGP_INST_DBG) && NGP_INST_DBG == 3  // K chains with a BayesR set per pass: the samplers of k_sweep_r (class coefficients one block ahead
                                                // through LDS, the lazy class search) -- the BayesR translation unit
__global__ __launch_bounds__(NGP_WG) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_sweep_multi_r(MultiArgs M) {
    constexpr bool TUPM = true;
    constexpr int RCLSM = 2;
#include "ngp_sweep_multi_body.inc"
}
#endif

}  // namespace ngp
