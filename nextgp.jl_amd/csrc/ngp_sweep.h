// ngp_sweep.h -- the persistent sweep kernel ("stage B"): ONE launch per Gibbs iteration walks all
// 64-SNP blocks.  Workgroups take fixed roles, one workgroup per CU:
//
//   sampler  (1)   wave 0 runs the serial 64-step recursion of every block; waves 1-3 prefetch the
//                  Gram blocks into LDS and compute the look-ahead corrections G[t,a] * dlt_a.
//   reducers (NG)  reducer g sums the partial X_t'y of shards 32g..32g+31 (fixed order).
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
#include "ngp_kernels.h"

#pragma clang fp contract(off)

#define NGP_RING 16        // slots of every communication ring (>= lag D)
#define NGP_MAX_LAG 8
#define NGP_SPIN_LIMIT (1u << 21)
#define NGP_WG 512          // threads per workgroup of the persistent kernel
#define NGP_DBG_STREAM (1u << 20)  // offset of streamer 0's stamps in the debug buffer

namespace ngp {

struct SweepArgs {
    const float *tiles;
    double *ycorr;
    const double *gramx;
    int D, R, S, NG, t0, t1;
    double *beta;
    uint8_t *delta;
    const double *c, *w, *q, *T, *chi;
    const int8_t *setof;
    const int32_t *vbidx;
    DSet *sets;
    double *varBeta;
    // communication (zeroed before every launch)
    double *part;        // [RING][S][64]
    double *gsum;        // [RING][NG][64]
    double *dlt;         // [RING][64]
    unsigned *cnt_part;  // [RING][NG] counters, one 128-B line each
    unsigned *cnt_gs;    // [RING] counters, one 128-B line each
    unsigned *flag_dlt;  // number of blocks the sampler has finished
    unsigned *abort_w;   // != 0: a spin timed out (code = role)
    unsigned long long *dbg;  // optional time stamps (diagnostic runs only), else nullptr
};

__device__ inline unsigned ld_u32(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_u32(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline double ld_f64(const double *p) {
    unsigned long long u = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __longlong_as_double((long long)u);
}
__device__ inline void st_f64(double *p, double v) {
    __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void drain_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

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
// 512-thread workgroups (8 waves).  Streamer: all waves load / update, waves 0-3 do the GEMV.
__device__ inline void role_streamer(const SweepArgs &A, const int s, char *smem) {
    const int R = A.R, S = A.S, D = A.D, tid = threadIdx.x;
    float *tl = (float *)smem;
    double *ys = (double *)(smem + (size_t)R * 256);
    double *red = ys + R;
    double *dl = red + 256;
    int *sflag = (int *)(dl + 64);
    const size_t tile_elems = (size_t)R * NGP_BLK;
    double *yg = A.ycorr + (size_t)s * R;
    for (int i = tid; i < R; i += NGP_WG) ys[i] = yg[i];
    __syncthreads();
    const int g = s / NGP_GRP;
    const int wv = tid >> 6, j = tid & 63;
    const int nb = A.t1 - A.t0;
    for (int u = 0; u < nb + D; ++u) {
        if (u >= D) {  // update with local block a = u - D
            const int a = u - D;
            if (tid == 0) {
                *sflag = wait_ge(A.flag_dlt, (unsigned)(a + 1), A.abort_w, 1u) ? 1 : 0;
                if (A.dbg && s == 0) A.dbg[NGP_DBG_STREAM + 2 * (size_t)u + 1] = wall_clock64();
            }
            __syncthreads();
            if (!*sflag) return;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (tid < 64) dl[tid] = ld_f64(&A.dlt[(size_t)(a % NGP_RING) * NGP_BLK + tid]);
            __syncthreads();
            const float *tp = A.tiles + ((size_t)(A.t0 + a) * S + s) * tile_elems;
            for (int i = tid; i < R; i += NGP_WG) {
                double yv = ys[i];
#pragma unroll 16
                for (int jj = 0; jj < NGP_BLK; jj++) yv = __builtin_fma(-(double)tp[(size_t)jj * R + i], dl[jj], yv);
                ys[i] = yv;
            }
        }
        if (u < nb) {
            const float4 *src = (const float4 *)(A.tiles + ((size_t)(A.t0 + u) * S + s) * tile_elems);
            float4 *dst = (float4 *)tl;
            for (int idx = tid; idx < R * 16; idx += NGP_WG) dst[idx] = src[idx];
            __syncthreads();
            if (wv < 4) {
                const float *col = tl + (size_t)j * R;
                double acc = 0.0;
                for (int qd = wv; qd < (R >> 2); qd += 4) {
                    float4 x = *(const float4 *)(col + 4 * qd);
                    const double *yq = ys + 4 * qd;
                    acc = __builtin_fma((double)x.x, yq[0], acc);
                    acc = __builtin_fma((double)x.y, yq[1], acc);
                    acc = __builtin_fma((double)x.z, yq[2], acc);
                    acc = __builtin_fma((double)x.w, yq[3], acc);
                }
                red[wv * 64 + j] = acc;
            }
            __syncthreads();
            if (wv == 0) {
                const int slot = u % NGP_RING;
                double p = ((red[j] + red[64 + j]) + red[128 + j]) + red[192 + j];
                st_f64(&A.part[((size_t)slot * S + s) * NGP_BLK + j], p);
                drain_vm();
                if (j == 0) {
                    atomicAdd(&A.cnt_part[((size_t)slot * A.NG + g) * 32], 1u);
                    if (A.dbg && s == 0) A.dbg[NGP_DBG_STREAM + 2 * (size_t)u] = wall_clock64();
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < R; i += NGP_WG) yg[i] = ys[i];
}

// ------------------------------------------------------------------------------------------
// reducer g: every wave works on its own blocks (u = wave, wave+8, ...), no workgroup barrier
__device__ inline void role_reducer(const SweepArgs &A, const int g) {
    const int S = A.S, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int s0 = g * NGP_GRP, s1 = min(s0 + NGP_GRP, S), gsize = s1 - s0;
    const int nb = A.t1 - A.t0;
    for (int u = wv; u < nb; u += NGP_WG / 64) {
        const int slot = u % NGP_RING, round = u / NGP_RING;
        int ok = 1;
        if (lane == 0) ok = wait_ge(&A.cnt_part[((size_t)slot * A.NG + g) * 32], (unsigned)((round + 1) * gsize), A.abort_w, 2u) ? 1 : 0;
        ok = __shfl(ok, 0);
        if (!ok) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double *p = A.part + ((size_t)slot * S + s0) * NGP_BLK + lane;
        double vals[NGP_GRP];
#pragma unroll
        for (int s = 0; s < NGP_GRP; s++) vals[s] = ld_f64(p + (size_t)min(s, gsize - 1) * NGP_BLK);
        double v = vals[0];
#pragma unroll
        for (int s = 1; s < NGP_GRP; s++)
            if (s < gsize) v = v + vals[s];
        st_f64(&A.gsum[((size_t)slot * A.NG + g) * NGP_BLK + lane], v);
        drain_vm();
        if (lane == 0) atomicAdd(&A.cnt_gs[(size_t)slot * 32], 1u);
    }
}

// ------------------------------------------------------------------------------------------
// sampler (8 waves): wave 0 = serial chain (LDS + ALU only), wave 1 = publisher of the finished
// block, waves 2-7 = Gram traffic (LDS prefetch of the next diagonal / lag-1 blocks, far
// corrections straight from global memory); wave 2 also fetches the next block's group sums.
// LDS: Gd[2][4096] | Gx[2][4096] | hist[RING][64] | vacc[RING][64] | r0[2][64] | outb[2][64] | outi[2][64] | flags
struct CoefRegs {
    double bo, cc, ww, qq, TT;
};
__device__ inline CoefRegs load_coef(const SweepArgs &A, long long k) {
    CoefRegs c;
    c.bo = A.beta[k];
    c.cc = A.c[k];
    c.ww = A.w[k];
    c.qq = A.q[k];
    c.TT = A.T[k];
    return c;
}

// group sums of local block u -> per-lane total (lane 0 polls, whole wave loads); false on abort
__device__ inline bool fetch_group_sums(const SweepArgs &A, int u, int j, double *tot_out) {
    const int NG = A.NG, slot = u % NGP_RING;
    int ok = 1;
    if (j == 0) ok = wait_ge(&A.cnt_gs[(size_t)slot * 32], (unsigned)((u / NGP_RING + 1) * NG), A.abort_w, 3u) ? 1 : 0;
    ok = __shfl(ok, 0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!ok) return false;
    const double *gp = A.gsum + (size_t)slot * NG * NGP_BLK + j;
    double gv[8];
#pragma unroll
    for (int g = 0; g < 8; g++) gv[g] = ld_f64(gp + (size_t)min(g, NG - 1) * NGP_BLK);
    double tot = gv[0];
#pragma unroll
    for (int g = 1; g < 8; g++)
        if (g < NG) tot = tot + gv[g];
    *tot_out = tot;
    return true;
}

__device__ inline void role_sampler(const SweepArgs &A, char *smem) {
    const int D = A.D, NG = A.NG, tid = threadIdx.x, wv = tid >> 6, j = tid & 63;
    double *Gd = (double *)smem;                // 2 x 4096
    double *Gx = Gd + 2 * 4096;                 // 2 x 4096 (lag-1 cross block, [k][j])
    double *hist = Gx + 2 * 4096;               // RING x 64
    double *vacc = hist + NGP_RING * NGP_BLK;   // RING x 64
    double *r0 = vacc + NGP_RING * NGP_BLK;     // 2 x 64
    double *outb = r0 + 2 * NGP_BLK;            // 2 x 64
    int *outi = (int *)(outb + 2 * NGP_BLK);    // 2 x 64
    int *sabort = outi + 2 * NGP_BLK;
    const int nb = A.t1 - A.t0;
    const size_t bsz = NGP_BLK * NGP_BLK;
    if (tid == 0) *sabort = 0;
    // prologue: diagonal Gram block of local block 0 (all waves) and its group sums (wave 2)
    {
        const double *gd = A.gramx + ((size_t)A.t0 * D + 0) * bsz;
        for (int idx = tid; idx < 4096; idx += NGP_WG) Gd[idx] = gd[idx];
    }
    __syncthreads();
    if (wv == 2) {
        double tot;
        if (fetch_group_sums(A, 0, j, &tot)) r0[j] = tot;
        else if (j == 0) *sabort = 1;
    }
    CoefRegs cur = {0, 0, 0, 0, 1}, nxt = {0, 0, 0, 0, 1};
    if (wv == 0) cur = load_coef(A, (long long)A.t0 * NGP_BLK + j);
    __syncthreads();
    if (*sabort) return;
    for (int u = 0; u < nb; ++u) {
        const int t = A.t0 + u, buf = u & 1, slot = u % NGP_RING;
        if (wv == 0) {
            // ---------------- critical wave: LDS + ALU only ----------------
            if (u + 1 < nb) nxt = load_coef(A, (long long)(t + 1) * NGP_BLK + j);
            if (A.dbg && j == 0) A.dbg[4 * (size_t)u] = wall_clock64();
            double tot = r0[buf * NGP_BLK + j];
            bool okc = true;
            if (D == 1 && u >= 1) okc = fetch_group_sums(A, u, j, &tot);  // lag 1: nothing can be fetched ahead
            if (!okc && j == 0) *sabort = 1;
            const bool have_far = (D >= 3) && (u >= 2);
            const bool have_one = (D >= 2) && (u >= 1);
            double cor = have_far ? vacc[slot * NGP_BLK + j] : 0.0;
            if (have_one) {
                const double *gx = Gx + buf * 4096;
                const double *dp = hist + ((u - 1) % NGP_RING) * NGP_BLK;
                double v1 = gemv4([&](int kk) { return gx[kk * NGP_BLK + j]; }, dp);
                cor = have_far ? cor + v1 : v1;
            }
            if (have_far || have_one) tot = tot - cor;
            const double *gdb = Gd + buf * 4096;
            const double bo = cur.bo, cc = cur.cc, ww = cur.ww, qq = cur.qq, TT = cur.TT;
            double r = __builtin_fma(gdb[j * NGP_BLK + j], bo, tot);
            double Gr[NGP_BLK];
#pragma unroll
            for (int kk = 0; kk < NGP_BLK; kk++) Gr[kk] = gdb[kk * NGP_BLK + j];
            double dsave = 0.0;
            int isave = 1;
#pragma unroll
            for (int kk = 0; kk < NGP_BLK; kk++) {
                double r2 = r * r;
                double lq = r2 * qq;
                int in = lq < TT;
                double d = __builtin_fma(r, cc, ww);
                double dlv = in ? d : -bo;
                if (j == kk) {
                    dsave = dlv;
                    isave = in;
                }
                double dk = readlane_d(dlv, kk);
                r = __builtin_fma(-Gr[kk], dk, r);
            }
            hist[slot * NGP_BLK + j] = dsave;
            outb[buf * NGP_BLK + j] = bo + dsave;
            outi[buf * NGP_BLK + j] = isave;
            if (A.dbg && j == 0) A.dbg[4 * (size_t)u + 1] = wall_clock64();
            cur = nxt;
        } else if (wv == 1) {
            // ---------------- publisher: results of local block u-1 ----------------
            if (u >= 1) {
                const int up = u - 1, pslot = up % NGP_RING, pbuf = up & 1;
                const long long k = (long long)(A.t0 + up) * NGP_BLK + j;
                const double dv = hist[pslot * NGP_BLK + j];
                st_f64(&A.dlt[(size_t)pslot * NGP_BLK + j], dv);
                drain_vm();
                if (j == 0) {
                    st_u32(A.flag_dlt, (unsigned)(up + 1));
                    if (A.dbg) A.dbg[4 * (size_t)up + 2] = wall_clock64();
                }
                const double bn = outb[pbuf * NGP_BLK + j];
                const int isave = outi[pbuf * NGP_BLK + j];
                A.beta[k] = bn;
                A.delta[k] = (uint8_t)isave;
                const int si = A.setof[k];
                if (si >= 0 && A.sets[si].method == 1) {
                    double vb = 0.0;
                    if (isave) {
                        double tt = A.sets[si].sdf;
                        double b2 = bn * bn;
                        tt = tt + b2;
                        vb = tt / A.chi[k];
                        atomicAdd(&A.sets[si].nloci, 1);
                    }
                    A.varBeta[A.vbidx[k]] = vb;
                }
            }
        } else {
            // ---------------- Gram waves 2..7: units x = 0..D-1, wave gw takes x = gw, gw+6 ----------------
            const int gw = wv - 2;
            double *r0n = r0 + (buf ^ 1) * NGP_BLK;
            for (int x = gw; x < D; x += 6) {
                if (x <= 1) {  // LDS prefetch of the next block's diagonal (x=0) / lag-1 (x=1) Gram block
                    if (u + 1 < nb) {
                        const double2 *gsrc = (const double2 *)(A.gramx + ((size_t)(t + 1) * D + x) * bsz);
                        double2 *gdst = (double2 *)((x == 0 ? Gd : Gx) + (buf ^ 1) * 4096);
                        double2 tmp[32];
#pragma unroll
                        for (int i = 0; i < 32; i++) tmp[i] = gsrc[i * 64 + j];
#pragma unroll
                        for (int i = 0; i < 32; i++) gdst[i * 64 + j] = tmp[i];
                    }
                } else if (u >= 1) {  // far correction with dlt of local block a = u-1 for target a + x
                    const int a = u - 1, upb = a + x;
                    if (upb < nb) {
                        const double *gx = A.gramx + ((size_t)(A.t0 + upb) * D + x) * bsz;
                        const double *dp = hist + (a % NGP_RING) * NGP_BLK;
                        double gr[NGP_BLK];
#pragma unroll
                        for (int kk = 0; kk < NGP_BLK; kk++) gr[kk] = gx[kk * NGP_BLK + j];
                        double v = gemv4([&](int kk) { return gr[kk]; }, dp);
                        double *va = vacc + (upb % NGP_RING) * NGP_BLK + j;
                        const bool first = (x == D - 1) || (a == 0);
                        *va = first ? v : *va + v;
                    }
                }
            }
            if (gw == 0 && u + 1 < nb && D >= 2) {  // group sums of the next block -> r0[next]
                double tot;
                if (fetch_group_sums(A, u + 1, j, &tot)) r0n[j] = tot;
                else if (j == 0) *sabort = 1;
            }
        }
        __syncthreads();
        if (*sabort) return;
    }
    // publish the last block
    if (wv == 1 && nb >= 1) {
        const int up = nb - 1, pslot = up % NGP_RING, pbuf = up & 1;
        const long long k = (long long)(A.t0 + up) * NGP_BLK + j;
        st_f64(&A.dlt[(size_t)pslot * NGP_BLK + j], hist[pslot * NGP_BLK + j]);
        drain_vm();
        if (j == 0) st_u32(A.flag_dlt, (unsigned)(up + 1));
        const double bn = outb[pbuf * NGP_BLK + j];
        const int isave = outi[pbuf * NGP_BLK + j];
        A.beta[k] = bn;
        A.delta[k] = (uint8_t)isave;
        const int si = A.setof[k];
        if (si >= 0 && A.sets[si].method == 1) {
            double vb = 0.0;
            if (isave) {
                double tt = A.sets[si].sdf;
                double b2 = bn * bn;
                tt = tt + b2;
                vb = tt / A.chi[k];
                atomicAdd(&A.sets[si].nloci, 1);
            }
            A.varBeta[A.vbidx[k]] = vb;
        }
    }
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NGP_WG) void k_sweep(SweepArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    if (b == 0)
        role_sampler(A, smem);
    else if (b <= A.NG)
        role_reducer(A, b - 1);
    else
        role_streamer(A, b - 1 - A.NG, smem);
}

}  // namespace ngp
