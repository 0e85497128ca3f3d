// Hand-off loop of the persistent sweep without the panel: S producer workgroups publish 64 values per block, ONE consumer needs
// their sum before it releases the producers LAG blocks later (the sampler's dlt).  Two forms of the reduction:
//   A  (round 3)  producers store doubles + count per group of 32; NG reducer workgroups add 32 partials in order, store the group
//                 sum + count; the consumer polls that counter and adds NG group sums          -- two cross-CU hops
//   B  (round 4)  producers atomicAdd fixed-point int64 into 8 accumulator copies + count; the consumer polls the 8 counters and
//                 adds the 8 copies (integer addition: order-free, deterministic)               -- one hop
// Prints microseconds per block for LAG = 1 (the loop's latency) and LAG = 6 (its throughput with six blocks in flight).
//   hipcc --offload-arch=gfx950 -O3 -o hop_bench hop_bench.hip && ./hop_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define RING 16
#define SPIN (1u << 22)
__device__ inline unsigned ld_u32(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_u32(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline double ld_f64(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ inline void st_f64(double *p, double v) {
    __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void drain_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ inline bool wait_ge(const unsigned *f, unsigned target, unsigned *abortw) {
    for (unsigned s = 0;; ++s) {
        if (ld_u32(f) >= target) return true;
        if ((s & 63u) == 63u && ld_u32(abortw) != 0u) return false;
        if (s > SPIN) { st_u32(abortw, 1u); return false; }
        __builtin_amdgcn_s_sleep(2);
    }
}
struct Args {
    int S, NG, NB, LAG, form;
    double *part, *gsum;            // A: [RING][S][64], [RING][NG][64]
    unsigned long long *acc;        // B: [RING][8][64] cumulative fixed-point sums
    unsigned *cnt_part, *cnt_gs;    // A: [RING][NG] x 32 words, [RING] x 32 words;  B: cnt_part = [RING][8] x 32 words
    unsigned *flag, *abortw;        // blocks the consumer has finished
    double *out;
    long long *cyc;
};
// exact round-to-nearest conversion of x (|x| < 2^62) to int64: rint, then split into an exact high and low half
__device__ inline long long f64_to_i64_rn(double x) {
    const double r = __builtin_rint(x);
    const double hi = __builtin_floor(r * 0x1p-32);
    const double lo = __builtin_fma(-hi, 0x1p32, r);
    return (long long)(((unsigned long long)(unsigned)(int)hi << 32) | (unsigned long long)(unsigned)lo);
}
__device__ inline double i64_to_f64(long long q) {
    const double hi = (double)(int)(q >> 32), lo = (double)(unsigned)q;
    return __builtin_fma(hi, 0x1p32, lo);
}
__global__ __launch_bounds__(512) void k(Args A) {
    const int b = blockIdx.x, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    if (b == 0) {  // consumer: wave 0 only
        if (wv != 0) return;
        __shared__ long long prev[RING][64];
        for (int i = 0; i < RING; i++) prev[i][lane] = 0;
        double sink = 0.0;
        const long long t0 = wall_clock64();
        for (int u = 0; u < A.NB; u++) {
            const int slot = u % RING, round = u / RING;
            double tot;
            if (A.form == 0) {
                int ok = 1;
                if (lane == 0) ok = wait_ge(&A.cnt_gs[slot * 32], (unsigned)((round + 1) * A.NG), A.abortw);
                if (!__builtin_amdgcn_readfirstlane(ok)) return;
                double gv[8];
#pragma unroll
                for (int g = 0; g < 8; g++) gv[g] = ld_f64(A.gsum + ((size_t)slot * A.NG + min(g, A.NG - 1)) * 64 + lane);
                tot = gv[0];
#pragma unroll
                for (int g = 1; g < 8; g++) if (g < A.NG) tot += gv[g];
            } else {
                // lanes 0..7 poll the 8 counters together
                const unsigned per = (unsigned)((A.S + 7 - (lane & 7)) / 8);  // producers of copy lane&7
                for (unsigned s = 0;; ++s) {
                    const unsigned v = ld_u32(&A.cnt_part[(slot * 8 + (lane & 7)) * 32]);
                    if (__ballot(v >= (unsigned)(round + 1) * per) == ~0ull) break;
                    if ((s & 63u) == 63u && ld_u32(A.abortw) != 0u) return;
                    if (s > SPIN) { st_u32(A.abortw, 2u); return; }
                    __builtin_amdgcn_s_sleep(2);
                }
                unsigned long long qv[8];
#pragma unroll
                for (int c = 0; c < 8; c++) qv[c] = __hip_atomic_load(A.acc + ((size_t)slot * 8 + c) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned long long q = 0;
#pragma unroll
                for (int c = 0; c < 8; c++) q += qv[c];
                long long cur = (long long)q;
                long long d = cur - prev[slot][lane];
                prev[slot][lane] = cur;
                tot = i64_to_f64(d) * 0x1p-20;
            }
            sink += tot;
            if (lane == 0) st_u32(A.flag, (unsigned)(u + 1));
        }
        const long long t1 = wall_clock64();
        A.out[lane] = sink;
        if (lane == 0) *A.cyc = t1 - t0;
        return;
    }
    if (A.form == 0 && b <= A.NG) {  // reducers: wave w takes blocks w, w+8, ..
        const int g = b - 1, s0 = g * 32, gsize = min(32, A.S - s0);
        for (int u = wv; u < A.NB; u += 8) {
            const int slot = u % RING, round = u / RING;
            int ok = 1;
            if (lane == 0) ok = wait_ge(&A.cnt_part[(slot * A.NG + g) * 32], (unsigned)((round + 1) * gsize), A.abortw);
            if (!__builtin_amdgcn_readfirstlane(ok)) return;
            double v[32];
#pragma unroll
            for (int s = 0; s < 32; s++) v[s] = ld_f64(A.part + ((size_t)slot * A.S + s0 + min(s, gsize - 1)) * 64 + lane);
            double t = v[0];
#pragma unroll
            for (int s = 1; s < 32; s++) if (s < gsize) t += v[s];
            st_f64(A.gsum + ((size_t)slot * A.NG + g) * 64 + lane, t);
            drain_vm();
            if (lane == 0) atomicAdd(&A.cnt_gs[slot * 32], 1u);
        }
        return;
    }
    const int s = (A.form == 0) ? b - 1 - A.NG : b - 1;
    if (s >= A.S || wv != 0) return;
    double x = 1.0 + 1e-3 * lane + 1e-6 * s;
    for (int u = 0; u < A.NB; u++) {
        if (u >= A.LAG) {
            int ok = 1;
            if (lane == 0) ok = wait_ge(A.flag, (unsigned)(u - A.LAG + 1), A.abortw);
            if (!__builtin_amdgcn_readfirstlane(ok)) return;
        }
        const int slot = u % RING;
        x = x * 1.0000001;
        if (A.form == 0) {
            st_f64(A.part + ((size_t)slot * A.S + s) * 64 + lane, x);
            drain_vm();
            if (lane == 0) atomicAdd(&A.cnt_part[(slot * A.NG + s / 32) * 32], 1u);
        } else {
            const long long q = f64_to_i64_rn(x * 0x1p20);
            __hip_atomic_fetch_add(A.acc + ((size_t)slot * 8 + (s & 7)) * 64 + lane, (unsigned long long)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            drain_vm();
            if (lane == 0) atomicAdd(&A.cnt_part[(slot * 8 + (s & 7)) * 32], 1u);
        }
    }
}
int main(int argc, char **argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 246, NB = 4000;
    const int NG = (S + 31) / 32;
    Args A;
    A.S = S; A.NG = NG; A.NB = NB;
    (void)hipMalloc(&A.part, (size_t)RING * S * 64 * 8); (void)hipMalloc(&A.gsum, (size_t)RING * NG * 64 * 8);
    (void)hipMalloc(&A.acc, (size_t)RING * 8 * 64 * 8);
    (void)hipMalloc(&A.cnt_part, (size_t)RING * 8 * 32 * 4); (void)hipMalloc(&A.cnt_gs, (size_t)RING * 32 * 4);
    (void)hipMalloc(&A.flag, 256); (void)hipMalloc(&A.abortw, 256); (void)hipMalloc(&A.out, 512); (void)hipMalloc(&A.cyc, 8);
    A.abortw += 32;  // (own line)
    for (int form = 0; form < 2; form++)
        for (int lag : {1, 2, 4, 6, 8}) {
            A.form = form; A.LAG = lag;
            (void)hipMemset(A.acc, 0, (size_t)RING * 8 * 64 * 8); (void)hipMemset(A.cnt_part, 0, (size_t)RING * 8 * 32 * 4);
            (void)hipMemset(A.cnt_gs, 0, (size_t)RING * 32 * 4); (void)hipMemset(A.flag, 0, 256);
            const int grid = 1 + (form == 0 ? NG : 0) + S;
            hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, A);
            (void)hipDeviceSynchronize();
            long long cy; unsigned ab; double o[64];
            (void)hipMemcpy(&cy, A.cyc, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&ab, A.abortw, 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(o, A.out, 512, hipMemcpyDeviceToHost);
            printf("form %c  S=%d  lag %d: %.3f us per block%s  (sum lane 0 %.6f)\n", form ? 'B' : 'A', S, lag, (double)cy / 100.0 / NB, ab ? "  ABORTED" : "", o[0] / NB);
        }
    return 0;
}
