// dma_bench.hip -- how should one CU stream its share of the panel?  (diagnostic, not product)
// 246 workgroups x 512 threads, one per CU; workgroup s streams tiles (t, s), t = 0..nb-1, each NQ KiB contiguous,
// into an LDS ring with global_load_lds_dwordx4; nothing is computed.  Variants:
//   0 burst:       per block: issuers wait vmcnt(0), request the whole next tile (split over NI issuer waves), barrier
//   1 continuous:  per block: each issuer requests its share of [tile u+1 quads H.., tile u+2 quads 0..H), waits until at
//                  most its share of H is outstanding, barrier
//   2 free:        issuers never meet a barrier: each keeps at most W requests outstanding (pure bandwidth ceiling)
// build: hipcc --offload-arch=gfx950 -O3 -o dma_bench dma_bench.hip ; run: ./dma_bench variant NI H_or_W [NQ] [nb]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define QS 1040
__device__ inline void dma16(const void *g, const void *l) {
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) const char *)l);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "s"(m), "v"(g) : "memory");
}
// lean form: LDS address and 64-bit global base in SGPRs (wave-uniform), per-lane 32-bit offset in one VGPR
__device__ inline void dma16_s(unsigned lds_addr, const void *gbase, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(gbase) : "memory");
}
__device__ inline void bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ inline void wait_le(int n) {
#define C(i) case i: asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); break;
    switch (n) { C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(18) C(19) C(20) C(21) C(22) C(23) C(24)
                 C(25) C(26) C(27) C(28) C(29) C(30) C(31) C(32) C(40) C(48) C(56) C(63) default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#undef C
}
__global__ __launch_bounds__(512) void k(const char *tiles, int nb, int S, int NQ, int variant, int NI, int HW, int nbar, unsigned long long *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int s = blockIdx.x, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const size_t tile_bytes = (size_t)NQ * 1024;
    const int myi = wv - (8 - NI);  // issuer index, < 0: not an issuer
    unsigned long long acc = 0;
    if (variant == 0) {
        for (int u = 0; u < nb; ++u) {
            if (myi >= 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (u + 1 < nb) {
                    const char *src = tiles + ((size_t)(u + 1) * S + s) * tile_bytes + (size_t)lane * 16;
                    char *dst = smem + (size_t)((u + 1) & 1) * NQ * QS;
                    for (int q = myi; q < NQ; q += NI) dma16(src + (size_t)q * 1024, dst + (size_t)q * QS);
                }
            }
            for (int b = 0; b < nbar; b++) bar();
            acc += *(const unsigned *)(smem + (size_t)(u & 1) * NQ * QS + lane * 16);
        }
    } else if (variant == 1) {
        const int H = HW, RQ = 2 * NQ + H;
        int base = 0;
        auto wrap = [&](int p) { return p >= RQ ? p - RQ : p; };
        if (myi >= 0) {
            const char *src0 = tiles + ((size_t)s) * tile_bytes + (size_t)lane * 16;
            for (int q = myi; q < NQ; q += NI) dma16(src0 + (size_t)q * 1024, smem + (size_t)q * QS);
            for (int q = myi; q < H; q += NI) dma16(src0 + (size_t)S * tile_bytes + (size_t)q * 1024, smem + (size_t)(NQ + q) * QS);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        bar();
        for (int u = 0; u < nb; ++u) {
            int n2 = 0;
            if (myi >= 0) {
                const int base1 = wrap(base + NQ), base2 = wrap(base1 + NQ);
                if (u + 1 < nb) {
                    const char *src = tiles + ((size_t)(u + 1) * S + s) * tile_bytes + (size_t)lane * 16;
                    for (int q = H + ((NI + myi - (H % NI)) % NI); q < NQ; q += NI) dma16(src + (size_t)q * 1024, smem + (size_t)wrap(base1 + q) * QS);
                }
                if (u + 2 < nb) {
                    const char *src = tiles + ((size_t)(u + 2) * S + s) * tile_bytes + (size_t)lane * 16;
                    for (int q = myi; q < H; q += NI, ++n2) dma16(src + (size_t)q * 1024, smem + (size_t)wrap(base2 + q) * QS);
                }
                wait_le(n2);
            }
            for (int b = 0; b < nbar; b++) bar();
            acc += *(const unsigned *)(smem + (size_t)base * QS + lane * 16);
            base = wrap(base + NQ);
        }
    } else if (variant == 3) {
        // variant 1 with the lean issue sequence
        const int H = HW, RQ = 2 * NQ + H;
        int base = 0;
        const unsigned ring0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)smem;
        const unsigned voff = lane * 16;
        auto wrap = [&](int p) { return p >= RQ ? p - RQ : p; };
        auto issue = [&](const char *tile_src, int q0, int q1, int tbase) {
            int n = 0;
            int q = q0 + ((NI + myi - (q0 % NI)) % NI);
            int p = wrap(tbase + q);
            const char *g = tile_src + (size_t)q * 1024;
            for (; q < q1; q += NI, ++n) {
                dma16_s(__builtin_amdgcn_readfirstlane(ring0 + (unsigned)p * QS), (const void *)__builtin_amdgcn_readfirstlane((unsigned)(size_t)g) ? g : g, voff);
                p += NI; if (p >= RQ) p -= RQ;
                g += (size_t)NI * 1024;
            }
            return n;
        };
        if (myi >= 0) {
            const char *src0 = tiles + ((size_t)s) * tile_bytes;
            issue(src0, 0, NQ, 0);
            issue(src0 + (size_t)S * tile_bytes, 0, H, NQ);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        bar();
        for (int u = 0; u < nb; ++u) {
            int n2 = 0;
            if (myi >= 0) {
                const int base1 = wrap(base + NQ), base2 = wrap(base1 + NQ);
                if (u + 1 < nb) issue(tiles + ((size_t)(u + 1) * S + s) * tile_bytes, H, NQ, base1);
                if (u + 2 < nb) n2 = issue(tiles + ((size_t)(u + 2) * S + s) * tile_bytes, 0, H, base2);
                wait_le(n2);
            }
            for (int b = 0; b < nbar; b++) bar();
            acc += *(const unsigned *)(smem + (size_t)base * QS + lane * 16);
            base = wrap(base + NQ);
        }
    } else {
        // free-running: issuer i streams quads i, i+NI, ... of the whole shard stream, at most HW outstanding
        if (myi >= 0) {
            const long long total = (long long)nb * NQ;
            int ring = 0;
            const int slots = 120 / NI;  // ring slots of this issuer
            for (long long k = myi; k < total; k += NI) {
                const long long t = k / NQ; const int q = (int)(k - t * NQ);
                const char *src = tiles + ((size_t)t * S + s) * tile_bytes + (size_t)q * 1024 + (size_t)lane * 16;
                dma16(src, smem + (size_t)(myi * slots + ring) * QS);
                if (++ring == slots) ring = 0;
                wait_le(HW);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x123456789ull) sink[0] = acc;
}
int main(int argc, char **argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0, NI = argc > 2 ? atoi(argv[2]) : 3, HW = argc > 3 ? atoi(argv[3]) : 24;
    const int NQ = argc > 4 ? atoi(argv[4]) : 51, nb = argc > 5 ? atoi(argv[5]) : 4000, nbar = argc > 6 ? atoi(argv[6]) : 1;
    const int S = 246;
    const size_t bytes = (size_t)nb * S * NQ * 1024;
    char *d; unsigned long long *sink;
    if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 64);
    hipMemset(d, 1, bytes);
    const size_t lds = (size_t)(2 * NQ + 32) * QS + 1024;
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(S), dim3(512), lds, 0, d, nb, S, NQ, variant, NI, HW, nbar, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("variant %d NI %d HW %d NQ %d nbar %d: %.3f ms, %.2f us/block, %.0f GB/s (%.1f%% of 8 TB/s) err=%s\n", variant, NI, HW, NQ, nbar, ms, ms * 1e3 / nb,
                             bytes / ms / 1e6, bytes / ms / 1e6 / 80.0, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
