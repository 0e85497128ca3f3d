// The serial 64-step chain of the sampler (ngp_sweep.h role_sampler, BayesPR): e_j += H_k[j] e_k for k = 0..63, H_k[j] = 0 for j <= k.
// V0: as shipped -- d_k by v_readlane (2 per step) into SGPRs, one fma for all lanes: the next step waits for readlane AND fma.
// V1: two accumulators -- `ea` serves the active row of 16 lanes (d_k by DPP row_newbcast, no SGPR round trip), `el` is the full
//     chain fed by readlanes of ea, which run ahead of it.  Same fma sequence per lane => bit-identical results.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o chain_bench chain_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__device__ inline double readlane_d(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
template <int K>
__device__ inline double bcast_row(double v) {  // lane (K & 15) of each row of 16 to the whole row
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + (K & 15), 0xF, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + (K & 15), 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int K> struct Steps {
    __device__ static inline void run(double &ea, double &el, const double *G, const double cc) {
        if constexpr (K < 64) {
            if constexpr ((K & 15) == 0 && K > 0) ea = el;
            const double H = -(cc * G[K]);
            const double dr = bcast_row<K>(ea);
            const double s = readlane_d(ea, K);
            ea = __builtin_fma(H, dr, ea);
            el = __builtin_fma(H, s, el);
            Steps<K + 1>::run(ea, el, G, cc);
        }
    }
};
template <int K> struct Steps2 {  // V2: the row step as ONE instruction (v_fmac_f64 with a DPP row broadcast of its first factor);
    // the full chain `el` consumes the readlane results NGP_Q steps late, so that nothing waits for the VALU -> SGPR -> VALU trip
    static constexpr int Q = 4;
    __device__ static inline void run(double &ea, double &el, const double *G, const double cc, double (&sq)[Q], double (&hq)[Q]) {
        if constexpr (K < 64) {
            if constexpr ((K & 15) == 0 && K > 0) {  // drain: el must be complete before it seeds the next row
#pragma unroll
                for (int i = 0; i < Q; i++) el = __builtin_fma(hq[(K + i) % Q], sq[(K + i) % Q], el);
                ea = el;
            } else if constexpr (K >= Q) {
                if constexpr (((K - Q) >> 4) == (K >> 4)) el = __builtin_fma(hq[K % Q], sq[K % Q], el);
            }
            const double H = -(cc * G[K]);
            sq[K % Q] = readlane_d(ea, K);
            hq[K % Q] = H;
            asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(ea) : "v"(H), "n"(K & 15));
            Steps2<K + 1>::run(ea, el, G, cc, sq, hq);
        } else {
#pragma unroll
            for (int i = 0; i < Q; i++) el = __builtin_fma(hq[(K + i) % Q], sq[(K + i) % Q], el);
        }
    }
};
template <int K> struct Steps3 {  // V3: two steps per readlane trip -- every lane also carries its right neighbour's value
    // (en = e_{j+1}, kept up to date with the neighbour's own coefficients), so lane k forms d_{k+1} itself and both d_k, d_{k+1}
    // leave lane k in ONE trip.  Same fma sequence per lane => bit-identical.
    __device__ static inline void run(double &e, double &en, const double *G, const double *Gn, const double cc, const double ccn) {
        if constexpr (K < 64) {
            const double Hk = -(cc * G[K]), Hk1 = -(cc * G[K + 1]), Hnk = -(ccn * Gn[K]), Hnk1 = -(ccn * Gn[K + 1]);
            const double t = __builtin_fma(Hnk, e, en);  // in lane K: e_{K+1} after step K = d_{K+1}
            const double dk = readlane_d(e, K), dk1 = readlane_d(t, K);
            e = __builtin_fma(Hk, dk, e);
            en = __builtin_fma(Hnk, dk, en);
            e = __builtin_fma(Hk1, dk1, e);
            en = __builtin_fma(Hnk1, dk1, en);
            Steps3<K + 2>::run(e, en, G, Gn, cc, ccn);
        }
    }
};
template <int V>
__global__ __launch_bounds__(64) void k(const double *Gm, const double *e0, const double *c0, double *out, int n, long long *cyc) {
    const int j = threadIdx.x;
    double G[64];
#pragma unroll
    for (int kk = 0; kk < 64; kk++) G[kk] = (j > kk) ? Gm[kk * 64 + j] : 0.0;
    const double cc = c0[j];
    double e = e0[j];
    double Gn[64];
#pragma unroll
    for (int kk = 0; kk < 64; kk++) Gn[kk] = (V == 3 && j + 1 < 64 && j + 1 > kk) ? Gm[kk * 64 + j + 1] : 0.0;
    const double ccn = (j + 1 < 64) ? c0[j + 1] : 0.0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; it++) {
        if (V == 0) {
            double H0 = -(cc * G[0]), H1 = -(cc * G[1]), H2 = -(cc * G[2]), H3 = -(cc * G[3]);
#pragma unroll
            for (int kk = 0; kk < 64; kk += 4) {
                double dk;
                dk = readlane_d(e, kk + 0); e = __builtin_fma(H0, dk, e); H0 = -(cc * G[(kk + 4) & 63]);
                dk = readlane_d(e, kk + 1); e = __builtin_fma(H1, dk, e); H1 = -(cc * G[(kk + 5) & 63]);
                dk = readlane_d(e, kk + 2); e = __builtin_fma(H2, dk, e); H2 = -(cc * G[(kk + 6) & 63]);
                dk = readlane_d(e, kk + 3); e = __builtin_fma(H3, dk, e); H3 = -(cc * G[(kk + 7) & 63]);
            }
        } else if (V == 1) {
            double ea = e, el = e;
            Steps<0>::run(ea, el, G, cc);
            e = el;
        } else if (V == 3) {
            double en = __shfl_down(e, 1);
            if (j == 63) en = 0.0;
            Steps3<0>::run(e, en, G, Gn, cc, ccn);
        } else {
            double ea = e, el = e;
            double sq[4] = {0, 0, 0, 0}, hq[4] = {0, 0, 0, 0};
            Steps2<0>::run(ea, el, G, cc, sq, hq);
            e = el;
        }
        e = e * 0.5 + e0[j];  // keep the values bounded; same for both variants
    }
    long long t1 = __builtin_readcyclecounter();
    out[j] = e;
    if (j == 0) *cyc = t1 - t0;
}
int main() {
    double hG[4096], he[64], hc[64], o0[64], o1[64];
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / 16777216.0 - 0.5; };
    for (auto &v : hG) v = 0.05 * rnd();
    for (int i = 0; i < 64; i++) { he[i] = rnd(); hc[i] = 0.5 + 0.1 * rnd(); }
    double *G, *e, *c, *out; long long *cyc;
    hipMalloc(&G, sizeof hG); hipMalloc(&e, 512); hipMalloc(&c, 512); hipMalloc(&out, 512); hipMalloc(&cyc, 8);
    hipMemcpy(G, hG, sizeof hG, hipMemcpyHostToDevice); hipMemcpy(e, he, 512, hipMemcpyHostToDevice); hipMemcpy(c, hc, 512, hipMemcpyHostToDevice);
    const int n = 2000; long long cy;
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, G, e, c, out, n, cyc);
    hipDeviceSynchronize(); hipMemcpy(&cy, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(o0, out, 512, hipMemcpyDeviceToHost);
    printf("V0 readlane chain      : %7.1f clocks per block of 64 steps (%.1f per step)\n", (double)cy / n, (double)cy / n / 64);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, G, e, c, out, n, cyc);
    hipDeviceSynchronize(); hipMemcpy(&cy, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(o1, out, 512, hipMemcpyDeviceToHost);
    printf("V1 row broadcast + tail: %7.1f clocks per block of 64 steps (%.1f per step)\n", (double)cy / n, (double)cy / n / 64);
    printf("bit-identical: %s\n", memcmp(o0, o1, 512) == 0 ? "yes" : "NO");
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, G, e, c, out, n, cyc);
    hipDeviceSynchronize(); hipMemcpy(&cy, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(o1, out, 512, hipMemcpyDeviceToHost);
    printf("V2 fmac_dpp + tail     : %7.1f clocks per block of 64 steps (%.1f per step)\n", (double)cy / n, (double)cy / n / 64);
    printf("bit-identical: %s\n", memcmp(o0, o1, 512) == 0 ? "yes" : "NO");
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, G, e, c, out, n, cyc);
    hipDeviceSynchronize(); hipMemcpy(&cy, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(o1, out, 512, hipMemcpyDeviceToHost);
    printf("V3 two steps per trip  : %7.1f clocks per block of 64 steps (%.1f per step)\n", (double)cy / n, (double)cy / n / 64);
    printf("bit-identical: %s\n", memcmp(o0, o1, 512) == 0 ? "yes" : "NO");
    return 0;
}
