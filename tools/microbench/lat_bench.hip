// Latency / issue cost of the instructions on the sampler's serial chain, one wave alone on its SIMD (clocks of s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ inline double readlane_d(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
template <int V>
__global__ __launch_bounds__(64) void k(const double *in, double *out, int n, long long *cyc) {
    const int j = threadIdx.x;
    double e = in[j], h = in[64 + j] * 1e-3, a0 = e, a1 = e + 1, a2 = e + 2, a3 = e + 3;
    float f = (float)e, g = (float)h;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int kk = 0; kk < 64; kk++) {
            if (V == 0) e = __builtin_fma(h, e, e);                       // dependent fp64 fma
            if (V == 1) { a0 = __builtin_fma(h, a0, a0); a1 = __builtin_fma(h, a1, a1); a2 = __builtin_fma(h, a2, a2); a3 = __builtin_fma(h, a3, a3); }  // 4 independent
            if (V == 2) f = __builtin_fmaf(g, f, f);                      // dependent fp32 fma
            if (V == 3) { double d = readlane_d(e, kk); e = e + d * 1e-9; }  // readlane x2 -> add (dependent through e)
            if (V == 4) { double d = readlane_d(e, kk); e = __builtin_fma(h, d, e); }  // the shipped step without the H product
            if (V == 6) {  // 2 x v_mov_b32_dpp row_newbcast + fma (dependent)
                int lo = __builtin_amdgcn_update_dpp(0, __double2loint(e), 0x153, 0xF, 0xF, false);
                int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(e), 0x153, 0xF, 0xF, false);
                e = __builtin_fma(h, __hiloint2double(hi, lo), e);
            }
            if (V == 7) {  // 1 x v_mov_b32_dpp quad_perm + 32-bit op (dependent)
                int lo = __builtin_amdgcn_update_dpp(0, __double2loint(e), 0xB1, 0xF, 0xF, false);
                e = __hiloint2double(__double2hiint(e), lo ^ __double2loint(e));
            }
            if (V == 8) {  // v_mov_b64_dpp row_newbcast (one DP DPP move) + fma
                double d;
                asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "=v"(d) : "v"(e));
                e = __builtin_fma(h, d, e);
            }
            if (V == 5) { int lo = __builtin_amdgcn_readlane(__double2loint(e), kk); e = __hiloint2double(__double2hiint(e), lo ^ __double2loint(e)); }  // readlane -> 32-bit op
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[j] = e + a0 + a1 + a2 + a3 + (double)f;
    if (j == 0) *cyc = t1 - t0;
}
template <int V> void run(const char *name, double *in, double *out, long long *cyc, double per) {
    const int n = 2000; long long cy;
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, in, out, n, cyc);
    (void)hipDeviceSynchronize(); (void)hipMemcpy(&cy, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s %6.2f clocks\n", name, (double)cy / n / 64 / per);
}
int main() {
    double h[128]; for (int i = 0; i < 128; i++) h[i] = 0.001 * (i + 1);
    double *in, *out; long long *cyc;
    (void)hipMalloc(&in, sizeof h); (void)hipMalloc(&out, 512); (void)hipMalloc(&cyc, 8); (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    run<0>("dependent v_fma_f64", in, out, cyc, 1);
    run<1>("independent v_fma_f64 (per instruction)", in, out, cyc, 4);
    run<2>("dependent v_fma_f32", in, out, cyc, 1);
    run<3>("2 x v_readlane + v_fma_f64 (add form)", in, out, cyc, 1);
    run<4>("2 x v_readlane + v_fma_f64 (chain step)", in, out, cyc, 1);
    run<5>("v_readlane + 32-bit op", in, out, cyc, 1);
    run<6>("2 x v_mov_b32_dpp row_newbcast + v_fma_f64", in, out, cyc, 1);
    run<7>("v_mov_b32_dpp quad_perm + 32-bit op", in, out, cyc, 1);
    run<8>("v_mov_b64_dpp row_newbcast + v_fma_f64", in, out, cyc, 1);
    return 0;
}
