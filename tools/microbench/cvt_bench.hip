// Cost of turning a genotype byte into the fp64 operand of an fma (compact storage, ngp_sweep.h variant 3).
// One workgroup of 512 threads (2 waves per SIMD), every variant: N iterations of 16 elements (one uint4) per lane.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o cvt_bench cvt_bench.hip ; prints SIMD cycles per element.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int V>
__global__ __launch_bounds__(512) void k(const uint4 *in, const double *yv, double *out, int n, long long *cyc) {
    const int tid = threadIdx.x;
    uint4 x = in[tid];
    double y[4] = {yv[0], yv[1], yv[2], yv[3]};
    double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    const double magic = 4503599627370496.0;  // 2^52
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; it++) {
        unsigned w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            unsigned ww = w[e];
            double d0, d1, d2, d3;
            if (V == 0) {  // and/bfe + cvt_f64_u32
                d0 = (double)(ww & 0xffu); d1 = (double)((ww >> 8) & 0xffu); d2 = (double)((ww >> 16) & 0xffu); d3 = (double)(ww >> 24);
            } else if (V == 1) {  // cvt_f32_ubyteN + cvt_f64_f32
                float f0, f1, f2, f3;
                asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(f0) : "v"(ww));
                asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(f1) : "v"(ww));
                asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(f2) : "v"(ww));
                asm volatile("v_cvt_f32_ubyte3 %0, %1" : "=v"(f3) : "v"(ww));
                d0 = (double)f0; d1 = (double)f1; d2 = (double)f2; d3 = (double)f3;
            } else if (V == 2) {  // magic: {0x43300000, byte} - 2^52
                d0 = __hiloint2double(0x43300000, (int)(ww & 0xffu)) - magic;
                d1 = __hiloint2double(0x43300000, (int)((ww >> 8) & 0xffu)) - magic;
                d2 = __hiloint2double(0x43300000, (int)((ww >> 16) & 0xffu)) - magic;
                d3 = __hiloint2double(0x43300000, (int)(ww >> 24)) - magic;
            } else if (V == 3) {  // fp32 tile element: cvt_f64_f32 only
                d0 = (double)__uint_as_float(ww); d1 = (double)__uint_as_float(ww + 1); d2 = (double)__uint_as_float(ww + 2); d3 = (double)__uint_as_float(ww + 3);
            } else {  // no conversion at all
                d0 = __hiloint2double((int)ww, 1); d1 = __hiloint2double((int)ww, 2); d2 = __hiloint2double((int)ww, 3); d3 = __hiloint2double((int)ww, 4);
            }
            acc0 = __builtin_fma(d0, y[0], acc0);
            acc1 = __builtin_fma(d1, y[1], acc1);
            acc2 = __builtin_fma(d2, y[2], acc2);
            acc3 = __builtin_fma(d3, y[3], acc3);
        }
        x.x += 0x01010101u; x.y ^= x.x; x.z += x.y; x.w ^= x.z;
    }
    long long t1 = __builtin_readcyclecounter();
    out[tid] = (acc0 + acc1) + (acc2 + acc3);
    if (tid == 0) *cyc = t1 - t0;
}
template <int V>
void run(const char *name, uint4 *in, double *y, double *out, long long *cyc) {
    const int n = 4096;
    hipLaunchKernelGGL(k<V>, dim3(1), dim3(512), 0, 0, in, y, out, n, cyc);
    hipLaunchKernelGGL(k<V>, dim3(1), dim3(512), 0, 0, in, y, out, n, cyc);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // 2 waves per SIMD, 16 elements per lane and iteration: SIMD cycles per wave-element = c / (n * 16 * 2)
    printf("%-28s %8.2f shader clocks per element and wave (2 waves per SIMD)\n", name, (double)c / (n * 16.0 * 2.0));
}
int main() {
    uint4 *in; double *y, *out; long long *cyc;
    hipMalloc(&in, 512 * 16); hipMalloc(&y, 64); hipMalloc(&out, 512 * 8); hipMalloc(&cyc, 8);
    hipMemset(in, 1, 512 * 16); double hy[4] = {1.5, 2.5, 3.5, 4.5}; hipMemcpy(y, hy, 32, hipMemcpyHostToDevice);
    run<0>("bfe + cvt_f64_u32 + fma", in, y, out, cyc);
    run<1>("cvt_f32_ubyte + cvt_f64_f32", in, y, out, cyc);
    run<2>("bfe + {2^52|b} - 2^52", in, y, out, cyc);
    run<3>("fp32: cvt_f64_f32 + fma", in, y, out, cyc);
    run<4>("fma only", in, y, out, cyc);
    return 0;
}
