// Micro-benchmark (one wave, dependent chain, like the panel kernel's best wave): cycles of eig3_top_fast, of its
// trigonometric root alone, of the DPP wave reductions and of the w = u^T row product.  hipcc --offload-arch=gfx950 -O3 -I../animsnapbases_amd/csrc -I../include
#include <hip/hip_runtime.h>
#include <cstdio>
#include "asb_common.h"
#include "asb_kernels.h"
__global__ void k(double* out, long long* cyc, int reps) {
    double a00 = 2000.0 + threadIdx.x * 0.0, a01 = 31.0, a02 = -17.0, a11 = 1950.0, a12 = 44.0, a22 = 2075.0;
    double lam = 0, u0 = 0, u1 = 0, u2 = 0, acc = 0;
    long long t0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
        eig3_top_fast(a00 + acc * 1e-30, a01, a02, a11, a12, a22, lam, u0, u1, u2);
        acc += lam + u0;
    }
    long long t1 = wall_clock64();
    double acc2 = 0;
    for (int r = 0; r < reps; ++r) {
        double rr = 0.3 + acc2 * 1e-30;
        acc2 += cos(acos(rr) / 3.0);
    }
    long long t2 = wall_clock64();
    double g[6] = {a00, a01, a02, a11, a12, a22};
    for (int r = 0; r < reps; ++r) {
        wave_sum_dpp<6>(g);
#pragma unroll
        for (int q = 0; q < 6; ++q) g[q] = g[q] * 1e-3 + 1.0;
    }
    long long t3 = wall_clock64();
    float accf = 0;
    for (int r = 0; r < reps; ++r) {
        float rr = 0.3f + accf * 1e-30f;
        accf += cosf(acosf(rr) / 3.0f);
    }
    long long t4 = wall_clock64();
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; }
    out[threadIdx.x] = acc + acc2 + g[0] + accf;
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 64);
    const int reps = 2000;
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, reps);
    long long h[4]; hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    printf("per call (100 MHz clock -> ns): eig3_top_fast %.0f ns | f64 cos(acos/3) %.0f ns | wave_sum_dpp<6> %.0f ns | f32 cosf(acosf/3) %.0f ns\n",
           h[0] * 10.0 / reps, h[1] * 10.0 / reps, h[2] * 10.0 / reps, h[3] * 10.0 / reps);
    return 0;
}
