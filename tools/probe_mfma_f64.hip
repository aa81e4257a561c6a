// Probe: v_mfma_f64_16x16x4_f64 operand/result layout (exact integer data) and issue rate on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_f64.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(const double* A /*16x4*/, const double* B /*4x16*/, double* C /*16x16*/) {
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];      // A[i = l&15][k = l>>4]
    const double b = B[(l >> 4) * 16 + (l & 15)];     // B[k = l>>4][j = l&15]
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];   // row=(l>>4)+4r, col=l&15
}

__global__ void k_rate(double* out, int iters) {
    const int l = threadIdx.x & 63;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

__global__ void k_rate_fma(double* out, int iters) {
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = fma(a, b, c[j]);
    double s = 0;
    for (int j = 0; j < 8; ++j) s += c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// both pipes at once: waves 0..3 of an 8-wave block issue MFMAs, waves 4..7 v_fma_f64 (one of each per SIMD); n_fma FMA groups
// of 8 per 4 MFMAs.  Is the f64 throughput of a CU the matrix pipe's, or matrix + vector?
__global__ __launch_bounds__(512) void k_rate_both(double* out, int iters, int fma_per_iter) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    if (w < 4) {
        d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < iters; ++i)
            for (int q = 0; q < fma_per_iter; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) c[j] = fma(a, b, c[j]);
        double s = 0;
        for (int j = 0; j < 8; ++j) s += c[j];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

int main() {
    std::vector<double> A(64), B(64), C(256), Cref(256, 0.0);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i * 7 + k * 3;      // asymmetric integers
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 2 + k * 11 - j * 5;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) Cref[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dC;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 2048);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    k_layout<<<1, 64>>>(dA, dB, dC);
    hipMemcpy(C.data(), dC, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += (C[i] != Cref[i]);
    printf("layout check: %d mismatches of 256\n", bad);

    double* out; hipMalloc(&out, 256 * 4 * 256 * 8 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpb = 4; wpb <= 8; wpb += 4) {     // waves per block = waves per CU (1 block/CU)
        const int iters = 20000, blocks = 256;
        k_rate<<<blocks, 64 * wpb>>>(out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0); k_rate<<<blocks, 64 * wpb>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * wpb * iters * 4 * 2048.0;
        printf("mfma f64 16x16x4: %d waves/CU: %.1f TFLOP/s (%.2f ms)\n", wpb, flops / ms * 1e-9, ms);
    }
    for (int wpb = 4; wpb <= 16; wpb *= 2) {
        const int iters = 20000, blocks = 256;
        k_rate_fma<<<blocks, 64 * wpb>>>(out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0); k_rate_fma<<<blocks, 64 * wpb>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * wpb * 64 * iters * 8 * 2.0;
        printf("v_fma_f64: %d waves/CU: %.1f TFLOP/s (%.2f ms)\n", wpb, flops / ms * 1e-9, ms);
    }
    for (int fpi = 0; fpi <= 8; fpi += 2) {
        const int iters = 20000, blocks = 256;
        k_rate_both<<<blocks, 512>>>(out, 100, fpi);
        hipDeviceSynchronize();
        hipEventRecord(e0); k_rate_both<<<blocks, 512>>>(out, iters, fpi); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fm = (double)blocks * 4 * iters * 4 * 2048.0, fv = (double)blocks * 4 * 64 * iters * fpi * 8 * 2.0;
        printf("both pipes, %d x 8 v_fma_f64 per 4 MFMAs: matrix %.1f + vector %.1f = %.1f TFLOP/s (%.2f ms)\n", fpi, fm / ms * 1e-9,
               fv / ms * 1e-9, (fm + fv) / ms * 1e-9, ms);
    }
    return bad != 0;
}
