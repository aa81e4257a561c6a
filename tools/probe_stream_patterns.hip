// Probe: HBM read rate of the MFMA-A-operand access pattern (16 rows x 128 B per wave instruction,
// rows Fp*8 bytes apart) versus a fully contiguous stream, on a 4.8 GB tensor (300000 rows x 2000 f64).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// contiguous: every lane reads 32 B, wave covers 2 KB contiguous, grid-stride
__global__ __launch_bounds__(256) void k_contig(const double4* __restrict__ X, long long n4, double* out) {
    double s = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const double4 v = X[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678) out[0] = s;
}

// tile pattern: wave handles 16 rows; per "chunk" lane (i = l&15, g = l>>4) reads 32 B at row i, frames 16c+4g;
// G chunks are issued back to back before use; frames [f0, f0+nf)
template <int G>
__global__ __launch_bounds__(1024) void k_tile(const double* __restrict__ X, long long rows, int Fp, int f0, int nf,
                                               unsigned* counter, double* out) {
    const int l = threadIdx.x & 63, i = l & 15, g = l >> 4;
    const long long ntiles = rows / 16;
    double s = 0;
    for (;;) {
        unsigned t = 0;
        if (l == 0) t = atomicAdd(counter, 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= ntiles) break;
        const double4* xp = reinterpret_cast<const double4*>(X + ((long long)t * 16 + i) * Fp + f0 + 4 * g);
        const int nchunk = nf / 16;
        for (int c = 0; c + G <= nchunk; c += G) {
            double4 v[G];
#pragma unroll
            for (int q = 0; q < G; ++q) v[q] = xp[4 * (c + q)];
#pragma unroll
            for (int q = 0; q < G; ++q) s += v[q].x + v[q].y + v[q].z + v[q].w;
        }
    }
    if (s == 12345.678) out[0] = s;
}

typedef double d4 __attribute__((ext_vector_type(4)));
// same tile pattern, double-buffered groups of 4 chunks, plus NM f64 MFMAs per chunk (B from a register)
template <int NM>
__global__ __launch_bounds__(512) void k_tile_mfma(const double* __restrict__ X, long long rows, int Fp, unsigned* counter, double* out) {
    const int l = threadIdx.x & 63, i = l & 15, g = l >> 4;
    const long long ntiles = rows / 16;
    d4 acc = {0, 0, 0, 0};
    const double b = 1.0 + l * 1e-3;
    for (;;) {
        unsigned t = 0;
        if (l == 0) t = atomicAdd(counter, 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= ntiles) break;
        const double4* xp = reinterpret_cast<const double4*>(X + ((long long)t * 16 + i) * Fp + 4 * g);
        const int nchunk = Fp / 16;
        double4 a0 = xp[0], a1 = xp[4], a2 = xp[8], a3 = xp[12];
        for (int c = 0; c + 8 <= nchunk; c += 4) {
            const double4 n0 = xp[4 * (c + 4)], n1 = xp[4 * (c + 5)], n2 = xp[4 * (c + 6)], n3 = xp[4 * (c + 7)];
#define USE(v) { if (NM >= 1) acc = __builtin_amdgcn_mfma_f64_16x16x4f64((v).x, b, acc, 0, 0, 0); \
                 if (NM >= 2) acc = __builtin_amdgcn_mfma_f64_16x16x4f64((v).y, b, acc, 0, 0, 0); \
                 if (NM >= 4) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64((v).z, b, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64((v).w, b, acc, 0, 0, 0); } \
                 if (NM == 0) acc[0] += (v).x + (v).y + (v).z + (v).w; else if (NM < 4) acc[1] += (v).z + (v).w + (NM < 2 ? (v).y : 0.0); }
            USE(a0) USE(a1) USE(a2) USE(a3)
            a0 = n0; a1 = n1; a2 = n2; a3 = n3;
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678) out[0] = acc[0];
}

int main() {
    const long long rows = 300000; const int Fp = 2000;
    double* X; double* out; unsigned* cnt;
    CK(hipMalloc(&X, rows * Fp * 8)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&cnt, 64));
    CK(hipMemset(X, 0, rows * Fp * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto report = [&](const char* name, float ms, double bytes) { printf("%-44s %.3f ms  %.0f GB/s\n", name, ms, bytes / ms * 1e-6); };
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0)); k_contig<<<2048, 256>>>((const double4*)X, rows * Fp / 4, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); report("contiguous 32B/lane", ms, rows * Fp * 8.0);
    }
#define RUN(G, F0, NF, LABEL) { CK(hipMemset(cnt, 0, 64)); CK(hipEventRecord(e0)); k_tile<G><<<256, 1024>>>(X, rows, Fp, F0, NF, cnt, out); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); report(LABEL, ms, rows * (double)(NF) * 8.0); }
    for (int rep = 0; rep < 2; ++rep) {
        RUN(4, 0, 1008, "tile G=4  half rows (1008 frames)")
        RUN(8, 0, 1008, "tile G=8  half rows")
        RUN(16, 0, 1008, "tile G=16 half rows (nchunk 63 -> 48 used)")
        RUN(4, 0, 1984, "tile G=4  full rows (1984 frames)")
        RUN(8, 0, 1984, "tile G=8  full rows")
        RUN(16, 0, 1984, "tile G=16 full rows (112 chunks)")
    }
#define RUNM(NM, LABEL) { CK(hipMemset(cnt, 0, 64)); CK(hipEventRecord(e0)); k_tile_mfma<NM><<<512, 512>>>(X, rows, Fp, cnt, out); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); report(LABEL, ms, rows * (double)Fp * 8.0); }
    for (int rep = 0; rep < 2; ++rep) {
        RUNM(0, "tile dbuf, no MFMA")
        RUNM(1, "tile dbuf, 1 MFMA / chunk")
        RUNM(2, "tile dbuf, 2 MFMA / chunk")
        RUNM(4, "tile dbuf, 4 MFMA / chunk (as k_project)")
    }
    return 0;
}
