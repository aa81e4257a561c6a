// asb_dense.hip -- dense f64 building blocks for meshes small enough that an N x N matrix is cheap in 288 GB of HBM:
//   k_gemm_nn            C = beta C + alpha A B, row-major, LDS-tiled v_mfma_f64_16x16x4_f64 (128 x 128 tile per block)
//   asb_dense_spd_inverse in-place inverse of a symmetric positive definite matrix by blocked Gauss-Jordan sweeps
//                        (block 128: one LDS-resident block inverse + three GEMMs per sweep, 2 n^3 flop in all)
// Used by the device heat-method geodesics (asb_geodesic.hip): the two SPD systems of utils/support.py:170-171 are
// inverted once (N = 15 000: 1.75 GB each) and every later solve is a column gather / one GEMM.  gfx950 only.
#include "asb_common.h"

#include <chrono>
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));

#define DG_BM 128
#define DG_KC 16
#define DG_SA 18      // LDS row stride of the A stage  As[i][k]
#define DG_SB 144     // LDS row stride of the B stage  Bs[k][j]

// grid (tiles_n, tiles_m, S).  All of M, N, Kc, lda, ldb, ldc are multiples of 2 and the pointers 16-byte aligned
// (the callers pad to 16).  S > 1: the slab's product goes to part[z] (M x N, ld N) and k_gemm_finish combines.
__global__ __launch_bounds__(256, 2) void k_gemm_nn(const double* __restrict__ A, long long lda, const double* __restrict__ B,
                                                   long long ldb, double* __restrict__ C, long long ldc, int M, int N, int Kc,
                                                   double alpha, double beta, int slab, double* __restrict__ part, int tri, int cinit) {
    // tri: only the tiles on and above the diagonal (a symmetric rank-k update of the upper triangle)
    // cinit (C -= A B, one slab): the accumulators START as the C tile -- its loads travel beside the first operand stage -- and A is
    // negated on its way into LDS: the epilogue only stores (the rank-256 updates of the Gauss-Jordan inverse: 16 stages per tile,
    // where a read-modify-write epilogue is a fifth of the tile's time)
    if (tri && (int)blockIdx.x < (int)blockIdx.y) return;
    __shared__ double As[2][DG_BM][DG_SA];
    __shared__ double Bs[2][DG_KC][DG_SB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int wi = wave >> 1, wj = wave & 1;
    const int i0 = blockIdx.y * DG_BM, j0 = blockIdx.x * DG_BM;
    const int k_begin = blockIdx.z * slab;
    int k_end = k_begin + slab;
    if (k_end > Kc) k_end = Kc;
    // A stage: thread -> row ar, 8 consecutive k (4 double2); B stage: wave -> rows wave + 4q, lane -> columns 2 lane, 2 lane + 1
    const int ar = tid >> 1, ak = (tid & 1) * 8;
    const bool a_ok = (i0 + ar) < M;
    const int cb = j0 + 2 * lane;
    const bool b_ok = cb < N;
    double2 ra[4], rb[4];
    auto fetch = [&](int k0) {
        const double* pa = A + (long long)(i0 + ar) * lda + k0 + ak;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            ra[q] = (a_ok && k0 + ak + 2 * q < k_end) ? *reinterpret_cast<const double2*>(pa + 2 * q) : make_double2(0.0, 0.0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = k0 + wave + 4 * q;
            rb[q] = (b_ok && k < k_end) ? *reinterpret_cast<const double2*>(B + (long long)k * ldb + cb) : make_double2(0.0, 0.0);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<double2*>(&As[buf][ar][ak + 2 * q]) = cinit ? make_double2(-ra[q].x, -ra[q].y) : ra[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<double2*>(&Bs[buf][wave + 4 * q][2 * lane]) = rb[q];
    };
    d4 acc[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = d4{0.0, 0.0, 0.0, 0.0};
    if (cinit) {
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int oi = i0 + wi * 64 + x * 16 + g + 4 * q, oj = j0 + wj * 64 + y * 16 + li;
                    if (oi < M && oj < N) acc[x][y][q] = C[(long long)oi * ldc + oj];
                }
    }
    if (k_begin < k_end) {
        fetch(k_begin);
        stash(0);
    }
    __syncthreads();
    int cur = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += DG_KC) {
        const bool more = k0 + DG_KC < k_end;
        if (more) fetch(k0 + DG_KC);
#pragma unroll
        for (int ks = 0; ks < DG_KC / 4; ++ks) {
            double a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[q] = As[cur][wi * 64 + q * 16 + li][ks * 4 + g];
                b[q] = Bs[cur][ks * 4 + g][wj * 64 + q * 16 + li];
            }
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[x][y], 0, 0, 0);
        }
        if (more) stash(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    const bool split = gridDim.z > 1;
    double* o = split ? part + (long long)blockIdx.z * M * N : C;
    const long long ldo = split ? N : ldc;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int oi = i0 + wi * 64 + x * 16 + g + 4 * q, oj = j0 + wj * 64 + y * 16 + li;
                if (oi < M && oj < N) {
                    double* dst = o + (long long)oi * ldo + oj;
                    if (split || cinit) *dst = acc[x][y][q];
                    else *dst = (beta == 0.0 ? 0.0 : beta * *dst) + alpha * acc[x][y][q];
                }
            }
}

__global__ __launch_bounds__(256) void k_gemm_finish(const double* __restrict__ part, int S, int M, int N, double alpha, double beta,
                                                     double* __restrict__ C, long long ldc) {
    const long long total = (long long)M * N;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        double s = 0.0;
        for (int q = 0; q < S; ++q) s += part[(long long)q * total + e];
        double* dst = C + (e / N) * ldc + (e % N);
        *dst = (beta == 0.0 ? 0.0 : beta * *dst) + alpha * s;
    }
}

int asb_gemm_nn(asb_ctx* ctx, const double* A, long long lda, const double* B, long long ldb, double* C, long long ldc, int M,
                int N, int Kc, double alpha, double beta, int tri) {
    if ((lda | ldb | ldc | M | N | Kc) & 1) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_gemm_nn: odd dimension");
    if (tri && M != N) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_gemm_nn: the triangular form needs a square result");
    const int tm = (M + DG_BM - 1) / DG_BM, tn = (N + DG_BM - 1) / DG_BM;
    // split the contraction when the tile grid alone cannot fill the chip (skinny products with a long contraction)
    int S = 1;
    if ((long long)tm * tn < 512 && Kc >= 1024 && !tri) {
        S = (int)(1024 / ((long long)tm * tn));
        const int maxS = Kc / 512;
        if (S > maxS) S = maxS;
        if (S > 32) S = 32;
        if (S < 1) S = 1;
    }
    int slab = ((Kc + S - 1) / S + DG_KC - 1) / DG_KC * DG_KC;
    S = (Kc + slab - 1) / slab;
    if (S > 1) {
        const size_t need = (size_t)S * M * N;
        if (need > ctx->la_part_cap) {
            int rc = asb_alloc(ctx, &ctx->la_part, need);
            if (rc) return rc;
            ctx->la_part_cap = need;
        }
    }
    static const int cinit_on = getenv("ASB_GEMM_CINIT") ? atoi(getenv("ASB_GEMM_CINIT")) : 1;
    const int cinit = (cinit_on && S == 1 && alpha == -1.0 && beta == 1.0) ? 1 : 0;
    hipLaunchKernelGGL(k_gemm_nn, dim3(tn, tm, S), dim3(256), 0, ctx->stream, A, lda, B, ldb, C, ldc, M, N, Kc, alpha, beta, slab,
                       ctx->la_part, tri, cinit);
    if (S > 1) {
        const long long total = (long long)M * N;
        const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(k_gemm_finish, dim3(grid), dim3(256), 0, ctx->stream, ctx->la_part, S, M, N, alpha, beta, C, ldc);
    }
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

// in-LDS Gauss-Jordan inverse of one symmetric positive definite b x b block (b <= 128), no pivoting.
// src (ld lds) -> dst (contiguous b x b).  status[0] = 1 when a pivot is not positive.
__global__ __launch_bounds__(1024) void k_block_inverse(const double* __restrict__ src, long long lds, int b,
                                                        double* __restrict__ dst, int* __restrict__ status) {
    extern __shared__ double Sm[];       // b x (b + 1)
    __shared__ double prow[128], pcol[128];
    __shared__ int bad;
    const int tid = threadIdx.x, nt = blockDim.x, ld = b + 1;
    if (tid == 0) bad = 0;
    for (int e = tid; e < b * b; e += nt) Sm[(e / b) * ld + (e % b)] = src[(long long)(e / b) * lds + (e % b)];
    __syncthreads();
    for (int p = 0; p < b; ++p) {
        const double piv = Sm[p * ld + p];
        if (!(piv > 0.0)) {
            if (tid == 0) bad = 1;
            break;               // uniform: every thread reads the same pivot
        }
        const double d = 1.0 / piv;
        for (int j = tid; j < b; j += nt) {
            prow[j] = Sm[p * ld + j] * d;      // new row p (j != p)
            pcol[j] = Sm[j * ld + p];          // old column p
        }
        __syncthreads();
        for (int e = tid; e < b * b; e += nt) {
            const int i = e / b, j = e % b;
            double v;
            if (i == p) v = (j == p) ? d : prow[j];
            else if (j == p) v = -pcol[i] * d;
            else v = Sm[i * ld + j] - pcol[i] * prow[j];
            Sm[i * ld + j] = v;
        }
        __syncthreads();
    }
    __syncthreads();
    if (bad) {
        if (tid == 0) status[0] = 1;
        return;
    }
    for (int e = tid; e < b * b; e += nt) dst[e] = Sm[(e / b) * ld + (e % b)];
}

// M (np x np, ld np, np a multiple of 16) <- M^-1 by plain blocked Gauss-Jordan sweeps over the FULL matrix (2 np^3 flop): the
// pivot blocks of the symmetric form below and matrices of at most one such block.  Enqueues only; a pivot that is not
// positive sets ctx->la_status[0] (the caller clears and reads it).
static int spd_inverse_full(asb_ctx* ctx, double* Mx, int np) {
    const int b = 128;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->dn_work, (size_t)2 * np * b + (size_t)b * b))) return rc;
    double* Cbuf = ctx->dn_work;
    double* Rbuf = Cbuf + (size_t)np * b;
    double* Dk = Rbuf + (size_t)np * b;
    const size_t lds = (size_t)b * (b + 1) * sizeof(double);
    ASB_HIP(ctx, hipFuncSetAttribute((const void*)k_block_inverse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const size_t pitch = (size_t)np * sizeof(double);
    for (int k0 = 0; k0 < np; k0 += b) {
        const int bk = (np - k0) < b ? (np - k0) : b;
        const size_t wb = (size_t)bk * sizeof(double);
        double* Mkk = Mx + (size_t)k0 * np + k0;
        hipLaunchKernelGGL(k_block_inverse, dim3(1), dim3(1024), (size_t)bk * (bk + 1) * sizeof(double), ctx->stream, Mkk,
                           (long long)np, bk, Dk, ctx->la_status);
        // C = M[:, k] with the pivot rows zeroed; R = D M[k, :] with the pivot columns zeroed
        ASB_HIP(ctx, hipMemcpy2DAsync(Cbuf, wb, Mx + k0, pitch, wb, (size_t)np, hipMemcpyDeviceToDevice, ctx->stream));
        ASB_HIP(ctx, hipMemsetAsync(Cbuf + (size_t)k0 * bk, 0, (size_t)bk * wb, ctx->stream));
        if ((rc = asb_gemm_nn(ctx, Dk, bk, Mx + (size_t)k0 * np, np, Rbuf, np, bk, np, bk, 1.0, 0.0))) return rc;
        ASB_HIP(ctx, hipMemset2DAsync(Rbuf + k0, pitch, 0, wb, (size_t)bk, ctx->stream));
        // every other block: M_ij -= M_ik D M_kj
        if ((rc = asb_gemm_nn(ctx, Cbuf, bk, Rbuf, np, Mx, np, np, np, bk, -1.0, 1.0))) return rc;
        // pivot row / column / block
        ASB_HIP(ctx, hipMemcpyAsync(Mx + (size_t)k0 * np, Rbuf, (size_t)bk * pitch, hipMemcpyDeviceToDevice, ctx->stream));
        if ((rc = asb_gemm_nn(ctx, Cbuf, bk, Dk, bk, Mx + k0, np, np, bk, bk, -1.0, 0.0))) return rc;
        ASB_HIP(ctx, hipMemcpy2DAsync(Mkk, pitch, Dk, wb, wb, (size_t)bk, hipMemcpyDeviceToDevice, ctx->stream));
    }
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

// ---- pivot blocks of the symmetric form: SPD inverse of a b x b block (b <= 128, a multiple of 16) in ONE block's LDS, Gauss-Jordan
// by 16 x 16 tiles (round 3).  k_block_inverse above sweeps the whole block once per pivot (16 LDS read-modify-writes per thread and
// pivot: 570 us for b = 128); here a pivot TILE costs one 16-step sweep of a 16 x 16 tile plus one register-tiled rank-16 update
// (4 x 4 outputs per thread: an eighth of the LDS traffic) -- about 70 us.  src (ld lds) -> dst (b x b contiguous, ld b).
__global__ __launch_bounds__(1024) void k_block_inverse16(const double* __restrict__ src, long long lds, int b,
                                                          double* __restrict__ dst, int* __restrict__ status) {
    extern __shared__ double Sm[];       // b x 129
    __shared__ double D[16][17];
    __shared__ double R[16][129];
    __shared__ int bad;
    constexpr int ld = 129;
    const int tid = threadIdx.x;
    if (tid == 0) bad = 0;
    for (int i = tid >> 7; i < b; i += 8) {
        const int j = tid & 127;
        if (j < b) Sm[i * ld + j] = src[(long long)i * lds + j];
    }
    __syncthreads();
    const int nT = b / 16;
    for (int p = 0; p < nT; ++p) {
        const int p0 = p * 16;
        // the pivot tile's own inverse: ONE wave (lane l: column l & 15, rows (l >> 4) + 4 q), 16 Gauss-Jordan steps through LDS
        // with no block barrier inside -- LDS operations of a wave complete in order; the other 15 waves wait below
        if (tid < 64) {
            const int c = tid & 15, rb = tid >> 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) D[rb + 4 * q][c] = Sm[(p0 + rb + 4 * q) * ld + p0 + c];
            for (int j = 0; j < 16; ++j) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const double piv = D[j][j];
                const double d = 1.0 / piv;
                const double prow = D[j][c] * d;
                double v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = rb + 4 * q;
                    const double pcol = D[r][j];
                    v[q] = (r == j) ? (c == j ? d : prow) : (c == j ? -pcol * d : D[r][c] - pcol * prow);
                }
                if (tid == 0 && !(piv > 0.0)) bad = 1;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int q = 0; q < 4; ++q) D[rb + 4 * q][c] = v[q];
            }
        }
        __syncthreads();
        // row panel R = Dinv S[p, :] outside the pivot columns
        {
            const int j = tid & 127;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t = (tid >> 7) + 8 * h;
                if (j < b && (j < p0 || j >= p0 + 16)) {
                    double acc = 0.0;
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc += D[t][q] * Sm[(p0 + q) * ld + j];
                    R[t][j] = acc;
                }
            }
        }
        __syncthreads();
        // rank-16 update of everything outside the pivot tile's rows and columns: 4 x 4 outputs per thread
        {
            const int i0 = (tid >> 5) * 4, j0 = (tid & 31) * 4;
            if (i0 < b && j0 < b && (i0 < p0 || i0 >= p0 + 16) && (j0 < p0 || j0 >= p0 + 16)) {
                double acc[4][4];
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 4; ++y) acc[x][y] = Sm[(i0 + x) * ld + j0 + y];
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    double cc[4], rr[4];
#pragma unroll
                    for (int x = 0; x < 4; ++x) cc[x] = Sm[(i0 + x) * ld + p0 + t];
#pragma unroll
                    for (int y = 0; y < 4; ++y) rr[y] = R[t][j0 + y];
#pragma unroll
                    for (int x = 0; x < 4; ++x)
#pragma unroll
                        for (int y = 0; y < 4; ++y) acc[x][y] -= cc[x] * rr[y];
                }
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 4; ++y) Sm[(i0 + x) * ld + j0 + y] = acc[x][y];
            }
        }
        __syncthreads();
        // new pivot column -S[:, p] Dinv (into registers, written behind the barrier), new pivot row R, pivot tile Dinv
        double ncol[2] = {0.0, 0.0};
        {
            const int i = tid & 127;
            if (i < b && (i < p0 || i >= p0 + 16)) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int cc = (tid >> 7) + 8 * h;
                    double acc = 0.0;
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc += Sm[i * ld + p0 + q] * D[q][cc];
                    ncol[h] = -acc;
                }
            }
        }
        __syncthreads();
        {
            const int i = tid & 127;
            if (i < b && (i < p0 || i >= p0 + 16)) {
#pragma unroll
                for (int h = 0; h < 2; ++h) Sm[i * ld + p0 + (tid >> 7) + 8 * h] = ncol[h];
            }
            const int j = tid & 127;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t = (tid >> 7) + 8 * h;
                if (j < b) Sm[(p0 + t) * ld + j] = (j >= p0 && j < p0 + 16) ? D[t][j - p0] : R[t][j];
            }
        }
        __syncthreads();
    }
    if (bad) {
        if (tid == 0) status[0] = 1;
        return;
    }
    for (int i = tid >> 7; i < b; i += 8) {
        const int j = tid & 127;
        if (j < b) dst[(long long)i * b + j] = Sm[i * ld + j];
    }
}
__global__ __launch_bounds__(256) void k_copy_block(const double* __restrict__ src, long long lds, int rows, int cols,
                                                    double* __restrict__ dst, long long ldd) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < rows * cols; e += gridDim.x * 256) {
        const int i = e / cols, j = e % cols;
        dst[(long long)i * ldd + j] = src[(long long)i * lds + j];
    }
}
// inverse of [A11 A12; A12^T A22] from I11 = A11^-1, X = I11 A12, I22 = (A22 - A12^T X)^-1, Y = X I22:
// [I11 + Y X^T, -Y; -Y^T, I22] into D (ld bk); b1 = 128 rows in front, b2 behind
__global__ __launch_bounds__(256) void k_schur_assemble(const double* __restrict__ I11, const double* __restrict__ X,
                                                        const double* __restrict__ Y, const double* __restrict__ I22, int b1, int b2,
                                                        double* __restrict__ D) {
    const int bk = b1 + b2;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < bk * bk; e += gridDim.x * 256) {
        const int i = e / bk, j = e % bk;
        double v;
        if (i < b1 && j < b1) {
            v = I11[(long long)i * b1 + j];
            for (int t = 0; t < b2; ++t) v += Y[(long long)i * b2 + t] * X[(long long)j * b2 + t];
        } else if (i < b1) v = -Y[(long long)i * b2 + (j - b1)];
        else if (j < b1) v = -Y[(long long)j * b2 + (i - b1)];
        else v = I22[(long long)(i - b1) * b2 + (j - b1)];
        D[e] = v;
    }
}
// D (bk x bk contiguous, symmetric positive definite, bk <= 256 a multiple of 16) <- D^-1; W: 5 x 128 x 128 doubles of work
static int pivot_block_inverse(asb_ctx* ctx, double* D, int bk, double* W) {
    const size_t lds = (size_t)128 * 129 * sizeof(double);
    ASB_HIP(ctx, hipFuncSetAttribute((const void*)k_block_inverse16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    double *I11 = W, *X = W + 16384, *Sc = X + 16384, *I22 = Sc + 16384, *Y = I22 + 16384;
    if (bk <= 128) {
        hipLaunchKernelGGL(k_block_inverse16, dim3(1), dim3(1024), lds, ctx->stream, D, (long long)bk, bk, I11, ctx->la_status);
        hipLaunchKernelGGL(k_copy_block, dim3(64), dim3(256), 0, ctx->stream, I11, (long long)bk, bk, bk, D, (long long)bk);
        return ASB_OK;
    }
    const int b1 = 128, b2 = bk - 128;
    int rc;
    hipLaunchKernelGGL(k_block_inverse16, dim3(1), dim3(1024), lds, ctx->stream, D, (long long)bk, b1, I11, ctx->la_status);
    if ((rc = asb_gemm_nn(ctx, I11, b1, D + b1, bk, X, b2, b1, b2, b1, 1.0, 0.0))) return rc;                      // X = I11 A12
    hipLaunchKernelGGL(k_copy_block, dim3(64), dim3(256), 0, ctx->stream, D + (size_t)b1 * bk + b1, (long long)bk, b2, b2, Sc, (long long)b2);
    if ((rc = asb_gemm_nn(ctx, D + (size_t)b1 * bk, bk, X, b2, Sc, b2, b2, b2, b1, -1.0, 1.0))) return rc;        // A22 - A21 X
    hipLaunchKernelGGL(k_block_inverse16, dim3(1), dim3(1024), lds, ctx->stream, Sc, (long long)b2, b2, I22, ctx->la_status);
    if ((rc = asb_gemm_nn(ctx, X, b2, I22, b2, Y, b2, b1, b2, b2, 1.0, 0.0))) return rc;                          // Y = X I22
    hipLaunchKernelGGL(k_schur_assemble, dim3(256), dim3(256), 0, ctx->stream, I11, X, Y, I22, b1, b2, D);
    return ASB_OK;
}

// ---- symmetric form (round 3).  A Gauss-Jordan sweep keeps the matrix symmetric up to a sign: with the pivot blocks swept in
// order, M_ji = -M_ij^T when exactly one of the block indices i, j has been swept and +M_ij^T otherwise.  So only the tiles on and
// above the diagonal are stored and updated (np^3 flop and half the traffic instead of 2 np^3), with pivot blocks of 256
// (half as many sweeps over the matrix as with 128; the 256 x 256 pivot block itself is inverted by two sweeps of the plain
// form).  Sweep of pivot block k (rows k0 .. k0 + bk):
//   B (bk x np)  the pivot ROW as it would read in a full matrix: B_t = M_kt for t behind the pivot (stored), -M_tk^T for t in
//                front of it (t swept, k not yet), 0 in the pivot columns;
//   R = D B      with D = M_kk^-1;   Ct[c][r] = s_c B[r][c], s = -1 in front of the pivot, +1 behind it (M_ik = s_i B_i^T);
//   M_ij -= Ct_i R_j  for the tiles i <= j (asb_gemm_nn, triangular form);
//   new pivot row M_kj = R_j (j behind), new pivot column M_ik = R_i^T (i in front), M_kk = D.
// After the last sweep every index is swept: the lower triangle is the mirror image of the upper one.
#define GJ_BK 256
__global__ __launch_bounds__(256) void k_gj_pivot_copy(const double* __restrict__ M, int np, int k0, int bk, double* __restrict__ D) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < bk * bk; e += gridDim.x * 256) {
        const int r = e / bk, c = e % bk;
        D[e] = c >= r ? M[(long long)(k0 + r) * np + k0 + c] : M[(long long)(k0 + c) * np + k0 + r];
    }
}
// 32 x 32 tiles through LDS: tile (tr, tc) of the bk x np panel.  Behind the pivot the source is the pivot row (read along c),
// in front of it the pivot column (read along r); B is written along c, Ct along r.
__global__ __launch_bounds__(256) void k_gj_panel(const double* __restrict__ M, int np, int k0, int bk, double* __restrict__ B,
                                                  double* __restrict__ Ct) {
    __shared__ double T[32][33];
    const int tc = blockIdx.x, tr = blockIdx.y, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = tc * 32, r0 = tr * 32;
    const bool front = c0 + 32 <= k0, behind = c0 >= k0 + bk;          // (k0 and bk are multiples of 16: a tile may straddle)
    for (int q = ty; q < 32; q += 8) {
        double v = 0.0;
        if (behind || (!front && c0 + tx >= k0 + bk)) {                // element (r0 + q, c0 + tx), read along c
            const int r = r0 + q, c = c0 + tx;
            if (r < bk && c < np && c >= k0 + bk) v = M[(long long)(k0 + r) * np + c];
            T[q][tx] = v;
        }
    }
    __syncthreads();
    for (int q = ty; q < 32; q += 8) {
        // element (r0 + tx, c0 + q), read along r from the pivot column
        const int r = r0 + tx, c = c0 + q;
        if (r < bk && c < k0) T[tx][q] = -M[(long long)c * np + k0 + r];
    }
    __syncthreads();
    for (int q = ty; q < 32; q += 8) {
        const int r = r0 + q, c = c0 + tx;
        if (r < bk && c < np) {
            const bool piv = c >= k0 && c < k0 + bk;
            B[(long long)r * np + c] = piv ? 0.0 : T[q][tx];
        }
    }
    for (int q = ty; q < 32; q += 8) {
        const int r = r0 + tx, c = c0 + q;
        if (r < bk && c < np) {
            const bool piv = c >= k0 && c < k0 + bk;
            Ct[(long long)c * bk + r] = piv ? 0.0 : (c < k0 ? -T[tx][q] : T[tx][q]);
        }
    }
}
// new pivot row (behind the pivot), pivot column (in front of it) and pivot block from R (bk x np) and D (bk x bk)
__global__ __launch_bounds__(256) void k_gj_write(double* __restrict__ M, int np, int k0, int bk, const double* __restrict__ R,
                                                  const double* __restrict__ D) {
    __shared__ double T[32][33];
    const int tc = blockIdx.x, tr = blockIdx.y, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = tc * 32, r0 = tr * 32;
    for (int q = ty; q < 32; q += 8) {
        const int r = r0 + q, c = c0 + tx;
        double v = 0.0;
        if (r < bk && c < np) v = R[(long long)r * np + c];
        T[q][tx] = v;
        if (r < bk && c < np) {
            if (c >= k0 + bk) M[(long long)(k0 + r) * np + c] = v;
            else if (c >= k0) M[(long long)(k0 + r) * np + c] = D[(long long)r * bk + (c - k0)];
        }
    }
    __syncthreads();
    for (int q = ty; q < 32; q += 8) {
        const int r = r0 + tx, c = c0 + q;
        if (r < bk && c < k0) M[(long long)c * np + k0 + r] = T[tx][q];
    }
}
__global__ __launch_bounds__(256) void k_mirror_upper(double* __restrict__ M, int np) {
    __shared__ double T[32][33];
    const int tj = blockIdx.x, ti = blockIdx.y, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    if (tj < ti) return;
    for (int q = ty; q < 32; q += 8) {
        const int i = ti * 32 + q, j = tj * 32 + tx;
        T[q][tx] = (i < np && j < np) ? M[(long long)i * np + j] : 0.0;
    }
    __syncthreads();
    for (int q = ty; q < 32; q += 8) {
        const int j = tj * 32 + q, i = ti * 32 + tx;          // element (j, i) of the lower triangle = (i, j) of the upper
        if (i < np && j < np && j > i) M[(long long)j * np + i] = T[tx][q];
    }
}

// M (np x np, ld np, np a multiple of 16) <- M^-1 for symmetric positive definite M.
int asb_dense_spd_inverse(asb_ctx* ctx, double* Mx, int np) {
    if (!ctx || !Mx || np < 16 || (np & 15)) return ASB_ERR_ARG;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->la_status, (size_t)4))) return rc;
    ASB_HIP(ctx, hipMemsetAsync(ctx->la_status, 0, 4 * sizeof(int), ctx->stream));
    static const int sym = getenv("ASB_DENSE_SYM") ? atoi(getenv("ASB_DENSE_SYM")) : 1;
    const bool timing = getenv("ASB_DEBUG_GJ") != nullptr;
    if (timing) (void)hipStreamSynchronize(ctx->stream);
    const auto t_start = std::chrono::steady_clock::now();
    if (np <= GJ_BK && sym) {
        // one pivot block: the Schur step over the tiled in-LDS inverse alone (SPLOCS' K x K systems: 40 - 130 us)
        if ((rc = asb_alloc(ctx, &ctx->dn_sym, (size_t)5 * 128 * 128))) return rc;
        if ((rc = pivot_block_inverse(ctx, Mx, np, ctx->dn_sym))) return rc;
    } else if (!sym) {
        if ((rc = spd_inverse_full(ctx, Mx, np))) return rc;
    } else {
        if ((rc = asb_alloc(ctx, &ctx->dn_sym, (size_t)3 * GJ_BK * np + (size_t)GJ_BK * GJ_BK + (size_t)5 * 128 * 128))) return rc;
        double* B = ctx->dn_sym;
        double* R = B + (size_t)GJ_BK * np;
        double* Ct = R + (size_t)GJ_BK * np;
        double* D = Ct + (size_t)GJ_BK * np;
        double* W = D + (size_t)GJ_BK * GJ_BK;
        for (int k0 = 0; k0 < np; k0 += GJ_BK) {
            const int bk = (np - k0) < GJ_BK ? (np - k0) : GJ_BK;
            hipLaunchKernelGGL(k_gj_pivot_copy, dim3(64), dim3(256), 0, ctx->stream, Mx, np, k0, bk, D);
            if ((rc = pivot_block_inverse(ctx, D, bk, W))) return rc;
            const dim3 pg((np + 31) / 32, (bk + 31) / 32);
            hipLaunchKernelGGL(k_gj_panel, pg, dim3(256), 0, ctx->stream, Mx, np, k0, bk, B, Ct);
            if ((rc = asb_gemm_nn(ctx, D, bk, B, np, R, np, bk, np, bk, 1.0, 0.0))) return rc;
            if ((rc = asb_gemm_nn(ctx, Ct, bk, R, np, Mx, np, np, np, bk, -1.0, 1.0, 1))) return rc;
            hipLaunchKernelGGL(k_gj_write, pg, dim3(256), 0, ctx->stream, Mx, np, k0, bk, R, D);
        }
        hipLaunchKernelGGL(k_mirror_upper, dim3((np + 31) / 32, (np + 31) / 32), dim3(256), 0, ctx->stream, Mx, np);
    }
    ASB_CHECK_LAUNCH(ctx);
    int st[4];
    ASB_HIP(ctx, hipMemcpyAsync(st, ctx->la_status, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (timing)
        fprintf(stderr, "[asb] dense SPD inverse of %d x %d: %.1f ms\n", np, np,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
    if (st[0]) ASB_FAIL(ctx, ASB_ERR_NUMERIC, "dense inverse: the matrix is not positive definite");
    return ASB_OK;
}

// test hook: inverse of a host SPD matrix (n x n) through the device path above
extern "C" int asb_test_spd_inverse(asb_ctx* ctx, const double* A_host, int64_t n, double* Ainv_host) {
    if (!ctx || !A_host || !Ainv_host || n < 1) return ASB_ERR_ARG;
    const int np = (int)((n + 15) / 16 * 16);
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->dn_test, (size_t)np * np))) return rc;
    std::vector<double> pad((size_t)np * np, 0.0);
    for (int i = 0; i < np; ++i)
        for (int j = 0; j < np; ++j) pad[(size_t)i * np + j] = (i < n && j < n) ? A_host[(size_t)i * n + j] : (i == j ? 1.0 : 0.0);
    ASB_HIP(ctx, hipMemcpyAsync(ctx->dn_test, pad.data(), pad.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = asb_dense_spd_inverse(ctx, ctx->dn_test, np))) return rc;
    ASB_HIP(ctx, hipMemcpyAsync(pad.data(), ctx->dn_test, pad.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j) Ainv_host[i * n + j] = pad[(size_t)i * np + j];
    return ASB_OK;
}
