// asb_dense.hip -- dense f64 building blocks for meshes small enough that an N x N matrix is cheap in 288 GB of HBM:
//   k_gemm_nn            C = beta C + alpha A B, row-major, LDS-tiled v_mfma_f64_16x16x4_f64 (128 x 128 tile per block)
//   asb_dense_spd_inverse in-place inverse of a symmetric positive definite matrix by blocked Gauss-Jordan sweeps
//                        (block 128: one LDS-resident block inverse + three GEMMs per sweep, 2 n^3 flop in all)
// Used by the device heat-method geodesics (asb_geodesic.hip): the two SPD systems of utils/support.py:170-171 are
// inverted once (N = 15 000: 1.75 GB each) and every later solve is a column gather / one GEMM.  gfx950 only.
#include "asb_common.h"

typedef double d4 __attribute__((ext_vector_type(4)));

#define DG_BM 128
#define DG_KC 16
#define DG_SA 18      // LDS row stride of the A stage  As[i][k]
#define DG_SB 144     // LDS row stride of the B stage  Bs[k][j]

// grid (tiles_n, tiles_m, S).  All of M, N, Kc, lda, ldb, ldc are multiples of 2 and the pointers 16-byte aligned
// (the callers pad to 16).  S > 1: the slab's product goes to part[z] (M x N, ld N) and k_gemm_finish combines.
__global__ __launch_bounds__(256, 2) void k_gemm_nn(const double* __restrict__ A, long long lda, const double* __restrict__ B,
                                                   long long ldb, double* __restrict__ C, long long ldc, int M, int N, int Kc,
                                                   double alpha, double beta, int slab, double* __restrict__ part) {
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
        for (int q = 0; q < 4; ++q) *reinterpret_cast<double2*>(&As[buf][ar][ak + 2 * q]) = ra[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<double2*>(&Bs[buf][wave + 4 * q][2 * lane]) = rb[q];
    };
    d4 acc[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = d4{0.0, 0.0, 0.0, 0.0};
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
                    if (split) *dst = acc[x][y][q];
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
                int N, int Kc, double alpha, double beta) {
    if ((lda | ldb | ldc | M | N | Kc) & 1) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_gemm_nn: odd dimension");
    const int tm = (M + DG_BM - 1) / DG_BM, tn = (N + DG_BM - 1) / DG_BM;
    // split the contraction when the tile grid alone cannot fill the chip (skinny products with a long contraction)
    int S = 1;
    if ((long long)tm * tn < 512 && Kc >= 1024) {
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
    hipLaunchKernelGGL(k_gemm_nn, dim3(tn, tm, S), dim3(256), 0, ctx->stream, A, lda, B, ldb, C, ldc, M, N, Kc, alpha, beta, slab,
                       ctx->la_part);
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

// M (np x np, ld np, np a multiple of 16) <- M^-1 for symmetric positive definite M.
int asb_dense_spd_inverse(asb_ctx* ctx, double* Mx, int np) {
    if (!ctx || !Mx || np < 16 || (np & 15)) return ASB_ERR_ARG;
    const int b = 128;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->dn_work, (size_t)2 * np * b + (size_t)b * b))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->la_status, (size_t)4))) return rc;
    double* Cbuf = ctx->dn_work;
    double* Rbuf = Cbuf + (size_t)np * b;
    double* Dk = Rbuf + (size_t)np * b;
    ASB_HIP(ctx, hipMemsetAsync(ctx->la_status, 0, 4 * sizeof(int), ctx->stream));
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
    int st[4];
    ASB_HIP(ctx, hipMemcpyAsync(st, ctx->la_status, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
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
