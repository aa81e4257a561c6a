// Small dense linear algebra on the device (f64): MFMA "TN" GEMM, transpose, a one-block
// Jacobi eigen-solver, and on top of them the per-dimension orthogonalisation of the basis --
// posComponents.post_process_components, snapbases/posComponents.py:284-287
// (`comps[:,:,l] = orth(comps[:,:,l].T).T`, scipy's SVD-based orth).  gfx950 only.
#include "asb_kernels.h"

typedef double d4 __attribute__((ext_vector_type(4)));

// --------------------------------------------------------------------------------------
// k_gemm_tn:  part[s][i][j] = sum_{r in slab s} A[r*lda + i*sa] * B[r*ldb + j]   (v_mfma_f64_16x16x4_f64)
// One wave per (16x16 output tile, slab).  Lane (i = l&15, g = l>>4) feeds A[r+g][i0+i] and
// B[r+g][j0+i].  With one slab the result goes straight to out[i*so_i + j*so_j].
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_gemm_tn(const double* __restrict__ A, long long lda, long long sa,
                                                const double* __restrict__ B, long long ldb, long long Rn, int I,
                                                int J, long long slab, double* __restrict__ out, long long so_i,
                                                long long so_j, long long slab_stride) {
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    const int tj = (J + 15) / 16;
    const int ti = blockIdx.x / tj, tjx = blockIdx.x % tj;
    const int i0 = ti * 16, j0 = tjx * 16;
    const long long r0 = (long long)blockIdx.y * slab;
    long long r1 = r0 + slab;
    if (r1 > Rn) r1 = Rn;
    const bool ai = (i0 + i) < I, bj = (j0 + i) < J;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    const double* pa = A + (long long)(i0 + i) * sa;
    const double* pb = B + (j0 + i);
    for (long long rb = r0; rb < r1; rb += 4) {
        const long long r = rb + g;
        const bool in = r < r1;
        const double a = (in && ai) ? pa[r * lda] : 0.0;
        const double b = (in && bj) ? pb[r * ldb] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    double* o = out + (long long)blockIdx.y * slab_stride;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int oi = i0 + g + 4 * q, oj = j0 + i;
        if (oi < I && oj < J) o[(long long)oi * so_i + (long long)oj * so_j] = acc[q];
    }
}

__global__ __launch_bounds__(256) void k_sum_slabs(const double* __restrict__ part, int S, int I, int J,
                                                   double* __restrict__ out, long long so_i, long long so_j) {
    const long long n = (long long)I * J;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int q = 0; q < S; ++q) s += part[(long long)q * n + e];
        out[(e / J) * so_i + (e % J) * so_j] = s;
    }
}

// --------------------------------------------------------------------------------------
// k_syrk_tn:  G = X^T X for a tall X (R x n, row stride ld) -- the F x F Gram matrix of the POD (config 5,
// 2 R n^2 flop: the one place on the path where the contraction is really dense, so this is the MFMA kernel).
// Block = 4 waves, 128 x 128 output tile (only tiles on or above the diagonal), each wave a 64 x 64 sub-tile =
// 4 x 4 v_mfma_f64_16x16x4_f64 accumulators.  The contraction runs over rows in stages of 16: both 16 x 128 operand
// slabs go global -> registers -> LDS (double-buffered, one barrier per stage), row stride 144 doubles so that the four
// k-rows a wave reads per instruction fall into different bank halves.  blockIdx.y = row slab (split-K into
// `part`, summed in fixed order by k_syrk_finish, which also mirrors the lower triangle).
// --------------------------------------------------------------------------------------
#define SY_BM 128
#define SY_KC 16
#define SY_LD 144
// SYM = false (round 3): the same tiling for a general Out (n x nj) = X^T Y, Y = R rows of stride ldy -- every tile, no mirror
// (B = Q^T A of the POD's Rayleigh-Ritz step: 345 GFLOP that the one-wave-per-tile k_gemm_tn did at 15 TFLOP/s).
template <bool SYM>
__global__ __launch_bounds__(256, 2) void k_syrk_tn(const double* __restrict__ X, long long ld, long long R, int n, long long slab,
                                                   double* __restrict__ part, int nb, const double* __restrict__ Y = nullptr,
                                                   long long ldy = 0, int nj = 0, int nbj = 0) {
    __shared__ double As[2][SY_KC][SY_LD];
    __shared__ double Bs[2][SY_KC][SY_LD];
    int t = blockIdx.x, ib = 0, jb;
    if (SYM) {
        while (t >= nb - ib) { t -= nb - ib; ++ib; }
        jb = ib + t;
        Y = X;
        ldy = ld;
        nj = n;
    } else {
        ib = t / nbj;
        jb = t % nbj;
    }
    const int i0 = ib * SY_BM, j0 = jb * SY_BM;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int wi = wave >> 1, wj = wave & 1;
    const long long r_begin = (long long)blockIdx.y * slab;
    long long r_end = r_begin + slab;
    if (r_end > R) r_end = R;
    const int ca = i0 + 2 * lane, cb = j0 + 2 * lane;
    const bool oka = ca < ld, okb = cb < ldy;
    double2 ra[4], rb[4];
    auto fetch = [&](long long r0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long r = r0 + wave + 4 * q;
            const bool in = r < r_end;
            ra[q] = (in && oka) ? *reinterpret_cast<const double2*>(X + r * ld + ca) : make_double2(0.0, 0.0);
            rb[q] = (in && okb) ? *reinterpret_cast<const double2*>(Y + r * ldy + cb) : make_double2(0.0, 0.0);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<double2*>(&As[buf][wave + 4 * q][2 * lane]) = ra[q];
            *reinterpret_cast<double2*>(&Bs[buf][wave + 4 * q][2 * lane]) = rb[q];
        }
    };
    d4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
    fetch(r_begin);
    stash(0);
    __syncthreads();
    int cur = 0;
    for (long long r0 = r_begin; r0 < r_end; r0 += SY_KC) {
        const bool more = r0 + SY_KC < r_end;
        if (more) fetch(r0 + SY_KC);
#pragma unroll
        for (int ks = 0; ks < SY_KC / 4; ++ks) {
            double a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[q] = As[cur][ks * 4 + g][wi * 64 + q * 16 + li];
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
    double* o = part + (long long)blockIdx.y * n * nj;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int oi = i0 + wi * 64 + x * 16 + g + 4 * q, oj = j0 + wj * 64 + y * 16 + li;
                if (oi < n && oj < nj) o[(long long)oi * nj + oj] = acc[x][y][q];
            }
}

// out[i][j] = out[j][i] = sum_s part[s][i][j] over the tiles on/above the diagonal (fixed summation order)
__global__ __launch_bounds__(256) void k_syrk_finish(const double* __restrict__ part, int S, int n, double* __restrict__ out) {
    const long long total = (long long)n * n;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int i = (int)(e / n), j = (int)(e % n);
        if (j / SY_BM < i / SY_BM) continue;
        double s = 0.0;
        for (int q = 0; q < S; ++q) s += part[(long long)q * total + e];
        out[e] = s;
        if (j / SY_BM > i / SY_BM) out[(long long)j * n + i] = s;
    }
}

// G (n x n) = X^T X, X = R rows of stride ld (columns >= n up to ld must be readable: the zero padding of the rows)
int asb_syrk_tn(asb_ctx* ctx, const double* X, long long ld, long long R, int n, double* out) {
    const int nb = (n + SY_BM - 1) / SY_BM;
    const int tiles = nb * (nb + 1) / 2;
    // enough blocks for >= 8 rounds of the chip's 512 resident blocks, slabs of at least 512 rows
    int S = (8 * 512 + tiles - 1) / tiles;
    const long long maxS = (R + 511) / 512;
    if (S > maxS) S = (int)(maxS < 1 ? 1 : maxS);
    if (S > 64) S = 64;
    long long slab = ((R + S - 1) / S + SY_KC - 1) / SY_KC * SY_KC;
    if (slab < SY_KC) slab = SY_KC;
    S = (int)((R + slab - 1) / slab);
    if (S < 1) S = 1;
    const size_t need = (size_t)S * n * n;
    if (need > ctx->la_part_cap) {
        int rc = asb_alloc(ctx, &ctx->la_part, need);
        if (rc) return rc;
        ctx->la_part_cap = need;
    }
    hipLaunchKernelGGL(k_syrk_tn<true>, dim3(tiles, S), dim3(256), 0, ctx->stream, X, ld, R, n, slab, ctx->la_part, nb);
    hipLaunchKernelGGL(k_syrk_finish, dim3(2048), dim3(256), 0, ctx->stream, ctx->la_part, S, n, out);
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

__global__ __launch_bounds__(256) void k_sum_parts(const double* __restrict__ part, int S, long long total, double* __restrict__ out) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        double s = 0.0;
        for (int q = 0; q < S; ++q) s += part[(long long)q * total + e];
        out[e] = s;
    }
}
// Out (I x J, row-major) = X^T Y for tall X (R x I, stride ldx) and Y (R x J, stride ldy); even strides, 16-byte aligned rows
int asb_gemm_tn_big(asb_ctx* ctx, const double* X, long long ldx, const double* Y, long long ldy, long long R, int I, int J, double* out) {
    const int nbi = (I + SY_BM - 1) / SY_BM, nbj = (J + SY_BM - 1) / SY_BM, tiles = nbi * nbj;
    int S = (8 * 512 + tiles - 1) / tiles;
    const long long maxS = (R + 511) / 512;
    if (S > maxS) S = (int)(maxS < 1 ? 1 : maxS);
    if (S > 64) S = 64;
    long long slab = ((R + S - 1) / S + SY_KC - 1) / SY_KC * SY_KC;
    if (slab < SY_KC) slab = SY_KC;
    S = (int)((R + slab - 1) / slab);
    if (S < 1) S = 1;
    const size_t need = (size_t)S * I * J;
    if (need > ctx->la_part_cap) {
        int rc = asb_alloc(ctx, &ctx->la_part, need);
        if (rc) return rc;
        ctx->la_part_cap = need;
    }
    hipLaunchKernelGGL(k_syrk_tn<false>, dim3(tiles, S), dim3(256), 0, ctx->stream, X, ldx, R, I, slab, ctx->la_part, nbi, Y, ldy, J, nbj);
    hipLaunchKernelGGL(k_sum_parts, dim3(2048), dim3(256), 0, ctx->stream, ctx->la_part, S, (long long)I * J, out);
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

__global__ __launch_bounds__(256) void k_transpose_small(const double* __restrict__ in, long long rows, long long cols,
                                                         double* __restrict__ out) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long long c0 = (long long)blockIdx.x * 32, r0 = (long long)blockIdx.y * 32;
    for (int q = 0; q < 4; ++q) {
        const long long r = r0 + ty + q * 8, c = c0 + tx;
        tile[ty + q * 8][tx] = (r < rows && c < cols) ? in[r * cols + c] : 0.0;
    }
    __syncthreads();
    for (int q = 0; q < 4; ++q) {
        const long long c = c0 + ty + q * 8, r = r0 + tx;
        if (r < rows && c < cols) out[c * rows + r] = tile[tx][ty + q * 8];
    }
}

int asb_gemm_tn_s(asb_ctx* ctx, const double* A, long long lda, long long sa, const double* B, long long ldb, long long Rn,
                  int I, int J, double* out, long long so_i, long long so_j) {
    const int tiles = ((I + 15) / 16) * ((J + 15) / 16);
    int S = (int)(4096 / (tiles > 0 ? tiles : 1));
    if (S < 1) S = 1;
    long long maxS = (Rn + 63) / 64;
    if (S > maxS) S = (int)maxS;
    if (S > 64) S = 64;
    long long slab = ((Rn + S - 1) / S + 3) / 4 * 4;
    S = (int)((Rn + slab - 1) / slab);
    if (S > 1) {
        const size_t need = (size_t)S * I * J;
        if (need > ctx->la_part_cap) {
            int rc = asb_alloc(ctx, &ctx->la_part, need);
            if (rc) return rc;
            ctx->la_part_cap = need;
        }
        hipLaunchKernelGGL(k_gemm_tn, dim3(tiles, S), dim3(64), 0, ctx->stream, A, lda, sa, B, ldb, Rn, I, J, slab,
                           ctx->la_part, (long long)J, (long long)1, (long long)I * J);
        const long long n = (long long)I * J;
        hipLaunchKernelGGL(k_sum_slabs, dim3((unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024)), dim3(256), 0,
                           ctx->stream, ctx->la_part, S, I, J, out, so_i, so_j);
    } else {
        hipLaunchKernelGGL(k_gemm_tn, dim3(tiles, 1), dim3(64), 0, ctx->stream, A, lda, sa, B, ldb, Rn, I, J, slab, out, so_i,
                           so_j, (long long)0);
    }
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

int asb_transpose(asb_ctx* ctx, const double* in, long long rows, long long cols, double* out) {
    dim3 tg((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    hipLaunchKernelGGL(k_transpose_small, tg, dim3(256), 0, ctx->stream, in, rows, cols, out);
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

// --------------------------------------------------------------------------------------
// k_jacobi_eig: cyclic two-sided Jacobi with the round-robin (tournament) ordering: n/2
// disjoint rotations per round run in parallel, n-1 rounds per sweep.  A lives in LDS, V in
// global memory.  One block.  Output: eigenvalues sorted descending, V columns accordingly.
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_jacobi_eig(const double* __restrict__ Ain, int n, double* __restrict__ lam,
                                                     double* __restrict__ V, double* __restrict__ Vtmp,
                                                     int* __restrict__ status) {
    extern __shared__ double sm[];
    const int ne = n + (n & 1);                 // even size (a decoupled dummy index when n is odd)
    double* A = sm;                             // ne x ne
    double* cs = A + ne * ne;                   // ne/2 cosines
    double* sn = cs + ne / 2;                   // ne/2 sines
    int* perm = reinterpret_cast<int*>(sn + ne / 2);      // ne
    __shared__ double red[32];
    __shared__ int conv;
    const int tid = threadIdx.x, nt = blockDim.x, half = ne / 2;
    for (int e = tid; e < ne * ne; e += nt) {
        const int r = e / ne, c = e % ne;
        A[e] = (r < n && c < n) ? Ain[r * n + c] : 0.0;
    }
    for (int e = tid; e < n * n; e += nt) Vtmp[e] = (e / n == e % n) ? 1.0 : 0.0;
    for (int e = tid; e < ne; e += nt) perm[e] = e;
    if (tid == 0) conv = 0;
    __syncthreads();
    for (int sweep = 0; sweep < 30; ++sweep) {
        // convergence: off-diagonal energy against the diagonal's
        double off = 0.0, dia = 0.0;
        for (int e = tid; e < ne * ne; e += nt) {
            const double v = A[e];
            if (e / ne == e % ne) dia += v * v; else off += v * v;
        }
        off = wave_sum(off); dia = wave_sum(dia);
        if ((tid & 63) == 0) { red[tid >> 6] = off; red[16 + (tid >> 6)] = dia; }
        __syncthreads();
        if (tid == 0) {
            double o = 0.0, d = 0.0;
            for (int q = 0; q < (nt >> 6); ++q) { o += red[q]; d += red[16 + q]; }
            conv = (o <= 1.0e-60 || o <= 1.0e-34 * d) ? 1 : 0;
        }
        __syncthreads();
        if (conv) break;
        for (int round = 0; round < ne - 1; ++round) {
            // rotation angles of this round's pairs (perm[m], perm[ne-1-m])
            if (tid < half) {
                int p = perm[tid], q = perm[ne - 1 - tid];
                if (p > q) { const int t = p; p = q; q = t; }
                const double apq = A[p * ne + q];
                double c = 1.0, s = 0.0;
                if (apq != 0.0) {
                    const double theta = (A[q * ne + q] - A[p * ne + p]) / (2.0 * apq);
                    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    c = 1.0 / sqrt(t * t + 1.0);
                    s = t * c;
                }
                cs[tid] = c; sn[tid] = s;
            }
            __syncthreads();
            // columns: A <- A J
            for (int e = tid; e < half * ne; e += nt) {
                const int m = e / ne, i = e % ne;
                int p = perm[m], q = perm[ne - 1 - m];
                if (p > q) { const int t = p; p = q; q = t; }
                const double c = cs[m], s = sn[m];
                const double aip = A[i * ne + p], aiq = A[i * ne + q];
                A[i * ne + p] = c * aip - s * aiq;
                A[i * ne + q] = s * aip + c * aiq;
            }
            __syncthreads();
            // rows: A <- J^T A ; eigenvectors: V <- V J
            for (int e = tid; e < half * ne; e += nt) {
                const int m = e / ne, j = e % ne;
                int p = perm[m], q = perm[ne - 1 - m];
                if (p > q) { const int t = p; p = q; q = t; }
                const double c = cs[m], s = sn[m];
                const double apj = A[p * ne + j], aqj = A[q * ne + j];
                A[p * ne + j] = c * apj - s * aqj;
                A[q * ne + j] = s * apj + c * aqj;
                if (j < n && p < n && q < n) {
                    const double vp = Vtmp[j * n + p], vq = Vtmp[j * n + q];
                    Vtmp[j * n + p] = c * vp - s * vq;
                    Vtmp[j * n + q] = s * vp + c * vq;
                }
            }
            __syncthreads();
            if (tid < half) {                   // exact zeros where the rotation annihilated
                int p = perm[tid], q = perm[ne - 1 - tid];
                A[p * ne + q] = 0.0; A[q * ne + p] = 0.0;
            }
            if (tid == 0) {                     // rotate the tournament: perm[1..ne-1] cyclically
                const int last = perm[ne - 1];
                for (int e = ne - 1; e > 1; --e) perm[e] = perm[e - 1];
                perm[1] = last;
            }
            __syncthreads();
        }
    }
    if (tid == 0 && !conv) status[0] = 2;
    // sort descending (selection sort by thread 0 on <= 128 values), then permute the columns
    if (tid == 0) {
        for (int i = 0; i < n; ++i) perm[i] = i;
        for (int i = 0; i < n; ++i) {
            int b = i;
            for (int j = i + 1; j < n; ++j)
                if (A[perm[j] * ne + perm[j]] > A[perm[b] * ne + perm[b]]) b = j;
            const int t = perm[i]; perm[i] = perm[b]; perm[b] = t;
        }
    }
    __syncthreads();
    for (int j = tid; j < n; j += nt) lam[j] = A[perm[j] * ne + perm[j]];
    for (int e = tid; e < n * n; e += nt) {
        const int r = e / n, j = e % n;
        V[e] = Vtmp[r * n + perm[j]];
    }
}

int asb_sym_eig(asb_ctx* ctx, const double* A_dev, int n, double* lam_dev, double* V_dev) {
    if (n < 1) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_sym_eig: n = %d", n);
    if (n > 128) {      // beyond one block's LDS: one-sided Jacobi on the rows, one launch per round (positive semi-definite input)
        int rc0;
        if ((rc0 = asb_alloc(ctx, &ctx->la_status, (size_t)4))) return rc0;
        ASB_HIP(ctx, hipMemsetAsync(ctx->la_status, 0, 4 * sizeof(int), ctx->stream));
        return asb_sym_eig_large(ctx, A_dev, n, lam_dev, V_dev);
    }
    const int ne = n + (n & 1);
    const size_t lds = ((size_t)ne * ne + ne) * sizeof(double) + (size_t)ne * sizeof(int) + 64;
    static size_t attr = 0;
    if (lds > 48 * 1024 && lds > attr) {
        ASB_HIP(ctx, hipFuncSetAttribute((const void*)k_jacobi_eig, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->la_vtmp, (size_t)n * n))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->la_status, (size_t)4))) return rc;
    ASB_HIP(ctx, hipMemsetAsync(ctx->la_status, 0, 4 * sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_jacobi_eig, dim3(1), dim3(n <= 32 ? 256 : 1024), lds, ctx->stream, A_dev, n, lam_dev, V_dev,
                       ctx->la_vtmp, ctx->la_status);
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

// --------------------------------------------------------------------------------------
// per-dimension orthogonalisation of the basis
// --------------------------------------------------------------------------------------
// Vs[k][j] = V[k][j] / sqrt(lam[j]);  sing[j] = sqrt(lam[j]);  status[1] = 1 if rank-deficient
__global__ __launch_bounds__(256) void k_scale_eigvecs(const double* __restrict__ V, const double* __restrict__ lam, int n,
                                                       double tol_rel, double* __restrict__ Vs, double* __restrict__ sing,
                                                       int* __restrict__ status) {
    const double smax = sqrt(fmax(lam[0], 0.0));
    for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
        const int j = e % n;
        const double sj = sqrt(fmax(lam[j], 0.0));
        Vs[e] = (sj > 0.0) ? V[e] / sj : 0.0;
        if (e < n) {
            const double se = sqrt(fmax(lam[e], 0.0));
            sing[e] = se;
            if (!(se > tol_rel * smax)) status[1] = 1;
        }
    }
}

// partial Gram matrices of the three coordinate slices: G[l] = A_l^T A_l (K x K), A_l[v][k] = comps[k][v][l]
extern "C" int asb_orth_gram(asb_ctx* ctx, double* G_dev) {
    if (!ctx || !ctx->comps) return ASB_ERR_ARG;
    const int64_t K = ctx->K, n = ctx->n_loc;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->oct, (size_t)3 * n * K + 128))) return rc;          // (+128: the Gram kernel's last tile reads on)
    if ((rc = asb_alloc(ctx, &ctx->og, (size_t)3 * K * K))) return rc;
    if ((rc = asb_transpose(ctx, ctx->comps, K, 3 * n, ctx->oct))) return rc;          // (3n x K): row 3v+l
    double* G = G_dev ? G_dev : ctx->og;
    // K >= 64 (the constraint bases: K = 288 at config 5): the 128 x 128-tile Gram kernel of the POD on the rows l, l + 3, ... --
    // 0.15 ms per slice against 1.7 ms for the one-wave-per-16 x 16-tile kernel, whose tile count explodes with K^2
    static const int big = getenv("ASB_ORTH_SYRK") ? atoi(getenv("ASB_ORTH_SYRK")) : 1;
    for (int l = 0; l < 3; ++l) {
        if (big && K >= 64) rc = asb_syrk_tn(ctx, ctx->oct + l * K, 3 * K, n, (int)K, G + (size_t)l * K * K);
        else rc = asb_gemm_tn(ctx, ctx->oct + l * K, 3 * K, ctx->oct + l * K, 3 * K, n, (int)K, (int)K, G + (size_t)l * K * K);
        if (rc) return rc;
    }
    return ASB_OK;
}

// with the (all-reduced) Gram matrices: eigen-solve, U_l = A_l V S^-1, comps[:,:,l] = U_l^T.
// sing_out (host, 3*K, optional): the singular values per dimension.  N_glob is used for scipy's
// rank tolerance max(N, K) * eps * s_max.
extern "C" int asb_orth_apply(asb_ctx* ctx, const double* G_dev, double* sing_out) {
    if (!ctx || !ctx->comps || !ctx->oct) return ASB_ERR_ARG;
    const int64_t K = ctx->K, n = ctx->n_loc;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->comps2, (size_t)K * 3 * n))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->olam, (size_t)3 * K))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->la_status, (size_t)4))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->ovec, (size_t)3 * K * K))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->osing, (size_t)3 * K))) return rc;
    if (G_dev) ASB_HIP(ctx, hipMemcpyAsync(ctx->og, G_dev, (size_t)3 * K * K * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    const double Nmax = (double)(ctx->N_glob > K ? ctx->N_glob : K);
    int st_host[4] = {0, 0, 0, 0};
    for (int l = 0; l < 3; ++l) {
        double* Gl = ctx->og + (size_t)l * K * K;
        double* Vl = ctx->ovec + (size_t)l * K * K;
        if ((rc = asb_sym_eig(ctx, Gl, (int)K, ctx->olam + l * K, Vl))) return rc;
        hipLaunchKernelGGL(k_scale_eigvecs, dim3(1), dim3(256), 0, ctx->stream, Vl, ctx->olam + l * K, (int)K,
                           Nmax * 2.220446049250313e-16, Gl, ctx->osing + l * K, ctx->la_status);
        ASB_CHECK_LAUNCH(ctx);
        int st[4];
        ASB_HIP(ctx, hipMemcpyAsync(st, ctx->la_status, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        st_host[0] |= st[0]; st_host[1] |= st[1];
        // U_l^T written straight into the new basis: out[v][j] -> comps2[j][3v + l]
        if ((rc = asb_gemm_tn_s(ctx, ctx->comps + l, 3 * n, 3, Gl, K, K, (int)n, (int)K, ctx->comps2 + l, 3, 3 * n))) return rc;
    }
    if (st_host[0] == 2) ASB_FAIL(ctx, ASB_ERR_NUMERIC, "orthogonalisation: the Jacobi eigen-solver did not converge");
    if (st_host[1]) ASB_FAIL(ctx, ASB_ERR_NUMERIC, "orthogonalisation: the basis is rank deficient in one dimension "
                                                  "(scipy.linalg.orth would drop vectors)");
    ASB_HIP(ctx, hipMemcpyAsync(ctx->comps, ctx->comps2, (size_t)K * 3 * n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    if (sing_out) {
        ASB_HIP(ctx, hipMemcpyAsync(sing_out, ctx->osing, (size_t)3 * K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return ASB_OK;
}

// T = 1.5 I - 0.5 G (one Newton-Schulz / Loewdin step towards G = I)
__global__ __launch_bounds__(256) void k_newton_schulz_T(double* __restrict__ G, int n) {
    for (int e = threadIdx.x; e < n * n; e += blockDim.x) G[e] = ((e / n == e % n) ? 1.5 : 0.0) - 0.5 * G[e];
}

// Second pass of the orthogonalisation.  U = A V S^-1 from the Gram route is orthonormal only to eps * cond(A)^2; with
// the Gram matrices of that U (asb_orth_gram again, all-reduced) one symmetric Newton-Schulz step U <- U (1.5 I - 0.5 U^T U)
// squares the defect (1e-10 -> 1e-20) and, being symmetric, does not rotate U away from the singular vectors scipy's orth returns.
extern "C" int asb_orth_refine(asb_ctx* ctx, const double* G_dev) {
    if (!ctx || !ctx->comps || !ctx->oct || !ctx->comps2) return ASB_ERR_ARG;
    const int64_t K = ctx->K, n = ctx->n_loc;
    int rc;
    if (G_dev) ASB_HIP(ctx, hipMemcpyAsync(ctx->og, G_dev, (size_t)3 * K * K * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    for (int l = 0; l < 3; ++l) {
        double* Gl = ctx->og + (size_t)l * K * K;
        hipLaunchKernelGGL(k_newton_schulz_T, dim3(1), dim3(256), 0, ctx->stream, Gl, (int)K);
        ASB_CHECK_LAUNCH(ctx);
        if ((rc = asb_gemm_tn_s(ctx, ctx->comps + l, 3 * n, 3, Gl, K, K, (int)n, (int)K, ctx->comps2 + l, 3, 3 * n))) return rc;
    }
    ASB_HIP(ctx, hipMemcpyAsync(ctx->comps, ctx->comps2, (size_t)K * 3 * n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return ASB_OK;
}

// the three K x K Gram matrices asb_orth_gram left in the context (single rank) -> host
extern "C" int asb_orth_gram_get(asb_ctx* ctx, double* G_host) {
    if (!ctx || !ctx->og || !G_host) return ASB_ERR_ARG;
    ASB_HIP(ctx, hipMemcpyAsync(G_host, ctx->og, (size_t)3 * ctx->K * ctx->K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ASB_OK;
}

// comps2 = T^T comps for ONE K x K factor T shared by the three coordinate slices (the joint QR of the POD's Rayleigh-Ritz basis,
// its rotation): a plain (K x K) (K x 3n) product on the 128 x 128-tile MFMA GEMM instead of three strided products of the
// one-wave-per-tile kernel (0.6 ms against 3 x 1.7 at K = 288, 3n = 150 000)
bool asb_combine_rows_ok(const asb_ctx* ctx) {
    static const int big = getenv("ASB_ORTH_SYRK") ? atoi(getenv("ASB_ORTH_SYRK")) : 1;
    return big && ctx->K >= 64 && !(ctx->K & 1) && !((3 * ctx->n_loc) & 1);
}
int asb_combine_rows(asb_ctx* ctx, const double* T_dev) {
    const int64_t K = ctx->K, n3 = 3 * ctx->n_loc;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->kk_tmp, (size_t)K * K))) return rc;
    if ((rc = asb_transpose(ctx, T_dev, K, K, ctx->kk_tmp))) return rc;
    return asb_gemm_nn(ctx, ctx->kk_tmp, K, ctx->comps, n3, ctx->comps2, n3, (int)K, (int)n3, (int)K, 1.0, 0.0);
}

// comps[:, :, l] <- T_l^T-combination of the components: new_j = sum_i comps_i T[l][i][j]  (T host, (3, K, K) row-major).
// The K x K factor of an orthogonalisation whose small dense step ran on the host (K > 128: orth = V S^-1 of the Gram
// matrix's eigen-decomposition, qr = L^-T of its Cholesky factor); needs asb_orth_gram to have been called.
// device form: T_dev (K x K) for all three slices (same_T) or (3, K, K); no synchronisation
int asb_components_transform_dev(asb_ctx* ctx, const double* T_dev, int same_T) {
    const int64_t K = ctx->K, n = ctx->n_loc;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->comps2, (size_t)K * 3 * n))) return rc;
    if (same_T && asb_combine_rows_ok(ctx)) {
        if ((rc = asb_combine_rows(ctx, T_dev))) return rc;
    } else {
        for (int l = 0; l < 3; ++l)
            if ((rc = asb_gemm_tn_s(ctx, ctx->comps + l, 3 * n, 3, T_dev + (same_T ? 0 : (size_t)l * K * K), K, K, (int)n, (int)K,
                                    ctx->comps2 + l, 3, 3 * n)))
                return rc;
    }
    ASB_HIP(ctx, hipMemcpyAsync(ctx->comps, ctx->comps2, (size_t)K * 3 * n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return ASB_OK;
}

extern "C" int asb_components_transform(asb_ctx* ctx, const double* T_host) {
    if (!ctx || !ctx->comps || !T_host) return ASB_ERR_ARG;
    const int64_t K = ctx->K;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->ovec, (size_t)3 * K * K))) return rc;
    ASB_HIP(ctx, hipMemcpyAsync(ctx->ovec, T_host, (size_t)3 * K * K * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = asb_components_transform_dev(ctx, ctx->ovec, 0))) return rc;
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ASB_OK;
}

// out (host, F x n_loc x 3): out[f][e][l] = sum_{j < r} comps[j][e][l] coef[l][j][f] -- the reconstruction product of
// geom_constructed (constraintsComponents.py:517-519: V_r[:, :, l] @ x for every frame) as one MFMA product per dimension
extern "C" int asb_components_expand(asb_ctx* ctx, const double* coef_host, int64_t r, int64_t Fo, double* out_host) {
    if (!ctx || !ctx->comps || !coef_host || !out_host || r < 1 || r > ctx->K || Fo < 1) return ASB_ERR_ARG;
    const int64_t n = ctx->n_loc;
    double *dc = nullptr, *dout = nullptr;
    ASB_HIP(ctx, hipMalloc((void**)&dc, (size_t)3 * r * Fo * sizeof(double)));
    hipError_t e = hipMalloc((void**)&dout, (size_t)Fo * 3 * n * sizeof(double));
    int rc = ASB_OK;
    if (e == hipSuccess) e = hipMemcpyAsync(dc, coef_host, (size_t)3 * r * Fo * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    for (int l = 0; l < 3 && e == hipSuccess && rc == ASB_OK; ++l)
        rc = asb_gemm_tn_s(ctx, ctx->comps + l, 3 * n, 3, dc + (size_t)l * r * Fo, Fo, r, (int)n, (int)Fo, dout + l, 3, 3 * n);
    if (e == hipSuccess && rc == ASB_OK)
        e = hipMemcpyAsync(out_host, dout, (size_t)Fo * 3 * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(dc);
    if (dout) (void)hipFree(dout);
    if (rc) return rc;
    if (e != hipSuccess) ASB_FAIL(ctx, ASB_ERR_HIP, "asb_components_expand: %s", hipGetErrorString(e));
    return ASB_OK;
}

// keeps the first K components of the device-resident basis (rows are contiguous: nothing moves)
extern "C" int asb_components_truncate(asb_ctx* ctx, int64_t K) {
    if (!ctx || !ctx->comps || K < 1 || K > ctx->K) return ASB_ERR_ARG;
    ctx->K = K;
    return ASB_OK;
}

// replaces the device-resident basis by a host array (K, n_loc, 3) (a caller-assigned `comps`)
extern "C" int asb_components_upload(asb_ctx* ctx, const double* comps_host, int64_t K) {
    if (!ctx || !ctx->X || !comps_host || K < 1) return ASB_ERR_ARG;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->comps, (size_t)K * 3 * ctx->n_loc))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->s_dev, (size_t)ctx->n_loc))) return rc;
    ctx->K = K;
    ASB_HIP(ctx, hipMemcpyAsync(ctx->comps, comps_host, (size_t)K * 3 * ctx->n_loc * sizeof(double), hipMemcpyHostToDevice,
                                ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ASB_OK;
}

extern "C" int asb_components_download(asb_ctx* ctx, double* comps_out) {
    if (!ctx || !ctx->comps || !comps_out) return ASB_ERR_ARG;
    ASB_HIP(ctx, hipMemcpyAsync(comps_out, ctx->comps, (size_t)ctx->K * 3 * ctx->n_loc * sizeof(double), hipMemcpyDeviceToHost,
                                ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ASB_OK;
}

// ---- the basis into pinned host memory, overlapped with the run (include/asb.h)
// The buffer is either the context's own (asb_components_stream) or the CALLER's (asb_components_stream_into: the Python
// engine hands out ndarray views of it, so its lifetime must follow those views, not this context).
static void dl_drop(asb_ctx* ctx) {
    if (ctx->dl_host && ctx->dl_host_owned) (void)hipHostFree(ctx->dl_host);
    ctx->dl_host = nullptr;
    ctx->dl_host_count = 0;
    ctx->dl_host_owned = 1;
}
static int dl_reserve(asb_ctx* ctx) {
    const size_t want = (size_t)ctx->K * 3 * ctx->n_loc;
    if (ctx->dl_host && ctx->dl_host_count >= want) return ASB_OK;
    if (ctx->dl_host && !ctx->dl_host_owned)
        ASB_FAIL(ctx, ASB_ERR_ARG, "the caller's pinned basis buffer holds %lld doubles, this run needs %lld (asb_components_stream_into)",
                 (long long)ctx->dl_host_count, (long long)want);
    dl_drop(ctx);
    if (want == 0) return ASB_OK;
    ASB_HIP(ctx, hipHostMalloc((void**)&ctx->dl_host, want * sizeof(double), hipHostMallocDefault));
    ctx->dl_host_count = want;
    ctx->dl_host_owned = 1;
    return ASB_OK;
}
static int dl_streams(asb_ctx* ctx) {
    if (!ctx->dl_stream) ASB_HIP(ctx, hipStreamCreateWithFlags(&ctx->dl_stream, hipStreamNonBlocking));
    if (!ctx->dl_event) ASB_HIP(ctx, hipEventCreateWithFlags(&ctx->dl_event, hipEventDisableTiming));
    return ASB_OK;
}
extern "C" int asb_components_stream(asb_ctx* ctx, int enable) {
    if (!ctx) return ASB_ERR_ARG;
    if (!enable) {
        if (ctx->dl_stream) (void)hipStreamSynchronize(ctx->dl_stream);
        dl_drop(ctx);
        ctx->dl_enabled = 0;
        return ASB_OK;
    }
    int rc;
    if ((rc = dl_streams(ctx))) return rc;
    ctx->dl_enabled = 1;
    ctx->dl_done = 0;
    return (ctx->K > 0 && ctx->n_loc > 0) ? dl_reserve(ctx) : ASB_OK;
}
// the same into a pinned buffer the CALLER owns (count doubles; NULL: streaming off).  The context never frees it and stops
// writing to it with the next asb_components_stream_into / asb_components_stream(ctx, 0) / asb_destroy.
extern "C" int asb_components_stream_into(asb_ctx* ctx, double* pinned_host, int64_t count) {
    if (!ctx || count < 0 || (pinned_host && count == 0)) return ASB_ERR_ARG;
    if (ctx->dl_stream) ASB_HIP(ctx, hipStreamSynchronize(ctx->dl_stream));
    dl_drop(ctx);
    if (!pinned_host) {
        ctx->dl_enabled = 0;
        return ASB_OK;
    }
    int rc;
    if ((rc = dl_streams(ctx))) return rc;
    ctx->dl_host = pinned_host;
    ctx->dl_host_count = (size_t)count;
    ctx->dl_host_owned = 0;
    ctx->dl_enabled = 1;
    ctx->dl_done = 0;
    return ASB_OK;
}
// pinned host memory for such a buffer (plain hipHostMalloc / hipHostFree: no context needed)
extern "C" int asb_host_alloc(int64_t count, double** out) {
    if (!out || count < 1) return ASB_ERR_ARG;
    *out = nullptr;
    return hipHostMalloc((void**)out, (size_t)count * sizeof(double), hipHostMallocDefault) == hipSuccess ? ASB_OK : ASB_ERR_HIP;
}
extern "C" int asb_host_free(double* p) {
    if (!p) return ASB_OK;
    return hipHostFree(p) == hipSuccess ? ASB_OK : ASB_ERR_HIP;
}
// called by asb_deflate_begin: a new run starts (K is known): the pinned buffer is (re)sized, nothing is copied yet
int asb_dl_begin(asb_ctx* ctx) {
    if (!ctx->dl_enabled) return ASB_OK;
    ASB_HIP(ctx, hipStreamSynchronize(ctx->dl_stream));      // (copies of the last run still in flight read the old basis)
    ctx->dl_done = 0;
    return dl_reserve(ctx);
}
extern "C" int asb_components_pinned(asb_ctx* ctx, double** out) {
    if (!ctx || !out || !ctx->comps) return ASB_ERR_ARG;
    if (!ctx->dl_enabled || !ctx->dl_host) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_components_pinned: asb_components_stream is off");
    if ((size_t)ctx->K * 3 * ctx->n_loc > ctx->dl_host_count)
        ASB_FAIL(ctx, ASB_ERR_ARG, "asb_components_pinned: the pinned buffer was sized for a smaller run (K grew: asb_components_download)");
    int rc;
    if ((rc = asb_dl_enqueue(ctx, ctx->K))) return rc;      // what is left (everything, for the paths that do not stream)
    ASB_HIP(ctx, hipStreamSynchronize(ctx->dl_stream));
    *out = ctx->dl_host;
    return ASB_OK;
}
