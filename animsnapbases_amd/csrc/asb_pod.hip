// Constraint-projection bases (config 5): Gram-form POD, CholeskyQR2, and the device half of
// DEIM -- constraintsComponents.compute_pod_for_vectorized_nonlinear_snapshots_tensor
// (snapbases/constraintsComponents.py:298-320), post_process_components (:415-446) and deim
// (:797-860) of the reference.  gfx950 only.
//
// POD: the reference takes the SVD of A = (3ep x F).  Here G = A^T A (F x F) is accumulated with
// f64 MFMA over the row shard (all-reduced over ranks by the caller), its eigen-pairs give S and V
// (host LAPACK for the F x F problem -- the O(3ep F^2) work is on the device), and the K leading
// left vectors are U_K = A V_K S_K^-1, 16 columns per pass over A with the deflation's projection
// kernel.  Rows r = 3e + d are exactly the (F, ep, 3) -> (3ep, F) reshape of the reference.
#include "asb_kernels.h"

int asb_project_columns(asb_ctx* ctx, const double* Wfk, int64_t ldw, int64_t k0, int ncols, double* out_rows,
                        const double* col_scale);
extern "C" int asb_sym_eig_topk(asb_ctx* ctx, double* A_dev, int64_t n64, int64_t k64, double* lam_host, double* V_host, int64_t* n_bad);

// G = X^T X over this shard's rows (F x F) into G_dev (caller's device buffer, to be all-reduced) or
// G_host (single rank convenience; synchronises)
extern "C" int asb_pod_gram(asb_ctx* ctx, double* G_dev, double* G_host) {
    if (!ctx || !ctx->X) return ASB_ERR_ARG;
    const int64_t F = ctx->F;
    int rc;
    double* G = G_dev;
    if (!G) {
        if ((rc = asb_alloc(ctx, &ctx->pod_g, (size_t)F * F))) return rc;
        G = ctx->pod_g;
    }
    if ((rc = asb_syrk_tn(ctx, ctx->X, ctx->Fp, 3 * ctx->n_loc, (int)F, G))) return rc;
    if (G_host) {
        ASB_HIP(ctx, hipMemcpyAsync(G_host, G, (size_t)F * F * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return ASB_OK;
}

// comps[i] = X . V[:, i] / sigma[i]   for i < K   (V host (F x K) row-major, sigma host (K))
extern "C" int asb_pod_basis(asb_ctx* ctx, const double* V, const double* sigma, int64_t K) {
    if (!ctx || !ctx->X || !V || !sigma || K < 1) return ASB_ERR_ARG;
    const int64_t F = ctx->F, n3 = 3 * ctx->n_loc;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->pod_v, (size_t)F * K))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->pod_s, (size_t)K))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->comps, (size_t)K * n3))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->s_dev, (size_t)ctx->n_loc))) return rc;
    ctx->K = K;
    ASB_HIP(ctx, hipMemcpyAsync(ctx->pod_v, V, (size_t)F * K * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    ASB_HIP(ctx, hipMemcpyAsync(ctx->pod_s, sigma, (size_t)K * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    for (int64_t k0 = 0; k0 < K; k0 += 16) {
        const int nc = (int)((K - k0) < 16 ? (K - k0) : 16);
        if ((rc = asb_project_columns(ctx, ctx->pod_v, K, k0, nc, ctx->comps + (size_t)k0 * n3, ctx->pod_s))) return rc;
    }
    return ASB_OK;
}

// B = Q^T A for the device-resident basis Q (K rows of length 3 n_loc) and the snapshots: B[k][f] = sum_r Q[k][r] X[r][f]
// (K x F, this shard's partial sum) into B_dev (caller's device buffer, all-reduced over ranks) and/or B_host.
// The Rayleigh-Ritz step of the POD: the small SVD of B recovers, from A itself, the accuracy the Gram matrix lost.
extern "C" int asb_pod_project(asb_ctx* ctx, double* B_dev, double* B_host) {
    if (!ctx || !ctx->X || !ctx->comps) return ASB_ERR_ARG;
    const int64_t K = ctx->K, F = ctx->F, n3 = 3 * ctx->n_loc;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->oct, (size_t)n3 * K))) return rc;
    if ((rc = asb_transpose(ctx, ctx->comps, K, n3, ctx->oct))) return rc;
    double* B = B_dev;
    if (!B) {
        if ((rc = asb_alloc(ctx, &ctx->pod_v, (size_t)F * K))) return rc;
        B = ctx->pod_v;
    }
    // (K >= 64, even: the 128 x 128-tile kernel of the Gram matrix with two operands -- 345 GFLOP at config 5)
    static const int big = getenv("ASB_ORTH_SYRK") ? atoi(getenv("ASB_ORTH_SYRK")) : 1;
    if (big && K >= 64 && !(K & 1)) rc = asb_gemm_tn_big(ctx, ctx->oct, K, ctx->X, ctx->Fp, n3, (int)K, (int)F, B);
    else rc = asb_gemm_tn(ctx, ctx->oct, K, ctx->X, ctx->Fp, n3, (int)K, (int)F, B);
    if (rc) return rc;
    if (B_host) {
        ASB_HIP(ctx, hipMemcpyAsync(B_host, B, (size_t)K * F * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return ASB_OK;
}

// X <- (X * inv_scale + mean) * rowscale     (:421-428, :440-443: the reference also restores
// nonlinearSnapshots.snapTensor); rowscale host (n_loc) or NULL
__global__ __launch_bounds__(256) void k_affine_rows(double* __restrict__ X, long long nrows, int F, int Fp, double inv_scale,
                                                     const double* __restrict__ mean, const double* __restrict__ rowscale) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (long long r = (long long)blockIdx.x * 4 + wid; r < nrows; r += (long long)gridDim.x * 4) {
        double* row = X + r * Fp;
        const double m = mean ? mean[r] : 0.0, rs = rowscale ? rowscale[r / 3] : 1.0;
        for (int f = lane; f < F; f += 64) row[f] = (row[f] * inv_scale + m) * rs;
    }
}

extern "C" int asb_snapshots_affine(asb_ctx* ctx, double inv_scale, int add_mean, const double* rowscale) {
    if (!ctx || !ctx->X) return ASB_ERR_ARG;
    if (add_mean && !ctx->have_mean) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_snapshots_affine: no mean on the device");
    const double* rs = nullptr;
    int rc;
    if (rowscale) {
        if ((rc = asb_alloc(ctx, &ctx->s_dev, (size_t)ctx->n_loc))) return rc;
        ASB_HIP(ctx, hipMemcpyAsync(ctx->s_dev, rowscale, (size_t)ctx->n_loc * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        rs = ctx->s_dev;
    }
    long long want = (ctx->n_loc * 3 + 3) / 4;
    const int grid = (int)(want < ctx->nblk_cap ? want : ctx->nblk_cap);
    { ctx->e0_valid = false; ctx->ev_valid = false; }
    hipLaunchKernelGGL(k_affine_rows, dim3(grid), dim3(256), 0, ctx->stream, ctx->X, (long long)ctx->n_loc * 3, (int)ctx->F,
                       (int)ctx->Fp, inv_scale, add_mean ? ctx->mean : nullptr, rs);
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

// --------------------------------------------------------------------------------------
// CholeskyQR: Tt[k][j] = (L^-1)[j][k] with G = L L^T, so that Q = A L^-T = A . Tt   (one block, K <= 128)
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_chol_tinv(const double* __restrict__ G, int K, double* __restrict__ Tt,
                                                   int* __restrict__ status) {
    extern __shared__ double L[];
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int e = tid; e < K * K; e += nt) L[e] = G[e];
    __syncthreads();
    for (int j = 0; j < K; ++j) {
        const double d = L[j * K + j];
        if (!(d > 0.0)) {
            if (tid == 0) status[0] = 1;
            return;
        }
        const double sq = sqrt(d);
        __syncthreads();
        for (int i = j + tid; i < K; i += nt) L[i * K + j] = (i == j) ? sq : L[i * K + j] / sq;
        __syncthreads();
        for (int e = tid; e < (K - j - 1) * (K - j - 1); e += nt) {
            const int i = j + 1 + e / (K - j - 1), c = j + 1 + e % (K - j - 1);
            if (c <= i) L[i * K + c] -= L[i * K + j] * L[c * K + j];
        }
        __syncthreads();
    }
    // column c of T = L^-1 by forward substitution: T[i][c] for i >= c
    for (int c = tid; c < K; c += nt) {
        for (int i = 0; i < K; ++i) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int j = c; j < i; ++j) s -= L[i * K + j] * Tt[(long long)c * K + j];      // Tt[c][j] = T[j][c]
            Tt[(long long)c * K + i] = (i < c) ? 0.0 : s / L[i * K + i];
        }
    }
}

// sum of the three per-dimension Gram matrices (the joint Gram of the (3 n) x K basis) into slice 0
__global__ __launch_bounds__(256) void k_sum3(double* __restrict__ G, long long kk) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < kk; e += (long long)gridDim.x * 256)
        G[e] = (G[e] + G[kk + e]) + G[2 * kk + e];
}

// one CholeskyQR pass with the (all-reduced) Gram matrices of asb_orth_gram: comps[:,:,l] <- (A_l L_l^-T)^T per dimension
// (joint = 0: `qr(comps[:,:,l].T, mode='economic')[0].T`, constraintsComponents.py:431-435), or with ONE factor of the summed
// Gram matrix for all three slices (joint = 1: orthonormal (3 n)-vectors, the Rayleigh-Ritz basis of the POD refinement).
// K <= 128: Cholesky + inverse in one block's LDS; larger K: blocked (asb_smalldense.hip).  Call twice for CholeskyQR2.
static int qr_apply(asb_ctx* ctx, const double* G_dev, int joint) {
    if (!ctx || !ctx->comps || !ctx->oct) return ASB_ERR_ARG;
    const int64_t K = ctx->K, n = ctx->n_loc;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->comps2, (size_t)K * 3 * n))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->ovec, (size_t)3 * K * K))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->la_status, (size_t)4))) return rc;
    if (G_dev) ASB_HIP(ctx, hipMemcpyAsync(ctx->og, G_dev, (size_t)3 * K * K * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    ASB_HIP(ctx, hipMemsetAsync(ctx->la_status, 0, 4 * sizeof(int), ctx->stream));
    if (joint) hipLaunchKernelGGL(k_sum3, dim3(256), dim3(256), 0, ctx->stream, ctx->og, (long long)K * K);
    const size_t lds = (size_t)K * K * sizeof(double);
    if (K <= 128 && lds > 48 * 1024)
        ASB_HIP(ctx, hipFuncSetAttribute((const void*)k_chol_tinv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int l = 0; l < (joint ? 1 : 3); ++l) {
        double* Tt = ctx->ovec + (size_t)l * K * K;
        if (K <= 128) {
            hipLaunchKernelGGL(k_chol_tinv, dim3(1), dim3(256), lds, ctx->stream, ctx->og + (size_t)l * K * K, (int)K, Tt, ctx->la_status);
            ASB_CHECK_LAUNCH(ctx);
        } else if ((rc = asb_chol_tinv_dev(ctx, ctx->og + (size_t)l * K * K, (int)K, Tt, ctx->la_status))) {
            return rc;
        }
    }
    if (joint && asb_combine_rows_ok(ctx)) {
        if ((rc = asb_combine_rows(ctx, ctx->ovec))) return rc;
    } else {
        for (int l = 0; l < 3; ++l)
            if ((rc = asb_gemm_tn_s(ctx, ctx->comps + l, 3 * n, 3, ctx->ovec + (joint ? 0 : (size_t)l * K * K), K, K, (int)n, (int)K,
                                    ctx->comps2 + l, 3, 3 * n)))
                return rc;
    }
    int st[4];
    ASB_HIP(ctx, hipMemcpyAsync(st, ctx->la_status, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (st[0]) ASB_FAIL(ctx, ASB_ERR_NUMERIC, "QR: a coordinate slice of the basis is rank deficient");
    ASB_HIP(ctx, hipMemcpyAsync(ctx->comps, ctx->comps2, (size_t)K * 3 * n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return ASB_OK;
}
extern "C" int asb_qr_apply(asb_ctx* ctx, const double* G_dev) { return qr_apply(ctx, G_dev, 0); }
extern "C" int asb_qr_apply_joint(asb_ctx* ctx, const double* G_dev) { return qr_apply(ctx, G_dev, 1); }

// sigma[i] = sqrt(max(lam[i], 0))
__global__ __launch_bounds__(256) void k_sqrt_pos(const double* __restrict__ lam, long long n, double* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = sqrt(fmax(lam[i], 0.0));
}

// out (cols x rows) = in (rows x cols)^T with row c of the result divided by scale[c]
__global__ __launch_bounds__(256) void k_transpose_div(const double* __restrict__ in, long long rows, long long cols,
                                                       const double* __restrict__ scale, double* __restrict__ out) {
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
        if (r < rows && c < cols) out[c * rows + r] = tile[tx][ty + q * 8] / scale[c];
    }
}

// comps[i] = X . V[:, i] / sigma[i] for i < K from the eigen-pairs asb_sym_eig_topk left on the device (no host copy)
extern "C" int asb_pod_basis_dev(asb_ctx* ctx, int64_t K) {
    if (!ctx || !ctx->X || K < 1) return ASB_ERR_ARG;
    if (!ctx->eig_v || ctx->eig_n != ctx->F || K > ctx->eig_k)
        ASB_FAIL(ctx, ASB_ERR_ARG, "asb_pod_basis_dev: asb_sym_eig_topk has not left %lld vectors of an F = %lld problem", (long long)K,
                 (long long)ctx->F);
    const int64_t n3 = 3 * ctx->n_loc;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->pod_s, (size_t)K))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->comps, (size_t)K * n3))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->s_dev, (size_t)ctx->n_loc))) return rc;
    ctx->K = K;
    hipLaunchKernelGGL(k_sqrt_pos, dim3(4), dim3(256), 0, ctx->stream, ctx->eig_lam, (long long)K, ctx->pod_s);
    ASB_CHECK_LAUNCH(ctx);
    // K >= 64: one (3n x F)(F x K) product on the tiled MFMA GEMM and a scaled transposition instead of K / 16 passes of the
    // 16-column projection kernel over the 4.8 GB tensor (config 5: 18 x 0.91 ms -> ~ 8 ms)
    static const int big = getenv("ASB_ORTH_SYRK") ? atoi(getenv("ASB_ORTH_SYRK")) : 1;
    if (big && K >= 64 && !(K & 1) && !(n3 & 1) && !(ctx->F & 1) && !(ctx->eig_k & 1)) {
        if ((rc = asb_alloc(ctx, &ctx->comps2, (size_t)K * n3))) return rc;
        if ((rc = asb_gemm_nn(ctx, ctx->X, ctx->Fp, ctx->eig_v, ctx->eig_k, ctx->comps2, K, (int)n3, (int)K, (int)ctx->F, 1.0, 0.0))) return rc;
        hipLaunchKernelGGL(k_transpose_div, dim3((unsigned)((K + 31) / 32), (unsigned)((n3 + 31) / 32)), dim3(256), 0, ctx->stream,
                           ctx->comps2, (long long)n3, (long long)K, ctx->pod_s, ctx->comps);
        ASB_CHECK_LAUNCH(ctx);
        return ASB_OK;
    }
    for (int64_t k0 = 0; k0 < K; k0 += 16) {
        const int nc = (int)((K - k0) < 16 ? (K - k0) : 16);
        if ((rc = asb_project_columns(ctx, ctx->eig_v, ctx->eig_k, k0, nc, ctx->comps + (size_t)k0 * n3, ctx->pod_s))) return rc;
    }
    return ASB_OK;
}

// V[f][k] /= sqrt(lam[k]) (columns scaled); status[1] = 1 when lam[k] is not resolved (<= tol_rel^2 lam[0])
__global__ __launch_bounds__(256) void k_scale_cols_inv_sqrt(double* __restrict__ V, int F, int k, const double* __restrict__ lam,
                                                             double tol_rel, int* __restrict__ status) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < (long long)F * k; e += (long long)gridDim.x * 256) {
        const int j = (int)(e % k);
        const double l = lam[j];
        if (!(l > tol_rel * tol_rel * lam[0])) { if (e < k) status[1] = 1; V[e] = 0.0; }
        else V[e] /= sqrt(l);
    }
}

// constProj_basis_type 'pod' (compute_pod_for_nonlinear_snapshots_tensor, constraintsComponents.py:274-294): one SVD per
// (constraint row p_i, coordinate d) of the e x F matrix M[e_i][f] = snapshots[f][e_i p + p_i][d]; component k holds, in
// those rows, the k-th left singular vector.  Here per slice: Gram matrix (F x F) of the slice's rows -> eigen-problem on
// the device -> U = M V S^-1 written straight into the strided rows of the basis.  (The reference computes these SVDs in
// float32 on the CPU with torch; this is the float64 result.)  One rank holds all rows.
extern "C" int asb_pod_slices(asb_ctx* ctx, int p, int64_t K) {
    if (!ctx || !ctx->X || p < 1 || K < 1) return ASB_ERR_ARG;
    if (ctx->n_loc != ctx->N_glob) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_pod_slices needs all rows on one rank");
    if (ctx->n_loc % p) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_pod_slices: %lld rows are not whole constraints of %d", (long long)ctx->n_loc, p);
    const int64_t F = ctx->F, n3 = 3 * ctx->n_loc, e = ctx->n_loc / p;
    if (K > F || K > e) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_pod_slices: K = %lld exceeds min(e, F) = %lld", (long long)K, (long long)(e < F ? e : F));
    if (F < 3) ASB_FAIL(ctx, ASB_ERR_LIMIT, "asb_pod_slices needs F >= 3");
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->pod_g, (size_t)F * F))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->comps, (size_t)K * n3))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->s_dev, (size_t)ctx->n_loc))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->la_status, (size_t)4))) return rc;
    ctx->K = K;
    std::vector<double> lam((size_t)F);
    const long long stride = 3LL * p * ctx->Fp;                 // between the rows of consecutive constraints of one slice
    for (int s = 0; s < 3 * p; ++s) {                           // slice s: rows r = 3 (e_i p + p_i) + d with 3 p_i + d = s
        const double* Xs = ctx->X + (long long)s * ctx->Fp;
        if ((rc = asb_gemm_tn_s(ctx, Xs, stride, 1, Xs, stride, e, (int)F, (int)F, ctx->pod_g, F, 1))) return rc;
        int64_t bad = 0;
        if ((rc = asb_sym_eig_topk(ctx, ctx->pod_g, F, K, lam.data(), nullptr, &bad))) return rc;
        ASB_HIP(ctx, hipMemsetAsync(ctx->la_status, 0, 4 * sizeof(int), ctx->stream));
        hipLaunchKernelGGL(k_scale_cols_inv_sqrt, dim3(64), dim3(256), 0, ctx->stream, ctx->eig_v, (int)F, (int)K, ctx->eig_lam, 1e-7,
                           ctx->la_status);
        ASB_CHECK_LAUNCH(ctx);
        // U[e_i][k] = sum_f M[e_i][f] V[f][k]  ->  comps[k][r(e_i)]
        if ((rc = asb_gemm_tn_s(ctx, Xs, 1, stride, ctx->eig_v, K, F, (int)e, (int)K, ctx->comps + s, 3LL * p, n3))) return rc;
        int st[4];
        ASB_HIP(ctx, hipMemcpyAsync(st, ctx->la_status, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (st[1]) ASB_FAIL(ctx, ASB_ERR_NUMERIC, "pod: slice %d has fewer than %lld singular values above 1e-7 of its largest", s, (long long)K);
    }
    return ASB_OK;
}

// Rayleigh-Ritz rotation: the (all-reduced) K x F matrix B = Q^T A (B_dev, or the one asb_pod_project left in the context;
// overwritten) -> its singular values S_host (K, descending) and left vectors U_B by one-sided Jacobi on its rows, then
// basis <- Q U_B.  The device counterpart of the small SVD the reference gets from gesdd on A itself (:307).
extern "C" int asb_pod_rotate(asb_ctx* ctx, double* B_dev, double* S_host) {
    if (!ctx || !ctx->comps || !S_host) return ASB_ERR_ARG;
    const int64_t K = ctx->K, F = ctx->F;
    double* B = B_dev ? B_dev : ctx->pod_v;
    if (!B) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_pod_rotate: no B (run asb_pod_project first or pass B_dev)");
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->ovec, (size_t)3 * K * K))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->olam, (size_t)3 * K))) return rc;
    if ((rc = asb_jacobi_rows_dev(ctx, B, (int)K, (int)F, F, ctx->ovec, 1, ctx->olam, nullptr))) return rc;
    if ((rc = asb_components_transform_dev(ctx, ctx->ovec, 1))) return rc;
    ASB_HIP(ctx, hipMemcpyAsync(S_host, ctx->olam, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ASB_OK;
}

// One step of subspace iteration on A A^T from the Rayleigh-Ritz result (round 4): the rows of the rotated B are sigma_i v_i^T
// (v_i = A^T u_i / sigma_i, the right Ritz vectors -- exact images of the basis under A^T), so basis <- A V Sigma^-1 is A A^T
// applied to every basis vector and scaled back to unit length.  What it buys: the Gram route leaves each weak vector with an
// error ~ eps (sigma_0 / sigma_k)^2 that points OUT of the K + 32-dimensional Ritz subspace -- Rayleigh-Ritz on A cannot see it
// -- spread over all the directions the subspace lacks; this step multiplies the part along direction j by (sigma_j / sigma_k)^2,
// and most of those directions have singular values far below sigma_k.  The caller re-orthogonalises (CholeskyQR2: the new
// vectors are the old ones plus small corrections) and repeats project + rotate.  B_dev: the buffer asb_pod_rotate worked on.
__global__ __launch_bounds__(256) void k_power_v(const double* __restrict__ B, int F, int K, const int* __restrict__ where,
                                                 const double* __restrict__ sig, double* __restrict__ Vn) {
    // Vn (F x K) row-major: column r = row where[r] of B / sigma_r^2
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < (long long)F * K; e += (long long)gridDim.x * 256) {
        const int f = (int)(e / K), r = (int)(e % K);
        const double s = sig[r];
        Vn[e] = s > 0.0 ? B[(long long)where[r] * F + f] / (s * s) : 0.0;
    }
}
extern "C" int asb_pod_power(asb_ctx* ctx, const double* B_dev) {
    if (!ctx || !ctx->X || !ctx->comps || !ctx->jac_where || !ctx->olam) return ASB_ERR_ARG;
    const int64_t K = ctx->K, F = ctx->F, n3 = 3 * ctx->n_loc;
    const double* B = B_dev ? B_dev : ctx->pod_v;
    if (!B) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_pod_power: no rotated B (asb_pod_rotate first)");
    if ((K | F | n3) & 1) ASB_FAIL(ctx, ASB_ERR_LIMIT, "asb_pod_power: odd dimension (K = %lld, F = %lld, rows = %lld)", (long long)K,
                                   (long long)F, (long long)n3);
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->pod_vn, (size_t)F * K))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->comps2, (size_t)K * n3))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->pod_s, (size_t)K))) return rc;
    hipLaunchKernelGGL(k_power_v, dim3(1024), dim3(256), 0, ctx->stream, B, (int)F, (int)K, ctx->jac_where, ctx->olam, ctx->pod_vn);
    ASB_CHECK_LAUNCH(ctx);
    // (3n x F)(F x K) on the tiled MFMA GEMM, then the transposition back to component-major rows (scale 1)
    if ((rc = asb_gemm_nn(ctx, ctx->X, ctx->Fp, ctx->pod_vn, K, ctx->comps2, K, (int)n3, (int)K, (int)F, 1.0, 0.0))) return rc;
    return asb_transpose(ctx, ctx->comps2, n3, K, ctx->comps);
}

// ---- the POD in LEVELS (round 4): what the Gram matrix of A cannot resolve -- singular values below ~1e-8 sigma_0, where its
// entries are rounding noise of the strong directions -- is resolved by the Gram matrix of the DEFLATED snapshots A_2 = A - U_1
// (U_1^T A), whose largest singular value is the first one level 1 left out.  asb_pod_deflate_begin keeps the level's basis U_1
// (the first `keep` rows of the context's basis; appended to what earlier levels kept) and replaces the context's snapshot
// tensor by A_2 (a second buffer: the original is restored by asb_pod_deflate_end); every POD entry point then works on A_2 as
// it did on A.  asb_pod_deflate_end restores the snapshots and installs [kept bases of all levels ; first `last` rows of the
// current basis] as the context's basis.  B_dev: the (all-reduced) buffer asb_pod_rotate worked on (rows sigma_i v_i^T, unsorted).
__global__ __launch_bounds__(256) void k_gather_b(const double* __restrict__ B, int F, int K, const int* __restrict__ where,
                                                  double* __restrict__ out) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < (long long)K * F; e += (long long)gridDim.x * 256)
        out[e] = B[(long long)where[e / F] * F + e % F];
}
extern "C" int asb_pod_deflate_begin(asb_ctx* ctx, const double* B_dev, int64_t keep) {
    if (!ctx || !ctx->X || !ctx->comps || !ctx->jac_where) return ASB_ERR_ARG;
    const int64_t K = ctx->K, F = ctx->F, n3 = 3 * ctx->n_loc;
    const double* B = B_dev ? B_dev : ctx->pod_v;
    if (!B || keep < 1 || keep > K) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_pod_deflate_begin: keep = %lld of %lld", (long long)keep, (long long)K);
    if ((keep | F | n3 | ctx->Fp) & 1) ASB_FAIL(ctx, ASB_ERR_LIMIT, "asb_pod_deflate_begin: odd dimension");
    int rc;
    // the kept rows join the earlier levels' (pod_u1 grows: old content first)
    const size_t have = (size_t)ctx->pod_u1_rows * n3, add = (size_t)keep * n3;
    double* u1 = nullptr;
    ASB_HIP(ctx, hipMalloc((void**)&u1, (have + add) * sizeof(double)));
    if (have) ASB_HIP(ctx, hipMemcpyAsync(u1, ctx->pod_u1, have * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    ASB_HIP(ctx, hipMemcpyAsync(u1 + have, ctx->comps, add * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->pod_u1) (void)hipFree(ctx->pod_u1);
    ctx->pod_u1 = u1;
    ctx->pod_u1_rows += keep;
    // B_1 = U_1^T A in sorted order = the gathered rows of the rotated B; oct_1 = U_1^T (3n x keep)
    if ((rc = asb_alloc(ctx, &ctx->pod_vn, (size_t)F * K))) return rc;
    hipLaunchKernelGGL(k_gather_b, dim3(1024), dim3(256), 0, ctx->stream, B, (int)F, (int)keep, ctx->jac_where, ctx->pod_vn);
    ASB_CHECK_LAUNCH(ctx);
    if ((rc = asb_alloc(ctx, &ctx->comps2, (size_t)K * n3))) return rc;
    if ((rc = asb_transpose(ctx, ctx->comps, keep, n3, ctx->comps2))) return rc;          // (3n x keep)
    // A_2 = A - oct_1 B_1 into the second snapshot buffer
    const size_t nx = (size_t)n3 * ctx->Fp;
    if (!ctx->X_deflated) {
        ASB_HIP(ctx, hipMalloc((void**)&ctx->X_deflated, nx * sizeof(double)));
        ASB_HIP(ctx, hipMemcpyAsync(ctx->X_deflated, ctx->X, nx * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        ctx->X_original = ctx->X;
        ctx->X = ctx->X_deflated;
    }
    if ((rc = asb_gemm_nn(ctx, ctx->comps2, keep, ctx->pod_vn, F, ctx->X, ctx->Fp, (int)n3, (int)F, (int)keep, -1.0, 1.0))) return rc;
    ctx->e0_valid = false;
    ctx->ev_valid = false;
    return ASB_OK;
}
extern "C" int asb_pod_deflate_end(asb_ctx* ctx, int64_t last) {
    if (!ctx || !ctx->X) return ASB_ERR_ARG;
    if (!ctx->X_original) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_pod_deflate_end without asb_pod_deflate_begin");
    const int64_t n3 = 3 * ctx->n_loc;
    if (last < 0 || last > ctx->K) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_pod_deflate_end: last = %lld of %lld", (long long)last, (long long)ctx->K);
    const int64_t Kt = ctx->pod_u1_rows + last;
    double* nb = nullptr;
    ASB_HIP(ctx, hipMalloc((void**)&nb, (size_t)Kt * n3 * sizeof(double)));
    ASB_HIP(ctx, hipMemcpyAsync(nb, ctx->pod_u1, (size_t)ctx->pod_u1_rows * n3 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    if (last)
        ASB_HIP(ctx, hipMemcpyAsync(nb + (size_t)ctx->pod_u1_rows * n3, ctx->comps, (size_t)last * n3 * sizeof(double),
                                    hipMemcpyDeviceToDevice, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // the context's basis buffer is replaced (asb_alloc's bookkeeping follows the pointer variable)
    if (ctx->comps) (void)hipFree(ctx->comps);
    ctx->comps = nb;
    ctx->alloc_bytes[(void*)&ctx->comps] = (size_t)Kt * n3 * sizeof(double);
    ctx->K = Kt;
    ctx->X = ctx->X_original;
    ctx->X_original = nullptr;
    (void)hipFree(ctx->X_deflated);
    ctx->X_deflated = nullptr;
    (void)hipFree(ctx->pod_u1);
    ctx->pod_u1 = nullptr;
    ctx->pod_u1_rows = 0;
    ctx->e0_valid = false;
    ctx->ev_valid = false;
    return ASB_OK;
}

// --------------------------------------------------------------------------------------
// DEIM, device half (:820-836): r[e,i] = sum_{j<k} coef[i][j] V[e,j,i] - V[e,k,i],  idx = argmax_e sum_i r^2.
// V[e,j,i] = comps[j][3e+i].  The k x k interpolation solves stay on the host (numpy lstsq, the
// routine the reference calls), fed by asb_deim_row.
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_deim_residual(const double* __restrict__ comps, long long n_vert, int k,
                                                       const double* __restrict__ coef, long long v0,
                                                       double* __restrict__ pmax, long long* __restrict__ pidx,
                                                       double* __restrict__ pabs = nullptr) {
    extern __shared__ double cf[];          // 3 * k
    __shared__ double sh_d[256];
    __shared__ long long sh_i[256];
    __shared__ double sh_a[4];
    double amax = 0.0;
    for (int q = threadIdx.x; q < 3 * k; q += blockDim.x) cf[q] = coef[q];
    __syncthreads();
    const long long stride = 3 * n_vert;
    double be = -1.0;
    long long bi = 0x7fffffffffffffffLL;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n_vert; e += (long long)gridDim.x * blockDim.x) {
        double r0 = 0.0, r1 = 0.0, r2 = 0.0;
        for (int j = 0; j < k; ++j) {
            const double* p = comps + (long long)j * stride + 3 * e;
            r0 += cf[j] * p[0]; r1 += cf[k + j] * p[1]; r2 += cf[2 * k + j] * p[2];
        }
        const double* q = comps + (long long)k * stride + 3 * e;
        r0 -= q[0]; r1 -= q[1]; r2 -= q[2];
        const double en = r0 * r0 + r1 * r1 + r2 * r2;
        if (am_better(en, e, be, bi)) { be = en; bi = e; }
        amax = fmax(amax, fmax(fabs(r0), fmax(fabs(r1), fabs(r2))));
    }
    if (pabs) {
        amax = wave_max(amax);
        if ((threadIdx.x & 63) == 0) sh_a[threadIdx.x >> 6] = amax;
    }
    sh_d[threadIdx.x] = be; sh_i[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o && am_better(sh_d[threadIdx.x + o], sh_i[threadIdx.x + o], sh_d[threadIdx.x], sh_i[threadIdx.x])) {
            sh_d[threadIdx.x] = sh_d[threadIdx.x + o];
            sh_i[threadIdx.x] = sh_i[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        pmax[blockIdx.x] = sh_d[0]; pidx[blockIdx.x] = v0 + sh_i[0];
        if (pabs) pabs[blockIdx.x] = fmax(fmax(sh_a[0], sh_a[1]), fmax(sh_a[2], sh_a[3]));
    }
}

// ---- the whole DEIM loop on the device (one rank holds every row) -------------------------------------------------------
// State: Mx (3, K, K) with Mx[i][m][j] = V[Pt[m], j, i] (row m = the m-th interpolation point, all K columns), Minv (3, K, K)
// whose leading k x k block is (Mx[i][:k, :k])^-1, carried along by the bordering (Schur complement) update -- what the
// reference recomputes from scratch with lstsq at every step (:829).
// k_deim_solve (one block per dimension): grows the inverse by last step's point / vector, then coef[i] = Minv b with
// b = Mx[i][:k, k]; the solve is verified (|M x - b| against rounding level), a failure raises flags[0] and the caller
// repeats the loop with the reference's lstsq on the host.
// Both the matrices of points (Mx, and MxT[i][j][m] = Mx[i][m][j]) and the inverse (Minv and its transpose MinvT) are kept in
// two layouts so that every product below reads consecutive words across a wave.
// 1024 threads: thread (r = tid % 256, part = tid / 256) takes every fourth term of row r's dot products (four times shorter
// latency chains), the parts meet in LDS.
#define DS_T 1024
__global__ __launch_bounds__(DS_T) void k_deim_solve(const double* __restrict__ Mx, const double* __restrict__ MxT,
                                                    double* __restrict__ Minv, double* __restrict__ MinvT, int K, int k,
                                                    double* __restrict__ coef, int* __restrict__ flags) {
    extern __shared__ double sh[];               // u (K), w (K), x (K), pu (4 K), pw (4 K), red (64)
    double* u = sh;
    double* w = u + K;
    double* x = w + K;
    double* pu = x + K;
    double* pw = pu + 4 * K;
    double* red = pw + 4 * K;
    const int i = blockIdx.x, tid = threadIdx.x, lane_r = tid & 255, part = tid >> 8;
    const double* M = Mx + (size_t)i * K * K;
    const double* MT = MxT + (size_t)i * K * K;
    double* A = Minv + (size_t)i * K * K;
    double* AT = MinvT + (size_t)i * K * K;
    const int n = k - 1;                         // size of the inverse carried over
    if (k == 1) {
        if (tid == 0) {
            const double m00 = M[0];
            if (m00 == 0.0 || !(m00 == m00)) flags[0] = 1;
            A[0] = 1.0 / m00;
            AT[0] = 1.0 / m00;
        }
    } else {
        // u = A bcol, w = crow A   (bcol[q] = M[q][n] = MT[n][q], crow[q] = M[n][q])
        const double* bcol = MT + (size_t)n * K;
        const double* crow = M + (size_t)n * K;
        for (int r0 = 0; r0 < n; r0 += 256) {
            const int r = r0 + lane_r;
            double su = 0.0, sw = 0.0;
            if (r < n)
                for (int q = part; q < n; q += 4) {
                    su += AT[(size_t)q * K + r] * bcol[q];        // A[r][q]
                    sw += crow[q] * A[(size_t)q * K + r];
                }
            if (r < n) { pu[part * K + r] = su; pw[part * K + r] = sw; }
        }
        __syncthreads();
        for (int r = tid; r < n; r += DS_T) {
            u[r] = (pu[r] + pu[K + r]) + (pu[2 * K + r] + pu[3 * K + r]);
            w[r] = (pw[r] + pw[K + r]) + (pw[2 * K + r] + pw[3 * K + r]);
        }
        __syncthreads();
        double part1[1] = {0.0};
        for (int q = tid; q < n; q += DS_T) part1[0] += crow[q] * u[q];
        block_sum<1>(part1, red);
        const double sch = crow[n] - part1[0];
        if (sch == 0.0 || !(sch == sch) || fabs(sch) > 1.7e308) {
            if (tid == 0) flags[0] = 1;
            return;
        }
        const double is = 1.0 / sch;
        for (int e = tid; e < n * n; e += DS_T) {
            const int r = e / n, c = e % n;
            A[(size_t)r * K + c] += u[r] * w[c] * is;
            AT[(size_t)r * K + c] += u[c] * w[r] * is;        // AT[r][c] = A[c][r]
        }
        for (int r = tid; r < n; r += DS_T) {
            A[(size_t)r * K + n] = -u[r] * is;
            A[(size_t)n * K + r] = -w[r] * is;
            AT[(size_t)n * K + r] = -u[r] * is;
            AT[(size_t)r * K + n] = -w[r] * is;
        }
        if (tid == 0) { A[(size_t)n * K + n] = is; AT[(size_t)n * K + n] = is; }
    }
    __threadfence_block();
    __syncthreads();
    // x = A[:k, :k] b,  b[m] = M[m][k] = MT[k][m]
    const double* b = MT + (size_t)k * K;
    double nm2 = 0.0;
    for (int r0 = 0; r0 < k; r0 += 256) {
        const int r = r0 + lane_r;
        double sx = 0.0;
        if (r < k)
            for (int q = part; q < k; q += 4) {
                sx += AT[(size_t)q * K + r] * b[q];
                const double m = MT[(size_t)q * K + r];
                nm2 += m * m;
            }
        if (r < k) pu[part * K + r] = sx;
    }
    __syncthreads();
    double nb2 = 0.0, nx2 = 0.0;
    for (int r = tid; r < k; r += DS_T) {
        const double sx = (pu[r] + pu[K + r]) + (pu[2 * K + r] + pu[3 * K + r]);
        x[r] = sx;
        coef[(size_t)i * k + r] = sx;
        nx2 += sx * sx;
        nb2 += b[r] * b[r];
    }
    __syncthreads();
    for (int r0 = 0; r0 < k; r0 += 256) {
        const int r = r0 + lane_r;
        double sr = 0.0;
        if (r < k)
            for (int q = part; q < k; q += 4) sr += MT[(size_t)q * K + r] * x[q];     // M[r][q]
        if (r < k) pw[part * K + r] = sr;
    }
    __syncthreads();
    double res2 = 0.0;
    for (int r = tid; r < k; r += DS_T) {
        const double sr = (pw[r] + pw[K + r]) + (pw[2 * K + r] + pw[3 * K + r]) - b[r];
        res2 += sr * sr;
    }
    double v[4] = {nb2, nx2, nm2, res2};
    block_sum<4>(v, red);
    if (tid == 0) {
        const double lim = 1e-10 * (sqrt(v[0]) + sqrt(v[2]) * sqrt(v[1]));
        if (!(sqrt(v[3]) <= lim)) flags[0] = 1;
    }
}

// reduces the residual kernel's partials to the step's point: Pt[k], its largest |r| entry, and row k of Mx
__global__ __launch_bounds__(256) void k_deim_pick(const double* __restrict__ pmax, const long long* __restrict__ pidx,
                                                   const double* __restrict__ pabs, int nblk, const double* __restrict__ comps,
                                                   long long n_vert, int K, int k, long long* __restrict__ Pt,
                                                   double* __restrict__ maxabs, double* __restrict__ Mx, double* __restrict__ MxT) {
    __shared__ double sh_d[256];
    __shared__ long long sh_i[256];
    __shared__ double sh_a[256];
    double be = -1.0, am = 0.0;
    long long bi = 0x7fffffffffffffffLL;
    for (int b = threadIdx.x; b < nblk; b += 256) {
        if (am_better(pmax[b], pidx[b], be, bi)) { be = pmax[b]; bi = pidx[b]; }
        am = fmax(am, pabs[b]);
    }
    sh_d[threadIdx.x] = be; sh_i[threadIdx.x] = bi; sh_a[threadIdx.x] = am;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            if (am_better(sh_d[threadIdx.x + o], sh_i[threadIdx.x + o], sh_d[threadIdx.x], sh_i[threadIdx.x])) {
                sh_d[threadIdx.x] = sh_d[threadIdx.x + o];
                sh_i[threadIdx.x] = sh_i[threadIdx.x + o];
            }
            sh_a[threadIdx.x] = fmax(sh_a[threadIdx.x], sh_a[threadIdx.x + o]);
        }
        __syncthreads();
    }
    const long long idx = sh_i[0];
    if (threadIdx.x == 0) { Pt[k] = idx; maxabs[k] = sh_a[0]; }
    for (int q = threadIdx.x; q < 3 * K; q += 256) {
        const int i = q / K, j = q % K;
        const double val = comps[(long long)j * 3 * n_vert + 3 * idx + i];
        Mx[(size_t)i * K * K + (size_t)k * K + j] = val;
        MxT[(size_t)i * K * K + (size_t)j * K + k] = val;
    }
}

// deim (:797-860) for a basis whose every row is on this device: Pt_out (K) the interpolation rows in order, maxabs_out (K)
// the largest |residual| entry of each step (the reference stops with "zero residual" when np.allclose(r, 0), i.e. when it
// is <= 1e-8), *solve_failed != 0 when a bordered solve missed its check (the caller then uses lstsq on the host).  One
// host synchronisation for the whole loop.
extern "C" int asb_deim_run(asb_ctx* ctx, int64_t* Pt_out, double* maxabs_out, int* solve_failed) {
    if (!ctx || !ctx->comps || !Pt_out || !maxabs_out || !solve_failed) return ASB_ERR_ARG;
    if (ctx->n_loc != ctx->N_glob) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_deim_run needs all rows on one rank");
    ASB_HIP(ctx, hipSetDevice(ctx->dev));         // (the host mirror calls this from a worker thread, beside its LAPACK rank check)
    const int K = (int)ctx->K;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->deim_m, (size_t)12 * K * K + 4 * (size_t)K + 2048 * 2))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->deim_pt, (size_t)K + 2048))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->la_status, (size_t)4))) return rc;
    double* Mx = ctx->deim_m;
    double* MxT = Mx + (size_t)3 * K * K;
    double* Minv = MxT + (size_t)3 * K * K;
    double* MinvT = Minv + (size_t)3 * K * K;
    double* coef = MinvT + (size_t)3 * K * K;         // 3 K
    double* maxabs = coef + (size_t)3 * K;            // K
    double* pmax = maxabs + K;                        // 2048
    double* pabs = pmax + 2048;                       // 2048
    long long* Pt = ctx->deim_pt;
    long long* pidx = Pt + K;
    ASB_HIP(ctx, hipMemsetAsync(ctx->la_status, 0, 4 * sizeof(int), ctx->stream));
    long long want = (ctx->n_loc + 255) / 256;
    const int grid = (int)(want < 1024 ? want : 1024);
    const size_t lds_solve = ((size_t)11 * K + 64) * sizeof(double);
    if (lds_solve > 48 * 1024)
        ASB_HIP(ctx, hipFuncSetAttribute((const void*)k_deim_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_solve));
    for (int k = 0; k < K; ++k) {
        if (k > 0) {
            hipLaunchKernelGGL(k_deim_solve, dim3(3), dim3(DS_T), lds_solve, ctx->stream, Mx, MxT, Minv, MinvT, K, k, coef, ctx->la_status);
        }
        hipLaunchKernelGGL(k_deim_residual, dim3(grid), dim3(256), (size_t)(3 * k + 1) * sizeof(double), ctx->stream, ctx->comps,
                           (long long)ctx->n_loc, k, coef, (long long)ctx->v0, pmax, pidx, pabs);
        hipLaunchKernelGGL(k_deim_pick, dim3(1), dim3(256), 0, ctx->stream, pmax, pidx, pabs, grid, ctx->comps, (long long)ctx->n_loc,
                           K, k, Pt, maxabs, Mx, MxT);
    }
    ASB_CHECK_LAUNCH(ctx);
    int st[4];
    ASB_HIP(ctx, hipMemcpyAsync(Pt_out, Pt, (size_t)K * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipMemcpyAsync(maxabs_out, maxabs, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipMemcpyAsync(st, ctx->la_status, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *solve_failed = st[0];
    return ASB_OK;
}

// ---- block DEIM (deim_blocksForm :733-795, geom_block_form_utilizing_differential_operator :619-731 in the constraint
// space): step k works on the p basis vectors of block k.  r[row][m][i] = sum_{j < k p} coef[i][j][m] V[row][j][i] -
// V[row][k p + m][i]; the per-row energies sum_{m, i} r^2 go to ctx->energy, the caller takes the arg-max over rows
// (group = 1) or over constraints of p rows (group = p) with asb_deflate_block_argmax.  coef host (3, k p, p), NULL at k = 0.
__global__ __launch_bounds__(256) void k_deim_block_residual(const double* __restrict__ comps, long long n_rows, int kp, int p,
                                                             const double* __restrict__ coef, double* __restrict__ energy,
                                                             double* __restrict__ pabs) {
    __shared__ double sh_a[4];
    const long long stride = 3 * n_rows;
    double amax = 0.0;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n_rows; e += (long long)gridDim.x * 256) {
        double en = 0.0;
        for (int m = 0; m < p; ++m) {
            double r[3] = {0.0, 0.0, 0.0};
            for (int j = 0; j < kp; ++j) {
                const double* q = comps + (long long)j * stride + 3 * e;
#pragma unroll
                for (int i = 0; i < 3; ++i) r[i] += coef[((size_t)i * kp + j) * p + m] * q[i];
            }
            const double* q = comps + (long long)(kp + m) * stride + 3 * e;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                r[i] -= q[i];
                en += r[i] * r[i];
                amax = fmax(amax, fabs(r[i]));
            }
        }
        energy[e] = en;
    }
    amax = wave_max(amax);
    if ((threadIdx.x & 63) == 0) sh_a[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) pabs[blockIdx.x] = fmax(fmax(sh_a[0], sh_a[1]), fmax(sh_a[2], sh_a[3]));
}

extern "C" int asb_deim_block_residual(asb_ctx* ctx, int64_t k, int p, const double* coef, double* maxabs_out) {
    if (!ctx || !ctx->comps || k < 0 || p < 1 || (k + 1) * p > ctx->K || !maxabs_out) return ASB_ERR_ARG;
    if (k > 0 && !coef) return ASB_ERR_ARG;
    const int kp = (int)(k * p);
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->pod_coef, (size_t)3 * ctx->K * p + 8))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->energy, (size_t)ctx->n_loc))) return rc;
    if (k > 0) ASB_HIP(ctx, hipMemcpyAsync(ctx->pod_coef, coef, (size_t)3 * kp * p * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    long long want = (ctx->n_loc + 255) / 256;
    const int grid = (int)(want < 1024 ? want : 1024);
    hipLaunchKernelGGL(k_deim_block_residual, dim3(grid), dim3(256), 0, ctx->stream, ctx->comps, (long long)ctx->n_loc, kp, p,
                       ctx->pod_coef, ctx->energy, ctx->pmax);
    ASB_CHECK_LAUNCH(ctx);
    std::vector<double> h(grid);
    ASB_HIP(ctx, hipMemcpyAsync(h.data(), ctx->pmax, grid * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double am = 0.0;
    for (int b = 0; b < grid; ++b) am = h[b] > am ? h[b] : am;
    *maxabs_out = am;
    return ASB_OK;
}

// shard arg-max of the DEIM residual of basis vector k given the interpolation coefficients
// coef (host, 3 x k; NULL for k = 0); returns the global row index and its residual energy.
extern "C" int asb_deim_step(asb_ctx* ctx, int64_t k, const double* coef, int64_t* idx_out, double* val_out) {
    if (!ctx || !ctx->comps || k < 0 || k >= ctx->K || !idx_out) return ASB_ERR_ARG;
    if (k > 0 && !coef) return ASB_ERR_ARG;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->pod_coef, (size_t)3 * ctx->K))) return rc;
    if (k > 0) ASB_HIP(ctx, hipMemcpyAsync(ctx->pod_coef, coef, (size_t)3 * k * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    long long want = (ctx->n_loc + 255) / 256;
    const int grid = (int)(want < 1024 ? want : 1024);
    hipLaunchKernelGGL(k_deim_residual, dim3(grid), dim3(256), (size_t)(3 * k + 1) * sizeof(double), ctx->stream, ctx->comps,
                       (long long)ctx->n_loc, (int)k, ctx->pod_coef, (long long)ctx->v0, ctx->pmax, ctx->pidx);
    ASB_CHECK_LAUNCH(ctx);
    std::vector<double> hm(grid);
    std::vector<long long> hi(grid);
    ASB_HIP(ctx, hipMemcpyAsync(hm.data(), ctx->pmax, grid * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipMemcpyAsync(hi.data(), ctx->pidx, grid * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double be = -1.0;
    long long bi = 0x7fffffffffffffffLL;
    for (int b = 0; b < grid; ++b)
        if (hm[b] > be || (hm[b] == be && hi[b] < bi)) { be = hm[b]; bi = hi[b]; }
    *idx_out = bi;
    if (val_out) *val_out = be;
    return ASB_OK;
}

// row_out (K, 3): V[gidx, :, :] = comps[:, 3*(gidx - v0) + i]  (only on the rank that owns gidx; others get rc 1)
extern "C" int asb_deim_row(asb_ctx* ctx, int64_t gidx, double* row_out) {
    if (!ctx || !ctx->comps || !row_out) return ASB_ERR_ARG;
    if (gidx < ctx->v0 || gidx >= ctx->v0 + ctx->n_loc) return 1;
    const long long e = gidx - ctx->v0;
    ASB_HIP(ctx, hipMemcpy2DAsync(row_out, 3 * sizeof(double), ctx->comps + 3 * e, (size_t)3 * ctx->n_loc * sizeof(double),
                                  3 * sizeof(double), (size_t)ctx->K, hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ASB_OK;
}

// --------------------------------------------------------------------------------------
// The weighted differential operator S^T (sparse, position-space vertices x constraint rows) of the constraint path:
// 'pca_blocks_with_St' picks the position-space vertex where S^T R is largest (constraintsComponents.py:180), the
// position-space variant of the geometric interpolation measures the residual of a basis block through S^T (:652, :672).
// Both are "squared row norms of S^T M" for a dense row-major M on the device: one wave per row of S^T, the lanes over
// M's columns, a chunk of 64 x U columns at a time (the CSR row is re-read per chunk: it is short and cached).
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_st_row_energy(const long long* __restrict__ indptr, const long long* __restrict__ indices,
                                                       const double* __restrict__ data, long long n_rows, const double* __restrict__ M,
                                                       long long ldm, long long ncols, double* __restrict__ E,
                                                       double* __restrict__ Amax) {          // Amax (optional): largest |entry| per row
    constexpr int U = 4;
    const int lane = threadIdx.x & 63;
    for (long long v = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); v < n_rows; v += (long long)gridDim.x * 4) {
        const long long a = indptr[v], b = indptr[v + 1];
        double e = 0.0, am = 0.0;
        for (long long c0 = 0; c0 < ncols; c0 += 64 * U) {
            double acc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = 0.0;
            for (long long q = a; q < b; ++q) {
                const double w = data[q];
                const double* row = M + indices[q] * ldm;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const long long c = c0 + lane + 64 * u;
                    if (c < ncols) acc[u] += w * row[c];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                e += acc[u] * acc[u];
                am = fmax(am, fabs(acc[u]));
            }
        }
        e = wave_sum(e);
        if (lane == 0) E[v] = e;
        if (Amax) {
            am = wave_max(am);
            if (lane == 0) Amax[v] = am;
        }
    }
}

extern "C" int asb_st_upload(asb_ctx* ctx, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr, const int64_t* indices,
                             const double* data) {
    if (!ctx || n_rows < 1 || n_cols < 1 || nnz < 0 || !indptr || (nnz > 0 && (!indices || !data))) return ASB_ERR_ARG;
    for (int64_t q = 0; q < nnz; ++q)
        if (indices[q] < 0 || indices[q] >= n_cols) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_st_upload: column index out of range");
    if (indptr[0] != 0 || indptr[n_rows] != nnz) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_st_upload: bad row pointers");
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->st_indptr, (size_t)n_rows + 1))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->st_indices, (size_t)(nnz > 0 ? nnz : 1)))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->st_data, (size_t)(nnz > 0 ? nnz : 1)))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->st_energy, (size_t)n_rows))) return rc;
    static_assert(sizeof(long long) == sizeof(int64_t), "index width");
    ASB_HIP(ctx, hipMemcpyAsync(ctx->st_indptr, indptr, (size_t)(n_rows + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    if (nnz > 0) {
        ASB_HIP(ctx, hipMemcpyAsync(ctx->st_indices, indices, (size_t)nnz * 8, hipMemcpyHostToDevice, ctx->stream));
        ASB_HIP(ctx, hipMemcpyAsync(ctx->st_data, data, (size_t)nnz * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->st_rows = n_rows;
    ctx->st_cols = n_cols;
    ctx->st_nnz = nnz;
    return ASB_OK;
}

// first arg-max of E[0 .. n) (k_block_argmax with blocks of one row), read back
static int st_argmax(asb_ctx* ctx, const double* E, long long n, int64_t* idx_out, double* val_out) {
    const int grid = (int)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256);
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->bam_val, (size_t)256))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->bam_idx, (size_t)256))) return rc;
    hipLaunchKernelGGL(k_block_argmax, dim3(grid), dim3(256), 0, ctx->stream, E, n, 1, (long long)0, ctx->bam_val, ctx->bam_idx);
    ASB_CHECK_LAUNCH(ctx);
    double hv[256];
    long long hi[256];
    ASB_HIP(ctx, hipMemcpyAsync(hv, ctx->bam_val, grid * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipMemcpyAsync(hi, ctx->bam_idx, grid * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double be = -1.0;
    long long bi = 0x7fffffffffffffffLL;
    for (int b = 0; b < grid; ++b)
        if (hv[b] > be || (hv[b] == be && hi[b] < bi)) { be = hv[b]; bi = hi[b]; }
    *idx_out = bi;
    if (val_out) *val_out = be;
    return ASB_OK;
}
static int st_row_energies(asb_ctx* ctx, const double* M, long long ldm, long long ncols, double* amax = nullptr) {
    const long long want = (ctx->st_rows + 3) / 4;
    const int grid = (int)(want < 8LL * ctx->n_cu ? want : 8LL * ctx->n_cu);
    hipLaunchKernelGGL(k_st_row_energy, dim3(grid), dim3(256), 0, ctx->stream, ctx->st_indptr, ctx->st_indices, ctx->st_data,
                       (long long)ctx->st_rows, M, ldm, ncols, ctx->st_energy, amax);
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

extern "C" int asb_st_residual_argmax(asb_ctx* ctx, int64_t* v_out, double* val_out) {
    if (!ctx || !ctx->R || !v_out) return ASB_ERR_ARG;
    if (ctx->mode != ASB_DEFLATE_RESIDUAL) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_st_residual_argmax needs the residual mode");
    if (!ctx->st_indptr || ctx->st_cols != ctx->n_loc || ctx->n_loc != ctx->N_glob)
        ASB_FAIL(ctx, ASB_ERR_ARG, "asb_st_residual_argmax: S^T (%lld columns) does not match the %lld constraint rows of this (single) shard",
                 (long long)ctx->st_cols, (long long)ctx->n_loc);
    // the residual of constraint row j is 3 Fp contiguous doubles (x, y, z sub-rows; the padding is zero)
    int rc = st_row_energies(ctx, ctx->R, 3 * ctx->Fp, 3 * ctx->Fp);
    if (rc) return rc;
    return st_argmax(ctx, ctx->st_energy, ctx->st_rows, v_out, val_out);
}

// |R|^2 = sum of the per-row energies the last pass left (asb_deflate_begin / asb_deflate_apply)
__global__ __launch_bounds__(1024) void k_sum_all(const double* __restrict__ x, long long n, double* __restrict__ out) {
    __shared__ double sh[16];
    double s = 0.0;
    for (long long i = threadIdx.x; i < n; i += 1024) s += x[i];
    double v[1] = {s};
    block_sum<1>(v, sh);
    if (threadIdx.x == 0) *out = v[0];
}
extern "C" int asb_deflate_residual_norm2(asb_ctx* ctx, double* out) {
    if (!ctx || !ctx->R || !ctx->energy || !out) return ASB_ERR_ARG;
    if (ctx->mode != ASB_DEFLATE_RESIDUAL) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_deflate_residual_norm2 needs the residual mode");
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->bam_val, (size_t)256))) return rc;
    hipLaunchKernelGGL(k_sum_all, dim3(1), dim3(1024), 0, ctx->stream, ctx->energy, (long long)ctx->n_loc, ctx->bam_val);
    ASB_CHECK_LAUNCH(ctx);
    ASB_HIP(ctx, hipMemcpyAsync(out, ctx->bam_val, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ASB_OK;
}

// residual of basis block k (k_deim_block_residual's arithmetic) written out: resid[e][m][i], e < n_rows, m < p, i < 3
__global__ __launch_bounds__(256) void k_deim_block_residual_out(const double* __restrict__ comps, long long n_rows, int kp, int p,
                                                                 const double* __restrict__ coef, double* __restrict__ resid,
                                                                 double* __restrict__ pabs) {
    __shared__ double sh_a[4];
    const long long stride = 3 * n_rows;
    double amax = 0.0;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n_rows; e += (long long)gridDim.x * 256) {
        for (int m = 0; m < p; ++m) {
            double r[3] = {0.0, 0.0, 0.0};
            for (int j = 0; j < kp; ++j) {
                const double* q = comps + (long long)j * stride + 3 * e;
#pragma unroll
                for (int i = 0; i < 3; ++i) r[i] += coef[((size_t)i * kp + j) * p + m] * q[i];
            }
            const double* q = comps + (long long)(kp + m) * stride + 3 * e;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                r[i] -= q[i];
                resid[(e * p + m) * 3 + i] = r[i];
                amax = fmax(amax, fabs(r[i]));
            }
        }
    }
    amax = wave_max(amax);
    if ((threadIdx.x & 63) == 0) sh_a[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) pabs[blockIdx.x] = fmax(fmax(sh_a[0], sh_a[1]), fmax(sh_a[2], sh_a[3]));
}

extern "C" int asb_deim_block_residual_st(asb_ctx* ctx, int64_t k, int p, const double* coef, double* maxabs_out, int64_t* v_out,
                                          double* val_out) {
    if (!ctx || !ctx->comps || k < 0 || p < 1 || (k + 1) * p > ctx->K || !maxabs_out || !v_out) return ASB_ERR_ARG;
    if (k > 0 && !coef) return ASB_ERR_ARG;
    // rows of the basis are constraint ROWS (e p of them); the reference reshapes the (e p, p, 3) block to (e p, 3 p) and S^T has
    // e p columns (:652 `self.St @ vk.reshape(vk.shape[0], -1)`)
    if (!ctx->st_indptr || ctx->st_cols != ctx->n_loc || ctx->n_loc != ctx->N_glob)
        ASB_FAIL(ctx, ASB_ERR_ARG, "asb_deim_block_residual_st: S^T (%lld columns) does not match the %lld rows of this (single) shard",
                 (long long)ctx->st_cols, (long long)ctx->n_loc);
    const int kp = (int)(k * p);
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->pod_coef, (size_t)3 * ctx->K * p + 8))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->st_resid, (size_t)ctx->n_loc * 3 * p))) return rc;
    if (k > 0) ASB_HIP(ctx, hipMemcpyAsync(ctx->pod_coef, coef, (size_t)3 * kp * p * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    long long want = (ctx->n_loc + 255) / 256;
    const int grid = (int)(want < 1024 ? want : 1024);
    hipLaunchKernelGGL(k_deim_block_residual_out, dim3(grid), dim3(256), 0, ctx->stream, ctx->comps, (long long)ctx->n_loc, kp, p,
                       ctx->pod_coef, ctx->st_resid, ctx->pmax);
    ASB_CHECK_LAUNCH(ctx);
    // (the reference's stop test np.allclose(r, 0) at :677 looks at r = S^T (c - v_k) ENTRY by entry, atol 1e-8: the largest
    // |entry| of S^T r comes back, not the largest row norm)
    if ((rc = asb_alloc(ctx, &ctx->st_amax, (size_t)ctx->st_rows))) return rc;
    if ((rc = st_row_energies(ctx, ctx->st_resid, 3 * p, 3 * p, ctx->st_amax))) return rc;
    int64_t am_row = 0;
    double am = 0.0;
    if ((rc = st_argmax(ctx, ctx->st_amax, ctx->st_rows, &am_row, &am))) return rc;
    if ((rc = st_argmax(ctx, ctx->st_energy, ctx->st_rows, v_out, val_out))) return rc;      // (synchronises)
    *maxabs_out = am;
    return ASB_OK;
}
