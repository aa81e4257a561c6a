// Heat-method geodesics on the device -- GeodesicDistanceComputation.__call__,
// utils/support.py:173-208 of the reference, for up to 64 source vertices at a time.
//
// The reference factorises (A - tL) and L once with SuperLU and back-substitutes per source.  Here the two SPD
// systems are solved by Jacobi-preconditioned conjugate gradients for 64 right-hand sides together: vectors are
// (N x 64) node-major, so one wave owns a matrix row and its 64 lanes are the 64 systems -- every neighbour fetch
// is one contiguous 512-byte line.  Gradient, normalisation and divergence are two more SpMMs with the operators
// assembled on the host (geodesic.py).  L is singular (constants): CG stays in the range for the consistent
// right-hand side and the constant is fixed afterwards by phi -= min(phi), exactly as the reference does (:206).
#include "asb_kernels.h"

#include <vector>

#define GB 64     // right-hand sides per batch = lanes of a wave

// One factorised block-tridiagonal SPD matrix (the "slab" mode, round 4): A = L D L^T with dense blocks --
// Dinv[k] = D_k^-1 (sp_k x sp_k), E[k] = A_{k,k-1} D_{k-1}^-1 (sp_k x sp_{k-1}) and its transpose Et[k]
struct asb_bt {
    std::vector<double*> Dinv, E, Et;
    void release() {
        for (auto* p : Dinv) if (p) (void)hipFree(p);
        for (auto* p : E) if (p) (void)hipFree(p);
        for (auto* p : Et) if (p) (void)hipFree(p);
        Dinv.clear(); E.clear(); Et.clear();
    }
};

struct asb_csr {
    int rows = 0, cols = 0;
    long long nnz = 0;
    int* rowptr = nullptr;
    int* colidx = nullptr;
    double* vals = nullptr;
};

struct asb_geo {
    int n = 0, m3 = 0;                 // vertices, 3 * triangles
    asb_csr heat, lap, grad, div;      // A - tL, -L, G (3M x N), D (N x 3M)
    double *dheat = nullptr, *dlap = nullptr;      // diagonals (Jacobi)
    double *x = nullptr, *r = nullptr, *p = nullptr, *ap = nullptr, *z = nullptr, *g = nullptr, *b = nullptr;
    double *part = nullptr, *sc = nullptr;         // (nblk, 64) partial sums; scalars rz[64], alpha[64], ...
    int nblk = 0;
    // dense mode (asb_geodesic_dense_setup): explicit inverses, np = roundup(n, 16), ld np
    bool dense = false;
    int np = 0;
    double *Hinv = nullptr, *Pinv = nullptr;
    double* sv_part = nullptr;      // k_symv_tiles partial sums
    // two-level preconditioner (asb_geodesic_coarse_setup): piecewise-constant aggregates, coarse inverses, coarse vectors
    bool coarse = false;
    double heat_omega = 1.0;
    int nc = 0, ncp = 0;
    int *agg = nullptr, *agg_ptr = nullptr, *agg_mem = nullptr;
    double *AcH = nullptr, *AcP = nullptr, *rc = nullptr, *zc = nullptr;
    // distance fields kept for SPLOCS (asb_geodesic_cache_add): slot q lives in slab[q / 64] at row q % 64
    double* slab[ASB_GEO_CACHE_SLABS] = {};
    long long cached = 0;
    // slab mode (asb_geodesic_bt_setup): both systems factorised DIRECTLY as block-tridiagonal matrices over breadth-first
    // slabs of the mesh; bt_off / bt_sz: padded offset / size of every slab, pos: vertex -> padded position
    bool bt = false;
    std::vector<int> bt_off, bt_sz;
    int bt_npad = 0;
    int* bt_pos = nullptr;
    asb_bt Hbt, Pbt;
    double *bt_z = nullptr, *bt_w = nullptr;       // (npad x 64) right-hand sides / solutions in slab order
    ~asb_geo() { Hbt.release(); Pbt.release(); }
};

// Y = A X  (+ optional per-column partial sums of X .* Y for CG's p^T A p)
__global__ __launch_bounds__(256) void k_spmm64(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                                const double* __restrict__ vals, int rows, const double* __restrict__ X,
                                                double* __restrict__ Y, double* __restrict__ part_xy) {
    __shared__ double sh[4][GB];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double acc_dot = 0.0;
    for (int r = blockIdx.x * 4 + wid; r < rows; r += gridDim.x * 4) {
        double y = 0.0;
        for (int j = rowptr[r]; j < rowptr[r + 1]; ++j) y += vals[j] * X[(long long)colidx[j] * GB + lane];
        Y[(long long)r * GB + lane] = y;
        if (part_xy) acc_dot += y * X[(long long)r * GB + lane];
    }
    if (part_xy) {
        sh[wid][lane] = acc_dot;
        __syncthreads();
        if (wid == 0) part_xy[(long long)blockIdx.x * GB + lane] = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
    }
}

// alpha = rz / sum(part pAp);  x += alpha p;  r -= alpha Ap;  z = r / diag;  partial r.z (and r.r in part2)
__global__ __launch_bounds__(256) void k_cg_update(int rows, int nblk_in, const double* __restrict__ part_pap,
                                                   const double* __restrict__ rz, const double* __restrict__ diag,
                                                   double* __restrict__ x, double* __restrict__ r, const double* __restrict__ p,
                                                   const double* __restrict__ ap, double* __restrict__ z,
                                                   double* __restrict__ part_rz, double* __restrict__ part_rr, double tol2) {
    __shared__ double sh[2][4][GB];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double pap = 0.0;
    for (int q = 0; q < nblk_in; ++q) pap += part_pap[(long long)q * GB + lane];
    // a column that has reached the tolerance (or was zero from the start) is frozen: it must not keep dividing
    // by a vanishing p^T A p while the other columns still iterate
    // (the test is on |r|^2 against its start value, sc[2 GB..] / sc[4 GB..], not on r.z: with the coarse term in the
    //  preconditioner r.z of a smooth start residual is orders of magnitude above what it is later for the same |r|)
    const bool active = rz[2 * GB + lane] > tol2 * rz[4 * GB + lane] && rz[lane] > 0.0 && pap > 0.0;
    const double a = active ? rz[lane] / pap : 0.0;
    double s_rz = 0.0, s_rr = 0.0;
    for (int i = blockIdx.x * 4 + wid; i < rows; i += gridDim.x * 4) {
        const long long e = (long long)i * GB + lane;
        x[e] += a * p[e];
        const double rn = r[e] - a * ap[e];
        r[e] = rn;
        const double zn = rn / diag[i];
        z[e] = zn;
        s_rz += rn * zn;
        s_rr += rn * rn;
    }
    sh[0][wid][lane] = s_rz; sh[1][wid][lane] = s_rr;
    __syncthreads();
    if (wid == 0) {
        part_rz[(long long)blockIdx.x * GB + lane] = (sh[0][0][lane] + sh[0][1][lane]) + (sh[0][2][lane] + sh[0][3][lane]);
        part_rr[(long long)blockIdx.x * GB + lane] = (sh[1][0][lane] + sh[1][1][lane]) + (sh[1][2][lane] + sh[1][3][lane]);
    }
}

// beta = rz_new / rz;  p = z + beta p;  block 0 publishes rz_new and rr
__global__ __launch_bounds__(256) void k_cg_direction(int rows, int nblk_in, const double* __restrict__ part_rz,
                                                      const double* __restrict__ part_rr, double* __restrict__ rz,
                                                      double* __restrict__ rr, const double* __restrict__ z,
                                                      double* __restrict__ p, int first, double tol2) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double rzn = 0.0, rrn = 0.0;
    for (int q = 0; q < nblk_in; ++q) { rzn += part_rz[(long long)q * GB + lane]; rrn += part_rr[(long long)q * GB + lane]; }
    const double old = rz[lane];
    const double beta = (first || old == 0.0 || !(rrn > tol2 * rz[4 * GB + lane])) ? 0.0 : rzn / old;      // (rrn: every block sums the same partials)
    for (int i = blockIdx.x * 4 + wid; i < rows; i += gridDim.x * 4) {
        const long long e = (long long)i * GB + lane;
        p[e] = z[e] + beta * p[e];
    }
    __threadfence();
    if (blockIdx.x == gridDim.x - 1 && wid == 0) {       // staged: see k_cg_commit
        rz[GB + lane] = rzn;
        rr[lane] = rrn;
        if (first) { rz[3 * GB + lane] = rzn; rz[4 * GB + lane] = rrn; }      // r0.z0 and |r0|^2 of the column: the freeze reference
    }
}

__global__ void k_cg_commit(double* __restrict__ rz) { rz[threadIdx.x] = rz[GB + threadIdx.x]; }

// ---- two-level additive preconditioner  z = D^-1 r + P Ac^-1 P^T r  (P: piecewise-constant prolongation from aggregates of
// ~50 mesh vertices, Ac = P^T A P inverted densely once).  Jacobi alone needs O(1/h) iterations on the Poisson system
// (thousands on a 14 000-vertex scan); the coarse term carries the smooth error components, and the count becomes a
// property of the aggregate size, not of the mesh size.
// rc[a] = sum of r over the members of aggregate a (one wave per aggregate, members in index order: deterministic)
__global__ __launch_bounds__(256) void k_restrict64(const int* __restrict__ agg_ptr, const int* __restrict__ agg_mem, int nc,
                                                    const double* __restrict__ r, double* __restrict__ rc) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int a = blockIdx.x * 4 + wid; a < nc; a += gridDim.x * 4) {
        double s = 0.0;
        for (int j = agg_ptr[a]; j < agg_ptr[a + 1]; ++j) s += r[(long long)agg_mem[j] * GB + lane];
        rc[(long long)a * GB + lane] = s;
    }
}
// z += zc[agg]; partial r.z with the complete z (replaces the Jacobi-only partials of k_cg_update / k_cg_start)
__global__ __launch_bounds__(256) void k_prolong_rz(int rows, const int* __restrict__ agg, const double* __restrict__ zc,
                                                    const double* __restrict__ r, double* __restrict__ z,
                                                    double* __restrict__ part_rz) {
    __shared__ double sh[4][GB];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double s = 0.0;
    for (int i = blockIdx.x * 4 + wid; i < rows; i += gridDim.x * 4) {
        const long long e = (long long)i * GB + lane;
        const double zn = z[e] + zc[(long long)agg[i] * GB + lane];
        z[e] = zn;
        s += r[e] * zn;
    }
    sh[wid][lane] = s;
    __syncthreads();
    if (wid == 0) part_rz[(long long)blockIdx.x * GB + lane] = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
}

// z = r / diag, partial r.z / r.r  (start of CG with x = 0, r = b)
__global__ __launch_bounds__(256) void k_cg_start(int rows, const double* __restrict__ diag, const double* __restrict__ r,
                                                  double* __restrict__ z, double* __restrict__ part_rz,
                                                  double* __restrict__ part_rr) {
    __shared__ double sh[2][4][GB];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double s_rz = 0.0, s_rr = 0.0;
    for (int i = blockIdx.x * 4 + wid; i < rows; i += gridDim.x * 4) {
        const long long e = (long long)i * GB + lane;
        const double rn = r[e], zn = rn / diag[i];
        z[e] = zn;
        s_rz += rn * zn; s_rr += rn * rn;
    }
    sh[0][wid][lane] = s_rz; sh[1][wid][lane] = s_rr;
    __syncthreads();
    if (wid == 0) {
        part_rz[(long long)blockIdx.x * GB + lane] = (sh[0][0][lane] + sh[0][1][lane]) + (sh[0][2][lane] + sh[0][3][lane]);
        part_rr[(long long)blockIdx.x * GB + lane] = (sh[1][0][lane] + sh[1][1][lane]) + (sh[1][2][lane] + sh[1][3][lane]);
    }
}

// X_f = -grad / |grad| per triangle (rows 3t..3t+2) and column
__global__ __launch_bounds__(256) void k_normalise_field(double* __restrict__ g, int ntri) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int t = blockIdx.x * 4 + wid; t < ntri; t += gridDim.x * 4) {
        double* q = g + (long long)3 * t * GB + lane;
        const double a = q[0], b = q[GB], c = q[2 * GB];
        const double len = sqrt(a * a + b * b + c * c);
        q[0] = -a / len; q[GB] = -b / len; q[2 * GB] = -c / len;
    }
}

__global__ __launch_bounds__(256) void k_scale_vec(double* __restrict__ x, long long n, double a) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) x[i] *= a;
}

// ---- heat step of the sparse mode: (A - tL) u = delta by plain Jacobi sweeps from u = 0.  u falls off like exp(-d / sqrt(t))
// -- 14 orders of magnitude across a 14 000-vertex mesh -- and the heat method needs the DIRECTION of grad u everywhere, i.e.
// u to relative accuracy in every component.  A Krylov method converges in a norm (absolute accuracy: the far field is lost,
// measured 0.7 % error in the distances); the Jacobi sweep of this M-matrix only ever adds non-negative terms, so every
// component is built up without cancellation and converges to rounding level relative to ITSELF -- what SuperLU's
// factorisation delivers in the reference.  Rate 1 - area / (area + t sum w) ~ 0.976 per sweep: ~2000 sweeps, each one SpMM.
// xout = (b - offdiag(A) xin) / diag;  part[block][lane] = max over the block's rows of |xout - xin| / |xout| (1 where 0)
__global__ __launch_bounds__(256) void k_heat_jacobi(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                                     const double* __restrict__ vals, const double* __restrict__ diag, int rows,
                                                     const double* __restrict__ b, const double* __restrict__ xin,
                                                     double* __restrict__ xout, double* __restrict__ part, int nsrc, double omega) {
    __shared__ double sh[4][GB];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double worst = 0.0;
    for (int r = blockIdx.x * 4 + wid; r < rows; r += gridDim.x * 4) {
        double acc = b[(long long)r * GB + lane];
        for (int j = rowptr[r]; j < rowptr[r + 1]; ++j) {
            const int c = colidx[j];
            if (c != r) acc -= vals[j] * xin[(long long)c * GB + lane];
        }
        const double xo = xin[(long long)r * GB + lane];
        const double xn = (1.0 - omega) * xo + omega * (acc / diag[r]);     // omega < 1 only where obtuse triangles break diagonal dominance
        xout[(long long)r * GB + lane] = xn;
        if (part && lane < nsrc) {
            const double rel = xn == 0.0 ? 1.0 : fabs(xn - xo) / fabs(xn);
            worst = fmax(worst, rel);
        }
    }
    if (part) {
        sh[wid][lane] = worst;
        __syncthreads();
        if (wid == 0) part[(long long)blockIdx.x * GB + lane] = fmax(fmax(sh[0][lane], sh[1][lane]), fmax(sh[2][lane], sh[3][lane]));
    }
}
__global__ __launch_bounds__(64) void k_colmax(const double* __restrict__ part, int nblk, double* __restrict__ out) {
    double m = 0.0;
    for (int q = 0; q < nblk; ++q) m = fmax(m, part[(long long)q * GB + threadIdx.x]);
    out[threadIdx.x] = m;
}

// b[:, c] -= mean(b[:, c]): the Poisson system is singular (constants); a right-hand side with a component along the null
// vector has no solution, its residual cannot fall below that component, and a preconditioner with a coarse level would
// amplify it into the search directions.  (The divergence of the normalised gradient field sums to zero only up to rounding.)
__global__ __launch_bounds__(256) void k_remove_mean(double* __restrict__ b, int n) {
    __shared__ double sh[4];
    const int c = blockIdx.x;
    double v[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += 256) v[0] += b[(long long)i * GB + c];
    block_sum<1>(v, sh);
    const double m = v[0] / (double)n;
    for (int i = threadIdx.x; i < n; i += 256) b[(long long)i * GB + c] -= m;
}

// b[idx[c]][c] = 1 for the batch's sources (others 0)
__global__ void k_set_sources(double* __restrict__ b, const long long* __restrict__ idx, int nsrc) {
    const int c = threadIdx.x;
    if (c < nsrc) b[idx[c] * GB + c] = 1.0;
}

// phi[:, c] -= min(phi[:, c]); out (nsrc, n) source-major
__global__ __launch_bounds__(256) void k_shift_min_out(const double* __restrict__ phi, int n, int nsrc,
                                                       double* __restrict__ out) {
    __shared__ double sh[256];
    const int c = blockIdx.x;
    double m = 1.0e300;
    for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmin(m, phi[(long long)i * GB + c]);
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] = fmin(sh[threadIdx.x], sh[threadIdx.x + o]);
        __syncthreads();
    }
    m = sh[0];
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[(long long)c * n + i] = phi[(long long)i * GB + c] - m;
    (void)nsrc;
}

// ------------------------------------------------------------------------------------- host
static int upload_csr(asb_ctx* ctx, asb_csr& A, int rows, int cols, const int* rowptr, const int* colidx, const double* vals) {
    A.rows = rows; A.cols = cols; A.nnz = rowptr[rows];
    int rc;
    if ((rc = asb_alloc(ctx, &A.rowptr, (size_t)rows + 1))) return rc;
    if ((rc = asb_alloc(ctx, &A.colidx, (size_t)A.nnz))) return rc;
    if ((rc = asb_alloc(ctx, &A.vals, (size_t)A.nnz))) return rc;
    ASB_HIP(ctx, hipMemcpyAsync(A.rowptr, rowptr, ((size_t)rows + 1) * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    ASB_HIP(ctx, hipMemcpyAsync(A.colidx, colidx, (size_t)A.nnz * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    ASB_HIP(ctx, hipMemcpyAsync(A.vals, vals, (size_t)A.nnz * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    return ASB_OK;
}

// operators in CSR (int32 indices): heat = A - tL (n x n), lap = -L (n x n), grad (3M x n), div (n x 3M); diagonals of the two SPD ones
extern "C" int asb_geodesic_setup(asb_ctx* ctx, int n, int m3, const int* heat_rp, const int* heat_ci, const double* heat_v,
                                  const int* lap_rp, const int* lap_ci, const double* lap_v, const int* grad_rp,
                                  const int* grad_ci, const double* grad_v, const int* div_rp, const int* div_ci,
                                  const double* div_v, const double* heat_diag, const double* lap_diag) {
    if (!ctx || n < 1 || m3 < 3) return ASB_ERR_ARG;
    ASB_HIP(ctx, hipSetDevice(ctx->dev));
    if (!ctx->geo) ctx->geo = new asb_geo();
    asb_geo* G = ctx->geo;
    G->n = n; G->m3 = m3;
    G->dense = false;                  // a new mesh: explicit inverses and cached fields of the old one are void
    G->coarse = false;
    G->bt = false;
    G->cached = 0;
    int rc;
    if ((rc = upload_csr(ctx, G->heat, n, n, heat_rp, heat_ci, heat_v))) return rc;
    if ((rc = upload_csr(ctx, G->lap, n, n, lap_rp, lap_ci, lap_v))) return rc;
    if ((rc = upload_csr(ctx, G->grad, m3, n, grad_rp, grad_ci, grad_v))) return rc;
    if ((rc = upload_csr(ctx, G->div, n, m3, div_rp, div_ci, div_v))) return rc;
    if ((rc = asb_alloc(ctx, &G->dheat, (size_t)n))) return rc;
    if ((rc = asb_alloc(ctx, &G->dlap, (size_t)n))) return rc;
    ASB_HIP(ctx, hipMemcpyAsync(G->dheat, heat_diag, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    ASB_HIP(ctx, hipMemcpyAsync(G->dlap, lap_diag, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    const size_t nv = (size_t)n * GB;
    double** vecs[] = {&G->x, &G->r, &G->p, &G->ap, &G->z, &G->b};
    for (auto v : vecs)
        if ((rc = asb_alloc(ctx, v, nv))) return rc;
    if ((rc = asb_alloc(ctx, &G->g, (size_t)m3 * GB))) return rc;
    G->nblk = (n + 3) / 4 < 1024 ? (n + 3) / 4 : 1024;
    if ((rc = asb_alloc(ctx, &G->part, (size_t)3 * G->nblk * GB))) return rc;
    if ((rc = asb_alloc(ctx, &G->sc, (size_t)5 * GB))) return rc;
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ASB_OK;
}

// x <- (A - tL)^-1 b for the 64 columns by Jacobi sweeps (see k_heat_jacobi); uses G->p as the second buffer
static int heat_jacobi64(asb_ctx* ctx, asb_geo* G, const double* b, double* x, int nsrc, int* iters_out) {
    const int n = G->n, nb = G->nblk;
    const asb_csr& A = G->heat;
    ASB_HIP(ctx, hipMemsetAsync(x, 0, (size_t)n * GB * sizeof(double), ctx->stream));
    double* cur = x;
    double* nxt = G->p;
    int it = 0;
    double prev_worst = 2.0;
    int flat = 0;
    double h[GB];
    double worst = 1.0;
    for (; it < 24000;) {
        for (int q = 0; q < 64; ++q, ++it) {
            const bool check = q == 63;
            hipLaunchKernelGGL(k_heat_jacobi, dim3(nb), dim3(256), 0, ctx->stream, A.rowptr, A.colidx, A.vals, G->dheat, n, b, cur, nxt,
                               check ? G->part : (double*)nullptr, nsrc, G->heat_omega);
            double* t = cur; cur = nxt; nxt = t;
        }
        hipLaunchKernelGGL(k_colmax, dim3(1), dim3(GB), 0, ctx->stream, G->part, nb, G->sc);
        ASB_CHECK_LAUNCH(ctx);
        ASB_HIP(ctx, hipMemcpyAsync(h, G->sc, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        worst = 0.0;
        for (int c = 0; c < nsrc; ++c) worst = h[c] > worst ? h[c] : worst;
        if (worst <= 2e-15) break;
        // rounding floor / unreachable vertices (another connected component): stop when the measure no longer moves
        if (worst < 1e-9 && worst > 0.97 * prev_worst) { if (++flat >= 6) break; } else flat = 0;
        prev_worst = worst;
    }
    if (cur != x) ASB_HIP(ctx, hipMemcpyAsync(x, cur, (size_t)n * GB * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    if (iters_out) *iters_out = it;
    // the sweep's rate is 1 - min_i area_i / (area_i + t sum_j w_ij): fine on quasi-uniform meshes (scans), hopeless where
    // element sizes differ by orders of magnitude -- say so instead of returning an unconverged heat field
    if (!(worst <= 1e-11))
        ASB_FAIL(ctx, ASB_ERR_NUMERIC, "device geodesics (sparse mode): the heat step's Jacobi sweeps reached a relative change of %.1e "
                 "after %d sweeps (badly graded mesh); use the host SuperLU backend (ASB_GEODESIC=host)", worst, it);
    return ASB_OK;
}

// Coarse level of the two-level preconditioner (after asb_geodesic_setup): agg (n) = aggregate of every vertex, its CSR
// form (agg_ptr nc + 1, agg_mem n), and the two coarse operators P^T (A - tL) P and P^T (-L) P + gauge (host, nc x nc, SPD),
// which are inverted here on the device (blocked Gauss-Jordan, asb_dense.hip).
extern "C" int asb_geodesic_coarse_setup(asb_ctx* ctx, int nc, const int* agg, const int* agg_ptr, const int* agg_mem,
                                         const double* heat_c, const double* lap_c, double heat_omega) {
    if (!ctx || !ctx->geo || nc < 1 || !agg || !agg_ptr || !agg_mem || !heat_c || !lap_c) return ASB_ERR_ARG;
    if (!(heat_omega > 0.0 && heat_omega <= 1.0)) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_geodesic_coarse_setup: damping %.3g not in (0, 1]", heat_omega);
    asb_geo* G = ctx->geo;
    G->heat_omega = heat_omega;
    const int n = G->n, ncp = (nc + 15) / 16 * 16;
    if (nc > 46000) ASB_FAIL(ctx, ASB_ERR_LIMIT, "geodesics: %d aggregates are too many for a dense coarse level", nc);
    int rc;
    if ((rc = asb_alloc(ctx, &G->agg, (size_t)n))) return rc;
    if ((rc = asb_alloc(ctx, &G->agg_ptr, (size_t)nc + 1))) return rc;
    if ((rc = asb_alloc(ctx, &G->agg_mem, (size_t)n))) return rc;
    if ((rc = asb_alloc(ctx, &G->AcH, (size_t)ncp * ncp))) return rc;
    if ((rc = asb_alloc(ctx, &G->AcP, (size_t)ncp * ncp))) return rc;
    if ((rc = asb_alloc(ctx, &G->rc, (size_t)ncp * GB))) return rc;
    if ((rc = asb_alloc(ctx, &G->zc, (size_t)ncp * GB))) return rc;
    if ((rc = asb_alloc(ctx, &G->z, (size_t)n * GB))) return rc;
    ASB_HIP(ctx, hipMemcpyAsync(G->agg, agg, (size_t)n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    ASB_HIP(ctx, hipMemcpyAsync(G->agg_ptr, agg_ptr, ((size_t)nc + 1) * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    ASB_HIP(ctx, hipMemcpyAsync(G->agg_mem, agg_mem, (size_t)n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    ASB_HIP(ctx, hipMemsetAsync(G->rc, 0, (size_t)ncp * GB * sizeof(double), ctx->stream));
    ASB_HIP(ctx, hipMemsetAsync(G->zc, 0, (size_t)ncp * GB * sizeof(double), ctx->stream));
    std::vector<double> pad((size_t)ncp * ncp);
    const double* src[2] = {heat_c, lap_c};
    double* dst[2] = {G->AcH, G->AcP};
    for (int q = 0; q < 2; ++q) {
        for (int i = 0; i < ncp; ++i)
            for (int j = 0; j < ncp; ++j)
                pad[(size_t)i * ncp + j] = (i < nc && j < nc) ? src[q][(size_t)i * nc + j] : (i == j ? 1.0 : 0.0);
        ASB_HIP(ctx, hipMemcpyAsync(dst[q], pad.data(), pad.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if ((rc = asb_dense_spd_inverse(ctx, dst[q], ncp))) return rc;
    }
    G->nc = nc;
    G->ncp = ncp;
    G->coarse = true;
    return ASB_OK;
}

// x <- A^-1 b by Jacobi-PCG on all 64 columns (b is consumed: it becomes the residual buffer)
static int coarse_correct(asb_ctx* ctx, asb_geo* G, const double* Acinv, const double* r, double* part_rz) {
    const int gb = (G->nc + 3) / 4 < 1024 ? (G->nc + 3) / 4 : 1024;
    hipLaunchKernelGGL(k_restrict64, dim3(gb), dim3(256), 0, ctx->stream, G->agg_ptr, G->agg_mem, G->nc, r, G->rc);
    int rc = asb_gemm_nn(ctx, Acinv, G->ncp, G->rc, GB, G->zc, GB, G->ncp, GB, G->ncp, 1.0, 0.0);
    if (rc) return rc;
    hipLaunchKernelGGL(k_prolong_rz, dim3(G->nblk), dim3(256), 0, ctx->stream, G->n, G->agg, G->zc, r, G->z, part_rz);
    return ASB_OK;
}

static int cg64(asb_ctx* ctx, asb_geo* G, const asb_csr& A, const double* diag, double* b, double* x, int nsrc, double tol,
                int max_iter, int* iters_out, const double* Acinv = nullptr) {
    const int n = G->n, nb = G->nblk;
    double *part_pap = G->part, *part_rz = G->part + (size_t)nb * GB, *part_rr = G->part + (size_t)2 * nb * GB;
    double *rz = G->sc, *rr = G->sc + 2 * GB;       // rz[0..63] current, [64..127] staged, [192..255] initial; rr = sc[128..191]
    const double tol2 = tol * tol;
    ASB_HIP(ctx, hipMemsetAsync(x, 0, (size_t)n * GB * sizeof(double), ctx->stream));
    ASB_HIP(ctx, hipMemsetAsync(G->sc, 0, (size_t)5 * GB * sizeof(double), ctx->stream));
    double* r = b;
    int rcc;
    hipLaunchKernelGGL(k_cg_start, dim3(nb), dim3(256), 0, ctx->stream, n, diag, r, G->z, part_rz, part_rr);
    if (Acinv && (rcc = coarse_correct(ctx, G, Acinv, r, part_rz))) return rcc;
    hipLaunchKernelGGL(k_cg_direction, dim3(nb), dim3(256), 0, ctx->stream, n, nb, part_rz, part_rr, rz, rr, G->z, G->p, 1, tol2);
    hipLaunchKernelGGL(k_cg_commit, dim3(1), dim3(GB), 0, ctx->stream, rz);
    double rr0[GB], rrk[GB];
    ASB_HIP(ctx, hipMemcpyAsync(rr0, rr, sizeof(rr0), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int c = 0; c < GB; ++c) rrk[c] = rr0[c];
    int it = 0, stalled = 0;
    double prev[GB];
    for (int c = 0; c < GB; ++c) prev[c] = rr0[c];
    for (; it < max_iter;) {
        for (int q = 0; q < 25 && it < max_iter; ++q, ++it) {
            hipLaunchKernelGGL(k_spmm64, dim3(nb), dim3(256), 0, ctx->stream, A.rowptr, A.colidx, A.vals, n, G->p, G->ap, part_pap);
            hipLaunchKernelGGL(k_cg_update, dim3(nb), dim3(256), 0, ctx->stream, n, nb, part_pap, rz, diag, x, r, G->p, G->ap, G->z,
                               part_rz, part_rr, tol2);
            if (Acinv && (rcc = coarse_correct(ctx, G, Acinv, r, part_rz))) return rcc;
            hipLaunchKernelGGL(k_cg_direction, dim3(nb), dim3(256), 0, ctx->stream, n, nb, part_rz, part_rr, rz, rr, G->z, G->p, 0, tol2);
            hipLaunchKernelGGL(k_cg_commit, dim3(1), dim3(GB), 0, ctx->stream, rz);
        }
        ASB_CHECK_LAUNCH(ctx);
        ASB_HIP(ctx, hipMemcpyAsync(rrk, rr, sizeof(rrk), hipMemcpyDeviceToHost, ctx->stream));
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        // stop at the requested relative residual, or when rounding has been reached: no column still above the
        // tolerance has set a new best residual for 200 iterations (CG on these systems floors near eps * cond)
        bool done = true, progress = false;
        for (int c = 0; c < nsrc; ++c)
            if (!(rrk[c] <= tol * tol * rr0[c])) {
                done = false;
                if (rrk[c] < 0.9 * prev[c]) progress = true;
            }
        if (done) break;
        if (progress) {
            stalled = 0;
            for (int c = 0; c < nsrc; ++c) prev[c] = rrk[c] < prev[c] ? rrk[c] : prev[c];
        } else if (++stalled >= 8) {      // 200 iterations without a new best residual anywhere
            break;
        }
    }
    if (iters_out) *iters_out = it;
    // loud failure instead of silently wrong distances: Jacobi-PCG is only adequate for well-shaped meshes
    for (int c = 0; c < nsrc; ++c)
        if (!(rrk[c] <= 1e-16 * rr0[c]))      // relative residual 1e-8 is the least we accept
            ASB_FAIL(ctx, ASB_ERR_NUMERIC, "device geodesics: PCG stopped at relative residual %.2e after %d iterations "
                     "(badly conditioned mesh); use the host SuperLU backend", sqrt(rrk[c] / (rr0[c] > 0 ? rr0[c] : 1.0)), it);
    return ASB_OK;
}

// ------------------------------------------------------------------------------------- dense mode
// leading n x n block <- shift (the rank-one gauge term), identity on the padding; the rest of `out` is already zero
__global__ __launch_bounds__(256) void k_dense_init(int n, int np, double shift, double* __restrict__ out) {
    const long long total = (long long)np * np;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int r = (int)(e / np), c = (int)(e % np);
        if (r < n && c < n) { if (shift != 0.0) out[e] = shift; }
        else if (r == c) out[e] = 1.0;
    }
}

// out += CSR matrix (one wave per row; a CSR row holds each column once)
__global__ __launch_bounds__(256) void k_csr_to_dense(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                                      const double* __restrict__ vals, int n, int np, double* __restrict__ out) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int r = blockIdx.x * 4 + wid; r < n; r += gridDim.x * 4) {
        double* row = out + (long long)r * np;
        for (int j = rowptr[r] + lane; j < rowptr[r + 1]; j += 64) row[colidx[j]] += vals[j];
    }
}

// heat step with the explicit inverse: u_c = column src_c of (A - tL)^-1 = its row (symmetric); x is (np x 64) node-major
__global__ __launch_bounds__(256) void k_gather_sources(const double* __restrict__ Hinv, int np, const long long* __restrict__ src,
                                                        int nsrc, double* __restrict__ x) {
    const long long total = (long long)np * GB;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % GB);
        const long long i = e / GB;
        x[e] = (c < nsrc) ? Hinv[src[c] * np + i] : 0.0;
    }
}

// Replaces the two PCG solves by explicit inverses (blocked Gauss-Jordan, asb_dense.hip): (A - tL)^-1 and
// (-L + (gamma / n) 1 1^T)^-1 -- the rank-one term fixes the constant null vector of the Laplacian, which the final
// phi -= min(phi) removes again, so the distances are those of the singular system (utils/support.py:171, 205-206).
extern "C" int asb_geodesic_dense_setup(asb_ctx* ctx) {
    if (!ctx || !ctx->geo) return ASB_ERR_ARG;
    asb_geo* G = ctx->geo;
    const int n = G->n, np = (n + 15) / 16 * 16;
    if (n > 46000) ASB_FAIL(ctx, ASB_ERR_LIMIT, "dense geodesics: %d vertices are too many for explicit inverses", n);
    int rc;
    if ((rc = asb_alloc(ctx, &G->Hinv, (size_t)np * np))) return rc;
    if ((rc = asb_alloc(ctx, &G->Pinv, (size_t)np * np))) return rc;
    const size_t nv = (size_t)np * GB;             // vectors get the padded length (padding rows stay zero)
    double** vecs[] = {&G->x, &G->b};
    for (auto v : vecs) {
        if ((rc = asb_alloc(ctx, v, nv))) return rc;
        ASB_HIP(ctx, hipMemsetAsync(*v, 0, nv * sizeof(double), ctx->stream));
    }
    // gamma = mean diagonal of -L keeps the added eigenvalue inside the spectrum
    std::vector<double> dl((size_t)n);
    ASB_HIP(ctx, hipMemcpyAsync(dl.data(), G->dlap, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double gamma = 0.0;
    for (double v : dl) gamma += v;
    gamma /= n;
    if (!(gamma > 0.0)) ASB_FAIL(ctx, ASB_ERR_NUMERIC, "dense geodesics: the Laplacian has a non-positive mean diagonal");
    const int grid = (np + 3) / 4 < 2048 ? (np + 3) / 4 : 2048;
    ASB_HIP(ctx, hipMemsetAsync(G->Hinv, 0, (size_t)np * np * sizeof(double), ctx->stream));
    ASB_HIP(ctx, hipMemsetAsync(G->Pinv, 0, (size_t)np * np * sizeof(double), ctx->stream));
    hipLaunchKernelGGL(k_dense_init, dim3(2048), dim3(256), 0, ctx->stream, n, np, 0.0, G->Hinv);
    hipLaunchKernelGGL(k_dense_init, dim3(2048), dim3(256), 0, ctx->stream, n, np, gamma / n, G->Pinv);
    hipLaunchKernelGGL(k_csr_to_dense, dim3(grid), dim3(256), 0, ctx->stream, G->heat.rowptr, G->heat.colidx, G->heat.vals, n, np,
                       G->Hinv);
    hipLaunchKernelGGL(k_csr_to_dense, dim3(grid), dim3(256), 0, ctx->stream, G->lap.rowptr, G->lap.colidx, G->lap.vals, n, np,
                       G->Pinv);
    ASB_CHECK_LAUNCH(ctx);
    if ((rc = asb_dense_spd_inverse(ctx, G->Hinv, np))) return rc;
    if ((rc = asb_dense_spd_inverse(ctx, G->Pinv, np))) return rc;
    G->dense = true;
    G->np = np;
    return ASB_OK;
}

// --------------------------------------------------------------------------------------------------------------------
// Slab mode (round 4): a DIRECT sparse solver out of dense building blocks.  Breadth-first level sets of the mesh graph,
// grouped into slabs of ~1500 vertices (geodesic.py: bfs_slabs), make both matrices block tridiagonal: a vertex's
// neighbours lie in its own or an adjacent level.  Block LDL^T: D_0 = A_00, E_k = A_{k,k-1} D_{k-1}^-1, D_k = A_kk - E_k
// A_{k-1,k}, each D_k^-1 by the symmetric blocked Gauss-Jordan of asb_dense.hip, the products on the tiled f64-MFMA GEMM.
// A solve is a forward sweep z_k -= E_k z_{k-1}, w_k = D_k^-1 z_k, and a backward sweep w_k -= E_{k+1}^T w_{k+1} -- 3 small
// GEMMs per slab for 64 right-hand sides.  What it buys over the Jacobi-sweep / PCG sparse mode: no convergence question
// at all -- the heat system's solution spans 14 orders of magnitude and must be right in every component (utils/support.py:
// 181-190 needs the DIRECTION of grad u everywhere); for an M-matrix every term of these sweeps has the same sign, as in
// SuperLU's factorisation, whatever the element sizes (slivers, graded meshes) -- and no size limit but memory: ~3 ns s^2
// doubles per matrix (100 000 vertices, 50 slabs of 2000: 4.8 GB).
// The Laplacian is singular (constants): ONE vertex is grounded (gamma on its diagonal); for the consistent right-hand side
// div X the grounded solution is an exact solution of the singular system, and phi -= min(phi) removes the constant.
// --------------------------------------------------------------------------------------------------------------------
// out (nr_pad x ld) <- block [r0, r0 + nr) x [c0, c0 + nc) of a CSR matrix; pad_diag: 1 on the diagonal of the padding rows
__global__ __launch_bounds__(256) void k_csr_block_dense(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                                         const double* __restrict__ vals, int r0, int nr, int c0, int nc,
                                                         double* __restrict__ out, int ld, int nr_pad, int pad_diag) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int r = blockIdx.x * 4 + wid; r < nr_pad; r += gridDim.x * 4) {
        double* row = out + (long long)r * ld;
        if (r >= nr) {
            if (pad_diag && lane == 0 && r < ld) row[r] = 1.0;
            continue;
        }
        for (int j = rowptr[r0 + r] + lane; j < rowptr[r0 + r + 1]; j += 64) {
            const int c = colidx[j] - c0;
            if (c >= 0 && c < nc) row[c] = vals[j];
        }
    }
}
// z[pos[v]] = b[v] (64 columns) / x[v] = w[pos[v]]
__global__ __launch_bounds__(256) void k_bt_permute(const double* __restrict__ in, const int* __restrict__ pos, int n,
                                                    double* __restrict__ out, int gather) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int v = blockIdx.x * 4 + wid; v < n; v += gridDim.x * 4) {
        if (gather) out[(long long)pos[v] * GB + lane] = in[(long long)v * GB + lane];
        else out[(long long)v * GB + lane] = in[(long long)pos[v] * GB + lane];
    }
}
__global__ void k_add_one(double* __restrict__ p, double v) { *p += v; }

// out (M x 64) = beta out + alpha A (M x Kc, ld lda) Z (Kc x 64): the product of a slab sweep.  The tiled GEMM of asb_dense.hip
// makes 14 x 1 tiles of it (a 3-way split of the contraction + a finishing launch: ~90 us); here a block of four waves owns 16
// rows of A, the waves take alternate 16-wide chunks of the contraction (A rows as 128-byte runs, Z from L2), f64 MFMA 16x16x4,
// and meet through LDS in a fixed order: M / 16 blocks, one launch.  M, Kc multiples of 16.
typedef double bt_d4 __attribute__((ext_vector_type(4)));
template <int NCT>
__global__ __launch_bounds__(256) void k_slab_gemm64(const double* __restrict__ A, long long lda, const double* __restrict__ Z,
                                                     double* __restrict__ out, int M, int Kc, double alpha, double beta) {
    constexpr int nct = NCT;
    // nct: column tiles of 16 that carry fields (a batch of nsrc <= 64 sources fills the first ceil(nsrc / 16)); the rest is left as it is
    __shared__ double red[3][4][4][64];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, i = l & 15, g = l >> 4;
    const int r0 = blockIdx.x * 16;
    bt_d4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = (bt_d4){0.0, 0.0, 0.0, 0.0};
    const double* arow = A + (long long)(r0 + i) * lda + 4 * g;
    for (int c = w; c < Kc / 16; c += 4) {
        const double4 a = *reinterpret_cast<const double4*>(arow + 16 * c);
        const double* zr = Z + (long long)(16 * c + 4 * g) * GB + i;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            if (ct >= nct) break;
            acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, zr[0 * GB + 16 * ct], acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, zr[1 * GB + 16 * ct], acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.z, zr[2 * GB + 16 * ct], acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.w, zr[3 * GB + 16 * ct], acc[ct], 0, 0, 0);
        }
    }
    if (w > 0) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (ct < nct) red[w - 1][ct][q][l] = acc[ct][q];
    }
    __syncthreads();
    if (w == 0) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (ct >= nct) continue;
                const double s = ((acc[ct][q] + red[0][ct][q][l]) + red[1][ct][q][l]) + red[2][ct][q][l];
                double* dst = out + (long long)(r0 + g + 4 * q) * GB + 16 * ct + i;      // C[row 4 q + g][column i] of the 16 x 16 tile
                *dst = (beta == 0.0 ? 0.0 : beta * *dst) + alpha * s;
            }
    }
    (void)M;
}
static int slab_gemm64(asb_ctx* ctx, const double* A, long long lda, const double* Z, double* out, int M, int Kc, double alpha, double beta,
                       int nct = 4) {
    if ((M | Kc) & 15) ASB_FAIL(ctx, ASB_ERR_ARG, "slab_gemm64: %d x %d is not a multiple of 16", M, Kc);
    switch (nct < 1 ? 1 : (nct > 4 ? 4 : nct)) {
        case 1: hipLaunchKernelGGL(k_slab_gemm64<1>, dim3(M / 16), dim3(256), 0, ctx->stream, A, lda, Z, out, M, Kc, alpha, beta); break;
        case 2: hipLaunchKernelGGL(k_slab_gemm64<2>, dim3(M / 16), dim3(256), 0, ctx->stream, A, lda, Z, out, M, Kc, alpha, beta); break;
        case 3: hipLaunchKernelGGL(k_slab_gemm64<3>, dim3(M / 16), dim3(256), 0, ctx->stream, A, lda, Z, out, M, Kc, alpha, beta); break;
        default: hipLaunchKernelGGL(k_slab_gemm64<4>, dim3(M / 16), dim3(256), 0, ctx->stream, A, lda, Z, out, M, Kc, alpha, beta); break;
    }
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

static int bt_factor(asb_ctx* ctx, asb_geo* G, asb_bt& bt, const int* rp, const int* ci, const double* va, const std::vector<int>& ptr,
                     int ground_row, double ground_val) {
    const int ns = (int)G->bt_sz.size();
    bt.release();
    bt.Dinv.assign(ns, nullptr);
    bt.E.assign(ns, nullptr);
    bt.Et.assign(ns, nullptr);
    int rc;
    double* Ct = nullptr;                      // A_{k-1,k} (sp_{k-1} x sp_k): scratch, sized for the largest pair
    size_t ct_cap = 0;
    for (int k = 0; k < ns; ++k) {
        const int sp = G->bt_sz[k], s = ptr[k + 1] - ptr[k];
        ASB_HIP(ctx, hipMalloc((void**)&bt.Dinv[k], (size_t)sp * sp * sizeof(double)));
        ASB_HIP(ctx, hipMemsetAsync(bt.Dinv[k], 0, (size_t)sp * sp * sizeof(double), ctx->stream));
        const int grid = (sp + 3) / 4 < 2048 ? (sp + 3) / 4 : 2048;
        hipLaunchKernelGGL(k_csr_block_dense, dim3(grid), dim3(256), 0, ctx->stream, rp, ci, va, ptr[k], s, ptr[k], s, bt.Dinv[k], sp, sp, 1);
        if (ground_row >= ptr[k] && ground_row < ptr[k + 1]) {
            const int q = ground_row - ptr[k];
            hipLaunchKernelGGL(k_add_one, dim3(1), dim3(1), 0, ctx->stream, bt.Dinv[k] + (size_t)q * sp + q, ground_val);
        }
        if (k > 0) {
            const int spm = G->bt_sz[k - 1], sm = ptr[k] - ptr[k - 1];
            ASB_HIP(ctx, hipMalloc((void**)&bt.E[k], (size_t)sp * spm * sizeof(double)));
            ASB_HIP(ctx, hipMalloc((void**)&bt.Et[k], (size_t)sp * spm * sizeof(double)));
            const size_t need = (size_t)sp * spm;
            if (need > ct_cap) {
                if (Ct) (void)hipFree(Ct);
                ASB_HIP(ctx, hipMalloc((void**)&Ct, 2 * need * sizeof(double)));
                ct_cap = need;
            }
            double* C = Ct + ct_cap;           // A_{k,k-1} (sp x sp_{k-1})
            ASB_HIP(ctx, hipMemsetAsync(Ct, 0, 2 * ct_cap * sizeof(double), ctx->stream));
            hipLaunchKernelGGL(k_csr_block_dense, dim3(grid), dim3(256), 0, ctx->stream, rp, ci, va, ptr[k], s, ptr[k - 1], sm, C, spm, sp, 0);
            const int gridm = (spm + 3) / 4 < 2048 ? (spm + 3) / 4 : 2048;
            hipLaunchKernelGGL(k_csr_block_dense, dim3(gridm), dim3(256), 0, ctx->stream, rp, ci, va, ptr[k - 1], sm, ptr[k], s, Ct, sp, spm, 0);
            ASB_CHECK_LAUNCH(ctx);
            // E_k = A_{k,k-1} D_{k-1}^-1 ; D_k -= E_k A_{k-1,k}
            if ((rc = asb_gemm_nn(ctx, C, spm, bt.Dinv[k - 1], spm, bt.E[k], spm, sp, spm, spm, 1.0, 0.0))) return rc;
            if ((rc = asb_gemm_nn(ctx, bt.E[k], spm, Ct, sp, bt.Dinv[k], sp, sp, sp, spm, -1.0, 1.0))) return rc;
            if ((rc = asb_transpose(ctx, bt.E[k], sp, spm, bt.Et[k]))) return rc;
        }
        ASB_CHECK_LAUNCH(ctx);
        if ((rc = asb_dense_spd_inverse(ctx, bt.Dinv[k], sp))) return rc;
    }
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (Ct) (void)hipFree(Ct);
    return ASB_OK;
}

// slab_ptr (nslab + 1): slab boundaries in the PERMUTED numbering; perm_of_vertex (n): vertex -> permuted index; the two CSR
// matrices in the permuted numbering (A - tL and -L)
extern "C" int asb_geodesic_bt_setup(asb_ctx* ctx, int nslab, const int* slab_ptr, const int* perm_of_vertex, const int* heat_rp,
                                     const int* heat_ci, const double* heat_v, const int* lap_rp, const int* lap_ci,
                                     const double* lap_v) {
    if (!ctx || !ctx->geo || nslab < 1 || !slab_ptr || !perm_of_vertex || !heat_rp || !lap_rp) return ASB_ERR_ARG;
    asb_geo* G = ctx->geo;
    const int n = G->n;
    if (slab_ptr[0] != 0 || slab_ptr[nslab] != n) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_geodesic_bt_setup: the slabs do not cover the %d vertices", n);
    std::vector<int> ptr(slab_ptr, slab_ptr + nslab + 1);
    G->bt = false;
    G->bt_off.assign(nslab, 0);
    G->bt_sz.assign(nslab, 0);
    int off = 0;
    for (int k = 0; k < nslab; ++k) {
        const int s = ptr[k + 1] - ptr[k];
        if (s < 1) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_geodesic_bt_setup: empty slab %d", k);
        G->bt_off[k] = off;
        G->bt_sz[k] = (s + 15) / 16 * 16;
        off += G->bt_sz[k];
    }
    G->bt_npad = off;
    // vertex -> padded position
    std::vector<int> slab_of(n), pos(n);
    for (int k = 0; k < nslab; ++k)
        for (int q = ptr[k]; q < ptr[k + 1]; ++q) slab_of[q] = k;
    for (int v = 0; v < n; ++v) {
        const int q = perm_of_vertex[v];
        if (q < 0 || q >= n) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_geodesic_bt_setup: bad permutation");
        pos[v] = G->bt_off[slab_of[q]] + (q - ptr[slab_of[q]]);
    }
    int rc;
    if ((rc = asb_alloc(ctx, &G->bt_pos, (size_t)n))) return rc;
    if ((rc = asb_alloc(ctx, &G->bt_z, (size_t)G->bt_npad * GB))) return rc;
    if ((rc = asb_alloc(ctx, &G->bt_w, (size_t)G->bt_npad * GB))) return rc;
    ASB_HIP(ctx, hipMemcpyAsync(G->bt_pos, pos.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    // the permuted matrices go to the device for the block extraction only
    auto up = [&](const int* rp, const int* ci, const double* va, int*& drp, int*& dci, double*& dva) -> int {
        const long long nnz = rp[n];
        ASB_HIP(ctx, hipMalloc((void**)&drp, (size_t)(n + 1) * sizeof(int)));
        ASB_HIP(ctx, hipMalloc((void**)&dci, (size_t)(nnz > 0 ? nnz : 1) * sizeof(int)));
        ASB_HIP(ctx, hipMalloc((void**)&dva, (size_t)(nnz > 0 ? nnz : 1) * sizeof(double)));
        ASB_HIP(ctx, hipMemcpyAsync(drp, rp, (size_t)(n + 1) * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        ASB_HIP(ctx, hipMemcpyAsync(dci, ci, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        ASB_HIP(ctx, hipMemcpyAsync(dva, va, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        return ASB_OK;
    };
    int *hrp = nullptr, *hci = nullptr, *lrp = nullptr, *lci = nullptr;
    double *hva = nullptr, *lva = nullptr;
    auto body = [&]() -> int {
        int r;
        if ((r = up(heat_rp, heat_ci, heat_v, hrp, hci, hva))) return r;
        if ((r = up(lap_rp, lap_ci, lap_v, lrp, lci, lva))) return r;
        if ((r = bt_factor(ctx, G, G->Hbt, hrp, hci, hva, ptr, -1, 0.0))) return r;
        // ground the LAST vertex of the permuted numbering with the mean diagonal of -L (keeps the conditioning)
        double gamma = 0.0;
        for (int q = 0; q < n; ++q)
            for (int j = lap_rp[q]; j < lap_rp[q + 1]; ++j)
                if (lap_ci[j] == q) gamma += lap_v[j];
        gamma /= n;
        if (!(gamma > 0.0)) ASB_FAIL(ctx, ASB_ERR_NUMERIC, "slab geodesics: the Laplacian has a non-positive mean diagonal");
        return bt_factor(ctx, G, G->Pbt, lrp, lci, lva, ptr, n - 1, gamma);
    };
    rc = body();
    (void)hipStreamSynchronize(ctx->stream);
    for (void* p : {(void*)hrp, (void*)hci, (void*)hva, (void*)lrp, (void*)lci, (void*)lva})
        if (p) (void)hipFree(p);
    if (rc) return rc;
    G->bt = true;
    G->dense = false;
    G->coarse = false;
    return ASB_OK;
}

// x (n x 64, vertex order) <- A^-1 b for a factorised matrix
static int bt_solve(asb_ctx* ctx, asb_geo* G, const asb_bt& bt, const double* b, double* x, int nct) {
    const int ns = (int)G->bt_sz.size(), n = G->n;
    int rc;
    const int grid = (n + 3) / 4 < 2048 ? (n + 3) / 4 : 2048;
    ASB_HIP(ctx, hipMemsetAsync(G->bt_z, 0, (size_t)G->bt_npad * GB * sizeof(double), ctx->stream));
    hipLaunchKernelGGL(k_bt_permute, dim3(grid), dim3(256), 0, ctx->stream, b, G->bt_pos, n, G->bt_z, 1);
    ASB_CHECK_LAUNCH(ctx);
    for (int k = 0; k < ns; ++k) {             // forward: z_k -= E_k z_{k-1};  w_k = D_k^-1 z_k
        const int sp = G->bt_sz[k];
        double* zk = G->bt_z + (size_t)G->bt_off[k] * GB;
        if (k > 0) {
            const int spm = G->bt_sz[k - 1];
            if ((rc = slab_gemm64(ctx, bt.E[k], spm, G->bt_z + (size_t)G->bt_off[k - 1] * GB, zk, sp, spm, -1.0, 1.0, nct))) return rc;
        }
        if ((rc = slab_gemm64(ctx, bt.Dinv[k], sp, zk, G->bt_w + (size_t)G->bt_off[k] * GB, sp, sp, 1.0, 0.0, nct))) return rc;
    }
    for (int k = ns - 2; k >= 0; --k) {        // backward: w_k -= E_{k+1}^T w_{k+1}
        const int sp = G->bt_sz[k], spn = G->bt_sz[k + 1];
        if ((rc = slab_gemm64(ctx, bt.Et[k + 1], spn, G->bt_w + (size_t)G->bt_off[k + 1] * GB, G->bt_w + (size_t)G->bt_off[k] * GB, sp, spn, -1.0,
                              1.0, nct))) return rc;
    }
    hipLaunchKernelGGL(k_bt_permute, dim3(grid), dim3(256), 0, ctx->stream, G->bt_w, G->bt_pos, n, x, 0);
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

// distances of nsrc (<= 64) sources whose vertex ids are on the device; result (nsrc, n), min-shifted, in ctx->geo_out
static int geodesic_solve_dev(asb_ctx* ctx, const long long* src_dev, int nsrc, double tol, int* it1_out, int* it2_out) {
    asb_geo* G = ctx->geo;
    const int n = G->n;
    int rc;
    if (!G->dense && (rc = asb_alloc(ctx, &G->z, (size_t)n * GB))) return rc;
    // heat step: (A - tL) u = delta
    int it1 = 0, it2 = 0;
    if (G->dense) {
        hipLaunchKernelGGL(k_gather_sources, dim3(1024), dim3(256), 0, ctx->stream, G->Hinv, G->np, src_dev, nsrc, G->x);
    } else if (G->bt) {
        ASB_HIP(ctx, hipMemsetAsync(G->b, 0, (size_t)n * GB * sizeof(double), ctx->stream));
        hipLaunchKernelGGL(k_set_sources, dim3(1), dim3(GB), 0, ctx->stream, G->b, src_dev, nsrc);
        if ((rc = bt_solve(ctx, G, G->Hbt, G->b, G->x, (nsrc + 15) / 16))) return rc;
    } else {
        ASB_HIP(ctx, hipMemsetAsync(G->b, 0, (size_t)n * GB * sizeof(double), ctx->stream));
        hipLaunchKernelGGL(k_set_sources, dim3(1), dim3(GB), 0, ctx->stream, G->b, src_dev, nsrc);
        if (G->coarse) {        // sparse mode proper: componentwise-accurate heat step (Jacobi sweeps), PCG only for Poisson
            if ((rc = heat_jacobi64(ctx, G, G->b, G->x, nsrc, &it1))) return rc;
        } else if ((rc = cg64(ctx, G, G->heat, G->dheat, G->b, G->x, nsrc, tol, 4000, &it1))) return rc;
    }
    // gradient, normalise, divergence
    const int gb = (G->m3 + 3) / 4 < 1024 ? (G->m3 + 3) / 4 : 1024;
    hipLaunchKernelGGL(k_spmm64, dim3(gb), dim3(256), 0, ctx->stream, G->grad.rowptr, G->grad.colidx, G->grad.vals, G->m3, G->x, G->g,
                       (double*)nullptr);
    hipLaunchKernelGGL(k_normalise_field, dim3(gb), dim3(256), 0, ctx->stream, G->g, G->m3 / 3);
    hipLaunchKernelGGL(k_spmm64, dim3(G->nblk), dim3(256), 0, ctx->stream, G->div.rowptr, G->div.colidx, G->div.vals, n, G->g, G->b,
                       (double*)nullptr);
    ASB_CHECK_LAUNCH(ctx);
    // Poisson step: L phi = div  <=>  (-L) phi = -div ; solve (-L) y = div and negate through the min shift (phi = -y)
    if (G->dense) {        // y = (-L + gamma/n 1 1^T)^-1 div : one (np x np) by (np x 64) product
        // (16 rows of the inverse per block, 925 blocks at 14 800 vertices: the 128 x 128 tiles of asb_gemm_nn made 116 blocks of it -- 0.98 ms)
        if ((rc = slab_gemm64(ctx, G->Pinv, G->np, G->b, G->x, G->np, G->np, 1.0, 0.0, (nsrc + 15) / 16))) return rc;
    } else if (G->bt) {    // the grounded Laplacian, factorised: an exact solution of the singular system for the consistent div X
        if ((rc = asb_alloc(ctx, &G->z, (size_t)n * GB))) return rc;
        ASB_HIP(ctx, hipMemcpyAsync(G->z, G->b, (size_t)n * GB * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        if ((rc = bt_solve(ctx, G, G->Pbt, G->z, G->x, (nsrc + 15) / 16))) return rc;
    } else {
        hipLaunchKernelGGL(k_remove_mean, dim3(GB), dim3(256), 0, ctx->stream, G->b, n);
        if ((rc = cg64(ctx, G, G->lap, G->dlap, G->b, G->x, nsrc, tol, 8000, &it2, G->coarse ? G->AcP : nullptr))) return rc;
    }
    // phi = -y; phi -= min(phi)  ==  max(y) - y : done by negating in place first
    hipLaunchKernelGGL(k_scale_vec, dim3(G->nblk), dim3(256), 0, ctx->stream, G->x, (long long)n * GB, -1.0);
    if ((rc = asb_alloc(ctx, &ctx->geo_out, (size_t)GB * n))) return rc;
    hipLaunchKernelGGL(k_shift_min_out, dim3(nsrc), dim3(256), 0, ctx->stream, G->x, n, nsrc, ctx->geo_out);
    ASB_CHECK_LAUNCH(ctx);
    if (it1_out) *it1_out = it1;
    if (it2_out) *it2_out = it2;
    return ASB_OK;
}

// ---- single source, dense backend: the per-component query of support='local'.  Column vectors instead of the 64-wide
// batch: u = row src of (A - tL)^-1 (symmetric), X = -grad u / |grad u| per face, b = div X, y = Pinv b through the upper
// triangle of the symmetric Pinv only (each 128 x 128 tile is read once and used for y_I += T b_J and y_J += T^T b_I),
// phi = max(y) - y.  All sums have a fixed order.
#define SV_T 128
__global__ __launch_bounds__(256) void k_face_field1(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                                     const double* __restrict__ vals, int ntri, const double* __restrict__ Hinv,
                                                     int np, const long long* __restrict__ src, double* __restrict__ g) {
    const double* u = Hinv + src[0] * (long long)np;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < ntri; t += gridDim.x * 256) {
        double q[3];
        for (int d = 0; d < 3; ++d) {
            const int r = 3 * t + d;
            double y = 0.0;
            for (int j = rowptr[r]; j < rowptr[r + 1]; ++j) y += vals[j] * u[colidx[j]];
            q[d] = y;
        }
        const double len = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
        g[3 * t] = -q[0] / len; g[3 * t + 1] = -q[1] / len; g[3 * t + 2] = -q[2] / len;
    }
}
// y = A x, eight lanes per row (lane q of the group takes the entries q, q + 8, ... of the row; fixed combination order)
__global__ __launch_bounds__(256) void k_spmv1(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                               const double* __restrict__ vals, int rows, int rows_pad,
                                               const double* __restrict__ x, double* __restrict__ y) {
    const int sub = threadIdx.x & 7;
    for (int r = (blockIdx.x * 256 + threadIdx.x) >> 3; r < rows_pad; r += gridDim.x * 32) {      // uniform per group
        double a = 0.0;
        if (r < rows)
            for (int j = rowptr[r] + sub; j < rowptr[r + 1]; j += 8) a += vals[j] * x[colidx[j]];
        a += __shfl_xor(a, 1);
        a += __shfl_xor(a, 2);
        a += __shfl_xor(a, 4);
        if (sub == 0) y[r] = a;                    // rows..rows_pad: explicit zeros (the buffer is shared with the batch path)
    }
}
// one block per tile (ib <= jb) of the symmetric matrix: prow[jb][i] = (T b_J)_i, pcol[ib][j] = (T^T b_I)_j (off-diagonal only)
__global__ __launch_bounds__(256) void k_symv_tiles(const double* __restrict__ P, int np, int nb, const double* __restrict__ b,
                                                    double* __restrict__ prow, double* __restrict__ pcol) {
    __shared__ double sh[4][SV_T];
    int t = blockIdx.x, ib = 0;
    while (t >= nb - ib) { t -= nb - ib; ++ib; }
    const int jb = ib + t;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int i0 = ib * SV_T, j0 = jb * SV_T, c = j0 + 2 * lane;
    const bool okc = c < np;                       // np is even: the pair is in or out together
    const double bj0 = okc ? b[c] : 0.0, bj1 = okc ? b[c + 1] : 0.0;
    double ca0 = 0.0, ca1 = 0.0;
    for (int rr = 0; rr < SV_T / 4; rr += 4) {
        double2 a[4];
        double bi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = i0 + (rr + q) * 4 + wid;
            const bool ok = okc && r < np;
            a[q] = ok ? *reinterpret_cast<const double2*>(P + (long long)r * np + c) : make_double2(0.0, 0.0);
            bi[q] = r < np ? b[r] : 0.0;
        }
        double sr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sr[q] = a[q].x * bj0 + a[q].y * bj1;
            ca0 += a[q].x * bi[q];
            ca1 += a[q].y * bi[q];
        }
        wave_sum_dpp<4>(sr);                      // (four row sums per turn through the DPP / permlane tree: no LDS crossbar)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = i0 + (rr + q) * 4 + wid;
            if (lane == 0 && r < np) prow[(long long)jb * np + r] = sr[q];
        }
    }
    if (ib == jb) return;
    sh[wid][2 * lane] = ca0; sh[wid][2 * lane + 1] = ca1;
    __syncthreads();
    if (threadIdx.x < SV_T && j0 + (int)threadIdx.x < np) {
        const int x = threadIdx.x;
        pcol[(long long)ib * np + j0 + x] = (sh[0][x] + sh[1][x]) + (sh[2][x] + sh[3][x]);
    }
}
__global__ __launch_bounds__(256) void k_symv_finish(const double* __restrict__ prow, const double* __restrict__ pcol, int np, int nb,
                                                     double* __restrict__ y) {
    __shared__ double sh[4][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;      // 64 entries per block, the nb partials in 4 ordered chunks
    const int i = blockIdx.x * 64 + lane;
    double s = 0.0;
    if (i < np) {
        const int mb = i / SV_T, q0 = g * nb / 4, q1 = (g + 1) * nb / 4;
        for (int q = q0; q < q1; ++q) s += (q < mb ? pcol : prow)[(long long)q * np + i];
    }
    sh[g][lane] = s;
    __syncthreads();
    if (g == 0 && i < np) y[i] = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
}
// out[i] = max(y) - y[i]
__global__ __launch_bounds__(1024) void k_max_minus(const double* __restrict__ y, int n, double* __restrict__ out) {
    __shared__ double sh[1024];
    double m = -1.0e300;
    for (int i = threadIdx.x; i < n; i += 1024) m = fmax(m, y[i]);
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + o]);
        __syncthreads();
    }
    m = sh[0];
    for (int i = threadIdx.x; i < n; i += 1024) out[i] = m - y[i];
}

static int geodesic_solve1_dense(asb_ctx* ctx, const long long* src_dev) {
    asb_geo* G = ctx->geo;
    const int n = G->n, np = G->np, nb = (np + SV_T - 1) / SV_T, ntri = G->m3 / 3;
    int rc;
    if ((rc = asb_alloc(ctx, &G->sv_part, (size_t)2 * nb * np))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->geo_out, (size_t)GB * n))) return rc;
    // G->g (3M x 64), G->b and G->x (np x 64) are the batch buffers: their heads serve as the single columns
    hipLaunchKernelGGL(k_face_field1, dim3((ntri + 255) / 256), dim3(256), 0, ctx->stream, G->grad.rowptr, G->grad.colidx, G->grad.vals,
                       ntri, G->Hinv, np, src_dev, G->g);
    hipLaunchKernelGGL(k_spmv1, dim3((np + 31) / 32), dim3(256), 0, ctx->stream, G->div.rowptr, G->div.colidx, G->div.vals, n, np,
                       G->g, G->b);
    double* prow = G->sv_part;
    double* pcol = G->sv_part + (size_t)nb * np;
    hipLaunchKernelGGL(k_symv_tiles, dim3(nb * (nb + 1) / 2), dim3(256), 0, ctx->stream, G->Pinv, np, nb, G->b, prow, pcol);
    hipLaunchKernelGGL(k_symv_finish, dim3((np + 63) / 64), dim3(256), 0, ctx->stream, prow, pcol, np, nb, G->x);
    hipLaunchKernelGGL(k_max_minus, dim3(1), dim3(1024), 0, ctx->stream, G->x, n, ctx->geo_out);
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

// distances from each of nsrc (<= 64) source vertices: out (nsrc, n), host.  iters (optional): CG iterations of the two solves.
extern "C" int asb_geodesic_solve(asb_ctx* ctx, const int64_t* sources, int nsrc, double tol, double* out, int* iters) {
    if (!ctx || !ctx->geo || !sources || nsrc < 1 || nsrc > GB || !out) return ASB_ERR_ARG;
    if (!(tol >= 1e-14)) tol = 1e-14;      // below the rounding floor CG only wanders (and can blow up)
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->geo_src, (size_t)GB))) return rc;
    ASB_HIP(ctx, hipMemcpyAsync(ctx->geo_src, sources, (size_t)nsrc * sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
    int it1 = 0, it2 = 0;
    if ((rc = geodesic_solve_dev(ctx, ctx->geo_src, nsrc, tol, &it1, &it2))) return rc;
    ASB_HIP(ctx, hipMemcpyAsync(out, ctx->geo_out, (size_t)nsrc * ctx->geo->n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (iters) { iters[0] = it1; iters[1] = it2; }
    return ASB_OK;
}

// ---- distance fields kept on the device (SPLOCS asks for the fields of its K centres in every outer iteration,
// posComponents.py:158-165, and most centres stay): solve the fields of nsrc (<= 64) new sources and append them to the
// cache; *slot0 = slot of the first one.  ASB_ERR_LIMIT when the cache is full (asb_geodesic_cache_clear empties it).
extern "C" int asb_geodesic_cache_add(asb_ctx* ctx, const int64_t* sources, int nsrc, double tol, int64_t* slot0) {
    if (!ctx || !ctx->geo || !sources || nsrc < 1 || nsrc > GB || !slot0) return ASB_ERR_ARG;
    asb_geo* G = ctx->geo;
    if (G->cached + nsrc > (long long)ASB_GEO_CACHE_SLABS * GB) ASB_FAIL(ctx, ASB_ERR_LIMIT, "geodesic field cache is full (%lld fields)", G->cached);
    for (int c = 0; c < nsrc; ++c)
        if (sources[c] < 0 || sources[c] >= G->n) ASB_FAIL(ctx, ASB_ERR_ARG, "geodesic source %lld outside the mesh", (long long)sources[c]);
    if (!(tol >= 1e-14)) tol = 1e-14;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->geo_src, (size_t)GB))) return rc;
    ASB_HIP(ctx, hipMemcpyAsync(ctx->geo_src, sources, (size_t)nsrc * sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = geodesic_solve_dev(ctx, ctx->geo_src, nsrc, tol, nullptr, nullptr))) return rc;
    const size_t n = (size_t)G->n;
    for (int c = 0; c < nsrc;) {                 // consecutive slots of one slab are consecutive rows: one copy per slab touched
        const long long q = G->cached + c;
        if ((rc = asb_alloc(ctx, &G->slab[q / GB], (size_t)GB * n))) return rc;      // keeps an existing slab (same size)
        const int room = (int)(GB - q % GB), cnt = nsrc - c < room ? nsrc - c : room;
        ASB_HIP(ctx, hipMemcpyAsync(G->slab[q / GB] + (size_t)(q % GB) * n, ctx->geo_out + (size_t)c * n, (size_t)cnt * n * sizeof(double),
                                    hipMemcpyDeviceToDevice, ctx->stream));
        c += cnt;
    }
    *slot0 = G->cached;
    G->cached += nsrc;
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));      // `sources` is the caller's: its upload must not outlive this call
    return ASB_OK;
}
extern "C" int asb_geodesic_cache_clear(asb_ctx* ctx) {
    if (!ctx) return ASB_ERR_ARG;
    if (ctx->geo) ctx->geo->cached = 0;
    return ASB_OK;
}
// device address of a cached field (nullptr: no such slot) and the mesh size, for asb_splocs.hip
const double* asb_geo_cached_field(asb_ctx* ctx, long long slot, long long* n_out) {
    if (!ctx->geo || slot < 0 || slot >= ctx->geo->cached) return nullptr;
    if (n_out) *n_out = ctx->geo->n;
    return ctx->geo->slab[slot / GB] + (size_t)(slot % GB) * ctx->geo->n;
}

// ---- support='local' without a host round trip per component (posComponents.py:87-105 with the dense geodesics):
// the vertex asb_deflate_pick chose is read from the device, its distance field solved, s = 1 - support_map (:61-64)
// formed for this shard's vertices and the deflation pass applied -- everything queued on the context's stream.
__global__ void k_src_from_pick(const double* __restrict__ scal, long long k, long long* __restrict__ src) {
    src[0] = __double_as_longlong(scal[k * 4 + 2]);
}
__global__ __launch_bounds__(256) void k_support_weights(const double* __restrict__ phi, long long v0, long long n_loc, double dmin,
                                                         double dmax, double* __restrict__ s) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_loc; i += (long long)gridDim.x * 256) {
        const double p = fmin(fmax(phi[v0 + i], dmin), dmax);
        s[i] = 1.0 - (p - dmin) / (dmax - dmin);
    }
}

extern "C" int asb_deflate_apply_geodesic(asb_ctx* ctx, int64_t k, double dmin, double dmax) {
    if (!ctx || !ctx->geo || !ctx->scal) return ASB_ERR_ARG;
    if (!ctx->geo->dense && !ctx->geo->bt) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_deflate_apply_geodesic needs the dense or the slab geodesic backend");
    if (ctx->geo->n != ctx->N_glob) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_deflate_apply_geodesic: the mesh has %d vertices, the snapshots %lld",
                                            ctx->geo->n, (long long)ctx->N_glob);
    if (k < 0 || k >= ctx->K) return ASB_ERR_ARG;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->geo_src, (size_t)GB))) return rc;
    hipLaunchKernelGGL(k_src_from_pick, dim3(1), dim3(1), 0, ctx->stream, ctx->scal, (long long)k, ctx->geo_src);
    if (ctx->geo->dense) rc = geodesic_solve1_dense(ctx, ctx->geo_src);
    else rc = geodesic_solve_dev(ctx, ctx->geo_src, 1, 1e-13, nullptr, nullptr);      // (slab mode: the batch solver with one source)
    if (rc) return rc;
    const int grid = (int)((ctx->n_loc + 255) / 256 < 1024 ? (ctx->n_loc + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_support_weights, dim3(grid), dim3(256), 0, ctx->stream, ctx->geo_out, (long long)ctx->v0, (long long)ctx->n_loc,
                       dmin, dmax, ctx->s_dev);
    ASB_CHECK_LAUNCH(ctx);
    return asb_deflate_apply_dev(ctx, k, ctx->s_dev);
}

void asb_geo_free(asb_ctx* ctx) {
    if (ctx->geo) {
        delete ctx->geo;
        ctx->geo = nullptr;
    }
}
