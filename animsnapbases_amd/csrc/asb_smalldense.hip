// asb_smalldense.hip -- the SMALL dense solvers that sit between the big device steps, on the device, so that no
// host LAPACK call is left on the path (gfx950 only):
//
//   * symmetric TRIDIAGONAL eigen-problem (after asb_eig.hip's Householder reduction of the F x F Gram matrix of
//     config 5): all eigenvalues by bisection on Sturm counts (one thread per eigenvalue), the k leading vectors by
//     inverse iteration (EISPACK TINVIT's elimination with interchanges, one thread per vector, work arrays
//     interleaved so that neighbouring threads touch neighbouring words) -- replaces LAPACK sterf / stemr behind the
//     `svd` of snapbases/constraintsComponents.py:307;
//   * one-sided (Hestenes) JACOBI on the rows of a small matrix: singular values + left vectors of the K x F factor of
//     the Rayleigh-Ritz step (the reference's accuracy comes from LAPACK's gesdd on A itself, :307), and the
//     eigen-decomposition of K x K Gram matrices with K > 128 (scipy.linalg.orth of snapbases/posComponents.py:287);
//     round-robin (circle-method) pair ordering, one launch per round, one block per pair;
//   * blocked CHOLESKY factor + triangular inverse of a K x K SPD matrix for any K (CholeskyQR of
//     `qr(.., mode='economic')`, constraintsComponents.py:433, with K = 200 ... 1000 in the reference's configurations).
//
// Everything is deterministic (ordered reductions, no floating-point atomics): ranks that hold the same small matrix
// compute bit-identical factors.
#include "asb_common.h"

#include <cfloat>
#include <cstring>
#include <cmath>
#include <vector>

#define ASB_EPS 2.220446049250313e-16

// ======================================================================================================================
// 1. symmetric tridiagonal eigen-problem
// ======================================================================================================================
enum { TRI_GL = 0, TRI_GU = 1, TRI_BNORM = 2, TRI_PIVMIN = 3, TRI_ATOL = 4 };

// e2[j] = e[j]^2, Gershgorin interval, norm, smallest admissible pivot of the Sturm recurrence, absolute tolerance
__global__ __launch_bounds__(256) void k_tri_prepare(const double* __restrict__ d, const double* __restrict__ e, int n,
                                                     double* __restrict__ e2, double* __restrict__ sc) {
    __shared__ double s_lo[4], s_hi[4], s_em[4];
    double gl = DBL_MAX, gu = -DBL_MAX, em = 0.0;
    for (int j = threadIdx.x; j < n; j += 256) {
        const double r = (j > 0 ? fabs(e[j - 1]) : 0.0) + (j < n - 1 ? fabs(e[j]) : 0.0);
        gl = fmin(gl, d[j] - r);
        gu = fmax(gu, d[j] + r);
        if (j < n - 1) {
            const double ee = e[j] * e[j];
            e2[j] = ee;
            em = fmax(em, ee);
        }
    }
    gl = -wave_max(-gl);
    gu = wave_max(gu);
    em = wave_max(em);
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = gl; s_hi[threadIdx.x >> 6] = gu; s_em[threadIdx.x >> 6] = em; }
    __syncthreads();
    if (threadIdx.x == 0) {
        gl = fmin(fmin(s_lo[0], s_lo[1]), fmin(s_lo[2], s_lo[3]));
        gu = fmax(fmax(s_hi[0], s_hi[1]), fmax(s_hi[2], s_hi[3]));
        em = fmax(fmax(s_em[0], s_em[1]), fmax(s_em[2], s_em[3]));
        const double bnorm = fmax(fabs(gl), fabs(gu));
        const double pivmin = DBL_MIN * fmax(1.0, em);
        const double slack = 2.0 * bnorm * ASB_EPS * (double)n + 2.0 * pivmin;
        sc[TRI_GL] = gl - slack;
        sc[TRI_GU] = gu + slack;
        sc[TRI_BNORM] = bnorm;
        sc[TRI_PIVMIN] = pivmin;
        sc[TRI_ATOL] = 0.25 * ASB_EPS * bnorm;
    }
}

// Multi-section (round 3): one WAVE per eigenvalue, its 64 lanes count at 64 interior points of the bracket at once, so a
// round narrows it 65-fold: 9-10 rounds of the sequential Sturm recurrence instead of ~53, on 64 times as many waves (one
// thread per eigenvalue left the GPU at 63 waves: 23 ms at n = 4000).  Same counts, same pivmin guard, same tolerance.
__global__ __launch_bounds__(64) void k_tri_multisect(const double* __restrict__ d, const double* __restrict__ e2, int n,
                                                      const double* __restrict__ sc, double* __restrict__ lam_desc) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const double pivmin = sc[TRI_PIVMIN], atol = sc[TRI_ATOL];
    double lo = sc[TRI_GL], hi = sc[TRI_GU];
    for (int it = 0; it < 400; ++it) {
        if (hi - lo <= atol + 2.0 * ASB_EPS * fmax(fabs(lo), fabs(hi))) break;
        const double h = (hi - lo) / 65.0;
        double x = lo + h * (double)(lane + 1);
        if (!(x > lo)) x = lo;                      // (bracket at rounding level: the points collapse onto its ends)
        if (!(x < hi)) x = hi;
        double q = d[0] - x;
        if (fabs(q) < pivmin) q = -pivmin;
        int cnt = q < 0.0 ? 1 : 0;
        for (int j = 1; j < n; ++j) {
            q = d[j] - x - e2[j - 1] / q;
            if (fabs(q) < pivmin) q = -pivmin;
            cnt += q < 0.0 ? 1 : 0;
        }
        // the eigenvalue (i-th smallest, 0-based) lies left of the first point with more than i eigenvalues below it
        const unsigned long long above = __ballot(cnt > i);
        double nlo, nhi;
        if (above == 0ull) {
            nlo = __shfl(x, 63, 64);
            nhi = hi;
        } else {
            const int f = __ffsll((long long)above) - 1;
            nhi = __shfl(x, f, 64);
            nlo = f > 0 ? __shfl(x, f - 1, 64) : lo;
        }
        if (!(nhi - nlo < hi - lo)) break;          // no progress: rounding level
        lo = nlo;
        hi = nhi;
    }
    if (lane == 0) lam_desc[n - 1 - i] = 0.5 * (lo + hi);
}

// thread i -> the (i+1)-th smallest eigenvalue by bisection; count(x) = #{eigenvalues < x} from the signs of the
// Sturm sequence q_0 = d_0 - x, q_j = d_j - x - e_{j-1}^2 / q_{j-1}.  Written in DESCENDING order.
__global__ __launch_bounds__(64) void k_tri_bisect(const double* __restrict__ d, const double* __restrict__ e2, int n,
                                                   const double* __restrict__ sc, double* __restrict__ lam_desc) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const double pivmin = sc[TRI_PIVMIN], atol = sc[TRI_ATOL];
    double lo = sc[TRI_GL], hi = sc[TRI_GU];
    for (int it = 0; it < 1200; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (!(mid > lo && mid < hi)) break;
        double q = d[0] - mid;
        if (fabs(q) < pivmin) q = -pivmin;
        int cnt = q < 0.0 ? 1 : 0;
        for (int j = 1; j < n; ++j) {
            q = d[j] - mid - e2[j - 1] / q;
            if (fabs(q) < pivmin) q = -pivmin;
            cnt += q < 0.0 ? 1 : 0;
        }
        if (cnt > i) hi = mid; else lo = mid;
        if (hi - lo <= atol + 2.0 * ASB_EPS * fmax(fabs(lo), fabs(hi))) break;
    }
    lam_desc[n - 1 - i] = 0.5 * (lo + hi);
}

// shifts of the inverse iteration: the k leading eigenvalues, pushed apart where two of them coincide to rounding
// (identical shifts would give identical vectors)
__global__ void k_tri_shifts(const double* __restrict__ lam_desc, int k, const double* __restrict__ sc, double* __restrict__ sh) {
    if (threadIdx.x || blockIdx.x) return;
    const double floor_ = 1e-3 * ASB_EPS * sc[TRI_BNORM];
    sh[0] = lam_desc[0];
    for (int j = 1; j < k; ++j) {
        const double pert = 10.0 * ASB_EPS * fabs(lam_desc[j]) + floor_;
        sh[j] = (sh[j - 1] - lam_desc[j] < pert) ? sh[j - 1] - pert : lam_desc[j];
    }
}

// Inverse iteration for vector v (thread v): Gaussian elimination with row interchanges of T - shift I (EISPACK TINVIT:
// multipliers rv4, rows of U in rv1 / rv2 / rv3), then back substitution from a constant start vector, at most five
// refinement steps, stop as soon as the iterate has grown to norm >= 1.  No re-orthogonalisation inside clusters: the
// caller only needs the SPAN of the vectors (Rayleigh-Ritz on the snapshot matrix follows) -- see DESIGN.md.
// All per-vector arrays are interleaved: element i of vector v at [i * k + v].
__global__ __launch_bounds__(64) void k_tri_invit(const double* __restrict__ d, const double* __restrict__ e, int n, int k,
                                                  const double* __restrict__ shifts, const double* __restrict__ sc,
                                                  double* __restrict__ work, unsigned char* __restrict__ swp,
                                                  double* __restrict__ Z, int* __restrict__ status) {
    const int v = blockIdx.x * 64 + threadIdx.x;
    if (v >= k) return;
    const size_t nk = (size_t)n * k;
    double* rv1 = work + v;
    double* rv2 = rv1 + nk;
    double* rv3 = rv2 + nk;
    double* rv4 = rv3 + nk;
    double* rv6 = Z + v;
    unsigned char* sw = swp + v;
    const double x1 = shifts[v];
    const double norm = sc[TRI_BNORM];
    const double eps3 = ASB_EPS * (norm > 0.0 ? norm : 1.0);
    const double eps4 = (double)n * eps3;
    const double uk = eps4 / sqrt((double)n);
    if (n == 1) { rv6[0] = 1.0; return; }
    double u = d[0] - x1, vv = e[0];
    for (int i = 1; i < n; ++i) {
        const double ei = e[i - 1];
        const double enext = (i < n - 1) ? e[i] : 0.0;
        const size_t a = (size_t)(i - 1) * k, b = (size_t)i * k;
        if (fabs(ei) >= fabs(u) && ei != 0.0) {      // interchange rows i-1 and i
            const double xu = u / ei;
            rv4[b] = xu;
            rv1[a] = ei;
            const double r2 = d[i] - x1;
            rv2[a] = r2;
            rv3[a] = enext;
            u = vv - xu * r2;
            vv = -xu * enext;
            sw[b] = 1;
        } else {
            if (u == 0.0) u = eps3;             // (only with e[i-1] == 0: the matrix splits here)
            const double xu = ei / u;
            rv4[b] = xu;
            rv1[a] = u;
            rv2[a] = vv;
            rv3[a] = 0.0;
            u = d[i] - x1 - xu * vv;
            vv = enext;
            sw[b] = 0;
        }
    }
    if (u == 0.0) u = eps3;
    {
        const size_t a = (size_t)(n - 1) * k;
        rv1[a] = u; rv2[a] = 0.0; rv3[a] = 0.0;
    }
    for (int i = 0; i < n; ++i) rv6[(size_t)i * k] = uk;
    bool ok = false;
    for (int its = 0; its < 6 && !ok; ++its) {
        double bu = 0.0, bv = 0.0, nrm = 0.0;
        for (int i = n - 1; i >= 0; --i) {      // back substitution with U
            const size_t a = (size_t)i * k;
            const double x = (rv6[a] - bu * rv2[a] - bv * rv3[a]) / rv1[a];
            rv6[a] = x;
            bv = bu;
            bu = x;
            nrm += fabs(x);
        }
        if (nrm >= 1.0) { ok = true; break; }
        if (its == 5) break;
        if (nrm == 0.0 || !(nrm == nrm)) {      // exactly orthogonal start (or overflow): a unit vector instead
            for (int i = 0; i < n; ++i) rv6[(size_t)i * k] = 0.0;
            rv6[(size_t)(its % n) * k] = eps4;
        } else {
            const double s = eps4 / nrm;
            for (int i = 0; i < n; ++i) rv6[(size_t)i * k] *= s;
        }
        for (int i = 1; i < n; ++i) {           // forward elimination of the right-hand side with L (and the interchanges)
            const size_t a = (size_t)(i - 1) * k, b = (size_t)i * k;
            double t = rv6[b];
            if (sw[b]) {
                t = rv6[a];
                rv6[a] = rv6[b];
            }
            rv6[b] = t - rv4[b] * rv6[a];
        }
    }
    double s2 = 0.0;
    for (int i = 0; i < n; ++i) { const double x = rv6[(size_t)i * k]; s2 += x * x; }
    if (!ok || !(s2 > 0.0) || !(s2 < DBL_MAX)) atomicAdd(status, 1);
    const double inv = (s2 > 0.0 && s2 < DBL_MAX) ? 1.0 / sqrt(s2) : 0.0;
    for (int i = 0; i < n; ++i) rv6[(size_t)i * k] *= inv;
}

// d, e (device, n and n-1 entries) -> lam_desc (device, n, descending) and, for k > 0, the k leading unit eigenvectors in
// Z (device, n x k row-major).  *n_bad (optional): vectors whose inverse iteration did not reach the growth criterion.
int asb_tri_eig_dev(asb_ctx* ctx, const double* d, const double* e, int n, int k, double* lam_desc, double* Z, int* n_bad) {
    if (n < 1 || k < 0 || k > n) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_tri_eig_dev: n = %d, k = %d", n, k);
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->tri_work, (size_t)n + 16 + (size_t)k + 4 * (size_t)n * (k > 0 ? k : 0)))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->tri_swp, (size_t)n * (k > 0 ? k : 1)))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->la_status, (size_t)4))) return rc;
    double* e2 = ctx->tri_work;
    double* sc = e2 + n;
    double* sh = sc + 16;
    double* work = sh + k;
    ASB_HIP(ctx, hipMemsetAsync(ctx->la_status, 0, 4 * sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_tri_prepare, dim3(1), dim3(256), 0, ctx->stream, d, e, n, e2, sc);
    static const int multisect = getenv("ASB_TRI_MULTISECT") ? atoi(getenv("ASB_TRI_MULTISECT")) : 1;
    if (multisect) hipLaunchKernelGGL(k_tri_multisect, dim3(n), dim3(64), 0, ctx->stream, d, e2, n, sc, lam_desc);
    else hipLaunchKernelGGL(k_tri_bisect, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, d, e2, n, sc, lam_desc);
    ASB_CHECK_LAUNCH(ctx);
    if (k > 0) {
        hipLaunchKernelGGL(k_tri_shifts, dim3(1), dim3(1), 0, ctx->stream, lam_desc, k, sc, sh);
        hipLaunchKernelGGL(k_tri_invit, dim3((k + 63) / 64), dim3(64), 0, ctx->stream, d, e, n, k, sh, sc, work, ctx->tri_swp, Z,
                           ctx->la_status);
        ASB_CHECK_LAUNCH(ctx);
    }
    if (n_bad) {
        int st[4];
        ASB_HIP(ctx, hipMemcpyAsync(st, ctx->la_status, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        *n_bad = st[0];
    }
    return ASB_OK;
}

// ======================================================================================================================
// 2. one-sided Jacobi on the rows of A (nv x m, row-major, leading dimension lda)
// ======================================================================================================================
// Round r of the circle method on n2 (even) players: block p = 0 pairs (r mod (n2-1), n2-1), block p > 0 pairs
// ((r + p) mod (n2-1), (r - p) mod (n2-1)).  Every player meets every other exactly once per sweep of n2 - 1 rounds and
// the n2 / 2 pairs of a round are disjoint: one launch per round, one block per pair, no two blocks touch the same row.
// Pair (i, j): alpha = |a_i|^2, beta = |a_j|^2, gamma = a_i . a_j; if |gamma| > tol sqrt(alpha beta) the plane rotation
// that makes the two rows orthogonal is applied to rows i, j of A and of Q (accumulated left factor, Q A_0 = A).
__global__ __launch_bounds__(256) void k_jacobi_round(double* __restrict__ A, int m, long long lda, double* __restrict__ Q, int nq,
                                                      int n2, int r, int nv, double tol, const double* __restrict__ fro2,
                                                      unsigned* __restrict__ n_rot) {
    __shared__ double sh[12];
    const int p = blockIdx.x, np = n2 - 1;
    int i = (p == 0) ? (r % np) : ((r + p) % np);
    int j = (p == 0) ? (n2 - 1) : ((r - p + np) % np);
    if (i > j) { const int t = i; i = j; j = t; }
    if (j >= nv) return;                          // the padding player of an odd count
    double* ai = A + (long long)i * lda;
    double* aj = A + (long long)j * lda;
    // rows of up to 4096 entries stay in registers between the three dot products and the rotation (one read, one write)
    constexpr int EPT = 16;
    const bool cached = m <= 256 * EPT;
    double xr[EPT], yr[EPT];
    double v[3] = {0.0, 0.0, 0.0};
    if (cached) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int c = threadIdx.x + 256 * q;
            xr[q] = c < m ? ai[c] : 0.0;
            yr[q] = c < m ? aj[c] : 0.0;
            v[0] += xr[q] * xr[q]; v[1] += yr[q] * yr[q]; v[2] += xr[q] * yr[q];
        }
    } else {
        for (int c = threadIdx.x; c < m; c += 256) {
            const double x = ai[c], y = aj[c];
            v[0] += x * x; v[1] += y * y; v[2] += x * y;
        }
    }
    block_sum<3>(v, sh);
    const double al = v[0], be = v[1], ga = v[2];
    // rows at rounding level of the whole matrix (rank deficiency: more rows than columns, dependent rows) carry no
    // direction of their own: they are left alone, or the sweep over mutually "correlated" noise never ends
    const double floor2 = 1e-28 * fro2[0];
    if (!(al > floor2) || !(be > floor2) || !(fabs(ga) > tol * sqrt(al) * sqrt(be))) return;
    const double zeta = (be - al) / (2.0 * ga);
    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
    if (cached) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int c = threadIdx.x + 256 * q;
            if (c < m) {
                ai[c] = cs * xr[q] - sn * yr[q];
                aj[c] = sn * xr[q] + cs * yr[q];
            }
        }
    } else {
        for (int c = threadIdx.x; c < m; c += 256) {
            const double x = ai[c], y = aj[c];
            ai[c] = cs * x - sn * y;
            aj[c] = sn * x + cs * y;
        }
    }
    if (Q) {
        double* qi = Q + (long long)i * nq;
        double* qj = Q + (long long)j * nq;
        for (int c = threadIdx.x; c < nq; c += 256) {
            const double x = qi[c], y = qj[c];
            qi[c] = cs * x - sn * y;
            qj[c] = sn * x + cs * y;
        }
    }
    if (threadIdx.x == 0) atomicAdd(n_rot, 1u);
}

__global__ __launch_bounds__(256) void k_sumsq(const double* __restrict__ sig, int nv, double* __restrict__ out) {
    __shared__ double sh[4];
    double v[1] = {0.0};
    for (int i = threadIdx.x; i < nv; i += 256) v[0] += sig[i] * sig[i];
    block_sum<1>(v, sh);
    if (threadIdx.x == 0) out[0] = v[0];
}

__global__ __launch_bounds__(256) void k_row_norms(const double* __restrict__ A, int m, long long lda, double* __restrict__ sig) {
    __shared__ double sh[4];
    const double* a = A + (long long)blockIdx.x * lda;
    double v[1] = {0.0};
    for (int c = threadIdx.x; c < m; c += 256) v[0] += a[c] * a[c];
    block_sum<1>(v, sh);
    if (threadIdx.x == 0) sig[blockIdx.x] = sqrt(v[0]);
}

// rank of every row by descending norm (ties: lower index first) -> where[rank] = row
__global__ __launch_bounds__(256) void k_rank_desc(const double* __restrict__ sig, int nv, int* __restrict__ where,
                                                   double* __restrict__ sig_sorted) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nv; i += gridDim.x * 256) {
        const double s = sig[i];
        int rank = 0;
        for (int j = 0; j < nv; ++j) rank += (sig[j] > s || (sig[j] == s && j < i)) ? 1 : 0;
        where[rank] = i;
        sig_sorted[rank] = s;
    }
}

__global__ __launch_bounds__(256) void k_identity(double* __restrict__ Q, int n) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < (long long)n * n; e += (long long)gridDim.x * 256)
        Q[e] = (e / n == e % n) ? 1.0 : 0.0;
}

// out row `rank` = in row where[rank]  (transpose_out: out[c][rank] instead -- eigenvectors as COLUMNS)
__global__ __launch_bounds__(256) void k_gather_rows(const double* __restrict__ in, int ncols, const int* __restrict__ where,
                                                     double* __restrict__ out, int nrows, int transpose_out) {
    const int rk = blockIdx.x;
    const double* src = in + (long long)where[rk] * ncols;
    for (int c = threadIdx.x; c < ncols; c += 256) {
        if (transpose_out) out[(long long)c * nrows + rk] = src[c];
        else out[(long long)rk * ncols + c] = src[c];
    }
}

// A (nv x m, device, overwritten: its rows end up mutually orthogonal, NOT sorted), Q_sorted (nv x nv device, optional):
// row r = the r-th left singular vector of A_0 (q_transposed: column r instead), sig_sorted (nv, device): singular values,
// descending.  *sweeps_out (optional).  Synchronises once per sweep (the rotation count).
int asb_jacobi_rows_dev(asb_ctx* ctx, double* A, int nv, int m, long long lda, double* Q_sorted, int q_transposed,
                        double* sig_sorted, int* sweeps_out) {
    if (nv < 1 || m < 1) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_jacobi_rows_dev: %d x %d", nv, m);
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->jac_q, (size_t)nv * nv))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->jac_sig, (size_t)nv + 1))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->jac_where, (size_t)nv + 4))) return rc;
    unsigned* n_rot = reinterpret_cast<unsigned*>(ctx->jac_where + nv);
    if (Q_sorted) hipLaunchKernelGGL(k_identity, dim3(256), dim3(256), 0, ctx->stream, ctx->jac_q, nv);
    const int n2 = nv + (nv & 1);
    const double tol = 4.0 * ASB_EPS;
    double* fro2 = ctx->jac_sig + nv;            // |A|_F^2 of the input (rotations preserve it)
    hipLaunchKernelGGL(k_row_norms, dim3(nv), dim3(256), 0, ctx->stream, A, m, lda, ctx->jac_sig);
    hipLaunchKernelGGL(k_sumsq, dim3(1), dim3(256), 0, ctx->stream, ctx->jac_sig, nv, fro2);
    int sweep = 0;
    bool converged = nv == 1;
    { int rcp = asb_pin_alloc(ctx); if (rcp) return rcp; }
    for (; sweep < 60 && !converged; ++sweep) {
        ASB_HIP(ctx, hipMemsetAsync(n_rot, 0, sizeof(unsigned), ctx->stream));
        for (int r = 0; r < n2 - 1; ++r)
            hipLaunchKernelGGL(k_jacobi_round, dim3(n2 / 2), dim3(256), 0, ctx->stream, A, m, lda, Q_sorted ? ctx->jac_q : (double*)nullptr,
                               nv, n2, r, nv, tol, fro2, n_rot);
        ASB_CHECK_LAUNCH(ctx);
        ASB_HIP(ctx, hipMemcpyAsync(ctx->host_pin + 384, n_rot, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        unsigned h;
        memcpy(&h, ctx->host_pin + 384, sizeof(h));
        converged = h == 0;
        if (getenv("ASB_DEBUG_JACOBI")) fprintf(stderr, "[asb] one-sided Jacobi %d x %d: sweep %d, %u rotations\n", nv, m, sweep, h);
    }
    if (!converged) ASB_FAIL(ctx, ASB_ERR_NUMERIC, "one-sided Jacobi: no convergence in %d sweeps (%d x %d)", sweep, nv, m);
    if (sweeps_out) *sweeps_out = sweep;
    hipLaunchKernelGGL(k_row_norms, dim3(nv), dim3(256), 0, ctx->stream, A, m, lda, ctx->jac_sig);
    hipLaunchKernelGGL(k_rank_desc, dim3((nv + 255) / 256), dim3(256), 0, ctx->stream, ctx->jac_sig, nv, ctx->jac_where, sig_sorted);
    if (Q_sorted)
        hipLaunchKernelGGL(k_gather_rows, dim3(nv), dim3(256), 0, ctx->stream, ctx->jac_q, nv, ctx->jac_where, Q_sorted, nv, q_transposed);
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

// eigen-decomposition of a symmetric POSITIVE SEMI-DEFINITE n x n matrix (a Gram matrix) of any size: its rows are
// rotated until orthogonal, Q A = diag(lam) P with P orthonormal, so lam are the eigenvalues and the rows of Q the
// eigenvectors.  lam (n, descending), V (n x n, eigenvectors as columns) -- the contract of asb_sym_eig.
int asb_sym_eig_large(asb_ctx* ctx, const double* A_dev, int n, double* lam_dev, double* V_dev) {
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->jac_a, (size_t)n * n))) return rc;
    ASB_HIP(ctx, hipMemcpyAsync(ctx->jac_a, A_dev, (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return asb_jacobi_rows_dev(ctx, ctx->jac_a, n, n, n, V_dev, 1, lam_dev, nullptr);
}

// ======================================================================================================================
// 3. blocked Cholesky + inverse of the factor, any K
// ======================================================================================================================
#define CB 32

// diagonal block jb: L11 = chol(G11) in LDS, written back (upper part zeroed), and its inverse to Dinv[jb / CB]
__global__ __launch_bounds__(1024) void k_chol_diag(double* __restrict__ Lw, int Kp, int jb, double* __restrict__ Dinv,
                                                    int* __restrict__ status) {
    __shared__ double L[CB][CB + 1];
    __shared__ double T[CB][CB + 1];
    const int r = threadIdx.x / CB, c = threadIdx.x % CB;
    L[r][c] = Lw[(long long)(jb + r) * Kp + jb + c];
    __syncthreads();
    for (int j = 0; j < CB; ++j) {
        const double dj = L[j][j];
        __syncthreads();
        if (!(dj > 0.0)) {
            if (threadIdx.x == 0) status[0] = 1;
            return;
        }
        const double sq = sqrt(dj);
        if (c == j && r >= j) L[r][j] = (r == j) ? sq : L[r][j] / sq;
        __syncthreads();
        if (r > j && c > j && c <= r) L[r][c] -= L[r][j] * L[c][j];
        __syncthreads();
    }
    // column c of L^-1 by forward substitution (32 threads)
    if (r == 0) {
        for (int i = 0; i < CB; ++i) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int j = c; j < i; ++j) s -= L[i][j] * T[j][c];
            T[i][c] = (i < c) ? 0.0 : s / L[i][i];
        }
    }
    __syncthreads();
    Lw[(long long)(jb + r) * Kp + jb + c] = (c <= r) ? L[r][c] : 0.0;
    Dinv[(long long)(jb / CB) * CB * CB + r * CB + c] = T[r][c];
}

// panel below the diagonal block: L21 = G21 L11^-T, i.e. L21[i][c] = sum_t G21[i][t] (L11^-1)[c][t]; 8 rows per block
__global__ __launch_bounds__(256) void k_chol_panel(double* __restrict__ Lw, int Kp, int jb, const double* __restrict__ Dinv) {
    __shared__ double T[CB][CB + 1];
    __shared__ double G[8][CB + 1];
    const int r = threadIdx.x / CB, c = threadIdx.x % CB;
    const double* Di = Dinv + (long long)(jb / CB) * CB * CB;
    for (int q = threadIdx.x; q < CB * CB; q += 256) T[q / CB][q % CB] = Di[q];
    const int i = jb + CB + blockIdx.x * 8 + r;
    const bool on = i < Kp;
    G[r][c] = on ? Lw[(long long)i * Kp + jb + c] : 0.0;
    __syncthreads();
    double s = 0.0;
#pragma unroll 8
    for (int t = 0; t < CB; ++t) s += G[r][t] * T[c][t];
    if (on) Lw[(long long)i * Kp + jb + c] = s;
}

// trailing block (rows, columns >= jb + CB), lower triangle by 32 x 32 tiles: G22[i][c] -= sum_t L21[i][t] L21[c][t]
__global__ __launch_bounds__(1024) void k_chol_trail(double* __restrict__ Lw, int Kp, int jb) {
    __shared__ double Ar[CB][CB + 1];
    __shared__ double Ac[CB][CB + 1];
    // tile index -> (bi, bc) with bc <= bi
    int tile = blockIdx.x, bi = 0;
    while (tile > bi) { tile -= bi + 1; ++bi; }
    const int bc = tile;
    const int r = threadIdx.x / CB, c = threadIdx.x % CB;
    const int i0 = jb + CB + bi * CB, c0 = jb + CB + bc * CB;
    Ar[r][c] = Lw[(long long)(i0 + r) * Kp + jb + c];
    Ac[r][c] = Lw[(long long)(c0 + r) * Kp + jb + c];
    __syncthreads();
    double s = 0.0;
#pragma unroll 8
    for (int t = 0; t < CB; ++t) s += Ar[r][t] * Ac[c][t];
    if (c0 + c <= i0 + r) Lw[(long long)(i0 + r) * Kp + c0 + c] -= s;
}

// T = L^-1 by block columns (one block of 32 x 32 threads per block column jc): T[jc][jc] = Dinv[jc];
// T[ib][jc] = -Dinv[ib] sum_{t = jc}^{ib-1} L[ib][t] T[t][jc].  Written TRANSPOSED and un-padded: Tt[c][i] = T[i][c].
__global__ __launch_bounds__(1024) void k_chol_inverse(const double* __restrict__ Lw, int Kp, int K, const double* __restrict__ Dinv,
                                                       double* __restrict__ Tw, double* __restrict__ Tt) {
    __shared__ double S[CB][CB + 1];
    const int jc = blockIdx.x, nb = Kp / CB;
    const int r = threadIdx.x / CB, c = threadIdx.x % CB;
    // Tw: padded work copy (Kp x Kp, row-major) holding T's block column jc
    Tw[(long long)(jc * CB + r) * Kp + jc * CB + c] = Dinv[(long long)jc * CB * CB + r * CB + c];
    __syncthreads();
    for (int ib = jc + 1; ib < nb; ++ib) {
        double s = 0.0;
        for (int t = jc; t < ib; ++t) {
            const double* Lrow = Lw + (long long)(ib * CB + r) * Kp + t * CB;
            const double* Tcol = Tw + (long long)(t * CB) * Kp + jc * CB + c;
#pragma unroll 8
            for (int q = 0; q < CB; ++q) s += Lrow[q] * Tcol[(long long)q * Kp];
        }
        S[r][c] = s;
        __syncthreads();
        const double* Di = Dinv + (long long)ib * CB * CB;
        double o = 0.0;
#pragma unroll 8
        for (int q = 0; q < CB; ++q) o -= Di[r * CB + q] * S[q][c];
        __syncthreads();
        Tw[(long long)(ib * CB + r) * Kp + jc * CB + c] = o;
        __threadfence_block();
        __syncthreads();
    }
    // transposed, un-padded output of this block column (rows above the diagonal block are zero)
    const int col = jc * CB + c;
    for (int ib = 0; ib < nb; ++ib) {
        const int row = ib * CB + r;
        if (row < K && col < K) Tt[(long long)col * K + row] = (ib < jc) ? 0.0 : Tw[(long long)row * Kp + col];
    }
}

__global__ __launch_bounds__(256) void k_chol_load(const double* __restrict__ G, int K, int Kp, double* __restrict__ Lw) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < (long long)Kp * Kp; e += (long long)gridDim.x * 256) {
        const int i = (int)(e / Kp), j = (int)(e % Kp);
        Lw[e] = (i < K && j < K) ? 0.5 * (G[(long long)i * K + j] + G[(long long)j * K + i]) : (i == j ? 1.0 : 0.0);
    }
}

// G (K x K SPD, device) -> Tt (K x K, device) with Tt[c][i] = (L^-1)[i][c], G = L L^T: Q = A . Tt is the Q of the
// economic QR of A (CholeskyQR).  status[0] (device int) is set to 1 when a pivot is not positive.
int asb_chol_tinv_dev(asb_ctx* ctx, const double* G, int K, double* Tt, int* status_dev) {
    const int Kp = (K + CB - 1) / CB * CB, nb = Kp / CB;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->chol_w, (size_t)2 * Kp * Kp + (size_t)nb * CB * CB))) return rc;
    double* Lw = ctx->chol_w;
    double* Tw = Lw + (size_t)Kp * Kp;
    double* Dinv = Tw + (size_t)Kp * Kp;
    hipLaunchKernelGGL(k_chol_load, dim3(256), dim3(256), 0, ctx->stream, G, K, Kp, Lw);
    for (int b = 0; b < nb; ++b) {
        const int jb = b * CB, rest = Kp - jb - CB;
        hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(1024), 0, ctx->stream, Lw, Kp, jb, Dinv, status_dev);
        if (rest > 0) {
            hipLaunchKernelGGL(k_chol_panel, dim3((rest + 7) / 8), dim3(256), 0, ctx->stream, Lw, Kp, jb, Dinv);
            const int tb = rest / CB;
            hipLaunchKernelGGL(k_chol_trail, dim3(tb * (tb + 1) / 2), dim3(1024), 0, ctx->stream, Lw, Kp, jb);
        }
    }
    hipLaunchKernelGGL(k_chol_inverse, dim3(nb), dim3(1024), 0, ctx->stream, Lw, Kp, K, Dinv, Tw, Tt);
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

// ======================================================================================================================
// host-side probes of the three solvers (tests; host arrays in and out)
// ======================================================================================================================
extern "C" int asb_test_tridiag_eig(asb_ctx* ctx, const double* d, const double* e, int64_t n, int64_t k, double* lam_desc,
                                    double* Z, int64_t* n_bad) {
    if (!ctx || !d || !lam_desc || n < 1 || k < 0 || k > n || (n > 1 && !e) || (k > 0 && !Z)) return ASB_ERR_ARG;
    double *dd = nullptr, *dl = nullptr, *dz = nullptr;
    ASB_HIP(ctx, hipMalloc((void**)&dd, (size_t)2 * n * sizeof(double)));
    ASB_HIP(ctx, hipMalloc((void**)&dl, (size_t)n * sizeof(double)));
    ASB_HIP(ctx, hipMalloc((void**)&dz, (size_t)(k > 0 ? n * k : 1) * sizeof(double)));
    ASB_HIP(ctx, hipMemcpyAsync(dd, d, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (n > 1) ASB_HIP(ctx, hipMemcpyAsync(dd + n, e, (size_t)(n - 1) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    int bad = 0;
    int rc = asb_tri_eig_dev(ctx, dd, dd + n, (int)n, (int)k, dl, dz, &bad);
    if (rc == ASB_OK) {
        (void)hipMemcpyAsync(lam_desc, dl, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        if (k > 0) (void)hipMemcpyAsync(Z, dz, (size_t)n * k * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (n_bad) *n_bad = bad;
    (void)hipFree(dd); (void)hipFree(dl); (void)hipFree(dz);
    return rc;
}

extern "C" int asb_test_jacobi_rows(asb_ctx* ctx, const double* A, int64_t nv, int64_t m, double* U /* nv x nv, columns */,
                                    double* sig, int64_t* sweeps) {
    if (!ctx || !A || !sig || nv < 1 || m < 1) return ASB_ERR_ARG;
    double *da = nullptr, *dq = nullptr, *ds = nullptr;
    ASB_HIP(ctx, hipMalloc((void**)&da, (size_t)nv * m * sizeof(double)));
    ASB_HIP(ctx, hipMalloc((void**)&dq, (size_t)nv * nv * sizeof(double)));
    ASB_HIP(ctx, hipMalloc((void**)&ds, (size_t)nv * sizeof(double)));
    ASB_HIP(ctx, hipMemcpyAsync(da, A, (size_t)nv * m * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    int sw = 0;
    int rc = asb_jacobi_rows_dev(ctx, da, (int)nv, (int)m, m, U ? dq : nullptr, 1, ds, &sw);
    if (rc == ASB_OK) {
        (void)hipMemcpyAsync(sig, ds, (size_t)nv * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        if (U) (void)hipMemcpyAsync(U, dq, (size_t)nv * nv * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (sweeps) *sweeps = sw;
    (void)hipFree(da); (void)hipFree(dq); (void)hipFree(ds);
    return rc;
}

extern "C" int asb_test_chol_tinv(asb_ctx* ctx, const double* G, int64_t K, double* Tt) {
    if (!ctx || !G || !Tt || K < 1) return ASB_ERR_ARG;
    double *dg = nullptr, *dt = nullptr;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->la_status, (size_t)4))) return rc;
    ASB_HIP(ctx, hipMalloc((void**)&dg, (size_t)K * K * sizeof(double)));
    ASB_HIP(ctx, hipMalloc((void**)&dt, (size_t)K * K * sizeof(double)));
    ASB_HIP(ctx, hipMemcpyAsync(dg, G, (size_t)K * K * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    ASB_HIP(ctx, hipMemsetAsync(ctx->la_status, 0, 4 * sizeof(int), ctx->stream));
    rc = asb_chol_tinv_dev(ctx, dg, (int)K, dt, ctx->la_status);
    int st[4] = {0, 0, 0, 0};
    if (rc == ASB_OK) {
        (void)hipMemcpyAsync(Tt, dt, (size_t)K * K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        (void)hipMemcpyAsync(st, ctx->la_status, sizeof(st), hipMemcpyDeviceToHost, ctx->stream);
        (void)hipStreamSynchronize(ctx->stream);
    }
    (void)hipFree(dg); (void)hipFree(dt);
    if (rc == ASB_OK && st[0]) ASB_FAIL(ctx, ASB_ERR_NUMERIC, "Cholesky: the matrix is not positive definite");
    return rc;
}
