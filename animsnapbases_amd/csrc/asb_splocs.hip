// SPLOCS global optimisation on the GPU -- posComponents.splocs_glob_optimization,
// snapbases/posComponents.py:132-189 of the reference.  gfx950 (MI355X) only.
//
// The reference sweeps the F x 3N residual ~5K times per outer iteration (rank-1 updates per
// component).  Here the residual is never formed.  With C fixed during the weight sweep,
//     P = X C^T (F x K),  M = C C^T (K x K)           -- ONE pass over X (f64 MFMA)
//     opt_k = (P[:,k] - W M[:,k]) / M[k,k] + W[:,k]    -- O(F K) per component, sequential in k
// (identical to  Rflat += w_k c_k^T; opt = Rflat c_k / |c_k|^2  of :153-154 because Rflat = X - W C
// throughout), the ADMM right-hand side c = W^T X is the same 16-column projection kernel as the
// deflation, the K x K solve is an explicit (G + rho I)^-1 applied with MFMA, and
//     |X - W C|^2 = |X|^2 - 2 <W, X C^T> + <W^T W, C C^T>
// needs only the next iteration's P and M.  Support maps (Lambda) come from the host geodesics.
#include "asb_kernels.h"

#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));

// defined in asb_project.hip
int asb_project_columns_wide(asb_ctx* ctx, const double* Wfk, int64_t ldw, int64_t k0, int ncols, double* out_rows, const double* col_scale);
int asb_project_columns(asb_ctx* ctx, const double* Wfk, int64_t ldw, int64_t k0, int ncols, double* out_rows,
                        const double* col_scale);

// --------------------------------------------------------------------------------------
// k_bcd: the block-coordinate-descent sweep over the K weight columns (:144-156), one block.
// W (F x K) row-major is updated in place;  opt = (P[:,k] - W M[:,k]) / M[k,k] + W[:,k];
// W[:,k] = project_weight(opt)  (clamp at 0, divide by the max unless it is 0).
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_bcd(double* __restrict__ W, const double* __restrict__ P,
                                              const double* __restrict__ M, int F, int K, const int* __restrict__ only_if) {
    extern __shared__ double sm[];       // K (column of M) + 16 (wave maxima)
    if (only_if && !only_if[0]) return;  // fallback of k_bcd_wide: runs only when that one gave up
    double* mcol = sm;
    double* wmax = sm + K;
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int k = 0; k < K; ++k) {
        const double nk = M[(long long)k * K + k];
        if (nk <= 1.e-8) {               // component is zero everywhere: zero activation (:147-150)
            for (int f = tid; f < F; f += nt) W[(long long)f * K + k] = 0.0;
            __syncthreads();
            continue;
        }
        for (int j = tid; j < K; j += nt) mcol[j] = M[(long long)j * K + k];
        __syncthreads();
        double mx = 0.0;
        for (int f = tid; f < F; f += nt) {
            const double* wr = W + (long long)f * K;
            double s = 0.0;
            for (int j = 0; j < K; ++j) s += wr[j] * mcol[j];
            double opt = (P[(long long)f * K + k] - s) / nk + wr[k];
            opt = fmax(0.0, opt);
            W[(long long)f * K + k] = opt;            // own row only: no cross-thread hazard
            mx = fmax(mx, opt);
        }
        mx = wave_max(mx);
        if ((tid & 63) == 0) wmax[tid >> 6] = mx;
        __syncthreads();
        mx = 0.0;
        for (int q = 0; q < (nt >> 6); ++q) mx = fmax(mx, wmax[q]);
        if (mx != 0.0)
            for (int f = tid; f < F; f += nt) W[(long long)f * K + k] /= mx;
        __syncthreads();
    }
}

// k_bcd_wide: the same sweep spread over co-resident blocks, one weight row per thread held in LDS (column-major:
// sw[j * nt + tid]), so the K dot products of a row never leave the CU.  The only coupling between rows is the column
// maximum of project_weight: block b publishes its maximum of column k in slot[k * nblk + b] (armed to -1 by the host;
// a maximum is >= 0) and every block polls the nblk slots of the column -- one exchange per column instead of the
// single block's K passes through L2.  A poll that does not complete (blocks not co-resident) raises flag[0]; W is only
// written at the very end, so the single-block kernel can redo the sweep from the untouched input in that case.
__global__ __launch_bounds__(256) void k_bcd_wide(double* __restrict__ W, const double* __restrict__ P, const double* __restrict__ M,
                                                  int F, int K, double* slot, int* flag) {
    extern __shared__ double sm[];       // nt * K (rows) + K (column of M) + 4 (wave maxima)
    const int tid = threadIdx.x, nt = blockDim.x, nblk = gridDim.x;
    double* sw = sm;
    double* mcol = sm + (size_t)nt * K;
    double* wmax = mcol + K;
    const int f = blockIdx.x * nt + tid;
    const bool live = f < F;
    for (int j = 0; j < K; ++j) sw[(size_t)j * nt + tid] = live ? W[(long long)f * K + j] : 0.0;
    for (int k = 0; k < K; ++k) {
        const double nk = M[(long long)k * K + k];
        if (nk <= 1.e-8) {               // (:147-150)
            sw[(size_t)k * nt + tid] = 0.0;
            continue;
        }
        __syncthreads();                 // the previous column's readers of mcol / wmax are done
        for (int j = tid; j < K; j += nt) mcol[j] = M[(long long)j * K + k];
        __syncthreads();
        double s = 0.0;
        for (int j = 0; j < K; ++j) s += sw[(size_t)j * nt + tid] * mcol[j];
        double opt = 0.0;
        if (live) opt = fmax(0.0, (P[(long long)f * K + k] - s) / nk + sw[(size_t)k * nt + tid]);
        double mx = wave_max(opt);
        if ((tid & 63) == 0) wmax[tid >> 6] = mx;
        __syncthreads();
        if (tid == 0) {
            mx = 0.0;
            for (int q = 0; q < (nt >> 6); ++q) mx = fmax(mx, wmax[q]);
            __hip_atomic_store(slot + (long long)k * nblk + blockIdx.x, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // all blocks' maxima of this column: thread q polls blocks q, q + nt, ...
        mx = 0.0;
        for (int q = tid; q < nblk; q += nt) {
            double v = -1.0;
            for (int spin = 0; spin < (1 << 20); ++spin) {
                v = __hip_atomic_load(slot + (long long)k * nblk + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v >= 0.0) break;
                if ((spin & 1023) == 1023 && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (!(v >= 0.0)) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v = 0.0; }
            mx = fmax(mx, v);
        }
        mx = wave_max(mx);
        __syncthreads();                 // wmax was read by thread 0 above
        if ((tid & 63) == 0) wmax[tid >> 6] = mx;
        __syncthreads();
        mx = 0.0;
        for (int q = 0; q < (nt >> 6); ++q) mx = fmax(mx, wmax[q]);
        sw[(size_t)k * nt + tid] = (mx != 0.0) ? opt / mx : opt;
    }
    __syncthreads();
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    // (a block that gets here after some column got stuck has either timed out on that column itself or seen the flag,
    // so nobody writes once the sweep is abandoned)
    if (live)
        for (int j = 0; j < K; ++j) W[(long long)f * K + j] = sw[(size_t)j * nt + tid];
}
__global__ void k_bcd_arm(double* slot, long long n, int* flag) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) slot[i] = -1.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) flag[0] = 0;
}

// per component: vertex with the largest |C_k[v]|^2 (first on ties) (:161)
__global__ __launch_bounds__(256) void k_centres(const double* __restrict__ C, long long n_vert, long long v0,
                                                 long long* __restrict__ idx_out, double* __restrict__ val_out) {
    __shared__ double sh_d[256];
    __shared__ long long sh_i[256];
    const double* c = C + (long long)blockIdx.x * n_vert * 3;
    double be = -1.0;
    long long bi = 0x7fffffffffffffffLL;
    for (long long v = threadIdx.x; v < n_vert; v += blockDim.x) {
        const double e = c[3 * v] * c[3 * v] + c[3 * v + 1] * c[3 * v + 1] + c[3 * v + 2] * c[3 * v + 2];
        if (am_better(e, v, be, bi)) { be = e; bi = v; }
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
    if (threadIdx.x == 0) { idx_out[blockIdx.x] = v0 + sh_i[0]; val_out[blockIdx.x] = sh_d[0]; }
}


// out (np x np) = [G + rho I, 0; 0, I]
__global__ __launch_bounds__(256) void k_pad_spd(const double* __restrict__ G, double rho, int K, int np, double* __restrict__ out) {
    const long long total = (long long)np * np;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int i = (int)(e / np), j = (int)(e % np);
        out[e] = (i < K && j < K) ? G[(long long)i * K + j] + (i == j ? rho : 0.0) : (i == j ? 1.0 : 0.0);
    }
}

// rhs = c + rho (Z - U)     (:176)
__global__ __launch_bounds__(256) void k_admm_rhs(const double* __restrict__ c, const double* __restrict__ Z,
                                                  const double* __restrict__ U, double rho, long long n,
                                                  double* __restrict__ rhs) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
        rhs[e] = c[e] + rho * (Z[e] - U[e]);
}

// Z = prox_l1l2(Lambda, C + U, 1/rho);  U = U + C - Z     (:177-178, :252-256), one thread per (k, v)
__global__ __launch_bounds__(256) void k_admm_prox(const double* __restrict__ C, double* __restrict__ Z,
                                                   double* __restrict__ U, const double* __restrict__ Lambda,
                                                   double beta, long long kn) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < kn; e += (long long)gridDim.x * blockDim.x) {
        const double c0 = C[3 * e], c1 = C[3 * e + 1], c2 = C[3 * e + 2];
        const double u0 = U[3 * e], u1 = U[3 * e + 1], u2 = U[3 * e + 2];
        const double x0 = c0 + u0, x1 = c1 + u1, x2 = c2 + u2;
        const double len = sqrt(x0 * x0 + x1 * x1 + x2 * x2);
        const double shrink = fmax(0.0, 1.0 - beta * Lambda[e] / len);      // len == 0 -> -inf -> 0
        const double z0 = x0 * shrink, z1 = x1 * shrink, z2 = x2 * shrink;
        Z[3 * e] = z0; Z[3 * e + 1] = z1; Z[3 * e + 2] = z2;
        U[3 * e] = u0 + c0 - z0; U[3 * e + 1] = u1 + c1 - z1; U[3 * e + 2] = u2 + c2 - z2;
    }
}

// The whole inner ADMM loop in ONE launch for K <= 64 (round 3).  Its three steps -- rhs = c + rho (Z - U), C = (G + rho I)^-1 rhs,
// Z = prox(C + U), U += C - Z (:171-178) -- couple nothing but the K components of ONE vertex's three columns, so a block takes 16
// vertices (48 columns: 384 contiguous bytes per component row) through all iterations: Z, U, c in registers (thread = component
// row k, four vertices), the K x 48 right-hand side and result through LDS, the K x K product on the f64 matrix cores (wave w: rows
// 16 w .. 16 w + 15 of the inverse as its A operand, in registers for the whole launch).  20 x 10 x (16 + 64 + 17) us of launches
// per config-3 run become 20 launches.
__global__ __launch_bounds__(256) void k_admm_fused(const double* __restrict__ c, double* __restrict__ Z, double* __restrict__ U,
                                                    const double* __restrict__ Ginv, const double* __restrict__ Lambda, double rho,
                                                    int K, long long n_vert, int n_iter) {
    __shared__ double rhs_s[64][48 + 1];
    __shared__ double c_s[64][48 + 1];
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const long long n3 = 3 * n_vert, v0 = (long long)blockIdx.x * 16;
    // MFMA A operand: Ginv[16 w + (l & 15)][4 s + (l >> 4)], s = 0 .. 15 (rows / columns beyond K are zero)
    double ga[16];
#pragma unroll
    for (int sidx = 0; sidx < 16; ++sidx) {
        const int r = 16 * w + (l & 15), cidx = 4 * sidx + (l >> 4);
        ga[sidx] = (r < K && cidx < K) ? Ginv[(long long)r * K + cidx] : 0.0;
    }
    // state: component row k = tid / 4, vertices v0 + 4 (tid % 4) .. + 3 (12 contiguous columns)
    const int k = tid >> 2, vq = (tid & 3) * 4;
    double zc[12], uc[12], cc[12], lam[4];
    const bool krow = k < K;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long long v = v0 + vq + q;
        const bool on = krow && v < n_vert;
        lam[q] = on ? Lambda[(long long)k * n_vert + v] : 0.0;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const long long e = (long long)k * n3 + 3 * v + d;
            zc[3 * q + d] = on ? Z[e] : 0.0;
            uc[3 * q + d] = on ? U[e] : 0.0;
            cc[3 * q + d] = on ? c[e] : 0.0;
        }
    }
    const double beta = 1.0 / rho;
    for (int it = 0; it < n_iter; ++it) {
#pragma unroll
        for (int j = 0; j < 12; ++j) rhs_s[k][3 * vq + j] = cc[j] + rho * (zc[j] - uc[j]);
        __syncthreads();
        d4 acc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int sidx = 0; sidx < 16; ++sidx) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double b = rhs_s[4 * sidx + (l >> 4)][16 * j + (l & 15)];
                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[sidx], b, acc[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) c_s[16 * w + (l >> 4) + 4 * q][16 * j + (l & 15)] = acc[j][q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double c0 = c_s[k][3 * (vq + q)], c1 = c_s[k][3 * (vq + q) + 1], c2 = c_s[k][3 * (vq + q) + 2];
            const double u0 = uc[3 * q], u1 = uc[3 * q + 1], u2 = uc[3 * q + 2];
            const double x0 = c0 + u0, x1 = c1 + u1, x2 = c2 + u2;
            const double len = sqrt(x0 * x0 + x1 * x1 + x2 * x2);
            const double shrink = fmax(0.0, 1.0 - beta * lam[q] / len);      // len == 0 -> -inf -> 0
            const double z0 = x0 * shrink, z1 = x1 * shrink, z2 = x2 * shrink;
            zc[3 * q] = z0; zc[3 * q + 1] = z1; zc[3 * q + 2] = z2;
            uc[3 * q] = u0 + c0 - z0; uc[3 * q + 1] = u1 + c1 - z1; uc[3 * q + 2] = u2 + c2 - z2;
        }
        // (the next iteration's first writes go to rhs_s, which nobody reads any more; c_s is rewritten behind the next barrier)
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long long v = v0 + vq + q;
        if (krow && v < n_vert) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const long long e = (long long)k * n3 + 3 * v + d;
                Z[e] = zc[3 * q + d];
                U[e] = uc[3 * q + d];
            }
        }
    }
}

// block partials of sum a[e]*b[e]  (b == nullptr: sum Lambda-weighted group norms, see host)
__global__ __launch_bounds__(256) void k_dot_part(const double* __restrict__ a, const double* __restrict__ b,
                                                  long long n, double* __restrict__ part) {
    __shared__ double sh[4];
    double v[1] = {0.0};
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
        v[0] += a[e] * b[e];
    block_sum<1>(v, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = v[0];
}

// block partials of sum_{k,v} Lambda[k,v] * |C[k,v,:]|     (:184)
__global__ __launch_bounds__(256) void k_sparsity_part(const double* __restrict__ C, const double* __restrict__ Lambda,
                                                       long long kn, double* __restrict__ part) {
    __shared__ double sh[4];
    double v[1] = {0.0};
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < kn; e += (long long)gridDim.x * blockDim.x)
        v[0] += Lambda[e] * sqrt(C[3 * e] * C[3 * e] + C[3 * e + 1] * C[3 * e + 1] + C[3 * e + 2] * C[3 * e + 2]);
    block_sum<1>(v, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = v[0];
}

__global__ __launch_bounds__(256) void k_sum1(const double* __restrict__ in, int n, double* __restrict__ out) {
    __shared__ double sh[4];
    double v[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += blockDim.x) v[0] += in[i];
    block_sum<1>(v, sh);
    if (threadIdx.x == 0) out[0] = v[0];
}

// W (K x Fp) component-major -> Wfk (F x K) frame-major
__global__ __launch_bounds__(256) void k_w_to_fk(const double* __restrict__ W, int Fp, int F, int K,
                                                 double* __restrict__ Wfk) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < (long long)F * K;
         e += (long long)gridDim.x * blockDim.x) {
        const int f = (int)(e / K), k = (int)(e % K);
        Wfk[e] = W[(long long)k * Fp + f];
    }
}

// --------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------
struct asb_splocs {
    int64_t K = 0;
    double *C = nullptr, *Z = nullptr, *U = nullptr, *c = nullptr, *rhs = nullptr, *Lambda = nullptr, *Ct = nullptr;
    double *Wfk = nullptr, *P = nullptr, *M = nullptr, *G = nullptr, *Ginv = nullptr, *red = nullptr;
    double *cen_val = nullptr, *bcd_slot = nullptr;
    const double** field_ptr = nullptr;
    long long* cen_idx = nullptr;
    int* status = nullptr;
    double* trace = nullptr;          // (its, 3): <W, P>, <G, M>, sum Lambda |C_v| of every outer iteration, read once at the end
    int64_t trace_cap = 0;
    bool defer_status = false;        // the ADMM's status word is looked at with the trace, not after every outer iteration
    double* pin = nullptr;            // 4096 doubles of pinned host memory for the centres' read-back
    ~asb_splocs() { if (pin) (void)hipHostFree(pin); }
};

static int dot_to_dev(asb_ctx* ctx, asb_splocs* s, const double* a, const double* b, long long n, double* out_dev) {
    const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_dot_part, dim3(grid), dim3(256), 0, ctx->stream, a, b, n, s->red);
    hipLaunchKernelGGL(k_sum1, dim3(1), dim3(256), 0, ctx->stream, s->red, grid, out_dev);
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}
static int dot_to_host(asb_ctx* ctx, asb_splocs* s, const double* a, const double* b, long long n, double* out) {
    const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_dot_part, dim3(grid), dim3(256), 0, ctx->stream, a, b, n, s->red);
    hipLaunchKernelGGL(k_sum1, dim3(1), dim3(256), 0, ctx->stream, s->red, grid, s->red + 1024);
    ASB_CHECK_LAUNCH(ctx);
    ASB_HIP(ctx, hipMemcpyAsync(out, s->red + 1024, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ASB_OK;
}

extern "C" int asb_splocs_begin(asb_ctx* ctx) {
    if (!ctx || !ctx->comps || !ctx->W) return ASB_ERR_ARG;
    if (!ctx->splocs) ctx->splocs = new asb_splocs();
    asb_splocs* s = ctx->splocs;
    const int64_t K = ctx->K, n3 = 3 * ctx->n_loc, F = ctx->F;
    s->K = K;
    s->defer_status = false;
    int rc;
    if ((rc = asb_alloc(ctx, &s->C, (size_t)K * n3))) return rc;
    if ((rc = asb_alloc(ctx, &s->Z, (size_t)K * n3))) return rc;
    if ((rc = asb_alloc(ctx, &s->U, (size_t)K * n3))) return rc;
    if ((rc = asb_alloc(ctx, &s->c, (size_t)K * n3))) return rc;
    if ((rc = asb_alloc(ctx, &s->rhs, (size_t)K * n3))) return rc;
    if ((rc = asb_alloc(ctx, &s->Ct, (size_t)K * n3))) return rc;
    if ((rc = asb_alloc(ctx, &s->Lambda, (size_t)K * ctx->n_loc))) return rc;
    if ((rc = asb_alloc(ctx, &s->Wfk, (size_t)F * K))) return rc;
    if ((rc = asb_alloc(ctx, &s->P, (size_t)F * K))) return rc;
    if ((rc = asb_alloc(ctx, &s->M, (size_t)K * K))) return rc;
    if ((rc = asb_alloc(ctx, &s->G, (size_t)K * K))) return rc;
    if ((rc = asb_alloc(ctx, &s->Ginv, (size_t)K * K))) return rc;
    if ((rc = asb_alloc(ctx, &s->red, (size_t)2048))) return rc;
    if ((rc = asb_alloc(ctx, &s->cen_val, (size_t)K))) return rc;
    if ((rc = asb_alloc(ctx, &s->cen_idx, (size_t)K))) return rc;
    if ((rc = asb_alloc(ctx, &s->status, (size_t)4))) return rc;
    // C = comps.copy(); W = weigs.copy(); U = 0     (:136-138)
    ASB_HIP(ctx, hipMemcpyAsync(s->C, ctx->comps, (size_t)K * n3 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    ASB_HIP(ctx, hipMemsetAsync(s->U, 0, (size_t)K * n3 * sizeof(double), ctx->stream));
    hipLaunchKernelGGL(k_w_to_fk, dim3(256), dim3(256), 0, ctx->stream, ctx->W, (int)ctx->Fp, (int)F, (int)K, s->Wfk);
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

// local Gram pieces of the current C:  P = X C^T (F x K),  M = C C^T (K x K), written to the
// caller's device buffers (to be all-reduced over ranks) or, when NULL, kept inside the context;
// normX2_local (optional): |X|^2 of the shard.
extern "C" int asb_splocs_gram(asb_ctx* ctx, double* P_dev, double* M_dev, double* normX2_local) {
    if (!ctx || !ctx->splocs) return ASB_ERR_ARG;
    asb_splocs* s = ctx->splocs;
    const int64_t K = s->K, n3 = 3 * ctx->n_loc;
    double* Pout = P_dev ? P_dev : s->P;
    double* Mout = M_dev ? M_dev : s->M;
    int rc;
    if ((rc = asb_transpose(ctx, s->C, K, n3, s->Ct))) return rc;
    // P = X^T C^T is the one pass over X of an outer iteration: on the 128 x 128-tile MFMA kernel of the POD's Gram matrix (two
    // operands, split over row slabs) where the shapes allow -- the one-wave-per-16 x 16-tile kernel takes ~0.4 ms for config 3's
    // 44 379 x 1000 by 44 379 x 64 (it re-reads X once per 16 columns of C)
    static const int big = getenv("ASB_SPLOCS_GRAM_BIG") ? atoi(getenv("ASB_SPLOCS_GRAM_BIG")) : 1;
    if (big && K >= 32 && !(K & 1)) rc = asb_gemm_tn_big(ctx, ctx->X, ctx->Fp, s->Ct, K, n3, (int)ctx->F, (int)K, Pout);
    else rc = asb_gemm_tn(ctx, ctx->X, ctx->Fp, s->Ct, K, n3, (int)ctx->F, (int)K, Pout);
    if (rc) return rc;
    if ((rc = asb_gemm_tn(ctx, s->Ct, K, s->Ct, K, n3, (int)K, (int)K, Mout))) return rc;
    if (normX2_local) return dot_to_host(ctx, s, ctx->X, ctx->X, (long long)n3 * ctx->Fp, normX2_local);
    return ASB_OK;
}

// weight sweep with the (all-reduced) P, M; then G = W^T W.  centres: per component the
// shard's vertex of largest displacement (global index) and its value.
extern "C" int asb_splocs_weights(asb_ctx* ctx, const double* P_dev, const double* M_dev, int64_t* centre_idx,
                                  double* centre_val) {
    if (!ctx || !ctx->splocs) return ASB_ERR_ARG;
    asb_splocs* s = ctx->splocs;
    const int K = (int)s->K, F = (int)ctx->F;
    if (P_dev) ASB_HIP(ctx, hipMemcpyAsync(s->P, P_dev, (size_t)F * K * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    if (M_dev) ASB_HIP(ctx, hipMemcpyAsync(s->M, M_dev, (size_t)K * K * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    // rows per block so that the rows fit the LDS; the blocks must all be resident (one per CU at most)
    int nt = (int)((150 * 1024 / sizeof(double) - 8) / (size_t)(K + 1)) / 64 * 64;
    if (nt > 256) nt = 256;
    const int nblk = nt >= 64 ? (F + nt - 1) / nt : 0;
    const char* env = getenv("ASB_BCD_WIDE");
    if (nt >= 64 && nblk <= ctx->n_cu && !(env && env[0] == '0')) {
        int rc2;
        if ((rc2 = asb_alloc(ctx, &s->bcd_slot, (size_t)K * nblk + 1))) return rc2;
        int* flag = reinterpret_cast<int*>(s->bcd_slot + (size_t)K * nblk);
        const size_t lds = ((size_t)nt * K + K + 4) * sizeof(double);
        if (lds > 48 * 1024) ASB_HIP(ctx, hipFuncSetAttribute((const void*)k_bcd_wide, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_bcd_arm, dim3(64), dim3(256), 0, ctx->stream, s->bcd_slot, (long long)K * nblk, flag);
        hipLaunchKernelGGL(k_bcd_wide, dim3(nblk), dim3(nt), lds, ctx->stream, s->Wfk, s->P, s->M, F, K, s->bcd_slot, flag);
        hipLaunchKernelGGL(k_bcd, dim3(1), dim3(1024), (K + 16) * sizeof(double), ctx->stream, s->Wfk, s->P, s->M, F, K, (const int*)flag);
    } else
        hipLaunchKernelGGL(k_bcd, dim3(1), dim3(1024), (K + 16) * sizeof(double), ctx->stream, s->Wfk, s->P, s->M, F, K, (const int*)nullptr);
    ASB_CHECK_LAUNCH(ctx);
    int rc = asb_gemm_tn(ctx, s->Wfk, K, s->Wfk, K, F, K, K, s->G);
    if (rc) return rc;
    hipLaunchKernelGGL(k_centres, dim3(K), dim3(256), 0, ctx->stream, s->C, (long long)ctx->n_loc, (long long)ctx->v0,
                       s->cen_idx, s->cen_val);
    ASB_CHECK_LAUNCH(ctx);
    // the centres come back through PINNED memory: a device-to-host copy into pageable memory is a blocking staged copy (~100 us
    // of idle GPU each, twice per outer iteration); into pinned memory it is one small asynchronous copy
    if (!s->pin) ASB_HIP(ctx, hipHostMalloc((void**)&s->pin, 4096 * sizeof(double), hipHostMallocDefault));
    const bool fits = (size_t)K <= 2048;
    if (fits && (centre_idx || centre_val)) {
        ASB_HIP(ctx, hipMemcpyAsync(s->pin, s->cen_idx, K * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
        ASB_HIP(ctx, hipMemcpyAsync(s->pin + 2048, s->cen_val, K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (centre_idx) memcpy(centre_idx, s->pin, K * sizeof(long long));
        if (centre_val) memcpy(centre_val, s->pin + 2048, K * sizeof(double));
        return ASB_OK;
    }
    if (centre_idx) ASB_HIP(ctx, hipMemcpyAsync(centre_idx, s->cen_idx, K * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
    if (centre_val) ASB_HIP(ctx, hipMemcpyAsync(centre_val, s->cen_val, K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ASB_OK;
}

// ADMM (:168-181) with Lambda (K, n_loc) already in s->Lambda; leaves C = Z.
static int splocs_admm_run(asb_ctx* ctx, double rho, int n_iter);

extern "C" int asb_splocs_admm(asb_ctx* ctx, const double* Lambda, double rho, int n_iter) {
    if (!ctx || !ctx->splocs || !Lambda) return ASB_ERR_ARG;
    asb_splocs* s = ctx->splocs;
    ASB_HIP(ctx, hipMemcpyAsync(s->Lambda, Lambda, (size_t)s->K * ctx->n_loc * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    return splocs_admm_run(ctx, rho, n_iter);
}

// Lambda[k][i] = lambda * (clip(phi_k[v0 + i], dmin, dmax) - dmin) / (dmax - dmin)   (:162-165, utils/support.py:61-64)
struct FieldPtrs { const double* p[64]; };
__global__ __launch_bounds__(256) void k_lambda_fields_arg(FieldPtrs fp, long long v0, long long n_loc, double lambda, double dmin,
                                                           double dmax, double* __restrict__ L) {
    const double* phi = fp.p[blockIdx.y] + v0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_loc; i += (long long)gridDim.x * 256) {
        const double p = fmin(fmax(phi[i], dmin), dmax);
        L[(long long)blockIdx.y * n_loc + i] = lambda * ((p - dmin) / (dmax - dmin));
    }
}
__global__ __launch_bounds__(256) void k_lambda_fields(const double* const* __restrict__ field, long long v0, long long n_loc,
                                                       double lambda, double dmin, double dmax, double* __restrict__ L) {
    const double* phi = field[blockIdx.y] + v0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_loc; i += (long long)gridDim.x * 256) {
        const double p = fmin(fmax(phi[i], dmin), dmax);
        L[(long long)blockIdx.y * n_loc + i] = lambda * ((p - dmin) / (dmax - dmin));
    }
}

// the same step with the support maps built on the device from cached distance fields (asb_geodesic_cache_add):
// slots (K, host) = cache slot of each component's centre
extern "C" int asb_splocs_admm_fields(asb_ctx* ctx, const int64_t* slots, double lambda, double dmin, double dmax, double rho,
                                      int n_iter) {
    if (!ctx || !ctx->splocs || !slots) return ASB_ERR_ARG;
    asb_splocs* s = ctx->splocs;
    const int64_t K = s->K;
    std::vector<const double*> ptrs((size_t)K);
    for (int64_t k = 0; k < K; ++k) {
        long long n = 0;
        ptrs[(size_t)k] = asb_geo_cached_field(ctx, slots[k], &n);
        if (!ptrs[(size_t)k]) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_splocs_admm_fields: no cached field in slot %lld", (long long)slots[k]);
        if (n != ctx->N_glob) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_splocs_admm_fields: the mesh has %lld vertices, the snapshots %lld", n,
                                       (long long)ctx->N_glob);
    }
    int rc;
    const int gx = (int)((ctx->n_loc + 255) / 256 < 64 ? (ctx->n_loc + 255) / 256 : 64);
    if (K <= 64) {              // the K field addresses travel as a kernel argument: no upload, no synchronisation
        FieldPtrs fp{};
        for (int64_t k = 0; k < K; ++k) fp.p[k] = ptrs[(size_t)k];
        hipLaunchKernelGGL(k_lambda_fields_arg, dim3(gx, (unsigned)K), dim3(256), 0, ctx->stream, fp, (long long)ctx->v0,
                           (long long)ctx->n_loc, lambda, dmin, dmax, s->Lambda);
    } else {
        if ((rc = asb_alloc(ctx, &s->field_ptr, (size_t)K))) return rc;
        ASB_HIP(ctx, hipMemcpyAsync(s->field_ptr, ptrs.data(), (size_t)K * sizeof(double*), hipMemcpyHostToDevice, ctx->stream));
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));      // ptrs is a local
        hipLaunchKernelGGL(k_lambda_fields, dim3(gx, (unsigned)K), dim3(256), 0, ctx->stream, s->field_ptr, (long long)ctx->v0,
                           (long long)ctx->n_loc, lambda, dmin, dmax, s->Lambda);
    }
    ASB_CHECK_LAUNCH(ctx);
    return splocs_admm_run(ctx, rho, n_iter);
}

static int splocs_admm_run(asb_ctx* ctx, double rho, int n_iter) {
    asb_splocs* s = ctx->splocs;
    const int64_t K = s->K, n3 = 3 * ctx->n_loc, kn = K * ctx->n_loc;
    // c = W^T X  (K x 3n): the deflation's projection kernel, 16 columns per pass over X
    // (round 4: 64 columns per pass through the panel reads' four-tile kernel; config 3: 4 x 89 us + 8 small launches -> one pass)
    static const int wide = getenv("ASB_SPLOCS_WIDE") ? atoi(getenv("ASB_SPLOCS_WIDE")) : 1;
    const int step = wide ? 64 : 16;
    for (int64_t k0 = 0; k0 < K; k0 += step) {
        const int nc = (int)((K - k0) < step ? (K - k0) : step);
        int rc = wide ? asb_project_columns_wide(ctx, s->Wfk, K, k0, nc, s->c + (size_t)k0 * n3, nullptr)
                      : asb_project_columns(ctx, s->Wfk, K, k0, nc, s->c + (size_t)k0 * n3, nullptr);
        if (rc) return rc;
    }
    if (!s->defer_status) ASB_HIP(ctx, hipMemsetAsync(s->status, 0, 4 * sizeof(int), ctx->stream));
    {                   // blocked Gauss-Jordan on the matrix padded to a multiple of 16 (asb_dense.hip: one in-LDS pivot block up to K = 256)
        const int np = (int)((K + 15) / 16 * 16);
        int rc2;
        if ((rc2 = asb_alloc(ctx, &ctx->dn_test, (size_t)np * np))) return rc2;
        hipLaunchKernelGGL(k_pad_spd, dim3(256), dim3(256), 0, ctx->stream, s->G, rho, (int)K, np, ctx->dn_test);
        ASB_CHECK_LAUNCH(ctx);
        if ((rc2 = asb_dense_spd_inverse(ctx, ctx->dn_test, np))) return rc2;
        ASB_HIP(ctx, hipMemcpy2DAsync(s->Ginv, (size_t)K * sizeof(double), ctx->dn_test, (size_t)np * sizeof(double),
                                      (size_t)K * sizeof(double), (size_t)K, hipMemcpyDeviceToDevice, ctx->stream));
    }
    ASB_HIP(ctx, hipMemcpyAsync(s->Z, s->C, (size_t)K * n3 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));   // Z = C.copy()
    const long long n = (long long)K * n3;
    const int eg = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    const int pg = (int)((kn + 255) / 256 < 4096 ? (kn + 255) / 256 : 4096);
    static const int fused = getenv("ASB_ADMM_FUSED") ? atoi(getenv("ASB_ADMM_FUSED")) : 1;
    if (fused && K <= 64 && n_iter > 0) {
        hipLaunchKernelGGL(k_admm_fused, dim3((unsigned)((ctx->n_loc + 15) / 16)), dim3(256), 0, ctx->stream, s->c, s->Z, s->U, s->Ginv,
                           s->Lambda, rho, (int)K, (long long)ctx->n_loc, n_iter);
        n_iter = 0;
    }
    for (int it = 0; it < n_iter; ++it) {
        hipLaunchKernelGGL(k_admm_rhs, dim3(eg), dim3(256), 0, ctx->stream, s->c, s->Z, s->U, rho, n, s->rhs);
        int rc = asb_gemm_tn(ctx, s->Ginv, K, s->rhs, n3, K, (int)K, (int)n3, s->C);      // C = (G + rho I)^-1 rhs
        if (rc) return rc;
        hipLaunchKernelGGL(k_admm_prox, dim3(pg), dim3(256), 0, ctx->stream, s->C, s->Z, s->U, s->Lambda, 1.0 / rho, kn);
    }
    ASB_HIP(ctx, hipMemcpyAsync(s->C, s->Z, (size_t)K * n3 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));   // C = Z
    ASB_CHECK_LAUNCH(ctx);
    if (s->defer_status) return ASB_OK;          // (asb_splocs_trace looks at it, once, with the objective trace)
    int st[4];
    ASB_HIP(ctx, hipMemcpyAsync(st, s->status, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (st[0]) ASB_FAIL(ctx, ASB_ERR_NUMERIC, "SPLOCS: W^T W + rho I is not positive definite");
    return ASB_OK;
}

// objective pieces after asb_splocs_gram() was called for the NEW C (with all-reduced P, M):
// wp = <W, P>, gm = <G, M>, sparsity_local = sum Lambda |C_v|   (:183-186)
extern "C" int asb_splocs_objective(asb_ctx* ctx, const double* P_dev, const double* M_dev, double* wp, double* gm,
                                    double* sparsity_local) {
    if (!ctx || !ctx->splocs) return ASB_ERR_ARG;
    asb_splocs* s = ctx->splocs;
    const int64_t K = s->K;
    if (P_dev) ASB_HIP(ctx, hipMemcpyAsync(s->P, P_dev, (size_t)ctx->F * K * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    if (M_dev) ASB_HIP(ctx, hipMemcpyAsync(s->M, M_dev, (size_t)K * K * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    int rc;
    if (wp && (rc = dot_to_host(ctx, s, s->Wfk, s->P, (long long)ctx->F * K, wp))) return rc;
    if (gm && (rc = dot_to_host(ctx, s, s->G, s->M, (long long)K * K, gm))) return rc;
    if (sparsity_local) {
        const long long kn = K * ctx->n_loc;
        const int grid = (int)((kn + 255) / 256 < 1024 ? (kn + 255) / 256 : 1024);
        hipLaunchKernelGGL(k_sparsity_part, dim3(grid), dim3(256), 0, ctx->stream, s->C, s->Lambda, kn, s->red);
        hipLaunchKernelGGL(k_sum1, dim3(1), dim3(256), 0, ctx->stream, s->red, grid, s->red + 1024);
        ASB_CHECK_LAUNCH(ctx);
        ASB_HIP(ctx, hipMemcpyAsync(sparsity_local, s->red + 1024, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return ASB_OK;
}

// The same three numbers of outer iteration `it` left ON THE DEVICE (no synchronisation): the trace of a whole run -- the lines
// the reference prints at :186-189 -- is read once, by asb_splocs_trace, when the loop is over.  asb_splocs_trace_begin(n_its)
// sizes the trace and defers the ADMM's status check (W^T W + rho I not positive definite) to that read as well.
extern "C" int asb_splocs_trace_begin(asb_ctx* ctx, int64_t n_its) {
    if (!ctx || !ctx->splocs || n_its < 1) return ASB_ERR_ARG;
    asb_splocs* s = ctx->splocs;
    int rc;
    if ((rc = asb_alloc(ctx, &s->trace, (size_t)n_its * 3))) return rc;
    s->trace_cap = n_its;
    s->defer_status = true;
    ASB_HIP(ctx, hipMemsetAsync(s->trace, 0, (size_t)n_its * 3 * sizeof(double), ctx->stream));
    ASB_HIP(ctx, hipMemsetAsync(s->status, 0, 4 * sizeof(int), ctx->stream));
    return ASB_OK;
}
extern "C" int asb_splocs_objective_dev(asb_ctx* ctx, const double* P_dev, const double* M_dev, int64_t it) {
    if (!ctx || !ctx->splocs) return ASB_ERR_ARG;
    asb_splocs* s = ctx->splocs;
    if (!s->trace || it < 0 || it >= s->trace_cap) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_splocs_objective_dev: iteration %lld outside the trace", (long long)it);
    const int64_t K = s->K;
    if (P_dev) ASB_HIP(ctx, hipMemcpyAsync(s->P, P_dev, (size_t)ctx->F * K * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    if (M_dev) ASB_HIP(ctx, hipMemcpyAsync(s->M, M_dev, (size_t)K * K * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    int rc;
    double* t = s->trace + it * 3;
    if ((rc = dot_to_dev(ctx, s, s->Wfk, s->P, (long long)ctx->F * K, t))) return rc;
    if ((rc = dot_to_dev(ctx, s, s->G, s->M, (long long)K * K, t + 1))) return rc;
    const long long kn = K * ctx->n_loc;
    const int grid = (int)((kn + 255) / 256 < 1024 ? (kn + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_sparsity_part, dim3(grid), dim3(256), 0, ctx->stream, s->C, s->Lambda, kn, s->red);
    hipLaunchKernelGGL(k_sum1, dim3(1), dim3(256), 0, ctx->stream, s->red, grid, t + 2);
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}
extern "C" int asb_splocs_trace(asb_ctx* ctx, int64_t n_its, double* out) {
    if (!ctx || !ctx->splocs || !out) return ASB_ERR_ARG;
    asb_splocs* s = ctx->splocs;
    if (!s->trace || n_its < 1 || n_its > s->trace_cap) return ASB_ERR_ARG;
    int st[4];
    ASB_HIP(ctx, hipMemcpyAsync(out, s->trace, (size_t)n_its * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipMemcpyAsync(st, s->status, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    s->defer_status = false;
    if (st[0]) ASB_FAIL(ctx, ASB_ERR_NUMERIC, "SPLOCS: W^T W + rho I is not positive definite");
    return ASB_OK;
}

// refined components (K, n_loc, 3) and weights (F, K); either may be NULL
extern "C" int asb_splocs_results(asb_ctx* ctx, double* C_out, double* W_out) {
    if (!ctx || !ctx->splocs) return ASB_ERR_ARG;
    asb_splocs* s = ctx->splocs;
    if (C_out) ASB_HIP(ctx, hipMemcpyAsync(C_out, s->C, (size_t)s->K * 3 * ctx->n_loc * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (W_out) ASB_HIP(ctx, hipMemcpyAsync(W_out, s->Wfk, (size_t)ctx->F * s->K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ASB_OK;
}

void asb_splocs_free(asb_ctx* ctx) {
    if (ctx->splocs) {
        delete ctx->splocs;       // device buffers are released through ctx->alloc_bytes
        ctx->splocs = nullptr;
    }
}
