// asb_eig.hip -- symmetric eigen-problem of a LARGE dense matrix (config 5: the F x F Gram matrix of the
// constraint snapshots, F = 4000) on the device: Householder tridiagonalisation A = Q T Q^T and the
// back-transformation of T's eigenvectors.  The tridiagonal problem itself (O(n^2) for the values, O(n k) for k
// vectors) is solved by the caller between the two calls.  Replaces the LAPACK `gesdd` of
// constraintsComponents.py:307 together with asb_pod_gram / asb_pod_basis.  gfx950 only.
//
// Layout: A is n x n row-major with BOTH triangles valid (128 MB at n = 4000: resident in the 256 MB MALL).
// Step j (j = 0 .. n-3) annihilates A[j+2.., j]:  H_j = I - tau_j v_j v_j^T, v_j[j+1] = 1, and updates the
// trailing block A22 <- A22 - v w^T - w v^T with p = tau A22 v, w = p - (tau/2)(p.v) v.  Two launches per step:
//   k_td_small (1 block)  : p.v, w_j, then the NEXT Householder vector from row j+1 of the not yet updated
//                           matrix (row j+1 of A22 - v w^T - w v^T is formed on the fly), d[j+1], e[j+1], tau_{j+1}
//   k_td_update (grid)    : trailing block update fused with the next step's symmetric mat-vec
//                           p_{j+1} = tau_{j+1} A22' v_{j+1}  -- ONE read + ONE write of the trailing block per step
// All reductions are ordered (no atomics): every rank of a multi-GPU run gets bit-identical T and vectors.
// v_j is kept in row j of A (columns j+1..n-1), which the trailing block no longer touches.
#include "asb_common.h"

#include <cstdlib>

#define TD_T 1024

// step `j` (j = -1: only the first Householder vector from row 0).  vcur = v_j, vnext = v_{j+1} (absolute row index).
// p arrives as `nch` partial vectors (one per column chunk of k_td_update, each n long): summed here in chunk order
__global__ __launch_bounds__(TD_T) void k_td_small(double* __restrict__ A, int n, int j, const double* __restrict__ p, int nch,
                                                  const double* __restrict__ vcur, double* __restrict__ w,
                                                  double* __restrict__ vnext, double* __restrict__ xbuf,
                                                  double* __restrict__ tau, double* __restrict__ d, double* __restrict__ e, int fresh = 0) {
    // fresh: the matrix is explicitly up to date from row j + 1 on (behind the panel kernels): no reflector j to finish
    __shared__ double sh[TD_T / 64 * 2];
    __shared__ double bc[4];
    const int tid = threadIdx.x;
    const bool have = j >= 0 && !fresh;
    const int r0 = j + 1;                       // first row of the current trailing block
    double wfirst = 0.0;
    if (have) {
        double acc[1] = {0.0};
        for (int r = r0 + tid; r < n; r += TD_T) {
            double pr = p[r];
            for (int c = 1; c < nch; ++c) pr += p[(long long)c * n + r];
            xbuf[r] = pr;                       // (xbuf is free until the row is formed below)
            acc[0] += pr * vcur[r];
        }
        block_sum<1>(acc, sh);
        const double half = 0.5 * tau[j] * acc[0];
        for (int r = r0 + tid; r < n; r += TD_T) {
            const double wr = xbuf[r] - half * vcur[r];
            w[r] = wr;
            if (r == r0) bc[0] = wr;
        }
        __syncthreads();
        wfirst = bc[0];
    }
    // row r0 of the updated matrix, columns c >= r0 (vcur[r0] == 1)
    const double* row = A + (long long)r0 * n;
    double acc2[1] = {0.0};
    for (int c = r0 + tid; c < n; c += TD_T) {
        double x = row[c];
        if (have) x -= w[c] + wfirst * vcur[c];
        xbuf[c] = x;
        if (c == r0) bc[1] = x;                 // the new diagonal entry
        if (c == r0 + 1) bc[2] = x;             // alpha
        if (c >= r0 + 2) acc2[0] += x * x;
    }
    block_sum<1>(acc2, sh);
    __syncthreads();
    const double sigma = acc2[0];
    if (tid == 0) d[r0] = bc[1];
    if (r0 + 1 >= n) return;                    // (never launched that far)
    const double alpha = bc[2];
    if (r0 + 2 >= n) {                          // last 2 x 2 block: no reflector left
        if (tid == 0) {
            e[r0] = alpha;
            tau[r0] = 0.0;
            // d[n-1] = A[n-1][n-1] - 2 v[n-1] w[n-1]
            double last = A[(long long)(n - 1) * n + (n - 1)];
            if (have) last -= 2.0 * vcur[n - 1] * w[n - 1];
            d[n - 1] = last;
        }
        return;
    }
    double beta, t, scale;
    if (sigma == 0.0) { beta = alpha; t = 0.0; scale = 0.0; }
    else {
        beta = -copysign(sqrt(alpha * alpha + sigma), alpha);
        t = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
    }
    if (tid == 0) { e[r0] = beta; tau[r0] = t; }
    double* arow = A + (long long)r0 * n;
    for (int c = r0 + 1 + tid; c < n; c += TD_T) {
        const double v = (c == r0 + 1) ? 1.0 : xbuf[c] * scale;
        vnext[c] = v;
        arow[c] = v;
    }
}

// The same step with everything a thread owns in REGISTERS (trailing block of at most EPT x 1024 rows): the partial vectors, v and
// row r0 of the matrix are requested together up front, and no value makes a round trip through global scratch between the
// phases -- the step is a chain of dependent memory latencies, not work (8.3 -> about 6 us).  Same thread-to-element mapping
// and summation order as k_td_small: bit-identical results.
template <int EPT>
__global__ __launch_bounds__(TD_T) void k_td_small_reg(double* __restrict__ A, int n, int j, const double* __restrict__ p, int nch,
                                                      const double* __restrict__ vcur, double* __restrict__ w,
                                                      double* __restrict__ vnext, double* __restrict__ tau, double* __restrict__ d,
                                                      double* __restrict__ e, int fresh = 0) {
    __shared__ double sh[TD_T / 64 * 2];
    __shared__ double bc[4];
    const int tid = threadIdx.x;
    const bool have = j >= 0 && !fresh;
    const int r0 = j + 1;
    const double* row = A + (long long)r0 * n;
    double pr[EPT], vc[EPT], x[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int r = r0 + tid + q * TD_T;
        const bool on = r < n;
        x[q] = on ? row[r] : 0.0;
        vc[q] = (on && have) ? vcur[r] : 0.0;
        double s = 0.0;
        if (on && have) {
            s = p[r];
            for (int c = 1; c < nch; ++c) s += p[(long long)c * n + r];
        }
        pr[q] = s;
    }
    double wfirst = 0.0;
    if (have) {
        double acc[1] = {0.0};
#pragma unroll
        for (int q = 0; q < EPT; ++q) acc[0] += pr[q] * vc[q];
        block_sum<1>(acc, sh);
        const double half = 0.5 * tau[j] * acc[0];
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int r = r0 + tid + q * TD_T;
            pr[q] = pr[q] - half * vc[q];                 // w
            if (r < n) {
                w[r] = pr[q];
                if (r == r0) bc[0] = pr[q];
            }
        }
        __syncthreads();
        wfirst = bc[0];
    }
    double acc2[1] = {0.0};
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int c = r0 + tid + q * TD_T;
        if (c < n) {
            if (have) x[q] -= pr[q] + wfirst * vc[q];
            if (c == r0) bc[1] = x[q];
            if (c == r0 + 1) bc[2] = x[q];
            if (c >= r0 + 2) acc2[0] += x[q] * x[q];
        }
    }
    block_sum<1>(acc2, sh);
    __syncthreads();
    const double sigma = acc2[0];
    if (tid == 0) d[r0] = bc[1];
    if (r0 + 1 >= n) return;
    const double alpha = bc[2];
    if (r0 + 2 >= n) {
        if (tid == 0) {
            e[r0] = alpha;
            tau[r0] = 0.0;
            double last = A[(long long)(n - 1) * n + (n - 1)];
            if (have) last -= 2.0 * vcur[n - 1] * w[n - 1];
            d[n - 1] = last;
        }
        return;
    }
    double beta, t, scale;
    if (sigma == 0.0) { beta = alpha; t = 0.0; scale = 0.0; }
    else {
        beta = -copysign(sqrt(alpha * alpha + sigma), alpha);
        t = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
    }
    if (tid == 0) { e[r0] = beta; tau[r0] = t; }
    double* arow = A + (long long)r0 * n;
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int c = r0 + tid + q * TD_T;
        if (c >= r0 + 1 && c < n) {
            const double v = (c == r0 + 1) ? 1.0 : x[q] * scale;
            vnext[c] = v;
            arow[c] = v;
        }
    }
}

// trailing block rows/cols >= r1 = j + 2:  A -= vcur w^T + w vcur^T (when `update`), then the next step's symmetric mat-vec
// p[i] = taun * sum_c A[i][c] vnext[c].  Work item = (group of R rows) x (chunk of TD_CW columns): a wave handles one item
// and writes its partial row sums to p[chunk * n + i] (k_td_small adds the chunks in order: deterministic) -- a 2-D
// decomposition, so that even the 128 MB start matrix gives every SIMD several waves and the sweep runs at the rate of the
// Infinity Cache instead of one wave's latency chain per four rows.
#define TD_CW 1024
template <int R, int CU = 2>
__global__ __launch_bounds__(256) void k_td_update(double* __restrict__ A, int n, int r1, int update,
                                                  const double* __restrict__ vcur, const double* __restrict__ w,
                                                  const double* __restrict__ vnext, const double* __restrict__ tau_next,
                                                  double* __restrict__ p, int nch) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * 256) >> 6;
    const double tn = *tau_next;
    const int ngroups = (n - r1 + R - 1) / R;
    for (long long item = wave; item < (long long)ngroups * nch; item += nwaves) {
        const int i0 = r1 + (int)(item / nch) * R, cc = (int)(item % nch);
        const int c_lo = r1 + cc * TD_CW, c_hi = (c_lo + TD_CW < n) ? c_lo + TD_CW : n;
        double vi[R], wi[R], acc[R];
        double* rowp[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int i = (i0 + q < n) ? i0 + q : n - 1;      // clamped rows repeat the last one (not stored twice: see below)
            vi[q] = update ? vcur[i] : 0.0;
            wi[q] = update ? w[i] : 0.0;
            acc[q] = 0.0;
            rowp[q] = A + (long long)i * n;
        }
        for (int c = c_lo + lane; c < c_hi; c += 64 * CU) {   // CU column positions per lane in flight (x R rows)
            double vc[CU], wc[CU], vn[CU], a[CU][R];
            bool on[CU];
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int cu = c + 64 * u;
                on[u] = cu < c_hi;
                vc[u] = (update && on[u]) ? vcur[cu] : 0.0;
                wc[u] = (update && on[u]) ? w[cu] : 0.0;
                vn[u] = on[u] ? vnext[cu] : 0.0;
#pragma unroll
                for (int q = 0; q < R; ++q) a[u][q] = on[u] ? rowp[q][cu] : 0.0;
            }
            // (the sums keep round 3's order for CU = 2: a[0] vn[0] + a[1] vn[1] per step)
#pragma unroll
            for (int q = 0; q < R; ++q) {
                double s = 0.0;
#pragma unroll
                for (int u = 0; u < CU; ++u) {
                    if (update) {
                        a[u][q] -= vi[q] * wc[u] + wi[q] * vc[u];
                        if (i0 + q < n && on[u]) rowp[q][c + 64 * u] = a[u][q];
                    }
                    s = u == 0 ? a[u][q] * vn[u] : s + a[u][q] * vn[u];
                }
                acc[q] += s;
            }
        }
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const double s = wave_sum(acc[q]);
            if (lane == 0 && i0 + q < n) p[(long long)cc * n + i0 + q] = tn * s;
        }
    }
}

// --------------------------------------------------------------------------------------
// k_td_panel (round 4): TDP_NB reflectors per LAUNCH of one co-resident grid, the trailing block only READ.
// The two-launch step above moves 16 m^2 bytes per reflector (read + write of the trailing block, 340 GB in all for n = 4000:
// 77 of the 104 ms, at the 4.4 TB/s the Infinity Cache gives a read/write mix).  LAPACK's dlatrd form defers the writes: inside a
// panel the matrix stays as it is and every use of it is corrected by the panel's reflectors so far,
//     A_cur = A - sum_k (v_k w_k^T + w_k v_k^T),
// the rank-2 TDP_NB update being applied once per panel (k_td_rank2k).  Per reflector that needs three grid-wide steps -- the
// corrected row and its norm, the mat-vec with the dots W^T v, V^T v, the new w -- which as launches would cost what they
// save; here they are phases of ONE kernel with two exchanges per reflector:
//   phase I  (by index slices): w of the previous reflector for the slice, then row r of A_cur for the slice -> x, |x|^2 partial
//   -- exchange A (one word per block: the partial) -> sigma, the reflector's beta / tau / scale, identically on every block
//   phase II (2-D items, as k_td_update): z[chunk][i] = sum_c A[i][c] v[c] (v in LDS), v^T A v partial; one wave per dot
//            p_k = W_k . v, q_k = V_k . v; v -> row r of A (the back-transformation reads it there) and V[jj]
//   -- exchange B (one word per block: the v^T A v partial) -> y . v = v^T A v - 2 p . q
// Exchanges as in k_panel_multi (asb_project.hip): agent-scope (write-through) stores and loads, self-validating words in a ring
// of three, no cache maintenance; every vector another block reads (x, z, p, q, V, W) goes through the same stores / loads.
// All sums in a fixed order: every rank of a multi-GPU run gets the same T.  A spin limit turns a lost block into an error flag.
#define TDP_NB 32
#define TDP_T 512
#define TDP_SENT 0xFFFFFFFFFFFFFFFFull
struct TdpBuf { double* x; double* zp; double* pq; double* V; double* W; unsigned long long* rec; unsigned* flags; unsigned long long* tlog; int test_stall; };
#define TDP_STAMP(slot) do { if (B.tlog && b == 0 && tid == 0) B.tlog[(size_t)jj * 8 + (slot)] = wall_clock64_td(); } while (0)
__device__ __forceinline__ unsigned long long wall_clock64_td() { return wall_clock64(); }      // 100 MHz
__device__ __forceinline__ void tdp_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double tdp_load(const double* p) {
    return __hip_atomic_load(const_cast<double*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every block: the ordered sum of all blocks' words of generation `gen` (-> LDS red[0]); false: timed out / aborted.
// Two halves, so that loads which do not depend on the exchange can be put in flight between them.
__device__ __forceinline__ void tdp_post(const TdpBuf& B, int G, int gen, double mine, int* dead_sh) {
    __builtin_amdgcn_s_waitcnt(0);               // this wave's write-through stores of the phase are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        if (!(B.test_stall && (int)blockIdx.x == G - 1))        // (tests: the last block never signals)
        __hip_atomic_store(B.rec + (size_t)(gen % 3) * G + blockIdx.x, (unsigned long long)__double_as_longlong(mine), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        *dead_sh = 0;                            // (read behind the poll's barrier; a timed-out poller sets it to 1 after this)
    }
}
__device__ __forceinline__ bool tdp_poll(const TdpBuf& B, int G, int gen, double* red, int* dead_sh) {
    const int tid = threadIdx.x;
    unsigned long long* ring = B.rec + (size_t)(gen % 3) * G;
    double v = 0.0;
    if (tid < G) {
        long long spins = 0;
        for (;;) {
            const unsigned long long w = __hip_atomic_load(ring + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (w != TDP_SENT) { v = __longlong_as_double((long long)w); break; }
            if ((++spins & 255) == 0 && (spins > (B.test_stall ? (1LL << 12) : (1LL << 24)) || __hip_atomic_load(B.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(B.flags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *dead_sh = 1;
                break;
            }
        }
    }
    // the sum in a fixed order: lane butterflies inside each wave, then the waves in turn (the same on every block); ONE barrier
    const double wsum = wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = wsum;
    __syncthreads();
    // Everybody has posted generation `gen`, so everybody is past its poll of generation gen - 1: this block's word of that
    // generation goes back to "not written" (its slot is written again at gen + 2, behind the drain in front of gen + 1's post)
    if (tid == 0 && *dead_sh == 0)
        __hip_atomic_store(B.rec + (size_t)((gen + 2) % 3) * G + blockIdx.x, TDP_SENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double sacc = 0.0;
#pragma unroll
    for (int q = 0; q < TDP_T / 64; ++q) sacc += red[q];
    const bool ok = *dead_sh == 0;
    __syncthreads();                             // (red and dead_sh are free again)
    red[0] = sacc;                               // every thread the same value: callers read red[0] behind their next barrier
    return ok;
}
__device__ __forceinline__ bool tdp_exchange(const TdpBuf& B, int G, int gen, double mine, double* red, int* dead_sh) {
    tdp_post(B, G, gen, mine, dead_sh);
    return tdp_poll(B, G, gen, red, dead_sh);
}
__global__ __launch_bounds__(TDP_T, 2) void k_td_panel(double* __restrict__ A, int n, int j0, int nb, TdpBuf B, double* __restrict__ tau,
                                                       double* __restrict__ d, double* __restrict__ e) {
    extern __shared__ double tdp_lds[];
    double* v_sh = tdp_lds;                       // n doubles: the current reflector (0 below its support)
    double* red = tdp_lds + n;                    // TDP_T
    double* pq_sh = red + TDP_T;                  // 2 TDP_NB: p_k = W_k . v, q_k = V_k . v
    double* rowk = pq_sh + 2 * TDP_NB;            // 2 TDP_NB: V_k[r], W_k[r] of the current row
    __shared__ int dead_sh;
    __shared__ double bc[8];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, G = gridDim.x, b = blockIdx.x;
    constexpr int NWB = TDP_T / 64;
    const long long gw = (long long)b * NWB + wv, NW = (long long)G * NWB;
    const int sl0 = (int)((long long)n * b / G), sl1 = (int)((long long)n * (b + 1) / G);      // this block's index slice
    const int half = lane >> 5, kl = lane & 31;   // phase I: half a wave per index, lane <-> reflector of the panel
    int gen = 0;
    double t_prev = 0.0, yv_prev = 0.0;
    for (int jj = 0; jj <= nb; ++jj) {
        const int r = j0 + jj;                    // the reflector built in this round (jj == nb: only the last w is finished)
        TDP_STAMP(0);
        // ---- phase I
        // w_{r-1}[r], needed by every block for the row: from z[.][r], the panel's rows at r and p, q
        if (jj > 0) {
            if (tid < 2 * TDP_NB) {
                const int k = tid & (TDP_NB - 1);
                rowk[tid] = (k < jj - 1) ? tdp_load((tid < TDP_NB ? B.V : B.W) + (size_t)k * n + r) : 0.0;
            }
            __syncthreads();
            if (wv == 0) {
                const int rp = r - 1, nch = (n - (rp + 1) + TD_CW - 1) / TD_CW;
                double y = 0.0;
                if (lane == 0) {
                    for (int c0 = 0; c0 < nch; c0 += 8) {
                        double zz[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) zz[u] = c0 + u < nch ? tdp_load(B.zp + (size_t)(c0 + u) * n + r) : 0.0;
#pragma unroll
                        for (int u = 0; u < 8; ++u) y += zz[u];
                    }
                }
                double corr = (lane < jj - 1 && lane < TDP_NB) ? rowk[lane] * pq_sh[lane] + rowk[TDP_NB + lane] * pq_sh[TDP_NB + lane] : 0.0;
                for (int o = 16; o > 0; o >>= 1) corr += __shfl_xor(corr, o, 64);
                if (lane == 0) {
                    y -= corr;
                    const double wr = t_prev * (y - 0.5 * t_prev * yv_prev * 1.0);      // v_{r-1}[r] = 1
                    bc[0] = wr;
                }
            }
            __syncthreads();
            if (tid == 0) { rowk[jj - 1] = 1.0; rowk[TDP_NB + jj - 1] = bc[0]; }
            __syncthreads();
        }
        double sig_part = 0.0;
        for (int i0 = sl0; i0 < sl1; i0 += 2 * NWB) {
            const int i = i0 + 2 * wv + half;
            const bool on = i < sl1 && i >= r && i < n;
            // lane kl: reflector k = kl of the panel (k < jj - 1 from memory; k = jj - 1 is the one being finished)
            double vk = 0.0, wk = 0.0;
            if (on && kl < jj - 1) {
                vk = tdp_load(B.V + (size_t)kl * n + i);
                wk = tdp_load(B.W + (size_t)kl * n + i);
            }
            double s3 = (kl < jj - 1) ? vk * pq_sh[kl] + wk * pq_sh[TDP_NB + kl] : 0.0;
            double s1 = (kl < jj - 1) ? rowk[kl] * wk + rowk[TDP_NB + kl] * vk : 0.0;
            for (int o = 16; o > 0; o >>= 1) { s3 += __shfl_xor(s3, o, 64); s1 += __shfl_xor(s1, o, 64); }
            if (kl == 0 && on) {
                double xi = (jj < nb) ? A[(size_t)r * n + i] : 0.0;
                if (jj > 0) {
                    const int rp = r - 1, nch = (n - (rp + 1) + TD_CW - 1) / TD_CW;
                    double y = 0.0;
                    for (int c0 = 0; c0 < nch; c0 += 8) {
                        double zz[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) zz[u] = c0 + u < nch ? tdp_load(B.zp + (size_t)(c0 + u) * n + i) : 0.0;
#pragma unroll
                        for (int u = 0; u < 8; ++u) y += zz[u];
                    }
                    y -= s3;
                    const double vi = v_sh[i];
                    const double wi = t_prev * (y - 0.5 * t_prev * yv_prev * vi);
                    tdp_store(B.W + (size_t)(jj - 1) * n + i, wi);
                    xi -= s1 + (1.0 * wi + bc[0] * vi);       // k = jj - 1: V[r] = 1, W[r] = w_{r-1}[r]
                }
                if (jj < nb) {
                    tdp_store(B.x + i, xi);
                    if (i >= r + 2) sig_part += xi * xi;
                }
            }
        }
        if (jj == nb) break;                       // (the kernel's end publishes the last w)
        // block partial of sigma, fixed order: lanes 0 / 32 of each wave hold one
        red[tid] = sig_part;
        __syncthreads();
        if (tid == 0) {
            double sacc = 0.0;
            for (int q = 0; q < TDP_T; q += 32) sacc += red[q];
            bc[1] = sacc;
        }
        __syncthreads();
        TDP_STAMP(1);
        // the matrix rows of this wave's FIRST mat-vec item do not depend on the exchange: requested behind the post, they travel
        // while the words do (at n = 4000 a wave has one or two items: up to the whole mat-vec's memory time is hidden)
        const int r1 = r + 1, rows = n - r1;
        const int nch = (rows + TD_CW - 1) / TD_CW, ngroups = (rows + 3) / 4;
        const long long n_items = (long long)ngroups * nch;
        double a_pre[8][4];
        tdp_post(B, G, gen, bc[1], &dead_sh);
        if (gw < n_items) {
            const int i0 = r1 + (int)(gw / nch) * 4, cc = (int)(gw % nch);
            const int c_lo = r1 + cc * TD_CW, c_hi = (c_lo + TD_CW < n) ? c_lo + TD_CW : n;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int cu = c_lo + lane + 64 * u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = (i0 + q < n) ? i0 + q : n - 1;
                    a_pre[u][q] = cu < c_hi ? A[(size_t)i * n + cu] : 0.0;
                }
            }
        }
        if (!tdp_poll(B, G, gen++, red, &dead_sh)) return;
        TDP_STAMP(2);
        const double sigma = red[0];
        __syncthreads();
        if (tid == 0) { bc[2] = tdp_load(B.x + r); bc[3] = tdp_load(B.x + r + 1); }
        __syncthreads();
        double vav = 0.0;
        const double diag = bc[2], alpha = bc[3];
        double beta, t, scale;
        if (sigma == 0.0) { beta = alpha; t = 0.0; scale = 0.0; }
        else {
            beta = -copysign(sqrt(alpha * alpha + sigma), alpha);
            t = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        if (b == 0 && tid == 0) { d[r] = diag; e[r] = beta; tau[r] = t; }
        // ---- phase II
        for (int c0 = 0; c0 < n; c0 += 8 * TDP_T) {             // (eight loads per thread in flight)
            double xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + tid + u * TDP_T;
                xv[u] = (c >= r + 2 && c < n) ? tdp_load(B.x + c) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + tid + u * TDP_T;
                if (c < n) v_sh[c] = (c <= r) ? 0.0 : (c == r + 1 ? 1.0 : xv[u] * scale);
            }
        }
        __syncthreads();
        for (int i = sl0 + tid; i < sl1; i += TDP_T)
            if (i >= r + 1) {
                tdp_store(B.V + (size_t)jj * n + i, v_sh[i]);
                A[(size_t)r * n + i] = v_sh[i];
            }
        TDP_STAMP(3);
        // the dots with the panel so far, one BLOCK each from the far end of the grid (all loads of a thread in flight at once; a
        // wave walking a vector alone was 60 dependent round trips at n = 4000)
        for (int q = G - 1 - b; q < 2 * jj; q += G) {
            const int k = q >> 1;
            const double* src = ((q & 1) ? B.V : B.W) + (size_t)k * n;
            double sdot = 0.0;
            for (int c0 = r1; c0 < n; c0 += 8 * TDP_T) {
                double xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + tid + u * TDP_T;
                    xv[u] = c < n ? tdp_load(src + c) : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + tid + u * TDP_T;
                    if (c < n) sdot += xv[u] * v_sh[c];
                }
            }
            red[tid] = sdot;
            __syncthreads();
            for (int o = TDP_T / 2; o > 0; o >>= 1) {
                if (tid < o) red[tid] += red[tid + o];
                __syncthreads();
            }
            if (tid == 0) tdp_store(B.pq + (q & 1) * TDP_NB + k, red[0]);
            __syncthreads();
        }
        TDP_STAMP(4);
        // items of 4 rows x TD_CW columns, round-robin over the waves of the grid (a wave's first batch of 32 loads is in registers).
        // Measured and rejected: items of 4 x 256 columns (no faster, and 12 - 16 partial sums per index double phase I); one wave
        // per SIMD with 3 or 5 items of 4 x 512 requested ahead (99.8 / 103.4 ms against 93.1: the requests delay the poll behind
        // them, and finishing an item -- four wave sums -- is 1.7 us of its own)
        for (long long item = gw; item < n_items; item += NW) {
            const int i0 = r1 + (int)(item / nch) * 4, cc = (int)(item % nch);
            const int c_lo = r1 + cc * TD_CW, c_hi = (c_lo + TD_CW < n) ? c_lo + TD_CW : n;
            const double* rowp[4];
            double acc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = (i0 + q < n) ? i0 + q : n - 1;
                rowp[q] = A + (size_t)i * n;
                acc[q] = 0.0;
            }
            for (int c = c_lo + lane; c < c_hi; c += 512) {
                double a[8][4], vn[8];
                const bool pre = item == gw && c == c_lo + lane;      // (wave-uniform: the batch requested behind exchange A's post)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int cu = c + 64 * u;
                    const bool on = cu < c_hi;
                    vn[u] = on ? v_sh[cu] : 0.0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) a[u][q] = pre ? a_pre[u][q] : (on ? rowp[q][cu] : 0.0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    double sq = a[0][q] * vn[0];
#pragma unroll
                    for (int u = 1; u < 8; ++u) sq += a[u][q] * vn[u];
                    acc[q] += sq;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double sq = wave_sum(acc[q]);
                if (lane == 0 && i0 + q < n) {
                    tdp_store(B.zp + (size_t)cc * n + i0 + q, sq);
                    vav += v_sh[i0 + q] * sq;
                }
            }
        }
        TDP_STAMP(5);
        red[tid] = (lane == 0) ? vav : 0.0;
        __syncthreads();
        if (tid == 0) {
            double sacc = 0.0;
            for (int q = 0; q < TDP_T; q += 64) sacc += red[q];
            bc[1] = sacc;
        }
        __syncthreads();
        if (!tdp_exchange(B, G, gen++, bc[1], red, &dead_sh)) return;
        TDP_STAMP(6);
        const double vAv = red[0];
        __syncthreads();
        if (tid < 2 * TDP_NB) pq_sh[tid] = ((tid & (TDP_NB - 1)) < jj) ? tdp_load(B.pq + tid) : 0.0;
        __syncthreads();
        double pqs = 0.0;
        for (int k = 0; k < jj; ++k) pqs += pq_sh[k] * pq_sh[TDP_NB + k];
        t_prev = t;
        yv_prev = vAv - 2.0 * pqs;
    }
}
// the panel's rank-2 nb update of the trailing block (rows, columns >= j1), both triangles: A -= sum_k v_k w_k^T + w_k v_k^T
// (like k_td_update, the two triangles agree to rounding, not bit for bit: the mat-vec reads whole rows)
__global__ __launch_bounds__(256) void k_td_rank2k(double* __restrict__ A, int n, int j1, const double* __restrict__ V,
                                                   const double* __restrict__ W, int nb) {
    __shared__ double Vi[TDP_NB][64], Wi[TDP_NB][64], Vc[TDP_NB][64], Wc[TDP_NB][64];
    const int ti = j1 + blockIdx.y * 64, tc = j1 + blockIdx.x * 64, tid = threadIdx.x;
    for (int q = tid; q < nb * 64; q += 256) {
        const int k = q >> 6, o = q & 63;
        const int i = ti + o, c = tc + o;
        Vi[k][o] = i < n ? V[(size_t)k * n + i] : 0.0;
        Wi[k][o] = i < n ? W[(size_t)k * n + i] : 0.0;
        Vc[k][o] = c < n ? V[(size_t)k * n + c] : 0.0;
        Wc[k][o] = c < n ? W[(size_t)k * n + c] : 0.0;
    }
    __syncthreads();
    const int cx = tid & 63, iy = tid >> 6;      // thread: column cx, rows iy, iy + 4, ... (a wave shares its rows: broadcast reads)
    const int c = tc + cx;
    double s[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) s[q] = 0.0;
    for (int k = 0; k < nb; ++k) {
        const double wc = Wc[k][cx], vc = Vc[k][cx];
#pragma unroll
        for (int q = 0; q < 16; ++q) s[q] = fma(Wi[k][iy + 4 * q], vc, fma(Vi[k][iy + 4 * q], wc, s[q]));
    }
    if (c < n) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = ti + iy + 4 * q;
            if (i < n) A[(size_t)i * n + c] -= s[q];
        }
    }
}

// Z (n x k, row-major) <- Q Z with Q = H_0 H_1 ... H_{n-3}: one WAVE per column, the column in LDS,
// reflectors applied last to first; v_j is read from row j of A (shared by all waves through L2).
__global__ __launch_bounds__(256) void k_td_back(const double* __restrict__ A, const double* __restrict__ tau, int n,
                                                const double* __restrict__ Z, int k, double* __restrict__ V, int wpb) {
    extern __shared__ double zsh[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = blockIdx.x * wpb + wv;
    if (wv >= wpb || col >= k) return;
    double* z = zsh + (size_t)wv * n;
    for (int r = lane; r < n; r += 64) z[r] = Z[(long long)r * k + col];
    for (int j = n - 3; j >= 0; --j) {
        const double t = tau[j];
        if (t == 0.0) continue;
        const double* v = A + (long long)j * n;
        // lane l owns the rows r = l (mod 64) for every reflector: no z entry is ever touched by two lanes
        const int rs = j + 1 + ((lane - (j + 1)) & 63);
        double acc = 0.0;
        for (int r = rs; r < n; r += 64) acc += v[r] * z[r];
        const double s = t * wave_sum(acc);
        for (int r = rs; r < n; r += 64) z[r] -= s * v[r];
    }
    for (int r = lane; r < n; r += 64) V[(long long)r * k + col] = z[r];
}

// A <- (A + A^T) / 2: the Gram tiles above and below the diagonal come from different MFMA tiles
__global__ __launch_bounds__(256) void k_symmetrize(double* __restrict__ A, int n) {
    const long long total = (long long)n * n;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int i = (int)(e / n), j = (int)(e % n);
        if (j > i) {
            const double m = 0.5 * (A[e] + A[(long long)j * n + i]);
            A[e] = m;
            A[(long long)j * n + i] = m;
        }
    }
}

// device part of asb_sym_tridiag: d, e stay in ctx->td_work (d at 6 n, e at 7 n); host copies when the pointers are given
static int sym_tridiag(asb_ctx* ctx, double* A_dev, int64_t n64, double* d_host, double* e_host) {
    if (!ctx || n64 < 1) return ASB_ERR_ARG;
    double* A = A_dev ? A_dev : ctx->pod_g;
    if (!A) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_sym_tridiag: no matrix (run asb_pod_gram first or pass A_dev)");
    if (n64 > 46000) ASB_FAIL(ctx, ASB_ERR_LIMIT, "asb_sym_tridiag: n = %lld too large", (long long)n64);
    const int n = (int)n64;
    int rc;
    const int nch_max = (n + TD_CW - 1) / TD_CW;
    if ((rc = asb_alloc(ctx, &ctx->td_work, (size_t)8 * n))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->td_ppart, (size_t)nch_max * n))) return rc;
    double* vb0 = ctx->td_work;
    double* vb1 = vb0 + n;
    double* w = vb1 + n;
    double* p = ctx->td_ppart;                  // partial mat-vec results, one vector per column chunk
    double* xbuf = ctx->td_work + (size_t)4 * n;      // (slot 3 n, the old single p vector, stays unused)
    double* tau = xbuf + n;
    double* d = tau + n;
    double* e = d + n;
    ASB_HIP(ctx, hipMemsetAsync(ctx->td_work, 0, (size_t)8 * n * sizeof(double), ctx->stream));
    hipLaunchKernelGGL(k_symmetrize, dim3(2048), dim3(256), 0, ctx->stream, A, n);
    if (n <= 2) {
        if (!d_host || !e_host) ASB_FAIL(ctx, ASB_ERR_ARG, "the device eigen-solver needs n >= 3 (n = %d)", n);
        double h[4] = {0, 0, 0, 0};
        ASB_HIP(ctx, hipMemcpyAsync(h, A, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        d_host[0] = h[0];
        if (n == 2) { d_host[1] = h[3]; e_host[0] = h[2]; }
        ctx->td_n = n;
        return ASB_OK;
    }
    // d[0] is never touched by a reflector
    ASB_HIP(ctx, hipMemcpyAsync(d, A, sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    constexpr int R = 4;
    int nch_prev = 1;
    // panels of TDP_NB reflectors in one co-resident launch each (k_td_panel) while the trailing block is large; the
    // two-launch loop below finishes from the first column they left (js), on the explicitly updated matrix
    int js = 0;
    {
        static const int panel_min = getenv("ASB_TD_PANEL_MIN") ? atoi(getenv("ASB_TD_PANEL_MIN")) : 1536;
        // (measured at n = 4000: tails of 512 / 1536 / 2048 / 2560 columns left to the two-launch loop give 91 - 93 / 92.5 / 94.6 / 98.6 ms)
        static const int panel_tail = getenv("ASB_TD_PANEL_TAIL") ? atoi(getenv("ASB_TD_PANEL_TAIL")) : 512;
        const size_t lds = ((size_t)n + TDP_T + 4 * TDP_NB) * sizeof(double);
        if (panel_min > 0 && n >= panel_min && lds <= 150 * 1024) {
            int per_cu = 0;
            ASB_HIP(ctx, hipFuncSetAttribute((const void*)k_td_panel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ASB_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_td_panel, TDP_T, lds));
            int G = per_cu >= 1 ? ctx->n_cu : 0;
            if (G > TDP_T) G = TDP_T;
            if (G >= 16) {
                const int nch_p = nch_max;
                const size_t nd = (size_t)n * (1 + nch_p + 2 * TDP_NB) + 2 * TDP_NB;
                if ((rc = asb_alloc(ctx, &ctx->td_panel, nd))) return rc;
                if ((rc = asb_alloc(ctx, &ctx->td_rec, (size_t)3 * TDP_T + 16))) return rc;
                TdpBuf B;
                B.x = ctx->td_panel;
                B.zp = B.x + n;
                B.V = B.zp + (size_t)nch_p * n;
                B.W = B.V + (size_t)TDP_NB * n;
                B.pq = B.W + (size_t)TDP_NB * n;
                B.rec = ctx->td_rec;
                B.flags = reinterpret_cast<unsigned*>(ctx->td_rec + 3 * TDP_T);
                B.tlog = nullptr;
                B.test_stall = getenv("ASB_TD_TEST_STALL") ? atoi(getenv("ASB_TD_TEST_STALL")) : 0;
                // the matrix as it is now, for the case that a panel's exchange times out: the two-launch loop then starts over on it
                if ((rc = asb_alloc(ctx, &ctx->td_backup, (size_t)n * n))) return rc;
                ASB_HIP(ctx, hipMemcpyAsync(ctx->td_backup, A, (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
                static unsigned long long* tlog_dev = nullptr;
                if (getenv("ASB_DEBUG_TD") && !tlog_dev) (void)hipMalloc((void**)&tlog_dev, (TDP_NB + 1) * 8 * sizeof(unsigned long long));
                ASB_HIP(ctx, hipMemsetAsync(B.flags, 0, 16 * sizeof(unsigned long long) / 2, ctx->stream));
                const int tail = panel_tail < 64 ? 64 : panel_tail;
                while (n - js > tail + TDP_NB) {
                    ASB_HIP(ctx, hipMemsetAsync(B.rec, 0xFF, (size_t)3 * TDP_T * sizeof(unsigned long long), ctx->stream));
                    B.tlog = (tlog_dev && js == 1024) ? tlog_dev : nullptr;
                    hipLaunchKernelGGL(k_td_panel, dim3(G), dim3(TDP_T), lds, ctx->stream, A, n, js, TDP_NB, B, tau, d, e);
                    const int j1 = js + TDP_NB, tiles = (n - j1 + 63) / 64;
                    hipLaunchKernelGGL(k_td_rank2k, dim3(tiles, tiles), dim3(256), 0, ctx->stream, A, n, j1, B.V, B.W, TDP_NB);
                    js = j1;
                }
                ASB_CHECK_LAUNCH(ctx);
                unsigned fl = 0;
                ASB_HIP(ctx, hipMemcpyAsync(&fl, B.flags, sizeof(fl), hipMemcpyDeviceToHost, ctx->stream));
                ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
                if (tlog_dev) {
                    unsigned long long h[(TDP_NB + 1) * 8];
                    (void)hipMemcpy(h, tlog_dev, sizeof(h), hipMemcpyDeviceToHost);
                    for (int jj = 0; jj < TDP_NB; jj += 5)
                        fprintf(stderr, "[asb] panel at 1024, reflector %2d (10 ns ticks): phase I %llu | exchange A %llu | stage v %llu | dots %llu | mat-vec %llu | exchange B %llu | round %llu\n", jj,
                                h[jj * 8 + 1] - h[jj * 8], h[jj * 8 + 2] - h[jj * 8 + 1], h[jj * 8 + 3] - h[jj * 8 + 2], h[jj * 8 + 4] - h[jj * 8 + 3],
                                h[jj * 8 + 5] - h[jj * 8 + 4], h[jj * 8 + 6] - h[jj * 8 + 5], h[(jj + 1) * 8] - h[jj * 8]);
                }
                if (fl) {
                    // a block of some panel never arrived (not co-resident, or lost): nothing of the panels is trusted
                    ASB_HIP(ctx, hipMemcpyAsync(A, ctx->td_backup, (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
                    ASB_HIP(ctx, hipMemsetAsync(ctx->td_work, 0, (size_t)8 * n * sizeof(double), ctx->stream));
                    ASB_HIP(ctx, hipMemcpyAsync(d, A, sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
                    js = 0;
                    ctx->n_coop_fallbacks++;
                    if (getenv("ASB_DEBUG_TD") || B.test_stall)
                        fprintf(stderr, "[asb] tridiagonalisation: the panel kernel's grid exchange timed out; two-launch loop from the first column\n");
                }
            }
        }
    }
    for (int j = js - 1; j <= n - 3; ++j) {
        const int fresh = (js > 0 && j == js - 1) ? 1 : 0;
        double* vcur = (j & 1) ? vb1 : vb0;          // j = -1 -> vb1 (unused)
        double* vnext = ((j + 1) & 1) ? vb1 : vb0;
        // k_td_small(j) writes d[j+1]; for j = -1 it would overwrite d[0] with the same value A[0][0]
        static const int reg = getenv("ASB_TD_SMALL_REG") ? atoi(getenv("ASB_TD_SMALL_REG")) : 1;
        const int rows_left = n - (j + 1);
        if (reg && rows_left <= 4 * TD_T)
            hipLaunchKernelGGL((k_td_small_reg<4>), dim3(1), dim3(TD_T), 0, ctx->stream, A, n, j, p, nch_prev, vcur, w, vnext, tau, d, e, fresh);
        else if (reg && rows_left <= 8 * TD_T)
            hipLaunchKernelGGL((k_td_small_reg<8>), dim3(1), dim3(TD_T), 0, ctx->stream, A, n, j, p, nch_prev, vcur, w, vnext, tau, d, e, fresh);
        else
            hipLaunchKernelGGL(k_td_small, dim3(1), dim3(TD_T), 0, ctx->stream, A, n, j, p, nch_prev, vcur, w, vnext, xbuf, tau, d, e, fresh);
        if (j + 1 <= n - 3) {
            const int r1 = j + 2, rows = n - r1;
            const int nch = (rows + TD_CW - 1) / TD_CW;
            const long long items = (long long)((rows + R - 1) / R) * nch;
            long long gridl = (items + 3) / 4;
            int grid = (int)(gridl > 4096 ? 4096 : gridl);
            if (grid < 1) grid = 1;
            static const int tdv = getenv("ASB_TD_VARIANT") ? atoi(getenv("ASB_TD_VARIANT")) : 1;
            if (tdv == 1)          // four column positions per lane in flight (round 4: POD 237 -> 233 ms on one box; 0: two)
                hipLaunchKernelGGL((k_td_update<R, 4>), dim3(grid), dim3(256), 0, ctx->stream, A, n, r1, (j >= 0 && !fresh) ? 1 : 0, vcur, w, vnext,
                                   tau + (j + 1), p, nch);
            else
            hipLaunchKernelGGL((k_td_update<R>), dim3(grid), dim3(256), 0, ctx->stream, A, n, r1, (j >= 0 && !fresh) ? 1 : 0, vcur, w, vnext,
                               tau + (j + 1), p, nch);
            nch_prev = nch;
        }
    }
    ASB_CHECK_LAUNCH(ctx);
    ctx->td_n = n;
    if (d_host && e_host) {
        ASB_HIP(ctx, hipMemcpyAsync(d_host, d, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        ASB_HIP(ctx, hipMemcpyAsync(e_host, e, (size_t)(n - 1) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return ASB_OK;
}

extern "C" int asb_sym_tridiag(asb_ctx* ctx, double* A_dev, int64_t n64, double* d_host, double* e_host) {
    if (!ctx || !d_host || !e_host || n64 < 1) return ASB_ERR_ARG;
    return sym_tridiag(ctx, A_dev, n64, d_host, e_host);
}

// ---- blocked back-transformation (round 3): the reflectors in groups of TB_NB, each group as I - V T V^T (compact WY), so
// that Q Z is a sequence of GEMMs instead of 4000 dependent wave reductions per column (42 ms at n = 4000, k = 288).
#define TB_NB 64
// T factors of all groups in one launch (block b: reflectors j0 = b TB_NB ...): G = V^T V in LDS, then LAPACK's larft
// recurrence T_ii = tau_i, T[0:i, i] = -tau_i T[0:i, 0:i] G[0:i, i] (a reflector with tau = 0 is the identity)
// G_b = V_b^T V_b of every group, 16 blocks per group (each a slice of the 64 x 64 pairs): one WAVE per pair (i, m >= i), its
// lanes along the rows (coalesced), ordered wave sum.  (Round 3: this part alone took 6.6 ms when each group was one block.)
__global__ __launch_bounds__(256) void k_td_wy_g(const double* __restrict__ A, int n, int nrefl, double* __restrict__ G_all) {
    const int b = blockIdx.x, sl = blockIdx.y, j0 = b * TB_NB, tid = threadIdx.x;
    const int nbb = nrefl - j0 < TB_NB ? nrefl - j0 : TB_NB;
    const int lane = tid & 63, wv = tid >> 6;
    double* G = G_all + (size_t)b * TB_NB * TB_NB;
    for (int e = sl * 4 + wv; e < TB_NB * TB_NB; e += 4 * gridDim.y) {
        const int i = e / TB_NB, m = e % TB_NB;
        double acc = 0.0;
        if (i < nbb && m < nbb && m >= i) {
            const double* vi = A + (long long)(j0 + i) * n;
            const double* vm = A + (long long)(j0 + m) * n;
            for (int c = j0 + m + 1 + lane; c < n; c += 64) acc += vi[c] * vm[c];
            acc = wave_sum(acc);
        }
        if (lane == 0) G[e] = acc;
    }
}
__global__ __launch_bounds__(256) void k_td_wy_t(const double* __restrict__ G_all, const double* __restrict__ tau, int n, int nrefl,
                                                 double* __restrict__ T_all) {
    __shared__ double G[TB_NB][TB_NB + 1];
    __shared__ double Ts[TB_NB][TB_NB + 1];
    __shared__ double col[TB_NB];
    const int b = blockIdx.x, j0 = b * TB_NB, tid = threadIdx.x;
    const int nbb = nrefl - j0 < TB_NB ? nrefl - j0 : TB_NB;
    for (int e = tid; e < TB_NB * TB_NB; e += 256) {
        Ts[e / TB_NB][e % TB_NB] = 0.0;
        G[e / TB_NB][e % TB_NB] = G_all[(size_t)b * TB_NB * TB_NB + e];
    }
    __syncthreads();
    for (int i = 0; i < nbb; ++i) {
        const double t = tau[j0 + i];
        if (tid < i) {                           // col = T[0:i, 0:i] G[0:i, i]
            double acc = 0.0;
            for (int q = tid; q < i; ++q) acc += Ts[tid][q] * G[q][i];
            col[tid] = acc;
        }
        __syncthreads();
        if (tid < i) Ts[tid][i] = -t * col[tid];
        if (tid == i) Ts[i][i] = t;
        __syncthreads();
    }
    double* T = T_all + (size_t)b * TB_NB * TB_NB;
    for (int e = tid; e < TB_NB * TB_NB; e += 256) T[e] = Ts[e / TB_NB][e % TB_NB];
}
// the group's reflectors as dense panels: Vp (TB_NB x n) and its transpose VpT (n x TB_NB)
__global__ __launch_bounds__(256) void k_td_wy_panel(const double* __restrict__ A, int n, int nrefl, int j0, double* __restrict__ Vp,
                                                     double* __restrict__ VpT) {
    const long long total = (long long)TB_NB * n;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int i = (int)(e / n), c = (int)(e % n);
        const double v = (j0 + i < nrefl && c >= j0 + i + 1) ? A[(long long)(j0 + i) * n + c] : 0.0;
        Vp[e] = v;
        VpT[(long long)c * TB_NB + i] = v;
    }
}
static int sym_backtransform_blocked(asb_ctx* ctx, const double* A, int n, const double* Z, int k, double* V) {
    const double* tau = ctx->td_work + (size_t)5 * n;
    const int nrefl = n - 2, nblk = (nrefl + TB_NB - 1) / TB_NB;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->td_wy, (size_t)2 * nblk * TB_NB * TB_NB + (size_t)2 * TB_NB * n + (size_t)2 * TB_NB * k))) return rc;
    double* Gw = ctx->td_wy + (size_t)nblk * TB_NB * TB_NB + (size_t)2 * TB_NB * n + (size_t)2 * TB_NB * k;
    double* T_all = ctx->td_wy;
    double* Vp = T_all + (size_t)nblk * TB_NB * TB_NB;
    double* VpT = Vp + (size_t)TB_NB * n;
    double* W1 = VpT + (size_t)TB_NB * n;
    double* W2 = W1 + (size_t)TB_NB * k;
    hipLaunchKernelGGL(k_td_wy_g, dim3(nblk, 16), dim3(256), 0, ctx->stream, A, n, nrefl, Gw);
    hipLaunchKernelGGL(k_td_wy_t, dim3(nblk), dim3(256), 0, ctx->stream, Gw, tau, n, nrefl, T_all);
    ASB_HIP(ctx, hipMemcpyAsync(V, Z, (size_t)n * k * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    for (int b = nblk - 1; b >= 0; --b) {           // Q = Q_0 Q_1 ... : the last group acts first
        hipLaunchKernelGGL(k_td_wy_panel, dim3(256), dim3(256), 0, ctx->stream, A, n, nrefl, b * TB_NB, Vp, VpT);
        if ((rc = asb_gemm_nn(ctx, Vp, n, V, k, W1, k, TB_NB, k, n, 1.0, 0.0))) return rc;                       // W1 = V^T Z
        if ((rc = asb_gemm_nn(ctx, T_all + (size_t)b * TB_NB * TB_NB, TB_NB, W1, k, W2, k, TB_NB, k, TB_NB, 1.0, 0.0))) return rc;
        if ((rc = asb_gemm_nn(ctx, VpT, TB_NB, W2, k, V, k, n, k, TB_NB, -1.0, 1.0))) return rc;                 // Z -= V (T W1)
    }
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

// Z (device, n x k row-major) -> V = Q Z (device, n x k) with the reflectors asb_sym_tridiag left in A
static int sym_backtransform_dev(asb_ctx* ctx, const double* A, int n, const double* Z, int k, double* V) {
    static const int blocked = getenv("ASB_BACKTRANSFORM_BLOCKED") ? atoi(getenv("ASB_BACKTRANSFORM_BLOCKED")) : 1;
    if (blocked && !(n & 1) && !(k & 1) && n >= 4 * TB_NB) return sym_backtransform_blocked(ctx, A, n, Z, k, V);
    const double* tau = ctx->td_work + (size_t)5 * n;
    int wpb = (int)((size_t)(160 * 1024 - 1024) / ((size_t)n * sizeof(double)));
    if (wpb < 1) ASB_FAIL(ctx, ASB_ERR_LIMIT, "asb_sym_backtransform: n = %d does not fit one LDS column", n);
    if (wpb > 4) wpb = 4;
    const size_t lds = (size_t)wpb * n * sizeof(double);
    if (lds > 48 * 1024)
        ASB_HIP(ctx, hipFuncSetAttribute((const void*)k_td_back, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_td_back, dim3((k + wpb - 1) / wpb), dim3(256), lds, ctx->stream, A, tau, n, Z, k, V, wpb);
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

// The whole symmetric eigen-problem on the device (n >= 3): Householder tridiagonalisation, bisection + inverse iteration
// on the tridiagonal matrix (asb_smalldense.hip), back-transformation.  A_dev (NULL: the Gram matrix of asb_pod_gram) is
// overwritten.  lam_host (n): ALL eigenvalues, descending.  The k leading eigenvectors stay on the device (n x k,
// row-major) for asb_pod_basis_dev; V_host (optional, n x k) receives a copy.  *n_bad (optional): vectors whose inverse
// iteration missed its growth criterion (0 in every case seen; they are still normalised iterates).
extern "C" int asb_sym_eig_topk(asb_ctx* ctx, double* A_dev, int64_t n64, int64_t k64, double* lam_host, double* V_host,
                                int64_t* n_bad) {
    if (!ctx || !lam_host || n64 < 3 || k64 < 1 || k64 > n64) return ASB_ERR_ARG;
    int rc;
    if ((rc = sym_tridiag(ctx, A_dev, n64, nullptr, nullptr))) return rc;
    const int n = (int)n64, k = (int)k64;
    const double* A = A_dev ? A_dev : ctx->pod_g;
    if ((rc = asb_alloc(ctx, &ctx->td_z, (size_t)2 * n * k))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->eig_lam, (size_t)n))) return rc;
    double* Z = ctx->td_z;
    double* V = Z + (size_t)n * k;
    const double* d = ctx->td_work + (size_t)6 * n;
    const double* e = ctx->td_work + (size_t)7 * n;
    int bad = 0;
    if ((rc = asb_tri_eig_dev(ctx, d, e, n, k, ctx->eig_lam, Z, &bad))) return rc;
    if ((rc = sym_backtransform_dev(ctx, A, n, Z, k, V))) return rc;
    ctx->eig_v = V;
    ctx->eig_n = n;
    ctx->eig_k = k;
    ASB_HIP(ctx, hipMemcpyAsync(lam_host, ctx->eig_lam, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (V_host) ASB_HIP(ctx, hipMemcpyAsync(V_host, V, (size_t)n * k * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n_bad) *n_bad = bad;
    return ASB_OK;
}

extern "C" int asb_sym_backtransform(asb_ctx* ctx, const double* A_dev, int64_t n64, const double* Z_host, int64_t k64,
                                     double* V_host) {
    if (!ctx || !Z_host || !V_host || n64 < 1 || k64 < 1) return ASB_ERR_ARG;
    const double* A = A_dev ? A_dev : ctx->pod_g;
    if (!A || !ctx->td_work || ctx->td_n != n64)
        ASB_FAIL(ctx, ASB_ERR_ARG, "asb_sym_backtransform: asb_sym_tridiag has not been run on an n = %lld matrix", (long long)n64);
    const int n = (int)n64, k = (int)k64;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->td_z, (size_t)2 * n * k))) return rc;
    double* Z = ctx->td_z;
    double* V = Z + (size_t)n * k;
    ASB_HIP(ctx, hipMemcpyAsync(Z, Z_host, (size_t)n * k * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = sym_backtransform_dev(ctx, A, n, Z, k, V))) return rc;
    ASB_HIP(ctx, hipMemcpyAsync(V_host, V, (size_t)n * k * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ASB_OK;
}
