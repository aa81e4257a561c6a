// Greedy deflation ("PCA") kernels shared by the residual and the panel (projection) paths --
// posComponents.extract_k_components, snapbases/posComponents.py:67-122 of the reference.
// gfx950 (MI355X) only.
//
// Data layout: residual rows r = 3*v + d of Fp doubles (vertex-major), so one vertex's
// 3 x F trajectory is a contiguous 24*Fp-byte run that a group of T threads streams with
// 16-byte loads, keeps in registers across the dot product and writes back once:
// one HBM read + one HBM write of R per component (the reference makes ~6 passes).
#pragma once
#include "asb_common.h"

#include <cmath>
#include <cstring>



// --------------------------------------------------------------------------------------
// k_stream: the dominant kernel.  For every vertex of the shard:
//   UPDATE: dot_d = w . R[v,d,:]  ->  c[v,d] = dot_d * s[v] / |w|^2   (:101-105)
//           R[v,d,:] -= w * c[v,d]                                      (:111)
//   always: energy[v] = sum R[v,:,:]^2 (:78-80), per-block (max, first index, sum).
// T threads per vertex (T = 64..1024), E2 double2 per thread per row.
// --------------------------------------------------------------------------------------
template <int T, int E2, bool UPDATE>
__global__ __launch_bounds__((T >= 256 ? T : 256)) void k_stream(
    double* __restrict__ R, const double* __restrict__ wk, double* __restrict__ scal_k,
    const double* __restrict__ s, double* __restrict__ ck_out, double* __restrict__ energy,
    double* __restrict__ pmax, long long* __restrict__ pidx, double* __restrict__ psum,
    long long n_loc, int F2, const PanelState* __restrict__ panel) {
    if (panel != nullptr) {
        if (panel->done) return;                      // panel path: this step was not committed
        if (panel->n_cand < n_loc) n_loc = panel->n_cand;
    }
    constexpr int BLOCK = (T >= 256 ? T : 256);
    constexpr int VPB = BLOCK / T;
    constexpr int NW = T / 64;
    const int tid = threadIdx.x;
    const int g = tid / T, t = tid % T;
    const int wig = t >> 6, lane = tid & 63;
    __shared__ double red[VPB][NW][4];
    __shared__ double lead_e[VPB];
    __shared__ long long lead_i[VPB];
    __shared__ double lead_s[VPB];

    double2 w[E2];
    double wn2 = 1.0;
    if (UPDATE) {
#pragma unroll
        for (int i = 0; i < E2; ++i) {
            const int j = t + i * T;
            w[i] = (j < F2) ? reinterpret_cast<const double2*>(wk)[j] : make_double2(0.0, 0.0);
        }
        wn2 = scal_k[1];
    }
    double bmax = -1.0, bsum = 0.0;
    long long bidx = 0x7fffffffffffffffLL;

    for (long long base = (long long)blockIdx.x * VPB; base < n_loc; base += (long long)gridDim.x * VPB) {
        const long long v = base + g;
        const bool valid = v < n_loc;
        double2* row = reinterpret_cast<double2*>(R) + (valid ? v : 0) * 3 * (long long)F2;
        double2 x[3][E2];
        double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int i = 0; i < E2; ++i) {
                const int j = t + i * T;
                x[d][i] = (valid && j < F2) ? row[(long long)d * F2 + j] : make_double2(0.0, 0.0);
                if (UPDATE) acc[d] += x[d][i].x * w[i].x + x[d][i].y * w[i].y;
            }
        double c[3] = {0.0, 0.0, 0.0};
        if (UPDATE) {
#pragma unroll
            for (int d = 0; d < 3; ++d) acc[d] = wave_sum(acc[d]);
            if (NW > 1) {
                __syncthreads();
                if (lane == 0) {
                    red[g][wig][0] = acc[0];
                    red[g][wig][1] = acc[1];
                    red[g][wig][2] = acc[2];
                }
                __syncthreads();
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    double sum = 0.0;
#pragma unroll
                    for (int q = 0; q < NW; ++q) sum += red[g][q][d];
                    acc[d] = sum;
                }
            }
            const double sv = (s != nullptr && valid) ? s[v] : 1.0;
#pragma unroll
            for (int d = 0; d < 3; ++d) c[d] = (acc[d] * sv) / wn2;
        }
        double e = 0.0;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int i = 0; i < E2; ++i) {
                const int j = t + i * T;
                if (UPDATE) {
                    x[d][i].x -= w[i].x * c[d];
                    x[d][i].y -= w[i].y * c[d];
                    if (valid && j < F2) row[(long long)d * F2 + j] = x[d][i];
                }
                e += x[d][i].x * x[d][i].x + x[d][i].y * x[d][i].y;
            }
        e = wave_sum(e);
        if (NW > 1) {
            __syncthreads();
            if (lane == 0) red[g][wig][3] = e;
            __syncthreads();
            double sum = 0.0;
#pragma unroll
            for (int q = 0; q < NW; ++q) sum += red[g][q][3];
            e = sum;
        }
        if (t == 0 && valid) {
            energy[v] = e;
            if (UPDATE) {
                ck_out[v * 3 + 0] = c[0];
                ck_out[v * 3 + 1] = c[1];
                ck_out[v * 3 + 2] = c[2];
            }
            bsum += e;
            if (am_better(e, v, bmax, bidx)) {
                bmax = e;
                bidx = v;
            }
        }
    }
    if (t == 0) {
        lead_e[g] = bmax;
        lead_i[g] = bidx;
        lead_s[g] = bsum;
    }
    __syncthreads();
    if (tid == 0) {
        double be = lead_e[0], bs = lead_s[0];
        long long bi = lead_i[0];
#pragma unroll
        for (int q = 1; q < VPB; ++q) {
            bs += lead_s[q];
            if (am_better(lead_e[q], lead_i[q], be, bi)) {
                be = lead_e[q];
                bi = lead_i[q];
            }
        }
        pmax[blockIdx.x] = be;
        pidx[blockIdx.x] = bi;
        psum[blockIdx.x] = bs;
    }
}

// --------------------------------------------------------------------------------------
// Symmetric 3x3 eigen-solve (cyclic Jacobi, f64): largest eigenvalue and its vector.
// Replaces LAPACK gesdd on the 3 x F slab (:83): sigma_1^2 / u_1 of S S^T.
// --------------------------------------------------------------------------------------
#define ASB_JROT(app, aqq, apq, arp, arq, vp0, vq0, vp1, vq1, vp2, vq2)                       \
    if (apq != 0.0) {                                                                         \
        const double theta = (aqq - app) / (2.0 * apq);                                       \
        const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0)); \
        const double cc = 1.0 / sqrt(tt * tt + 1.0), ss = tt * cc;                            \
        const double napp = app - tt * apq, naqq = aqq + tt * apq;                            \
        const double nrp = cc * arp - ss * arq, nrq = ss * arp + cc * arq;                    \
        app = napp; aqq = naqq; apq = 0.0; arp = nrp; arq = nrq;                              \
        double tp, tq;                                                                        \
        tp = cc * vp0 - ss * vq0; tq = ss * vp0 + cc * vq0; vp0 = tp; vq0 = tq;               \
        tp = cc * vp1 - ss * vq1; tq = ss * vp1 + cc * vq1; vp1 = tp; vq1 = tq;               \
        tp = cc * vp2 - ss * vq2; tq = ss * vp2 + cc * vq2; vp2 = tp; vq2 = tq;               \
    }

__host__ __device__ inline void eig3_top(double a00, double a01, double a02, double a11, double a12, double a22,
                         double& lam, double& u0, double& u1, double& u2) {
    // work on A / max|a_ij|: the convergence test squares the entries, which must neither overflow nor flush to zero
    // for snapshots scaled like 1e+-120
    const double sc = fmax(fmax(fabs(a00), fabs(a11)), fmax(fabs(a22), fmax(fabs(a01), fmax(fabs(a02), fabs(a12)))));
    const bool scaled = sc > 0.0 && sc < 1.0e300;
    if (scaled) {
        const double is = 1.0 / sc;
        a00 *= is; a01 *= is; a02 *= is; a11 *= is; a12 *= is; a22 *= is;
    }
    // eigenvector matrix V, column j = (v0j, v1j, v2j)
    double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = a01 * a01 + a02 * a02 + a12 * a12;
        const double dia = a00 * a00 + a11 * a11 + a22 * a22;
        if (off == 0.0 || off <= 1e-40 * dia) break;
        // (p,q) = (0,1): third index r = 2 couples through a02 (rp) and a12 (rq)
        ASB_JROT(a00, a11, a01, a02, a12, v00, v01, v10, v11, v20, v21)
        // (p,q) = (0,2): r = 1, rp = a01, rq = a12
        ASB_JROT(a00, a22, a02, a01, a12, v00, v02, v10, v12, v20, v22)
        // (p,q) = (1,2): r = 0, rp = a01, rq = a02
        ASB_JROT(a11, a22, a12, a01, a02, v01, v02, v11, v12, v21, v22)
    }
    lam = a00; u0 = v00; u1 = v10; u2 = v20;
    if (a11 > lam) { lam = a11; u0 = v01; u1 = v11; u2 = v21; }
    if (a22 > lam) { lam = a22; u0 = v02; u1 = v12; u2 = v22; }
    // canonical sign (LAPACK's is arbitrary): largest-magnitude entry positive
    const double m0 = fabs(u0), m1 = fabs(u1), m2 = fabs(u2);
    const double lead = (m0 >= m1 && m0 >= m2) ? u0 : (m1 >= m2 ? u1 : u2);
    if (lead < 0.0) { u0 = -u0; u1 = -u1; u2 = -u2; }
    const double nn = sqrt(u0 * u0 + u1 * u1 + u2 * u2);
    u0 /= nn; u1 /= nn; u2 /= nn;
    if (scaled) lam *= sc;
}

// Largest eigen-pair of a symmetric 3 x 3 matrix without the Jacobi sweeps (~6 sweeps x 3 rotations of dependent
// sqrt / divide chains, about 5 us on one lane): trigonometric root of the characteristic polynomial, eigenvector from
// the largest cross product of two rows of A - lambda I, one to three Rayleigh-quotient refinements, and a residual test
// |A u - lambda u| <= 4 eps |A|; anything that fails it (near-degenerate top eigenvalues) goes to eig3_top.
__host__ __device__ inline void eig3_top_fast(double a00, double a01, double a02, double a11, double a12, double a22,
                                              double& lam, double& u0, double& u1, double& u2) {
    const double sc = fmax(fmax(fabs(a00), fabs(a11)), fmax(fabs(a22), fmax(fabs(a01), fmax(fabs(a02), fabs(a12)))));
    if (!(sc > 0.0) || !(sc < 1.0e300)) { eig3_top(a00, a01, a02, a11, a12, a22, lam, u0, u1, u2); return; }
    const double is = 1.0 / sc;
    const double b00 = a00 * is, b01 = a01 * is, b02 = a02 * is, b11 = a11 * is, b12 = a12 * is, b22 = a22 * is;
    const double q = (b00 + b11 + b22) / 3.0;
    const double p1 = b01 * b01 + b02 * b02 + b12 * b12;
    const double d0 = b00 - q, d1 = b11 - q, d2 = b22 - q;
    const double p = sqrt((d0 * d0 + d1 * d1 + d2 * d2 + 2.0 * p1) / 6.0);
    bool ok = p > 0.0;
    double l = q, x = 0.0, y = 0.0, z = 0.0;
    if (ok) {
        const double ip = 1.0 / p;
        const double c00 = d0 * ip, c11 = d1 * ip, c22 = d2 * ip, c01 = b01 * ip, c02 = b02 * ip, c12 = b12 * ip;
        double r = 0.5 * (c00 * (c11 * c22 - c12 * c12) - c01 * (c01 * c22 - c12 * c02) + c02 * (c01 * c12 - c11 * c02));
        r = fmin(1.0, fmax(-1.0, r));
        l = q + 2.0 * p * cos(acos(r) / 3.0);
        for (int it = 0; it < 3 && ok; ++it) {
            const double m00 = b00 - l, m11 = b11 - l, m22 = b22 - l;
            // cross products of the rows of B - l I
            const double x0 = b01 * b12 - b02 * m11, y0 = b02 * b01 - m00 * b12, z0 = m00 * m11 - b01 * b01;      // r0 x r1
            const double x1 = b01 * m22 - b02 * b12, y1 = b02 * b02 - m00 * m22, z1 = m00 * b12 - b01 * b02;      // r0 x r2
            const double x2 = m11 * m22 - b12 * b12, y2 = b12 * b02 - b01 * m22, z2 = b01 * b12 - m11 * b02;      // r1 x r2
            const double n0 = x0 * x0 + y0 * y0 + z0 * z0, n1 = x1 * x1 + y1 * y1 + z1 * z1, n2 = x2 * x2 + y2 * y2 + z2 * z2;
            double nn;
            if (n0 >= n1 && n0 >= n2) { x = x0; y = y0; z = z0; nn = n0; }
            else if (n1 >= n2) { x = x1; y = y1; z = z1; nn = n1; }
            else { x = x2; y = y2; z = z2; nn = n2; }
            if (!(nn > 1.0e-20)) { ok = false; break; }        // rank(B - l I) < 2: the top eigenvalue is (nearly) double
            const double inn = 1.0 / sqrt(nn);
            x *= inn; y *= inn; z *= inn;
            const double ln = x * (b00 * x + b01 * y + b02 * z) + y * (b01 * x + b11 * y + b12 * z) + z * (b02 * x + b12 * y + b22 * z);
            // the trigonometric root is already good to a few ulps unless the top eigenvalues are close: one refinement
            // that moves it by no more than rounding ends the loop (the residual test below still decides)
            const bool settled = fabs(ln - l) <= 8.9e-16 * fabs(ln);
            l = ln;
            if (settled) break;
        }
    }
    if (ok) {
        const double r0 = b00 * x + b01 * y + b02 * z - l * x, r1 = b01 * x + b11 * y + b12 * z - l * y,
                     r2 = b02 * x + b12 * y + b22 * z - l * z;
        // is l really the LARGEST eigenvalue?  trace and the 2 x 2 minors give the other two: both must be <= l
        const double tr = b00 + b11 + b22 - l;                       // l2 + l3
        const double mm = b00 * b11 - b01 * b01 + b00 * b22 - b02 * b02 + b11 * b22 - b12 * b12 - l * tr;      // l2 l3
        const double disc = tr * tr - 4.0 * mm;
        const double l2 = 0.5 * (tr + sqrt(fmax(disc, 0.0)));
        ok = fmax(fabs(r0), fmax(fabs(r1), fabs(r2))) <= 1.0e-15 && l2 <= l * (1.0 - 1.0e-6);
    }
    if (!ok) { eig3_top(a00, a01, a02, a11, a12, a22, lam, u0, u1, u2); return; }
    const double m0 = fabs(x), m1 = fabs(y), m2 = fabs(z);
    const double lead = (m0 >= m1 && m0 >= m2) ? x : (m1 >= m2 ? y : z);
    if (lead < 0.0) { x = -x; y = -y; z = -z; }
    lam = l * sc; u0 = x; u1 = y; u2 = z;
}

// Reduce the per-block partial records of the last k_stream pass (one block).
// Result in every thread: (be, bi) winner, bs sum of energies.
__device__ inline void reduce_partials(const double* pmax, const long long* pidx, const double* psum,
                                int nblk, double* sh_d, long long* sh_i, double& be, long long& bi,
                                double& bs) {
    const int tid = threadIdx.x, nt = blockDim.x;
    double e = -1.0, s = 0.0;
    long long ix = 0x7fffffffffffffffLL;
    for (int b = tid; b < nblk; b += nt) {
        s += psum[b];
        if (am_better(pmax[b], pidx[b], e, ix)) {
            e = pmax[b];
            ix = pidx[b];
        }
    }
    sh_d[tid] = e;
    sh_d[nt + tid] = s;
    sh_i[tid] = ix;
    __syncthreads();
    for (int o = nt >> 1; o > 0; o >>= 1) {
        if (tid < o) {
            sh_d[nt + tid] += sh_d[nt + tid + o];
            if (am_better(sh_d[tid + o], sh_i[tid + o], sh_d[tid], sh_i[tid])) {
                sh_d[tid] = sh_d[tid + o];
                sh_i[tid] = sh_i[tid + o];
            }
        }
        __syncthreads();
    }
    be = sh_d[0];
    bi = sh_i[0];
    bs = sh_d[nt];
    __syncthreads();
}

// --------------------------------------------------------------------------------------
// k_local_best: shard winner + its slab -> exchange record; local ||R||^2 of comp k-1.
// --------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_local_best(const double* __restrict__ R,
                                                    const double* pmax, const long long* pidx,
                                                    const double* psum, int nblk, long long v0,
                                                    int Fp, double* __restrict__ rec,
                                                    double* __restrict__ scal, long long k, long long forced,
                                                    long long n_loc) {
    __shared__ double sh_d[512];
    __shared__ long long sh_i[256];
    double be, bs;
    long long bi;
    reduce_partials(pmax, pidx, psum, nblk, sh_d, sh_i, be, bi, bs);
    if (forced >= 0) {       // the caller names the row ('pca_blocks'): its owner wins the exchange, the others abstain
        const bool mine = forced >= v0 && forced < v0 + n_loc;
        be = mine ? 1.0e300 : -1.0;
        bi = mine ? forced - v0 : 0;
    }
    if (threadIdx.x == 0) {
        rec[0] = be;
        rec[1] = __longlong_as_double(v0 + bi);
        if (k > 0) scal[(k - 1) * 4 + 3] = bs;
    }
    const double* slab = R + bi * 3 * (long long)Fp;
    for (int j = threadIdx.x; j < 3 * Fp; j += blockDim.x) rec[2 + j] = slab[j];
}

// --------------------------------------------------------------------------------------
// k_pick: winner over the records (or over this shard's partials when recs == nullptr),
// rank-1 SVD of its slab, w_k (+ the +-projection test of support='local'), scalars.
// One block of 256 threads.  k == K: only finalises the local norm of component K-1.
// --------------------------------------------------------------------------------------
// Panel mode (panel != nullptr): R is the compact candidate buffer, partial indices are
// candidate slots mapped to global vertex ids by cand_idx; the pick is only committed while
// the best candidate provably beats every non-candidate (energy > theta + margin).
static __global__ __launch_bounds__(256) void k_pick(const double* __restrict__ R, const double* pmax,
                                              const long long* pidx, const double* psum, int nblk,
                                              const double* __restrict__ recs, int n_rec,
                                              long long xlen, long long v0, int F, int Fp,
                                              double* __restrict__ W, double* __restrict__ scal,
                                              long long k, long long K, int local_mode,
                                              PanelState* __restrict__ panel,
                                              const long long* __restrict__ cand_idx, long long k_panel0,
                                              long long forced) {
    __shared__ double sh_d[512];
    __shared__ long long sh_i[256];
    __shared__ double u_sh[4];
    const int tid = threadIdx.x, nt = blockDim.x;
    const double* slab;
    long long gidx;
    if (panel != nullptr) {
        if (panel->done) return;
        double be, bs;
        long long bi;
        reduce_partials(pmax, pidx, psum, nblk, sh_d, sh_i, be, bi, bs);
        if (!(be > panel->theta + panel->margin) || bi >= panel->n_cand) {
            if (tid == 0) panel->done = 1;
            return;
        }
        slab = R + bi * 3 * (long long)Fp;
        gidx = cand_idx[bi];
    } else if (recs == nullptr) {
        double be, bs;
        long long bi;
        reduce_partials(pmax, pidx, psum, nblk, sh_d, sh_i, be, bi, bs);
        if (tid == 0 && k > 0) scal[(k - 1) * 4 + 3] = bs;
        if (k >= K) return;
        if (forced >= 0) bi = forced - v0;      // the caller names the row ('pca_blocks')
        slab = R + bi * 3 * (long long)Fp;
        gidx = v0 + bi;
    } else {
        if (k >= K) return;
        double be = -1.0;
        long long bi = 0x7fffffffffffffffLL;
        int bw = 0;
        for (int r = 0; r < n_rec; ++r) {   // n_rec <= #GPUs: every thread scans
            const double e = recs[r * xlen];
            const long long ix = __double_as_longlong(recs[r * xlen + 1]);
            if (am_better(e, ix, be, bi)) { be = e; bi = ix; bw = r; }
        }
        slab = recs + bw * xlen + 2;
        gidx = bi;
    }
    // Gram of the 3 x F slab
    double gsum[6] = {0, 0, 0, 0, 0, 0};
    for (int f = tid; f < F; f += nt) {
        const double a = slab[f], b = slab[Fp + f], c = slab[2 * Fp + f];
        gsum[0] += a * a; gsum[1] += a * b; gsum[2] += a * c;
        gsum[3] += b * b; gsum[4] += b * c; gsum[5] += c * c;
    }
    block_sum<6>(gsum, sh_d);
    if (tid == 0) {
        double lam, u0, u1, u2;
        eig3_top(gsum[0], gsum[1], gsum[2], gsum[3], gsum[4], gsum[5], lam, u0, u1, u2);
        u_sh[0] = u0; u_sh[1] = u1; u_sh[2] = u2; u_sh[3] = lam;
    }
    __syncthreads();
    const double u0 = u_sh[0], u1 = u_sh[1], u2 = u_sh[2];
    const double sigma = sqrt(fmax(u_sh[3], 0.0));
    double* wk = W + k * (long long)Fp;
    double sign = 1.0, scale = 1.0;
    if (local_mode) {
        // project_weight(+w) vs project_weight(-w), keep the larger norm (:90-94)
        double mp = 0.0, mn = 0.0, sq[2] = {0.0, 0.0};
        for (int f = tid; f < F; f += nt) {
            const double wv = u0 * slab[f] + u1 * slab[Fp + f] + u2 * slab[2 * Fp + f];
            const double p = fmax(wv, 0.0), q = fmax(-wv, 0.0);
            mp = fmax(mp, p); mn = fmax(mn, q);
            sq[0] += p * p; sq[1] += q * q;
        }
        block_sum<2>(sq, sh_d);
        mp = wave_max(mp); mn = wave_max(mn);
        __syncthreads();
        if ((tid & 63) == 0) { sh_d[tid >> 6] = mp; sh_d[8 + (tid >> 6)] = mn; }
        __syncthreads();
        mp = 0.0; mn = 0.0;
        for (int q = 0; q < (nt >> 6); ++q) { mp = fmax(mp, sh_d[q]); mn = fmax(mn, sh_d[8 + q]); }
        __syncthreads();
        const double npos = (mp == 0.0) ? sqrt(sq[0]) : sqrt(sq[0]) / mp;
        const double nneg = (mn == 0.0) ? sqrt(sq[1]) : sqrt(sq[1]) / mn;
        if (npos > nneg) { sign = 1.0; scale = (mp == 0.0) ? 1.0 : mp; }
        else { sign = -1.0; scale = (mn == 0.0) ? 1.0 : mn; }
    }
    double wn[1] = {0.0};
    for (int f = tid; f < Fp; f += nt) {
        double wv = 0.0;
        if (f < F) {
            wv = u0 * slab[f] + u1 * slab[Fp + f] + u2 * slab[2 * Fp + f];
            if (local_mode) wv = fmax(sign * wv, 0.0) / scale;
        }
        wk[f] = wv;
        wn[0] += wv * wv;
    }
    block_sum<1>(wn, sh_d);
    if (tid == 0) {
        scal[k * 4 + 0] = sigma;
        scal[k * 4 + 1] = wn[0];
        scal[k * 4 + 2] = __longlong_as_double(gidx);
        if (panel != nullptr) panel->committed = k - k_panel0 + 1;
    }
}


// --------------------------------------------------------------------------------------
// host-side launch helpers shared by the residual and the panel paths
// --------------------------------------------------------------------------------------

// threads per vertex T and double2-per-thread E2 such that T * E2 * 2 >= Fp
static inline bool pick_cfg(int64_t Fp, StreamCfg& c) {
    const int64_t F2 = Fp / 2;
    const int Ts[5] = {64, 128, 256, 512, 1024};
    for (int e2 = 4; e2 <= 16; e2 *= 2)
        for (int i = 0; i < 5; ++i)
            if ((int64_t)Ts[i] * e2 >= F2) {
                c.T = Ts[i];
                c.E2 = e2;
                c.block = Ts[i] >= 256 ? Ts[i] : 256;
                c.vpb = c.block / Ts[i];
                return true;
            }
    return false;
}

static inline int stream_grid(const asb_ctx* ctx, const StreamCfg& c, int64_t n) {
    int64_t want = (n + c.vpb - 1) / c.vpb;
    int grid = (int)(want < ctx->nblk_cap ? want : ctx->nblk_cap);
    return grid < 1 ? 1 : grid;
}

struct StreamArgs {
    double* R;
    const double* wk;
    double* scal_k;
    const double* s;
    double* ck;
    double* energy;
    double* pmax;
    long long* pidx;
    double* psum;
    long long n;
    const PanelState* panel;
};

template <int T, int E2>
static void launch_stream_te(asb_ctx* ctx, bool update, int grid, const StreamArgs& a) {
    constexpr int BLOCK = (T >= 256 ? T : 256);
    const int F2 = (int)(ctx->Fp / 2);
    if (update)
        hipLaunchKernelGGL((k_stream<T, E2, true>), dim3(grid), dim3(BLOCK), 0, ctx->stream, a.R, a.wk, a.scal_k, a.s,
                           a.ck, a.energy, a.pmax, a.pidx, a.psum, a.n, F2, a.panel);
    else
        hipLaunchKernelGGL((k_stream<T, E2, false>), dim3(grid), dim3(BLOCK), 0, ctx->stream, a.R, a.wk, a.scal_k,
                           a.s, a.ck, a.energy, a.pmax, a.pidx, a.psum, a.n, F2, a.panel);
}

template <int E2>
static void launch_stream_e(asb_ctx* ctx, int T, bool update, int grid, const StreamArgs& a) {
    switch (T) {
        case 64: launch_stream_te<64, E2>(ctx, update, grid, a); break;
        case 128: launch_stream_te<128, E2>(ctx, update, grid, a); break;
        case 256: launch_stream_te<256, E2>(ctx, update, grid, a); break;
        case 512: launch_stream_te<512, E2>(ctx, update, grid, a); break;
        default: launch_stream_te<1024, E2>(ctx, update, grid, a); break;
    }
}

static inline void launch_stream(asb_ctx* ctx, const StreamCfg& c, bool update, int grid, const StreamArgs& a) {
    switch (c.E2) {
        case 4: launch_stream_e<4>(ctx, c.T, update, grid, a); break;
        case 8: launch_stream_e<8>(ctx, c.T, update, grid, a); break;
        default: launch_stream_e<16>(ctx, c.T, update, grid, a); break;
    }
}

static inline int prof_begin(asb_ctx* ctx, size_t& slot) {
    slot = (size_t)-1;
    if (!ctx->prof) return ASB_OK;
    if (ctx->ev_used == ctx->ev_pool.size()) {
        hipEvent_t a, b;
        ASB_HIP(ctx, hipEventCreate(&a));
        ASB_HIP(ctx, hipEventCreate(&b));
        ctx->ev_pool.emplace_back(a, b);
    }
    slot = ctx->ev_used++;
    ASB_HIP(ctx, hipEventRecord(ctx->ev_pool[slot].first, ctx->stream));
    return ASB_OK;
}
static inline int prof_end(asb_ctx* ctx, size_t slot) {
    if (slot == (size_t)-1) return ASB_OK;
    ASB_HIP(ctx, hipEventRecord(ctx->ev_pool[slot].second, ctx->stream));
    return ASB_OK;
}

// block energies: sum of the current residual energies of p consecutive rows; first maximum of this shard
// (:86-92, indxLargestDeformation).  The shard must hold whole blocks.
static __global__ __launch_bounds__(256) void k_block_argmax(const double* __restrict__ energy, long long nblocks, int p, long long b0,
                                                      double* __restrict__ pmax, long long* __restrict__ pidx) {
    __shared__ double sh_d[256];
    __shared__ long long sh_i[256];
    double be = -1.0;
    long long bi = 0x7fffffffffffffffLL;
    for (long long b = (long long)blockIdx.x * 256 + threadIdx.x; b < nblocks; b += (long long)gridDim.x * 256) {
        double s = 0.0;
        for (int i = 0; i < p; ++i) s += energy[b * p + i];
        if (am_better(s, b0 + b, be, bi)) { be = s; bi = b0 + b; }
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
    if (threadIdx.x == 0) { pmax[blockIdx.x] = sh_d[0]; pidx[blockIdx.x] = sh_i[0]; }
}

