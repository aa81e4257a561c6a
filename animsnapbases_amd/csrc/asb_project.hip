// Residual-free "panel" form of the greedy deflation for support='global' --
// posComponents.extract_k_components, snapbases/posComponents.py:67-122.  gfx950 only.
//
// With global support every deflation is an orthogonal projection, the weights w_k are
// mutually orthogonal and
//       c_k = X^T w_k / |w_k|^2,     R_k = X - sum_{j<k} w_j (x) c_j,
//       energy_k[v] = |X_v|^2 - sum_{j<k} |w_j|^2 |c_j[v]|^2          (non-increasing in k)
// so X is never modified.  The K strictly sequential arg-max / SVD steps only ever need
// the residual rows of the few vertices that can still win.  Per panel:
//   1. threshold tau such that ~M vertices have energy > tau          (k_hist / k_tau)
//   2. those candidates' residual rows are rebuilt EXACTLY in a compact buffer
//      (X_v - sum_j c_j[v] w_j), everyone else is bounded by tau      (k_compact / k_gather)
//   3. up to 16 greedy steps run on the compact buffer alone (the same k_pick / k_stream
//      kernels as the residual path); a step is committed only while the best candidate's
//      exact energy beats tau + margin, i.e. while it is provably the global arg-max
//   4. ONE pass over X projects every vertex on the panel's new weights with f64 MFMA
//      (Y = X . W_panel, 16 columns): c_k for all vertices, energies -= |w|^2 |c|^2
//                                                                     (k_project_mfma)
// HBM traffic: one read of X per PANEL instead of a read + write of R per COMPONENT.
#include "asb_kernels.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));

#define ASB_NBINS 2048

// layout of ctx->scalar_dev (doubles)
enum { SC_NORMX2 = 0, SC_E0MAX = 1, SC_EMAX = 2, SC_LO = 3, SC_HI = 4, SC_ABOVE = 5, SC_TAU = 6, SC_OVERFLOW = 7,
       SC_TAU_HI = 11, SC_TAU2 = 12, SC_BANDMAX = 13, SC_TAUG = 16, SC_GG = 24, SC_HH = 32,
       SC_TAU_DIV = 40, SC_DIV_SEED = 41 };  // 8..10: k_best_energy; 11..13: super-panels (band); 40 / 41: the diversity family
                                                                     // 16..16+ASB_NG-1: thresholds of a guessed selection,
                                                                     // 24.. / 32..: their g and h (k_tau_multi writes all three)
// Guessed candidates of a first panel (see asb_project_run): the scores EV + g (E - EV), g on a geometric grid -- E - EV
// is the energy along the constant-in-time direction, of which the first components leave a falling share g behind
#define ASB_NG 6
// (the g of each score sits beside its threshold: a selection by the sketch predictor's scores -- asb_sketch.hip -- is the
// same predicate with ev = score, g = 0 and one live threshold, the others at +inf)
// score q of a vertex with energy e, of which m = e - ev along the constant direction: ev + g m + h sqrt(m ev).  The last term
// (round 3) is an UPPER CONFIDENCE bound: the first components' coefficients c_v = alpha (-a_v sqrt(F)) + beta (n_v . n_w) carry
// the cross term 2 alpha beta a_v sqrt(F) (n_v . n_w), zero-mean with variance 4 alpha^2 beta^2 M_v EV_v / (3 F) per step -- about
// 1 % of a vertex's energy in all, as large as the spread of EV among the leading few hundred vertices -- so a vertex with much
// energy along the constant direction can rise by that much; h = kappa / sqrt(F) (guess_thresholds)
__device__ __forceinline__ double guess_score(double ev, double m, double s, double g, double h) { return ev + g * m + h * s; }
__device__ __forceinline__ bool in_guess(double e, double ev, const double* __restrict__ sc) {
    const double m = e - ev, s = sqrt(fmax(m, 0.0) * fmax(ev, 0.0));
    bool in = false;
#pragma unroll
    for (int q = 0; q < ASB_NG; ++q) in = in || (guess_score(ev, m, s, sc[SC_GG + q], sc[SC_HH + q]) > sc[SC_TAUG + q]);
    return in;
}
// The DIVERSITY family of a candidate selection (round 4).  The panel kernel's greedy steps on the candidate rows do two jobs:
// they find the winners, and the weight vectors of the steps the pass rejects become the sketch the next read's candidates are
// predicted from (asb_sketch.hip).  For the second job the largest energies are the wrong rows on localised data: they all lie
// in the one or two strongest modes, the steps behind those modes run on noise, and the sketch misses every other mode.  An
// energy-WEIGHTED RANDOM SAMPLE of all vertices spans the dominant frame subspace of the whole residual (row sampling with
// probability ~ a power of the squared row norm): vertex i belongs to it when e_i^(1/4) / Exp_i > tau_div, Exp_i = -log(u_i) a unit
// exponential from a hash of (i, seed) -- the m largest keys are an exact weighted sample without replacement (Efraimidis-
// Spirakis).  The weight is e^(1/4), not e: with weights ~ e the sample crowds into the strongest modes again and the WEAK modes,
// whose directions the replay needs once the strong ones are gone, get one or two rows -- CPU replay (tools/sim_sketch2.py, 50
// bumps, 190 sampled rows beside the first read's candidates): the read predicted from that sketch keeps 18 / 24 components (two
// data sets) with weights e, 19 / 35 with e^(1/2), 37 with e^(1/4).  Computed on
// the fly from the energy at the start of the read and the vertex's index, by the compaction and by the pass's check alike;
// off while sc[SC_TAU_DIV] is "infinite".  Like every other family it only names candidates: nothing rests on it.
// WHEN: the family pays where the energy is unevenly spread over the vertices -- localised modes: the largest energies all lie in
// one or two of them -- and costs a read or two where it is not (global modes: rank 50 + noise 13.1 -> 14.9 ms, a slowly decaying
// spectrum 17.8 -> 21.0 ms with it, 50 bumps 34.2 -> 16.9 ms).  The squared coefficient of variation of the per-vertex energies
// outside the constant direction, a by-product of the standardisation sweep, separates the two by orders of magnitude (0.02 - 0.07
// for global modes, 0 for noise, 3.6 for the bumps): the family is on above ASB_DIVERSE_CV2 (0.5).
static inline bool diverse_on(const asb_ctx* ctx) {
    static const double thr = getenv("ASB_DIVERSE_CV2") ? atof(getenv("ASB_DIVERSE_CV2")) : 0.5;
    return ctx->diverse && (!ctx->ev_valid || ctx->ev_cv2 > thr);
}
#define ASB_DIV_Q (ASB_NG + 1)              // its slot among the thresholds of a multi-score selection
#define ASB_NQ (ASB_NG + 2)
__device__ __forceinline__ double div_key(double e, long long i, double seed_bits) {
    unsigned long long x = ((unsigned long long)i + 1ull) * 0x9E3779B97F4A7C15ull ^ (unsigned long long)__double_as_longlong(seed_bits);
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;      // splitmix64
    const double u = ((double)(x >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    return sqrt(sqrt(fmax(e, 0.0))) / -log(u);
}
__device__ __forceinline__ bool in_div(double e, long long i, const double* __restrict__ sc) {
    const double t = sc[SC_TAU_DIV];
    return t < 1.0e299 && div_key(e, i, sc[SC_DIV_SEED]) > t;
}
__global__ void k_sc_set(double* __restrict__ sc, int slot, double v) { sc[slot] = v; }

// --------------------------------------------------------------------------------------
// k_gather: item s -> vertex v = idx_map[s] - v0 (or s); rebuilds its residual row
//   R_v = X_v - sum_{j<k0} w_j (x) c_j[v]   exactly, optionally stores it (dst) and
// always returns its energy (+ per-block max / first index / sum).
// --------------------------------------------------------------------------------------
// CPT items per thread group: the weights w_j (read from L2 by every block: K x 8 F bytes per item, the kernel's bound)
// are loaded once for CPT items.
template <int T, int E2, int CPT = 1>
__global__ __launch_bounds__((T >= 256 ? T : 256)) void k_gather(
    const double* __restrict__ X, const long long* __restrict__ idx_map, long long v0, long long n_items,
    const PanelState* __restrict__ panel, const double* __restrict__ comps, long long comp_stride,
    const double* __restrict__ W, int k0, int F2, double* __restrict__ dst, double* __restrict__ energy_out,
    double* __restrict__ pmax, long long* __restrict__ pidx, double* __restrict__ psum) {
    constexpr int BLOCK = (T >= 256 ? T : 256);
    constexpr int VPB = BLOCK / T;
    constexpr int NW = T / 64;
    const int tid = threadIdx.x, g = tid / T, t = tid % T, wig = t >> 6, lane = tid & 63;
    __shared__ double red[VPB][NW];
    __shared__ double lead_e[VPB];
    __shared__ long long lead_i[VPB];
    __shared__ double lead_s[VPB];
    if (panel != nullptr && panel->n_cand < n_items) n_items = panel->n_cand;
    double bmax = -1.0, bsum = 0.0;
    long long bidx = 0x7fffffffffffffffLL;
    for (long long base = (long long)blockIdx.x * VPB * CPT; base < n_items; base += (long long)gridDim.x * VPB * CPT) {
        long long s[CPT], v[CPT];
        bool valid[CPT];
        double2 x[CPT][3][E2];
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc) {
            s[cc] = base + (long long)g * CPT + cc;
            valid[cc] = s[cc] < n_items;
            v[cc] = valid[cc] ? (idx_map ? idx_map[s[cc]] - v0 : s[cc]) : 0;
            const double2* row = reinterpret_cast<const double2*>(X) + v[cc] * 3 * (long long)F2;
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int i = 0; i < E2; ++i) {
                    const int j = t + i * T;
                    x[cc][d][i] = (valid[cc] && j < F2) ? row[(long long)d * F2 + j] : make_double2(0.0, 0.0);
                }
        }
        constexpr int U = E2 * CPT <= 4 ? 4 : (E2 * CPT <= 8 ? 2 : 1);      // components in flight (bounded by the register file)
        for (int q0 = 0; q0 < k0; q0 += U) {      // U components' loads in flight (same subtraction order as one by one)
            double cf[U][CPT][3];
            double2 w[U][E2];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + u < k0 ? q0 + u : k0 - 1;
                const bool on = q0 + u < k0;
#pragma unroll
                for (int cc = 0; cc < CPT; ++cc) {
                    const double* cq = comps + (long long)q * comp_stride + v[cc] * 3;
                    cf[u][cc][0] = on ? cq[0] : 0.0; cf[u][cc][1] = on ? cq[1] : 0.0; cf[u][cc][2] = on ? cq[2] : 0.0;
                }
                const double2* wq = reinterpret_cast<const double2*>(W) + (long long)q * F2;
#pragma unroll
                for (int i = 0; i < E2; ++i) {
                    const int j = t + i * T;
                    w[u][i] = (on && j < F2) ? wq[j] : make_double2(0.0, 0.0);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (q0 + u < k0) {
#pragma unroll
                    for (int cc = 0; cc < CPT; ++cc)
#pragma unroll
                        for (int i = 0; i < E2; ++i) {
                            x[cc][0][i].x -= w[u][i].x * cf[u][cc][0]; x[cc][0][i].y -= w[u][i].y * cf[u][cc][0];
                            x[cc][1][i].x -= w[u][i].x * cf[u][cc][1]; x[cc][1][i].y -= w[u][i].y * cf[u][cc][1];
                            x[cc][2][i].x -= w[u][i].x * cf[u][cc][2]; x[cc][2][i].y -= w[u][i].y * cf[u][cc][2];
                        }
                }
        }
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc) {
            double e = 0.0;
            double2* out = dst ? reinterpret_cast<double2*>(dst) + s[cc] * 3 * (long long)F2 : nullptr;
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int i = 0; i < E2; ++i) {
                    const int j = t + i * T;
                    if (out && valid[cc] && j < F2) out[(long long)d * F2 + j] = x[cc][d][i];
                    e += x[cc][d][i].x * x[cc][d][i].x + x[cc][d][i].y * x[cc][d][i].y;
                }
            e = wave_sum(e);
            if (NW > 1) {
                __syncthreads();
                if (lane == 0) red[g][wig] = e;
                __syncthreads();
                double sum = 0.0;
#pragma unroll
                for (int q = 0; q < NW; ++q) sum += red[g][q];
                e = sum;
            }
            if (t == 0 && valid[cc]) {
                energy_out[s[cc]] = e;
                bsum += e;
                if (am_better(e, s[cc], bmax, bidx)) { bmax = e; bidx = s[cc]; }
            }
        }
    }
    if (t == 0) { lead_e[g] = bmax; lead_i[g] = bidx; lead_s[g] = bsum; }
    __syncthreads();
    if (tid == 0) {
        double be = lead_e[0], bs = lead_s[0];
        long long bi = lead_i[0];
#pragma unroll
        for (int q = 1; q < VPB; ++q) {
            bs += lead_s[q];
            if (am_better(lead_e[q], lead_i[q], be, bi)) { be = lead_e[q]; bi = lead_i[q]; }
        }
        pmax[blockIdx.x] = be; pidx[blockIdx.x] = bi; psum[blockIdx.x] = bs;
    }
}

// --------------------------------------------------------------------------------------
// threshold selection: two-level histogram of the energies, all on the device
// --------------------------------------------------------------------------------------
// reduces the per-block maxima of the last pass: sc[SC_EMAX]; level-1 range [0, Emax];
// first = 1 also records the initial maximum and the total (|X|^2).
__global__ __launch_bounds__(256) void k_range_init(const double* pmax, const long long* pidx, const double* psum,
                                                    int nblk, double* __restrict__ sc, int first) {
    __shared__ double sh_d[512];
    __shared__ long long sh_i[256];
    double be, bs;
    long long bi;
    reduce_partials(pmax, pidx, psum, nblk, sh_d, sh_i, be, bi, bs);
    if (threadIdx.x == 0) {
        sc[SC_EMAX] = be;
        sc[SC_LO] = 0.0;
        sc[SC_HI] = be;
        sc[SC_ABOVE] = 0.0;
        if (first) { sc[SC_NORMX2] = bs; sc[SC_E0MAX] = be; }
    }
}

// the same from the energies the standardisation sweep left behind (e0 = [|X|^2, largest energy])
__global__ void k_range_restore(double* __restrict__ sc, const double* __restrict__ e0) {
    sc[SC_EMAX] = e0[1];
    sc[SC_LO] = 0.0;
    sc[SC_HI] = e0[1];
    sc[SC_ABOVE] = 0.0;
    sc[SC_NORMX2] = e0[0];
    sc[SC_E0MAX] = e0[1];
}

// level 1 (by_exponent): hist[b] = #{ e : biased exponent of e == b } -- exact, needs no range, and
// covers the whole dynamic range of the energies (they fall by 1e8 on low-rank data);
// level 2: linear bins inside the crossing binade [lo, hi).
__global__ __launch_bounds__(256) void k_hist(const double* __restrict__ E, long long n, const double* __restrict__ sc,
                                              int* __restrict__ hist, int by_exponent) {
    __shared__ int lh[ASB_NBINS];
    for (int i = threadIdx.x; i < ASB_NBINS; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const double lo = sc[SC_LO], hi = sc[SC_HI];
    const double scale = (hi > lo) ? (double)ASB_NBINS / (hi - lo) : 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double e = E[i];
        int b;
        if (by_exponent) {
            if (!(e >= 0.0)) continue;
            b = (int)((__double_as_longlong(e) >> 52) & 0x7FF);
        } else {
            if (e < lo || e >= hi) continue;
            b = (int)((e - lo) * scale);
            if (b > ASB_NBINS - 1) b = ASB_NBINS - 1;
        }
        atomicAdd(&lh[b], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ASB_NBINS; i += blockDim.x)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// level 1: find the bin where the count from the top reaches m_target -> next range.
// level 2: final tau (lower edge of the crossing bin; upper edge if that overflows m_cap).
__device__ __forceinline__ void tau_body(int* __restrict__ hist, double* __restrict__ sc, int level, long long m_target,
                                         long long m_cap, int* lh, int* seg) {
    constexpr int PER = ASB_NBINS / 256;
    int ssum = 0;
    for (int q = 0; q < PER; ++q) {
        const int v = hist[threadIdx.x * PER + q];
        hist[threadIdx.x * PER + q] = 0;          // consumed: the next k_hist accumulates into zeros again
        lh[threadIdx.x * PER + q] = v;
        ssum += v;
    }
    seg[threadIdx.x] = ssum;
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double lo = sc[SC_LO], hi = sc[SC_HI];
    const double width = (hi - lo) / (double)ASB_NBINS;
    long long acc = (level == 1) ? 0 : (long long)sc[SC_ABOVE];      // vertices above the binade found at level 1
    int sg = 255;
    for (; sg >= 0; --sg) {                 // coarse: segments of PER bins from the top
        if (acc + seg[sg] >= m_target) break;
        acc += seg[sg];
    }
    int b = -1;
    if (sg >= 0)
        for (b = sg * PER + PER - 1; b >= sg * PER; --b) {
            if (acc + lh[b] >= m_target) break;
            acc += lh[b];
        }
    if (b < 0) {                       // fewer than m_target vertices in range: take them all
        if (level == 1) { sc[SC_LO] = 0.0; sc[SC_HI] = 0.0; sc[SC_ABOVE] = (double)acc; sc[SC_TAU] = -1.0; }
        else sc[SC_TAU] = nextafter(lo, -1.0e300);
        return;
    }
    double edge_lo, edge_hi;
    if (level == 1) {                  // bin b = biased exponent: the binade [2^(b-1023), 2^(b-1022))
        edge_lo = (b == 0) ? 0.0 : ldexp(1.0, b - 1023);
        edge_hi = (b >= 2046) ? 1.7976931348623157e308 : ldexp(1.0, b - 1022);
    } else {
        edge_lo = lo + b * width;
        edge_hi = (b == ASB_NBINS - 1) ? hi : lo + (b + 1) * width;
    }
    if (level == 1) {
        sc[SC_LO] = edge_lo;
        sc[SC_HI] = edge_hi;
        sc[SC_ABOVE] = (double)acc;
        sc[SC_TAU] = edge_lo;          // provisional
    } else {
        // candidates are E > tau.  The lower edge keeps the crossing bin (>= m_target candidates),
        // the upper edge drops it when it would overflow the buffer.
        double tau = edge_lo;
        if (acc + lh[b] > m_cap && acc > 0) tau = edge_hi;
        sc[SC_TAU] = nextafter(tau, -1.0e300);      // strict '>' in the compaction keeps e == edge
    }
}

__global__ __launch_bounds__(256) void k_tau(int* __restrict__ hist, double* __restrict__ sc, int level,
                                             long long m_target, long long m_cap) {
    __shared__ int lh[ASB_NBINS];
    __shared__ int seg[256];
    tau_body(hist, sc, level, m_target, m_cap, lh, seg);
}

// ---- the thresholds of a guessed selection in one go: ASB_NG + 1 scores EV + g (E - EV) (the last one, g = 1, is the
// energy itself), each with its own histogram, range block scm[q * 8 ..] (slots as in sc) and target; block q of k_tau_multi
// is k_tau for score q, and at level 2 it leaves the threshold where the compaction looks for it
struct GuessTargets { double g[ASB_NQ], h[ASB_NQ]; long long m_target[ASB_NQ], m_cap[ASB_NQ]; double div_seed; };
__global__ __launch_bounds__(256) void k_hist_multi(const double* __restrict__ E, const double* __restrict__ EV, long long n,
                                                    const double* __restrict__ scm, int* __restrict__ hist, int by_exponent,
                                                    GuessTargets gt, int nq, unsigned qmask = 0xffffffffu) {
    // one score at a time through the same LDS histogram: E / EV are re-read from cache, LDS holds 8 KB
    __shared__ int lh[ASB_NBINS];
    for (int q = 0; q < nq; ++q) {
        if (!((qmask >> q) & 1u)) continue;
        for (int i = threadIdx.x; i < ASB_NBINS; i += blockDim.x) lh[i] = 0;
        __syncthreads();
        const double lo = scm[q * 8 + SC_LO], hi = scm[q * 8 + SC_HI], g = gt.g[q], h = gt.h[q];
        const double scale = (hi > lo) ? (double)ASB_NBINS / (hi - lo) : 0.0;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
            double e;
            if (q == ASB_DIV_Q) e = div_key(E[i], i, gt.div_seed);
            else {
                const double ev = EV ? EV[i] : E[i], m = E[i] - ev;
                e = guess_score(ev, m, h != 0.0 ? sqrt(fmax(m, 0.0) * fmax(ev, 0.0)) : 0.0, g, h);
            }
            int b;
            if (by_exponent) {
                if (!(e >= 0.0)) continue;
                b = (int)((__double_as_longlong(e) >> 52) & 0x7FF);
            } else {
                if (e < lo || e >= hi) continue;
                b = (int)((e - lo) * scale);
                if (b > ASB_NBINS - 1) b = ASB_NBINS - 1;
            }
            atomicAdd(&lh[b], 1);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < ASB_NBINS; i += blockDim.x)
            if (lh[i]) atomicAdd(&hist[q * ASB_NBINS + i], lh[i]);
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_tau_multi(int* __restrict__ hist, double* __restrict__ scm, double* __restrict__ sc, int level,
                                                   GuessTargets gt, unsigned qmask = 0xffffffffu) {
    __shared__ int lh[ASB_NBINS];
    __shared__ int seg[256];
    const int q = blockIdx.x;
    if (!((qmask >> q) & 1u)) {              // a score that takes no part: nothing lies above its threshold
        if (level == 2 && threadIdx.x == 0 && q < ASB_NG) { sc[SC_TAUG + q] = 1.0e300; sc[SC_GG + q] = 0.0; sc[SC_HH + q] = 0.0; }
        if (level == 2 && threadIdx.x == 0 && q == ASB_DIV_Q) sc[SC_TAU_DIV] = 1.0e300;
        return;
    }
    tau_body(hist + q * ASB_NBINS, scm + q * 8, level, gt.m_target[q], gt.m_cap[q], lh, seg);
    if (level == 2 && threadIdx.x == 0) {
        if (q == ASB_DIV_Q) {
            const double t = scm[q * 8 + SC_TAU];
            sc[SC_TAU_DIV] = t < 0.0 ? 0.0 : t;           // (fewer vertices than the target: all with a positive key)
            sc[SC_DIV_SEED] = gt.div_seed;
        } else {
            sc[q < ASB_NG ? SC_TAUG + q : SC_TAU] = scm[q * 8 + SC_TAU];
            if (q < ASB_NG) { sc[SC_GG + q] = gt.g[q]; sc[SC_HH + q] = gt.h[q]; }
        }
    }
}

// ordered compaction, two stages: cand_idx = global ids of { v : E[v] > tau } in increasing
// order (so that slot order == vertex order and arg-max ties resolve to the lowest vertex), at
// most m_cap of them; stage B also initialises the panel state.
#define ASB_CBLOCKS 128
__global__ __launch_bounds__(256) void k_compact_a(const double* __restrict__ E, long long n, long long v0,
                                                   const double* __restrict__ sc, int take_all, long long m_cap,
                                                   long long* __restrict__ tmp, long long* __restrict__ cnt, int hi_slot = -1,
                                                   const double* __restrict__ E2 = nullptr) {
    __shared__ long long pre[256];
    const int tid = threadIdx.x;
    const double tau = take_all ? -1.0e300 : sc[SC_TAU];
    const double tau_hi = hi_slot >= 0 ? sc[hi_slot] : 1.7976931348623157e308;       // band: tau < E <= tau_hi
    const long long seg = (n + gridDim.x - 1) / gridDim.x;
    const long long s0 = blockIdx.x * seg, s1 = (s0 + seg < n) ? s0 + seg : n;
    const long long sub = (seg + 255) / 256;
    const long long a = s0 + tid * sub, b = (a + sub < s1) ? a + sub : s1;
    long long c = 0;
    const bool div = !take_all && hi_slot < 0;
    for (long long i = a; i < b; ++i)
        c += ((E[i] > tau && !(E[i] > tau_hi)) || (E2 && in_guess(E[i], E2[i], sc)) || (div && in_div(E[i], i, sc)));
    pre[tid] = c;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const long long add = (tid >= o) ? pre[tid - o] : 0;
        __syncthreads();
        pre[tid] += add;
        __syncthreads();
    }
    long long pos = pre[tid] - c;
    long long* out = tmp + (long long)blockIdx.x * m_cap;
    for (long long i = a; i < b; ++i)
        if ((E[i] > tau && !(E[i] > tau_hi)) || (E2 && in_guess(E[i], E2[i], sc)) || (div && in_div(E[i], i, sc))) {
            if (pos < m_cap) out[pos] = v0 + i;
            ++pos;
        }
    if (tid == 255) cnt[blockIdx.x] = pre[255];
}

__global__ __launch_bounds__(64) void k_compact_b(const long long* __restrict__ tmp, const long long* __restrict__ cnt,
                                                  int nb, long long m_cap, long long* __restrict__ cand_idx,
                                                  PanelState* __restrict__ panel) {
    // block b copies its own piece to its offset = sum of the counts of the blocks before it
    const int b = blockIdx.x;
    long long off = 0;
    for (int q = threadIdx.x; q < b; q += 64) off += cnt[q];
    off = (long long)wave_sum((double)off);            // counts are small integers: exact in f64
    const long long c = cnt[b];
    for (long long q = threadIdx.x; q < c && q < m_cap; q += 64)
        if (off + q < m_cap) cand_idx[off + q] = tmp[(long long)b * m_cap + q];
    if (b == nb - 1 && threadIdx.x == 0) {
        const long long total = off + c;
        panel->n_cand = total > m_cap ? m_cap : total;
        panel->pad = total > m_cap ? 1 : 0;          // overflow: some candidates were dropped
    }
}

// arms the panel: every vertex outside the candidate buffer has energy <= theta
__global__ __launch_bounds__(256) void k_panel_arm(PanelState* __restrict__ panel, const double* __restrict__ sc, int global_all,
                            long long n_slots, double margin_rel, unsigned* __restrict__ coop_flags,
                            unsigned long long* __restrict__ coop_words, int n_words, int theta_from_band = 0,
                            long long spec_max = 0) {
    // the co-resident panel kernel's flags; its exchange words (records + weight buffers) start out as "not written"
    // (k_panel_multi: all bits set -- a NaN no energy or weight can be, -1 for slots)
    if (coop_flags && threadIdx.x < 4) coop_flags[threadIdx.x] = 0u;
    if (coop_words)
        for (int r = threadIdx.x; r < n_words; r += blockDim.x) coop_words[r] = 0xFFFFFFFFFFFFFFFFull;
    if (threadIdx.x != 0) return;
    if (n_slots >= 0) panel->n_cand = n_slots;
    panel->pad &= 1;
    // later sub-panels of a super-panel: every vertex outside candidates and band is below tau2 (stale but valid), the
    // band's energies are exact (k_correct on the band after every sub-panel)
    panel->theta = global_all ? -1.0e300
                              : (panel->pad ? 1.0e300 : (theta_from_band ? fmax(sc[SC_TAU2], sc[SC_BANDMAX]) : sc[SC_TAU]));
    panel->margin = margin_rel * sc[SC_E0MAX];
    panel->done = 0;
    panel->committed = 0;
    panel->proven = -1;
    panel->spec_max = spec_max;
    panel->spec_ok = ASB_PANEL_COLS;
}

__global__ void k_sc_copy(double* __restrict__ sc, int dst, int src) { sc[dst] = sc[src]; }
// forced single candidate (degenerate ties): global vertex id gidx if this shard owns it
__global__ void k_force_single(long long gidx, long long v0, long long n_loc, long long* __restrict__ cand_idx,
                               PanelState* __restrict__ panel) {
    const bool mine = gidx >= v0 && gidx < v0 + n_loc;
    if (mine) cand_idx[0] = gidx;
    panel->n_cand = mine ? 1 : 0;
    panel->pad = 0;
}

// first arg-max of the energies over the last pass's block partials -> out[0] = energy, out[1] = local idx bits
__global__ __launch_bounds__(256) void k_best_energy(const double* pmax, const long long* pidx, const double* psum, int nblk,
                                                     double* __restrict__ out) {
    __shared__ double sh_d[512];
    __shared__ long long sh_i[256];
    double be, bs;
    long long bi;
    reduce_partials(pmax, pidx, psum, nblk, sh_d, sh_i, be, bi, bs);
    if (threadIdx.x == 0) { out[0] = be; out[1] = __longlong_as_double(bi); out[2] = bs; }
}

// --------------------------------------------------------------------------------------
// Orthogonality correction of a panel.  In exact arithmetic the new weights are orthogonal
// to every earlier w_j; numerically the candidate rows they are built from carry the
// cancellation error eps*|X|/|R| along those directions, and since the projection is taken
// against X (not the residual), Y = X.W_panel picks up  sum_j c_j[v] (w_j . w_t).  Left in, it
// costs eps*kappa^2 (the classical-Gram-Schmidt effect); removing it restores eps*kappa:
//     c_t[v] = ( X_v . w_t - sum_{j<k0} c_j[v] |w_j|^2-free Gram term ) / |w_t|^2
// k_panel_gram: G[j][t] = w_j . w_{k0+t}  (j < k0 + ncols).   k_correct: applies it -- earlier panels first, then the
// panel's own columns in order -- and updates the
// energies E[v] -= sum_t |w_t|^2 |c_t[v]|^2 and the per-block partial records.
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_panel_gram(const double* __restrict__ W, const double* __restrict__ Wt,
                                                    int Fp, double* __restrict__ G, double* __restrict__ Gs = nullptr,
                                                    const double* __restrict__ wn2 = nullptr, const double* __restrict__ scal = nullptr) {
    __shared__ double sh[4 * 16];
    const double* wj = W + (long long)blockIdx.x * Fp;
    double acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = 0.0;
    for (int f = threadIdx.x; f < Fp; f += blockDim.x) {
        const double a = wj[f];
        const double* wt = Wt + (long long)f * ASB_PANEL_COLS;
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] += a * wt[t];
    }
    block_sum<16>(acc, sh);
    if (threadIdx.x < 16) {
        // scal != NULL (k_orth_wt's operand): (w_j . w_t) / |w_j|^2 instead
        G[(long long)blockIdx.x * 16 + threadIdx.x] = scal ? acc[threadIdx.x] / scal[(long long)blockIdx.x * 4 + 1] : acc[threadIdx.x];
        if (Gs) Gs[(long long)blockIdx.x * 16 + threadIdx.x] = acc[threadIdx.x] / wn2[threadIdx.x];      // (w_j . w_t) / |w_t|^2
    }
}

// vmap != nullptr: only the vertices vmap[0 .. bstate->n_cand) (global ids; the band of a super-panel) are treated, E is
// their compact energy array and colpart is not written (the full pass accounts for every vertex later).
// SPEC (panels with unproven steps, full shard only): the corrected coefficients are written, but the energies are
// left alone; instead every vertex outside the candidate set (E <= tau at panel start) follows its energy through
// the columns and reports the first unproven step t >= proven whose winner it would have beaten (or tied with,
// within the margin): spec->spec_ok = the smallest such t.  k_commit_energy then applies the columns before it.
template <bool SPEC>
__global__ __launch_bounds__(256) void k_correct(double* __restrict__ comps, long long comp_stride, long long n_vert,
                                                 int k0, int ncols, const double* __restrict__ G,
                                                 const double* __restrict__ wn2, double* __restrict__ E,
                                                 double* __restrict__ pmax, long long* __restrict__ pidx,
                                                 double* __restrict__ psum, double* __restrict__ colpart,
                                                 const long long* __restrict__ vmap = nullptr,
                                                 const PanelState* __restrict__ bstate = nullptr, long long v0 = 0,
                                                 PanelState* __restrict__ spec = nullptr, const double* __restrict__ sc = nullptr,
                                                 const double* __restrict__ E2 = nullptr, const double* __restrict__ Ecl = nullptr) {
    if (vmap != nullptr) n_vert = bstate->n_cand;
    __shared__ int sh_viol[4];
    int viol = ASB_PANEL_COLS;
    double ew[16], sp_tau = 0.0, sp_margin = 0.0;
    int sp_proven = 0;
    if (SPEC) {
#pragma unroll
        for (int t = 0; t < 16; ++t) ew[t] = spec->e_win[t];
        sp_tau = sc[SC_TAU]; sp_margin = spec->margin; sp_proven = (int)spec->proven;
    }
    __shared__ double gs[64 * 16];
    __shared__ double sh_d[512];
    __shared__ long long sh_i[256];
    const int tid = threadIdx.x;
    double bmax = -1.0, bsum = 0.0, csum[16];
    long long bidx = 0x7fffffffffffffffLL;
#pragma unroll
    for (int t = 0; t < 16; ++t) csum[t] = 0.0;
    double inv[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) inv[t] = wn2[t];
    for (long long base = (long long)blockIdx.x * 256; base < n_vert; base += (long long)gridDim.x * 256) {
        const long long vi = base + tid;                      // index into E
        const bool valid = vi < n_vert;
        const long long v = (vmap != nullptr) ? (valid ? vmap[vi] - v0 : 0) : vi;       // local vertex
        double c[16][3];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const double* p = comps + (long long)(k0 + t) * comp_stride + v * 3;
            const bool on = valid && t < ncols;
            c[t][0] = on ? p[0] : 0.0; c[t][1] = on ? p[1] : 0.0; c[t][2] = on ? p[2] : 0.0;
        }
        for (int j0 = 0; j0 < k0; j0 += 64) {
            const int jn = (k0 - j0 < 64) ? k0 - j0 : 64;
            __syncthreads();
            for (int q = tid; q < jn * 16; q += 256) gs[q] = G[(long long)j0 * 16 + q] / inv[q & 15];
            __syncthreads();
            for (int j = 0; j < jn; j += 4) {      // four earlier components in flight: the loop is latency-bound otherwise
                double a[4][3];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool on = valid && j + u < jn;
                    const double* p = comps + (long long)(j0 + j + u) * comp_stride + v * 3;
                    a[u][0] = on ? p[0] : 0.0; a[u][1] = on ? p[1] : 0.0; a[u][2] = on ? p[2] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (j + u < jn) {
#pragma unroll
                        for (int t = 0; t < 16; ++t) {
                            const double gq = gs[(j + u) * 16 + t];
                            c[t][0] -= a[u][0] * gq; c[t][1] -= a[u][1] * gq; c[t][2] -= a[u][2] * gq;
                        }
                    }
            }
        }
        // the same correction among the panel's own columns, in order (column t needs the corrected columns j < t):
        // their mutual orthogonality is only eps relative to the rows at panel start, which is not small against a
        // later, much weaker component of the same panel
        __syncthreads();
        for (int q = tid; q < 16 * 16; q += 256)       // rows beyond ncols do not exist in G (it has K rows in all)
            gs[q] = ((q >> 4) < ncols) ? G[(long long)k0 * 16 + q] / inv[q & 15] : 0.0;
        __syncthreads();
#pragma unroll
        for (int t = 1; t < 16; ++t)
#pragma unroll
            for (int j = 0; j < t; ++j) {
                const double gq = (t < ncols) ? gs[j * 16 + t] : 0.0;
                c[t][0] -= c[j][0] * gq; c[t][1] -= c[j][1] * gq; c[t][2] -= c[j][2] * gq;
            }
        if (SPEC) {
            if (valid) {
                double e = E[vi];
                // the compaction took E > tau (and, for a guessed selection, the scores above their thresholds) -- E as it
                // was when the candidates were chosen (Ecl: a later tile of a double panel sees updated energies in E)
                const double es = Ecl ? Ecl[vi] : e;
                const bool outside = !(es > sp_tau) && !(E2 && in_guess(es, E2[vi], sc)) && !in_div(es, vi, sc);
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    if (t < ncols) {
                        double* p = comps + (long long)(k0 + t) * comp_stride + v * 3;
                        p[0] = c[t][0]; p[1] = c[t][1]; p[2] = c[t][2];
                        if (outside && t >= sp_proven && t < viol && !(ew[t] > e + sp_margin)) viol = t;
                        e -= (c[t][0] * c[t][0] + c[t][1] * c[t][1] + c[t][2] * c[t][2]) * inv[t];
                    }
            }
            continue;
        }
        if (valid) {
            double loss = 0.0;
#pragma unroll
            for (int t = 0; t < 16; ++t)
                if (t < ncols) {
                    double* p = comps + (long long)(k0 + t) * comp_stride + v * 3;
                    p[0] = c[t][0]; p[1] = c[t][1]; p[2] = c[t][2];
                    const double q = (c[t][0] * c[t][0] + c[t][1] * c[t][1] + c[t][2] * c[t][2]) * inv[t];
                    loss += q;
                    csum[t] += q;
                }
            double e = E[vi] - loss;
            if (e < 0.0) e = 0.0;
            E[vi] = e;
            bsum += e;
            if (am_better(e, v, bmax, bidx)) { bmax = e; bidx = v; }
        }
    }
    if (SPEC) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int ov = __shfl_xor(viol, o, 64);
            viol = ov < viol ? ov : viol;
        }
        if ((tid & 63) == 0) sh_viol[tid >> 6] = viol;
        __syncthreads();
        if (tid == 0) {
            for (int q = 1; q < 4; ++q) viol = sh_viol[q] < viol ? sh_viol[q] : viol;
            if (viol < ASB_PANEL_COLS) atomicMin(reinterpret_cast<long long*>(&spec->spec_ok), (long long)viol);
        }
        return;
    }
    __syncthreads();
    block_sum<16>(csum, sh_d);
    if (tid < 16 && vmap == nullptr) colpart[(long long)blockIdx.x * 16 + tid] = csum[tid];
    __syncthreads();
    sh_d[tid] = bmax; sh_d[256 + tid] = bsum; sh_i[tid] = bidx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            sh_d[256 + tid] += sh_d[256 + tid + o];
            if (am_better(sh_d[tid + o], sh_i[tid + o], sh_d[tid], sh_i[tid])) { sh_d[tid] = sh_d[tid + o]; sh_i[tid] = sh_i[tid + o]; }
        }
        __syncthreads();
    }
    if (tid == 0) { pmax[blockIdx.x] = sh_d[0]; pidx[blockIdx.x] = sh_i[0]; psum[blockIdx.x] = sh_d[256]; }
}

// k_correct_rows: the same correction with ONE THREAD PER ROW (r = 3 v + d) instead of one per vertex: 16 accumulators
// per thread instead of 48 (about 80 VGPRs against 432, six waves per SIMD against one), every load of an earlier
// component is 512 contiguous bytes per wave, and the multipliers Gs[j][t] = (w_j . w_t) / |w_t|^2 are wave-uniform
// (scalar loads).  A block is 192 threads = 64 whole vertices; the three rows of a vertex meet again through LDS, where
// the first wave does the per-vertex part (energies, block records, or -- SPEC -- the check of the unproven steps).
template <bool SPEC>
__global__ __launch_bounds__(192) void k_correct_rows(double* __restrict__ comps, long long comp_stride, long long n_vert,
                                                      int k0, int ncols, const double* __restrict__ Gs,
                                                      const double* __restrict__ wn2, double* __restrict__ E,
                                                      double* __restrict__ pmax, long long* __restrict__ pidx,
                                                      double* __restrict__ psum, double* __restrict__ colpart,
                                                      PanelState* __restrict__ spec, const double* __restrict__ sc,
                                                      const double* __restrict__ E2 = nullptr, const double* __restrict__ Ecl = nullptr,
                                                      int pre_orth = 0, double* __restrict__ Etmp = nullptr) {
    // Etmp (SPEC only): besides the check, the energies AS IF every column stood go to Etmp and the block records / column
    // sums are written as in the plain case -- k_tile_decide / k_apply_tmp adopt them if the check finds nothing
    // pre_orth: the pass projected on weights already orthogonalised against everything before them (k_orth_wt): the
    // coefficients are final, only energies / records / the check are left
    __shared__ double qs[16 * 192];
    const int tid = threadIdx.x;
    const long long n_rows = 3 * n_vert;
    int viol = ASB_PANEL_COLS;
    double bmax = -1.0, bsum = 0.0, csum[16];
    long long bidx = 0x7fffffffffffffffLL;
#pragma unroll
    for (int t = 0; t < 16; ++t) csum[t] = 0.0;
    for (long long vb = (long long)blockIdx.x * 64; vb < n_vert; vb += (long long)gridDim.x * 64) {
        const long long r = 3 * vb + tid;
        const bool valid = r < n_rows;
        double c[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) c[t] = (valid && t < ncols) ? comps[(long long)(k0 + t) * comp_stride + r] : 0.0;
        int j = pre_orth ? k0 : 0;
        for (; j + 8 <= k0; j += 8) {          // eight earlier components in flight
            double a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = valid ? comps[(long long)(j + u) * comp_stride + r] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double* g = Gs + (long long)(j + u) * 16;
#pragma unroll
                for (int t = 0; t < 16; ++t) c[t] -= a[u] * g[t];
            }
        }
        for (; j < k0; ++j) {
            const double a = valid ? comps[(long long)j * comp_stride + r] : 0.0;
            const double* g = Gs + (long long)j * 16;
#pragma unroll
            for (int t = 0; t < 16; ++t) c[t] -= a * g[t];
        }
        // among the panel's own columns, in order (column t needs the corrected columns j < t)
#pragma unroll
        for (int t = 1; t < 16; ++t)
            if (t < ncols && !pre_orth) {
                const double* g = Gs + (long long)k0 * 16 + t;
#pragma unroll
                for (int jj = 0; jj < t; ++jj) c[t] -= c[jj] * g[jj * 16];
            }
        __syncthreads();                       // the previous group's reads of qs are done
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (valid && t < ncols && !pre_orth) comps[(long long)(k0 + t) * comp_stride + r] = c[t];
            qs[t * 192 + tid] = c[t] * c[t];
        }
        __syncthreads();
        if (tid < 64) {
            const long long v = vb + tid;
            if (v < n_vert) {
                double e = E[v];
                if (SPEC) {
                    // the compaction took E > tau (and, for a guessed selection, the scores above their thresholds) -- E as
                    // it was when the candidates were chosen (Ecl: a later tile of a double panel sees updated energies)
                    const double es = Ecl ? Ecl[v] : e;
                    const bool outside = !(es > sc[SC_TAU]) && !(E2 && in_guess(es, E2[v], sc)) && !in_div(es, v, sc);
                    const double margin = spec->margin;
                    const int proven = (int)spec->proven;
                    const double e_start = e;
                    double loss = 0.0;
#pragma unroll
                    for (int t = 0; t < 16; ++t)
                        if (t < ncols) {
                            if (outside && t >= proven && t < viol && !(spec->e_win[t] > e + margin)) viol = t;
                            const double q = ((qs[t * 192 + 3 * tid] + qs[t * 192 + 3 * tid + 1]) + qs[t * 192 + 3 * tid + 2]) * wn2[t];
                            e -= q;
                            loss += q;
                            csum[t] += q;
                        }
                    if (Etmp) {                                  // exactly k_commit_energy's arithmetic
                        double en = e_start - loss;
                        if (en < 0.0) en = 0.0;
                        Etmp[v] = en;
                        bsum += en;
                        if (am_better(en, v, bmax, bidx)) { bmax = en; bidx = v; }
                    }
                } else {
                    double loss = 0.0;
#pragma unroll
                    for (int t = 0; t < 16; ++t)
                        if (t < ncols) {
                            const double q = ((qs[t * 192 + 3 * tid] + qs[t * 192 + 3 * tid + 1]) + qs[t * 192 + 3 * tid + 2]) * wn2[t];
                            loss += q;
                            csum[t] += q;
                        }
                    e -= loss;
                    if (e < 0.0) e = 0.0;
                    E[v] = e;
                    bsum += e;
                    if (am_better(e, v, bmax, bidx)) { bmax = e; bidx = v; }
                }
            }
        }
    }
    if (tid >= 64) return;                     // the per-vertex records live in the first wave
    if (SPEC) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int ov = __shfl_xor(viol, o, 64);
            viol = ov < viol ? ov : viol;
        }
        if (tid == 0 && viol < ASB_PANEL_COLS) atomicMin(reinterpret_cast<long long*>(&spec->spec_ok), (long long)viol);
        if (!Etmp) return;
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) csum[t] = wave_sum(csum[t]);
    bsum = wave_sum(bsum);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double om = __shfl_xor(bmax, o, 64);
        const long long oi = __shfl_xor(bidx, o, 64);
        if (am_better(om, oi, bmax, bidx)) { bmax = om; bidx = oi; }
    }
    if (tid == 0) {
#pragma unroll
        for (int t = 0; t < 16; ++t) colpart[(long long)blockIdx.x * 16 + t] = csum[t];
        pmax[blockIdx.x] = bmax; pidx[blockIdx.x] = bidx; psum[blockIdx.x] = bsum;
    }
}

// second half of a panel with unproven steps: E -= sum over the columns that survived the check (t < spec_ok), block
// records for the next selection and the per-column sums -- what k_correct<false> does in one go when nothing is in doubt
__global__ __launch_bounds__(256) void k_commit_energy(const double* __restrict__ comps, long long comp_stride, long long n_vert,
                                                       int k0, const PanelState* __restrict__ spec, const double* __restrict__ wn2,
                                                       double* __restrict__ E, double* __restrict__ pmax,
                                                       long long* __restrict__ pidx, double* __restrict__ psum,
                                                       double* __restrict__ colpart, int force_ncols) {
    __shared__ double sh_d[512];
    __shared__ long long sh_i[256];
    const int tid = threadIdx.x;
    // multi-rank: the host passes the minimum over the ranks
    const int ncols = force_ncols >= 0 ? force_ncols : (int)(spec->spec_ok < spec->committed ? spec->spec_ok : spec->committed);
    double bmax = -1.0, bsum = 0.0, csum[16], inv[16];
    long long bidx = 0x7fffffffffffffffLL;
#pragma unroll
    for (int t = 0; t < 16; ++t) { csum[t] = 0.0; inv[t] = wn2[t]; }
    for (long long v = (long long)blockIdx.x * 256 + tid; v < n_vert; v += (long long)gridDim.x * 256) {
        double loss = 0.0;
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (t < ncols) {
                const double* p = comps + (long long)(k0 + t) * comp_stride + v * 3;
                const double q = (p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) * inv[t];
                loss += q;
                csum[t] += q;
            }
        double e = E[v] - loss;
        if (e < 0.0) e = 0.0;
        E[v] = e;
        bsum += e;
        if (am_better(e, v, bmax, bidx)) { bmax = e; bidx = v; }
    }
    block_sum<16>(csum, sh_d);
    if (tid < 16) colpart[(long long)blockIdx.x * 16 + tid] = csum[tid];
    __syncthreads();
    sh_d[tid] = bmax; sh_d[256 + tid] = bsum; sh_i[tid] = bidx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            sh_d[256 + tid] += sh_d[256 + tid + o];
            if (am_better(sh_d[tid + o], sh_i[tid + o], sh_d[tid], sh_i[tid])) { sh_d[tid] = sh_d[tid + o]; sh_i[tid] = sh_i[tid + o]; }
        }
        __syncthreads();
    }
    if (tid == 0) { pmax[blockIdx.x] = sh_d[0]; pidx[blockIdx.x] = sh_i[0]; psum[blockIdx.x] = sh_d[256]; }
}

// --------------------------------------------------------------------------------------
// panel weights in B-operand order: Wt[f][t] (Fp x 16), zero columns beyond ncols
// --------------------------------------------------------------------------------------
// Wt[f][t] -= sum_{j < kb + t} W[j][f] (w_j . w_t) / |w_j|^2: the panel's weights made orthogonal, in exact arithmetic
// terms, to every earlier weight vector (earlier panels, earlier sub-panels of the same read, earlier columns of the
// panel).  The w_k are orthogonal up to rounding already; what is removed here is the eps-sized leakage that k_correct
// otherwise takes out of the COEFFICIENTS after the pass (X w~_t = X w_t - sum_j (X w_j)(w_j . w_t) / |w_j|^2, which is the
// correction with the uncorrected c_j in place of the corrected ones: a second-order difference) -- at the cost of
// K x 16 x F flops instead of a sweep over all earlier coefficient columns.
__global__ __launch_bounds__(256) void k_orth_wt(const double* __restrict__ W, const double* __restrict__ G,
                                                 long long kb, int ncols, int Fp, double* __restrict__ Wt) {
    const long long total = (long long)Fp * ASB_PANEL_COLS;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(i % ASB_PANEL_COLS);
        const long long f = i / ASB_PANEL_COLS;
        if (t >= ncols) continue;
        double s = 0.0;
        double s2 = 0.0;
        long long j = 0;
        for (; j + 2 <= kb + t; j += 2) {          // G: already divided by |w_j|^2 (k_panel_gram)
            s += W[j * Fp + f] * G[j * 16 + t];
            s2 += W[(j + 1) * Fp + f] * G[(j + 1) * 16 + t];
        }
        if (j < kb + t) s += W[j * Fp + f] * G[j * 16 + t];
        Wt[i] -= s + s2;
    }
}
__global__ __launch_bounds__(256) void k_build_wt(const double* __restrict__ W, const double* __restrict__ scal,
                                                  long long k0, int ncols, int Fp, double* __restrict__ Wt,
                                                  double* __restrict__ wn2) {
    const long long total = (long long)Fp * ASB_PANEL_COLS;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(i % ASB_PANEL_COLS);
        const long long f = i / ASB_PANEL_COLS;
        Wt[i] = (t < ncols) ? W[(k0 + t) * Fp + f] : 0.0;
    }
    if (blockIdx.x == 0 && threadIdx.x < ASB_PANEL_COLS)
        wn2[threadIdx.x] = (threadIdx.x < ncols) ? scal[(k0 + threadIdx.x) * 4 + 1] : 1.0;
}

// B-operand panel from a frame-major (F x ldw) matrix: columns k0 .. k0+ncols-1, unit norms
__global__ __launch_bounds__(256) void k_build_wt_fk(const double* __restrict__ Wfk, long long ldw, long long k0,
                                                     int ncols, int F, int Fp, double* __restrict__ Wt,
                                                     double* __restrict__ wn2, const double* __restrict__ col_scale) {
    const long long total = (long long)Fp * ASB_PANEL_COLS;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(i % ASB_PANEL_COLS);
        const long long f = i / ASB_PANEL_COLS;
        Wt[i] = (t < ncols && f < F) ? Wfk[f * ldw + k0 + t] : 0.0;
    }
    if (blockIdx.x == 0 && threadIdx.x < ASB_PANEL_COLS)
        wn2[threadIdx.x] = (col_scale && (int)threadIdx.x < ncols) ? col_scale[k0 + threadIdx.x] : 1.0;
}

// --------------------------------------------------------------------------------------
// k_project_lds: the same product Y = X . Wt, organised for memory-level parallelism.
// One SWEEP covers nf <= 1008 frames: that slice of Wt sits in LDS (<= 126 KB, rows permuted so
// that the four 16-lane groups of a ds_read_b64 hit disjoint bank halves), every WAVE is
// independent: it pulls 16-row tiles from an atomic counter, streams the tile's half-rows from
// HBM straight into the MFMA A position (32 B per lane = whole 128-byte lines per row) and keeps
// only 4 accumulator registers -- no cross-wave reduction, no barrier in the loop, 16 waves per CU
// with several KB in flight each.  F > 1008 takes several sweeps (launches); the 16x16 partial
// tiles travel through a small scratch buffer (rows x 16 doubles).
// --------------------------------------------------------------------------------------
__device__ __forceinline__ int wt_perm(int f) { return (f & ~5) | ((f & 1) << 2) | ((f >> 2) & 1); }

// --------------------------------------------------------------------------------------
// k_project_l2: single sweep over the FULL rows.  The B operand comes from L2 instead of LDS: the panel
// is stored in MFMA-lane order, Wq[chunk][g][i][j] = W_panel[16*chunk + 4g + j][i], so a lane fetches its four
// B values of a chunk with one 32-byte load -- the same instruction count as the A operand -- and the
// 256 KB panel stays L2-resident.  No LDS, no partial tiles, one launch per pass; whole rows are streamed
// (the access pattern measured at 5.7 TB/s by tools/probe_stream_patterns.hip).
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_build_wq(const double* __restrict__ Wt, int Fp, double* __restrict__ Wq,
                                                  unsigned* __restrict__ tile_counter) {
    if (blockIdx.x == 0 && threadIdx.x < 16) tile_counter[threadIdx.x] = 0u;      // the projection kernel's work queue
    const long long total = (long long)Fp * 16;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(e & 3), i = (int)((e >> 2) & 15), g = (int)((e >> 6) & 3);
        const long long chunk = e >> 8;
        Wq[e] = Wt[(chunk * 16 + 4 * g + j) * ASB_PANEL_COLS + i];
    }
}

// k_project_l2s: the same product with S waves sharing one row tile (each takes every S-th group of G frame chunks and
// the partial accumulators are summed through LDS in a fixed order).  A wave's pass over its tile is a chain of dependent
// load -> MFMA groups, i.e. latency-bound (~60 groups x the HBM latency for F = 2000): with one wave per tile the queue
// drains for that long at the end of every launch with fewer and fewer loads in flight; S waves per tile cut that tail
// S times while the bytes in flight per CU stay the same.  The quad also reads S * G * 128 B contiguous bytes of a row at
// a time instead of G * 128.
template <int NT, int G, int S, int NQ, int OCC = 1>
__global__ __launch_bounds__(64 * S * NQ, OCC) void k_project_l2s(
    const double* __restrict__ X, long long rows, int Fp, const double* __restrict__ Wq, const double* __restrict__ wn2,
    int ncols, double* __restrict__ comps, long long comp_stride, unsigned int* __restrict__ counter) {
    // NQ tiles in flight per block (NQ * S waves): NQ = 1 keeps the barriers inside the group of waves that shares a tile
    __shared__ double red[NQ][S - 1][NT][4][64];
    __shared__ unsigned int tile_sh[NQ];
    const int l = threadIdx.x & 63, i = l & 15, g = l >> 4, w = threadIdx.x >> 6, quad = w / S, sub = w % S;
    constexpr int TR = 16 * NT;
    const long long ntiles = (rows + TR - 1) / TR;
    const int nchunk = Fp / 16;
    const double4* wq = reinterpret_cast<const double4*>(Wq) + (g * 16 + i);      // chunk c: wq[64 * c]
    for (;;) {
        if (sub == 0 && l == 0) tile_sh[quad] = atomicAdd(counter, 1u);
        __syncthreads();
        const unsigned int t = tile_sh[quad];
        const bool live = (long long)t < ntiles;
        if (!__syncthreads_or(live)) break;                     // tile ids only grow: every group of waves is past the end
        d4 acc[NT];
#pragma unroll
        for (int m = 0; m < NT; ++m) acc[m] = (d4){0.0, 0.0, 0.0, 0.0};
        if (live) {
            const double4* xp[NT];                              // chunk c: xp[m][4 * c]
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                long long r = (long long)t * TR + 16 * m + i;
                if (r >= rows) r = rows - 1;
                xp[m] = reinterpret_cast<const double4*>(X + r * Fp + 4 * g);
            }
            double4 a[NT][G], bq[G], an[NT][G], bn[G];
            int gi = sub;                                       // group gi = chunks gi * G .. gi * G + G - 1
            if (gi * G < nchunk) {
#pragma unroll
                for (int q = 0; q < G; ++q) {
                    const int c = gi * G + q < nchunk ? gi * G + q : nchunk - 1;
                    bq[q] = wq[64 * c];
#pragma unroll
                    for (int m = 0; m < NT; ++m) a[m][q] = xp[m][4 * c];
                }
            }
            while (gi * G < nchunk) {
                const int gn = gi + S;
                if (gn * G < nchunk) {                          // the next group's operands fly while this one's MFMAs issue
#pragma unroll
                    for (int q = 0; q < G; ++q) {
                        const int c = gn * G + q < nchunk ? gn * G + q : nchunk - 1;
                        bn[q] = wq[64 * c];
#pragma unroll
                        for (int m = 0; m < NT; ++m) an[m][q] = xp[m][4 * c];
                    }
                }
#pragma unroll
                for (int q = 0; q < G; ++q)
                    if (gi * G + q < nchunk) {
#pragma unroll
                        for (int m = 0; m < NT; ++m) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m][q].x, bq[q].x, acc[m], 0, 0, 0);
#pragma unroll
                        for (int m = 0; m < NT; ++m) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m][q].y, bq[q].y, acc[m], 0, 0, 0);
#pragma unroll
                        for (int m = 0; m < NT; ++m) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m][q].z, bq[q].z, acc[m], 0, 0, 0);
#pragma unroll
                        for (int m = 0; m < NT; ++m) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m][q].w, bq[q].w, acc[m], 0, 0, 0);
                    }
#pragma unroll
                for (int q = 0; q < G; ++q) {
                    bq[q] = bn[q];
#pragma unroll
                    for (int m = 0; m < NT; ++m) a[m][q] = an[m][q];
                }
                gi = gn;
            }
            if (sub > 0) {
#pragma unroll
                for (int m = 0; m < NT; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q) red[quad][sub - 1][m][q][l] = acc[m][q];
            }
        }
        __syncthreads();
        if (live && sub == 0) {
#pragma unroll
            for (int s2 = 0; s2 < S - 1; ++s2)
#pragma unroll
                for (int m = 0; m < NT; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[m][q] += red[quad][s2][m][q][l];
            if (i < ncols) {
                const double inv = wn2[i];
                double* dst = comps + (long long)i * comp_stride + (long long)t * TR + g;
#pragma unroll
                for (int m = 0; m < NT; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if ((long long)t * TR + 16 * m + g + 4 * q < rows) dst[16 * m + 4 * q] = acc[m][q] / inv;
            }
        }
        // the next iteration's first barrier (tile ids) also orders these reads of `red` before its next writes
    }
}

// --------------------------------------------------------------------------------------
// k_project_wide: the same product for NCT column tiles in ONE read of X (super-panels: up to 3 sub-panels of <= 16
// committed components each).  Wq / wn2 hold the tiles back to back (tile ct at Wq + ct * Fp * 16, wn2 + 16 ct); tile
// ct writes its columns i < nc[ct] to component rows kb[ct] + i.  MAP: the rows are those of the vertices vmap[.]
// (the band of a super-panel) instead of all of the shard's.
// --------------------------------------------------------------------------------------
#define ASB_MAX_SUB 8          // sub-panels (16-column tiles) per read of X
struct WideArgs { long long kb[ASB_MAX_SUB]; int nc[ASB_MAX_SUB]; };
// k_project_l2w: the NCT-tile product with S waves per row tile (k_project_l2s's split: wave `sub` takes every S-th group of
// G frame chunks, partial accumulators summed through LDS in a fixed order) and BOTH operands of the next group in flight
// while the current group's 4 * NT * NCT * G MFMAs issue.  k_project_wide gives a whole 64-row tile to one wave -- 125 chunks
// x 32 MFMAs x 64 cycles with two waves per SIMD = 210 us per tile, 2.3 tiles per wave on config 4: a third of the launch
// is the queue draining -- and fetches its L2 operand right in front of the MFMAs that need it.
template <int NT, int G, int S, int NCT, int OCC = 1, int PD = 1, int MODE = 0>      // MODE != 0: timing probes (asb_test_l2w_probe)
__global__ __launch_bounds__(64 * S, OCC) void k_project_l2w(
    const double* __restrict__ X, long long rows, int Fp, const double* __restrict__ Wq, const double* __restrict__ wn2,
    WideArgs wa, double* __restrict__ comps, long long comp_stride, unsigned int* __restrict__ counter) {
    extern __shared__ double l2w_lds[];                        // (S - 1) * NT * NCT * 4 * 64 doubles (64 KB for 8 tiles) + the tile id
    typedef double (*red_t)[NT][NCT][4][64];
    red_t red = reinterpret_cast<red_t>(l2w_lds);
    unsigned int& tile_sh = *reinterpret_cast<unsigned int*>(l2w_lds + (size_t)(S - 1) * NT * NCT * 4 * 64);
    const int l = threadIdx.x & 63, i = l & 15, g = l >> 4, sub = threadIdx.x >> 6;
    constexpr int TR = 16 * NT;
    const long long ntiles = (rows + TR - 1) / TR;
    const int nchunk = Fp / 16;
    const double4* wq[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) wq[ct] = reinterpret_cast<const double4*>(Wq + (long long)ct * Fp * 16) + (g * 16 + i);
    for (;;) {
        if (threadIdx.x == 0) tile_sh = atomicAdd(counter, 1u);
        __syncthreads();
        const unsigned int t = tile_sh;
        if ((long long)t >= ntiles) break;                      // the same for every wave of the block
        d4 acc[NT][NCT];
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[m][ct] = (d4){0.0, 0.0, 0.0, 0.0};
        const double4* xp[NT];                                  // chunk c: xp[m][4 * c]
#pragma unroll
        for (int m = 0; m < NT; ++m) {
            long long r = (long long)(MODE == 1 ? t % 4 : t) * TR + 16 * m + i;      // probe 1: four cache-resident tiles
            if (r >= rows) r = rows - 1;
            xp[m] = reinterpret_cast<const double4*>(X + r * Fp + 4 * g);
        }
        double4 a[NT][G], an[NT][G], b[NCT][G], bn[NCT][G];
        int gi = sub;                                           // group gi = chunks gi * G .. gi * G + G - 1
#define ASB_L2W_LOAD_A(DST, GRP)                                                                       \
    _Pragma("unroll") for (int q = 0; q < G; ++q) {                                                   \
        const int c = (GRP) * G + q < nchunk ? (GRP) * G + q : nchunk - 1;                             \
        _Pragma("unroll") for (int m = 0; m < NT; ++m) DST[m][q] = xp[m][4 * c];                       \
    }
#define ASB_L2W_LOAD_B(DST, GRP)                                                                       \
    _Pragma("unroll") for (int q = 0; q < G; ++q) {                                                   \
        const int c = (GRP) * G + q < nchunk ? (GRP) * G + q : nchunk - 1;                             \
        _Pragma("unroll") for (int ct = 0; ct < NCT; ++ct) DST[ct][q] = wq[ct][64 * c];                \
    }
#define ASB_L2W_MFMA(GRP)                                                                              \
    _Pragma("unroll") for (int q = 0; q < G; ++q)                                                     \
        if (MODE == 2) {          /* probe 2: the loads alone */                                       \
            _Pragma("unroll") for (int ct = 0; ct < NCT; ++ct)                                        \
                _Pragma("unroll") for (int m = 0; m < NT; ++m)                                        \
                    acc[m][ct][0] += (a[m][q].x + a[m][q].y) * b[ct][q].x + (a[m][q].z + a[m][q].w) * b[ct][q].w; \
        } else if ((GRP) * G + q < nchunk) {                                                           \
            _Pragma("unroll") for (int ct = 0; ct < NCT; ++ct) {                                      \
                _Pragma("unroll") for (int m = 0; m < NT; ++m)                                        \
                    acc[m][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m][q].x, b[ct][q].x, acc[m][ct], 0, 0, 0); \
                _Pragma("unroll") for (int m = 0; m < NT; ++m)                                        \
                    acc[m][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m][q].y, b[ct][q].y, acc[m][ct], 0, 0, 0); \
                _Pragma("unroll") for (int m = 0; m < NT; ++m)                                        \
                    acc[m][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m][q].z, b[ct][q].z, acc[m][ct], 0, 0, 0); \
                _Pragma("unroll") for (int m = 0; m < NT; ++m)                                        \
                    acc[m][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m][q].w, b[ct][q].w, acc[m][ct], 0, 0, 0); \
            }                                                                                          \
        }
        if (PD == 2) {
            // the HBM operand two groups ahead, the L2 operand one: 2 * NT * G * 2 KB in flight per wave
            double4 a2[NT][G];
            if (gi * G < nchunk) {
                ASB_L2W_LOAD_B(b, gi)
                ASB_L2W_LOAD_A(a, gi)
                if ((gi + S) * G < nchunk) { ASB_L2W_LOAD_A(an, gi + S) }
            }
            while (gi * G < nchunk) {
                // B first: loads retire in order, and this group's MFMAs wait for the B issued one iteration ago -- anything
                // issued in front of it (an A two groups ahead) would have to have landed as well
                if ((gi + S) * G < nchunk) { ASB_L2W_LOAD_B(bn, gi + S) }
                if ((gi + 2 * S) * G < nchunk) { ASB_L2W_LOAD_A(a2, gi + 2 * S) }
                ASB_L2W_MFMA(gi)
#pragma unroll
                for (int q = 0; q < G; ++q) {
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) b[ct][q] = bn[ct][q];
#pragma unroll
                    for (int m = 0; m < NT; ++m) { a[m][q] = an[m][q]; an[m][q] = a2[m][q]; }
                }
                gi += S;
            }
        } else {
            if (gi * G < nchunk) {
                ASB_L2W_LOAD_B(b, gi)
                ASB_L2W_LOAD_A(a, gi)
            }
            while (gi * G < nchunk) {
                const int gn = gi + S;
                if (gn * G < nchunk) {
                    ASB_L2W_LOAD_B(bn, gn)
                    ASB_L2W_LOAD_A(an, gn)
                }
                ASB_L2W_MFMA(gi)
#pragma unroll
                for (int q = 0; q < G; ++q) {
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) b[ct][q] = bn[ct][q];
#pragma unroll
                    for (int m = 0; m < NT; ++m) a[m][q] = an[m][q];
                }
                gi = gn;
            }
        }
#undef ASB_L2W_LOAD_A
#undef ASB_L2W_LOAD_B
#undef ASB_L2W_MFMA
        if (sub > 0) {
#pragma unroll
            for (int m = 0; m < NT; ++m)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int q = 0; q < 4; ++q) red[sub - 1][m][ct][q][l] = acc[m][ct][q];
        }
        __syncthreads();
        if (sub == 0) {
#pragma unroll
            for (int s2 = 0; s2 < S - 1; ++s2)
#pragma unroll
                for (int m = 0; m < NT; ++m)
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[m][ct][q] += red[s2][m][ct][q][l];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                if (i < wa.nc[ct]) {
                    const double inv = wn2[16 * ct + i];
                    double* dst = comps + (wa.kb[ct] + i) * comp_stride + (long long)t * TR + g;
#pragma unroll
                    for (int m = 0; m < NT; ++m)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if ((long long)t * TR + 16 * m + g + 4 * q < rows) dst[16 * m + 4 * q] = acc[m][ct][q] / inv;
                }
        }
        // the next iteration's first barrier (tile id) also orders these reads of `red` before its next writes
    }
}

// --------------------------------------------------------------------------------------
// k_project_l2d (round 3): k_project_l2c with the two things its probes said it was losing.
//  (1) BALANCE.  1172 tiles of 256 rows drawn from a queue by 256 resident blocks are 4.58 rounds of work done in 5: the
//      last round keeps 58 % of the CUs busy (8 % of an MFMA-bound launch).  Here every block owns a contiguous range of
//      16-row groups (73 or 74 of the 18 750 at config 4), cut into rounds of at most 16 groups spread evenly over the four
//      row-tile waves: 4 4 4 4 groups per wave and round, then 3 2 2 2 -- 19 group-times per block against 20, where 18.3 is
//      the ideal.  A wave's number of groups NTV in {0 .. 4} is a template parameter of the tile body (one instantiation
//      each: the counted waits need the same number of loads on every path of a body).
//  (2) NO BLOCK BARRIER PER STAGE.  A stage's weights are complete when all 8 waves' direct loads have landed; a wave knows
//      that of its OWN loads behind the next wait for an X chunk requested after them, so it then adds 1 to the stage's
//      arrival counter in LDS, and a wave entering stage s spins on that counter (usually already 8: the arrivals happened
//      two chunk pairs earlier) -- the waves no longer meet every stage, they only cannot run ahead of the data.  Three
//      stage buffers instead of two: the loads of stage s + 1 go out while stage s is computed and overwrite stage s - 2,
//      which every wave has left -- a wave can only be inside stage s once all 8 have ARRIVED for s, and a wave arrives
//      for s while it computes s - 1.
// Everything else as in k_project_l2c: 8 waves = (row tile rt) x (frame half sub), X chunks in two register sets used
// alternately, direct loads / LDS reads by hand so that the compiler counts one kind of pending load.
// --------------------------------------------------------------------------------------
template <int NCT, int P, int NTV, int SYNC, int NSB, int MODE, int XDP>
__device__ __forceinline__ void l2d_tile(const double* __restrict__ X, long long rows, int Fp, const double* __restrict__ Wq,
                                         const double* __restrict__ wn2, const WideArgs& wa, double* __restrict__ comps,
                                         long long comp_stride, long long group0, double* lds, unsigned cnt_byte) {
    constexpr int RT = 4, NTM = 4, NI = 4 * P * NCT, PER = NI / (2 * RT), NS = NSB;
    // XDP = XD + 10 BPF.  BPF (round 4): the weights of column tile ct + 1 are read from LDS while tile ct's 16 MFMAs issue (a second
    // pair of operand registers), instead of read-and-wait in front of every tile: one exposed LDS latency per chunk instead of NCT
    constexpr int XD = XDP % 10;
    constexpr bool BPF = XDP >= 10;
    // SYNC: 0 = block barrier per stage (two buffers); 1 = arrival counters over all 8 waves; 5 = arrival counters PER FRAME-HALF
    // GROUP of four waves (each group stages the chunks only it reads: a slow wave holds up three others, not seven);
    // 2 / 3 = debugging forms (barrier with three buffers / counters and barrier)
    constexpr bool CNT = (SYNC & 1) != 0, BAR = SYNC == 0 || SYNC == 2 || SYNC == 3, GRP = SYNC == 5;
    constexpr int DIST = CNT ? (NS - 1) / 2 : 1;                // the direct loads run DIST stages ahead of the MFMAs
    // a wave inside stage s knows only that every wave has ARRIVED for s, which a wave does while it computes stage s - DIST: the
    // buffer the loads of stage s + DIST overwrite must belong to stage s - DIST - 1 or older (round 3 measured "four buffers,
    // two stages ahead" at 1.37-1.39 ms -- and one rejected tile in twenty steps: it overwrote a stage still being read)
    static_assert(!CNT || NS >= 2 * DIST + 1, "counters: the stage being overwritten must have been left by every wave");
    constexpr int STAGE_D = NI * 128;                           // doubles per stage
    constexpr int NTA = NTV > 0 ? NTV : 1;                      // (array bounds of a wave without rows)
    typedef double (*red_t)[NTM][4][64];
    red_t red = reinterpret_cast<red_t>(lds);                   // shares the stages' memory (used behind the tile's last MFMA)
    const int tid = threadIdx.x, l = tid & 63, i = l & 15, g = l >> 4, w = tid >> 6, rt = w % RT, sub = w / RT;
    const int nchunk = Fp / 16, npair = (nchunk + 1) / 2, nstage = (npair + P - 1) / P;
    const long long base = group0 * 16;
    auto issue_stage = [&](int s) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            // all 8 waves share a stage (instruction q of NI: chunk cc of the stage's 2 P, column tile, half), or each group of
            // four stages its own P chunks (instruction q of NI / 2 behind the group's base)
            const int q = (GRP ? rt : w) * PER + u, h = q & 1, ct = (q >> 1) % NCT, cc = (q >> 1) / NCT;
            const int c = GRP ? 2 * (s * P + cc) + sub : 2 * s * P + cc;
            if (c < nchunk) {
                const double* src = Wq + (long long)ct * Fp * 16 + (long long)c * 256 + l * 4 + h * 2;
                const unsigned lds_byte = __builtin_amdgcn_readfirstlane(
                    (unsigned)((GRP ? ((s % NS) * 2 + sub) * (STAGE_D / 2) + q * 128 : (s % NS) * STAGE_D + q * 128) * 8));
                unsigned m0_keep;                                // m0 is the compiler's: put back what it held
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(m0_keep) : "v"(src), "s"(lds_byte) : "memory");
            }
        }
    };
    // arrival for stage s: this wave's direct loads of it have landed (callers: behind a wait that covers them)
    auto arrive = [&](int s) {
        const unsigned addr = cnt_byte + 4u * (unsigned)(GRP ? (s % NS) * 2 + sub : s % NS);      // LDS byte address (the dynamic LDS starts at 0)
        const unsigned one = 1u;                                       // (every lane adds: 64 per wave, 512 / 256 per stage)
        asm volatile("ds_add_u32 %0, %1" :: "v"(addr), "v"(one) : "memory");
    };
    // entry into stage s: all 8 waves have arrived for it (the buffer's (s / NS + 1)-th use in this tile)
    auto enter = [&](int s) {
        const unsigned addr = cnt_byte + 4u * (unsigned)(GRP ? (s % NS) * 2 + sub : s % NS);
        const unsigned want = (GRP ? 256u : 512u) * (unsigned)(s / NS + 1);       // (ds_add_u32 is per lane: a wave adds 64)
        unsigned got, sg;
        asm volatile("1:\n\tds_read_b32 %0, %2\n\ts_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, %0\n\ts_cmp_lt_u32 %1, %3\n\ts_cbranch_scc1 1b"
                     : "=&v"(got), "=&s"(sg) : "v"(addr), "s"(want) : "memory", "scc");
    };
    const double4* xp[NTA];                                     // chunk c: xp[m][4 * c]
#pragma unroll
    for (int m = 0; m < NTA; ++m) {
        long long r = base + 16 * m + i;
        if (r >= rows) r = rows - 1;
        xp[m] = reinterpret_cast<const double4*>(X + r * Fp + 4 * g);
    }
    d4 acc[NTA][NCT];
#pragma unroll
    for (int m = 0; m < NTA; ++m)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[m][ct] = (d4){0.0, 0.0, 0.0, 0.0};
    // XD = 2: a THIRD register set, the X chunk of pair j + 2 is requested while pair j computes (twice the tolerance for
    // a slow HBM access; 32 more registers)
    static_assert(XD == 1 || XD == 2, "prefetch distance");
    static_assert(!CNT || XD < P, "the arrival for the next stage must fall inside the stage");
    double4 a0[NTA], a1[NTA], a2[XD == 2 ? NTA : 1];
#pragma unroll
    for (int d = 0; d < DIST; ++d)
        if (d < nstage) issue_stage(d);
    if (NTV > 0) {
        const int c = sub < nchunk ? sub : nchunk - 1;
#pragma unroll
        for (int m = 0; m < NTA; ++m) a0[m] = xp[m][4 * c];
        if (XD == 2) {
            const int c1 = 2 + sub < nchunk ? 2 + sub : nchunk - 1;
#pragma unroll
            for (int m = 0; m < NTA; ++m) a1[m] = xp[m][4 * c1];
        }
    }
    __builtin_amdgcn_s_waitcnt(0);                              // first stage(s) and first X chunk of the tile: one exposed latency
    if (CNT) {
#pragma unroll
        for (int d = 0; d < DIST; ++d)
            if (d < nstage) arrive(d);
    }
    auto pair_step = [&](int j, double4 (&cur)[NTA], double4 (&nxt)[NTA]) {
        const int s = j / P;
        if (j == s * P) {
            if (CNT && MODE != 5) enter(s);
            if (BAR && MODE != 5) __syncthreads();
        }
        if (NTV > 0 && (MODE == 3 || MODE == 4)) {                    // probes: no X traffic inside the loop
#pragma unroll
            for (int m = 0; m < NTA; ++m) nxt[m] = cur[m];
        } else if (NTV > 0) {
            // the next pair's X chunk flies while this pair's MFMAs issue -- unconditionally (behind the last pair: the last
            // chunk once more): the same number of loads on every path keeps the compiler's waits counted ones
            const int cn = 2 * (j + XD) + sub < nchunk ? 2 * (j + XD) + sub : nchunk - 1;
#pragma unroll
            for (int m = 0; m < NTA; ++m) nxt[m] = xp[m][4 * cn];
        }
        // pair XD of a stage: the loads of stage s + DIST (issued in the first pair, older than the XD chunks requested since)
        // have landed once everything but the newest XD * 2 NTV loads is back -- the chunk this pair computes from among them
        if (CNT && j == s * P + XD && s + DIST < nstage) {
            if (MODE == 3 || MODE == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (XD * NTV == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (XD * NTV == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (XD * NTV == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (XD * NTV == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (XD * NTV == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (XD * NTV == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            arrive(s + DIST);
        }
        bool issued = false;
        if (NTV > 0 && 2 * j + sub < nchunk) {
            const int cc = 2 * (j - s * P) + sub;
            auto b_addr = [&](int ct) {
                return GRP ? (unsigned)((((s % NS) * 2 + sub) * (STAGE_D / 2) + (((j - s * P) * NCT + ct) * 2) * 128 + l * 2) * 8)
                           : (unsigned)(((s % NS) * STAGE_D + ((cc * NCT + ct) * 2) * 128 + l * 2) * 8);
            };
            auto mfma16 = [&](int ct, const auto& b0, const auto& b1) {
#pragma unroll
                for (int m = 0; m < NTA; ++m) acc[m][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[m].x, b0.x, acc[m][ct], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < NTA; ++m) acc[m][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[m].y, b0.y, acc[m][ct], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < NTA; ++m) acc[m][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[m].z, b1.x, acc[m][ct], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < NTA; ++m) acc[m][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[m].w, b1.y, acc[m][ct], 0, 0, 0);
                // the next stage's direct loads go out BEHIND the wait for the current X chunk (the first MFMAs above)
                if (ct == 0 && j == s * P && s + DIST < nstage) { issue_stage(s + DIST); issued = true; }
            };
            if (BPF && MODE != 4) {
                // tile ct's operands are requested one tile earlier: request ct + 1's, then wait for everything but those (LDS
                // reads return in order; the in/out operands keep the tile's MFMAs behind the wait) -- two operand pairs, A and B
                typedef double d2v __attribute__((ext_vector_type(2)));      // (a native vector: in/out asm operands of a struct type are not supported)
                d2v pa0, pa1, pb0, pb1;
                auto ahead = [&](int ct, d2v& n0, d2v& n1, d2v& c0, d2v& c1) {
                    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\ts_waitcnt lgkmcnt(2)"
                                 : "=&v"(n0), "=&v"(n1), "+v"(c0), "+v"(c1) : "v"(b_addr(ct + 1)));
                };
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(pa0), "=&v"(pa1) : "v"(b_addr(0)));
                static_assert(!BPF || NCT == 4, "operands read ahead: written out for four column tiles");
                ahead(0, pb0, pb1, pa0, pa1);
                mfma16(0, pa0, pa1);
                ahead(1, pa0, pa1, pb0, pb1);
                mfma16(1, pb0, pb1);
                ahead(2, pb0, pb1, pa0, pa1);
                mfma16(2, pa0, pa1);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pb0), "+v"(pb1));
                mfma16(3, pb0, pb1);
            } else {
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    double2 b0, b1;
                    if (MODE == 4) {                            // probe: no LDS reads either
                        b0 = make_double2(cur[0].x, cur[0].y);
                        b1 = make_double2(cur[0].z, cur[0].w);
                    } else {
                        const unsigned lds_addr = b_addr(ct);
                        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(b0), "=&v"(b1) : "v"(lds_addr));
                    }
                    mfma16(ct, b0, b1);
                }
            }
        }
        if (!issued && j == s * P && s + DIST < nstage) issue_stage(s + DIST);
    };
    if (XD == 1) {
        for (int j = 0; j < npair; j += 2) {
            pair_step(j, a0, a1);
            if (j + 1 < npair) pair_step(j + 1, a1, a0);
        }
    } else {
        auto& b2 = reinterpret_cast<double4 (&)[NTA]>(a2);
        for (int j = 0; j < npair; j += 3) {
            pair_step(j, a0, b2);
            if (j + 1 < npair) pair_step(j + 1, a1, a0);
            if (j + 2 < npair) pair_step(j + 2, b2, a1);
        }
    }
    // the two frame halves of a row tile meet through LDS, one column tile at a time (fixed order: half 0 + half 1)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        __syncthreads();            // (first: every wave is behind its last read of a stage -- `red` shares that memory)
        if (NTV > 0 && sub == 1) {
#pragma unroll
            for (int m = 0; m < NTA; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) red[rt][m][q][l] = acc[m][ct][q];
        }
        __syncthreads();
        if (NTV > 0 && sub == 0 && i < wa.nc[ct]) {
            const double inv = wn2[16 * ct + i];
            double* dst = comps + (wa.kb[ct] + i) * comp_stride + base + g;
#pragma unroll
            for (int m = 0; m < NTA; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (base + 16 * m + g + 4 * q < rows) dst[16 * m + 4 * q] = (acc[m][ct][q] + red[rt][m][q][l]) / inv;
        }
    }
}

template <int NCT, int P, int SYNC = 1, int NSB = (SYNC ? 3 : 2), int MODE = 0, int XD = 1>      // MODE: timing probes (asb_test_l2w_probe)
__global__ __launch_bounds__(512, 2) void k_project_l2d(
    const double* __restrict__ X, long long rows, int Fp, const double* __restrict__ Wq, const double* __restrict__ wn2,
    WideArgs wa, double* __restrict__ comps, long long comp_stride) {
    constexpr int NI = 4 * P * NCT, NS = NSB, STAGE_D = NI * 128, RED_D = 4 * 4 * 4 * 64;
    constexpr int BODY_D = NS * STAGE_D > RED_D ? NS * STAGE_D : RED_D;
    static_assert(NI % 8 == 0, "stage instructions must divide over the 8 waves");
    static_assert(P >= 2, "a stage must hold at least two chunk pairs (arrival / completion of the direct loads)");
    static_assert((BODY_D + 4) * 8 <= 160 * 1024, "the stages must fit the LDS of a CU (gfx950: 160 KB)");
    extern __shared__ double l2d_lds[];
    unsigned* cnt = reinterpret_cast<unsigned*>(l2d_lds + BODY_D);          // NS arrival counters
    const int tid = threadIdx.x, rt = (tid >> 6) & 3;
    const long long ngroups = (rows + 15) / 16;
    const long long g0 = ngroups * blockIdx.x / gridDim.x, g1 = ngroups * (blockIdx.x + 1) / gridDim.x;
    constexpr unsigned CNT_BYTE = (unsigned)BODY_D * 8u;
    for (long long gr = g0; gr < g1; gr += 16) {
        const int n = (int)(g1 - gr < 16 ? g1 - gr : 16);                   // groups of this round, spread evenly over the 4 row-tile waves
        const int bs = n >> 2, ex = n & 3;
        const int ntv = bs + (rt < ex ? 1 : 0);
        const long long start = gr + rt * bs + (rt < ex ? rt : ex);
        __syncthreads();                                                    // the last round's reads of `red` are done
        if (tid < 2 * NS) cnt[tid] = 0u;
        __syncthreads();
        switch (ntv) {
            case 4: l2d_tile<NCT, P, 4, SYNC, NSB, MODE, XD>(X, rows, Fp, Wq, wn2, wa, comps, comp_stride, start, l2d_lds, CNT_BYTE); break;
            case 3: l2d_tile<NCT, P, 3, SYNC, NSB, MODE, XD>(X, rows, Fp, Wq, wn2, wa, comps, comp_stride, start, l2d_lds, CNT_BYTE); break;
            case 2: l2d_tile<NCT, P, 2, SYNC, NSB, MODE, XD>(X, rows, Fp, Wq, wn2, wa, comps, comp_stride, start, l2d_lds, CNT_BYTE); break;
            case 1: l2d_tile<NCT, P, 1, SYNC, NSB, MODE, XD>(X, rows, Fp, Wq, wn2, wa, comps, comp_stride, start, l2d_lds, CNT_BYTE); break;
            default: l2d_tile<NCT, P, 0, SYNC, NSB, MODE, XD>(X, rows, Fp, Wq, wn2, wa, comps, comp_stride, start, l2d_lds, CNT_BYTE); break;
        }
    }
}

// --------------------------------------------------------------------------------------
// k_project_l2e (round 4): the 64-column pass WITHOUT any synchronisation between waves.  k_project_l2d shares one staged copy of
// the weights among the 8 waves of a block and pays for it with a stage protocol (arrival counters; 0.11 of its 1.41 ms) and a
// cross-wave reduction.  A lane reads back from LDS exactly the 32 bytes per column tile it loaded itself (Wq is in MFMA lane
// order), so LDS is only a register-free FIFO: here every wave keeps a PRIVATE ring of two chunks (2 x 8 KB; 8 waves = 128 KB) that
// it fills itself with direct loads, one chunk ahead, next to its own X chunk -- 8x the L2 traffic for the weights (4.8 GB per
// launch, 13 % of what the L2s deliver), no barrier, no counters, no reduction: a wave owns its rows for all frames.
// MEASURED (round 4, ASB_WIDE_VARIANT=60; parity green): 1.456 - 1.475 ms per launch against 1.400 - 1.408 for k_project_l2d on the
// same box -- and 1.372 - 1.388 with the wait for the weights left out (wrong results): the stage protocol is not what k_project_l2d
// loses; a wave that waits for its own weights chunk by chunk loses more than eight waves sharing a stage.  Kept as a variant.
// --------------------------------------------------------------------------------------
template <int NCT, int NTV, int DBG>
__device__ __forceinline__ void l2e_rows(const double* __restrict__ X, long long rows, int Fp, const double* __restrict__ Wq,
                                         const double* __restrict__ wn2, const WideArgs& wa, double* __restrict__ comps,
                                         long long comp_stride, long long group0, unsigned lds_byte0) {
    static_assert(NTV >= 1 && NTV <= 4, "row groups per wave");
    typedef double d2v __attribute__((ext_vector_type(2)));
    const int l = threadIdx.x & 63, i = l & 15, g = l >> 4;
    const int nchunk = Fp / 16;
    const long long base = group0 * 16;
    const double4* xp[NTV];                                     // chunk c: xp[m][4 * c]
#pragma unroll
    for (int m = 0; m < NTV; ++m) {
        long long r = base + 16 * m + i;
        if (r >= rows) r = rows - 1;
        xp[m] = reinterpret_cast<const double4*>(X + r * Fp + 4 * g);
    }
    d4 acc[NTV][NCT];
#pragma unroll
    for (int m = 0; m < NTV; ++m)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[m][ct] = (d4){0.0, 0.0, 0.0, 0.0};
    double4 a0[NTV], a1[NTV];
    // the weights of chunk c into ring slot c & 1 (asm: the compiler counts only the X loads; the hardware retires both kinds in
    // the order issued, so the compiler's wait for the X chunk issued BEHIND these covers them -- as in k_project_l2d)
    auto issue_w = [&](int c) {
#pragma unroll
        for (int q = 0; q < 2 * NCT; ++q) {
            const int ct = q >> 1, h = q & 1;
            const double* src = Wq + (long long)ct * Fp * 16 + (long long)c * 256 + l * 4 + h * 2;
            const unsigned lds_byte = __builtin_amdgcn_readfirstlane(lds_byte0 + (unsigned)(((c & 1) * 2 * NCT + q) * 128 * 8));
            unsigned m0_keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(m0_keep) : "v"(src), "s"(lds_byte) : "memory");
        }
    };
    auto load_x = [&](int c, double4 (&a)[NTV]) {
        const int cc = c < nchunk ? c : nchunk - 1;            // (behind the end: the last chunk once more -- the same loads on every path)
#pragma unroll
        for (int m = 0; m < NTV; ++m) a[m] = xp[m][4 * cc];
    };
    auto compute = [&](int c, double4 (&a)[NTV]) {
        // this chunk's weights have landed once everything but the X chunk requested last (2 NTV loads) is back
        if (NTV == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (NTV == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (NTV == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            d2v b0, b1;
            const unsigned addr = lds_byte0 + (unsigned)((((c & 1) * 2 * NCT + 2 * ct) * 128 + l * 2) * 8);
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=&v"(b0), "=&v"(b1) : "v"(addr));
#pragma unroll
            for (int m = 0; m < NTV; ++m) acc[m][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].x, b0.x, acc[m][ct], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < NTV; ++m) acc[m][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].y, b0.y, acc[m][ct], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < NTV; ++m) acc[m][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].z, b1.x, acc[m][ct], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < NTV; ++m) acc[m][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].w, b1.y, acc[m][ct], 0, 0, 0);
            // the next chunk's weights go out BEHIND the wait for this chunk's X (the first MFMAs above): they have this chunk's
            // remaining 48 MFMAs and the other wave's 64 to arrive, and the next chunk's X wait (issued behind them) covers them
            if (ct == 0 && c + 1 < nchunk) issue_w(c + 1);
        }
    };
    load_x(0, a0);
    issue_w(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // first chunk of the rows: one exposed latency
    for (int c = 0; c < nchunk; c += 2) {
        load_x(c + 1, a1);
        compute(c, a0);
        load_x(c + 2, a0);
        if (c + 1 < nchunk) compute(c + 1, a1);
    }
    (void)DBG;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        if (i < wa.nc[ct]) {
            const double inv = wn2[16 * ct + i];
            double* dst = comps + (wa.kb[ct] + i) * comp_stride + base + g;
#pragma unroll
            for (int m = 0; m < NTV; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (base + 16 * m + g + 4 * q < rows) dst[16 * m + 4 * q] = acc[m][ct][q] / inv;
        }
    }
}
template <int NCT, int DBG = 0>
__global__ __launch_bounds__(512, 2) void k_project_l2e(const double* __restrict__ X, long long rows, int Fp, const double* __restrict__ Wq,
                                                        const double* __restrict__ wn2, WideArgs wa, double* __restrict__ comps,
                                                        long long comp_stride) {
    extern __shared__ double l2e_lds[];
    const int w = threadIdx.x >> 6;
    const unsigned lds_byte0 = (unsigned)(w * 2 * 2 * NCT * 128 * 8);        // this wave's ring: 2 chunks x 2 NCT KB
    const long long ngroups = (rows + 15) / 16;
    const long long g0 = ngroups * blockIdx.x / gridDim.x, g1 = ngroups * (blockIdx.x + 1) / gridDim.x;
    // the block's groups in contiguous shares per wave (9 or 10 of 73 - 74 at config 4), each share in turns of at most four
    const long long n = g1 - g0, w0 = g0 + n * w / 8, w1 = g0 + n * (w + 1) / 8;
    for (long long gr = w0; gr < w1; gr += 4) {
        const int ntv = (int)(w1 - gr < 4 ? w1 - gr : 4);
        switch (ntv) {
            case 4: l2e_rows<NCT, 4, DBG>(X, rows, Fp, Wq, wn2, wa, comps, comp_stride, gr, lds_byte0); break;
            case 3: l2e_rows<NCT, 3, DBG>(X, rows, Fp, Wq, wn2, wa, comps, comp_stride, gr, lds_byte0); break;
            case 2: l2e_rows<NCT, 2, DBG>(X, rows, Fp, Wq, wn2, wa, comps, comp_stride, gr, lds_byte0); break;
            default: l2e_rows<NCT, 1, DBG>(X, rows, Fp, Wq, wn2, wa, comps, comp_stride, gr, lds_byte0); break;
        }
    }
    (void)l2e_lds;
}

// scal[(k0+t)*4+3] = sum over blocks of colpart[b][t]  (= |w_t|^2 |c_t|_F^2 on this shard)
// scal[(k0 + t) * 4 + 3] = sum over the blocks' partial column sums; one wave per column (launch with 1024 threads)
__global__ __launch_bounds__(1024) void k_colsum(const double* __restrict__ colpart, int nblk, int ncols, long long k0,
                                                 double* __restrict__ scal, PanelState* __restrict__ spec = nullptr) {
    if (spec != nullptr) {              // panel with unproven steps: the columns that survived; the host reads `committed`
        ncols = (int)(spec->spec_ok < spec->committed ? spec->spec_ok : spec->committed);
        __syncthreads();
        if (threadIdx.x == 0) spec->committed = ncols;
    }
    const int t = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (t >= ncols) return;
    double v = 0.0;
    for (int b = lane; b < nblk; b += 64) v += colpart[(long long)b * 16 + t];
    v = wave_sum(v);
    if (lane == 0) scal[(k0 + t) * 4 + 3] = v;
}

// --------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------
template <int T, int E2>
static void launch_gather_te(asb_ctx* ctx, int grid, const long long* idx_map, long long n_items, const PanelState* panel,
                             int k0, double* dst, double* e_out, double* pm, long long* pi, double* ps) {
    constexpr int BLOCK = (T >= 256 ? T : 256);
    if constexpr (T == 256 && E2 == 4) {
        if (dst != nullptr && k0 >= 8 && ctx->gather_cpt == 2) {      // candidate rows late in a run: the L2 reads of W dominate
            // (same grid: the blocks without items still write their neutral partial records, which the caller counts)
            hipLaunchKernelGGL((k_gather<T, E2, 2>), dim3(grid), dim3(BLOCK), 0, ctx->stream, ctx->X, idx_map,
                               (long long)ctx->v0, n_items, panel, ctx->comps, (long long)(3 * ctx->n_loc), ctx->W, k0,
                               (int)(ctx->Fp / 2), dst, e_out, pm, pi, ps);
            return;
        }
    }
    hipLaunchKernelGGL((k_gather<T, E2>), dim3(grid), dim3(BLOCK), 0, ctx->stream, ctx->X, idx_map, (long long)ctx->v0,
                       n_items, panel, ctx->comps, (long long)(3 * ctx->n_loc), ctx->W, k0, (int)(ctx->Fp / 2), dst,
                       e_out, pm, pi, ps);
}
template <int E2>
static void launch_gather_e(asb_ctx* ctx, int T, int grid, const long long* idx_map, long long n_items,
                            const PanelState* panel, int k0, double* dst, double* e_out, double* pm, long long* pi,
                            double* ps) {
    switch (T) {
        case 64: launch_gather_te<64, E2>(ctx, grid, idx_map, n_items, panel, k0, dst, e_out, pm, pi, ps); break;
        case 128: launch_gather_te<128, E2>(ctx, grid, idx_map, n_items, panel, k0, dst, e_out, pm, pi, ps); break;
        case 256: launch_gather_te<256, E2>(ctx, grid, idx_map, n_items, panel, k0, dst, e_out, pm, pi, ps); break;
        case 512: launch_gather_te<512, E2>(ctx, grid, idx_map, n_items, panel, k0, dst, e_out, pm, pi, ps); break;
        default: launch_gather_te<1024, E2>(ctx, grid, idx_map, n_items, panel, k0, dst, e_out, pm, pi, ps); break;
    }
}
static void launch_gather(asb_ctx* ctx, const StreamCfg& c, int grid, const long long* idx_map, long long n_items,
                          const PanelState* panel, int k0, double* dst, double* e_out, double* pm, long long* pi,
                          double* ps) {
    switch (c.E2) {
        case 4: launch_gather_e<4>(ctx, c.T, grid, idx_map, n_items, panel, k0, dst, e_out, pm, pi, ps); break;
        case 8: launch_gather_e<8>(ctx, c.T, grid, idx_map, n_items, panel, k0, dst, e_out, pm, pi, ps); break;
        default: launch_gather_e<16>(ctx, c.T, grid, idx_map, n_items, panel, k0, dst, e_out, pm, pi, ps); break;
    }
}

static void launch_project(asb_ctx* ctx, int ncols, double* out);

// Per-panel read-back of the panel state (and the panel kernel's flags) through pinned host memory: ONE wait, no staging
// copy (a pageable destination costs an extra copy kernel and a second round trip for the flags).
// Round 2: instead of copy + hipStreamSynchronize (an interrupt-driven wait of 15-25 us, twice per panel), a one-wave kernel
// writes the state straight into coherent pinned host memory, its sequence number last (system-scope fence in between), and
// the host polls that word: the wait ends a microsecond or two after the producing kernels do.  Falls back to the
// synchronising copy if the word does not arrive (or with ASB_HOST_POLL=0).
__global__ __launch_bounds__(64) void k_publish_state(const PanelState* __restrict__ st, const unsigned* __restrict__ flags,
                                                      unsigned char* __restrict__ pin, unsigned long long seq) {
    const int l = threadIdx.x;
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(st);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(pin);
    if (l < (int)(sizeof(PanelState) / 8)) dst[l] = src[l];
    if (flags && l < 4) reinterpret_cast<unsigned*>(pin + 256)[l] = flags[l];
    __threadfence_system();
    __syncthreads();
    if (l == 0) {
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(pin + 448), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static int read_panel_state_from(asb_ctx* ctx, const PanelState* st, PanelState* h, unsigned* flags) {
    int rc = asb_pin_alloc(ctx);
    if (rc) return rc;
    static_assert(sizeof(PanelState) % 8 == 0 && sizeof(PanelState) <= 256, "PanelState must fit its pinned slot");
    if (ctx->host_poll && ctx->host_pin_dev) {
        const unsigned long long seq = ++ctx->pin_seq;
        hipLaunchKernelGGL(k_publish_state, dim3(1), dim3(64), 0, ctx->stream, st, flags ? ctx->coop_bar : (const unsigned*)nullptr,
                           ctx->host_pin_dev, seq);
        ASB_CHECK_LAUNCH(ctx);
        volatile unsigned long long* word = reinterpret_cast<volatile unsigned long long*>(ctx->host_pin + 448);
        const auto t0 = std::chrono::steady_clock::now();
        bool arrived = false;
        for (unsigned spins = 0;; ++spins) {
            if (*word == seq) { arrived = true; break; }
            if ((spins & 1023) == 1023 &&
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) break;      // (a kernel fault: the sync below reports it)
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (arrived) {
            memcpy(h, ctx->host_pin, sizeof(PanelState));
            if (flags) memcpy(flags, ctx->host_pin + 256, 4 * sizeof(unsigned));
            return ASB_OK;
        }
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(h, ctx->host_pin, sizeof(PanelState));
        if (flags) memcpy(flags, ctx->host_pin + 256, 4 * sizeof(unsigned));
        return ASB_OK;
    }
    ASB_HIP(ctx, hipMemcpyAsync(ctx->host_pin, st, sizeof(PanelState), hipMemcpyDeviceToHost, ctx->stream));
    if (flags) ASB_HIP(ctx, hipMemcpyAsync(ctx->host_pin + 256, ctx->coop_bar, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(h, ctx->host_pin, sizeof(PanelState));
    if (flags) memcpy(flags, ctx->host_pin + 256, 4 * sizeof(unsigned));
    return ASB_OK;
}
static int read_panel_state(asb_ctx* ctx, PanelState* h, unsigned* flags) { return read_panel_state_from(ctx, ctx->pstate, h, flags); }

// the same for up to 16 device words (8 bytes each): pinned slot [512, 640), sequence word at 648
__global__ __launch_bounds__(64) void k_publish_words(const unsigned long long* __restrict__ src, int n, unsigned char* __restrict__ pin,
                                                      unsigned long long seq) {
    if ((int)threadIdx.x < n) reinterpret_cast<unsigned long long*>(pin + 512)[threadIdx.x] = src[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(pin + 648), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static int fetch_words(asb_ctx* ctx, const void* dev, int n, void* host) {
    int rc = asb_pin_alloc(ctx);
    if (rc) return rc;
    if (n < 1 || n > 16) ASB_FAIL(ctx, ASB_ERR_ARG, "fetch_words: %d words", n);
    if (ctx->host_poll && ctx->host_pin_dev) {
        const unsigned long long seq = ++ctx->pin_seq;
        hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(64), 0, ctx->stream, (const unsigned long long*)dev, n, ctx->host_pin_dev, seq);
        ASB_CHECK_LAUNCH(ctx);
        volatile unsigned long long* word = reinterpret_cast<volatile unsigned long long*>(ctx->host_pin + 648);
        const auto t0 = std::chrono::steady_clock::now();
        bool arrived = false;
        for (unsigned spins = 0;; ++spins) {
            if (*word == seq) { arrived = true; break; }
            if ((spins & 1023) == 1023 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) break;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (!arrived) ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(host, ctx->host_pin + 512, (size_t)n * 8);
        return ASB_OK;
    }
    ASB_HIP(ctx, hipMemcpyAsync(ctx->host_pin + 512, dev, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(host, ctx->host_pin + 512, (size_t)n * 8);
    return ASB_OK;
}

// the summary of a k_panel_multi launch in ONE read: word sp = committed | (proven + 1) << 32 of sub-panel sp
// (proven + 1 = 0: that sub-panel did not finish), word 8 = the kernel's flags [1] | [2] << 1
__global__ __launch_bounds__(64) void k_publish_multi(const PanelState* __restrict__ sub, const unsigned* __restrict__ flags,
                                                      unsigned char* __restrict__ pin, unsigned long long seq) {
    const int l = threadIdx.x;
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(pin + 512);
    if (l < 8) dst[l] = (unsigned long long)sub[l].committed | ((unsigned long long)(sub[l].proven + 1) << 32);
    if (l == 8) dst[8] = (unsigned long long)(flags[1] ? 1u : 0u) | ((unsigned long long)(flags[2] ? 1u : 0u) << 1);
    __threadfence_system();
    __syncthreads();
    if (l == 0)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(pin + 648), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// begin: the publication is enqueued right behind the panel kernel (*seq = 0: no polled slot, `end` copies synchronously);
// end: wait for it.  Whatever the caller enqueues in between runs while the host waits.
static int fetch_multi_begin(asb_ctx* ctx, unsigned long long* seq) {
    int rc = asb_pin_alloc(ctx);
    if (rc) return rc;
    *seq = 0;
    if (ctx->host_poll && ctx->host_pin_dev) {
        *seq = ++ctx->pin_seq;
        hipLaunchKernelGGL(k_publish_multi, dim3(1), dim3(64), 0, ctx->stream, ctx->pstate2, ctx->coop_bar, ctx->host_pin_dev, *seq);
        ASB_CHECK_LAUNCH(ctx);
    }
    return ASB_OK;
}
static int fetch_multi_end(asb_ctx* ctx, unsigned long long seq, unsigned long long* out9) {
    if (seq) {
        volatile unsigned long long* word = reinterpret_cast<volatile unsigned long long*>(ctx->host_pin + 648);
        const auto t0 = std::chrono::steady_clock::now();
        bool arrived = false;
        for (unsigned spins = 0;; ++spins) {
            if (*word == seq) { arrived = true; break; }
            if ((spins & 1023) == 1023 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 4.0) break;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (!arrived) ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(out9, ctx->host_pin + 512, 9 * 8);
        return ASB_OK;
    }
    // (no polled slot: a synchronising copy -- it also waits for whatever was enqueued behind the panel kernel)
    PanelState h[8];
    unsigned fl[4];
    ASB_HIP(ctx, hipMemcpyAsync(h, ctx->pstate2, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipMemcpyAsync(fl, ctx->coop_bar, sizeof(fl), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int l = 0; l < 8; ++l) out9[l] = (unsigned long long)h[l].committed | ((unsigned long long)(h[l].proven + 1) << 32);
    out9[8] = (unsigned long long)(fl[1] ? 1u : 0u) | ((unsigned long long)(fl[2] ? 1u : 0u) << 1);
    return ASB_OK;
}

// second half of a pass with unproven steps: energies of the columns that stood (force_ncols < 0: the count the check
// left on the device; otherwise the host's value, e.g. the minimum over the ranks), column sums, *kept
static int project_commit(asb_ctx* ctx, long long k0, int force_ncols, int64_t* kept) {
    long long cw = (ctx->n_loc + 255) / 256;
    const int cgrid = (int)(cw < ctx->nblk_cap ? cw : ctx->nblk_cap);
    hipLaunchKernelGGL(k_commit_energy, dim3(cgrid), dim3(256), 0, ctx->stream, ctx->comps, (long long)(3 * ctx->n_loc),
                       (long long)ctx->n_loc, (int)k0, ctx->pstate, ctx->wn2t, ctx->energy, ctx->pmax, ctx->pidx, ctx->psum,
                       ctx->colpart, force_ncols);
    ASB_CHECK_LAUNCH(ctx);
    ctx->nblk = cgrid;
    hipLaunchKernelGGL(k_colsum, dim3(1), dim3(1024), 0, ctx->stream, ctx->colpart, ctx->nblk, force_ncols, k0, ctx->scal,
                       force_ncols < 0 ? ctx->pstate : (PanelState*)nullptr);
    ASB_CHECK_LAUNCH(ctx);
    if (kept) {
        *kept = force_ncols;
        if (force_ncols < 0) {
            PanelState h;
            int rc = read_panel_state(ctx, &h, nullptr);
            if (rc) return rc;
            *kept = h.committed;
        }
    }
    return ASB_OK;
}

// one projection pass for components [k0, k0+ncols); proven < ncols: the steps from `proven` on were taken without
// proof -- they are checked against every vertex's energy and *kept (optional) returns how many columns survived
// (check_only: the energies are left to a later project_commit)
static int project_pass(asb_ctx* ctx, long long k0, int ncols, int proven = ASB_PANEL_COLS, int64_t* kept = nullptr,
                        bool check_only = false) {
    const bool spec = proven < ncols;
    const int NC = (int)(ctx->Fp / 16);
    const int nwg = (int)((3 * ctx->n_loc + 47) / 48);
    int grid = ctx->n_cu * ((NC <= 16) ? 2 : 1);     // persistent blocks: 1 per CU (2 for the 256-thread variant)
    if (grid > nwg) grid = nwg;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_build_wt, dim3(64), dim3(256), 0, ctx->stream, ctx->W, ctx->scal, k0, ncols, (int)ctx->Fp,
                       ctx->Wt, ctx->wn2t);
    ASB_CHECK_LAUNCH(ctx);
    launch_project(ctx, ncols, ctx->comps + (size_t)k0 * 3 * ctx->n_loc);      // HIP-event bracketed per kernel launch
    ASB_CHECK_LAUNCH(ctx);
    // rows j < k0: earlier panels' weights against this panel's; rows k0 .. k0+ncols-1: the panel's own Gram matrix
    hipLaunchKernelGGL(k_panel_gram, dim3((unsigned)(k0 + ncols)), dim3(256), 0, ctx->stream, ctx->W, ctx->Wt, (int)ctx->Fp, ctx->gram,
                       ctx->gram_s, ctx->wn2t);
    const bool rows_kernel = ctx->correct_rows != 0;
    long long cw = rows_kernel ? (ctx->n_loc + 63) / 64 : (ctx->n_loc + 255) / 256;
    const int cgrid = (int)(cw < ctx->nblk_cap ? cw : ctx->nblk_cap);
    if (spec) {
        if (rows_kernel)
            hipLaunchKernelGGL(k_correct_rows<true>, dim3(cgrid), dim3(192), 0, ctx->stream, ctx->comps, (long long)(3 * ctx->n_loc),
                               (long long)ctx->n_loc, (int)k0, ncols, ctx->gram_s, ctx->wn2t, ctx->energy, ctx->pmax, ctx->pidx,
                               ctx->psum, ctx->colpart, ctx->pstate, ctx->scalar_dev, ctx->sel_e2);
        else
            hipLaunchKernelGGL(k_correct<true>, dim3(cgrid), dim3(256), 0, ctx->stream, ctx->comps, (long long)(3 * ctx->n_loc),
                               (long long)ctx->n_loc, (int)k0, ncols, ctx->gram, ctx->wn2t, ctx->energy, ctx->pmax, ctx->pidx,
                               ctx->psum, ctx->colpart, (const long long*)nullptr, (const PanelState*)nullptr, (long long)0,
                               ctx->pstate, ctx->scalar_dev, ctx->sel_e2);
        ASB_CHECK_LAUNCH(ctx);
        if (check_only) return ASB_OK;
        return project_commit(ctx, k0, -1, kept);
    }
    if (rows_kernel)
        hipLaunchKernelGGL(k_correct_rows<false>, dim3(cgrid), dim3(192), 0, ctx->stream, ctx->comps, (long long)(3 * ctx->n_loc),
                           (long long)ctx->n_loc, (int)k0, ncols, ctx->gram_s, ctx->wn2t, ctx->energy, ctx->pmax, ctx->pidx,
                           ctx->psum, ctx->colpart, (PanelState*)nullptr, (const double*)nullptr);
    else
        hipLaunchKernelGGL(k_correct<false>, dim3(cgrid), dim3(256), 0, ctx->stream, ctx->comps, (long long)(3 * ctx->n_loc),
                           (long long)ctx->n_loc, (int)k0, ncols, ctx->gram, ctx->wn2t, ctx->energy, ctx->pmax, ctx->pidx,
                           ctx->psum, ctx->colpart);
    ASB_CHECK_LAUNCH(ctx);
    ctx->nblk = cgrid;
    hipLaunchKernelGGL(k_colsum, dim3(1), dim3(1024), 0, ctx->stream, ctx->colpart, ctx->nblk, ncols, k0, ctx->scal,
                       (PanelState*)nullptr);
    ASB_CHECK_LAUNCH(ctx);
    if (kept) *kept = ncols;
    return ASB_OK;
}

// the multi-tile projection kernel (bracketed by the profiling events); tiles built by wide_build_tile.
// Up to 3 tiles: compiled for two waves per SIMD (<= 256 registers: 8 waves x 8 KB of X in flight per CU); from 4 tiles on
// the accumulators alone take 128+ registers: one wave per SIMD, the MFMA time per chunk (1024 cycles per tile) hides the
// load latency instead.  Variants measured on config 4 (tools/wide_variants.sh): deeper prefetch (PD = 2), four waves per
// tile, 32- and 48-row tiles, two chunks per group -- all slower than this one for 2 tiles.
template <int NT, int G, int S, int NCT, int OCC, int PD>
static int launch_l2w_cfg(asb_ctx* ctx, const WideArgs& wa, int blocks_per_cu) {
    const long long rows = 3 * ctx->n_loc, ntiles = (rows + 16 * NT - 1) / (16 * NT);
    const size_t lds = ((size_t)(S - 1) * NT * NCT * 4 * 64 + 2) * sizeof(double);
    static bool attr_set_dev[64] = {false};              // per device: a second GPU in the same process needs its own
    bool& attr_set = attr_set_dev[ctx->dev & 63];
    if (!attr_set) {
        ASB_HIP(ctx, hipFuncSetAttribute((const void*)k_project_l2w<NT, G, S, NCT, OCC, PD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const long long cap = (long long)blocks_per_cu * ctx->n_cu;
    hipLaunchKernelGGL((k_project_l2w<NT, G, S, NCT, OCC, PD>), dim3((unsigned)(ntiles < cap ? ntiles : cap)), dim3(64 * S), lds, ctx->stream,
                       ctx->X, rows, (int)ctx->Fp, ctx->Wq3, ctx->wn2t3, wa, (ctx->wide_out ? ctx->wide_out : ctx->comps), rows, ctx->tile_counter);
    return ASB_OK;
}
template <int NCT, int P, int SYNC, int NSB = (SYNC ? 3 : 2), int XD = 1>
static int launch_l2d(asb_ctx* ctx, const WideArgs& wa) {
    const long long rows = 3 * ctx->n_loc, ngroups = (rows + 15) / 16;
    const size_t stage = (size_t)NSB * 4 * P * NCT * 128, redd = (size_t)4 * 4 * 4 * 64;
    const size_t lds = ((stage > redd ? stage : redd) + 4) * sizeof(double);
    static bool attr_set_dev[64] = {false};
    bool& attr_set = attr_set_dev[ctx->dev & 63];
    if (!attr_set) {
        ASB_HIP(ctx, hipFuncSetAttribute((const void*)k_project_l2d<NCT, P, SYNC, NSB, 0, XD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    // one block per CU, each with its own contiguous share of the 16-row groups (no work queue)
    const long long nb = ngroups < ctx->n_cu ? ngroups : ctx->n_cu;
    hipLaunchKernelGGL((k_project_l2d<NCT, P, SYNC, NSB, 0, XD>), dim3((unsigned)nb), dim3(512), lds, ctx->stream, ctx->X, rows, (int)ctx->Fp, ctx->Wq3,
                       ctx->wn2t3, wa, (ctx->wide_out ? ctx->wide_out : ctx->comps), rows);
    return ASB_OK;
}
template <int DBG>
static int launch_l2e(asb_ctx* ctx, const WideArgs& wa) {
    const long long rows = 3 * ctx->n_loc, ngroups = (rows + 15) / 16;
    const size_t lds = (size_t)8 * 2 * 2 * 4 * 128 * sizeof(double);
    static bool attr_set_dev[64] = {false};
    bool& attr_set = attr_set_dev[ctx->dev & 63];
    if (!attr_set) {
        ASB_HIP(ctx, hipFuncSetAttribute((const void*)k_project_l2e<4, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const long long nb = ngroups < ctx->n_cu ? ngroups : ctx->n_cu;
    hipLaunchKernelGGL((k_project_l2e<4, DBG>), dim3((unsigned)nb), dim3(512), lds, ctx->stream, ctx->X, rows, (int)ctx->Fp, ctx->Wq3, ctx->wn2t3, wa,
                       (ctx->wide_out ? ctx->wide_out : ctx->comps), rows);
    return ASB_OK;
}
template <int NCT>
static int launch_l2w(asb_ctx* ctx, int variant, const WideArgs& wa) {
    if (variant == 60 && NCT == 4) return launch_l2e<0>(ctx, wa);                // no synchronisation between waves (private weight rings): slower
    if (variant == 45 && NCT == 4) return launch_l2d<4, 3, 0>(ctx, wa);      // balanced partition, barrier per stage (two buffers)
    if (variant == 47 && NCT == 4) return launch_l2d<4, 2, 1>(ctx, wa);      // shorter stages
    if (variant == 52 && NCT == 4) return launch_l2d<4, 3, 1, 3, 11>(ctx, wa);  // + weights of the next column tile read ahead (BPF)
    if (variant == 51 && NCT == 4) return launch_l2d<4, 3, 5, 3>(ctx, wa);   // arrival counters per frame-half group of four waves (no gain: 1.43)
    // (round 2's variants -- one wave per tile, 32- / 48- / 96- / 128-row tiles, deeper prefetch, weights shared through LDS with a
    // barrier per chunk pair (k_project_l2b) or per stage from a tile queue (k_project_l2c) -- were measured and removed; their
    // numbers are in DESIGN.md section 5 and profiles/r02*)
    // default: up to 3 sub-panels every wave fetches its own weights from L2 (two waves per SIMD); 4 sub-panels only fit two waves
    // per SIMD with the weights staged in LDS
    if (NCT == 4) return launch_l2d<4, 3, 1>(ctx, wa);
    return launch_l2w_cfg<4, 1, 2, NCT, (NCT <= 3 ? 2 : 1), 1>(ctx, wa, NCT <= 3 ? 4 : 2);
}
static int launch_wide(asb_ctx* ctx, int ntile, const WideArgs& wa) {
    const long long rows = 3 * ctx->n_loc;
    static const int variant = getenv("ASB_WIDE_VARIANT") ? atoi(getenv("ASB_WIDE_VARIANT")) : 4;
    size_t slot;
    int rc;
    if ((rc = prof_begin(ctx, slot))) return rc;
    if (ntile == 1) {                 // one tile: the single-panel kernel on the tile's operands
        const long long ntiles = (rows + 63) / 64;
        hipLaunchKernelGGL((k_project_l2s<4, 2, 2, 1>), dim3((unsigned)(ntiles < 4 * ctx->n_cu ? ntiles : 4 * ctx->n_cu)), dim3(128), 0, ctx->stream,
                           ctx->X, rows, (int)ctx->Fp, ctx->Wq3, ctx->wn2t3, wa.nc[0], (ctx->wide_out ? ctx->wide_out : ctx->comps) + (size_t)wa.kb[0] * rows, rows, ctx->tile_counter);
    } else {
        switch (ntile) {
            case 2: rc = launch_l2w<2>(ctx, variant, wa); break;
            case 3: rc = launch_l2w<3>(ctx, variant, wa); break;
            case 4: rc = launch_l2w<4>(ctx, variant, wa); break;
            case 5: rc = launch_l2w<5>(ctx, variant, wa); break;
            case 6: rc = launch_l2w<6>(ctx, variant, wa); break;
            case 7: rc = launch_l2w<7>(ctx, variant, wa); break;
            default: rc = launch_l2w<8>(ctx, variant, wa); break;
        }
        if (rc) return rc;
    }
    if ((rc = prof_end(ctx, slot))) return rc;
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

// timing probe of the multi-tile kernel on the context's tensor (results are garbage): mode 0 = the kernel as it runs,
// 1 = X operand from four cache-resident tiles (MFMA + L2 operand alone), 2 = no MFMAs (the loads alone)
template <int NCT, int MODE>
static int l2w_probe_launch(asb_ctx* ctx, const WideArgs& wa) {
    constexpr int OCC = NCT <= 3 ? 2 : 1;
    const long long rows = 3 * ctx->n_loc, ntiles = (rows + 63) / 64;
    const size_t lds = ((size_t)4 * NCT * 4 * 64 + 2) * sizeof(double);
    ASB_HIP(ctx, hipFuncSetAttribute((const void*)k_project_l2w<4, 1, 2, NCT, OCC, 1, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long long cap = (long long)(NCT <= 3 ? 4 : 2) * ctx->n_cu;
    hipLaunchKernelGGL((k_project_l2w<4, 1, 2, NCT, OCC, 1, MODE>), dim3((unsigned)(ntiles < cap ? ntiles : cap)), dim3(128), lds, ctx->stream,
                       ctx->X, rows, (int)ctx->Fp, ctx->Wq3, ctx->wn2t3, wa, ctx->comps, rows, ctx->tile_counter);
    return ASB_OK;
}
extern "C" int asb_test_l2w_probe(asb_ctx* ctx, int nct, int mode, int reps, double* ms_out) {
    if (!ctx || !ctx->X || !ctx->comps || !ms_out || reps < 1 || ctx->K < 16 * nct) return ASB_ERR_ARG;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->Wt3, (size_t)ASB_MAX_SUB * ctx->Fp * 16))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->Wq3, (size_t)ASB_MAX_SUB * ctx->Fp * 16))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->wn2t3, (size_t)16 * ASB_MAX_SUB))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->tile_counter, (size_t)16))) return rc;
    // weights: constants (mode < 100) or pseudo-random in (-0.05, 0.05) (mode >= 100: mode - 100 is the mode proper) -- the
    // f64 MFMA rate of this part depends on how many operand bits toggle
    const bool random_w = mode >= 100;
    if (random_w) mode -= 100;
    std::vector<double> ones((size_t)ASB_MAX_SUB * ctx->Fp * 16, 1.0e-3);
    if (random_w) {
        unsigned long long z = 0x9E3779B97F4A7C15ull;
        for (double& v : ones) {
            z = z * 6364136223846793005ull + 1442695040888963407ull;
            v = ((double)(z >> 11) / 9007199254740992.0 - 0.5) * 0.1;
        }
    }
    ASB_HIP(ctx, hipMemcpy(ctx->Wq3, ones.data(), ones.size() * sizeof(double), hipMemcpyHostToDevice));
    for (double& v : ones) v = 1.0e-3;
    ASB_HIP(ctx, hipMemcpy(ctx->wn2t3, ones.data(), 16 * ASB_MAX_SUB * sizeof(double), hipMemcpyHostToDevice));
    WideArgs wa{};
    for (int ct = 0; ct < nct; ++ct) { wa.kb[ct] = 16 * ct; wa.nc[ct] = 16; }
    hipEvent_t e0, e1;
    ASB_HIP(ctx, hipEventCreate(&e0));
    ASB_HIP(ctx, hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        ASB_HIP(ctx, hipMemsetAsync(ctx->tile_counter, 0, 16 * sizeof(unsigned), ctx->stream));
        ASB_HIP(ctx, hipEventRecord(e0, ctx->stream));
        rc = ASB_ERR_ARG;
#define ASB_PROBE_CASE(N)                                                           \
        if (nct == N && mode < 10) rc = mode == 0 ? l2w_probe_launch<N, 0>(ctx, wa) : (mode == 1 ? l2w_probe_launch<N, 1>(ctx, wa) : l2w_probe_launch<N, 2>(ctx, wa));
        ASB_PROBE_CASE(2) ASB_PROBE_CASE(3) ASB_PROBE_CASE(4)
#undef ASB_PROBE_CASE
        if (nct == 4 && mode >= 20) {          // k_project_l2d<4, 3>: 20 = as it runs, 23 = no X loads in the loop, 24 = nor LDS reads, 25 = no stage sync
            const long long rows = 3 * ctx->n_loc;
#define ASB_L2D_PROBE(M)                                                                                                        \
            {                                                                                                                   \
                const size_t lds3 = ((size_t)3 * 4 * 3 * 4 * 128 + 4) * sizeof(double);                                         \
                const long long ngr = (rows + 15) / 16;                                                                         \
                ASB_HIP(ctx, hipFuncSetAttribute((const void*)k_project_l2d<4, 3, 1, 3, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3)); \
                hipLaunchKernelGGL((k_project_l2d<4, 3, 1, 3, M>), dim3((unsigned)(ngr < ctx->n_cu ? ngr : ctx->n_cu)), dim3(512), lds3, ctx->stream, \
                                   ctx->X, rows, (int)ctx->Fp, ctx->Wq3, ctx->wn2t3, wa, ctx->comps, rows);                     \
            }
            if (mode == 23) ASB_L2D_PROBE(3) else if (mode == 24) ASB_L2D_PROBE(4) else if (mode == 25) ASB_L2D_PROBE(5) else ASB_L2D_PROBE(0)
#undef ASB_L2D_PROBE
            rc = ASB_OK;
        }
        if (rc) return rc;
        ASB_HIP(ctx, hipEventRecord(e1, ctx->stream));
        ASB_HIP(ctx, hipEventSynchronize(e1));
        float ms = 0.f;
        ASB_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_out = best;
    return ASB_OK;
}

static int launch_project_l2(asb_ctx* ctx, int ncols, double* out) {
    const long long rows = 3 * ctx->n_loc;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->Wq, (size_t)ctx->Fp * 16))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->tile_counter, (size_t)16))) return rc;
    hipLaunchKernelGGL(k_build_wq, dim3(64), dim3(256), 0, ctx->stream, ctx->Wt, (int)ctx->Fp, ctx->Wq, ctx->tile_counter);
    const int variant = ctx->l2_variant;       // 4 (default): two waves (one 128-thread block) per 64-row tile; 5: four; 6: <= 256 registers
    const long long ntiles = (rows + 63) / 64;
    size_t slot;
    if ((rc = prof_begin(ctx, slot))) return rc;
    if (variant == 6)
        hipLaunchKernelGGL((k_project_l2s<4, 2, 2, 1, 2>), dim3((unsigned)(ntiles < 4 * ctx->n_cu ? ntiles : 4 * ctx->n_cu)), dim3(128), 0, ctx->stream, ctx->X, rows, (int)ctx->Fp, ctx->Wq,
                           ctx->wn2t, ncols, out, rows, ctx->tile_counter);
    else if (variant == 5)      // four waves per tile: loses at the barriers what it gains at the end of the launch
        hipLaunchKernelGGL((k_project_l2s<4, 2, 4, 1>), dim3((unsigned)(ntiles < 2 * ctx->n_cu ? ntiles : 2 * ctx->n_cu)), dim3(256), 0, ctx->stream, ctx->X, rows, (int)ctx->Fp, ctx->Wq,
                           ctx->wn2t, ncols, out, rows, ctx->tile_counter);
    else
        hipLaunchKernelGGL((k_project_l2s<4, 2, 2, 1>), dim3((unsigned)(ntiles < 4 * ctx->n_cu ? ntiles : 4 * ctx->n_cu)), dim3(128), 0, ctx->stream, ctx->X, rows, (int)ctx->Fp, ctx->Wq,
                           ctx->wn2t, ncols, out, rows, ctx->tile_counter);
    if ((rc = prof_end(ctx, slot))) return rc;
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

static void launch_project(asb_ctx* ctx, int ncols, double* out) {
    (void)launch_project_l2(ctx, ncols, out);      // (errors surface through the launch check of the caller)
}

// out_rows (ncols, 3 n_loc) = X . Wfk[:, k0:k0+ncols] / col_scale[k0 + t]   (col_scale NULL: raw products;
// used for c = W^T X of SPLOCS and for the POD back-projection U = A V S^-1)
int asb_project_columns(asb_ctx* ctx, const double* Wfk, int64_t ldw, int64_t k0, int ncols, double* out_rows,
                        const double* col_scale) {
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->Wt, (size_t)ctx->Fp * ASB_PANEL_COLS))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->wn2t, (size_t)ASB_PANEL_COLS))) return rc;
    hipLaunchKernelGGL(k_build_wt_fk, dim3(64), dim3(256), 0, ctx->stream, Wfk, (long long)ldw, (long long)k0, ncols,
                       (int)ctx->F, (int)ctx->Fp, ctx->Wt, ctx->wn2t, col_scale);
    launch_project(ctx, ncols, out_rows);
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

__global__ void k_build_wq_tiles(const double* __restrict__ Wt3, int Fp, double* __restrict__ Wq3, unsigned* __restrict__ tile_counter);
// the same product for up to 64 columns in ONE pass over X (the four-tile kernels of the panel reads; SPLOCS' c = W^T X took four
// 16-column passes): out_rows[(k0 + j) - k0] for j < ncols, ncols <= 64
int asb_project_columns_wide(asb_ctx* ctx, const double* Wfk, int64_t ldw, int64_t k0, int ncols, double* out_rows, const double* col_scale) {
    if (ncols < 1 || ncols > 64) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_project_columns_wide: %d columns", ncols);
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->Wt3, (size_t)ASB_MAX_SUB * ctx->Fp * 16))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->Wq3, (size_t)ASB_MAX_SUB * ctx->Fp * 16))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->wn2t3, (size_t)16 * ASB_MAX_SUB))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->tile_counter, (size_t)16))) return rc;
    const int ntile = (ncols + 15) / 16;
    WideArgs wa{};
    for (int ct = 0; ct < ntile; ++ct) {
        wa.kb[ct] = 16 * ct;
        wa.nc[ct] = ncols - 16 * ct < 16 ? ncols - 16 * ct : 16;
        hipLaunchKernelGGL(k_build_wt_fk, dim3(64), dim3(256), 0, ctx->stream, Wfk, (long long)ldw, (long long)(k0 + 16 * ct), wa.nc[ct], (int)ctx->F,
                           (int)ctx->Fp, ctx->Wt3 + (size_t)ct * ctx->Fp * 16, ctx->wn2t3 + 16 * ct, col_scale);
    }
    hipLaunchKernelGGL(k_build_wq_tiles, dim3(64, ntile), dim3(256), 0, ctx->stream, ctx->Wt3, (int)ctx->Fp, ctx->Wq3, ctx->tile_counter);
    ctx->wide_out = out_rows;
    rc = launch_wide(ctx, ntile, wa);
    ctx->wide_out = nullptr;
    return rc;
}

// start of a run on a tensor whose initial energies are known (E0): scal <- 0, hist <- 0, energy <- E0, range scalars restored --
// one launch instead of two memsets, a device-to-device copy and a one-thread kernel (each an API call with its own host cost)
__global__ __launch_bounds__(256) void k_begin_reset(double* __restrict__ scal, long long n_scal, int* __restrict__ hist, int n_hist,
                                                     double* __restrict__ energy, const double* __restrict__ E0, long long n,
                                                     double* __restrict__ sc, const double* __restrict__ e0) {
    const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x, st = (long long)gridDim.x * 256;
    for (long long i = i0; i < n_scal; i += st) scal[i] = 0.0;
    for (long long i = i0; i < n_hist; i += st) hist[i] = 0;
    for (long long i = i0; i < n; i += st) energy[i] = E0[i];
    if (i0 == 0) {                       // (k_range_restore)
        sc[SC_EMAX] = e0[1];
        sc[SC_LO] = 0.0;
        sc[SC_HI] = e0[1];
        sc[SC_ABOVE] = 0.0;
        sc[SC_NORMX2] = e0[0];
        sc[SC_E0MAX] = e0[1];
    }
}

int asb_project_begin(asb_ctx* ctx, int64_t K) {
    const size_t rows = (size_t)ctx->n_loc * 3;
    int rc;
    int64_t mt = 768;      // ~16-step panels on flat (random) energy landscapes at the least candidate traffic (tools/sweep_m.sh)
    if (const char* e = getenv("ASB_M_TARGET")) mt = atoll(e) > 16 ? atoll(e) : 16;      // tuning knob (candidates per panel)
    ctx->m_target = ctx->N_glob < mt ? ctx->N_glob : mt;
    ctx->m_cap = ctx->N_glob < 2 * mt ? ctx->N_glob : 2 * mt;
    if ((rc = asb_alloc(ctx, &ctx->energy, (size_t)ctx->n_loc))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->W, (size_t)K * ctx->Fp))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->comps, (size_t)K * rows))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->scal, (size_t)(K + 1) * 4))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->xrec, (size_t)(2 + 3 * ctx->Fp)))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->s_dev, (size_t)ctx->n_loc))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->Wt, (size_t)ctx->Fp * ASB_PANEL_COLS))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->wn2t, (size_t)ASB_PANEL_COLS))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->candR, (size_t)ctx->m_cap * 3 * ctx->Fp))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->cand_e, (size_t)ctx->m_cap))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->cand_idx, (size_t)ctx->m_cap))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->cpmax, (size_t)ctx->nblk_cap))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->cpidx, (size_t)ctx->nblk_cap))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->cpsum, (size_t)ctx->nblk_cap))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->colpart, (size_t)ctx->nblk_cap * 16))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->hist, (size_t)ASB_NBINS))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->pstate, (size_t)1))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->gram, (size_t)K * ASB_PANEL_COLS))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->gram_s, (size_t)K * ASB_PANEL_COLS))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->ctmp, (size_t)ASB_CBLOCKS * ctx->m_cap))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->ccnt, (size_t)ASB_CBLOCKS))) return rc;
    StreamCfg c;
    if (!pick_cfg(ctx->Fp, c)) ASB_FAIL(ctx, ASB_ERR_LIMIT, "F too large");
    hipLaunchKernelGGL(k_sc_set, dim3(1), dim3(1), 0, ctx->stream, ctx->scalar_dev, (int)SC_TAU_DIV, 1.0e300);      // diversity family off
    ctx->diverse_next = ctx->read_diverse = false;
    ctx->n_diverse_reads = 0;
    // the adaptive panel lengths start afresh: the same tensor gives the same panels, hence the same bits, on every call
    ctx->sub_cur = 0;
    for (int q = 0; q < 8; ++q) ctx->sub_budget[q] = ASB_PANEL_COLS;
    if (ctx->e0_valid && ctx->E0 && ctx->e0_reuse) {
        // the energies of the prepared tensor came with the sweep that wrote it (asb_snapshots_scale) or with an earlier
        // begin on the same tensor: X has not changed since, so nothing is read again
        const long long nmax = ctx->n_loc > (K + 1) * 4 ? ctx->n_loc : (K + 1) * 4;
        hipLaunchKernelGGL(k_begin_reset, dim3((unsigned)((nmax + 255) / 256 < 2048 ? (nmax + 255) / 256 : 2048)), dim3(256), 0, ctx->stream,
                           ctx->scal, (long long)(K + 1) * 4, ctx->hist, (int)ASB_NBINS, ctx->energy, ctx->E0, (long long)ctx->n_loc,
                           ctx->scalar_dev, ctx->e0_sc);
        ASB_CHECK_LAUNCH(ctx);
        ctx->nblk = 0;                 // no partial records yet: every consumer of them runs after a refresh
        ctx->n_energy_pass = 0;
        return ASB_OK;
    }
    // initial energies straight from X (read-only pass)
    ASB_HIP(ctx, hipMemsetAsync(ctx->hist, 0, ASB_NBINS * sizeof(int), ctx->stream));
    ASB_HIP(ctx, hipMemsetAsync(ctx->scal, 0, (size_t)(K + 1) * 4 * sizeof(double), ctx->stream));
    const int grid = stream_grid(ctx, c, ctx->n_loc);
    StreamArgs a{ctx->X, nullptr, nullptr, nullptr, nullptr, ctx->energy, ctx->pmax, ctx->pidx, ctx->psum,
                 (long long)ctx->n_loc, nullptr};
    launch_stream(ctx, c, false, grid, a);
    ASB_CHECK_LAUNCH(ctx);
    ctx->nblk = grid;
    hipLaunchKernelGGL(k_range_init, dim3(1), dim3(256), 0, ctx->stream, ctx->pmax, ctx->pidx, ctx->psum, ctx->nblk,
                       ctx->scalar_dev, 1);
    ASB_CHECK_LAUNCH(ctx);
    ctx->n_energy_pass = 1;
    ctx->mean_frac = 0.0;          // EV comes with the standardisation sweep only
    ctx->ev_valid = false;
    if ((rc = asb_alloc(ctx, &ctx->E0, (size_t)ctx->n_loc))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->e0_sc, (size_t)4))) return rc;
    ASB_HIP(ctx, hipMemcpyAsync(ctx->E0, ctx->energy, (size_t)ctx->n_loc * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    ASB_HIP(ctx, hipMemcpyAsync(ctx->e0_sc, ctx->scalar_dev + SC_NORMX2, 2 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    ctx->e0_valid = true;
    return ASB_OK;
}

// --------------------------------------------------------------------------------------
// Panel inner loop, read-only form.  The candidate rows stay at their panel-start state R0; because
// the panel's weights are mutually orthogonal, R_s^(t) . w_t = R0_s . w_t, so one step only needs
//   k_cand_dots : c_t[s] = R0_s . w_t / |w_t|^2 for every candidate (kept in cand_c), energies
//                 e_s -= |w_t|^2 |c_t[s]|^2 (relative to the EXACT panel-start energies), arg-max partials
//   k_pick_panel: the winner's current slab rebuilt explicitly, R0_b - sum_{t'<t} c_t'[b] w_t', then the
//                 3x3 Gram / Jacobi / w_t exactly as k_pick does
// -- half the L2 traffic of updating every candidate row, and no dependent write/read of the rows.
// --------------------------------------------------------------------------------------
template <int T, int E2>
__global__ __launch_bounds__((T >= 256 ? T : 256)) void k_cand_dots(
    const double* __restrict__ R0, const double* __restrict__ wk, const double* __restrict__ scal_k, int t_panel,
    long long m_cap, double* __restrict__ cand_c, double* __restrict__ energy, double* __restrict__ pmax,
    long long* __restrict__ pidx, double* __restrict__ psum, int F2, const PanelState* __restrict__ panel) {
    if (panel->done) return;
    constexpr int BLOCK = (T >= 256 ? T : 256);
    constexpr int VPB = BLOCK / T;
    constexpr int NW = T / 64;
    const int tid = threadIdx.x, g = tid / T, t = tid % T, wig = t >> 6, lane = tid & 63;
    __shared__ double red[VPB][NW][3];
    __shared__ double lead_e[VPB];
    __shared__ long long lead_i[VPB];
    const long long n = panel->n_cand;
    double2 w[E2];
#pragma unroll
    for (int i = 0; i < E2; ++i) {
        const int j = t + i * T;
        w[i] = (j < F2) ? reinterpret_cast<const double2*>(wk)[j] : make_double2(0.0, 0.0);
    }
    const double wn2 = scal_k[1];
    double bmax = -1.0;
    long long bidx = 0x7fffffffffffffffLL;
    for (long long base = (long long)blockIdx.x * VPB; base < n; base += (long long)gridDim.x * VPB) {
        const long long s = base + g;
        const bool valid = s < n;
        const double2* row = reinterpret_cast<const double2*>(R0) + (valid ? s : 0) * 3 * (long long)F2;
        double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int i = 0; i < E2; ++i) {
                const int j = t + i * T;
                const double2 x = (valid && j < F2) ? row[(long long)d * F2 + j] : make_double2(0.0, 0.0);
                acc[d] += x.x * w[i].x + x.y * w[i].y;
            }
#pragma unroll
        for (int d = 0; d < 3; ++d) acc[d] = wave_sum(acc[d]);
        if (NW > 1) {
            __syncthreads();
            if (lane == 0) { red[g][wig][0] = acc[0]; red[g][wig][1] = acc[1]; red[g][wig][2] = acc[2]; }
            __syncthreads();
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                double sum = 0.0;
#pragma unroll
                for (int q = 0; q < NW; ++q) sum += red[g][q][d];
                acc[d] = sum;
            }
        }
        if (t == 0 && valid) {
            double* c = cand_c + ((long long)t_panel * m_cap + s) * 3;
            c[0] = acc[0] / wn2; c[1] = acc[1] / wn2; c[2] = acc[2] / wn2;
            // c . acc, not acc^2 / |w|^2: the squares of dot products of 1e-120-scaled snapshots flush to zero
            double e = energy[s] - (c[0] * acc[0] + c[1] * acc[1] + c[2] * acc[2]);
            if (e < 0.0) e = 0.0;
            energy[s] = e;
            if (am_better(e, s, bmax, bidx)) { bmax = e; bidx = s; }
        }
    }
    if (t == 0) { lead_e[g] = bmax; lead_i[g] = bidx; }
    __syncthreads();
    if (tid == 0) {
        double be = lead_e[0];
        long long bi = lead_i[0];
#pragma unroll
        for (int q = 1; q < VPB; ++q)
            if (am_better(lead_e[q], lead_i[q], be, bi)) { be = lead_e[q]; bi = lead_i[q]; }
        pmax[blockIdx.x] = be; pidx[blockIdx.x] = bi; psum[blockIdx.x] = 0.0;
    }
}

// one block of 1024 threads; thread tid owns frames f = tid + 1024 i (i < 2) of the three rows when
// Fp <= 2048, otherwise the explicit slab goes through a global scratch row
#define ASB_PP_T 1024
#define ASB_PP_E 2
__global__ __launch_bounds__(ASB_PP_T) void k_pick_panel(const double* __restrict__ R0, const double* __restrict__ cand_c,
                                                         long long m_cap, const double* pmax, const long long* pidx,
                                                         int nblk, int F, int Fp, double* __restrict__ W,
                                                         double* __restrict__ scal, long long k, long long k0,
                                                         PanelState* __restrict__ panel,
                                                         const long long* __restrict__ cand_idx,
                                                         double* __restrict__ scratch) {
    __shared__ double sh_d[ASB_PP_T];
    __shared__ long long sh_i[ASB_PP_T];
    __shared__ double u_sh[4];
    __shared__ double cprev[16 * 3];
    if (panel->done) return;
    const int tid = threadIdx.x;
    // arg-max over the block partials of the last candidate pass
    double be = -1.0;
    long long bi = 0x7fffffffffffffffLL;
    for (int b = tid; b < nblk; b += ASB_PP_T)
        if (am_better(pmax[b], pidx[b], be, bi)) { be = pmax[b]; bi = pidx[b]; }
    sh_d[tid] = be; sh_i[tid] = bi;
    __syncthreads();
    for (int o = ASB_PP_T / 2; o > 0; o >>= 1) {
        if (tid < o && am_better(sh_d[tid + o], sh_i[tid + o], sh_d[tid], sh_i[tid])) { sh_d[tid] = sh_d[tid + o]; sh_i[tid] = sh_i[tid + o]; }
        __syncthreads();
    }
    be = sh_d[0]; bi = sh_i[0];
    __syncthreads();
    if (!(be > panel->theta + panel->margin) || bi >= panel->n_cand || cand_idx[bi] < 0) {
        if (tid == 0) panel->done = 1;
        return;
    }
    const int tp = (int)(k - k0);
    if (tid < tp * 3) cprev[tid] = cand_c[((long long)(tid / 3) * m_cap + bi) * 3 + (tid % 3)];
    __syncthreads();
    const double* r0 = R0 + bi * 3 * (long long)Fp;
    const bool in_regs = Fp <= ASB_PP_T * ASB_PP_E;
    const int iters = in_regs ? ASB_PP_E : (Fp + ASB_PP_T - 1) / ASB_PP_T;
    double x[3][ASB_PP_E];
    double gsum[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll 2
    for (int i = 0; i < iters; ++i) {
        const int f = tid + ASB_PP_T * i;
        double a = 0.0, b = 0.0, c = 0.0;
        if (f < Fp) {
            a = r0[f]; b = r0[Fp + f]; c = r0[2 * Fp + f];
            for (int q = 0; q < tp; ++q) {
                const double wq = W[(k0 + q) * Fp + f];
                a -= cprev[3 * q] * wq; b -= cprev[3 * q + 1] * wq; c -= cprev[3 * q + 2] * wq;
            }
            if (!in_regs) { scratch[f] = a; scratch[Fp + f] = b; scratch[2 * Fp + f] = c; }
        }
        if (in_regs && i < ASB_PP_E) { x[0][i] = a; x[1][i] = b; x[2][i] = c; }
        if (f < F) {
            gsum[0] += a * a; gsum[1] += a * b; gsum[2] += a * c;
            gsum[3] += b * b; gsum[4] += b * c; gsum[5] += c * c;
        }
    }
    block_sum<6>(gsum, sh_d);
    if (tid == 0) {
        double lam, u0, u1, u2;
        eig3_top(gsum[0], gsum[1], gsum[2], gsum[3], gsum[4], gsum[5], lam, u0, u1, u2);
        u_sh[0] = u0; u_sh[1] = u1; u_sh[2] = u2; u_sh[3] = lam;
    }
    __syncthreads();
    const double u0 = u_sh[0], u1 = u_sh[1], u2 = u_sh[2];
    double* wk = W + k * (long long)Fp;
    double wn[1] = {0.0};
#pragma unroll 2
    for (int i = 0; i < iters; ++i) {
        const int f = tid + ASB_PP_T * i;
        if (f < Fp) {
            double wv = 0.0;
            if (f < F)
                wv = (in_regs && i < ASB_PP_E) ? (u0 * x[0][i] + u1 * x[1][i] + u2 * x[2][i])
                                                : (u0 * scratch[f] + u1 * scratch[Fp + f] + u2 * scratch[2 * Fp + f]);
            wk[f] = wv;
            wn[0] += wv * wv;
        }
    }
    block_sum<1>(wn, sh_d);
    if (tid == 0) {
        scal[k * 4 + 0] = sqrt(fmax(u_sh[3], 0.0));
        scal[k * 4 + 1] = wn[0];
        scal[k * 4 + 2] = __longlong_as_double(cand_idx[bi]);
        panel->committed = k - k0 + 1;
    }
}

template <int T, int E2>
static void launch_cand_dots_te(asb_ctx* ctx, int grid, long long k, int t_panel) {
    constexpr int BLOCK = (T >= 256 ? T : 256);
    hipLaunchKernelGGL((k_cand_dots<T, E2>), dim3(grid), dim3(BLOCK), 0, ctx->stream, ctx->candR, ctx->W + k * ctx->Fp,
                       ctx->scal + k * 4, t_panel, (long long)ctx->m_cap, ctx->cand_c, ctx->cand_e, ctx->cpmax, ctx->cpidx,
                       ctx->cpsum, (int)(ctx->Fp / 2), ctx->pstate);
}
template <int E2>
static void launch_cand_dots_e(asb_ctx* ctx, int T, int grid, long long k, int t_panel) {
    switch (T) {
        case 64: launch_cand_dots_te<64, E2>(ctx, grid, k, t_panel); break;
        case 128: launch_cand_dots_te<128, E2>(ctx, grid, k, t_panel); break;
        case 256: launch_cand_dots_te<256, E2>(ctx, grid, k, t_panel); break;
        case 512: launch_cand_dots_te<512, E2>(ctx, grid, k, t_panel); break;
        default: launch_cand_dots_te<1024, E2>(ctx, grid, k, t_panel); break;
    }
}
static void launch_cand_dots(asb_ctx* ctx, const StreamCfg& c, int grid, long long k, int t_panel) {
    switch (c.E2) {
        case 4: launch_cand_dots_e<4>(ctx, c.T, grid, k, t_panel); break;
        case 8: launch_cand_dots_e<8>(ctx, c.T, grid, k, t_panel); break;
        default: launch_cand_dots_e<16>(ctx, c.T, grid, k, t_panel); break;
    }
}

#define ASB_MARGIN_REL 1.0e-11

static int hist_grid(const asb_ctx* ctx) {
    const long long n = ctx->n_loc;
    return (int)((n + 255) / 256 < 512 ? (n + 255) / 256 : 512);
}

// level 1: exponent histogram (range-free); level 2: linear bins inside the crossing binade
// threshold with ~m_target (at most m_cap) of the n values of E above it into sc[SC_TAU] (the sketch replay's subset, asb_sketch.hip);
// uses the panel selection's histogram and range scalars, which are free between two reads
int asb_sketch_subset_tau(asb_ctx* ctx, const double* E, long long n, long long m_target, long long m_cap) {
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->hist, (size_t)ASB_NBINS))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->scalar_dev, (size_t)48))) return rc;      // (a context that never uploaded snapshots: the test entry)
    ASB_HIP(ctx, hipMemsetAsync(ctx->hist, 0, ASB_NBINS * sizeof(int), ctx->stream));      // (between two reads the bins are zero anyway)
    long long want = (n + 255) / 256;
    const int grid = (int)(want < 4LL * ctx->n_cu ? want : 4LL * ctx->n_cu);
    for (int level = 1; level <= 2; ++level) {
        hipLaunchKernelGGL(k_hist, dim3(grid), dim3(256), 0, ctx->stream, E, n, ctx->scalar_dev, ctx->hist, level == 1 ? 1 : 0);
        hipLaunchKernelGGL(k_tau, dim3(1), dim3(256), 0, ctx->stream, ctx->hist, ctx->scalar_dev, level, m_target, m_cap);
    }
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

extern "C" int asb_panel_hist(asb_ctx* ctx, int level, int* hist_dev) {
    if (!ctx || !ctx->energy || ctx->mode != ASB_DEFLATE_PROJECT) return ASB_ERR_ARG;
    int* h = hist_dev ? hist_dev : ctx->hist;
    // the context's own histogram is zero here: cleared when the mode starts and by every asb_panel_tau that consumed
    // it; a caller's buffer (all-reduced in between) is cleared explicitly
    if (hist_dev) ASB_HIP(ctx, hipMemsetAsync(h, 0, ASB_NBINS * sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_hist, dim3(hist_grid(ctx)), dim3(256), 0, ctx->stream, ctx->energy, (long long)ctx->n_loc,
                       ctx->scalar_dev, h, level == 1 ? 1 : 0);
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

extern "C" int asb_panel_tau(asb_ctx* ctx, int level, const int* hist_dev) {
    if (!ctx || !ctx->energy || ctx->mode != ASB_DEFLATE_PROJECT) return ASB_ERR_ARG;
    hipLaunchKernelGGL(k_tau, dim3(1), dim3(256), 0, ctx->stream, hist_dev ? const_cast<int*>(hist_dev) : ctx->hist, ctx->scalar_dev, level,
                       (long long)(ctx->m_target_eff ? ctx->m_target_eff : ctx->m_target), (long long)ctx->m_cap);
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

// Multi-rank thresholding in ONE exchange.  After the two local histogram steps (asb_panel_hist / asb_panel_tau with
// NULL buffers: no all-reduce) this writes the shard's energies above its LOCAL threshold into out_dev[0 .. cap)
// (unordered, the rest -1) and the local threshold -- a bound on everything not exported -- into out_dev[cap].
// The ranks all-gather these (cap + 1 doubles each); the global threshold is then
//   tau = max( (m_target + 1)-th largest exported energy,  max over ranks of the local thresholds ),
// which every rank installs with asb_panel_set_tau before asb_panel_select.
__global__ __launch_bounds__(256) void k_export_above(const double* __restrict__ energy, long long n, const double* __restrict__ sc,
                                                      long long cap, double* __restrict__ out, unsigned* __restrict__ counter) {
    const double tau = sc[SC_TAU];
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < n; v += (long long)gridDim.x * 256) {
        const double e = energy[v];
        if (e > tau) {
            const unsigned slot = atomicAdd(counter, 1u);
            if (slot < cap) out[slot] = e;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[cap] = tau;
}
__global__ __launch_bounds__(256) void k_fill(double* __restrict__ out, long long n, double v, unsigned* __restrict__ counter) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = v;
    if (blockIdx.x == 0 && threadIdx.x == 0) *counter = 0u;
}

extern "C" int asb_panel_top_energies(asb_ctx* ctx, double* out_dev, int64_t cap) {
    if (!ctx || !ctx->energy || ctx->mode != ASB_DEFLATE_PROJECT || !out_dev || cap < 1) return ASB_ERR_ARG;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->tile_counter, (size_t)16))) return rc;
    unsigned* counter = ctx->tile_counter + 15;
    hipLaunchKernelGGL(k_fill, dim3(8), dim3(256), 0, ctx->stream, out_dev, (long long)cap + 1, -1.0, counter);
    hipLaunchKernelGGL(k_export_above, dim3(hist_grid(ctx)), dim3(256), 0, ctx->stream, ctx->energy, (long long)ctx->n_loc,
                       ctx->scalar_dev, (long long)cap, out_dev, counter);
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

// installs the candidate threshold (a device scalar): asb_panel_select then takes energy > tau, asb_panel_run uses
// tau as the bound on every non-candidate
extern "C" int asb_panel_set_tau(asb_ctx* ctx, const double* tau_dev) {
    if (!ctx || !ctx->energy || ctx->mode != ASB_DEFLATE_PROJECT || !tau_dev) return ASB_ERR_ARG;
    ASB_HIP(ctx, hipMemcpyAsync(ctx->scalar_dev + SC_TAU, tau_dev, sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return ASB_OK;
}

// ordered compaction of this shard's candidates (E > tau; all vertices when global_all; only
// forced_gidx when >= 0) and their exact residual rows.  rows_out / idx_out: caller's device
// buffers of capacity m_cap (multi-rank staging) or NULL for the context's own candidate buffer.
extern "C" int asb_panel_select(asb_ctx* ctx, int64_t k, int64_t forced_gidx, int global_all, double* rows_out,
                                long long* idx_out, int64_t* n_local, int* overflow) {
    if (!ctx || !ctx->energy || ctx->mode != ASB_DEFLATE_PROJECT) return ASB_ERR_ARG;
    if (!pick_cfg(ctx->Fp, ctx->cfg)) ASB_FAIL(ctx, ASB_ERR_LIMIT, "F too large");
    const StreamCfg c = ctx->cfg;
    const long long n = ctx->n_loc;
    double* rows = rows_out ? rows_out : ctx->candR;
    long long* idx = idx_out ? idx_out : ctx->cand_idx;
    if (forced_gidx >= 0) {
        hipLaunchKernelGGL(k_force_single, dim3(1), dim3(1), 0, ctx->stream, (long long)forced_gidx, (long long)ctx->v0, n,
                           idx, ctx->pstate);
    } else {
        hipLaunchKernelGGL(k_compact_a, dim3(ASB_CBLOCKS), dim3(256), 0, ctx->stream, ctx->energy, n, (long long)ctx->v0,
                           ctx->scalar_dev, global_all, (long long)ctx->m_cap, ctx->ctmp, ctx->ccnt, -1, ctx->sel_e2);
        hipLaunchKernelGGL(k_compact_b, dim3(ASB_CBLOCKS), dim3(64), 0, ctx->stream, ctx->ctmp, ctx->ccnt, ASB_CBLOCKS,
                           (long long)ctx->m_cap, idx, ctx->pstate);
    }
    ASB_CHECK_LAUNCH(ctx);
    const int ggrid = stream_grid(ctx, c, ctx->m_cap);
    launch_gather(ctx, c, ggrid, idx, (long long)ctx->m_cap, ctx->pstate, (int)k, rows, ctx->cand_e, ctx->cpmax, ctx->cpidx,
                  ctx->cpsum);
    ASB_CHECK_LAUNCH(ctx);
    ctx->cnblk = ggrid;
    if (n_local || overflow) {
        PanelState h;
        ASB_HIP(ctx, hipMemcpyAsync(&h, ctx->pstate, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (n_local) *n_local = h.n_cand;
        if (overflow) *overflow = (int)h.pad;
    }
    return ASB_OK;
}

// multi-rank: builds the (replicated) global candidate buffer from the all-gathered, padded
// per-rank pieces: rows_g (world, maxcount, 3, Fp), idx_g (world, maxcount), counts (world).
struct RankCounts { long long c[16]; };
// one block per candidate slot: slot -> (rank, position in that rank's padded piece) -> copy the row and its vertex id
// (stride_rows / stride_idx: distance between two ranks' pieces in 8-byte words)
__global__ __launch_bounds__(256) void k_assemble(const double* __restrict__ rows_g, const long long* __restrict__ idx_g, RankCounts cnt,
                                                  int world, long long stride_rows, long long stride_idx, long long row_len,
                                                  double* __restrict__ candR, long long* __restrict__ cand_idx, long long total) {
    for (long long s = blockIdx.x; s < total; s += gridDim.x) {
        long long off = s;
        int r = 0;
        while (r < world - 1 && off >= cnt.c[r]) { off -= cnt.c[r]; ++r; }
        const double2* src = reinterpret_cast<const double2*>(rows_g + (long long)r * stride_rows + off * row_len);
        double2* dst = reinterpret_cast<double2*>(candR + s * row_len);
        for (long long j = threadIdx.x; j < row_len / 2; j += 256) dst[j] = src[j];
        if (threadIdx.x == 0) cand_idx[s] = idx_g[(long long)r * stride_idx + off];
    }
}

static int panel_assemble(asb_ctx* ctx, const double* rows_g, const long long* idx_g, const int64_t* counts, int world,
                          int64_t maxcount, long long stride_rows, long long stride_idx);

extern "C" int asb_panel_assemble(asb_ctx* ctx, const double* rows_g, const long long* idx_g, const int64_t* counts,
                                  int world, int64_t maxcount) {
    if (!ctx) return ASB_ERR_ARG;
    return panel_assemble(ctx, rows_g, idx_g, counts, world, maxcount, (long long)maxcount * 3 * ctx->Fp, (long long)maxcount);
}
// the same from ONE all-gathered buffer: every rank's piece is its maxcount rows (3*Fp doubles each) followed by its
// maxcount vertex ids (int64), i.e. maxcount * (3*Fp + 1) eight-byte words per rank
extern "C" int asb_panel_assemble_packed(asb_ctx* ctx, const double* packed_g, const int64_t* counts, int world, int64_t maxcount) {
    if (!ctx || !packed_g || maxcount < 0) return ASB_ERR_ARG;
    const long long rl = 3 * ctx->Fp, stride = (long long)maxcount * (rl + 1);
    return panel_assemble(ctx, packed_g, reinterpret_cast<const long long*>(packed_g + (long long)maxcount * rl), counts, world, maxcount,
                          stride, stride);
}

static int panel_assemble(asb_ctx* ctx, const double* rows_g, const long long* idx_g, const int64_t* counts, int world,
                          int64_t maxcount, long long stride_rows, long long stride_idx) {
    if (!ctx || !ctx->candR || !rows_g || !idx_g || !counts) return ASB_ERR_ARG;
    if (world < 1 || world > 16) ASB_FAIL(ctx, ASB_ERR_LIMIT, "asb_panel_assemble: at most 16 ranks (got %d)", world);
    RankCounts rc_;
    int64_t off = 0;
    for (int r = 0; r < 16; ++r) rc_.c[r] = 0;
    for (int r = 0; r < world; ++r) {
        const int64_t c = counts[r];
        if (c < 0 || c > maxcount) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_panel_assemble: bad count");
        if (off + c > ctx->m_cap) ASB_FAIL(ctx, ASB_ERR_LIMIT, "asb_panel_assemble: %lld candidates exceed the capacity %lld",
                                           (long long)(off + c), (long long)ctx->m_cap);
        rc_.c[r] = c;
        off += c;
    }
    if (off > 0) {
        hipLaunchKernelGGL(k_assemble, dim3((unsigned)(off < 2048 ? off : 2048)), dim3(256), 0, ctx->stream, rows_g, idx_g, rc_, world,
                           stride_rows, stride_idx, (long long)(3 * ctx->Fp), ctx->candR, ctx->cand_idx, (long long)off);
        ASB_CHECK_LAUNCH(ctx);
    }
    ctx->n_slots_host = off;
    return ASB_OK;
}

// --------------------------------------------------------------------------------------
// Global threshold of the multi-rank panel from the all-gathered exports (asb_panel_top_energies): exact radix select
// of the (m_target + 1)-th largest energy in LDS (8 passes over the 64-bit patterns, which order like the non-negative
// doubles they encode), tau = max(that, the ranks' local thresholds) -> SC_TAU, and the per-rank candidate counts.
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_global_tau(const double* __restrict__ tab, int world, long long cap, long long want,
                                                     double* __restrict__ sc, long long* __restrict__ counts) {
    extern __shared__ unsigned long long keys[];          // world * cap
    __shared__ unsigned hist[256];
    __shared__ long long wsum[4];
    __shared__ unsigned long long prefix_sh;
    __shared__ long long remaining_sh;
    __shared__ double tau_sh;
    const int tid = threadIdx.x;
    const long long n = (long long)world * cap;
    for (long long i = tid; i < n; i += 1024) {
        const double e = tab[(i / cap) * (cap + 1) + (i % cap)];
        keys[i] = (e > 0.0) ? (unsigned long long)__double_as_longlong(e) : 0ull;
    }
    if (tid == 0) { prefix_sh = 0ull; remaining_sh = want; }
    __syncthreads();
    for (int byte = 7; byte >= 0; --byte) {
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned long long prefix = prefix_sh;
        for (long long i = tid; i < n; i += 1024) {
            const unsigned long long kx = keys[i];
            if (byte == 7 || (kx >> (8 * (byte + 1))) == prefix) atomicAdd(&hist[(kx >> (8 * byte)) & 255ull], 1u);
        }
        __syncthreads();
        // the bin where the count from the top reaches what is still wanted: a scan over the 256 bins by 256 threads
        // (thread i <-> bin 255 - i) instead of one thread walking them
        {
            const long long rem = remaining_sh;
            long long v = 0, incl = 0;
            if (tid < 256) {
                v = hist[255 - tid];
                incl = v;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const long long up = __shfl_up(incl, o, 64);
                    if ((tid & 63) >= o) incl += up;
                }
                if ((tid & 63) == 63) wsum[tid >> 6] = incl;
            }
            __syncthreads();
            if (tid < 256) {
                for (int w = 0; w < (tid >> 6); ++w) incl += wsum[w];
                const long long excl = incl - v;
                const int b = 255 - tid;
                // bins 255 .. 1: the first whose running count reaches rem; bin 0 takes what is left when none does
                if ((b > 0 && excl < rem && incl >= rem) || (b == 0 && excl < rem)) {
                    remaining_sh = rem - excl;
                    prefix_sh = (prefix << 8) | (unsigned long long)b;
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        double tau = (n >= want) ? __longlong_as_double((long long)prefix_sh) : 0.0;
        for (int r = 0; r < world; ++r) tau = fmax(tau, tab[(long long)r * (cap + 1) + cap]);
        tau_sh = tau;
        sc[SC_TAU] = tau;
    }
    __syncthreads();
    const double tau = tau_sh;
    const int wv = tid >> 6, lane = tid & 63;
    for (int r = wv; r < world; r += 16) {
        long long c = 0;
        for (long long i = lane; i < cap; i += 64) c += tab[(long long)r * (cap + 1) + i] > tau ? 1 : 0;
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if (lane == 0) counts[r] = c;
    }
}

// tab_dev: (world, cap + 1) all-gathered exports.  Installs tau on the device and returns the per-rank candidate counts.
extern "C" int asb_panel_global_tau(asb_ctx* ctx, const double* tab_dev, int world, int64_t cap, int64_t* counts_out) {
    if (!ctx || !ctx->energy || ctx->mode != ASB_DEFLATE_PROJECT || !tab_dev || !counts_out || world < 1 || cap < 1) return ASB_ERR_ARG;
    const size_t lds = (size_t)world * cap * sizeof(unsigned long long);
    if (world > 16 || lds > 150 * 1024) return ASB_ERR_LIMIT;        // the caller falls back to its own selection
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->bam_idx, (size_t)256))) return rc;
    if (lds > 48 * 1024)
        ASB_HIP(ctx, hipFuncSetAttribute((const void*)k_global_tau, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_global_tau, dim3(1), dim3(1024), lds, ctx->stream, tab_dev, world, (long long)cap,
                       (long long)(ctx->m_target_eff ? ctx->m_target_eff : ctx->m_target) + 1, ctx->scalar_dev, ctx->bam_idx);
    ASB_CHECK_LAUNCH(ctx);
    long long h[16];
    if ((rc = fetch_words(ctx, ctx->bam_idx, world, h))) return rc;
    for (int r = 0; r < world; ++r) counts_out[r] = h[r];
    return ASB_OK;
}

__device__ __forceinline__ void coop_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double coop_load(const double* p) {
    return __hip_atomic_load(const_cast<double*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// --------------------------------------------------------------------------------------
// k_panel_multi (round 3): ALL sub-panels of a read of X in ONE launch (up to ASB_MAX_SUB x 16 greedy steps), and a
// shorter step.  Rows live in registers as in k_panel_coop; what changed is the exchange:
//   * every word that crosses blocks is SELF-VALIDATING: it is reset to a bit pattern no payload can take (a NaN for
//     doubles, -1 for slots) two steps before it is written again, so a reader spins on the payload itself -- one memory
//     round trip per exchange instead of "poll the sequence number, then load the payload";
//   * a block publishes only (best energy, slot) -- 16 bytes -- not its tentative weights: the 3 x 3 eigen-pair of the
//     block's best candidate is computed WHILE the records travel (that wave does not poll), and only the block that
//     wins writes its w (16 KB instead of 256 x 16 KB of write-through traffic per step), straight from LDS, which the
//     other blocks read with the same spin-on-payload loads;
//   * every wave keeps the 3 x 3 Gram matrix of its own row current (six sums behind the deflation it does anyway), so
//     the energy is its trace and the winner's eigen-problem needs no further reduction.
// Rings of three (records: per block; weights: one buffer per step mod 3).  Reset protocol: a block that has passed the
// poll of step g's records knows that every block has finished step g - 1 (a block publishes step g only behind it), so
// it resets its own record of step g - 1, and block 0 the weight buffer of step g - 1; every wave drains its stores
// (s_waitcnt) in front of the barrier that precedes the next publication, so a reset is visible before anything that
// could lead another block to write or read that word again.  Hand-off form as in k_panel_coop: agent-scope (write-
// through) stores, agent-scope loads, no cache maintenance.
// Per sub-panel the kernel leaves what k_panel_coop left in a PanelState (sub[sp]: committed, proven, e_win, done); a
// sub-panel starts only behind one that ran its 16 steps.  Blocks without candidates leave at once.
// --------------------------------------------------------------------------------------
struct MultiArgs { long long kb[ASB_MAX_SUB]; int steps[ASB_MAX_SUB]; int spec_max[ASB_MAX_SUB]; int nsub; };
#define ASB_SENT_D 0xFFFFFFFFFFFFFFFFull          // quiet NaN with every payload bit set: never an energy or a weight
#define ASB_SENT_I (-1LL)                         // never a slot

__device__ __forceinline__ unsigned long long coop_load_u64(const unsigned long long* p) {
    return __hip_atomic_load(const_cast<unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void coop_store_u64(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void k_panel_arm_multi(PanelState* __restrict__ panel, PanelState* __restrict__ sub,
                                                         const double* __restrict__ sc, double margin_rel, unsigned* __restrict__ coop_flags,
                                                         unsigned long long* __restrict__ words, int n_words, MultiArgs ma,
                                                         long long n_slots = -1) {
    for (int r = threadIdx.x; r < n_words; r += blockDim.x) words[r] = ASB_SENT_D;     // (ASB_SENT_I has the same bits)
    if (threadIdx.x < 4) coop_flags[threadIdx.x] = 0u;
    if (threadIdx.x != 0) return;
    if (n_slots >= 0) panel->n_cand = n_slots;          // an ASSEMBLED candidate buffer (several ranks): the host knows its size
    panel->pad &= 1;
    panel->theta = panel->pad ? 1.0e300 : sc[SC_TAU];
    panel->margin = margin_rel * sc[SC_E0MAX];
    panel->done = 0;
    panel->committed = 0;
    panel->proven = -1;
    panel->spec_max = ma.spec_max[0];
    panel->spec_ok = ASB_PANEL_COLS;
    for (int sp = 0; sp < ASB_MAX_SUB; ++sp) {
        sub[sp] = *panel;
        sub[sp].spec_max = sp < ma.nsub ? ma.spec_max[sp] : 0;
    }
}

template <int NJ>
__global__ __launch_bounds__(256, (NJ > 16 ? 1 : 2)) void k_panel_multi(const double* __restrict__ R0, int F, int Fp,
                                                        double* __restrict__ W, double* __restrict__ scal, MultiArgs ma,
                                                        const PanelState* panel, PanelState* sub,
                                                        const long long* __restrict__ cand_idx, unsigned* __restrict__ bar,
                                                        unsigned long long* rec, unsigned long long* wbuf, unsigned long long* swbuf,
                                                        double* rows_out, int test_stall, int spec_rank) {
    constexpr int NQ = NJ / 4;                                   // weight words per thread
    __shared__ __attribute__((aligned(16))) double w_sh[NJ * 64 + 8];      // [NJ * 64] = |w|^2, [NJ * 64 + 1] = lambda
    __shared__ double wv_e[4];
    __shared__ long long wv_i[4];
    __shared__ double sh_e[4];
    __shared__ int sh_i[4];
    __shared__ int sh_c[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int G = gridDim.x;
    const long long n_cand = panel->n_cand;
    if (n_cand > 4LL * G) {          // more candidates than resident waves: the caller falls back to the two-kernel loop
        if (blockIdx.x == 0 && tid == 0) bar[2] = 1u;
        return;
    }
    const int Gact = n_cand > 0 ? (int)((n_cand + 3) / 4) : 1;  // blocks that hold candidates
    if ((int)blockIdx.x >= Gact) return;
    const int WROW = Fp + 8;                                     // a weight buffer: Fp words, |w|^2, lambda
    const double thr = panel->theta + panel->margin, e_floor = panel->margin;
    const bool overflow = panel->pad != 0;
    const long long s = (long long)blockIdx.x * 4 + wv;          // this wave's candidate slot
    const bool have = s < n_cand && cand_idx[s] >= 0;
    double x[3][NJ];
    {
        const double* row = R0 + s * 3 * (long long)Fp;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int f = lane + 64 * j;
                x[d][j] = (have && f < Fp) ? row[(long long)d * Fp + f] : 0.0;
            }
    }
    if (spec_rank > 0) {
        // this block's three speculative buffers start out as "not written": nobody is directed to them before a record of
        // THIS launch says so, and the block's first record goes out behind the drain of these stores (top of step 0)
#pragma unroll
        for (int r3 = 0; r3 < 3; ++r3) {
            unsigned long long* sown = swbuf + ((size_t)r3 * G + blockIdx.x) * WROW;
            for (int f = tid; f < WROW; f += 256) coop_store_u64(sown + f, ASB_SENT_D);
        }
    }
    double g[6] = {0, 0, 0, 0, 0, 0};                            // Gram matrix of the wave's row (00 01 02 11 12 22), every lane
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const double a = x[0][j], b = x[1][j], c = x[2][j];
        g[0] += a * a; g[1] += a * b; g[2] += a * c; g[3] += b * b; g[4] += b * c; g[5] += c * c;
    }
    wave_sum_dpp<6>(g);
    unsigned long long* tlog = reinterpret_cast<unsigned long long*>(bar + 4);      // [64][6] timestamps of block 0 (debug)
    const long long spin_max = test_stall ? (1LL << 10) : (1LL << 20);
    int gstep = 0;
    bool aborted = false, ended = false;
    // SPECULATIVE publication of w: a block whose best candidate ranked among the first `spec_rank` of the LAST step's records
    // stores its w (to its own buffer, straight from the wave that computed it) while the records still travel; if it wins,
    // the others find w already in memory when they know the winner -- the second exchange overlaps the first.  The record
    // says whether w was published that way (bit 32 of the slot word); a winner that did not publishes behind the poll as before.
    bool spec_now = false, spec_prev = false;
    int sp = 0;
    for (; sp < ma.nsub && !ended; ++sp) {
        const long long k0 = ma.kb[sp];
        const int steps = ma.steps[sp], spec_max = overflow ? 0 : ma.spec_max[sp];
        PanelState* st = sub + sp;
        int proven = -1, nrun = 0;
        for (int t = 0; t < steps; ++t, ++gstep) {
            const int ring = gstep % 3, ring_prev = (gstep + 2) % 3;
            if (blockIdx.x == 0 && tid == 0) tlog[gstep * 6 + 0] = wall_clock64();
            // ---- 1. block-local best; its (energy, slot) goes out at once
            const double e = (g[0] + g[3]) + g[5];
            if (lane == 0) { wv_e[wv] = have ? e : -1.0; wv_i[wv] = have ? s : 0x7fffffffffffffffLL; }
            __builtin_amdgcn_s_waitcnt(0);              // this wave's stores of the last step (resets, weights) are acknowledged
            __syncthreads();
            int ow = 0;
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (am_better(wv_e[q], wv_i[q], wv_e[ow], wv_i[ow])) ow = q;
            unsigned long long* myrec = rec + ((size_t)ring * G + blockIdx.x) * 2;
            if (tid == 0 && !(test_stall && (int)blockIdx.x == Gact - 1)) {      // (tests: the last block never signals)
                const unsigned long long slot32 = wv_i[ow] > 0x7ffffffeLL ? 0x7fffffffull : (unsigned long long)wv_i[ow];
                coop_store(reinterpret_cast<double*>(myrec), wv_e[ow]);
                coop_store_u64(myrec + 1, slot32 | (spec_now && wv_e[ow] > 0.0 ? (1ull << 32) : 0ull));
            }
            if (blockIdx.x == 0 && tid == 0) tlog[gstep * 6 + 1] = wall_clock64();
            // ---- 2. the best wave solves its 3 x 3 eigen-problem while the other three poll the records
            double lam = 0.0, u0 = 0.0, u1 = 0.0, u2 = 0.0;
            double be = -2.0;                            // (records carry -1 at the least)
            int bi = 0x7fffffff, dead = 0, better = 0;   // slots fit 30 bits (4 G <= 4096); better: records that outrank this block's
            if (wv == ow) {
                // the eigen-pair AND w = u^T row of the block's best candidate, into LDS, while the records travel: should
                // the block win, its w only has to be stored (w_sh is free here: the last reads of it -- the deflation of
                // step gstep - 1 -- lie in front of this step's first barrier)
                if (wv_e[ow] > 0.0) {
                    eig3_top_fast(g[0], g[1], g[2], g[3], g[4], g[5], lam, u0, u1, u2);     // every lane: identical
                    double wn1[1] = {0.0};
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const double wvv = u0 * x[0][j] + u1 * x[1][j] + u2 * x[2][j];      // (frames >= F are zero in the rows)
                        w_sh[lane + 64 * j] = wvv;
                        wn1[0] += wvv * wvv;
                    }
                    wave_sum_dpp<1>(wn1);
                    if (lane == 0) { w_sh[NJ * 64] = wn1[0]; w_sh[NJ * 64 + 1] = lam; }
                    if (spec_now) {
                        // 16-byte write-through stores of frame pairs (the wave's own LDS writes are in order for it)
                        double* sw = reinterpret_cast<double*>(swbuf) + ((size_t)ring * G + blockIdx.x) * WROW;
#pragma unroll
                        for (int q = 0; q < NJ / 2; ++q) {
                            const int f2 = 2 * lane + 128 * q;
                            if (f2 < Fp) {
                                typedef double d2v __attribute__((ext_vector_type(2)));
                                const d2v v = *reinterpret_cast<const d2v*>(&w_sh[f2]);
                                asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(sw + f2), "v"(v) : "memory");
                            }
                        }
                        if (lane == 0) {
                            typedef double d2v __attribute__((ext_vector_type(2)));
                            const d2v hd2 = {wn1[0], lam};
                            asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(sw + Fp), "v"(hd2) : "memory");
                        }
                    }
                }
            } else {
                const unsigned long long* recs = rec + (size_t)ring * G * 2;
                const int p = (wv - (wv > ow ? 1 : 0)) * 64 + lane;
                for (int b = p; b < Gact; b += 192) {
                    unsigned long long eb, ib;
                    long long spins = 0;
                    for (;;) {
                        eb = coop_load_u64(recs + 2 * b);
                        ib = coop_load_u64(recs + 2 * b + 1);
                        if (eb != ASB_SENT_D && ib != (unsigned long long)ASB_SENT_I) break;
                        // (the abort flag and the limit every 64 rounds: the fast path is the two loads alone)
                        if ((++spins & 63) == 0 &&
                            (spins > spin_max || __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { dead = 1; break; }
                    }
                    if (dead) break;
                    const double ebd = __longlong_as_double((long long)eb);
                    // key = 2 slot + (1 - published): the lowest slot wins a tie, the flag rides below it
                    const int sl = (int)(ib & 0x7fffffffull);
                    const int ibs = sl == 0x7fffffff ? 0x7fffffff : 2 * sl + ((ib >> 32) & 1ull ? 0 : 1);
                    if (ebd > be || (ebd == be && ibs < bi)) { be = ebd; bi = ibs; }
                    if (ebd > wv_e[ow] || (ebd == wv_e[ow] && (long long)sl < wv_i[ow])) ++better;
                }
                // arg-max over the wave: largest energy, then lowest slot among the lanes that hold it (NumPy's first max)
                const double em = wave_max_dpp(be);
                bi = wave_min_dpp(be == em ? bi : 0x7fffffff);
                be = em;
                better = wave_isum_dpp(better);
            }
            if (lane == 0) { sh_e[wv] = be; sh_i[wv] = bi; sh_c[wv] = better; }
            if (__syncthreads_or(dead)) { aborted = true; break; }
            be = sh_e[0]; bi = sh_i[0];
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (sh_e[q] > be || (sh_e[q] == be && sh_i[q] < bi)) { be = sh_e[q]; bi = sh_i[q]; }
            const bool win_spec = bi != 0x7fffffff && (bi & 1) == 0;       // the winner's w is (being) published already
            if (bi != 0x7fffffff) bi >>= 1;              // the winner's slot
            const int bb = bi >> 2;                      // its block (slot = 4 block + wave)
            const bool spec_mine = spec_now && wv_e[ow] > 0.0;             // this block published at this step
            // who publishes ahead at the NEXT step: the blocks whose record ranked among the first spec_rank now
            spec_now = spec_rank > 0 && wv_e[ow] > 0.0 && (sh_c[0] + sh_c[1] + sh_c[2] + sh_c[3]) < spec_rank;
            if (blockIdx.x == 0 && tid == 0) tlog[gstep * 6 + 2] = wall_clock64();
            // every block is past step gstep - 1: its words can be reset (see the header)
            if (tid == 0) {
                unsigned long long* old = rec + ((size_t)ring_prev * G + blockIdx.x) * 2;
                coop_store_u64(old, ASB_SENT_D);
                coop_store_u64(old + 1, (unsigned long long)ASB_SENT_I);
            }
            if (spec_prev) {                             // this block's own speculative buffer of step gstep - 1
                unsigned long long* sold = swbuf + ((size_t)ring_prev * G + blockIdx.x) * WROW;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int f = tid + 256 * q;
                    if (f < Fp) coop_store_u64(sold + f, ASB_SENT_D);
                }
                if (tid < 2) coop_store_u64(sold + Fp + tid, ASB_SENT_D);
            }
            spec_prev = spec_mine;
            {   // the weight buffer of step gstep - 1: every block resets its slice (same argument block by block: a block's
                // reset is visible before its next record, and whoever writes or reads the buffer again has seen ALL records)
                unsigned long long* wold = wbuf + (size_t)ring_prev * WROW;
                const int per = (WROW + Gact - 1) / Gact;
                const int f1 = ((int)blockIdx.x + 1) * per < WROW ? ((int)blockIdx.x + 1) * per : WROW;
                for (int f = (int)blockIdx.x * per + tid; f < f1; f += 256) coop_store_u64(wold + f, ASB_SENT_D);   // (few blocks: long slices)
            }
            if (!(be > thr) || bi >= n_cand) {           // cannot be proven to be the global arg-max ...
                if (proven < 0) proven = t;
                // ... the panel ends here, unless it may go on unproven (checked after the projection pass)
                if (bi >= n_cand || !(be > e_floor) || t - proven >= spec_max) {
                    if (blockIdx.x == 0 && tid == 0) st->done = 1;
                    ended = true;
                    break;
                }
            }
            nrun = t + 1;
            // ---- 3. the winner's w: its block writes it (LDS -> write-through stores by all four waves), the others spin on it
            unsigned long long* wb = win_spec ? swbuf + ((size_t)ring * G + bb) * WROW : wbuf + (size_t)ring * WROW;
            dead = 0;
            if (bb == (int)blockIdx.x && win_spec) {
                // (w_sh holds this block's w since the barrier behind the poll, and its best wave has stored it already)
            } else if (bb == (int)blockIdx.x) {      // (w_sh holds this block's w since the barrier behind the poll)
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int f = tid + 256 * q;
                    if (f < Fp) coop_store(reinterpret_cast<double*>(wb) + f, w_sh[f]);
                }
                if (tid < 2) coop_store(reinterpret_cast<double*>(wb) + Fp + tid, w_sh[NJ * 64 + tid]);
            } else {
                unsigned long long tmp[NQ], hd = 0ull;
                long long spins = 0;
                for (;;) {
                    bool all = true;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int f = tid + 256 * q;
                        tmp[q] = (f < Fp) ? coop_load_u64(wb + f) : 0ull;
                        all = all && tmp[q] != ASB_SENT_D;
                    }
                    if (tid < 2) { hd = coop_load_u64(wb + Fp + tid); all = all && hd != ASB_SENT_D; }
                    if (all) break;
                    if ((++spins & 63) == 0 &&
                        (spins > spin_max || __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { dead = 1; break; }
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q) w_sh[tid + 256 * q] = __longlong_as_double((long long)tmp[q]);
                if (tid < 2) w_sh[NJ * 64 + tid] = __longlong_as_double((long long)hd);
            }
            if (__syncthreads_or(dead)) { aborted = true; break; }
            if (blockIdx.x == 0 && tid == 0) tlog[gstep * 6 + 3] = wall_clock64();
            const double wn2 = w_sh[NJ * 64];
            if (bb == (int)blockIdx.x) {                 // the winner's block has w first and nothing to wait for: it writes the results
                const long long k = k0 + t;
                for (int f = tid; f < Fp; f += 256) W[k * (long long)Fp + f] = w_sh[f];
                if (tid == 0) {
                    scal[k * 4 + 0] = sqrt(fmax(w_sh[NJ * 64 + 1], 0.0));
                    scal[k * 4 + 1] = wn2;
                    scal[k * 4 + 2] = __longlong_as_double(cand_idx[bi]);
                    st->e_win[t] = be;
                }
            }
            // (words written once, by whichever block won; the running count is block 0's alone: one writer per address --
            // the XCDs' L2s are not coherent with each other for plain stores)
            if (blockIdx.x == 0 && tid == 0) st->committed = t + 1;
            // ---- 4. explicit deflation of this wave's row, and its new Gram matrix
            if (have) {
                // (w is read from LDS twice, for the dots and for the update: holding it in registers beside the row -- 64 + 192
                // VGPRs at F = 2000 -- pushes the loop into scratch memory, 9 us per step instead of 3)
                double acc[3] = {0.0, 0.0, 0.0};
                constexpr int JB = NJ < 8 ? NJ : 8;              // w in groups of JB values: the scheduler must not gather all NJ reads first
#pragma unroll
                for (int jb = 0; jb < NJ; jb += JB) {
                    double wj[JB];
#pragma unroll
                    for (int u = 0; u < JB; ++u) wj[u] = w_sh[lane + 64 * (jb + u)];
#pragma unroll
                    for (int u = 0; u < JB; ++u) {
                        acc[0] += x[0][jb + u] * wj[u]; acc[1] += x[1][jb + u] * wj[u]; acc[2] += x[2][jb + u] * wj[u];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                wave_sum_dpp<3>(acc);
#pragma unroll
                for (int d = 0; d < 3; ++d) acc[d] /= wn2;
#pragma unroll
                for (int q = 0; q < 6; ++q) g[q] = 0.0;
#pragma unroll
                for (int jb = 0; jb < NJ; jb += JB) {
                    double wj[JB];
#pragma unroll
                    for (int u = 0; u < JB; ++u) wj[u] = w_sh[lane + 64 * (jb + u)];
#pragma unroll
                    for (int u = 0; u < JB; ++u) {
                        const int j = jb + u;
                        const double a = x[0][j] - acc[0] * wj[u], b = x[1][j] - acc[1] * wj[u], c = x[2][j] - acc[2] * wj[u];
                        x[0][j] = a; x[1][j] = b; x[2][j] = c;
                        g[0] += a * a; g[1] += a * b; g[2] += a * c; g[3] += b * b; g[4] += b * c; g[5] += c * c;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                wave_sum_dpp<6>(g);
            }
            if (blockIdx.x == 0 && tid == 0) tlog[gstep * 6 + 4] = wall_clock64();
        }
        if (aborted) break;
        if (blockIdx.x == 0 && tid == 0) st->proven = proven < 0 ? nrun : proven;
        if (nrun < ASB_PANEL_COLS) ended = true;        // a later sub-panel runs only behind a full one
    }
    if (aborted) {
        if (tid == 0) __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // sub-panels that were not reached: "ran, committed nothing" (proven = -1 would read as a launch that did not finish)
    if (blockIdx.x == 0 && tid == 0)
        for (int q = sp; q < ma.nsub; ++q) { sub[q].proven = 0; sub[q].done = 1; }
    // launches that continue on the same candidates (one sub-panel per launch: the multi-rank driver, ASB_SUB_CHAIN=0):
    // the deflated rows become the next start rows -- behind the last step, past every abort exit
    if (rows_out != nullptr && have) {
        double* row = rows_out + s * 3 * (long long)Fp;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int f = lane + 64 * j;
                if (f < Fp) row[(long long)d * Fp + f] = x[d][j];
            }
    }
}

// the kernel's flags + debug timestamps, and its exchange words: records (3 rings x blocks x 2) then weight buffers (3 x (Fp + 8))
static int coop_buffers(asb_ctx* ctx, int* cgrid_all, size_t* n_words) {
    int rc;
    *cgrid_all = (int)((ctx->m_cap + 3) / 4);
    // records (3 rings x blocks x 2), the winner's weight buffers (3 x (Fp + 8)), the speculative ones (3 x blocks x (Fp + 8))
    *n_words = (size_t)3 * *cgrid_all * 2 + (size_t)3 * (ctx->Fp + 8) + (size_t)3 * *cgrid_all * (ctx->Fp + 8);
    if ((rc = asb_alloc(ctx, &ctx->coop_bar, (size_t)4 + 2 * 64 * 6))) return rc;
    return asb_alloc(ctx, &ctx->coop_rec, *n_words);
}
static int launch_panel_multi_any(asb_ctx* ctx, int grid, const MultiArgs& ma, bool* launched, PanelState* sub, bool writeback);

template <int NJ>
static int launch_panel_multi(asb_ctx* ctx, int grid, const MultiArgs& ma, bool* launched, PanelState* sub, bool writeback) {
    int per_cu = 0;
    ASB_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_panel_multi<NJ>, 256, 0));
    if (per_cu < 1) { *launched = false; return ASB_OK; }
    if ((long long)per_cu * ctx->n_cu < grid) grid = per_cu * ctx->n_cu;      // the kernel refuses panels with more candidates
    unsigned long long* words = (unsigned long long*)ctx->coop_rec;
    if (ctx->coop_launch == 1) {
        // A COOPERATIVE launch (ASB_COOP_LAUNCH=1) makes the runtime assert what the plain launch infers from the occupancy
        // query -- that all blocks are resident at once -- and refuse instead of running when they cannot be.  Not the default:
        // the cooperative path serialises against every other queue of the process; where the stream (or the device)
        // does not take it the launch falls back to the plain form, whose poll limit turns a surprise into an error, not a hang.
        const double* a0 = ctx->candR;
        int a1 = (int)ctx->F, a2 = (int)ctx->Fp;
        double *a3 = ctx->W, *a4 = ctx->scal;
        MultiArgs a5 = ma;
        const PanelState* a6 = ctx->pstate;
        PanelState* a7 = sub;
        const long long* a8 = ctx->cand_idx;
        unsigned* a9 = ctx->coop_bar;
        unsigned long long *a10 = words, *a11 = words + (size_t)3 * grid * 2, *a11b = a11 + (size_t)3 * (ctx->Fp + 8);
        double* a12 = writeback ? ctx->candR : (double*)nullptr;
        int a13 = ctx->coop_test_stall, a14 = ctx->spec_w_rank;
        void* args[] = {&a0, &a1, &a2, &a3, &a4, &a5, &a6, &a7, &a8, &a9, &a10, &a11, &a11b, &a12, &a13, &a14};
        const hipError_t e = hipLaunchCooperativeKernel((const void*)k_panel_multi<NJ>, dim3(grid), dim3(256), args, 0, ctx->stream);
        if (e == hipSuccess) {
            *launched = true;
            return ASB_OK;
        }
        (void)hipGetLastError();
        ctx->coop_launch = -1;                // this context's stream / device does not take cooperative launches
    }
    hipLaunchKernelGGL(k_panel_multi<NJ>, dim3(grid), dim3(256), 0, ctx->stream, ctx->candR, (int)ctx->F, (int)ctx->Fp, ctx->W, ctx->scal,
                       ma, ctx->pstate, sub, ctx->cand_idx, ctx->coop_bar, words, words + (size_t)3 * grid * 2,
                       words + (size_t)3 * grid * 2 + (size_t)3 * (ctx->Fp + 8), writeback ? ctx->candR : (double*)nullptr,
                       ctx->coop_test_stall, ctx->spec_w_rank);
    ASB_CHECK_LAUNCH(ctx);
    *launched = true;
    return ASB_OK;
}

static int launch_panel_multi_any(asb_ctx* ctx, int grid, const MultiArgs& ma, bool* launched, PanelState* sub, bool writeback) {
    if (ctx->Fp <= 256) return launch_panel_multi<4>(ctx, grid, ma, launched, sub, writeback);
    if (ctx->Fp <= 512) return launch_panel_multi<8>(ctx, grid, ma, launched, sub, writeback);
    if (ctx->Fp <= 1024) return launch_panel_multi<16>(ctx, grid, ma, launched, sub, writeback);
    return launch_panel_multi<32>(ctx, grid, ma, launched, sub, writeback);
}
static void print_multi_timeline(asb_ctx* ctx, int nsteps) {
    unsigned long long tl[64 * 6];
    (void)hipMemcpy(tl, ctx->coop_bar + 4, sizeof(tl), hipMemcpyDeviceToHost);
    for (int t = 0; t < 64 && t < nsteps; t += (nsteps > 16 ? 5 : 1))
        fprintf(stderr, "[asb]   step %2d: best+publish %.2f | poll+eigen %.2f | winner w %.2f | deflate+Gram %.2f | total %.2f us\n", t,
                (tl[t * 6 + 1] - tl[t * 6 + 0]) * 0.01, (tl[t * 6 + 2] - tl[t * 6 + 1]) * 0.01, (tl[t * 6 + 3] - tl[t * 6 + 2]) * 0.01,
                (tl[t * 6 + 4] - tl[t * 6 + 3]) * 0.01, (tl[t * 6 + 4] - tl[t * 6 + 0]) * 0.01);
}

// up to `steps` greedy steps on the context's candidate buffer (asb_panel_select with NULL
// buffers, or asb_panel_assemble); returns the number of components committed.
extern "C" int asb_panel_run(asb_ctx* ctx, int64_t k0, int steps, int global_all, int assembled, int64_t* committed) {
    if (!ctx || !ctx->candR || ctx->mode != ASB_DEFLATE_PROJECT || !committed) return ASB_ERR_ARG;
    if (steps < 1 || steps > ASB_PANEL_COLS || k0 < 0 || k0 + steps > ctx->K)
        ASB_FAIL(ctx, ASB_ERR_ARG, "asb_panel_run: bad range k0=%lld steps=%d", (long long)k0, steps);
    const StreamCfg c = ctx->cfg;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->cand_c, (size_t)ASB_PANEL_COLS * ctx->m_cap * 3))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->slab_scratch, (size_t)3 * ctx->Fp))) return rc;
    int cgrid_all = 0;
    size_t n_words = 0;
    const bool want_coop = ctx->panel_coop && ctx->Fp <= 2048;
    if (want_coop && (rc = coop_buffers(ctx, &cgrid_all, &n_words))) return rc;
    const long long spec_max = want_coop && !global_all ? ctx->run_spec_max : 0;
    hipLaunchKernelGGL(k_panel_arm, dim3(1), dim3(256), 0, ctx->stream, ctx->pstate, ctx->scalar_dev, global_all,
                       (long long)(assembled ? ctx->n_slots_host : -1), ASB_MARGIN_REL, want_coop ? ctx->coop_bar : (unsigned*)nullptr,
                       want_coop ? (unsigned long long*)ctx->coop_rec : (unsigned long long*)nullptr,
                       want_coop ? 3 * cgrid_all * 2 + 3 * ((int)ctx->Fp + 8) : 0,
                       ctx->run_theta_band, spec_max);
    const int grid = stream_grid(ctx, c, ctx->m_cap);
    bool coop = false;
    if (want_coop) {      // the whole inner loop in one launch of co-resident blocks, rows in registers (one sub-panel)
        MultiArgs ma{};
        ma.kb[0] = k0;
        ma.steps[0] = steps;
        ma.spec_max[0] = (int)spec_max;
        ma.nsub = 1;
        if ((rc = launch_panel_multi_any(ctx, cgrid_all, ma, &coop, ctx->pstate, ctx->run_writeback != 0))) return rc;
    }
    if (assembled && !coop) {      // two-kernel loop: energies / partial records of the assembled buffer (rows came from other ranks)
        StreamArgs a{ctx->candR, nullptr, nullptr, nullptr, nullptr, ctx->cand_e, ctx->cpmax, ctx->cpidx, ctx->cpsum,
                     (long long)ctx->m_cap, ctx->pstate};
        launch_stream(ctx, c, false, grid, a);
        ctx->cnblk = grid;
    }
    for (int t = 0; t < (coop ? 0 : steps); ++t) {
        const long long k = k0 + t;
        hipLaunchKernelGGL(k_pick_panel, dim3(1), dim3(ASB_PP_T), 0, ctx->stream, ctx->candR, ctx->cand_c, (long long)ctx->m_cap,
                           ctx->cpmax, ctx->cpidx, ctx->cnblk, (int)ctx->F, (int)ctx->Fp, ctx->W, ctx->scal, k,
                           (long long)k0, ctx->pstate, ctx->cand_idx, ctx->slab_scratch);
        if (t + 1 < steps) {          // the last step's dots would only feed a pick that never runs
            launch_cand_dots(ctx, c, grid, k, t);
            ctx->cnblk = grid;
        }
    }
    ASB_CHECK_LAUNCH(ctx);
    ctx->run_coop_used = coop ? 1 : 0;
    PanelState h;
    unsigned flags[4] = {0, 0, 0, 0};
    if ((rc = read_panel_state(ctx, &h, coop ? flags : nullptr))) return rc;
    *committed = h.committed;
    ctx->run_proven = (h.proven < 0 || h.proven > h.committed) ? h.committed : h.proven;
    ctx->n_panels++;
    if (coop) {
        if (flags[1]) {
            // The record exchange did not complete: the kernel's blocks were not all resident at once -- another stream,
            // an RCCL kernel or a second context holds part of the GPU.  Every block has left through the abort flag (the
            // stream is idle again); what the dead launch wrote (W / scal rows from k0 on, the panel state) is rewritten by
            // the two-kernel loop, which needs no co-residency.  The context stays on that loop from now on.
            // (Double / super panels let the kernel write the deflated rows back for the next sub-panel: it does so only
            // behind its last step, past every abort exit, so a launch that timed out has left the rows as it found them.)
            ctx->panel_coop = 0;
            ctx->coop_test_stall = 0;
            ctx->n_coop_fallbacks++;
            if (assembled) {
                // several ranks run this panel on identical data and must take identical decisions: the redo is the
                // DRIVER's, on every rank together (committed = -1 tells it; see _panels.py)
                ctx->n_panels--;
                *committed = -1;
                ctx->run_proven = 0;
                return ASB_OK;
            }
            ctx->n_panels--;
            rc = asb_panel_run(ctx, k0, steps, global_all, assembled, committed);
            ctx->run_coop_used = 0;
            return rc;
        }
        if (flags[2]) {               // more candidates than resident waves: this panel runs through the two-kernel loop
            const int save = ctx->panel_coop;
            ctx->panel_coop = 0;
            ctx->n_panels--;
            rc = asb_panel_run(ctx, k0, steps, global_all, assembled, committed);
            ctx->panel_coop = save;
            ctx->run_coop_used = 0;
            return rc;
        }
    }
    if (coop && getenv("ASB_DEBUG_PANELS")) print_multi_timeline(ctx, (int)h.committed);
    if (getenv("ASB_DEBUG_PANELS")) {
        double sc[8];
        (void)hipMemcpy(sc, ctx->scalar_dev, sizeof(sc), hipMemcpyDeviceToHost);
        std::vector<double> ce((size_t)h.n_cand);
        (void)hipMemcpy(ce.data(), ctx->cand_e, ce.size() * sizeof(double), hipMemcpyDeviceToHost);
        std::sort(ce.begin(), ce.end());
        const size_t m = ce.size();
        fprintf(stderr, "[asb] panel at k=%lld: n_cand=%lld committed=%lld theta=%.3f | candidate energies after the run: "
                        "max %.3f, 2nd %.3f, median %.3f, min %.3f\n", (long long)k0, h.n_cand, h.committed, h.theta,
                m ? ce[m - 1] : 0.0, m > 1 ? ce[m - 2] : 0.0, m ? ce[m / 2] : 0.0, m ? ce[0] : 0.0);
    }
    return ASB_OK;
}

// ONE pass over X: c_k for all vertices of the shard, energies, partial records
extern "C" int asb_panel_project(asb_ctx* ctx, int64_t k0, int ncols) {
    if (!ctx || !ctx->energy || ctx->mode != ASB_DEFLATE_PROJECT) return ASB_ERR_ARG;
    if (ncols < 1 || ncols > ASB_PANEL_COLS || k0 < 0 || k0 + ncols > ctx->K)
        ASB_FAIL(ctx, ASB_ERR_ARG, "asb_panel_project: bad range");
    int rc = project_pass(ctx, k0, ncols);
    if (rc) return rc;
    ctx->k_done = k0 + ncols;
    return ASB_OK;
}

// ---- the same with unproven steps, split for several ranks: every rank checks the steps against ITS vertices, the host
// takes the minimum over the ranks and hands it to asb_panel_commit
extern "C" int asb_panel_run_spec(asb_ctx* ctx, int64_t k0, int steps, int global_all, int assembled, int spec_max, int64_t* ran,
                                  int64_t* proven) {
    if (!ctx || !ran || !proven || spec_max < 0) return ASB_ERR_ARG;
    ctx->run_spec_max = ctx->spec_panels ? spec_max : 0;
    const int rc = asb_panel_run(ctx, k0, steps, global_all, assembled, ran);
    ctx->run_spec_max = 0;
    if (rc) return rc;
    *proven = ctx->run_proven;
    return ASB_OK;
}
// pass over X for all `ncols` steps of the panel (the first `proven` of them certain), coefficients written, energies
// NOT yet updated; *first_rejected = first unproven step one of this shard's vertices contradicts (ncols: none)
extern "C" int asb_panel_project_spec(asb_ctx* ctx, int64_t k0, int ncols, int proven, int64_t* first_rejected) {
    if (!ctx || !ctx->energy || ctx->mode != ASB_DEFLATE_PROJECT || !first_rejected) return ASB_ERR_ARG;
    if (ncols < 1 || ncols > ASB_PANEL_COLS || k0 < 0 || k0 + ncols > ctx->K || proven < 0 || proven >= ncols)
        ASB_FAIL(ctx, ASB_ERR_ARG, "asb_panel_project_spec: bad range");
    int rc = project_pass(ctx, k0, ncols, proven, nullptr, true);
    if (rc) return rc;
    ctx->n_spec_steps += ncols - proven;
    ctx->run_proven = proven;
    PanelState h;
    if ((rc = read_panel_state(ctx, &h, nullptr))) return rc;
    *first_rejected = h.spec_ok < ncols ? h.spec_ok : ncols;
    return ASB_OK;
}
// the same without a host read-back: the count goes into the caller's DEVICE word (a float64, to be min-all-reduced over
// the ranks and read once) -- one host synchronisation per panel less in the multi-rank driver
__global__ void k_spec_count(const PanelState* __restrict__ st, int ncols, double* __restrict__ out) {
    out[0] = (double)(st->spec_ok < ncols ? st->spec_ok : ncols);
}
extern "C" int asb_panel_project_spec_dev(asb_ctx* ctx, int64_t k0, int ncols, int proven, double* first_rejected_dev) {
    if (!ctx || !ctx->energy || ctx->mode != ASB_DEFLATE_PROJECT || !first_rejected_dev) return ASB_ERR_ARG;
    if (ncols < 1 || ncols > ASB_PANEL_COLS || k0 < 0 || k0 + ncols > ctx->K || proven < 0 || proven >= ncols)
        ASB_FAIL(ctx, ASB_ERR_ARG, "asb_panel_project_spec_dev: bad range");
    int rc = project_pass(ctx, k0, ncols, proven, nullptr, true);
    if (rc) return rc;
    ctx->n_spec_steps += ncols - proven;
    ctx->run_proven = proven;
    hipLaunchKernelGGL(k_spec_count, dim3(1), dim3(1), 0, ctx->stream, ctx->pstate, ncols, first_rejected_dev);
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}
// energies / column sums of the first `kept` columns of that pass (0: nothing stood, the energies stay as they were)
extern "C" int asb_panel_commit(asb_ctx* ctx, int64_t k0, int kept) {
    if (!ctx || !ctx->energy || ctx->mode != ASB_DEFLATE_PROJECT) return ASB_ERR_ARG;
    if (kept < 0 || kept > ASB_PANEL_COLS || k0 < 0 || k0 + kept > ctx->K) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_panel_commit: bad range");
    int rc = project_commit(ctx, k0, kept, nullptr);
    if (rc) return rc;
    if (kept > 0) ctx->k_done = k0 + kept;
    if (kept > ctx->run_proven) ctx->n_spec_kept += kept - ctx->run_proven;
    return ASB_OK;
}

// fallback: exact energies of ALL vertices of the shard (X_v - sum_j c_j w_j recomputed);
// best_energy / best_gidx (optional, synchronises): the shard's first arg-max afterwards.
extern "C" int asb_panel_refresh(asb_ctx* ctx, int64_t k, double* best_energy, int64_t* best_gidx) {
    if (!ctx || !ctx->energy || ctx->mode != ASB_DEFLATE_PROJECT) return ASB_ERR_ARG;
    if (!pick_cfg(ctx->Fp, ctx->cfg)) ASB_FAIL(ctx, ASB_ERR_LIMIT, "F too large");
    const int ggrid = stream_grid(ctx, ctx->cfg, ctx->n_loc);
    launch_gather(ctx, ctx->cfg, ggrid, nullptr, (long long)ctx->n_loc, nullptr, (int)k, nullptr, ctx->energy, ctx->pmax,
                  ctx->pidx, ctx->psum);
    ASB_CHECK_LAUNCH(ctx);
    ctx->nblk = ggrid;
    ctx->n_refresh++;
    if (best_energy || best_gidx) {
        hipLaunchKernelGGL(k_best_energy, dim3(1), dim3(256), 0, ctx->stream, ctx->pmax, ctx->pidx, ctx->psum, ctx->nblk,
                           ctx->scalar_dev + 8);
        double h[3];
        ASB_HIP(ctx, hipMemcpyAsync(h, ctx->scalar_dev + 8, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (best_energy) *best_energy = h[0];
        if (best_gidx) {
            long long li;
            memcpy(&li, &h[1], 8);
            *best_gidx = ctx->v0 + li;
        }
    }
    return ASB_OK;
}

// |X|^2 and the initial maximum energy of this shard; set_e0max >= 0 installs the GLOBAL maximum
// (histogram range and rounding margin must be identical on every rank).
extern "C" int asb_panel_scale(asb_ctx* ctx, double* normX2_local, double* e0max_local, double set_e0max) {
    if (!ctx || !ctx->energy || ctx->mode != ASB_DEFLATE_PROJECT) return ASB_ERR_ARG;
    if (normX2_local || e0max_local) {
        double sc[8];
        ASB_HIP(ctx, hipMemcpyAsync(sc, ctx->scalar_dev, sizeof(sc), hipMemcpyDeviceToHost, ctx->stream));
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (normX2_local) *normX2_local = sc[SC_NORMX2];
        if (e0max_local) *e0max_local = sc[SC_E0MAX];
    }
    if (set_e0max >= 0.0)
        ASB_HIP(ctx, hipMemcpyAsync(ctx->scalar_dev + SC_E0MAX, &set_e0max, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    return ASB_OK;
}

// one device word (the min-all-reduced count of a panel) to the host without a stream synchronisation: published into
// pinned memory behind whatever the stream still has to do (the collective included) and polled
extern "C" int asb_fetch_double(asb_ctx* ctx, const double* dev, double* out) {
    if (!ctx || !dev || !out) return ASB_ERR_ARG;
    return fetch_words(ctx, dev, 1, out);
}
extern "C" int asb_fetch_doubles(asb_ctx* ctx, const double* dev, int n, double* out) {
    if (!ctx || !dev || !out || n < 1 || n > 16) return ASB_ERR_ARG;
    return fetch_words(ctx, dev, n, out);
}
// multi-rank driver: switch the co-resident panel kernel on / off for this context (all ranks together); returns the old value
extern "C" int asb_panel_set_coop(asb_ctx* ctx, int on) {
    if (!ctx) return ASB_ERR_ARG;
    const int old = ctx->panel_coop;
    ctx->panel_coop = on ? 1 : 0;
    return old;
}
extern "C" int64_t asb_panel_capacity(const asb_ctx* ctx) { return ctx ? ctx->m_cap : 0; }
extern "C" int64_t asb_panel_target(const asb_ctx* ctx) { return ctx ? (ctx->m_target_eff ? ctx->m_target_eff : ctx->m_target) : 0; }

// ---- double panels (ASB_DOUBLE_PANELS=1, experimental): TWO sub-panels of up to 16 steps on the same candidate rows (the
// panel kernel writes the deflated rows back), then ONE read of X for their up to 32 columns.  The second sub-panel's
// steps are unproven almost by construction; they are checked like any unproven step, tile by tile: the first tile's
// check and energy update, and only if all of it stands the second tile's against the updated energies.
// ---- tiles of a read finished WITHOUT a host read in between: the check kernel also leaves the energies as if every column
// stood (Etmp) and the plain records / column sums; k_tile_decide (one block) looks at the check's verdict and at the chain --
// a tile counts only if every tile before it stood in full --, writes the column sums and res[ct] = columns kept (-1: not
// reached); k_apply_tmp adopts Etmp if the tile stood in full.  The host reads res[] once per read of X; a tile that did not
// stand in full (rare) is then committed the slow way (k_commit_energy), its energies being untouched.
__global__ __launch_bounds__(1024) void k_tile_decide(const double* __restrict__ colpart, int nblk, int ct, long long kb,
                                                      double* __restrict__ scal, PanelState* __restrict__ st, long long* __restrict__ res) {
    __shared__ int full_sh;
    if (threadIdx.x == 0) {
        const bool chain = ct == 0 || res[ASB_MAX_SUB] != 0;            // res[ASB_MAX_SUB]: every tile so far stood in full
        const int ran = (int)st->committed, ok = (int)(st->spec_ok < st->committed ? st->spec_ok : st->committed);
        const int full = chain && ok == ran;
        res[ct] = chain ? ok : -1;
        res[ASB_MAX_SUB] = full;
        st->committed = chain ? ok : 0;
        full_sh = full ? ran : 0;
    }
    __syncthreads();
    const int ncols = full_sh;
    const int t = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (t >= ncols) return;
    double v = 0.0;
    for (int b = lane; b < nblk; b += 64) v += colpart[(long long)b * 16 + t];
    v = wave_sum(v);
    if (lane == 0) scal[(kb + t) * 4 + 3] = v;
}
__global__ __launch_bounds__(256) void k_apply_tmp(double* __restrict__ E, const double* __restrict__ Etmp, long long n,
                                                   const long long* __restrict__ res) {
    if (res[ASB_MAX_SUB] == 0) return;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < n; v += (long long)gridDim.x * 256) E[v] = Etmp[v];
}

// ---- the checks of ALL tiles of a read in ONE launch (round 4; k_correct_rows<true> x ntile -> k_check_tiles): with weights
// orthogonalised before the pass a check is a streaming read of the tile's columns, and what a later tile sees of an earlier one
// is only the energies it would have left -- so one thread per vertex walks the tiles in order, carrying the energy AS IF every
// tile so far stood (the chain discards everything behind a tile that did not), and leaves per tile: the tentative energies,
// the block records and column sums, and the first rejected step (atomicMin into the tile's PanelState).  The arithmetic per
// tile is k_correct_rows<true>'s, term by term.  k_tiles_decide (one block) is k_tile_decide for all tiles in order;
// k_apply_tiles adopts the energies and records of the LAST tile that stood in full.  Four tiles: 3 launches instead of 12.
#define ASB_CHK_TILES 4
struct CheckOut { double* etmp; double* pmax; long long* pidx; double* psum; double* colpart; };
__global__ __launch_bounds__(256, 2) void k_check_tiles(const double* __restrict__ comps, long long comp_stride, long long n_vert, WideArgs wa,
                                                     int ntile, const double* __restrict__ wn2t3, const double* __restrict__ E,
                                                     const double* __restrict__ Ecl, const double* __restrict__ E2,
                                                     const double* __restrict__ sc, PanelState* __restrict__ st, CheckOut out) {
    __shared__ double sh[4 * 16];
    __shared__ double sh_m[4];
    __shared__ long long sh_i[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int viol[ASB_CHK_TILES];
    double bmax[ASB_CHK_TILES], bsum[ASB_CHK_TILES], csum[ASB_CHK_TILES][16];
    long long bidx[ASB_CHK_TILES];
#pragma unroll
    for (int ct = 0; ct < ASB_CHK_TILES; ++ct) {
        viol[ct] = ASB_PANEL_COLS; bmax[ct] = -1.0; bsum[ct] = 0.0; bidx[ct] = 0x7fffffffffffffffLL;
#pragma unroll
        for (int t = 0; t < 16; ++t) csum[ct][t] = 0.0;
    }
    for (long long v = (long long)blockIdx.x * 256 + tid; v < n_vert; v += (long long)gridDim.x * 256) {
        double e = E[v];
        const double es = Ecl ? Ecl[v] : e;
        const bool outside = !(es > sc[SC_TAU]) && !(E2 && in_guess(es, E2[v], sc)) && !in_div(es, v, sc);
#pragma unroll
        for (int ct = 0; ct < ASB_CHK_TILES; ++ct) {
            if (ct >= ntile) break;
            const int ncols = wa.nc[ct];
            const double* base = comps + wa.kb[ct] * comp_stride + 3 * v;
            const PanelState* sp = st + ct;
            const double margin = sp->margin;
            const int proven = (int)sp->proven;
            const double e_start = e;
            double loss = 0.0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {                // (eight columns' loads in flight at a time: 24 doubles per thread)
                double c[8][3];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = 8 * h + u;
                    const bool on = t < ncols;
                    c[u][0] = on ? base[(long long)t * comp_stride] : 0.0;
                    c[u][1] = on ? base[(long long)t * comp_stride + 1] : 0.0;
                    c[u][2] = on ? base[(long long)t * comp_stride + 2] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = 8 * h + u;
                    if (t < ncols) {
                        if (outside && t >= proven && t < viol[ct] && !(sp->e_win[t] > e + margin)) viol[ct] = t;
                        const double q = ((c[u][0] * c[u][0] + c[u][1] * c[u][1]) + c[u][2] * c[u][2]) * wn2t3[16 * ct + t];
                        e -= q;
                        loss += q;
                        csum[ct][t] += q;
                    }
                }
            }
            double en = e_start - loss;                  // exactly k_commit_energy's arithmetic
            if (en < 0.0) en = 0.0;
            out.etmp[(long long)ct * n_vert + v] = en;
            bsum[ct] += en;
            if (am_better(en, v, bmax[ct], bidx[ct])) { bmax[ct] = en; bidx[ct] = v; }
            e = en;                                      // what the next tile reads once this one has been adopted
        }
    }
#pragma unroll
    for (int ct = 0; ct < ASB_CHK_TILES; ++ct) {
        if (ct >= ntile) break;                          // (uniform)
        int vl = viol[ct];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int ov = __shfl_xor(vl, o, 64);
            vl = ov < vl ? ov : vl;
        }
        if (lane == 0 && vl < ASB_PANEL_COLS) atomicMin(reinterpret_cast<long long*>(&st[ct].spec_ok), (long long)vl);
#pragma unroll
        for (int t = 0; t < 16; ++t) csum[ct][t] = wave_sum(csum[ct][t]);
        double bs = wave_sum(bsum[ct]), bm = bmax[ct];
        long long bi = bidx[ct];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double om = __shfl_xor(bm, o, 64);
            const long long oi = __shfl_xor(bi, o, 64);
            if (am_better(om, oi, bm, bi)) { bm = om; bi = oi; }
        }
        __syncthreads();                                 // the previous tile's reads of sh are done
        if (lane == 0) {
#pragma unroll
            for (int t = 0; t < 16; ++t) sh[wv * 16 + t] = csum[ct][t];
            sh_m[wv] = bm; sh_i[wv] = bi;
        }
        // (the block's sum of energies goes through the same LDS slots one tile later: a fifth array would do as well)
        __syncthreads();
        if (tid < 16)
            out.colpart[((long long)ct * gridDim.x + blockIdx.x) * 16 + tid] = ((sh[tid] + sh[16 + tid]) + sh[32 + tid]) + sh[48 + tid];
        if (tid == 64) {
            double m = sh_m[0];
            long long ix = sh_i[0];
            for (int w = 1; w < 4; ++w)
                if (am_better(sh_m[w], sh_i[w], m, ix)) { m = sh_m[w]; ix = sh_i[w]; }
            out.pmax[(long long)ct * gridDim.x + blockIdx.x] = m;
            out.pidx[(long long)ct * gridDim.x + blockIdx.x] = ix;
        }
        __syncthreads();
        if (lane == 0) sh[wv] = bs;
        __syncthreads();
        if (tid == 0) out.psum[(long long)ct * gridDim.x + blockIdx.x] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    }
}
// The same with ONE WAVE PER TILE: block = 4 waves over the same 64 vertices, wave ct loads tile ct's columns (48 loads per lane in
// flight instead of eight dependent batches of 24), the tiles' losses meet through LDS, and every wave walks the (three-step) chain
// of clamped energies up to its own tile before it checks its columns -- term by term the arithmetic of k_check_tiles.  Each wave
// reduces its own tile's records: no cross-wave reduction.  62 -> 3x us per read at config 4.
__global__ __launch_bounds__(256) void k_check_tiles_w(const double* __restrict__ comps, long long comp_stride, long long n_vert, WideArgs wa,
                                                       int ntile, const double* __restrict__ wn2t3, const double* __restrict__ E,
                                                       const double* __restrict__ Ecl, const double* __restrict__ E2,
                                                       const double* __restrict__ sc, PanelState* __restrict__ st, CheckOut out) {
    __shared__ double loss_sh[ASB_CHK_TILES][64];
    const int lane = threadIdx.x & 63, ct = threadIdx.x >> 6;
    const bool active = ct < ntile;
    const int ncols = active ? wa.nc[ct] : 0;
    const long long kb = active ? wa.kb[ct] : 0;
    const PanelState* sp = st + (active ? ct : 0);
    const double margin = sp->margin;
    const int proven = (int)sp->proven;
    int viol = ASB_PANEL_COLS;
    double bmax = -1.0, bsum = 0.0, csum[16], wn[16], ew[16];
    long long bidx = 0x7fffffffffffffffLL;
#pragma unroll
    for (int t = 0; t < 16; ++t) { csum[t] = 0.0; wn[t] = active ? wn2t3[16 * ct + t] : 0.0; ew[t] = sp->e_win[t]; }
    for (long long vb = (long long)blockIdx.x * 64; vb < n_vert; vb += (long long)gridDim.x * 64) {
        const long long v = vb + lane;
        const bool valid = v < n_vert;
        double q[16], loss = 0.0;
        {
            const double* base = comps + kb * comp_stride + 3 * (valid ? v : 0);
            double c[16][3];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const bool on = active && valid && t < ncols;
                c[t][0] = on ? base[(long long)t * comp_stride] : 0.0;
                c[t][1] = on ? base[(long long)t * comp_stride + 1] : 0.0;
                c[t][2] = on ? base[(long long)t * comp_stride + 2] : 0.0;
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                q[t] = ((c[t][0] * c[t][0] + c[t][1] * c[t][1]) + c[t][2] * c[t][2]) * wn[t];
                if (t < ncols) loss += q[t];
            }
        }
        loss_sh[ct][lane] = loss;
        __syncthreads();
        if (active && valid) {
            double e = E[v];
            const double es = Ecl ? Ecl[v] : e;
            const bool outside = !(es > sc[SC_TAU]) && !(E2 && in_guess(es, E2[v], sc)) && !in_div(es, v, sc);
            for (int c2 = 0; c2 < ct; ++c2) {            // what the earlier tiles leave (each adopted: clamped at zero)
                double en = e - loss_sh[c2][lane];
                if (en < 0.0) en = 0.0;
                e = en;
            }
            const double e_start = e;
#pragma unroll
            for (int t = 0; t < 16; ++t)
                if (t < ncols) {
                    if (outside && t >= proven && t < viol && !(ew[t] > e + margin)) viol = t;
                    e -= q[t];
                    csum[t] += q[t];
                }
            double en = e_start - loss;                  // exactly k_commit_energy's arithmetic
            if (en < 0.0) en = 0.0;
            out.etmp[(long long)ct * n_vert + v] = en;
            bsum += en;
            if (am_better(en, v, bmax, bidx)) { bmax = en; bidx = v; }
        }
        __syncthreads();                                 // (loss_sh is written again by the next group)
    }
    if (!active) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int ov = __shfl_xor(viol, o, 64);
        viol = ov < viol ? ov : viol;
    }
    if (lane == 0 && viol < ASB_PANEL_COLS) atomicMin(reinterpret_cast<long long*>(&st[ct].spec_ok), (long long)viol);
    wave_sum_dpp<16>(csum);
    bsum = wave_sum(bsum);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double om = __shfl_xor(bmax, o, 64);
        const long long oi = __shfl_xor(bidx, o, 64);
        if (am_better(om, oi, bmax, bidx)) { bmax = om; bidx = oi; }
    }
    if (lane == 0) {
        const long long slot = (long long)ct * gridDim.x + blockIdx.x;
#pragma unroll
        for (int t = 0; t < 16; ++t) out.colpart[slot * 16 + t] = csum[t];
        out.pmax[slot] = bmax;
        out.pidx[slot] = bidx;
        out.psum[slot] = bsum;
    }
}
// res[ct] = columns of tile ct kept (-1: behind a tile that did not stand in full), res[ASB_MAX_SUB] = every tile stood in full,
// res[ASB_MAX_SUB + 1] = L = tiles that stood in full from the front; column sums of those tiles into scal
__global__ __launch_bounds__(1024) void k_tiles_decide(const double* __restrict__ colpart, int nblk, int ntile, WideArgs wa,
                                                       double* __restrict__ scal, PanelState* __restrict__ st, long long* __restrict__ res) {
    __shared__ int full_sh;
    if (threadIdx.x == 0) {
        bool chain = true;
        int L = 0;
        for (int ct = 0; ct < ntile; ++ct) {
            const int ran = (int)st[ct].committed, ok = (int)(st[ct].spec_ok < st[ct].committed ? st[ct].spec_ok : st[ct].committed);
            const bool full = chain && ok == ran;
            res[ct] = chain ? ok : -1;
            st[ct].committed = chain ? ok : 0;
            if (full) L = ct + 1;
            chain = full;
        }
        res[ASB_MAX_SUB] = chain ? 1 : 0;
        res[ASB_MAX_SUB + 1] = L;
        full_sh = L;
    }
    __syncthreads();
    const int L = full_sh, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int p = w; p < L * 16; p += 16) {
        const int ct = p >> 4, t = p & 15;
        if (t >= wa.nc[ct]) continue;
        double v = 0.0;
        for (int b0 = lane; b0 < nblk; b0 += 64 * 8) {          // (eight loads in flight; summed in order)
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + 64 * u;
                x[u] = b < nblk ? colpart[((long long)ct * nblk + b) * 16 + t] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) v += x[u];
        }
        v = wave_sum(v);
        if (lane == 0) scal[(wa.kb[ct] + t) * 4 + 3] = v;
    }
}
__global__ __launch_bounds__(256) void k_apply_tiles(double* __restrict__ E, long long n, int nblk, CheckOut in, const long long* __restrict__ res,
                                                     double* __restrict__ pmax, long long* __restrict__ pidx, double* __restrict__ psum) {
    const int L = (int)res[ASB_MAX_SUB + 1];
    if (L <= 0) return;
    const double* src = in.etmp + (long long)(L - 1) * n;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < n; v += (long long)gridDim.x * 256) E[v] = src[v];
    if (blockIdx.x == 0)
        for (int b = threadIdx.x; b < nblk; b += 256) {
            pmax[b] = in.pmax[(long long)(L - 1) * nblk + b];
            pidx[b] = in.pidx[(long long)(L - 1) * nblk + b];
            psum[b] = in.psum[(long long)(L - 1) * nblk + b];
        }
}

static int spec_tile_finish(asb_ctx* ctx, int ct, long long kb, int nc, PanelState* st, int64_t* kept) {
    const double* Wt = ctx->Wt3 + (size_t)ct * ctx->Fp * 16;
    const int pre = (ctx->pre_orth && ctx->correct_rows) ? 1 : 0;
    if (!pre)
        hipLaunchKernelGGL(k_panel_gram, dim3((unsigned)(kb + nc)), dim3(256), 0, ctx->stream, ctx->W, Wt, (int)ctx->Fp, ctx->gram,
                           ctx->gram_s, ctx->wn2t3 + 16 * ct);
    long long cw = (ctx->n_loc + 255) / 256;
    const int cgrid = (int)(cw < ctx->nblk_cap ? cw : ctx->nblk_cap);
    if (ctx->correct_rows) {
        long long cwr = (ctx->n_loc + 63) / 64;
        hipLaunchKernelGGL(k_correct_rows<true>, dim3((unsigned)(cwr < ctx->nblk_cap ? cwr : ctx->nblk_cap)), dim3(192), 0, ctx->stream,
                           ctx->comps, (long long)(3 * ctx->n_loc), (long long)ctx->n_loc, (int)kb, nc, ctx->gram_s, ctx->wn2t3 + 16 * ct,
                           ctx->energy, ctx->pmax, ctx->pidx, ctx->psum, ctx->colpart, st, ctx->scalar_dev, ctx->sel_e2, ctx->e_class, pre);
    } else
    hipLaunchKernelGGL(k_correct<true>, dim3(cgrid), dim3(256), 0, ctx->stream, ctx->comps, (long long)(3 * ctx->n_loc),
                       (long long)ctx->n_loc, (int)kb, nc, ctx->gram, ctx->wn2t3 + 16 * ct, ctx->energy, ctx->pmax, ctx->pidx,
                       ctx->psum, ctx->colpart, (const long long*)nullptr, (const PanelState*)nullptr, (long long)0, st,
                       ctx->scalar_dev, ctx->sel_e2, ctx->e_class);
    hipLaunchKernelGGL(k_commit_energy, dim3(cgrid), dim3(256), 0, ctx->stream, ctx->comps, (long long)(3 * ctx->n_loc),
                       (long long)ctx->n_loc, (int)kb, st, ctx->wn2t3 + 16 * ct, ctx->energy, ctx->pmax, ctx->pidx, ctx->psum,
                       ctx->colpart, -1);
    ctx->nblk = cgrid;
    hipLaunchKernelGGL(k_colsum, dim3(1), dim3(1024), 0, ctx->stream, ctx->colpart, ctx->nblk, -1, kb, ctx->scal, st);
    ASB_CHECK_LAUNCH(ctx);
    PanelState h;
    int rcs = read_panel_state_from(ctx, st, &h, nullptr);
    if (rcs) return rcs;
    *kept = h.committed;
    return ASB_OK;
}

// ---- the operands of ALL tiles of a read in four launches (blockIdx.y = tile) instead of four per tile: these kernels are
// a few microseconds of work each, their cost is the launch
__global__ __launch_bounds__(256) void k_build_wt_tiles(const double* __restrict__ W, const double* __restrict__ scal, WideArgs wa, int Fp,
                                                        double* __restrict__ Wt3, double* __restrict__ wn2t3) {
    const int ct = blockIdx.y, ncols = wa.nc[ct];
    const long long k0 = wa.kb[ct], total = (long long)Fp * ASB_PANEL_COLS;
    double* Wt = Wt3 + (size_t)ct * Fp * 16;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(i % ASB_PANEL_COLS);
        const long long f = i / ASB_PANEL_COLS;
        Wt[i] = (t < ncols) ? W[(k0 + t) * Fp + f] : 0.0;
    }
    if (blockIdx.x == 0 && threadIdx.x < ASB_PANEL_COLS)
        wn2t3[16 * ct + threadIdx.x] = (threadIdx.x < ncols) ? scal[(k0 + threadIdx.x) * 4 + 1] : 1.0;
}
// G3[ct][j][t] = (w_j . w_t) / |w_j|^2 for j < kb[ct] + nc[ct] (k_panel_gram with scal; K rows of 16 per tile)
__global__ __launch_bounds__(256) void k_panel_gram_tiles(const double* __restrict__ W, const double* __restrict__ Wt3, int Fp, long long K,
                                                          WideArgs wa, const double* __restrict__ scal, double* __restrict__ G3) {
    __shared__ double sh[4 * 16];
    const int ct = blockIdx.y;
    if ((long long)blockIdx.x >= wa.kb[ct] + wa.nc[ct]) return;
    const double* wj = W + (long long)blockIdx.x * Fp;
    const double* Wt = Wt3 + (size_t)ct * Fp * 16;
    double acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = 0.0;
    for (int f = threadIdx.x; f < Fp; f += blockDim.x) {
        const double a = wj[f];
        const double* wt = Wt + (long long)f * ASB_PANEL_COLS;
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] += a * wt[t];
    }
    block_sum<16>(acc, sh);
    if (threadIdx.x < 16)
        G3[((long long)ct * K + blockIdx.x) * 16 + threadIdx.x] = acc[threadIdx.x] / scal[(long long)blockIdx.x * 4 + 1];
}
__global__ __launch_bounds__(256) void k_orth_wt_tiles(const double* __restrict__ W, const double* __restrict__ G3, long long K, WideArgs wa,
                                                       int Fp, double* __restrict__ Wt3) {
    const int ct = blockIdx.y, ncols = wa.nc[ct];
    const long long kb = wa.kb[ct], total = (long long)Fp * ASB_PANEL_COLS;
    const double* G = G3 + (long long)ct * K * 16;
    double* Wt = Wt3 + (size_t)ct * Fp * 16;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(i % ASB_PANEL_COLS);
        const long long f = i / ASB_PANEL_COLS;
        if (t >= ncols) continue;
        double s = 0.0, s2 = 0.0;
        long long j = 0;
        for (; j + 2 <= kb + t; j += 2) {
            s += W[j * Fp + f] * G[j * 16 + t];
            s2 += W[(j + 1) * Fp + f] * G[(j + 1) * 16 + t];
        }
        if (j < kb + t) s += W[j * Fp + f] * G[j * 16 + t];
        Wt[i] -= s + s2;
    }
}
__global__ __launch_bounds__(256) void k_build_wq_tiles(const double* __restrict__ Wt3, int Fp, double* __restrict__ Wq3,
                                                        unsigned* __restrict__ tile_counter) {
    const int ct = blockIdx.y;
    if (blockIdx.x == 0 && ct == 0 && threadIdx.x < 16) tile_counter[threadIdx.x] = 0u;      // the projection kernel's work queue
    const double* Wt = Wt3 + (size_t)ct * Fp * 16;
    double* Wq = Wq3 + (size_t)ct * Fp * 16;
    const long long total = (long long)Fp * 16;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(e & 3), i = (int)((e >> 2) & 15), g = (int)((e >> 6) & 3);
        const long long chunk = e >> 8;
        Wq[e] = Wt[(chunk * 16 + 4 * g + j) * ASB_PANEL_COLS + i];
    }
}
// the same through the per-tile kernels when the weights are not orthogonalised beforehand (ASB_PRE_ORTH=0 / ASB_CORRECT_ROWS=0)
static int dbl_build_tiles(asb_ctx* ctx, int ntile, const WideArgs& wa);

// operands of tile ct of a multi-sub-panel read; pre_orth: orthogonalised weights (k_orth_wt), see spec_tile_finish
static void dbl_build_tile(asb_ctx* ctx, int ct, long long kb, int nc) {
    double* Wt = ctx->Wt3 + (size_t)ct * ctx->Fp * 16;
    hipLaunchKernelGGL(k_build_wt, dim3(64), dim3(256), 0, ctx->stream, ctx->W, ctx->scal, kb, nc, (int)ctx->Fp, Wt, ctx->wn2t3 + 16 * ct);
    if (ctx->pre_orth && ctx->correct_rows) {
        hipLaunchKernelGGL(k_panel_gram, dim3((unsigned)(kb + nc)), dim3(256), 0, ctx->stream, ctx->W, Wt, (int)ctx->Fp, ctx->gram,
                           (double*)nullptr, (const double*)nullptr, (const double*)ctx->scal);
        hipLaunchKernelGGL(k_orth_wt, dim3(128), dim3(256), 0, ctx->stream, ctx->W, ctx->gram, kb, nc, (int)ctx->Fp, Wt);
    }
    hipLaunchKernelGGL(k_build_wq, dim3(64), dim3(256), 0, ctx->stream, Wt, (int)ctx->Fp, ctx->Wq3 + (size_t)ct * ctx->Fp * 16,
                       ctx->tile_counter);
}
static int dbl_build_tiles(asb_ctx* ctx, int ntile, const WideArgs& wa) {
    if (!(ctx->pre_orth && ctx->correct_rows)) {
        for (int ct = 0; ct < ntile; ++ct) dbl_build_tile(ctx, ct, wa.kb[ct], wa.nc[ct]);
        ASB_CHECK_LAUNCH(ctx);
        return ASB_OK;
    }
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->gram3, (size_t)ASB_MAX_SUB * ctx->K * 16))) return rc;
    const long long rows_g = wa.kb[ntile - 1] + wa.nc[ntile - 1];
    hipLaunchKernelGGL(k_build_wt_tiles, dim3(64, ntile), dim3(256), 0, ctx->stream, ctx->W, ctx->scal, wa, (int)ctx->Fp, ctx->Wt3, ctx->wn2t3);
    hipLaunchKernelGGL(k_panel_gram_tiles, dim3((unsigned)rows_g, ntile), dim3(256), 0, ctx->stream, ctx->W, ctx->Wt3, (int)ctx->Fp,
                       (long long)ctx->K, wa, ctx->scal, ctx->gram3);
    hipLaunchKernelGGL(k_orth_wt_tiles, dim3(128, ntile), dim3(256), 0, ctx->stream, ctx->W, ctx->gram3, (long long)ctx->K, wa, (int)ctx->Fp,
                       ctx->Wt3);
    hipLaunchKernelGGL(k_build_wq_tiles, dim3(64, ntile), dim3(256), 0, ctx->stream, ctx->Wt3, (int)ctx->Fp, ctx->Wq3, ctx->tile_counter);
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}
// The sub-panels of a read in ONE launch (k_panel_multi): arm, kernel, one read of the summary.
// *ntile = -1: the launch did not run to the end of its first sub-panel (exchange timed out / too many candidates): nothing
// was committed, the caller takes the one-by-one path with its fallbacks.
static int dbl_build_tiles(asb_ctx* ctx, int ntile, const WideArgs& wa);
static int launch_wide(asb_ctx* ctx, int ntile, const WideArgs& wa);
// spec_ntile / spec_nc (optional): the read's PASS is enqueued right behind the panel kernel on the column counts the
// sub-panels are EXPECTED to reach (their step budgets), before the host knows what they reached -- it runs while the host waits
// for the summary instead of after it.  Safe: a sub-panel that ended early only means columns computed from rows of W nobody
// wrote (they lie at and beyond the first column that is not committed, and a later tile exists only behind a FULL one, so no
// committed column is ever orthogonalised against them); the caller checks the tiles on the counts really reached.
static int multi_chain_run(asb_ctx* ctx, long long k, long long k1, int nsub_max, int* ntile, int* nc, int* proven,
                           int* spec_ntile = nullptr, int* spec_nc = nullptr, bool assembled = false) {
    int rc;
    *ntile = -1;
    ctx->chain_timed_out = 0;
    if (spec_ntile) *spec_ntile = 0;
    int cgrid_all = 0;
    size_t n_words = 0;
    if ((rc = coop_buffers(ctx, &cgrid_all, &n_words))) return rc;
    MultiArgs ma{};
    int n = 0;
    for (int sp = 0; sp < nsub_max && sp < ASB_MAX_SUB && k + (long long)sp * ASB_PANEL_COLS < k1; ++sp) {
        const long long kb = k + (long long)sp * ASB_PANEL_COLS;
        int steps = (int)((k1 - kb) < ASB_PANEL_COLS ? (k1 - kb) : ASB_PANEL_COLS);
        if (sp > 0 && steps > ctx->sub_budget[sp]) steps = ctx->sub_budget[sp];
        ma.kb[n] = kb;
        ma.steps[n] = steps;
        ma.spec_max[n] = sp == 0 ? ctx->spec_budget : ASB_PANEL_COLS;
        ++n;
    }
    ma.nsub = n;
    // records and the winner's weight buffers start out as "not written" (all bits set: filled by the arm kernel); the
    // speculative buffers are reset by their owners at the start of the panel kernel
    const int n_small = 3 * cgrid_all * 2 + 3 * ((int)ctx->Fp + 8);
    hipLaunchKernelGGL(k_panel_arm_multi, dim3(1), dim3(256), 0, ctx->stream, ctx->pstate, ctx->pstate2, ctx->scalar_dev, ASB_MARGIN_REL,
                       ctx->coop_bar, (unsigned long long*)ctx->coop_rec, n_small, ma, (long long)(assembled ? ctx->n_slots_host : -1));
    bool launched = false;
    if ((rc = launch_panel_multi_any(ctx, cgrid_all, ma, &launched, ctx->pstate2, false))) return rc;
    if (!launched) return ASB_OK;
    unsigned long long sum[9], seq = 0;
    if ((rc = fetch_multi_begin(ctx, &seq))) return rc;
    if (spec_ntile && seq && ctx->spec_pass && ctx->spec_budget >= ASB_PANEL_COLS && ctx->pre_orth && ctx->correct_rows) {
        WideArgs wa{};
        int nt = 0;
        for (int sp = 0; sp < n; ++sp) {
            wa.kb[nt] = ma.kb[sp];
            wa.nc[nt] = ma.steps[sp];
            spec_nc[nt] = ma.steps[sp];
            ++nt;
            if (ma.steps[sp] < ASB_PANEL_COLS) break;
        }
        if ((rc = dbl_build_tiles(ctx, nt, wa))) return rc;
        if ((rc = launch_wide(ctx, nt, wa))) return rc;
        *spec_ntile = nt;
    }
    if ((rc = fetch_multi_end(ctx, seq, sum))) return rc;
    ctx->n_panels++;
    ctx->run_coop_used = 1;
    if (getenv("ASB_DEBUG_PANELS")) print_multi_timeline(ctx, (int)(sum[0] & 0xffffffffu) + 16 * (n - 1));
    int nt = 0;
    for (int sp = 0; sp < n; ++sp) {
        const long long committed = (long long)(sum[sp] & 0xffffffffu), prov = (long long)(sum[sp] >> 32) - 1;
        if (prov < 0) {                                  // the launch did not finish this sub-panel (sum[8] says why)
            ctx->chain_timed_out = 1;                    // (several ranks: the driver makes all of them leave the kernel together)
            if (sp == 0) return ASB_OK;                  // *ntile = -1: the one-by-one path meets the same and falls back
            ctx->panel_coop = 0;                         // a later one timed out: what stands stands, the context leaves the kernel
            ctx->coop_test_stall = 0;
            ctx->n_coop_fallbacks++;
            break;
        }
        if (committed <= 0) break;
        nc[nt] = (int)committed;
        proven[nt] = (int)(prov > committed ? committed : prov);
        ++nt;
        if (committed < ASB_PANEL_COLS) break;
    }
    *ntile = nt;
    return ASB_OK;
}
// the checks of all tiles of a read, enqueued back to back (pre-orthogonalised weights, one-thread-per-row check): per tile the
// check with tentative energies, the one-block decision (chain flag: a tile counts only behind tiles that stood in full) and the
// conditional adoption; ctx->tile_res holds the per-tile results afterwards
static int tiles_enqueue(asb_ctx* ctx, int ntile, const long long* kb, const int* nc, PanelState* const* st, int* rgrid_out, int* cgrid_out) {
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->e_tmp, (size_t)ctx->n_loc))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->tile_res, (size_t)ASB_MAX_SUB + 2))) return rc;
    long long cwr = (ctx->n_loc + 63) / 64;
    // (two blocks per CU, grid-strided: the one-block k_tile_decide sums a partial per block and column)
    static const int bpc = getenv("ASB_CHECK_BLOCKS_PER_CU") ? atoi(getenv("ASB_CHECK_BLOCKS_PER_CU")) : 2;
    const long long rcap = (long long)(bpc < 1 ? 1 : (bpc > 8 ? 8 : bpc)) * ctx->n_cu;
    const int rgrid = (int)(cwr < rcap ? cwr : rcap);
    long long cw = (ctx->n_loc + 255) / 256;
    const int cgrid = (int)(cw < ctx->nblk_cap ? cw : ctx->nblk_cap);
    static const int fused = getenv("ASB_CHECK_FUSED") ? atoi(getenv("ASB_CHECK_FUSED")) : 2;
    if (fused && ntile <= ASB_CHK_TILES) {
        long long cb = (ctx->n_loc + 255) / 256;
        int fgrid = (int)(cb < ctx->nblk_cap ? cb : ctx->nblk_cap);
        if ((rc = asb_alloc(ctx, &ctx->e_tmp4, (size_t)ASB_CHK_TILES * ctx->n_loc))) return rc;
        if ((rc = asb_alloc(ctx, &ctx->chk_rec, (size_t)ASB_CHK_TILES * ctx->nblk_cap * 18))) return rc;
        if ((rc = asb_alloc(ctx, &ctx->chk_idx, (size_t)ASB_CHK_TILES * ctx->nblk_cap))) return rc;
        CheckOut co{ctx->e_tmp4, ctx->chk_rec, ctx->chk_idx, ctx->chk_rec + (size_t)ASB_CHK_TILES * ctx->nblk_cap,
                    ctx->chk_rec + (size_t)2 * ASB_CHK_TILES * ctx->nblk_cap};
        WideArgs wa{};
        for (int ct = 0; ct < ntile; ++ct) {
            wa.kb[ct] = kb[ct];
            wa.nc[ct] = nc[ct];
            if (st[ct] != st[0] + ct) ASB_FAIL(ctx, ASB_ERR_ARG, "tiles_enqueue: the tiles' states are not contiguous");
        }
        if (fused >= 2) {                                // one wave per tile (blocks of 64 vertices)
            long long cg = (ctx->n_loc + 63) / 64;           // (two blocks per CU, grid-strided: the one-block decision sums a record per block)
            const long long cgc = 2LL * ctx->n_cu < ctx->nblk_cap ? 2LL * ctx->n_cu : ctx->nblk_cap;
            fgrid = (int)(cg < cgc ? cg : cgc);
            hipLaunchKernelGGL(k_check_tiles_w, dim3(fgrid), dim3(256), 0, ctx->stream, ctx->comps, (long long)(3 * ctx->n_loc),
                               (long long)ctx->n_loc, wa, ntile, ctx->wn2t3, ctx->energy, ctx->e_class, ctx->sel_e2, ctx->scalar_dev, st[0], co);
        } else
        hipLaunchKernelGGL(k_check_tiles, dim3(fgrid), dim3(256), 0, ctx->stream, ctx->comps, (long long)(3 * ctx->n_loc), (long long)ctx->n_loc,
                           wa, ntile, ctx->wn2t3, ctx->energy, ctx->e_class, ctx->sel_e2, ctx->scalar_dev, st[0], co);
        hipLaunchKernelGGL(k_tiles_decide, dim3(1), dim3(1024), 0, ctx->stream, co.colpart, fgrid, ntile, wa, ctx->scal, st[0], ctx->tile_res);
        hipLaunchKernelGGL(k_apply_tiles, dim3(cgrid), dim3(256), 0, ctx->stream, ctx->energy, (long long)ctx->n_loc, fgrid, co, ctx->tile_res,
                           ctx->pmax, ctx->pidx, ctx->psum);
        ASB_CHECK_LAUNCH(ctx);
        *rgrid_out = fgrid;
        *cgrid_out = cgrid;
        return ASB_OK;
    }
    for (int ct = 0; ct < ntile; ++ct) {
        hipLaunchKernelGGL(k_correct_rows<true>, dim3(rgrid), dim3(192), 0, ctx->stream, ctx->comps, (long long)(3 * ctx->n_loc),
                           (long long)ctx->n_loc, (int)kb[ct], nc[ct], ctx->gram_s, ctx->wn2t3 + 16 * ct, ctx->energy, ctx->pmax, ctx->pidx,
                           ctx->psum, ctx->colpart, st[ct], ctx->scalar_dev, ctx->sel_e2, ctx->e_class, 1, ctx->e_tmp);
        hipLaunchKernelGGL(k_tile_decide, dim3(1), dim3(1024), 0, ctx->stream, ctx->colpart, rgrid, ct, (long long)kb[ct], ctx->scal, st[ct],
                           ctx->tile_res);
        hipLaunchKernelGGL(k_apply_tmp, dim3(cgrid), dim3(256), 0, ctx->stream, ctx->energy, ctx->e_tmp, (long long)ctx->n_loc, ctx->tile_res);
    }
    ASB_CHECK_LAUNCH(ctx);
    *rgrid_out = rgrid;
    *cgrid_out = cgrid;
    return ASB_OK;
}
static int panel_candidates(asb_ctx* ctx, long long k, int stalled);
static int double_panel(asb_ctx* ctx, long long k, long long k1, int64_t* done_out) {
    int rc;
    *done_out = 0;
    const int nsub_lim = ctx->sub_panels < 1 ? 1 : (ctx->sub_panels > ASB_MAX_SUB ? ASB_MAX_SUB : ctx->sub_panels);
    if (ctx->sub_cur < 1) ctx->sub_cur = nsub_lim < ctx->sub_first ? nsub_lim : ctx->sub_first;
    const int nsub_max = ctx->sub_cur < nsub_lim ? ctx->sub_cur : nsub_lim;
    if ((rc = asb_alloc(ctx, &ctx->Wt3, (size_t)ASB_MAX_SUB * ctx->Fp * 16))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->Wq3, (size_t)ASB_MAX_SUB * ctx->Fp * 16))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->wn2t3, (size_t)16 * ASB_MAX_SUB))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->tile_counter, (size_t)16))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->pstate2, (size_t)ASB_MAX_SUB))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->e_class, (size_t)ctx->n_loc))) return rc;
    const auto t_read0 = std::chrono::steady_clock::now();
    if ((rc = panel_candidates(ctx, k, 0))) return rc;
    // who is a candidate is decided by the energies NOW; the later tiles' checks run after the earlier tiles' updates
    ASB_HIP(ctx, hipMemcpyAsync(ctx->e_class, ctx->energy, (size_t)ctx->n_loc * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    long long kb[ASB_MAX_SUB];
    int nc[ASB_MAX_SUB] = {0}, proven[ASB_MAX_SUB] = {0};
    for (int sp = 0; sp < ASB_MAX_SUB; ++sp) kb[sp] = k + (long long)sp * ASB_PANEL_COLS;
    int ntile = 0, spec_ntile = 0, spec_nc[ASB_MAX_SUB] = {0};
    bool chained_runs = false;
    if (ctx->sub_chain && ctx->panel_coop && ctx->spec_panels && ctx->Fp <= 2048 && nsub_max > 1) {
        int nt = -1;
        if ((rc = multi_chain_run(ctx, k, k1, nsub_max, &nt, nc, proven, &spec_ntile, spec_nc))) return rc;
        if (nt == 0) return ASB_OK;                      // nothing committed: the caller's refresh / forced path
        if (nt > 0) { ntile = nt; chained_runs = true; }
    }
    for (int sp = 0; !chained_runs && sp < nsub_max && kb[sp] < k1; ++sp) {
        int steps = (int)((k1 - kb[sp]) < ASB_PANEL_COLS ? (k1 - kb[sp]) : ASB_PANEL_COLS);
        if (sp > 0) {
            // a later sub-panel runs on rows chosen for the first, mostly unproven (the bound on the vertices outside is the
            // stale one): it is given as many steps as the last ones kept (+2) -- a rejected step costs a panel step and
            // everything behind it
            if (steps > ctx->sub_budget[sp]) steps = ctx->sub_budget[sp];
        }
        int64_t ran = 0;
        ctx->run_writeback = 1;
        ctx->run_spec_max = ctx->spec_panels ? (sp == 0 ? ctx->spec_budget : ASB_PANEL_COLS) : 0;
        rc = asb_panel_run(ctx, kb[sp], steps, 0, 0, &ran);      // sp > 0: same candidates, rows as the last sub-panel left them
        ctx->run_writeback = 0;
        ctx->run_spec_max = 0;
        if (rc) return rc;
        // this sub-panel's state (winner energies, provable head) is needed again after the pass; a later launch re-arms pstate
        ASB_HIP(ctx, hipMemcpyAsync(ctx->pstate2 + sp, ctx->pstate, sizeof(PanelState), hipMemcpyDeviceToDevice, ctx->stream));
        if (sp > 0) ctx->n_panels--;                     // statistics count reads of X
        if (sp == 0 && ran == 0) return ASB_OK;          // the caller's refresh / forced path
        if (ran == 0 || (sp > 0 && !ctx->run_coop_used)) break;
        nc[sp] = (int)ran;
        proven[sp] = (int)ctx->run_proven;
        ntile = sp + 1;
        // another sub-panel only behind a full one that ran in the co-resident kernel (the two-kernel loop leaves the rows
        // as they were) and may go on unproven
        if (ran < ASB_PANEL_COLS || !ctx->run_coop_used || !ctx->spec_panels) break;
    }
    PanelState* st[ASB_MAX_SUB];
    for (int ct = 0; ct < ntile; ++ct) st[ct] = ctx->pstate2 + ct;      // (k_panel_multi writes them there; the one-by-one loop copies)
    // one read of X for all tiles
    WideArgs wa{};
    for (int ct = 0; ct < ntile; ++ct) {
        wa.kb[ct] = kb[ct];
        wa.nc[ct] = nc[ct];
    }
    // (the pass may already be running: enqueued behind the panel kernel on the expected counts, which cover the reached ones)
    bool covered = chained_runs && spec_ntile >= ntile;
    for (int ct = 0; covered && ct < ntile; ++ct) covered = spec_nc[ct] >= nc[ct];
    if (!covered) {
        if ((rc = dbl_build_tiles(ctx, ntile, wa))) return rc;
        if ((rc = launch_wide(ctx, ntile, wa))) return rc;
    }
    int64_t total = 0;
    int full = 0;
    bool rejected = false;
    const bool chained = ctx->pre_orth && ctx->correct_rows && ctx->tile_chain;
    if (chained) {
        // all tiles enqueued back to back, ONE host read
        int rgrid = 0, cgrid = 0;
        if ((rc = tiles_enqueue(ctx, ntile, kb, nc, st, &rgrid, &cgrid))) return rc;
        long long res[ASB_MAX_SUB + 1];
        if ((rc = fetch_words(ctx, ctx->tile_res, ASB_MAX_SUB + 1, res))) return rc;
        ctx->nblk = rgrid;                           // the records of the last tile that stood in full
        for (int ct = 0; ct < ntile; ++ct) {
            int64_t kept = res[ct] < 0 ? 0 : res[ct];
            if (res[ct] >= 0 && kept < nc[ct]) {     // this tile did not stand in full: its energies are untouched, commit the head
                hipLaunchKernelGGL(k_commit_energy, dim3(cgrid), dim3(256), 0, ctx->stream, ctx->comps, (long long)(3 * ctx->n_loc),
                                   (long long)ctx->n_loc, (int)kb[ct], st[ct], ctx->wn2t3 + 16 * ct, ctx->energy, ctx->pmax, ctx->pidx,
                                   ctx->psum, ctx->colpart, (int)kept);
                ctx->nblk = cgrid;
                hipLaunchKernelGGL(k_colsum, dim3(1), dim3(1024), 0, ctx->stream, ctx->colpart, ctx->nblk, (int)kept, kb[ct], ctx->scal,
                                   (PanelState*)nullptr);
                ASB_CHECK_LAUNCH(ctx);
            }
            if (res[ct] < 0) break;                  // behind a tile that did not stand
            ctx->n_spec_steps += nc[ct] - proven[ct];
            ctx->n_spec_kept += kept > proven[ct] ? kept - proven[ct] : 0;
            total += kept;
            if (getenv("ASB_DEBUG_PANELS"))
                fprintf(stderr, "[asb] panel at k=%lld tile %d: %d proven + %lld of %d unproven steps kept\n", k, ct, proven[ct],
                        (long long)(kept > proven[ct] ? kept - proven[ct] : 0), nc[ct] - proven[ct]);
            if (ct >= 1) {
                const int want = (int)kept + 2;
                ctx->sub_budget[ct] = want < 4 ? 4 : (want > ASB_PANEL_COLS ? ASB_PANEL_COLS : want);
            }
            if (kept < nc[ct]) { rejected = true; break; }
            ++full;
        }
    }
    for (int ct = 0; ct < ntile && !chained; ++ct) {
        int64_t kept = 0;
        if ((rc = spec_tile_finish(ctx, ct, kb[ct], nc[ct], st[ct], &kept))) return rc;
        ctx->n_spec_steps += nc[ct] - proven[ct];
        ctx->n_spec_kept += kept > proven[ct] ? kept - proven[ct] : 0;
        total += kept;
        if (getenv("ASB_DEBUG_PANELS"))
            fprintf(stderr, "[asb] panel at k=%lld tile %d: %d proven + %lld of %d unproven steps kept\n", k, ct, proven[ct],
                    (long long)(kept > proven[ct] ? kept - proven[ct] : 0), nc[ct] - proven[ct]);
        if (ct >= 1) {                                // adapt the later sub-panels' lengths to what stands
            const int want = (int)kept + 2;
            ctx->sub_budget[ct] = want < 4 ? 4 : (want > ASB_PANEL_COLS ? ASB_PANEL_COLS : want);
        }
        if (kept < nc[ct]) { rejected = true; break; }      // what follows was built on a rejected step
        ++full;
    }
    // (chained tiles: the host has just read the tile results, the read's GPU work is done)
    const double read_ms = chained ? std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_read0).count() : -1.0;
    // how many sub-panels the next read of X gets: twice as many after a read whose sub-panels all stood, what stood
    // (+1) after a rejection -- a rejected sub-panel costs its panel steps and its share of the MFMA work
    if (rejected) ctx->sub_cur = full + 1 < nsub_lim ? full + 1 : nsub_lim;
    else if (ntile == nsub_max) ctx->sub_cur = 2 * nsub_max < nsub_lim ? 2 * nsub_max : nsub_lim;
    if (total > 0) ctx->k_done = k + total;
    // Structured data: the ranking reshuffled under this read's candidates.  The columns of its rejected steps are a sketch
    // of the residual of EVERY vertex (asb_sketch.hip): a greedy replay in that space names the next read's candidates, and
    // the next read gets all its sub-panels again (its rejected columns are the sketch after it).
    // Is a predicted read worth what it costs?  It reads X for all four sub-panels and pays the replay (about 2.9 ms at config 4's
    // size) where a plain read behind a rejection takes one or two sub-panels (1.3 - 1.6 ms): on low-rank data it commits 4x the
    // components, on a slowly decaying spectrum or on localised modes 1.3 - 1.5x (tools/structured_probe.py: 9 reads in 23.7 ms
    // against 12 in 16.3, 16 in 42.9 against 23 in 30.3 when every rejection was answered by a replay).  So both kinds of read are
    // rated, components per (modelled) millisecond, as exponential means; the better one is taken, the other tried again every
    // sixth read.
    {
        static const double pass_ms[5] = {0.0, 0.87, 1.0, 1.25, 1.41};
        // (a plain read is rated at the size the adaptation would have given it -- the sub-panels its kept steps fill -- not at the
        // full size a first read or a read behind a predicted one happens to have)
        int nt = ntile < 1 ? 1 : (ntile > 4 ? 4 : ntile);
        if (!ctx->read_by_score) {
            const int need = (int)((total + ASB_PANEL_COLS) / ASB_PANEL_COLS);
            nt = need < 1 ? 1 : (need < nt ? need : nt);
        }
        // What a read costs is MEASURED: the fastest of the first four reads of each size on this context and shape (host clock
        // from its selection to its tile results -- the host waits for the GPU there anyway; the first ones may pay for
        // allocations) and likewise of the first replays fix cost_nt[] / cost_replay, which stay frozen afterwards (the same
        // tensor takes the same decisions on every later call).  Until a size has been seen, the figures of the 100 000 x 2000
        // tensor on an MI355X scaled by the shard's size stand in for it.
        if (ctx->cost_n != ctx->n_loc || ctx->cost_Fp != ctx->Fp) {
            for (int q = 0; q < 5; ++q) { ctx->cost_nt[q] = -1.0; ctx->cost_cnt[q] = 0; }
            ctx->cost_replay = -1.0;
            ctx->cost_replay_cnt = 0;
            ctx->cost_n = ctx->n_loc;
            ctx->cost_Fp = ctx->Fp;
        }
        {
            const int nm = ntile < 1 ? 1 : (ntile > 4 ? 4 : ntile);
            if (ctx->cost_cnt[nm] < 4 && read_ms > 0.0 && !getenv("ASB_DEBUG_PANELS")) {
                ctx->cost_nt[nm] = (ctx->cost_nt[nm] < 0.0 || read_ms < ctx->cost_nt[nm]) ? read_ms : ctx->cost_nt[nm];
                ctx->cost_cnt[nm]++;
            }
        }
        const double scale = (double)ctx->n_loc * (double)ctx->Fp / (100000.0 * 2000.0);
        const double modelled = (0.45 + 0.23 * nt + pass_ms[nt]) * (scale > 0.05 ? scale : 0.05);
        const double replay_cost = ctx->cost_replay >= 0.0 ? ctx->cost_replay : 0.7 * (scale > 0.05 ? scale : 0.05);
        const double cost = (ctx->cost_nt[nt] >= 0.0 ? ctx->cost_nt[nt] : modelled) + (ctx->read_by_score ? replay_cost : 0.0);
        const double rate = (double)total / cost;
        double& ema = ctx->read_by_score ? ctx->rate_sketch : ctx->rate_plain;
        ema = ema < 0.0 ? rate : 0.5 * (ema + rate);
        ctx->mode_streak = (ctx->read_by_score == ctx->last_by_score) ? ctx->mode_streak + 1 : 1;
        ctx->last_by_score = ctx->read_by_score;
    }
    bool want_replay = true;
    if (ctx->rate_sketch >= 0.0 && ctx->rate_plain >= 0.0) want_replay = ctx->rate_sketch >= ctx->rate_plain;
    if (ctx->mode_streak >= ctx->probe_after && ctx->rate_sketch >= 0.0 && ctx->rate_plain >= 0.0) {
        want_replay = !ctx->last_by_score;                  // look at the other kind again, ever more rarely while the verdict stands
        ctx->probe_after = ctx->probe_after < 64 ? 2 * ctx->probe_after : 64;
    }
    if (getenv("ASB_DEBUG_PANELS") && ctx->read_by_score && ctx->sk_pred) {
        // how good was the replay that named this read's candidates?  its winners (local vertex ids) against the read's own
        long long pr[64];
        std::vector<double> sc4((size_t)(total + 1) * 4);
        (void)hipMemcpy(pr, ctx->sk_pred, sizeof(pr), hipMemcpyDeviceToHost);
        (void)hipMemcpy(sc4.data(), ctx->scal + k * 4, sc4.size() * sizeof(double), hipMemcpyDeviceToHost);
        int agree = 0;
        for (; agree < total && agree < 64; ++agree) {
            long long g;
            memcpy(&g, &sc4[(size_t)agree * 4 + 2], 8);
            if (pr[agree] + ctx->v0 != g) break;
        }
        long long gq = -1;
        if (total < 64) memcpy(&gq, &sc4[(size_t)total * 4 + 2], 8);
        fprintf(stderr, "[asb]   the replay named the first %d of the %lld winners kept; at the step that fell it had vertex %lld, the panel took %lld\n",
                agree, (long long)total, total < 64 ? pr[total] + ctx->v0 : -1LL, gq);
    }
    if (getenv("ASB_DEBUG_PANELS"))
        fprintf(stderr, "[asb] read at k=%lld (%s candidates) kept %lld: components per ms plain %.1f, predicted %.1f -> %s next\n", k,
                ctx->read_by_score ? "predicted" : "plain", (long long)total, ctx->rate_plain, ctx->rate_sketch,
                want_replay ? "replay" : "plain");
    if (ctx->sketch && want_replay && !ctx->sketch_run_off && rejected && total > 0 && k + total < k1) {
        long long ncols = 0;
        for (int ct = 0; ct < ntile; ++ct) ncols += nc[ct];
        const long long left = ncols - total;
        static const int sk_min_cols = getenv("ASB_SKETCH_MIN_COLS") ? atoi(getenv("ASB_SKETCH_MIN_COLS")) : 8;
        if (left >= sk_min_cols && ctx->n_loc > ctx->m_cap) {        // (above one co-resident launch the replay runs on the largest energies)
            const long long ks = k + total, todo = k1 - ks;
            const int r = (int)(left < 64 ? left : 64), steps = (int)(todo < 64 ? todo : 64);
            const auto t_rep0 = std::chrono::steady_clock::now();
            if ((rc = asb_sketch_predict(ctx, ctx->comps + (size_t)ks * 3 * ctx->n_loc, (long long)(3 * ctx->n_loc), ctx->scal + ks * 4 + 1, 4,
                                         ctx->energy, (long long)ctx->n_loc, r, steps)))
                return rc;
            // The kernel decides itself whether the sketch holds enough of the residual to be replayed (asb_sketch.hip).  The host
            // reads that verdict here -- a thin sketch ends the launch within its load phase, a replay costs the GPU no idle time
            // beyond the enqueue latency of the next read -- because it decides the next read's shape: predicted candidates get
            // all sub-panels; on noise-like data the plain adaptation stands, and this run launches no further replays.
            unsigned fl[4] = {0, 0, 0, 0};
            if ((rc = fetch_words(ctx, ctx->sk_flags, 2, fl))) return rc;
            const bool replayed = fl[2] != 0 && fl[1] == 0;
            if (replayed && ctx->cost_replay_cnt < 4 && steps == 64 && !getenv("ASB_DEBUG_PANELS")) {
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_rep0).count();
                ctx->cost_replay = (ctx->cost_replay < 0.0 || ms < ctx->cost_replay) ? ms : ctx->cost_replay;
                ctx->cost_replay_cnt++;
            }
            if (replayed) {
                ctx->sketch_valid = true;
                ctx->sub_cur = nsub_lim;
                for (int sp = 0; sp < 8; ++sp) ctx->sub_budget[sp] = ASB_PANEL_COLS;
            } else if (fl[3]) {
                ctx->sketch_run_off = true;
            }
            if (getenv("ASB_DEBUG_PANELS"))
                fprintf(stderr, "[asb] sketch of %d columns at k=%lld holds %.3f of the residual: %s\n", r, ks,
                        [&] { float f; memcpy(&f, &fl[0], 4); return (double)f; }(), replayed ? "replayed" : (fl[3] ? "holds too little of the residual: candidates by energy" : "exchange timed out"));
        }
    }
    // (noise-like data: a rejection comes from the random cross terms of the first components, not from a ranking that drifts --
    // the next read is given all its sub-panels again; measured over eight random tensors: 6.4 ms per step against 6.5 with the
    // cautious "what stood + 1")
    if (ctx->sketch_run_off && rejected) {
        ctx->sub_cur = nsub_lim;
        for (int sp = 0; sp < 8; ++sp) ctx->sub_budget[sp] = ASB_PANEL_COLS;
    }
    // Behind a rejection that no replay answers, the next read's candidates are half the largest energies, half an energy-weighted
    // random sample of everyone (in_div), and it gets all its sub-panels: its winners still come from the first half, and the
    // columns of its rejected steps -- greedy steps on rows from ALL over the mesh -- are the sketch the read after it is
    // predicted from.  CPU replay (tools/sim_sketch2.py, N = 100 000, F = 256, K = 128, 50 localised bumps): 6 reads (1, 39, 9, 20,
    // 35, 24 components) against 9 with candidates by energy and replay alone; low rank and the slow spectrum unchanged (6, 8).
    // (only where a predicted read is what the rating wants next and there is no sketch to predict it from -- too few rejected
    // columns, or a probe of the other kind in between: on data where plain reads rate better, e.g. a slowly decaying spectrum, a
    // diverse read would only be a dearer plain read: 27.5 against 17.5 ms when every rejection was answered by one)
    if (diverse_on(ctx) && want_replay && ctx->sketch && rejected && !ctx->sketch_valid && !ctx->sketch_run_off && total > 0 && k + total < k1) {
        ctx->diverse_next = true;
        ctx->sub_cur = nsub_lim;
        for (int sp = 0; sp < 8; ++sp) ctx->sub_budget[sp] = ASB_PANEL_COLS;
    }
    *done_out = total;
    return ASB_OK;
}

// ---- the same read of X with several sub-panels, in steps, for the multi-rank driver (_panels.py): every rank runs the
// sub-panels on the identical assembled candidates, projects its shard once, and the tiles are checked one at a time with
// a min over the ranks in between (the driver's all-reduce of the device word)
// (every sub-panel's state is copied to pstate2[sp] right behind its run: what a later launch -- which re-arms pstate --
// does or does not commit cannot change which record a tile is checked against)
static PanelState* sub_state(asb_ctx* ctx, int ct) { return ctx->pstate2 + ct; }
extern "C" int asb_panel_sub_run(asb_ctx* ctx, int sp, int64_t k0, int steps, int spec_max, int64_t* ran, int64_t* proven, int* may_continue) {
    if (!ctx || !ctx->candR || ctx->mode != ASB_DEFLATE_PROJECT || !ran || !proven || !may_continue) return ASB_ERR_ARG;
    if (sp < 0 || sp >= ASB_MAX_SUB || spec_max < 0) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_panel_sub_run: bad sub-panel");
    int rc;
    if (sp == 0) {
        if ((rc = asb_alloc(ctx, &ctx->Wt3, (size_t)ASB_MAX_SUB * ctx->Fp * 16))) return rc;
        if ((rc = asb_alloc(ctx, &ctx->Wq3, (size_t)ASB_MAX_SUB * ctx->Fp * 16))) return rc;
        if ((rc = asb_alloc(ctx, &ctx->wn2t3, (size_t)16 * ASB_MAX_SUB))) return rc;
        if ((rc = asb_alloc(ctx, &ctx->tile_counter, (size_t)16))) return rc;
        if ((rc = asb_alloc(ctx, &ctx->pstate2, (size_t)ASB_MAX_SUB))) return rc;
        if ((rc = asb_alloc(ctx, &ctx->e_class, (size_t)ctx->n_loc))) return rc;
        ASB_HIP(ctx, hipMemcpyAsync(ctx->e_class, ctx->energy, (size_t)ctx->n_loc * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    }
    ctx->run_writeback = 1;
    ctx->run_spec_max = ctx->spec_panels ? spec_max : 0;
    rc = asb_panel_run(ctx, k0, steps, 0, 1, ran);
    ctx->run_writeback = 0;
    ctx->run_spec_max = 0;
    if (rc) return rc;
    ASB_HIP(ctx, hipMemcpyAsync(ctx->pstate2 + sp, ctx->pstate, sizeof(PanelState), hipMemcpyDeviceToDevice, ctx->stream));
    if (sp > 0) ctx->n_panels--;                          // statistics count reads of X
    *proven = ctx->run_proven;
    *may_continue = (*ran == ASB_PANEL_COLS && ctx->run_coop_used && ctx->spec_panels) ? 1 : 0;
    return ASB_OK;
}
extern "C" int asb_panel_sub_project(asb_ctx* ctx, int64_t k0, int ntile, const int* nc) {
    if (!ctx || !ctx->energy || ctx->mode != ASB_DEFLATE_PROJECT || !nc || ntile < 1 || ntile > ASB_MAX_SUB) return ASB_ERR_ARG;
    WideArgs wa{};
    for (int ct = 0; ct < ntile; ++ct) {
        if (nc[ct] < 1 || nc[ct] > ASB_PANEL_COLS || k0 + 16 * ct + nc[ct] > ctx->K) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_panel_sub_project: bad range");
        wa.kb[ct] = k0 + 16 * ct;
        wa.nc[ct] = nc[ct];
    }
    int rc;
    if ((rc = dbl_build_tiles(ctx, ntile, wa))) return rc;
    ctx->sub_ntile = ntile;
    return launch_wide(ctx, ntile, wa);
}
extern "C" int asb_panel_sub_check(asb_ctx* ctx, int ct, int64_t kb, int nc, double* first_rejected_dev) {
    if (!ctx || !ctx->energy || ctx->mode != ASB_DEFLATE_PROJECT || !first_rejected_dev || ct < 0 || ct >= ctx->sub_ntile) return ASB_ERR_ARG;
    PanelState* st = sub_state(ctx, ct);
    const double* Wt = ctx->Wt3 + (size_t)ct * ctx->Fp * 16;
    const int pre = (ctx->pre_orth && ctx->correct_rows) ? 1 : 0;
    if (!pre)
        hipLaunchKernelGGL(k_panel_gram, dim3((unsigned)(kb + nc)), dim3(256), 0, ctx->stream, ctx->W, Wt, (int)ctx->Fp, ctx->gram,
                           ctx->gram_s, ctx->wn2t3 + 16 * ct);
    if (ctx->correct_rows) {
        long long cwr = (ctx->n_loc + 63) / 64;
        hipLaunchKernelGGL(k_correct_rows<true>, dim3((unsigned)(cwr < ctx->nblk_cap ? cwr : ctx->nblk_cap)), dim3(192), 0, ctx->stream,
                           ctx->comps, (long long)(3 * ctx->n_loc), (long long)ctx->n_loc, (int)kb, nc, ctx->gram_s, ctx->wn2t3 + 16 * ct,
                           ctx->energy, ctx->pmax, ctx->pidx, ctx->psum, ctx->colpart, st, ctx->scalar_dev, ctx->sel_e2, ctx->e_class, pre);
    } else {
        long long cw = (ctx->n_loc + 255) / 256;
        hipLaunchKernelGGL(k_correct<true>, dim3((unsigned)(cw < ctx->nblk_cap ? cw : ctx->nblk_cap)), dim3(256), 0, ctx->stream, ctx->comps,
                           (long long)(3 * ctx->n_loc), (long long)ctx->n_loc, (int)kb, nc, ctx->gram, ctx->wn2t3 + 16 * ct, ctx->energy,
                           ctx->pmax, ctx->pidx, ctx->psum, ctx->colpart, (const long long*)nullptr, (const PanelState*)nullptr,
                           (long long)0, st, ctx->scalar_dev, ctx->sel_e2, ctx->e_class);
    }
    hipLaunchKernelGGL(k_spec_count, dim3(1), dim3(1), 0, ctx->stream, st, nc, first_rejected_dev);
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}
extern "C" int asb_panel_sub_commit(asb_ctx* ctx, int ct, int64_t kb, int nc, int kept) {
    if (!ctx || !ctx->energy || ctx->mode != ASB_DEFLATE_PROJECT || ct < 0 || ct >= ctx->sub_ntile || kept < 0 || kept > nc) return ASB_ERR_ARG;
    PanelState* st = sub_state(ctx, ct);
    long long cw = (ctx->n_loc + 255) / 256;
    const int cgrid = (int)(cw < ctx->nblk_cap ? cw : ctx->nblk_cap);
    hipLaunchKernelGGL(k_commit_energy, dim3(cgrid), dim3(256), 0, ctx->stream, ctx->comps, (long long)(3 * ctx->n_loc),
                       (long long)ctx->n_loc, (int)kb, st, ctx->wn2t3 + 16 * ct, ctx->energy, ctx->pmax, ctx->pidx, ctx->psum,
                       ctx->colpart, kept);
    ctx->nblk = cgrid;
    hipLaunchKernelGGL(k_colsum, dim3(1), dim3(1024), 0, ctx->stream, ctx->colpart, ctx->nblk, kept, (long long)kb, ctx->scal,
                       (PanelState*)nullptr);
    ASB_CHECK_LAUNCH(ctx);
    if (kept > 0) ctx->k_done = kb + kept;
    ctx->n_spec_steps += nc;
    ctx->n_spec_kept += kept;
    return ASB_OK;
}

// ---- round 4: the multi-rank read in THREE calls and ONE exchange, level with the single-rank chain.
//   asb_panel_read_run     all sub-panels of the read in ONE launch of k_panel_multi on the ASSEMBLED candidates (identical on
//                          every rank, so the nine-word summary is too), the read's pass over this shard enqueued behind it on
//                          the expected column counts, the checks of all its tiles enqueued behind that with the LOCAL chain
//                          (a tile counts on this shard only behind tiles that stood in full on this shard); leaves
//                          words_dev[ct] = columns of tile ct that stand on this shard (nc[ct] for a tile not reached: neutral
//                          under min) and words_dev[ASB_MAX_SUB] = status (0, or -1: the launch's exchange timed out here)
//   -- the driver min-all-reduces words_dev (ASB_MAX_SUB + 1 doubles) over the ranks and reads it once --
//   asb_panel_read_commit  the verdict for all ranks: tiles stand in full while min == nc, the first one below keeps min columns,
//                          nothing behind it; where the local chain ran ahead of that (a tile stood here, not elsewhere) the
//                          energies go back to their values at the start of the read and the verdict is applied column by column
__global__ void k_read_words(const long long* __restrict__ res, int ntile, WideArgs wa, double status, double* __restrict__ words) {
    const int t = threadIdx.x;
    if (t < ASB_MAX_SUB) words[t] = t < ntile ? (res[t] < 0 ? (double)wa.nc[t] : (double)res[t]) : 0.0;
    if (t == ASB_MAX_SUB) words[t] = status;
    if (t == ASB_MAX_SUB + 1) {          // the local counts once more, packed (base 32), OUTSIDE the part the ranks reduce
        double p = 0.0, b = 1.0;
        for (int ct = 0; ct < ASB_MAX_SUB; ++ct, b *= 32.0) p += b * (ct < ntile ? (res[ct] < 0 ? (double)wa.nc[ct] : (double)res[ct]) : 0.0);
        words[t] = p;
    }
}
extern "C" int asb_panel_read_run(asb_ctx* ctx, int64_t k0, int64_t k1, int nsub_max, int spec_budget, const int* sub_budget,
                                  double* words_dev, int* ntile_out, int* nc_out, int* proven_out) {
    if (!ctx || !ctx->candR || !ctx->energy || ctx->mode != ASB_DEFLATE_PROJECT || !words_dev || !ntile_out || !nc_out || !proven_out)
        return ASB_ERR_ARG;
    if (k0 < 0 || k1 > ctx->K || k0 >= k1 || nsub_max < 1) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_panel_read_run: bad range");
    if (!(ctx->panel_coop && ctx->spec_panels && ctx->Fp <= 2048 && ctx->pre_orth && ctx->correct_rows))
        ASB_FAIL(ctx, ASB_ERR_ARG, "asb_panel_read_run needs the co-resident panel kernel (and F <= 2048)");
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->Wt3, (size_t)ASB_MAX_SUB * ctx->Fp * 16))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->Wq3, (size_t)ASB_MAX_SUB * ctx->Fp * 16))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->wn2t3, (size_t)16 * ASB_MAX_SUB))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->tile_counter, (size_t)16))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->pstate2, (size_t)ASB_MAX_SUB))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->e_class, (size_t)ctx->n_loc))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->tile_res, (size_t)ASB_MAX_SUB + 2))) return rc;
    ASB_HIP(ctx, hipMemcpyAsync(ctx->e_class, ctx->energy, (size_t)ctx->n_loc * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    const int save_budget = ctx->spec_budget;
    int save_sub[8];
    for (int q = 0; q < 8; ++q) { save_sub[q] = ctx->sub_budget[q]; if (sub_budget) ctx->sub_budget[q] = sub_budget[q]; }
    ctx->spec_budget = spec_budget;
    int ntile = -1, nc[ASB_MAX_SUB] = {0}, proven[ASB_MAX_SUB] = {0}, spec_ntile = 0, spec_nc[ASB_MAX_SUB] = {0};
    const int nlim = nsub_max > ASB_MAX_SUB ? ASB_MAX_SUB : nsub_max;
    rc = multi_chain_run(ctx, k0, k1, nlim, &ntile, nc, proven, &spec_ntile, spec_nc, true);
    ctx->spec_budget = save_budget;
    for (int q = 0; q < 8; ++q) ctx->sub_budget[q] = save_sub[q];
    if (rc) return rc;
    WideArgs wa{};
    double status = 0.0;
    if (ntile < 0 || ctx->chain_timed_out) {          // the exchange timed out on THIS rank: every rank must leave the kernel together
        status = -1.0;
        if (ntile < 0) ntile = 0;
        if (!ctx->chain_timed_out) {                   // (the launch could not be made at all)
            ctx->n_coop_fallbacks++;
        } else if (ctx->panel_coop) {
            ctx->panel_coop = 0;
            ctx->coop_test_stall = 0;
            ctx->n_coop_fallbacks++;
        }
        ntile = 0;
    }
    long long kb[ASB_MAX_SUB];
    PanelState* st[ASB_MAX_SUB];
    for (int ct = 0; ct < ASB_MAX_SUB; ++ct) {
        kb[ct] = k0 + (long long)ct * ASB_PANEL_COLS;
        st[ct] = ctx->pstate2 + ct;
        wa.kb[ct] = kb[ct];
        wa.nc[ct] = ct < ntile ? nc[ct] : 0;
    }
    ctx->rd_k0 = k0;
    ctx->rd_ntile = ntile;
    for (int ct = 0; ct < 8; ++ct) { ctx->rd_nc[ct] = ct < ntile ? nc[ct] : 0; ctx->rd_proven[ct] = ct < ntile ? proven[ct] : 0; }
    if (ntile > 0) {
        bool covered = spec_ntile >= ntile;
        for (int ct = 0; covered && ct < ntile; ++ct) covered = spec_nc[ct] >= nc[ct];
        if (!covered) {
            if ((rc = dbl_build_tiles(ctx, ntile, wa))) return rc;
            if ((rc = launch_wide(ctx, ntile, wa))) return rc;
        }
        int rgrid = 0, cgrid = 0;
        if ((rc = tiles_enqueue(ctx, ntile, kb, nc, st, &rgrid, &cgrid))) return rc;
        ctx->rd_rgrid = rgrid;
    }
    hipLaunchKernelGGL(k_read_words, dim3(1), dim3(64), 0, ctx->stream, ctx->tile_res, ntile, wa, status, words_dev);
    ASB_CHECK_LAUNCH(ctx);
    *ntile_out = ntile;
    for (int ct = 0; ct < ASB_MAX_SUB; ++ct) { nc_out[ct] = ct < ntile ? nc[ct] : 0; proven_out[ct] = ct < ntile ? proven[ct] : 0; }
    return ASB_OK;
}
// words (host, ASB_MAX_SUB + 2 doubles): [0, ASB_MAX_SUB] after the min over the ranks, [ASB_MAX_SUB + 1] this rank's own counts as
// asb_panel_read_run packed them.  *total = components the read commits (the same number on every rank).
extern "C" int asb_panel_read_commit(asb_ctx* ctx, const double* words, int64_t* total_out, int* full_out, int* rejected_out) {
    if (!ctx || !ctx->energy || ctx->mode != ASB_DEFLATE_PROJECT || !words || !total_out) return ASB_ERR_ARG;
    const int ntile = ctx->rd_ntile;
    const double* words_min = words;
    double words_local[ASB_MAX_SUB];
    {
        long long p = (long long)words[ASB_MAX_SUB + 1];
        for (int ct = 0; ct < ASB_MAX_SUB; ++ct, p /= 32) words_local[ct] = (double)(p % 32);
    }
    long long cw = (ctx->n_loc + 255) / 256;
    const int cgrid = (int)(cw < ctx->nblk_cap ? cw : ctx->nblk_cap);
    // the verdict
    int keep[ASB_MAX_SUB] = {0}, nstand = 0;             // keep[ct]: columns of tile ct that are committed; nstand: tiles that take part
    for (int ct = 0; ct < ntile; ++ct) {
        const int g = (int)words_min[ct];
        if (g < 0 || g > ctx->rd_nc[ct]) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_panel_read_commit: bad count %d for tile %d", g, ct);
        keep[ct] = g;
        nstand = ct + 1;
        if (g < ctx->rd_nc[ct]) break;
    }
    // what the local chain did: it adopted the tentative energies of tiles 0 .. l_full - 1 (stood in full HERE, consecutively)
    int l_full = 0;
    while (l_full < ntile && (int)words_local[l_full] == ctx->rd_nc[l_full]) {
        // (a tile "not reached" locally also reads nc: it lies behind a local failure, so the loop has stopped before it)
        ++l_full;
    }
    int g_full = 0;
    while (g_full < nstand && keep[g_full] == ctx->rd_nc[g_full]) ++g_full;
    int64_t total = 0;
    auto commit_cols = [&](int ct, int cols) {
        hipLaunchKernelGGL(k_commit_energy, dim3(cgrid), dim3(256), 0, ctx->stream, ctx->comps, (long long)(3 * ctx->n_loc),
                           (long long)ctx->n_loc, (int)(ctx->rd_k0 + 16 * ct), ctx->pstate2 + ct, ctx->wn2t3 + 16 * ct, ctx->energy, ctx->pmax,
                           ctx->pidx, ctx->psum, ctx->colpart, cols);
        ctx->nblk = cgrid;
        hipLaunchKernelGGL(k_colsum, dim3(1), dim3(1024), 0, ctx->stream, ctx->colpart, ctx->nblk, cols, (long long)(ctx->rd_k0 + 16 * ct),
                           ctx->scal, (PanelState*)nullptr);
    };
    if (l_full > g_full) {
        // this shard ran ahead of the verdict: back to the energies at the start of the read, then the verdict column by column
        ASB_HIP(ctx, hipMemcpyAsync(ctx->energy, ctx->e_class, (size_t)ctx->n_loc * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        for (int ct = 0; ct < nstand; ++ct)
            if (keep[ct] > 0) commit_cols(ct, keep[ct]);
    } else {
        // l_full == g_full (the minimum cannot stand where this shard did not): the full tiles are adopted, the partial one --
        // its energies untouched -- keeps its head
        ctx->nblk = ctx->rd_rgrid;
        if (g_full < nstand && keep[g_full] > 0) commit_cols(g_full, keep[g_full]);
    }
    ASB_CHECK_LAUNCH(ctx);
    bool rejected = false;
    int full = 0;
    for (int ct = 0; ct < nstand; ++ct) {
        total += keep[ct];
        ctx->n_spec_steps += ctx->rd_nc[ct] - ctx->rd_proven[ct];
        ctx->n_spec_kept += keep[ct] > ctx->rd_proven[ct] ? keep[ct] - ctx->rd_proven[ct] : 0;
        if (keep[ct] < ctx->rd_nc[ct]) rejected = true; else ++full;
    }
    if (total > 0) ctx->k_done = ctx->rd_k0 + total;
    *total_out = total;
    if (full_out) *full_out = full;
    if (rejected_out) *rejected_out = rejected ? 1 : 0;
    return ASB_OK;
}

// ---- guessed candidates of a first panel (see asb_project_run)
static bool guess_possible(const asb_ctx* ctx) {
    return ctx->first_panel_mean && ctx->spec_panels && ctx->panel_coop && ctx->Fp <= 2048 && ctx->EV && ctx->ev_valid && ctx->e0_valid &&
           ctx->n_energy_pass == 0 && ctx->m_target >= 256 && ctx->n_loc > ctx->m_cap;
}
// thresholds of the scores EV + g (E - EV) into sc[SC_TAUG ..]: about mq[q] / world of this shard's vertices above each.
// ~590 candidates on config 4 (the sets overlap), at most 830 + bin overshoot + the energies' own of the 1024 resident waves
static int guess_thresholds(asb_ctx* ctx, int world, bool with_energy) {
    // five shares g of the constant direction (what the first components leave behind: 0.107, 0.055, 0.028, 0.016 ... 0.004 on
    // config 4) and, round 3, the upper confidence bound on the cross terms: kappa = 1.35 is 2.6 standard deviations of the model
    // in in_guess's comment (sum over the steps of 4 alpha^2 beta^2 = 0.8).  CPU replay of six tensors (tools/sim_first_read.py):
    // first winner outside the union at step 42, 21, 41, 42, 57, 52 of 64 without it, 49-55, none, 53-none, none, 57, none with it
    static const double gq[ASB_NG] = {0.0, 0.02, 0.05, 0.12, 0.3, 0.0};
    static const double kq[ASB_NG] = {0.0, 0.0, 0.0, 0.0, 0.0, 1.35};
    static const int mq[ASB_NG] = {400, 140, 140, 90, 60, 400};
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->hist6, (size_t)ASB_NQ * ASB_NBINS))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->scm, (size_t)ASB_NQ * 8))) return rc;
    if (!ctx->hist6_clear) {          // every k_tau_multi leaves the bins it consumed at zero again
        ASB_HIP(ctx, hipMemsetAsync(ctx->hist6, 0, (size_t)ASB_NQ * ASB_NBINS * sizeof(int), ctx->stream));
        ctx->hist6_clear = true;
    }
    GuessTargets gt;
    for (int q = 0; q < ASB_NG; ++q) {
        long long m = mq[q] * ctx->m_target / 768 / world;
        if (m < 8) m = 8;
        gt.g[q] = gq[q];
        gt.h[q] = kq[q] / sqrt((double)ctx->F);
        gt.m_target[q] = m;
        gt.m_cap[q] = m + m / 8;
    }
    // g = 1: the energies proper (single rank: the few that carry the provable first steps; several ranks threshold them
    // through the usual exchange instead)
    const long long me = ctx->m_target / 12;
    gt.g[ASB_NG] = 1.0;
    gt.h[ASB_NG] = 0.0;
    gt.m_target[ASB_NG] = me;
    gt.m_cap[ASB_NG] = me + me / 2;
    // the diversity family (in_div; single rank): the guess fills ~740 of the 1024 candidate slots of the panel kernel; ~190
    // energy-weighted random vertices beside them cost the random tensor nothing (their rows ride along) and give a first read
    // on LOCALISED data weight vectors that span all its modes instead of the strongest one or two (the sketch of the next read)
    const long long md = (with_energy && diverse_on(ctx)) ? 190 * ctx->m_target / 768 : 0;
    gt.g[ASB_DIV_Q] = gt.h[ASB_DIV_Q] = 0.0;
    gt.m_target[ASB_DIV_Q] = md;
    gt.m_cap[ASB_DIV_Q] = md + md / 8;
    gt.div_seed = (double)(ctx->n_panels + ctx->n_refresh + 1);
    const int nq = with_energy ? ASB_NQ : ASB_NG;
    static const int ucb = getenv("ASB_GUESS_UCB") ? atoi(getenv("ASB_GUESS_UCB")) : 1;      // 0: without the confidence-bound family
    unsigned qmask = ucb ? 0xffffffffu : ~(1u << (ASB_NG - 1));
    if (md < 8) qmask &= ~(1u << ASB_DIV_Q);
    for (int level = 1; level <= 2; ++level) {
        hipLaunchKernelGGL(k_hist_multi, dim3(hist_grid(ctx)), dim3(256), 0, ctx->stream, ctx->energy, ctx->EV, (long long)ctx->n_loc,
                           ctx->scm, ctx->hist6, level == 1 ? 1 : 0, gt, nq, qmask);
        hipLaunchKernelGGL(k_tau_multi, dim3(nq), dim3(256), 0, ctx->stream, ctx->hist6, ctx->scm, ctx->scalar_dev, level, gt, qmask);
    }
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}
// candidates named by the sketch predictor (asb_sketch.hip): the 2/3 m_target largest scores united with the m_target / 3
// largest energies (which also carry the provable first steps) -- the guessed selection's machinery with one live score (g = 0: the score itself)
static int score_thresholds(asb_ctx* ctx, bool with_score = true, long long m_div = 0) {
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->hist6, (size_t)ASB_NQ * ASB_NBINS))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->scm, (size_t)ASB_NQ * 8))) return rc;
    if (!ctx->hist6_clear) {
        ASB_HIP(ctx, hipMemsetAsync(ctx->hist6, 0, (size_t)ASB_NQ * ASB_NBINS * sizeof(int), ctx->stream));
        ctx->hist6_clear = true;
    }
    GuessTargets gt;
    // (a third of the candidates by energy: the replay names who comes CLOSE to winning under its model, the energies who is large
    // now -- eight low-rank tensors: 16.0 ms / 6.25 reads in the mean with a third or a half by energy, 17.0 / 6.5 with a twelfth)
    static const int me_div = getenv("ASB_SKETCH_ME_DIV") ? atoi(getenv("ASB_SKETCH_ME_DIV")) : 3;
    // with_score: ms by the replay's scores, me by energy (+ m_div by the diversity family beside them); without: a PLAIN read
    // behind a rejection -- m_target - m_div by energy, m_div (half) energy-weighted random (in_div)
    const long long me = with_score ? ctx->m_target / (me_div > 1 ? me_div : 2) : ctx->m_target - m_div, ms = ctx->m_target - me;
    for (int q = 0; q < ASB_NQ; ++q) {
        gt.g[q] = q == ASB_NG ? 1.0 : 0.0;
        gt.h[q] = 0.0;
        gt.m_target[q] = q == ASB_NG ? me : (q == ASB_DIV_Q ? m_div : ms);
        gt.m_cap[q] = q == ASB_NG ? me + me / 2 : (q == ASB_DIV_Q ? m_div + m_div / 8 : ms + ms / 8);
    }
    gt.div_seed = (double)(ctx->n_panels + ctx->n_refresh + 1);
    const unsigned qmask = (with_score ? 1u : 0u) | (1u << ASB_NG) | (m_div >= 8 ? (1u << ASB_DIV_Q) : 0u);
    for (int level = 1; level <= 2; ++level) {
        hipLaunchKernelGGL(k_hist_multi, dim3(hist_grid(ctx)), dim3(256), 0, ctx->stream, ctx->energy,
                           with_score ? ctx->sk_score : (const double*)nullptr, (long long)ctx->n_loc, ctx->scm, ctx->hist6,
                           level == 1 ? 1 : 0, gt, ASB_NQ, qmask);
        hipLaunchKernelGGL(k_tau_multi, dim3(ASB_NQ), dim3(256), 0, ctx->stream, ctx->hist6, ctx->scm, ctx->scalar_dev, level, gt, qmask);
    }
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}
// multi-rank driver: what this shard contributes to the decision (the ranks sum both and compare), then begin / end
// around the first panel: begin installs the score thresholds and the smaller target for the energies proper, which
// asb_panel_tau / asb_panel_global_tau / asb_panel_target then use; asb_panel_select takes the union; the pass checks
// against it; end restores the plain selection.
extern "C" int asb_panel_guess_stats(asb_ctx* ctx, double* mean_energy_local, double* normx2_local, int* possible) {
    if (!ctx || !mean_energy_local || !normx2_local || !possible) return ASB_ERR_ARG;
    *possible = guess_possible(ctx) ? 1 : 0;
    *mean_energy_local = *possible ? ctx->mean_energy : 0.0;
    *normx2_local = *possible ? ctx->prep_normx2 : 0.0;
    return ASB_OK;
}
extern "C" int asb_panel_guess_begin(asb_ctx* ctx, int world) {
    if (!ctx || !ctx->energy || ctx->mode != ASB_DEFLATE_PROJECT || world < 1) return ASB_ERR_ARG;
    if (!guess_possible(ctx)) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_panel_guess_begin: no energies without the constant direction");
    int rc;
    if ((rc = guess_thresholds(ctx, world, false))) return rc;
    ctx->sel_e2 = ctx->EV;
    ctx->m_target_eff = ctx->m_target / 12;
    ctx->n_guess_panels++;
    return ASB_OK;
}
extern "C" int asb_panel_guess_end(asb_ctx* ctx) {
    if (!ctx) return ASB_ERR_ARG;
    ctx->sel_e2 = nullptr;
    ctx->m_target_eff = 0;
    return ASB_OK;
}

// candidate selection of the single-rank drivers: threshold + compaction + exact rows.
// First panel of a tensor whose energy sits largely in the constant-in-time direction (rest shape "first": every row
// carries its own offset): the first components remove that direction from EVERY vertex, after which the initial energies
// say nothing about who wins next, and a panel chosen by them alone ends after ~3 steps (config 4).  EV -- the energy
// without that direction, a by-product of the standardisation sweep -- is a good GUESS of the later ranking; the components
// do not remove the constant direction at once but leave a falling share g of it behind (0.1, 0.05, 0.04, 0.03 ... on
// config 4), so the ranking that matters at step t is that of EV + g_t (E - EV) with g_t unknown beforehand.  The
// candidates are { E > tau_E } (few: the provable first steps) united with the top vertices of that score for g on a
// geometric grid.  Nothing rests on the guess: steps beyond the provable ones are unproven steps, checked by the pass
// against every vertex outside the candidate set like any others.
static int panel_candidates(asb_ctx* ctx, long long k, int stalled) {
    int rc;
    const bool guess = k == 0 && stalled == 0 && ctx->mean_frac > 0.25 && guess_possible(ctx);
    const bool by_score = ctx->sketch_valid && stalled == 0 && k > 0;
    ctx->sketch_valid = false;
    ctx->read_by_score = by_score;
    // diversity (in_div): half of a plain read's candidates behind a rejection, an eighth beside the predicted ones
    const bool div_plain = diverse_on(ctx) && ctx->diverse_next && stalled == 0 && k > 0 && ctx->n_loc > 4 * ctx->m_cap;
    ctx->diverse_next = false;
    ctx->read_diverse = false;
    if (by_score) {
        if ((rc = score_thresholds(ctx, true, diverse_on(ctx) ? ctx->m_target / 8 : 0))) return rc;
        ctx->sel_e2 = ctx->sk_score;
        ctx->n_sketch_reads++;
    } else if (guess) {
        if ((rc = guess_thresholds(ctx, 1, true))) return rc;
        ctx->sel_e2 = ctx->EV;
        ctx->n_guess_panels++;
    } else if (div_plain) {
        if ((rc = score_thresholds(ctx, false, ctx->m_target / 2))) return rc;
        ctx->read_diverse = true;
        ctx->n_diverse_reads++;
    } else {
        hipLaunchKernelGGL(k_sc_set, dim3(1), dim3(1), 0, ctx->stream, ctx->scalar_dev, (int)SC_TAU_DIV, 1.0e300);
        for (int level = 1; level <= 2; ++level) {
            if ((rc = asb_panel_hist(ctx, level, nullptr))) return rc;
            if ((rc = asb_panel_tau(ctx, level, nullptr))) return rc;
        }
    }
    return asb_panel_select(ctx, k, -1, 0, nullptr, nullptr, nullptr, nullptr);
}

// ---- leaving the projection mode in mid-run (single rank: asb_project_run's stall rule; several ranks: the driver's, which then
// continues with the residual protocol): the residual R_k = X - sum_{j<k} w_j (x) c_j is written out once (one read of X, one
// write of R, the exact energies and their partial records with it), the per-component scalars change from the projection
// mode's |w|^2 |c|^2 to the residual mode's |R_j|^2, and the context IS in residual mode from here on: asb_deflate_pick /
// _apply / _local_best / _results work as if the run had begun there.
__global__ void k_scal_to_direct(double* __restrict__ scal, long long k, const double* __restrict__ sc) {
    double r2 = sc[SC_NORMX2];
    for (long long j = 0; j < k; ++j) {
        r2 -= scal[j * 4 + 3];
        scal[j * 4 + 3] = r2;
    }
}
extern "C" int asb_project_switch_residual(asb_ctx* ctx, int64_t k) {
    if (!ctx || !ctx->energy || !ctx->comps || !ctx->W) return ASB_ERR_ARG;
    if (ctx->mode != ASB_DEFLATE_PROJECT) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_project_switch_residual: not in projection mode");
    if (k < 0 || k > ctx->K) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_project_switch_residual: k = %lld out of range", (long long)k);
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->R, (size_t)ctx->n_loc * 3 * ctx->Fp))) return rc;
    if (!pick_cfg(ctx->Fp, ctx->cfg)) ASB_FAIL(ctx, ASB_ERR_LIMIT, "F too large");
    const int ggrid = stream_grid(ctx, ctx->cfg, ctx->n_loc);
    launch_gather(ctx, ctx->cfg, ggrid, nullptr, (long long)ctx->n_loc, nullptr, (int)k, ctx->R, ctx->energy, ctx->pmax, ctx->pidx,
                  ctx->psum);
    ASB_CHECK_LAUNCH(ctx);
    ctx->nblk = ggrid;
    ctx->n_refresh++;                                   // (a read of X)
    hipLaunchKernelGGL(k_scal_to_direct, dim3(1), dim3(1), 0, ctx->stream, ctx->scal, (long long)k, ctx->scalar_dev);
    ASB_CHECK_LAUNCH(ctx);
    ctx->mode = ASB_DEFLATE_RESIDUAL;
    ctx->local = 0;
    ctx->forced_row = -1;
    ctx->sel_e2 = nullptr;
    ctx->k_done = k;
    ctx->k_switch = k;
    if (getenv("ASB_DEBUG_PANELS"))
        fprintf(stderr, "[asb] k=%lld: %lld reads of X so far -- the run continues in the residual loop\n", (long long)k,
                (long long)(ctx->n_panels + ctx->n_refresh));
    return ASB_OK;
}
extern "C" int asb_deflate_switch_stats(asb_ctx* ctx, int64_t* k_switch) {
    if (!ctx || !k_switch) return ASB_ERR_ARG;
    *k_switch = ctx->k_switch;
    return ASB_OK;
}
// fewer than 3/4 of a component per read of X over the last >= 8 reads (refreshes included)?  A low-yield read costs ~1.5 ms at
// config 4's size (one sub-panel's pass + its greedy steps + selection), a refresh ~1 ms, a residual step (one read + one write
// of R) ~1.9 ms: below that rate the residual loop is cheaper, and far below it -- K beyond the numerical rank: ~5 reads per
// component -- it is the only sane way on.
static bool stall_rule(asb_ctx* ctx, long long k) {
    if (!ctx->stall_fallback) return false;
    const long long reads = ctx->n_panels + ctx->n_refresh;
    if (reads - ctx->fb_mark_reads < 8) return false;
    const bool slow = (k - ctx->fb_mark_k) * 4 < (reads - ctx->fb_mark_reads) * 3;
    ctx->fb_mark_reads = reads;
    ctx->fb_mark_k = k;
    return slow;
}
extern "C" int asb_deflate_pick(asb_ctx* ctx, int64_t k, const double* recs_dev, int64_t n_rec);
extern "C" int asb_deflate_apply(asb_ctx* ctx, int64_t k, const double* s);

int asb_project_run(asb_ctx* ctx, int64_t k0, int64_t k1) {
    int rc;
    long long k = k0;
    int stalled = 0;
    const int global_all = ctx->n_loc <= ctx->m_cap;
    const bool use_double = ctx->double_panels && !global_all && ctx->panel_coop && ctx->Fp <= 2048 && ctx->spec_panels;
    while (k < k1) {
        ctx->sel_e2 = nullptr;
        // rows below k are final (projection mode never rewrites a committed column): their copy to the pinned buffer runs
        // on the copy stream while the next read of X computes (asb_components_stream; no-op otherwise)
        if ((rc = asb_dl_enqueue(ctx, k))) return rc;
        if (stall_rule(ctx, k)) {
            if ((rc = asb_project_switch_residual(ctx, k))) return rc;
            for (long long kk = k; kk < k1; ++kk) {
                if ((rc = asb_deflate_pick(ctx, kk, nullptr, 0))) return rc;
                if ((rc = asb_deflate_apply(ctx, kk, nullptr))) return rc;
            }
            return ASB_OK;
        }
        if (use_double && stalled == 0) {
            int64_t done = 0;
            if ((rc = double_panel(ctx, k, k1, &done))) return rc;
            if (done > 0) { k += done; continue; }
            stalled = 1;                                 // nothing stood: exact energies, then the plain path below
            if ((rc = asb_panel_refresh(ctx, k, nullptr, nullptr))) return rc;
            continue;
        }
        int64_t forced = -1;
        if (stalled >= 2) {        // massive exact ties: the first arg-max of the (exact) energies alone
            double be;
            hipLaunchKernelGGL(k_best_energy, dim3(1), dim3(256), 0, ctx->stream, ctx->pmax, ctx->pidx, ctx->psum, ctx->nblk,
                               ctx->scalar_dev + 8);
            double h[3];
            ASB_HIP(ctx, hipMemcpyAsync(h, ctx->scalar_dev + 8, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
            ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
            long long li;
            memcpy(&li, &h[1], 8);
            forced = ctx->v0 + li;
            (void)be;
        } else if (!global_all) {
            if ((rc = panel_candidates(ctx, k, stalled))) return rc;
        } else if ((rc = asb_panel_select(ctx, k, forced, 1, nullptr, nullptr, nullptr, nullptr))) return rc;
        if (forced >= 0 && (rc = asb_panel_select(ctx, k, forced, 1, nullptr, nullptr, nullptr, nullptr))) return rc;
        const int steps = forced >= 0 ? 1 : (int)((k1 - k) < ASB_PANEL_COLS ? (k1 - k) : ASB_PANEL_COLS);
        int64_t done = 0;
        // unproven steps only on the plain path: a stalled panel is repeated with provable steps alone
        ctx->run_spec_max = (ctx->spec_panels && stalled == 0 && forced < 0 && !global_all) ? ctx->spec_budget : 0;
        rc = asb_panel_run(ctx, k, steps, forced >= 0 ? 1 : global_all, 0, &done);
        ctx->run_spec_max = 0;
        if (rc) return rc;
        if (done > ctx->run_proven) {          // the tail of the panel is unproven: the pass decides how much of it stands
            const int64_t proven = ctx->run_proven, tried = done - proven;
            if ((rc = project_pass(ctx, k, (int)done, (int)proven, &done))) return rc;
            const int64_t gain = done - proven;
            ctx->n_spec_steps += tried;
            ctx->n_spec_kept += gain;
            // a kept step saves 1/16 of a pass over X (~80 us on config 4), a rejected one costs one panel step (~14 us):
            // keep trying the full panel as long as anything stands, back off only after complete failures
            ctx->spec_budget = gain > 0 ? ASB_PANEL_COLS : (ctx->spec_budget / 2 > 2 ? ctx->spec_budget / 2 : 2);
            if (getenv("ASB_DEBUG_PANELS"))
                fprintf(stderr, "[asb] panel at k=%lld: %lld proven + %lld of %lld unproven steps kept\n", k, (long long)proven,
                        (long long)gain, (long long)tried);
            if (done > 0) {
                stalled = 0;
                ctx->k_done = k + done;
                k += done;
                continue;
            }
            // nothing stood (the very first step was already wrong): energies are untouched, fall through to the refresh
        }
        if (done == 0) {
            // the energy recurrence could not prove any candidate: refresh ALL energies exactly and retry;
            // a second failure (massive exact ties) forces the first arg-max as the only candidate
            if (++stalled > 3) ASB_FAIL(ctx, ASB_ERR_NUMERIC, "deflation made no progress at component %lld", k);
            if ((rc = asb_panel_refresh(ctx, k, nullptr, nullptr))) return rc;
            continue;
        }
        stalled = 0;
        if ((rc = asb_panel_project(ctx, k, (int)done))) return rc;
        k += done;
    }
    ctx->sel_e2 = nullptr;
    return ASB_OK;
}

// weigs (F, K) = W^T (W: K rows of Fp frames)
__global__ __launch_bounds__(256) void k_weights_fk(const double* __restrict__ W, long long Fp, long long F, long long K,
                                                    double* __restrict__ out) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < F * K; e += (long long)gridDim.x * 256)
        out[e] = W[(e % K) * Fp + e / K];
}

// the (K + 1) x 4 scalars of a run and the range scalars into a pinned host buffer of the context, its sequence word last
__global__ __launch_bounds__(256) void k_publish_results(const double* __restrict__ scal, long long n_scal, const double* __restrict__ sc,
                                                         double* __restrict__ pin, unsigned long long seq) {
    for (long long i = threadIdx.x; i < n_scal; i += 256) pin[2 + i] = scal[i];
    if (threadIdx.x < 8) pin[2 + n_scal + threadIdx.x] = sc[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(pin), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int asb_project_results(asb_ctx* ctx, double* comps, double* weigs, int64_t* idx, double* sigma, double* normR2_local) {
    const int64_t K = ctx->K;
    const size_t n_scal = (size_t)(K + 1) * 4;
    std::vector<double> h(n_scal);
    double sc[8];
    // one-thread-per-word kernel into mapped pinned memory + a polled sequence word: no copy call (each blocks the host for its
    // own round trip when the destination is pageable) -- the weights below are the only copy of a run's results
    bool published = false;
    unsigned long long seq = 0;
    if (ctx->host_poll) {
        if (ctx->res_pin_count < n_scal + 16) {
            if (ctx->res_pin) (void)hipHostFree(ctx->res_pin);
            ctx->res_pin = nullptr;
            ctx->res_pin_dev = nullptr;
            if (hipHostMalloc((void**)&ctx->res_pin, (n_scal + 16) * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess) {
                ctx->res_pin_count = n_scal + 16;
                ctx->res_pin[0] = 0.0;
                if (hipHostGetDevicePointer((void**)&ctx->res_pin_dev, ctx->res_pin, 0) != hipSuccess) ctx->res_pin_dev = nullptr;
            } else {
                (void)hipGetLastError();
                ctx->res_pin = nullptr;
                ctx->res_pin_count = 0;
            }
        }
        if (ctx->res_pin_dev) {
            seq = ++ctx->pin_seq;
            hipLaunchKernelGGL(k_publish_results, dim3(1), dim3(256), 0, ctx->stream, ctx->scal, (long long)n_scal, ctx->scalar_dev,
                               ctx->res_pin_dev, seq);
            ASB_CHECK_LAUNCH(ctx);
            published = true;
        }
    }
    if (!published) {
        ASB_HIP(ctx, hipMemcpyAsync(h.data(), ctx->scal, h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        ASB_HIP(ctx, hipMemcpyAsync(sc, ctx->scalar_dev, sizeof(sc), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (comps)
        ASB_HIP(ctx, hipMemcpyAsync(comps, ctx->comps, (size_t)K * 3 * ctx->n_loc * sizeof(double), hipMemcpyDeviceToHost,
                                    ctx->stream));
    if (weigs) {        // the reference's (F, K) order is produced on the device: one copy straight into the caller's array
        int rc = asb_alloc(ctx, &ctx->w_fk, (size_t)ctx->F * K);
        if (rc) return rc;
        const long long total = (long long)ctx->F * K;
        hipLaunchKernelGGL(k_weights_fk, dim3((unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024)), dim3(256), 0, ctx->stream,
                           ctx->W, (long long)ctx->Fp, (long long)ctx->F, (long long)K, ctx->w_fk);
        ASB_CHECK_LAUNCH(ctx);
        ASB_HIP(ctx, hipMemcpyAsync(weigs, ctx->w_fk, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (published) {
        volatile unsigned long long* word = reinterpret_cast<volatile unsigned long long*>(ctx->res_pin);
        const auto t0 = std::chrono::steady_clock::now();
        bool arrived = false;
        for (unsigned spins = 0;; ++spins) {
            if (*word == seq) { arrived = true; break; }
            if ((spins & 1023) == 1023 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 4.0) break;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (!arrived || comps || weigs) ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));      // (the copies above; a fault shows here)
        memcpy(h.data(), ctx->res_pin + 2, n_scal * sizeof(double));
        memcpy(sc, ctx->res_pin + 2 + n_scal, sizeof(sc));
    } else {
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    double r2 = sc[SC_NORMX2];
    for (int64_t k = 0; k < K; ++k) {
        if (sigma) sigma[k] = h[k * 4 + 0];
        if (idx) memcpy(&idx[k], &h[k * 4 + 2], 8);
        r2 -= h[k * 4 + 3];                       // |R_k|^2 = |R_{k-1}|^2 - |w_k|^2 |c_k|^2
        if (normR2_local) normR2_local[k] = r2;
    }
    return ASB_OK;
}
