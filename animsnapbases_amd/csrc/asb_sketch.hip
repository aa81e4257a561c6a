// Predicted candidates for the panel algorithm on STRUCTURED data ("sketch predictor").
//
// The problem.  A read of X commits the greedy steps (posComponents.py:76-96: arg-max of the residual energies, top right
// singular vector of the winner's 3 x F slab, deflation of every vertex) that the pass can verify; on noise-like data the
// candidates chosen by their energies at the start of the read contain the next 64 winners, on structured data (every
// component removes a direction ALL vertices share) they contain about five -- the whole ranking reshuffles (16 reads
// of X for K = 128 on bench.py's low-rank leg against 2 on the random tensor).
//
// The sketch.  The pass of such a read still computed, for EVERY vertex, the coefficients of all its 64 columns, and the
// columns of the rejected steps are projections of the residual on 64 - kept orthogonal directions w_t that span its
// dominant frame subspace (each is the top singular direction of a high-energy residual slab).  Z_v = (|w_t| c_t[v])_t
// (3 x r per vertex) is therefore a rank-r sketch of the residual, R_v = Z_v What^T + T_v, for free.  The greedy loop is
// replayed IN SKETCH SPACE for all vertices at once: energy e_v = |Z_v|^2 + tail_v (tail_v = E_v - |Z_v|^2 from the exact
// energies), winner = arg-max, its slab's Gram matrix Z_v^T Z_v + (tail_v / 3) I (the tail taken as isotropic: that is
// what leaves the measured share g = (1/3) / (|a|^2 + 1/3) of a shared direction behind on noise), top
// eigen-pair (lambda, u), q = Z_v u, and for every vertex Z_v -= q (q^T Z_v) / lambda -- the reference's deflation with
// every cross term between a vertex's tail and the winner's direction dropped.  Nothing rests on the replay: it only NAMES
// the next read's candidates (score_v = max over the steps of e_v / e_winner: the vertices that come closest to being
// selected), the panel kernel then runs on their exact rows and the pass checks every step against every vertex outside
// the set, exactly as for candidates chosen any other way.
//
// The kernel.  One thread per vertex holds its 3 x 64 sketch in registers as f32 (192 VGPRs; 512-thread blocks, one per CU:
// 131 072 vertices on 256 CUs); a step is the panel kernel's exchange in small: block best -> 16-byte record -> every block
// reduces all records -> the winner's thread, which solved its 3 x 3 eigen-problem and formed q while the records
// travelled, publishes q (33 words) -> every thread deflates its own sketch (2 x 192 FMAs) and re-sums its Gram matrix
// from the sketch itself (f32 rank-one updates would lose the energies once they have fallen by 1e6).  Words that cross
// blocks are self-validating and reset two steps before they are written again (k_panel_multi's protocol); a poll that
// does not complete raises the abort flag, and an aborted launch leaves score = energy: the selection then is the plain
// one.  Single rank (the multi-rank driver keeps the energy ranking).
#include "asb_common.h"
#include "asb_kernels.h"

#ifndef SK_R
#define SK_R 64                   // sketch dimensions (columns of a read)
#endif
#define SK_T 512                  // threads per block = vertices per block
#define SK_NP (SK_R / 2)           // pairs of f32 in a q message
#define SK_QW (SK_NP + 2)         // its words: the pairs, lambda (f64), pad
#define SK_SENT 0xFFFFFFFFFFFFFFFFull
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned long long sk_load(const unsigned long long* p) {
    return __hip_atomic_load(const_cast<unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sk_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int CTRL>
__device__ __forceinline__ unsigned sk_dpp(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ unsigned wave_max_u32_dpp(unsigned v) {
    v = max(v, sk_dpp<0xB1>(v));
    v = max(v, sk_dpp<0x4E>(v));
    v = max(v, sk_dpp<0x141>(v));
    v = max(v, sk_dpp<0x140>(v));
    auto s = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = max((unsigned)s[0], (unsigned)s[1]);
    s = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return max((unsigned)s[0], (unsigned)s[1]);
}
template <int CTRL>
__device__ __forceinline__ unsigned long long sk_dpp64(unsigned long long v) {
    return ((unsigned long long)sk_dpp<CTRL>((unsigned)(v >> 32)) << 32) | sk_dpp<CTRL>((unsigned)v);
}
__device__ __forceinline__ unsigned long long wave_max_u64_dpp(unsigned long long v) {
    unsigned long long o;
    o = sk_dpp64<0xB1>(v); v = o > v ? o : v;
    o = sk_dpp64<0x4E>(v); v = o > v ? o : v;
    o = sk_dpp64<0x141>(v); v = o > v ? o : v;
    o = sk_dpp64<0x140>(v); v = o > v ? o : v;
    {
        const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
        const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        const unsigned long long a = ((unsigned long long)h[0] << 32) | l[0], b = ((unsigned long long)h[1] << 32) | l[1];
        v = a > b ? a : b;
    }
    {
        const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
        const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        const unsigned long long a = ((unsigned long long)h[0] << 32) | l[0], b = ((unsigned long long)h[1] << 32) | l[1];
        v = a > b ? a : b;
    }
    return v;
}
__device__ __forceinline__ bool tl0(unsigned b, int t) { return b == 0 && t == 0; }
// out of line: one lane per block runs it per step, and inlined its ~60 registers of f64 temporaries would come on top of the
// 192 the sketch occupies everywhere
__device__ __attribute__((noinline)) void sk_eig3(const float* g, float t3f, double* out4) {
    const double t3 = (double)t3f;
    double lam, u0, u1, u2;
    eig3_top_fast((double)g[0] + t3, (double)g[1], (double)g[2], (double)g[3] + t3, (double)g[4], (double)g[5] + t3, lam, u0, u1, u2);
    out4[0] = lam; out4[1] = u0; out4[2] = u1; out4[3] = u2;
}

// cols: column i of the sketch source at cols + i * stride, entry 3 v + d; wn2[i * wn2_stride] = |w_i|^2.
// words: records 3 x G, then q messages 3 x SK_QW.  flags[1]: abort, flags[2]: ran to the end.
__global__ __launch_bounds__(SK_T, 2) void k_sketch_greedy(const double* __restrict__ cols, long long stride,
                                                           const double* __restrict__ wn2, int wn2_stride,
                                                           const double* __restrict__ E, long long n, int r, int steps,
                                                           double* __restrict__ score_out, long long* __restrict__ pred_out,
                                                           unsigned long long* words, unsigned* flags, int test_stall, float min_share,
                                                           unsigned* counts, const int* __restrict__ map = nullptr) {
    // map (optional): the replay runs on a SUBSET of the vertices -- slot s of the launch holds vertex map[s], n = subset size, in
    // increasing vertex order (so "lowest slot" is "lowest vertex"); cols / E / score_out stay indexed by the vertex
    __shared__ __attribute__((aligned(16))) float q_sh[SK_R];
    __shared__ double lam_sh;
    __shared__ float zs[4][64];
    __shared__ float sw[SK_R];
    __shared__ unsigned wv_e[8];
    __shared__ int wv_l[8];
    __shared__ unsigned long long sh_k[8];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int G = gridDim.x;
    const long long slot = (long long)blockIdx.x * SK_T + tid;
    const bool have = slot < n;
    const long long v = have ? (map ? (long long)map[slot] : slot) : 0;
    unsigned long long* rec = words;
    unsigned long long* qbuf = words + (size_t)3 * G;
    const double e_in = have ? E[v] : 0.0;
    if (__hip_atomic_load(flags + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {      // scheduled after the others gave up
        if (have) score_out[v] = e_in;
        return;
    }
    if (tid < SK_R) sw[tid] = tid < r ? (float)sqrt(fmax(wn2[(long long)tid * wn2_stride], 0.0)) : 0.0f;
    __syncthreads();
    f2 z[3][SK_R / 2];
    {
        // (no predicated loads: columns beyond r re-read the last one and are scaled by sw = 0; threads beyond n read vertex 0)
        const double* p0 = cols + 3 * (have ? v : 0);
        const float hm = have ? 1.0f : 0.0f;
#pragma unroll
        for (int i0 = 0; i0 < SK_R; i0 += 4) {         // four columns (12 loads) in flight
            double c[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ie = i0 + u < r ? i0 + u : r - 1;
                const double* p = p0 + (long long)ie * stride;
#pragma unroll
                for (int d = 0; d < 3; ++d) c[u][d] = p[d];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float s = sw[i0 + u] * hm;
#pragma unroll
                for (int d = 0; d < 3; ++d) z[d][(i0 + u) / 2][(i0 + u) & 1] = (float)c[u][d] * s;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float g[6];
    auto gram = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 6; ++q) g[q] = 0.0f;
#pragma unroll
        for (int i = 0; i < SK_R; ++i) {
            const float a = z[0][i / 2][i & 1], b = z[1][i / 2][i & 1], c = z[2][i / 2][i & 1];
            g[0] += a * a; g[1] += a * b; g[2] += a * c; g[3] += b * b; g[4] += b * c; g[5] += c * c;
        }
    };
    gram();
    // the part of the exact energy the sketch does not hold.  |Z_v|^2 is summed in f64 HERE, once (round 4): as an f32 sum it is
    // good to 1e-7 of the energy, which is 10 % of a tail of 1e-6 -- and on structured data (a sketch that holds 0.999 of the
    // residual) the tails are what ranks the vertices once the sketched energies have fallen to them
    double own64 = 0.0;
#pragma unroll
    for (int i = 0; i < SK_R; ++i) {
        const double a = (double)z[0][i / 2][i & 1], b = (double)z[1][i / 2][i & 1], c = (double)z[2][i / 2][i & 1];
        own64 += a * a + b * b + c * c;
    }
    float tail = (float)fmax(e_in - own64, 0.0);
    float e = (g[0] + g[3]) + g[5] + tail;
    float score = 0.0f;
    const long long spin_max = test_stall ? (1LL << 10) : (1LL << 20);
    bool aborted = false;
    // ---- is a replay worth its 0.7 ms?  Only if the sketch HOLDS the residual: its share of the energy, sum |Z_v|^2 / sum E_v, is
    // near 1 on structured data and a few per cent on noise (r of F directions), where candidates by energy do as well (measured on
    // eight random tensors: 6.6 ms per step without replays, 7.0 with).  One exchange: every block publishes its two sums, all
    // blocks add them in the same order and take the same decision; below min_share the launch leaves score = energy.
    {
        __shared__ double gsum[2][256];
        __shared__ double gate[2];
        double v2[2] = {have ? (double)((g[0] + g[3]) + g[5]) : 0.0, have ? e_in : 0.0};
        block_sum<2>(v2, &gsum[0][0]);
        __syncthreads();
        unsigned long long* grec = qbuf + (size_t)3 * SK_QW;           // 2 G words behind the q messages
        if (tid == 0 && !(test_stall && (int)blockIdx.x == G - 1)) {
            sk_store(grec + 2 * blockIdx.x, (unsigned long long)__double_as_longlong(v2[0]));
            sk_store(grec + 2 * blockIdx.x + 1, (unsigned long long)__double_as_longlong(v2[1]));
        }
        int dead = 0;
        for (int b = tid; b < G; b += SK_T) {
            unsigned long long w0, w1;
            long long spins = 0;
            for (;;) {
                w0 = sk_load(grec + 2 * b);
                w1 = sk_load(grec + 2 * b + 1);
                if (w0 != SK_SENT && w1 != SK_SENT) break;
                if ((++spins & 63) == 0 &&
                    (spins > spin_max || __hip_atomic_load(flags + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { dead = 1; break; }
            }
            if (dead) break;
            gsum[0][b] = __longlong_as_double((long long)w0);
            gsum[1][b] = __longlong_as_double((long long)w1);
        }
        if (__syncthreads_or(dead)) {
            if (tid == 0) __hip_atomic_store(flags + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (have) score_out[v] = e_in;
            return;
        }
        if (tid == 0) {
            double a = 0.0, b2 = 0.0;
            for (int b = 0; b < G; ++b) { a += gsum[0][b]; b2 += gsum[1][b]; }
            gate[0] = a; gate[1] = b2;
        }
        __syncthreads();
        if (tl0(blockIdx.x, tid)) flags[0] = __float_as_uint((float)(gate[1] > 0.0 ? gate[0] / gate[1] : 0.0));      // (diagnostics)
        if (!(gate[0] >= (double)min_share * gate[1])) {
            if (have) score_out[v] = e_in;
            if (tl0(blockIdx.x, tid)) { __hip_atomic_store(flags + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicAdd(counts + 1, 1u); }
            return;
        }
        if (tl0(blockIdx.x, tid)) atomicAdd(counts + 0, 1u);
    }
    unsigned long long* tlog = reinterpret_cast<unsigned long long*>(flags + 4);      // [64][6] timestamps of block 0 (debug)
    const bool tl = blockIdx.x == 0 && tid == 0;
    // a record is ONE word: (bits of the energy, an f32 >= 0: ordered like the value) << 32 | (2^32 - 1 - vertex): the largest
    // key is the largest energy and among equals the lowest vertex (NumPy's first max); all ones (a NaN) never occurs
    const unsigned vkey = 0xFFFFFFFFu - (unsigned)slot;
    for (int t = 0; t < steps; ++t) {
        const int ring = t % 3, ring_prev = (t + 2) % 3;
        if (tl) tlog[t * 6 + 0] = wall_clock64();
        // ---- 1. block best; the record goes out at once
        const unsigned eb = have ? __float_as_uint(fmaxf(e, 0.0f)) : 0u;
        const unsigned em = wave_max_u32_dpp(eb);
        const int bl = wave_min_dpp((have && eb == em) ? lane : 64);
        if (lane == 0) { wv_e[wv] = em; wv_l[wv] = bl; }
        __builtin_amdgcn_s_waitcnt(0);                  // this wave's resets of the last step are acknowledged
        __syncthreads();
        int ow = 0;
#pragma unroll
        for (int q = 1; q < 8; ++q)
            if (wv_e[q] > wv_e[ow]) ow = q;
        const int bt = ow * 64 + (wv_l[ow] & 63);       // the block's best thread (lowest vertex among equals)
        const unsigned bE = wv_e[ow];
        if (tid == 0 && !(test_stall && (int)blockIdx.x == G - 1)) {
            const unsigned long long key = bE ? (((unsigned long long)bE << 32) | (0xFFFFFFFFu - (unsigned)(blockIdx.x * SK_T + bt))) : 0ull;
            sk_store(rec + (size_t)ring * G + blockIdx.x, key);
        }
        if (tl) tlog[t * 6 + 1] = wall_clock64();
        // ---- 2. the best thread's Gram matrix, eigen-pair and q while the records travel; the other half of the block polls
        float my_u[3] = {0.0f, 0.0f, 0.0f};
        unsigned long long bk = 0ull;
        int dead = 0;
        const int pb = ow < 4 ? 256 : 0;
        if (wv == ow) {
            if (bE) {
                // the best thread's slab goes through LDS and ITS WAVE does the rest together: lane i owns sketch dimension i --
                // Gram matrix by a wave sum, eigen-pair (every lane the same), q_i = u . z_i (LDS is in order within a wave)
                if (tid == bt) {
#pragma unroll
                    for (int i = 0; i < SK_R; ++i) { zs[0][i] = z[0][i / 2][i & 1]; zs[1][i] = z[1][i / 2][i & 1]; zs[2][i] = z[2][i / 2][i & 1]; }
                    zs[3][0] = tail;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const float a = lane < SK_R ? zs[0][lane] : 0.0f, b = lane < SK_R ? zs[1][lane] : 0.0f, c = lane < SK_R ? zs[2][lane] : 0.0f;
                double gq[6] = {(double)(a * a), (double)(a * b), (double)(a * c), (double)(b * b), (double)(b * c), (double)(c * c)};
                wave_sum_dpp<6>(gq);
                float gl[6] = {(float)gq[0], (float)gq[1], (float)gq[2], (float)gq[3], (float)gq[4], (float)gq[5]};
                double o4[4];
                sk_eig3(gl, zs[3][0] / 3.0f, o4);
                my_u[0] = (float)o4[1]; my_u[1] = (float)o4[2]; my_u[2] = (float)o4[3];
                if (lane < SK_R) q_sh[lane] = my_u[0] * a + my_u[1] * b + my_u[2] * c;
                if (lane == 0) lam_sh = o4[0];
            }
        } else if (tid >= pb && tid < pb + 256) {
            const unsigned long long* recs = rec + (size_t)ring * G;
            for (int b = tid - pb; b < G; b += 256) {
                unsigned long long w;
                long long spins = 0;
                for (;;) {
                    w = sk_load(recs + b);
                    if (w != SK_SENT) break;
                    if ((++spins & 63) == 0 &&
                        (spins > spin_max || __hip_atomic_load(flags + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { dead = 1; break; }
                }
                if (dead) break;
                bk = w > bk ? w : bk;
            }
        }
        bk = wave_max_u64_dpp(bk);
        if (lane == 0) sh_k[wv] = bk;
        if (__syncthreads_or(dead)) { aborted = true; break; }
        bk = sh_k[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) bk = sh_k[q] > bk ? sh_k[q] : bk;
        if (tl) tlog[t * 6 + 2] = wall_clock64();
        // every block is past step t - 1: its words can be reset (k_panel_multi's argument)
        if (tid == 0) sk_store(rec + (size_t)ring_prev * G + blockIdx.x, SK_SENT);
        if (blockIdx.x == 0 && tid >= 64 && tid < 64 + SK_QW) sk_store(qbuf + (size_t)ring_prev * SK_QW + (tid - 64), SK_SENT);
        if (bk == 0ull) break;                           // nothing left to select (every block sees the same records)
        const float be = __uint_as_float((unsigned)(bk >> 32));
        const bool mine = have && (unsigned)(bk & 0xffffffffull) == vkey;      // (this thread is the winner)
        const int gb = (int)((0xFFFFFFFFu - (unsigned)(bk & 0xffffffffull)) / SK_T);
        if (have) score = fmaxf(score, e / be);
        if (tl && pred_out) {
            const long long ws = (long long)(0xFFFFFFFFu - (unsigned)(bk & 0xffffffffull));
            pred_out[t] = map ? (long long)map[ws] : ws;
        }
        // ---- 3. the winner's q: its block publishes it, the others spin on it
        unsigned long long* qb = qbuf + (size_t)ring * SK_QW;
        dead = 0;
        if (gb == (int)blockIdx.x) {
            if (tid < SK_NP) {
                const unsigned lo = __float_as_uint(q_sh[2 * tid]), hi = __float_as_uint(q_sh[2 * tid + 1]);
                sk_store(qb + tid, ((unsigned long long)hi << 32) | lo);
            } else if (tid == SK_NP) {
                sk_store(qb + SK_NP, (unsigned long long)__double_as_longlong(lam_sh));
            }
        } else if (tid <= SK_NP) {
            unsigned long long w;
            long long spins = 0;
            for (;;) {
                w = sk_load(qb + tid);
                if (w != SK_SENT) break;
                if ((++spins & 63) == 0 &&
                    (spins > spin_max || __hip_atomic_load(flags + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { dead = 1; break; }
            }
            if (tid < SK_NP) {
                q_sh[2 * tid] = __uint_as_float((unsigned)(w & 0xffffffffull));
                q_sh[2 * tid + 1] = __uint_as_float((unsigned)(w >> 32));
            } else {
                lam_sh = __longlong_as_double((long long)w);
            }
        }
        if (__syncthreads_or(dead)) { aborted = true; break; }
        if (tl) tlog[t * 6 + 3] = wall_clock64();
        // ---- 4. every vertex's sketch is deflated; the winner's as a whole (R' = (I - u u^T) R: sketch and tail alike).
        // The energy is re-summed from the sketch (f32 rank-one updates would lose it once it has fallen by 1e6); the full
        // Gram matrix is only the winner's business (step 2)
        if (have) {
            const float il = (float)(1.0 / lam_sh);
            // packed f32 throughout (v_pk_fma_f32): dots, update and the squares of the updated entries
            f2 acc0 = {0.0f, 0.0f}, acc1 = {0.0f, 0.0f}, acc2 = {0.0f, 0.0f};
#pragma unroll
            for (int i4 = 0; i4 < SK_R; i4 += 4) {
                const float4 q4 = *reinterpret_cast<const float4*>(&q_sh[i4]);
                const f2 qa = {q4.x, q4.y}, qb2 = {q4.z, q4.w};
                acc0 = __builtin_elementwise_fma(qa, z[0][i4 / 2], acc0); acc0 = __builtin_elementwise_fma(qb2, z[0][i4 / 2 + 1], acc0);
                acc1 = __builtin_elementwise_fma(qa, z[1][i4 / 2], acc1); acc1 = __builtin_elementwise_fma(qb2, z[1][i4 / 2 + 1], acc1);
                acc2 = __builtin_elementwise_fma(qa, z[2][i4 / 2], acc2); acc2 = __builtin_elementwise_fma(qb2, z[2][i4 / 2 + 1], acc2);
                if ((i4 & 12) == 12) __builtin_amdgcn_sched_barrier(0);
            }
            float a0 = (acc0.x + acc0.y) * il, a1 = (acc1.x + acc1.y) * il, a2 = (acc2.x + acc2.y) * il;
            if (mine) { a0 = my_u[0]; a1 = my_u[1]; a2 = my_u[2]; tail *= (2.0f / 3.0f); }
            const f2 m0 = {-a0, -a0}, m1 = {-a1, -a1}, m2 = {-a2, -a2};
            f2 sq0 = {0.0f, 0.0f}, sq1 = {0.0f, 0.0f}, sq2 = {0.0f, 0.0f};
#pragma unroll
            for (int i4 = 0; i4 < SK_R; i4 += 4) {
                const float4 q4 = *reinterpret_cast<const float4*>(&q_sh[i4]);
                const f2 qa = {q4.x, q4.y}, qb2 = {q4.z, q4.w};
                z[0][i4 / 2] = __builtin_elementwise_fma(m0, qa, z[0][i4 / 2]); z[0][i4 / 2 + 1] = __builtin_elementwise_fma(m0, qb2, z[0][i4 / 2 + 1]);
                z[1][i4 / 2] = __builtin_elementwise_fma(m1, qa, z[1][i4 / 2]); z[1][i4 / 2 + 1] = __builtin_elementwise_fma(m1, qb2, z[1][i4 / 2 + 1]);
                z[2][i4 / 2] = __builtin_elementwise_fma(m2, qa, z[2][i4 / 2]); z[2][i4 / 2 + 1] = __builtin_elementwise_fma(m2, qb2, z[2][i4 / 2 + 1]);
                sq0 = __builtin_elementwise_fma(z[0][i4 / 2], z[0][i4 / 2], sq0); sq0 = __builtin_elementwise_fma(z[0][i4 / 2 + 1], z[0][i4 / 2 + 1], sq0);
                sq1 = __builtin_elementwise_fma(z[1][i4 / 2], z[1][i4 / 2], sq1); sq1 = __builtin_elementwise_fma(z[1][i4 / 2 + 1], z[1][i4 / 2 + 1], sq1);
                sq2 = __builtin_elementwise_fma(z[2][i4 / 2], z[2][i4 / 2], sq2); sq2 = __builtin_elementwise_fma(z[2][i4 / 2 + 1], z[2][i4 / 2 + 1], sq2);
                if ((i4 & 12) == 12) __builtin_amdgcn_sched_barrier(0);
            }
            const float s0 = sq0.x + sq0.y, s1 = sq1.x + sq1.y, s2 = sq2.x + sq2.y;
            e = (s0 + s1) + s2 + tail;
        }
        if (tl) tlog[t * 6 + 4] = wall_clock64();
    }
    if (aborted) {
        if (tid == 0) __hip_atomic_store(flags + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (have) score_out[v] = e_in;
        return;
    }
    if (have) score_out[v] = (double)score;
    if (blockIdx.x == 0 && tid == 0) __hip_atomic_store(flags + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ran to the end
}

// vertices one launch can hold (one 512-thread block per CU, all co-resident)
long long asb_sketch_capacity(asb_ctx* ctx) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_sketch_greedy, SK_T, 0) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        return 0;
    }
    const long long cap = (long long)per_cu * ctx->n_cu * SK_T;
    return cap < 256LL * SK_T ? cap : 256LL * SK_T;          // (the gate's exchange holds 256 blocks)
}

// ---- the replay on a SUBSET (shards above one co-resident launch: 256 blocks x 512 vertices).  A vertex's energy never grows, so
// who can win within the next 64 steps lies among the largest energies NOW: the launch takes the ~118 000 largest (threshold by
// the panel's two-level histogram, ordered compaction so that slot order is vertex order) and everyone else keeps score 0.
__global__ __launch_bounds__(256) void k_subset_count(const double* __restrict__ E, long long n, const double* __restrict__ tau_p,
                                                      int* __restrict__ cnt) {
    __shared__ int sh[4];
    const double tau = *tau_p;
    const long long seg = (n + gridDim.x - 1) / gridDim.x, a = blockIdx.x * seg, b = (a + seg < n) ? a + seg : n;
    int c = 0;
    for (long long i = a + threadIdx.x; i < b; i += 256) c += E[i] > tau;
    c = wave_isum_dpp(c);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(256) void k_subset_fill(const double* __restrict__ E, long long n, const double* __restrict__ tau_p,
                                                     const int* __restrict__ cnt, long long cap, int* __restrict__ map,
                                                     int* __restrict__ total_out) {
    __shared__ int wsum[4];
    __shared__ long long base_sh;
    const double tau = *tau_p;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x < 64) {
        long long off = 0;
        for (int q = lane; q < (int)blockIdx.x; q += 64) off += cnt[q];
        off = (long long)wave_isum_dpp((int)off);
        if (lane == 0) base_sh = off;
    }
    __syncthreads();
    long long base = base_sh;
    const long long seg = (n + gridDim.x - 1) / gridDim.x, a = blockIdx.x * seg, b = (a + seg < n) ? a + seg : n;
    for (long long i0 = a; i0 < b; i0 += 256) {
        const long long i = i0 + threadIdx.x;
        const bool on = i < b && E[i] > tau;
        const unsigned long long m = __ballot(on);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wv] = __popcll(m);
        __syncthreads();
        int woff = 0;
        for (int q = 0; q < wv; ++q) woff += wsum[q];
        const long long pos = base + woff + before;
        if (on && pos < cap) map[pos] = (int)i;
        base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total_out = (int)(base < cap ? base : cap);
}

// scores of the n vertices whose sketch columns are cols[i * stride + 3 v + d] (i < r <= 64), exact energies E, into
// ctx->sk_score (n doubles); the replay's predicted winners into ctx->sk_pred (<= 64).  Enqueued on the context's stream.
int asb_sketch_predict(asb_ctx* ctx, const double* cols, long long stride, const double* wn2, int wn2_stride, const double* E,
                       long long n, int r, int steps) {
    if (!ctx || !cols || !wn2 || !E || n < 1 || r < 1 || r > SK_R || steps < 1) return ASB_ERR_ARG;
    if (steps > 64) steps = 64;
    const long long cap = asb_sketch_capacity(ctx);
    if (cap < SK_T) ASB_FAIL(ctx, ASB_ERR_LIMIT, "asb_sketch_predict: the replay kernel does not fit this device");
    if (n >= (1LL << 31)) ASB_FAIL(ctx, ASB_ERR_LIMIT, "asb_sketch_predict: more than 2^31 vertices per shard");
    int rc;
    const int* map = nullptr;
    long long n_run = n;
    if (n > cap) {
        // the ~0.9 cap largest energies take part (asb.h: energies never grow); threshold from the panel machinery's two-level
        // histogram (its scalars are free between two reads), ordered compaction, the subset's size read back
        extern int asb_sketch_subset_tau(asb_ctx * ctx, const double* E, long long n, long long m_target, long long m_cap);
        if ((rc = asb_sketch_subset_tau(ctx, E, n, cap - cap / 10, cap))) return rc;
        const int nb = 1024;
        if ((rc = asb_alloc(ctx, &ctx->sk_map, (size_t)cap))) return rc;
        if ((rc = asb_alloc(ctx, &ctx->sk_cnt, (size_t)nb + 1))) return rc;
        hipLaunchKernelGGL(k_subset_count, dim3(nb), dim3(256), 0, ctx->stream, E, n, ctx->scalar_dev + 6 /* SC_TAU */, ctx->sk_cnt);
        hipLaunchKernelGGL(k_subset_fill, dim3(nb), dim3(256), 0, ctx->stream, E, n, ctx->scalar_dev + 6, ctx->sk_cnt, cap, ctx->sk_map,
                           ctx->sk_cnt + nb);
        ASB_CHECK_LAUNCH(ctx);
        int tot = 0;
        ASB_HIP(ctx, hipMemcpyAsync(&tot, ctx->sk_cnt + nb, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (tot < 1) ASB_FAIL(ctx, ASB_ERR_NUMERIC, "asb_sketch_predict: empty replay subset");
        n_run = tot;
        map = ctx->sk_map;
    }
    const int G = (int)((n_run + SK_T - 1) / SK_T);
    if (G > 256) ASB_FAIL(ctx, ASB_ERR_LIMIT, "asb_sketch_predict: more than 256 blocks");
    const size_t n_words = (size_t)3 * G + (size_t)3 * SK_QW + (size_t)2 * G;
    if ((rc = asb_alloc(ctx, &ctx->sk_words, n_words))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->sk_flags, (size_t)4 + 2 * 64 * 6))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->sk_score, (size_t)n))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->sk_pred, (size_t)64))) return rc;
    if (!ctx->sk_counts) {
        if ((rc = asb_alloc(ctx, &ctx->sk_counts, (size_t)4))) return rc;
        ASB_HIP(ctx, hipMemsetAsync(ctx->sk_counts, 0, 4 * sizeof(unsigned), ctx->stream));
    }
    static const float min_share = getenv("ASB_SKETCH_MIN_SHARE") ? (float)atof(getenv("ASB_SKETCH_MIN_SHARE")) : 0.15f;
    ASB_HIP(ctx, hipMemsetAsync(ctx->sk_words, 0xFF, n_words * sizeof(unsigned long long), ctx->stream));
    ASB_HIP(ctx, hipMemsetAsync(ctx->sk_flags, 0, 4 * sizeof(unsigned), ctx->stream));
    ASB_HIP(ctx, hipMemsetAsync(ctx->sk_pred, 0xFF, 64 * sizeof(long long), ctx->stream));
    if (map) ASB_HIP(ctx, hipMemsetAsync(ctx->sk_score, 0, (size_t)n * sizeof(double), ctx->stream));      // everyone outside the subset
    hipLaunchKernelGGL(k_sketch_greedy, dim3(G), dim3(SK_T), 0, ctx->stream, cols, stride, wn2, wn2_stride, E, n_run, r, steps,
                       ctx->sk_score, ctx->sk_pred, ctx->sk_words, ctx->sk_flags, ctx->sk_test_stall, min_share, ctx->sk_counts, map);
    ASB_CHECK_LAUNCH(ctx);
    ctx->sk_test_stall = 0;
    ctx->n_sketch_runs++;
    if (getenv("ASB_DEBUG_PANELS")) {
        unsigned long long tlh[64 * 6];
        unsigned flh[4] = {0, 0, 0, 0};
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemcpy(flh, ctx->sk_flags, sizeof(flh), hipMemcpyDeviceToHost);
        (void)hipMemcpy(tlh, ctx->sk_flags + 4, sizeof(tlh), hipMemcpyDeviceToHost);
        if (!flh[2]) steps = 0;
        for (int t = 0; t < steps; t += 9)
            fprintf(stderr, "[asb]   replay step %2d: best+publish %.2f | poll+eigen %.2f | resets+score %.2f | winner q %.2f | deflate %.2f | total %.2f us\n", t,
                    (tlh[t * 6 + 1] - tlh[t * 6 + 0]) * 0.01, (tlh[t * 6 + 2] - tlh[t * 6 + 1]) * 0.01, 0.0,
                    (tlh[t * 6 + 3] - tlh[t * 6 + 2]) * 0.01, (tlh[t * 6 + 4] - tlh[t * 6 + 3]) * 0.01, (tlh[t * 6 + 4] - tlh[t * 6 + 0]) * 0.01);
        if (steps > 1) fprintf(stderr, "[asb]   replay: %d steps in %.1f us, first step starts %.1f us\n", steps,
                               (tlh[(steps - 1) * 6 + 4] - tlh[0]) * 0.01, 0.0);
    }
    return ASB_OK;
}

// test entry: the replay on caller-supplied host arrays (cols: r x 3 n, wn2: r, E: n) -> scores (n), predicted winners
// (steps, -1 where the replay ended), status (1: ran to the end, 0: aborted -> scores are the energies)
extern "C" int asb_test_sketch_predict(asb_ctx* ctx, const double* cols, const double* wn2, const double* E, int64_t n, int r,
                                       int steps, double* scores, int64_t* pred, int* status) {
    if (!ctx || !cols || !wn2 || !E || !scores || !pred || !status || n < 1 || r < 1 || r > SK_R || steps < 1 || steps > 64)
        return ASB_ERR_ARG;
    double *dc = nullptr, *dw = nullptr, *de = nullptr;
    auto body = [&]() -> int {
        ASB_HIP(ctx, hipMalloc((void**)&dc, (size_t)r * 3 * n * sizeof(double)));
        ASB_HIP(ctx, hipMalloc((void**)&dw, (size_t)r * sizeof(double)));
        ASB_HIP(ctx, hipMalloc((void**)&de, (size_t)n * sizeof(double)));
        ASB_HIP(ctx, hipMemcpy(dc, cols, (size_t)r * 3 * n * sizeof(double), hipMemcpyHostToDevice));
        ASB_HIP(ctx, hipMemcpy(dw, wn2, (size_t)r * sizeof(double), hipMemcpyHostToDevice));
        ASB_HIP(ctx, hipMemcpy(de, E, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
        const int rc = asb_sketch_predict(ctx, dc, 3 * n, dw, 1, de, n, r, steps);
        if (rc != ASB_OK) return rc;
        unsigned fl[4];
        long long pr[64];
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ASB_HIP(ctx, hipMemcpy(scores, ctx->sk_score, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
        ASB_HIP(ctx, hipMemcpy(pr, ctx->sk_pred, sizeof(pr), hipMemcpyDeviceToHost));
        ASB_HIP(ctx, hipMemcpy(fl, ctx->sk_flags, sizeof(fl), hipMemcpyDeviceToHost));
        for (int t = 0; t < steps; ++t) pred[t] = pr[t];
        *status = fl[2] ? 1 : 0;
        return ASB_OK;
    };
    const int rc = body();              // (the staging buffers go on every path)
    (void)hipFree(dc);
    (void)hipFree(dw);
    (void)hipFree(de);
    return rc;
}
