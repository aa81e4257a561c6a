// Internal definitions shared by the translation units of libasb_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "asb.h"

// Device-resident state of one panel of the projection path (asb_project.hip).
#define ASB_PANEL_COLS 16          // steps (weight columns) per panel = columns of one projection pass

struct PanelState {
    double theta;          // upper bound on the energy of every NON-candidate vertex
    double margin;         // absolute safety margin on that bound (rounding of the energy recurrence)
    long long done;        // set when the best candidate can no longer be proven to be the global arg-max
    long long committed;   // components committed in this panel
    long long n_cand;
    long long pad;
    // steps taken without proof (verified against every vertex's energy after the projection pass, k_correct<true>)
    long long proven;      // length of the provable head of the panel (-1 until the panel kernel has set it)
    long long spec_max;    // how many unproven steps the panel kernel may add
    long long spec_ok;     // first unproven step the verification rejected (>= committed: none)
    double e_win[16];      // energy of each step's winner
};

struct StreamCfg {
    int T, E2, block, vpb;
};

struct asb_splocs;
struct asb_geo;

struct asb_ctx {
    int dev = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    std::map<void*, size_t> alloc_bytes;   // capacity of each context-owned buffer

    // ---- snapshot shard, vertex-major: row r = 3*v + d, Fp doubles per row ----
    int64_t F = 0, Fp = 0, n_loc = 0, v0 = 0, N_glob = 0;
    double* X = nullptr;      // (3*n_loc, Fp)   prepared snapshots (snapTensor)
    double* mean = nullptr;   // (3*n_loc)
    bool have_mean = false;
    // per-vertex energies |X_v|^2 of the PREPARED tensor, a by-product of the last sweep that wrote it (k_scale_energy) or
    // of the first projection-mode begin after X changed; every writer of X clears e0_valid
    double* tr_part = nullptr;  // per-strip [sum, sum of squares] of the fused layout change
    double* E0 = nullptr;       // (n_loc)
    double* e0_sc = nullptr;    // [|X|^2 of the shard, largest energy, energy along the constant-in-time direction]
    // EV[v] = E0[v] - sum_d (sum_f X[v,d,f])^2 / F: the energy left once the constant-in-time direction is gone -- what the
    // first panel uses to GUESS its later winners when that direction carries much of |X|^2 (mean_frac; rest shape "first")
    double* EV = nullptr;
    int* hist6 = nullptr;             // histograms / range blocks of the scores of a guessed selection
    double* scm = nullptr;
    bool hist6_clear = false;
    double* mean_part = nullptr;      // per-block partials of that energy
    double ev_cv2 = 0.0;            // squared coefficient of variation of EV over the vertices (asb_snapshots_scale)
    double mean_frac = 0.0, mean_energy = 0.0, prep_normx2 = 0.0;     // share / energy along that direction, |X|^2 (host)
    int64_t m_target_eff = 0;   // != 0 while a guessed panel is being selected: the (smaller) target of the energies proper
    int first_panel_mean = 1;   // ASB_FIRST_PANEL_MEAN=0: first panel from the initial energies alone
    const double* sel_e2 = nullptr;      // != NULL while a panel's candidates are { E > tau } u { sel_e2 > tau_v }
    bool e0_valid = false;
    bool ev_valid = false;          // EV / mean_energy / prep_normx2 describe the CURRENT tensor (set by asb_snapshots_scale only; every
                                    // writer of X and asb_project_begin's own energy pass clear it)
    int64_t n_energy_pass = 0;  // reads of X the last asb_deflate_begin spent on initial energies (statistics)

    // ---- reduction scratch ----
    int nblk_cap = 0;
    double* pmax = nullptr;
    long long* pidx = nullptr;
    double* psum = nullptr;
    double* scalar_dev = nullptr;   // a few doubles for device-side scalars
    double* s_dev = nullptr;        // (n_loc) support factor

    // ---- deflation state ----
    int64_t K = 0;
    int mode = 0, local = 0;
    int64_t k_done = 0;
    // overlapped download of the basis (asb_components_stream): pinned (K, n_loc, 3) buffer, copy stream, rows enqueued so far
    int dl_enabled = 0;
    double* dl_host = nullptr;
    size_t dl_host_count = 0;
    int dl_host_owned = 1;            // 0: the caller's buffer (asb_components_stream_into): never freed here
    hipStream_t dl_stream = nullptr;
    hipEvent_t dl_event = nullptr;
    long long dl_done = 0;
    double* res_pin = nullptr;        // pinned host slot of a run's small results (asb_project_results): [seq, -, (K+1) x 4 scalars, 8 range scalars]
    double* res_pin_dev = nullptr;
    size_t res_pin_count = 0;
    double* td_wy = nullptr;          // blocked back-transformation: T factors, reflector panels, work
    // the sparse differential operator S^T of the constraint path (asb_st_upload: CSR on the device) and its work arrays
    long long* st_indptr = nullptr;
    long long* st_indices = nullptr;
    double* st_data = nullptr;
    long long st_rows = 0, st_cols = 0, st_nnz = 0;
    double* st_energy = nullptr;      // (st_rows) squared row norms of S^T M
    double* st_amax = nullptr;        // (st_rows) largest |entry| per row of S^T M
    double* st_resid = nullptr;       // (n_loc, 3 p) residual block of the position-space interpolation error
    unsigned* coop_bar = nullptr;     // k_panel_multi: flags [-, abort, too many candidates, -] + debug timestamps
    double* coop_rec = nullptr;       // (2, grid) records {e, lam, wn2, slot}
    int panel_coop = 1;               // ASB_PANEL_COOP=0 -> the two-kernel inner loop
    int e0_reuse = 1;                 // ASB_E0_REUSE=0 -> asb_project_begin always re-reads X for the initial energies
    int correct_rows = 1;             // ASB_CORRECT_ROWS=0 -> the one-thread-per-vertex correction kernel (k_correct)
    int coop_test_stall = 0;          // ASB_COOP_TEST_STALL=1 (tests): the first co-resident launch is made to time out
    int64_t n_guess_panels = 0;       // first panels of the last run whose candidates were guessed (asb_project_run)
    int64_t n_coop_fallbacks = 0;     // launches of k_panel_coop whose record exchange timed out (redone by the two-kernel loop)
    // super-panels (asb_project.hip): how the next asb_panel_run behaves / what it did
    int run_writeback = 0, run_theta_band = 0, run_coop_used = 0;
    int spec_panels = 1;              // ASB_SPEC_PANELS=0 -> provable steps only
    int gather_cpt = 2;               // ASB_GATHER_CPT=1 -> one candidate per block in k_gather<256,4>
    int run_spec_max = 0;             // unproven steps the next asb_panel_run may take (0 outside asb_project_run)
    int spec_budget = 16;             // adapted to how many unproven steps survived in the last panels
    long long run_proven = 0;         // provable head of the last asb_panel_run
    double* w_fk = nullptr;                 // weights in the reference's (F, K) order for the read-back
    unsigned char* host_pin = nullptr;      // pinned (coherent, device-mapped) host memory for the small read-backs
    unsigned char* host_pin_dev = nullptr;  // its device address: tiny kernels publish state there, the host polls (asb_pin_alloc)
    unsigned long long pin_seq = 0;         // sequence number of the last publication
    int host_poll = 1;                      // ASB_HOST_POLL=0: read-backs by copy + stream synchronisation
    long long n_spec_steps = 0, n_spec_kept = 0;      // statistics (asb_deflate_stats)
    int super_panels = 0;             // ASB_SUPER_PANELS=1
    long long band_target = 12288, band_cap = 16384;
    long long* band_idx = nullptr;
    long long *btmp = nullptr, *bcnt = nullptr;
    double *band_E = nullptr, *bpmax = nullptr, *bpsum = nullptr;
    long long* bpidx = nullptr;
    PanelState* bstate = nullptr;
    double *Wt3 = nullptr, *wn2t3 = nullptr, *Wq3 = nullptr, *gram3 = nullptr;
    int64_t forced_row = -1;      // asb_deflate_force_next: global row the next pick must take
    double* bam_val = nullptr;    // asb_deflate_block_argmax partials
    long long* bam_idx = nullptr;
    int nblk = 0;               // partial records written by the last streaming pass
    double* R = nullptr;        // (3*n_loc, Fp) residual (mode RESIDUAL)
    double* energy = nullptr;   // (n_loc)
    double* W = nullptr;        // (K, Fp)
    double* comps = nullptr;    // (K, 3*n_loc)
    double* scal = nullptr;     // (K+1, 4): sigma, |w|^2, idx bits, local ||R||^2 after comp k
    double* xrec = nullptr;     // one exchange record

    // ---- projection (panel) path, asb_project.hip ----
    int n_cu = 256;
    StreamCfg cfg{};
    int64_t m_target = 0, m_cap = 0;   // candidate-set size aimed for / capacity
    double* Wt = nullptr;        // (Fp, 16) panel weights, frame-major (MFMA B operand order)
    double* wn2t = nullptr;      // (16) |w_t|^2 of the panel
    double* candR = nullptr;     // (m_cap, 3, Fp) exact residual rows of the candidates
    double* cand_e = nullptr;    // (m_cap)
    double* cand_c = nullptr;    // (16, m_cap, 3) in-panel coefficients of the candidates
    double* slab_scratch = nullptr;
    long long* cand_idx = nullptr;
    double* cpmax = nullptr;     // partial records of passes over the candidate buffer
    long long* cpidx = nullptr;
    double* cpsum = nullptr;
    int cnblk = 0;
    int64_t n_slots_host = 0;   // candidates in the assembled (multi-rank) buffer
    double* colpart = nullptr;   // (blocks, 16)
    double* gram = nullptr;      // (K, 16) w_j . w_panel
    double* gram_s = nullptr;    // the same divided by |w_t|^2 of the panel's column t (k_correct_rows)
    double* ypart = nullptr;     // partial 16x16 tiles between sweeps of k_project_lds
    unsigned int* tile_counter = nullptr;
    double* Wq = nullptr;        // (Fp/16, 4, 16, 4) panel in MFMA lane order (k_project_l2)
    int l2_variant = 4;          // ASB_L2_VARIANT: 4 = k_project_l2s<4,2,2,1> (two waves per 64-row tile); 0..2 = k_project_l2 (one wave per tile: 32 rows x 4 chunks, 48 x 2, 64 x 2); 5 = four waves per tile
    int project_kernel = 3;      // 1: k_project_mfma (Wt in registers), 2: k_project_lds (Wt in LDS), 3: k_project_l2 (Wt from L2)
    long long* ctmp = nullptr;   // compaction scratch
    long long* ccnt = nullptr;
    int* hist = nullptr;
    PanelState* pstate = nullptr;
    PanelState* pstate2 = nullptr;         // double panels: the first sub-panel's state, kept for its check after the pass
    int double_panels = 1;                 // two sub-panels per read of X (ASB_DOUBLE_PANELS=0: one)
    double* e_class = nullptr;             // energies at the start of a double panel: who was a candidate (both tiles' checks)
    double* e_tmp = nullptr;               // energies as if a tile stood in full (k_correct_rows<true> -> k_apply_tmp)
    double* wide_out = nullptr;            // asb_project_columns_wide: where the multi-tile pass writes (default: comps)
    double* e_tmp4 = nullptr;              // k_check_tiles: tentative energies per tile (4 x n_loc)
    double* chk_rec = nullptr;             // its per-tile block records: pmax | psum | colpart (4 x nblk_cap x (1 + 1 + 16))
    long long* chk_idx = nullptr;
    long long* tile_res = nullptr;         // per tile: columns kept (-1: not reached); [ASB_MAX_SUB]: the chain flag
    int spec_w_rank = 24;                  // ASB_SPEC_W_RANK: blocks ranked below it publish their w ahead of the exchange (0: none)
    int spec_pass = 1;                     // ASB_SPEC_PASS=0: the read's pass is enqueued only once the host knows the sub-panels' counts
    int coop_launch = 0;                   // ASB_COOP_LAUNCH=1: hipLaunchCooperativeKernel for the panel kernel (-1: tried, refused)
    int sub_chain = 1;                     // the sub-panels of a read enqueued without host reads in between (ASB_SUB_CHAIN=0: one by one)
    int tile_chain = 1;                    // tiles of a read finished without host reads in between (ASB_TILE_CHAIN=0: one read per tile)
    int pre_orth = 1;                      // multi-sub-panel reads project on pre-orthogonalised weights (ASB_PRE_ORTH=0: correct after)
    int sub_panels = 4;                    // most sub-panels per read of X with double_panels (ASB_SUB_PANELS, 1..8; from 5 on
                                           // the projection kernel needs more than 256 registers and loses what the saved read gains)
    int sub_first = 4;                     // sub-panels of the first read (ASB_SUB_FIRST); then adapted: sub_cur
    int sub_cur = 0;
    int sub_ntile = 0;                     // tiles of the read in progress (multi-rank steps: asb_panel_sub_*)
    int chain_timed_out = 0;               // the last one-launch run of a read's sub-panels met a poll that did not complete
    // the read in progress of the multi-rank driver (asb_panel_read_*): its tiles, whether its pass is already enqueued
    long long rd_k0 = 0;
    int rd_ntile = 0, rd_nc[8] = {0}, rd_proven[8] = {0}, rd_rgrid = 0;
    double* rd_words = nullptr;            // (8) per tile: columns that stand on this shard; [ASB_MAX_SUB]: status
    int sub_budget[8] = {16, 16, 16, 16, 16, 16, 16, 16};      // steps given to the later sub-panels (adapted to what the last ones kept)
    int64_t n_panels = 0, n_refresh = 0;
    // stall cliff: a run whose reads of X commit fewer than 3/4 of a component each (K beyond the numerical rank: nothing is
    // provable at rounding level, every panel ends in an exact refresh) continues in the residual loop (asb_project_switch_residual)
    int stall_fallback = 1;                // ASB_STALL_FALLBACK=0: keep grinding through panels
    int64_t fb_mark_reads = 0, fb_mark_k = 0;
    int64_t k_switch = -1;                 // component at which this run left the projection mode (-1: it did not)
    // sketch predictor (asb_sketch.hip): candidates of the next read named by a greedy replay in the space of the columns
    // the last read computed for its rejected steps
    int sketch = 1;                        // ASB_SKETCH=0: candidates by energy (and the first panel's guess) only
    bool sketch_valid = false;             // sk_score holds the scores for the read about to start
    bool read_by_score = false, last_by_score = false;      // the read in progress / the one before took predicted candidates
    double rate_plain = -1.0, rate_sketch = -1.0;           // components per modelled ms of the two kinds of read (exponential means)
    int mode_streak = 0, probe_after = 2;
    // measured cost (ms) of a read with 1 .. 4 sub-panels and of a 64-step replay on this context and shape (-1: not seen yet)
    double cost_nt[5] = {-1.0, -1.0, -1.0, -1.0, -1.0}, cost_replay = -1.0;
    int cost_cnt[5] = {0, 0, 0, 0, 0}, cost_replay_cnt = 0;
    int64_t cost_n = 0, cost_Fp = 0;
    bool sketch_run_off = false;           // this run's data are noise-like (a sketch held too little of the residual): no more replays
    unsigned long long* sk_words = nullptr;
    unsigned* sk_flags = nullptr;          // [-, abort, ran to the end, -]
    int* sk_map = nullptr;                 // replay subset (shards above one co-resident launch): slot -> vertex, increasing
    int* sk_cnt = nullptr;
    double* sk_score = nullptr;            // (n_loc)
    long long* sk_pred = nullptr;          // (64) the replay's winners
    unsigned* sk_counts = nullptr;         // [replays run, launches that found the sketch too thin and left score = energy]
    int sk_test_stall = 0;                 // tests: the next launch is made to time out
    int64_t n_sketch_runs = 0, n_sketch_reads = 0;
    // diversity family of a candidate selection (asb_project.hip: in_div): energy-weighted random vertices beside the largest
    int diverse = 1;                       // ASB_DIVERSE=0: candidates by energy / guess / replay scores only
    bool diverse_next = false;             // the next plain read takes half of its candidates by the weighted sample
    bool read_diverse = false;             // the read in progress does
    int64_t n_diverse_reads = 0;

    asb_splocs* splocs = nullptr;   // SPLOCS state (asb_splocs.hip)
    asb_geo* geo = nullptr;         // device geodesics (asb_geodesic.hip)
    long long* geo_src = nullptr;
    double* geo_out = nullptr;

    // ---- small dense linear algebra scratch (asb_linalg.hip) ----
    double* la_part = nullptr;
    size_t la_part_cap = 0;
    double* comps2 = nullptr;     // second basis buffer (orthogonalisation output)
    double* og = nullptr;         // (3, K, K) per-dimension Gram / scaled eigenvectors
    double* olam = nullptr;       // (3, K)
    double* oct = nullptr;        // (3 n_loc, K) transposed basis
    double* ovec = nullptr;       // (3, K, K) eigenvectors
    double* osing = nullptr;      // (3, K) singular values
    double* la_vtmp = nullptr;
    double* kk_tmp = nullptr;     // K x K transposed factor (asb_combine_rows)
    double *pod_g = nullptr, *pod_v = nullptr, *pod_s = nullptr, *pod_coef = nullptr;   // asb_pod.hip
    double* pod_vn = nullptr;     // (F x K) right Ritz vectors / sigma of the power step
    // the POD in levels (asb_pod_deflate_begin / _end): bases kept by finished levels, the deflated copy of the snapshots
    double* pod_u1 = nullptr;
    int64_t pod_u1_rows = 0;
    double *X_deflated = nullptr, *X_original = nullptr;
    int* la_status = nullptr;
    double* dn_sym = nullptr;                     // symmetric Gauss-Jordan: pivot row panel, D x panel, signed transpose, pivot block
    double *dn_work = nullptr, *dn_test = nullptr;   // asb_dense.hip: Gauss-Jordan panels; test matrix
    double* td_backup = nullptr;                  // the matrix before the panels (restored if a panel's exchange times out)
    double* td_panel = nullptr;                   // k_td_panel: x | z partials | V | W | p, q
    unsigned long long* td_rec = nullptr;         // its exchange words (ring of three) and the abort flag
    double* td_ppart = nullptr;                   // partial mat-vec vectors of the tridiagonalisation (one per column chunk)
    double *td_work = nullptr, *td_z = nullptr;   // asb_eig.hip: Householder work vectors / tau / d / e; Z and Q Z
    int64_t td_n = 0;
    // asb_smalldense.hip: tridiagonal eigen-solver, one-sided Jacobi, blocked Cholesky
    double* tri_work = nullptr;
    unsigned char* tri_swp = nullptr;
    double *jac_q = nullptr, *jac_sig = nullptr, *jac_a = nullptr;
    int* jac_where = nullptr;
    double* chol_w = nullptr;
    double* deim_m = nullptr;         // asb_deim_run: Mx, Minv, coef, partials
    long long* deim_pt = nullptr;
    double *eig_lam = nullptr, *eig_v = nullptr;  // asb_sym_eig_topk: eigenvalues (n, descending), leading vectors (n x k)
    int64_t eig_n = 0, eig_k = 0;

    // ---- profiling of the dominant streaming kernel ----
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
};

// projection path entry points (asb_project.hip), dispatched on ctx->mode
int asb_project_begin(asb_ctx* ctx, int64_t K);
long long asb_sketch_capacity(asb_ctx* ctx);       // asb_sketch.hip
int asb_sketch_predict(asb_ctx* ctx, const double* cols, long long stride, const double* wn2, int wn2_stride, const double* E,
                       long long n, int r, int steps);
int asb_project_run(asb_ctx* ctx, int64_t k0, int64_t k1);
void asb_splocs_free(asb_ctx* ctx);
void asb_geo_free(asb_ctx* ctx);
#define ASB_GEO_CACHE_SLABS 64          // x 64 distance fields
const double* asb_geo_cached_field(asb_ctx* ctx, long long slot, long long* n_out);
// small dense linear algebra on the device (asb_linalg.hip)
// out[i*so_i + j*so_j] = sum_r A[r*lda + i*sa] * B[r*ldb + j]   (f64 MFMA; contraction index r slow in A and B)
int asb_gemm_tn_s(asb_ctx* ctx, const double* A, long long lda, long long sa, const double* B, long long ldb, long long Rn,
                  int I, int J, double* out, long long so_i, long long so_j);
static inline int asb_gemm_tn(asb_ctx* ctx, const double* A, long long lda, const double* B, long long ldb, long long Rn,
                              int I, int J, double* out) {
    return asb_gemm_tn_s(ctx, A, lda, 1, B, ldb, Rn, I, J, out, J, 1);
}
int asb_transpose(asb_ctx* ctx, const double* in, long long rows, long long cols, double* out);   // (rows x cols) -> (cols x rows)
// eigen-decomposition of a symmetric n x n matrix (n <= 128) on the device: lam (n) descending, V (n x n) columns
int asb_sym_eig(asb_ctx* ctx, const double* A_dev, int n, double* lam_dev, double* V_dev);
// asb_smalldense.hip
int asb_tri_eig_dev(asb_ctx* ctx, const double* d, const double* e, int n, int k, double* lam_desc, double* Z, int* n_bad);
int asb_jacobi_rows_dev(asb_ctx* ctx, double* A, int nv, int m, long long lda, double* Q_sorted, int q_transposed,
                        double* sig_sorted, int* sweeps_out);
int asb_sym_eig_large(asb_ctx* ctx, const double* A_dev, int n, double* lam_dev, double* V_dev);
int asb_chol_tinv_dev(asb_ctx* ctx, const double* G, int K, double* Tt, int* status_dev);
int asb_components_transform_dev(asb_ctx* ctx, const double* T_dev, int same_T);      // asb_linalg.hip
int asb_project_results(asb_ctx* ctx, double* comps, double* weigs, int64_t* idx, double* sigma, double* normR2_local);

#define ASB_FAIL(ctx, code, ...)                                   \
    do {                                                           \
        char _b[512];                                              \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                     \
        (ctx)->err = _b;                                           \
        return (code);                                             \
    } while (0)

#define ASB_HIP(ctx, call)                                                                   \
    do {                                                                                     \
        hipError_t _e = (call);                                                              \
        if (_e != hipSuccess)                                                                \
            ASB_FAIL(ctx, ASB_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), \
                     __FILE__, __LINE__);                                                    \
    } while (0)

// C = beta C + alpha A B (row-major, even dimensions; asb_dense.hip) and the in-place SPD inverse built on it
bool asb_combine_rows_ok(const asb_ctx* ctx);          // asb_linalg.hip
int asb_combine_rows(asb_ctx* ctx, const double* T_dev);
int asb_gemm_nn(asb_ctx* ctx, const double* A, long long lda, const double* B, long long ldb, double* C, long long ldc, int M,
                int N, int Kc, double alpha, double beta, int tri = 0);
int asb_dense_spd_inverse(asb_ctx* ctx, double* M, int np);
int asb_deflate_apply_dev(asb_ctx* ctx, int64_t k, const double* s_dev);      // asb_deflate.hip
// G = X^T X (n x n, both triangles) for a tall row-major X: LDS-tiled f64 MFMA kernel (asb_linalg.hip)
int asb_syrk_tn(asb_ctx* ctx, const double* X, long long ld, long long R, int n, double* out);
int asb_gemm_tn_big(asb_ctx* ctx, const double* X, long long ldx, const double* Y, long long ldy, long long R, int I, int J, double* out);

#define ASB_CHECK_LAUNCH(ctx) ASB_HIP(ctx, hipGetLastError())

// 1 KiB of pinned host memory shared by the small per-panel / per-sweep read-backs:
//   [0, 256) PanelState, [256, 272) panel-kernel flags, [384, 392) counters, [448, 456) publication sequence number
static inline int asb_pin_alloc(asb_ctx* ctx) {
    if (ctx->host_pin) return ASB_OK;
    ASB_HIP(ctx, hipHostMalloc((void**)&ctx->host_pin, 1024, hipHostMallocCoherent | hipHostMallocMapped));
    for (int i = 0; i < 1024; ++i) ctx->host_pin[i] = 0;
    if (hipHostGetDevicePointer((void**)&ctx->host_pin_dev, ctx->host_pin, 0) != hipSuccess) ctx->host_pin_dev = nullptr;
    return ASB_OK;
}

// (Re)allocates *p to hold `count` elements; an existing allocation that is already large
// enough (and not more than 2x too large) is kept, so repeated runs on one context do not
// pay hipMalloc/hipFree inside a timed region.
template <typename T>
static inline int asb_alloc(asb_ctx* ctx, T** p, size_t count) {
    const size_t want = count * sizeof(T);
    auto it = ctx->alloc_bytes.find((void*)p);
    if (*p && it != ctx->alloc_bytes.end() && it->second >= want && it->second <= 2 * want + 4096) return ASB_OK;
    if (*p) {
        (void)hipFree(*p);
        *p = nullptr;
        ctx->alloc_bytes.erase((void*)p);
    }
    if (count == 0) return ASB_OK;
    ASB_HIP(ctx, hipMalloc((void**)p, want));
    ctx->alloc_bytes[(void*)p] = want;
    return ASB_OK;
}

int asb_dl_begin(asb_ctx* ctx);          // asb_linalg.hip
// component rows [dl_done, k_to) are final: copy them to the pinned buffer on the copy stream, behind what the main stream
// has enqueued so far (no-op unless asb_components_stream is on)
static inline int asb_dl_enqueue(asb_ctx* ctx, long long k_to) {
    if (!ctx->dl_enabled || !ctx->dl_host || !ctx->comps || k_to <= ctx->dl_done) return ASB_OK;
    if (k_to > ctx->K) k_to = ctx->K;
    const size_t row = (size_t)3 * ctx->n_loc;
    if ((size_t)k_to * row > ctx->dl_host_count) return ASB_OK;
    ASB_HIP(ctx, hipEventRecord(ctx->dl_event, ctx->stream));
    ASB_HIP(ctx, hipStreamWaitEvent(ctx->dl_stream, ctx->dl_event, 0));
    ASB_HIP(ctx, hipMemcpyAsync(ctx->dl_host + (size_t)ctx->dl_done * row, ctx->comps + (size_t)ctx->dl_done * row,
                                (size_t)(k_to - ctx->dl_done) * row * sizeof(double), hipMemcpyDeviceToHost, ctx->dl_stream));
    ctx->dl_done = k_to;
    return ASB_OK;
}

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// Wave-wide reductions without the LDS crossbar (gfx950): four DPP stages inside a row of 16 lanes (quad_perm xor 1, xor 2,
// row_half_mirror, row_mirror), then v_permlane16_swap / v_permlane32_swap across rows -- a stage is two 32-bit moves and an
// add instead of two ds_bpermute round trips (~6 x 120 cycles for a butterfly of __shfl_xor).  Every stage combines two
// values that are each identical in the lanes they come from, so all 64 lanes end with bit-identical results.
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// the two values a lane sees after the swap: (own, partner) for rows (16) / halves (32)
__device__ __forceinline__ void swap16_f64(double v, double& a, double& b) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    a = __hiloint2double((int)h[0], (int)l[0]);
    b = __hiloint2double((int)h[1], (int)l[1]);
}
__device__ __forceinline__ void swap32_f64(double v, double& a, double& b) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    a = __hiloint2double((int)h[0], (int)l[0]);
    b = __hiloint2double((int)h[1], (int)l[1]);
}
template <int NV>
__device__ __forceinline__ void wave_sum_dpp(double (&v)[NV]) {
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] += dpp_mov_f64<0xB1>(v[q]);
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] += dpp_mov_f64<0x4E>(v[q]);
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] += dpp_mov_f64<0x141>(v[q]);
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] += dpp_mov_f64<0x140>(v[q]);
#pragma unroll
    for (int q = 0; q < NV; ++q) { double a, b; swap16_f64(v[q], a, b); v[q] = a + b; }
#pragma unroll
    for (int q = 0; q < NV; ++q) { double a, b; swap32_f64(v[q], a, b); v[q] = a + b; }
}
__device__ __forceinline__ double wave_max_dpp(double v) {
    v = fmax(v, dpp_mov_f64<0xB1>(v));
    v = fmax(v, dpp_mov_f64<0x4E>(v));
    v = fmax(v, dpp_mov_f64<0x141>(v));
    v = fmax(v, dpp_mov_f64<0x140>(v));
    double a, b;
    swap16_f64(v, a, b); v = fmax(a, b);
    swap32_f64(v, a, b); v = fmax(a, b);
    return v;
}
__device__ __forceinline__ int wave_min_dpp(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true));
    auto s = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    v = min((int)s[0], (int)s[1]);
    s = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return min((int)s[0], (int)s[1]);
}

__device__ __forceinline__ int wave_isum_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);
    auto s = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    v = (int)s[0] + (int)s[1];
    s = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (int)s[0] + (int)s[1];
}

// Block-wide sum of NV values per thread; result valid in every thread.
// scratch: at least NV * (blockDim.x/64) doubles of LDS.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* scratch) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    if (nw == 1) return;
    __syncthreads();
    if (lane == 0)
        for (int i = 0; i < NV; ++i) scratch[wid * NV + i] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double s = 0;
        for (int w = 0; w < nw; ++w) s += scratch[w * NV + i];
        v[i] = s;
    }
}

// "better" for arg-max with NumPy's first-max tie-break: larger value, then lower index.
__device__ __forceinline__ bool am_better(double e, long long i, double be, long long bi) {
    return (e > be) || (e == be && i < bi);
}
