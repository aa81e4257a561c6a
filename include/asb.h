/*
 * asb.h -- C ABI of libasb_hip.so: the MI355X (gfx950) implementation of the
 * animSnapBases snapshot-reduction hot path.
 *
 * The reference (ShMonem/animSnapBases) is pure Python and has NO FFI / plugin
 * interface for this path: the drop-in boundary is the Python class
 * API (posSnapshots / posComponents), mirrored in animsnapbases_amd/.  This header is
 * the thin C ABI underneath it; every entry point names the reference lines whose
 * arithmetic it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain C: pointers, int64_t sizes, doubles.  No torch / numpy types.
 *   - return 0 on success, a negative asb_status otherwise; text via asb_last_error().
 *   - host buffers are caller-owned, C-contiguous, float64 / int64.
 *   - pointers named *_dev are DEVICE pointers owned by the caller (e.g. the storage of
 *     a torch tensor used for an RCCL collective); all other device memory belongs to
 *     the context.
 *   - a context is bound to one GPU and one HIP stream and is not thread-safe.
 *   - all work is enqueued on the context's stream; calls that return host data
 *     synchronise that stream, all others are asynchronous.
 *
 * Device layout ("vertex-major"): the (F, N, 3) snapshot tensor of the reference is held
 * as rows r = 3*v + d of Fp = roundup(F, 16) doubles, row r at byte offset r*Fp*8, so that
 * a vertex's 3 x F trajectory is one contiguous 24*Fp-byte run.  Padding entries are 0.
 * Multi-GPU: each context owns the contiguous vertex range [v0, v0 + n_loc).
 */
#ifndef ASB_H
#define ASB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASB_ABI_VERSION 1

typedef struct asb_ctx asb_ctx;

typedef enum asb_status {
    ASB_OK = 0,
    ASB_ERR_ARG = -1,      /* bad argument / wrong call order            */
    ASB_ERR_HIP = -2,      /* HIP runtime error (text has the HIP string) */
    ASB_ERR_NODEV = -3,    /* no usable gfx950 device                     */
    ASB_ERR_LIMIT = -4,    /* size outside what the kernels are built for */
    ASB_ERR_NUMERIC = -5   /* numerical breakdown (e.g. Cholesky pivot)   */
} asb_status;

/* ---------------------------------------------------------------- context ---- */
int asb_abi_version(void);
/* stream: a hipStream_t to enqueue on (e.g. torch's current stream so that RCCL
 * collectives issued by torch.distributed are ordered with the kernels), NULL for a
 * private stream, or ASB_STREAM_DEFAULT for the device's default (null) stream -- which is
 * what torch's current stream is unless the caller switched streams. */
#define ASB_STREAM_DEFAULT ((void*)(intptr_t)-1)
int asb_create(int device_id, void* hip_stream, asb_ctx** out);
void asb_destroy(asb_ctx* ctx);
const char* asb_last_error(const asb_ctx* ctx);
int asb_sync(asb_ctx* ctx);
/* number of launches of the dominant streaming kernel and their total HIP-event time
 * (ms) since the last reset; used by bench.py for the roofline figure. */
int asb_prof_reset(asb_ctx* ctx, int enable);
int asb_prof_get(asb_ctx* ctx, int64_t* launches, double* total_ms);

/* ------------------------------------------------- snapshot preparation ------ */
/* posSnapshots.do_snapshots_precomputations, snapbases/posSnapshots.py:64-105.
 * X: host (F, N_glob, 3).  Uploads vertices [v0, v0+n_loc), multiplies row v by
 * massL[v0+v] when massL != NULL (:82) and stores the vertex-major layout. */
int asb_snapshots_upload(asb_ctx* ctx, const double* X, int64_t F, int64_t N_glob,
                         int64_t v0, int64_t n_loc, const double* massL);
/* Same, but X_dev is already in device memory in the reference layout (F, n_loc, 3)
 * (synthetic benchmark inputs generated on the GPU): this rank's vertices [v0, v0+n_loc)
 * of N_glob. */
int asb_snapshots_adopt_dev(asb_ctx* ctx, const double* X_dev, int64_t F, int64_t n_loc,
                            const double* massL_loc, int64_t v0, int64_t N_glob);
/* :85-89 + posSnapshots.standarize :168 -- mean = frame 0 (rest_shape 0) or the frame
 * average (1); when subtract != 0 the mean row is subtracted.  local_sum = sum of all
 * entries of the shard afterwards (for the global mean of np.std). */
int asb_snapshots_center(asb_ctx* ctx, int rest_shape, int subtract, double* local_sum);
/* asb_snapshots_upload / asb_snapshots_adopt_dev followed by asb_snapshots_center in ONE sweep over the data: with
 * rest_shape 0 ("first", :86) the rest row is frame 0 of the input, so the layout change subtracts it on the fly and also
 * returns sums_out[0] = sum(x) and sums_out[1] = sum(x^2) of the prepared shard (np.std of :170 then needs no sweep of its
 * own: var = sum(x^2)/n - mu^2, the caller falls back to asb_snapshots_sqdev when mu^2 >> var).  rest_shape 1 ("average")
 * runs the separate sweeps and returns sums_out[1] = -1. */
int asb_snapshots_upload_rest(asb_ctx* ctx, const double* X, int64_t F, int64_t N_glob, int64_t v0, int64_t n_loc,
                              const double* massL, int rest_shape, int subtract, double* sums_out);
int asb_snapshots_adopt_dev_rest(asb_ctx* ctx, const double* X_dev, int64_t F, int64_t n_loc, const double* massL_loc,
                                 int64_t v0, int64_t N_glob, int rest_shape, int subtract, double* sums_out);
/* sum over the shard of (x - mu)^2 -- second pass of np.std, :171 */
int asb_snapshots_sqdev(asb_ctx* ctx, double mu, double* local_sqdev);
/* snapTensor *= pre_scale_factor, :172 */
int asb_snapshots_scale(asb_ctx* ctx, double pre_scale_factor);
int asb_snapshots_get_mean(asb_ctx* ctx, double* mean_out /* (n_loc,3) */);
/* the prepared tensor back in the reference layout, out: (F, n_loc, 3) */
int asb_snapshots_download(asb_ctx* ctx, double* out);

/* ------------------------------------- greedy deflation ("PCA") -------------- */
/* posComponents.extract_k_components, snapbases/posComponents.py:67-122.
 *
 * mode ASB_DEFLATE_RESIDUAL keeps the residual tensor R in HBM and updates it per
 * component exactly as the reference does (needed for support='local').
 * mode ASB_DEFLATE_PROJECT is the residual-free form valid for support='global':
 * the weights w_k are mutually orthogonal there, so c_k = X^T w_k / |w_k|^2 and X is
 * only ever READ. */
#define ASB_DEFLATE_RESIDUAL 0
#define ASB_DEFLATE_PROJECT 1
int asb_deflate_begin(asb_ctx* ctx, int64_t K, int mode, int local_support);
/* number of doubles of one exchange record: [energy, idx (int64 bits), slab 3*Fp] */
int64_t asb_deflate_xchg_len(const asb_ctx* ctx);
/* :78-80 on this shard: best local vertex (max residual energy, first on ties) and
 * its 3 x F residual slab -> rec_dev (one record). */
int asb_deflate_local_best(asb_ctx* ctx, int64_t k, double* rec_dev);
/* :79-96: winner over n_rec records (max energy, lowest global index), rank-1 SVD of
 * its 3 x F slab, w_k = sigma_1 * Vt[0]; with local_support also the +-projection
 * test :90-94.  recs_dev == NULL: single rank, reads the shard directly. */
int asb_deflate_pick(asb_ctx* ctx, int64_t k, const double* recs_dev, int64_t n_rec);
/* selected vertex (global index) and sigma_1 of component k (synchronises) */
int asb_deflate_get_pick(asb_ctx* ctx, int64_t k, int64_t* idx, double* sigma);
/* 'pca_blocks' constraint bases (snapbases/constraintsComponents.py:324-412; residual mode).
 * asb_deflate_block_argmax: the constraint (block of p consecutive rows) with the largest residual energy on this
 * shard -- indxLargestDeformation, :86-92 (first maximum); the shard must hold whole blocks; block_out is global.
 * asb_deflate_force_next: the next asb_deflate_local_best / asb_deflate_pick takes global row gidx instead of the
 * arg-max (the reference deflates the p rows of the chosen constraint one after the other, :352-356). */
int asb_deflate_block_argmax(asb_ctx* ctx, int p, int64_t* block_out, double* val_out);
int asb_deflate_force_next(asb_ctx* ctx, int64_t gidx);
/* :101-111: c_k = (w_k . R)[* s] / |w_k|^2, R -= w_k (x) c_k, new energies.
 * s: host (n_loc) support factor 1 - support_map (:95), or NULL for global. */
int asb_deflate_apply(asb_ctx* ctx, int64_t k, const double* s);
/* components k0 .. k1-1 back to back on one rank with global support (no host round
 * trips inside). */
int asb_deflate_run_global(asb_ctx* ctx, int64_t k0, int64_t k1);
/* outputs.  comps (K, n_loc, 3); weigs (F, K); idx (K) global indices; sigma (K);
 * normR2_local (K): this shard's ||R||_F^2 after each component (:113, summed over
 * shards and square-rooted by the caller). Any pointer may be NULL. */
int asb_deflate_results(asb_ctx* ctx, double* comps, double* weigs, int64_t* idx,
                        double* sigma, double* normR2_local);
/* ---- projection mode, panel by panel (what asb_deflate_run_global does internally on one
 * rank; the multi-rank driver interleaves the collectives marked [ALL-*]) ------------------
 *   asb_panel_scale                     [ALL-REDUCE max of e0max, then set it on every rank]
 *   per panel:
 *     asb_panel_hist/tau(1), (2)  (local)  asb_panel_top_energies  [ALL-GATHER cap+1 doubles]  asb_panel_global_tau
 *     asb_panel_select(rows, ids behind the rows)  [ALL-GATHER rows+ids]  asb_panel_assemble_packed
 *     asb_panel_run(_spec) -> steps (identical on every rank: same data, same arithmetic)
 *     asb_panel_project    (local shard)   | with unproven steps: asb_panel_project_spec [ALL-REDUCE min] asb_panel_commit
 *   (older form, still supported: histograms all-reduced per level, rows and ids gathered separately)           */
int asb_panel_scale(asb_ctx* ctx, double* normX2_local, double* e0max_local, double set_e0max);
/* local energy histogram (ASB_NBINS = 2048 ints) into hist_dev (NULL: internal) */
int asb_panel_hist(asb_ctx* ctx, int level, int* hist_dev);
/* threshold step from the (summed) histogram; the histogram is consumed (cleared to zero) */
int asb_panel_tau(asb_ctx* ctx, int level, const int* hist_dev);
/* One-exchange thresholding for several ranks: after the LOCAL histogram steps (hist_dev = NULL, no all-reduce),
 * asb_panel_top_energies writes this shard's energies above its local threshold into out_dev[0 .. cap) (unordered,
 * padded with -1) and that local threshold into out_dev[cap].  The ranks all-gather the cap + 1 doubles; the global
 * threshold tau = max((m_target + 1)-th largest exported energy, max of the local thresholds) is installed with
 * asb_panel_set_tau (device scalar) on every rank.  asb_panel_target: m_target. */
int asb_panel_top_energies(asb_ctx* ctx, double* out_dev, int64_t cap);
/* the same selection on the device: tab_dev = the all-gathered (world, cap + 1) exports; installs tau and returns the
 * per-rank candidate counts (host, world).  ASB_ERR_LIMIT when world * cap does not fit the selection kernel's LDS. */
int asb_panel_global_tau(asb_ctx* ctx, const double* tab_dev, int world, int64_t cap, int64_t* counts_out);
int asb_panel_set_tau(asb_ctx* ctx, const double* tau_dev);
int64_t asb_panel_target(const asb_ctx* ctx);
/* this shard's candidates (energy > tau; every vertex when global_all; only forced_gidx when
 * >= 0), in vertex order, and their exact residual rows (3*Fp doubles each) into the caller's
 * device buffers of capacity asb_panel_capacity() (NULL: the context's own buffer).
 * n_local / overflow (optional) synchronise. */
int asb_panel_select(asb_ctx* ctx, int64_t k, int64_t forced_gidx, int global_all, double* rows_out_dev,
                     long long* idx_out_dev, int64_t* n_local, int* overflow);
int64_t asb_panel_capacity(const asb_ctx* ctx);
/* replicated candidate buffer from the all-gathered padded pieces:
 * rows_g_dev (world, maxcount, 3, Fp), idx_g_dev (world, maxcount), counts (host, world) */
int asb_panel_assemble(asb_ctx* ctx, const double* rows_g_dev, const long long* idx_g_dev,
                       const int64_t* counts, int world, int64_t maxcount);
/* the same from ONE all-gathered buffer (one collective instead of two): every rank's piece is its maxcount rows
 * followed by its maxcount vertex ids (int64), maxcount * (3*Fp + 1) eight-byte words per rank -- i.e. asb_panel_select
 * was given idx_out_dev = (long long*)(rows_out_dev + maxcount * 3*Fp) */
int asb_panel_assemble_packed(asb_ctx* ctx, const double* packed_g_dev, const int64_t* counts, int world, int64_t maxcount);
/* up to `steps` (<= 16) greedy steps on the candidate buffer; *committed of them are final */
int asb_panel_run(asb_ctx* ctx, int64_t k0, int steps, int global_all, int assembled, int64_t* committed);
/* one pass over X for components [k0, k0+ncols): c_k on this shard, energies */
int asb_panel_project(asb_ctx* ctx, int64_t k0, int ncols);
/* Multi-rank form of the unproven steps (asb_deflate_spec_stats): asb_panel_run_spec lets the panel kernel append up to
 * spec_max steps whose winner is not provable in advance (*ran steps in all, the first *proven of them certain; identical
 * on every rank).  asb_panel_project_spec is the pass over X for all of them with the energies left untouched;
 * *first_rejected = first unproven step that one of THIS shard's vertices contradicts (ncols: none).  The caller takes
 * the minimum over the ranks and asb_panel_commit applies the energies of that many columns (0: nothing stood). */
int asb_panel_run_spec(asb_ctx* ctx, int64_t k0, int steps, int global_all, int assembled, int spec_max, int64_t* ran,
                       int64_t* proven);
int asb_panel_project_spec(asb_ctx* ctx, int64_t k0, int ncols, int proven, int64_t* first_rejected);
int asb_panel_commit(asb_ctx* ctx, int64_t k0, int kept);
/* asb_panel_project_spec without the host read-back: the count is written (as a float64) to the caller's device word,
 * which the multi-rank driver min-all-reduces and reads once (one host synchronisation per panel instead of two). */
int asb_panel_project_spec_dev(asb_ctx* ctx, int64_t k0, int ncols, int proven, double* first_rejected_dev);
/* fallback: exact energies of every vertex of the shard; optional first arg-max */
int asb_panel_refresh(asb_ctx* ctx, int64_t k, double* best_energy, int64_t* best_gidx);

/* projection mode statistics of the last run: streaming passes over X (panels) and exact
 * energy refreshes (fallback when the energy recurrence could not prove a candidate). */
int asb_deflate_stats(asb_ctx* ctx, int64_t* n_panels, int64_t* n_refresh);
/* panels may append steps whose winner could not be proven in advance (bound on the vertices outside the candidate set
 * too stale); the projection pass then checks them against every vertex's energy and keeps the valid prefix, so the
 * selected sequence stays that of the reference loop (posComponents.py:75-77).  tried / kept: such steps in the last
 * run.  ASB_SPEC_PANELS=0 switches them off. */
int asb_deflate_spec_stats(asb_ctx* ctx, int64_t* tried, int64_t* kept);
/* reads of the snapshot tensor that the last asb_deflate_begin spent on the initial per-vertex energies ((R**2).sum of
 * posComponents.py:78-80 at k = 0): 0 when they came with the standardisation sweep (asb_snapshots_scale) or with an
 * earlier begin on the same, unchanged tensor; 1 otherwise.  ASB_E0_REUSE=0 always recomputes them. */
int asb_deflate_energy_passes(asb_ctx* ctx, int64_t* n_passes);
/* The panel's inner loop normally runs as ONE launch of co-resident blocks that exchange records through memory
 * (k_panel_coop).  If that exchange times out -- the blocks were not all resident because something else shares the GPU --
 * the panel is redone by the two-kernel loop and the context stays on it; *n = how often that happened (lifetime). */
int asb_deflate_coop_fallbacks(asb_ctx* ctx, int64_t* n);
/* First panel of a tensor whose energy sits largely (> 1/4) in the constant-in-time direction (rest shape "first" on
 * noise-like data): its candidates are guessed from the energies without that direction (a by-product of
 * asb_snapshots_scale) in addition to the few largest initial energies; the steps taken on the guess are unproven steps
 * like any others (asb_deflate_spec_stats), so the sequence is still that of posComponents.py:75-77.  *n = 1 if the last
 * run did so.  ASB_FIRST_PANEL_MEAN=0 switches it off. */
int asb_deflate_guessed_panels(asb_ctx* ctx, int64_t* n);
/* Structured data (every component removes a direction all vertices share): the energies at the start of a read no longer
 * say who wins a few steps later, and its pass rejects most of its unproven steps.  The coefficient columns of those rejected
 * steps are a rank-r sketch of every vertex's residual; a greedy replay of posComponents.py:76-96 in that space (asb_sketch.hip:
 * one thread per vertex, the sketch in registers) names the NEXT read's candidates.  Only the candidates change: every step is
 * still checked by the pass against every vertex outside them, so the selected sequence stays the reference's.
 * A replay runs only if the sketch holds at least ASB_SKETCH_MIN_SHARE (0.15) of the residual's energy -- on noise it holds r / F of it and
 * candidates by energy do as well; the kernel decides that itself (one exchange) and otherwise leaves the plain selection.
 * *runs = replays in the last run, *reads = reads of X whose candidates came from one.  ASB_SKETCH=0 switches it off. */
int asb_deflate_sketch_stats(asb_ctx* ctx, int64_t* runs, int64_t* reads);
/* Leaving the projection mode in mid-run.  The panel algorithm proves its winners against bounds kept in the energy recurrence;
 * where that cannot work -- K beyond the numerical rank of the data: the residual is rounding noise, nothing is provable and
 * every panel ends in an exact refresh, ~5 reads of X per component -- the reference's own loop (posComponents.py:76-96: keep R,
 * one read + one write per component) is the cheaper algorithm.  asb_project_switch_residual(ctx, k), k = components committed
 * so far: R_k = X - sum_{j<k} w_j (x) c_j is written out once, the per-component scalars are converted, and the context is in
 * residual mode from there on (asb_deflate_local_best / _pick / _apply / _results as if the run had begun in it).
 * asb_deflate_run_global does this by itself (single rank) once the reads of X of the last >= 8 committed fewer than 3/4 of a
 * component each (ASB_STALL_FALLBACK=0: never); several ranks: the driver applies the same rule (_panels.py).
 * asb_deflate_switch_stats: the component at which the last run switched, -1 if it did not. */
int asb_project_switch_residual(asb_ctx* ctx, int64_t k);
/* The multi-rank read of X in two calls and ONE exchange (round 4; what asb_deflate_run_global does inside the library on one
 * rank: posComponents.py:76-96 for up to 64 components per read).  Every rank holds the same assembled candidates
 * (asb_panel_assemble*).  asb_panel_read_run: all (<= nsub_max) sub-panels of the read in ONE launch of the panel kernel --
 * identical on every rank --, this shard's pass over X and the checks of all tiles behind it; words_dev (device, 10 doubles):
 * [ct] = columns of tile ct that stand on this shard (the tile's length if it was not reached: neutral under min), [8] = status
 * (0; -1: the kernel's exchange timed out here), [9] = the local counts again, packed, NOT to be reduced.  The driver
 * min-all-reduces words_dev[0 .. 8] over the ranks, reads all 10 words once (asb_fetch_doubles) and hands them to
 * asb_panel_read_commit: tiles stand in full while the minimum equals their length, the first one below keeps that many
 * columns, nothing behind it; a shard whose local chain ran ahead of the verdict rolls its energies back first.  With status
 * -1 the driver passes zeros (roll-back only), switches the kernel off on all ranks (asb_panel_set_coop) and repeats the panel. */
/* SPLOCS without a host read per outer iteration (posComponents.py:183-189 prints one line per iteration; nothing in the loop
 * depends on the printed numbers): asb_splocs_trace_begin sizes a device trace and defers the ADMM's status check,
 * asb_splocs_objective_dev leaves iteration it's <W,P>, <G,M>, sum Lambda |C_v| there, asb_splocs_trace reads all (n_its, 3) once. */
int asb_splocs_trace_begin(asb_ctx* ctx, int64_t n_its);
int asb_splocs_objective_dev(asb_ctx* ctx, const double* P_dev, const double* M_dev, int64_t it);
int asb_splocs_trace(asb_ctx* ctx, int64_t n_its, double* out_n_its_by_3);
int asb_panel_read_run(asb_ctx* ctx, int64_t k0, int64_t k1, int nsub_max, int spec_budget, const int* sub_budget8,
                       double* words_dev, int* ntile_out, int* nc_out8, int* proven_out8);
int asb_panel_read_commit(asb_ctx* ctx, const double* words10_host, int64_t* total_out, int* full_out, int* rejected_out);
int asb_fetch_doubles(asb_ctx* ctx, const double* dev, int n, double* out);
int asb_deflate_switch_stats(asb_ctx* ctx, int64_t* k_switch);
/* The same for the multi-rank driver (animsnapbases_amd/_panels.py).  asb_panel_guess_stats: this shard's energy along the
 * constant direction, its |X|^2 and whether the context could guess at all (0: both values are 0); the ranks sum all three
 * and guess only if every rank can and the share exceeds 1/4.  asb_panel_guess_begin (before the first panel's
 * asb_panel_hist): installs the score thresholds (each rank its share 1/world of the targets) and a smaller target for the
 * energies proper, which asb_panel_tau / asb_panel_global_tau / asb_panel_target then use; asb_panel_select takes the
 * union (the counts returned by asb_panel_global_tau do not include it: ask asb_panel_select for them);
 * asb_panel_project_spec* check against the same union.  asb_panel_guess_end (after that panel, whatever its outcome)
 * restores the plain selection. */
int asb_panel_guess_stats(asb_ctx* ctx, double* mean_energy_local, double* normx2_local, int* possible);
int asb_panel_guess_begin(asb_ctx* ctx, int world);
int asb_panel_guess_end(asb_ctx* ctx);
/* Several sub-panels per read of X, in steps, for the multi-rank driver (what asb_deflate_run does by itself on one rank):
 * asb_panel_sub_run      sub-panel sp = 0, 1, 2 of the read: up to `steps` greedy steps of the co-resident panel kernel on the
 *                        assembled candidates (sp = 0) or on the rows the previous sub-panel left behind (sp > 0), up to
 *                        spec_max of them unproven; *ran = -1: the kernel's record exchange timed out (as asb_panel_run_spec
 *                        in assembled mode: the driver redoes the read with the kernel off on every rank); *may_continue:
 *                        another sub-panel may follow (this one was full and ran in the co-resident kernel);
 * asb_panel_sub_project  ONE pass over the shard for the ntile sub-panels [k0 + 16 ct, + nc[ct]);
 * asb_panel_sub_check    tile ct against this shard's vertices: first rejected step (nc: none) into the caller's device
 *                        double -- min-all-reduce it over the ranks;
 * asb_panel_sub_commit   energies / column sums of the first `kept` columns of tile ct; tiles behind a tile that was not
 *                        kept in full are dropped by the driver. */
int asb_panel_sub_run(asb_ctx* ctx, int sp, int64_t k0, int steps, int spec_max, int64_t* ran, int64_t* proven, int* may_continue);
int asb_panel_sub_project(asb_ctx* ctx, int64_t k0, int ntile, const int* nc);
int asb_panel_sub_check(asb_ctx* ctx, int ct, int64_t kb, int nc, double* first_rejected_dev);
int asb_panel_sub_commit(asb_ctx* ctx, int ct, int64_t kb, int nc, int kept);
/* Multi-rank runs (assembled candidate buffer): a timed-out exchange is NOT redone locally -- the ranks must stay in
 * lock-step -- asb_panel_run / asb_panel_run_spec then return *committed = -1 with the kernel switched off for this context;
 * the driver min-reduces that over the ranks, switches it off everywhere (asb_panel_set_coop, returns the old setting) and
 * repeats the panel on every rank. */
int asb_panel_set_coop(asb_ctx* ctx, int on);
/* *out = *dev (one float64 in device memory, e.g. the min-all-reduced count of asb_panel_project_spec_dev) in stream order,
 * published into pinned host memory and polled there instead of a stream synchronisation */
int asb_fetch_double(asb_ctx* ctx, const double* dev, double* out);
/* the final residual in the reference layout (F, n_loc, 3) (R of :125) */
int asb_deflate_download_residual(asb_ctx* ctx, double* out);

/* ------------------------------------------------ post-processing ------------ */
/* posComponents.post_process_components, snapbases/posComponents.py:277-292, the
 * element-wise part, in place on the device-resident components:
 *   unscale != 0:  comps /= pre_scale_factor; comps += mean        (:279-282)
 *   invMassL != NULL (host, n_loc entries of this shard): comps *= invMassL[:,None] (:292)
 * comps_out (K, n_loc, 3) may be NULL. */
int asb_components_post(asb_ctx* ctx, int unscale, double pre_scale_factor,
                        const double* invMassL_loc, double* comps_out);

/* :284-287 `comps[:,:,l] = orth(comps[:,:,l].T).T` (scipy's SVD-based orth) per dimension,
 * as Gram (MFMA) -> K x K Jacobi eigen-solve -> U = A V S^-1, all on the device (K <= 128: parallel Jacobi in one
 * block's LDS; larger K: one-sided Jacobi on the rows of the Gram matrix, csrc/asb_smalldense.hip).
 * asb_orth_gram: this shard's three K x K Gram matrices into G_dev (3*K*K doubles, caller's
 * device buffer to be all-reduced over ranks) or into the context when NULL.
 * asb_orth_apply: finishes with the (summed) Gram matrices; sing_out (host, 3*K, optional).
 * Fails with ASB_ERR_NUMERIC when a slice is rank deficient (orth would drop vectors).
 * asb_orth_refine: second pass -- with the Gram matrices of the basis asb_orth_apply left (asb_orth_gram again,
 * summed), one symmetric Newton-Schulz step U <- U (1.5 I - 0.5 U^T U): squares the eps * cond^2 orthogonality
 * defect of the Gram route, so U^T U = I holds to round-off like scipy's SVD-based orth. */
int asb_orth_gram(asb_ctx* ctx, double* G_dev);
int asb_orth_apply(asb_ctx* ctx, const double* G_dev, double* sing_out);
int asb_orth_refine(asb_ctx* ctx, const double* G_dev);
/* K > 128 (the one-block eigen-solver / Cholesky of asb_orth_apply / asb_qr_apply stop there): the K x K step of
 * the orthogonalisation runs on the host.  asb_orth_gram_get: the three Gram matrices of asb_orth_gram (NULL buffer)
 * to the host; asb_components_transform: comps[:, :, l] <- sum_i comps_i T[l][i][j] with T host (3, K, K) --
 * T_l = V S^-1 (orth, :284-287) or L^-T (qr, constraintsComponents.py:431-435). */
int asb_orth_gram_get(asb_ctx* ctx, double* G_host);
int asb_components_transform(asb_ctx* ctx, const double* T_host);
/* geom_constructed, constraintsComponents.py:489-521, the large product: out (host, F x n_loc x 3),
 * out[f][e][l] = sum_{j < r} comps[j][e][l] coef[l][j][f] with coef host (3, r, F) (the interpolation coefficients of
 * every frame, solved by the caller from the r p x r p normal equations as the reference does) */
int asb_components_expand(asb_ctx* ctx, const double* coef_host, int64_t r, int64_t F, double* out_host);
/* the device-resident basis (K, n_loc, 3) to the host */
int asb_components_download(asb_ctx* ctx, double* comps_out);
/* ---- the weighted differential operator S^T of the constraint path (constraintsComponents.py:70-74: a scipy sparse matrix
 * read from an .npz; rows = position-space vertices, columns = the e p constraint rows).  One rank holds all constraint rows. */
int asb_st_upload(asb_ctx* ctx, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr_host,
                  const int64_t* indices_host, const double* data_host);
/* 'pca_blocks_with_St' (constraintsComponents.py:180): v = argmax_rows sum((S^T R_flat)^2) on the CURRENT residual of the
 * residual-mode deflation (R_flat = (e p) x (3 F)); first maximum */
int asb_st_residual_argmax(asb_ctx* ctx, int64_t* v_out, double* val_out);
/* |R|_F^2 of the residual-mode deflation right now (the loop condition `while norm(R) > tol`, :179, :252) */
int asb_deflate_residual_norm2(asb_ctx* ctx, double* out);
/* residual mode: room for K_new >= K components in all; W / comps / scalars grow, what the run produced so far is kept (loops
 * that end on a tolerance -- :179 `while norm(R) > tol` -- reserve a little and grow geometrically, as the reference's lists do) */
int asb_deflate_reserve(asb_ctx* ctx, int64_t K_new);
/* geom_block_form_utilizing_differential_operator(error_in_pos_space=True) (:652-672): residual of basis block k as
 * asb_deim_block_residual, mapped to position space by S^T ((|V|) x (3 p)); first arg-max of its squared row norms */
int asb_deim_block_residual_st(asb_ctx* ctx, int64_t k, int p, const double* coef_host, double* maxabs_out, int64_t* v_out,
                               double* val_out);
/* The basis into PINNED host memory, overlapped with the run that produces it (posComponents.py:119 leaves `comps` in host
 * memory: `self.comps = array(C)`).  asb_components_stream(ctx, 1) before asb_deflate_begin: the context keeps a pinned
 * (K, n_loc, 3) buffer and a copy stream; every component row is copied as soon as it is final (projection mode: after each
 * read of X, while the next read runs; otherwise at the end).  asb_components_pinned waits for the copies and returns the
 * buffer: valid until the next asb_deflate_begin / destroy on this context (the caller copies what it wants to keep).
 * asb_components_stream(ctx, 0) switches it off and frees the buffer.
 * asb_components_stream_into: the same into a pinned buffer the CALLER owns (`count` doubles >= K n_loc 3 of the runs that
 * follow, or asb_deflate_begin fails; NULL: off) -- for callers that hand the buffer on (the Python engine's ndarray views):
 * the context never frees it and stops writing to it with the next _into / asb_components_stream(ctx, 0) / asb_destroy.
 * asb_host_alloc / asb_host_free: pinned host memory for such a buffer (hipHostMalloc / hipHostFree; no context). */
int asb_components_stream(asb_ctx* ctx, int enable);
int asb_components_stream_into(asb_ctx* ctx, double* pinned_host, int64_t count);
int asb_host_alloc(int64_t count, double** out);
int asb_host_free(double* p);
int asb_components_pinned(asb_ctx* ctx, double** comps_pinned_out);
/* installs a caller-assigned basis (host, K x n_loc x 3) as the device-resident one */
int asb_components_upload(asb_ctx* ctx, const double* comps_host, int64_t K);

/* ------------------------------------------------ geodesics on the device (optional) ---- */
/* GeodesicDistanceComputation, utils/support.py:139-208, with the two SuperLU solves replaced by
 * Jacobi-PCG for up to 64 sources at a time.  Operators (host CSR, int32 indices), assembled by the caller
 * from the mesh: heat = A - tL, lap = -L (both n x n, SPD / SPSD), grad (m3 x n), div (n x m3), and the
 * diagonals of heat and lap. */
int asb_geodesic_setup(asb_ctx* ctx, int n, int m3, const int* heat_rp, const int* heat_ci, const double* heat_v,
                       const int* lap_rp, const int* lap_ci, const double* lap_v, const int* grad_rp,
                       const int* grad_ci, const double* grad_v, const int* div_rp, const int* div_ci,
                       const double* div_v, const double* heat_diag, const double* lap_diag);
/* Two-level preconditioner for the sparse (PCG) mode -- what lets it scale past the dense mode's 46 000 vertices: agg (n) =
 * aggregate (0 .. nc-1) of every vertex, agg_ptr / agg_mem its CSR form, heat_c / lap_c (host, nc x nc, SPD) the coarse
 * operators P^T (A - tL) P and P^T (-L) P + gauge with P the piecewise-constant prolongation; they are inverted on the
 * device and every PCG step of the POISSON system adds P Ac^-1 P^T r to the Jacobi step.  The HEAT system is solved by
 * (damped, heat_omega in (0, 1]) Jacobi sweeps from zero instead: its solution spans many orders of magnitude and the
 * method needs every component to relative accuracy, which a sweep that only adds non-negative terms delivers and a Krylov
 * method does not.  Quasi-uniform meshes only (the sweep's rate is 1 - min area / (area + t sum w)); otherwise the solve
 * fails with ASB_ERR_NUMERIC and the host backend is the way. */
int asb_geodesic_coarse_setup(asb_ctx* ctx, int nc, const int* agg, const int* agg_ptr, const int* agg_mem,
                              const double* heat_c, const double* lap_c, double heat_omega);
/* distances (min-shifted, :206) from nsrc <= 64 sources: out host (nsrc, n); tol = relative residual of the
 * CG solves; iters (optional, 2 ints) = iterations of the heat and the Poisson solve */
int asb_geodesic_solve(asb_ctx* ctx, const int64_t* sources, int nsrc, double tol, double* out, int* iters);
/* Dense mode (after asb_geodesic_setup): the counterpart of the reference's two SuperLU factorisations
 * (utils/support.py:170-171) for meshes whose N x N matrices are cheap in HBM (N <= 46 000: 2 x 17 GB).  Inverts
 * (A - tL) and the gauge-fixed -L once on the device (blocked Gauss-Jordan on f64 MFMA); asb_geodesic_solve then
 * needs no iteration: the heat step is a column gather, the Poisson step one dense product.  tol / iters unused. */
int asb_geodesic_dense_setup(asb_ctx* ctx);
/* Slab mode (round 4): a DIRECT factorisation of both systems as block-tridiagonal matrices over breadth-first slabs of the mesh
 * graph -- what the reference's two splu factorisations (utils/support.py:170-171) are for meshes beyond the dense inverses'
 * 46 000 vertices and for badly graded ones (no iteration, no convergence question).  slab_ptr (nslab + 1): slab boundaries in a
 * PERMUTED numbering in which every edge joins vertices of the same or of adjacent slabs; perm_of_vertex (n): vertex ->
 * permuted index; heat / lap: A - tL and -L as CSR in that numbering (host). */
int asb_geodesic_bt_setup(asb_ctx* ctx, int nslab, const int* slab_ptr, const int* perm_of_vertex, const int* heat_rp,
                          const int* heat_ci, const double* heat_v, const int* lap_rp, const int* lap_ci, const double* lap_v);
/* support='local' step without a host round trip (posComponents.py:87-105, dense geodesics): reads the vertex
 * asb_deflate_pick chose for component k on the device, solves its distance field, forms
 * s = 1 - (clip(phi, dmin, dmax) - dmin) / (dmax - dmin) (:61-64) for this shard and applies the deflation. */
int asb_deflate_apply_geodesic(asb_ctx* ctx, int64_t k, double dmin, double dmax);
/* Distance fields kept on the device for SPLOCS (posComponents.py:158-165 asks for the fields of the K centres in every
 * outer iteration): solves the fields of nsrc (<= 64) new sources and appends them to the context's cache (4096 slots);
 * *slot0 = slot of the first.  ASB_ERR_LIMIT when full; asb_geodesic_cache_clear empties the cache (buffers stay). */
int asb_geodesic_cache_add(asb_ctx* ctx, const int64_t* sources, int nsrc, double tol, int64_t* slot0);
int asb_geodesic_cache_clear(asb_ctx* ctx);

/* ------------------------------------------------ snapshot ingest --------------- */
/* align, utils/process.py:235-250 (find_rbm_procrustes :210-234 + transform :196-208 per frame): every
 * frame is moved onto frame 0 by the rigid-body motion of the orthogonal Procrustes problem.
 * frames: host (F, N, 3) float64, overwritten; T_out: optional host (F, 4, 4) matrices. */
int asb_align_frames(asb_ctx* ctx, double* frames, int64_t F, int64_t N, int rigid, double* T_out);

/* ------------------------------------------------ constraint-projection bases (config 5) ---- */
/* compute_pod_for_vectorized_nonlinear_snapshots_tensor, snapbases/constraintsComponents.py:298-320:
 * the reference takes svd(A), A = (3ep x F).  asb_pod_gram: G = A^T A (F x F) of this shard (f64 MFMA)
 * into G_dev (caller's device buffer, all-reduced over ranks by the caller) and/or G_host.
 * The F x F eigen-problem is solved by the caller (LAPACK); asb_pod_basis then forms the K leading left
 * vectors  comps[i] = A V[:, i] / sigma[i]  (V host F x K, sigma host K) as the device-resident basis. */
int asb_pod_gram(asb_ctx* ctx, double* G_dev, double* G_host);
int asb_pod_basis(asb_ctx* ctx, const double* V, const double* sigma, int64_t K);
/* Rayleigh-Ritz refinement of the POD basis: B = Q^T A (K x F, this shard's partial sum) for the device-resident
 * (orthonormalised) basis Q, into B_dev (to be all-reduced) and/or B_host.  The SVD of the small B, taken from A
 * itself, restores the eps * sigma_0 / sigma_k accuracy of the reference's `svd` (:307) that the Gram route
 * (eps * (sigma_0 / sigma_k)^2) loses on weak components; the rotation is applied with asb_components_transform. */
int asb_pod_project(asb_ctx* ctx, double* B_dev, double* B_host);
/* keeps the first K components of the device-resident basis (drops the oversampling vectors of the refinement) */
int asb_components_truncate(asb_ctx* ctx, int64_t K);

/* The F x F symmetric eigen-problem of the POD (the `svd` of constraintsComponents.py:307 in Gram form) on the
 * device.  asb_sym_tridiag: Householder tridiagonalisation A = Q T Q^T of the n x n symmetric matrix A_dev
 * (row-major, both triangles; NULL = the Gram matrix asb_pod_gram left in the context).  A is overwritten with
 * the reflectors; d_host (n) / e_host (n-1) receive T's diagonal / off-diagonal.  Ordered reductions only:
 * bit-identical on every rank.  The caller solves the tridiagonal problem (O(n^2)).
 * asb_sym_backtransform: V = Q Z for k eigenvectors Z of T (host, n x k row-major) -> V_host (n x k). */
int asb_sym_tridiag(asb_ctx* ctx, double* A_dev, int64_t n, double* d_host, double* e_host);
int asb_sym_backtransform(asb_ctx* ctx, const double* A_dev, int64_t n, const double* Z_host, int64_t k, double* V_host);
/* The same eigen-problem with NOTHING on the host (n >= 3): tridiagonalisation, then all eigenvalues of T by bisection
 * on Sturm counts and its k leading eigenvectors by inverse iteration (csrc/asb_smalldense.hip), then the
 * back-transformation.  lam_host (n): all eigenvalues, descending; the k vectors stay on the device for
 * asb_pod_basis_dev (V_host, optional n x k row-major, receives a copy); *n_bad (optional): inverse iterations that missed
 * their growth criterion.  Vectors of eigenvalues closer than ~eps |T| / 1e-5 are accurate as a SPAN, not one by one
 * (no re-orthogonalisation inside clusters): the POD follows with a Rayleigh-Ritz step on the snapshot matrix. */
int asb_sym_eig_topk(asb_ctx* ctx, double* A_dev, int64_t n, int64_t k, double* lam_host, double* V_host, int64_t* n_bad);
/* asb_pod_basis from the eigen-pairs asb_sym_eig_topk left on the device: comps[i] = A V[:, i] / sqrt(lam[i]), i < K */
int asb_pod_basis_dev(asb_ctx* ctx, int64_t K);
/* Rayleigh-Ritz rotation on the device: singular values (S_host, K, descending) and left vectors U_B of the (all-reduced)
 * K x F matrix B = Q^T A (B_dev, or the context's from asb_pod_project; overwritten) by one-sided Jacobi on its rows;
 * basis <- Q U_B.  Replaces the small host SVD between asb_pod_project and asb_components_transform. */
int asb_pod_rotate(asb_ctx* ctx, double* B_dev, double* S_host);
/* One step of subspace iteration from the Rayleigh-Ritz result: basis <- A V Sigma^-1 with the right Ritz vectors asb_pod_rotate
 * left in B (even K, F and row count).  Followed by CholeskyQR2 + asb_pod_project + asb_pod_rotate again it removes most of what the
 * Gram route's eps (sigma_0 / sigma_k)^2 leaves OUTSIDE the Ritz subspace of the weak vectors (constraintsComponents.py:307: the
 * reference's gesdd has no such loss). */
int asb_pod_power(asb_ctx* ctx, const double* B_dev);
/* The POD in LEVELS: singular values below ~1e-8 sigma_0 are invisible in the Gram matrix of A (where the reference's gesdd on A
 * itself still returns vectors, constraintsComponents.py:307-316) but not in the Gram matrix of A_2 = A - U_1 (U_1^T A).
 * asb_pod_deflate_begin(ctx, B_dev, keep): the first `keep` rows of the basis are kept (behind what earlier levels kept) and the
 * context's snapshots become A_2 (a second buffer; even `keep`, F and row count); the POD entry points then work on A_2.
 * asb_pod_deflate_end(ctx, last): the snapshots are the original ones again, the basis is [kept rows of all levels ; the first
 * `last` rows of the current basis]. */
int asb_pod_deflate_begin(asb_ctx* ctx, const double* B_dev, int64_t keep);
int asb_pod_deflate_end(asb_ctx* ctx, int64_t last);
/* constProj_basis_type 'pod' (compute_pod_for_nonlinear_snapshots_tensor, :274-294): one SVD per (constraint row, coordinate)
 * slice of the snapshots -- e x F matrices -- by Gram matrix + device eigen-solver; the K leading left vectors of every
 * slice become the device-resident basis (K, e p, 3).  The reference does this in float32 with torch on the CPU. */
int asb_pod_slices(asb_ctx* ctx, int p, int64_t K);
/* asb_qr_apply with ONE Cholesky factor of the sum of the three Gram matrices for all three slices: the basis becomes
 * orthonormal as (3 n)-vectors (the Q of the Rayleigh-Ritz step).  Any K. */
int asb_qr_apply_joint(asb_ctx* ctx, const double* G_dev);
/* :421-428 / :440-443 the reference also restores the snapshot tensor:
 * X <- (X * inv_scale + mean) * rowscale[v]   (rowscale host n_loc or NULL) */
int asb_snapshots_affine(asb_ctx* ctx, double inv_scale, int add_mean, const double* rowscale);
/* :430-433 `qr(comps[:,:,l].T, mode='economic')[0].T`: one CholeskyQR pass per dimension with the
 * Gram matrices of asb_orth_gram (NULL: the context's).  Call gram/apply twice (CholeskyQR2).  K <= 128: one block's
 * LDS; larger K (the reference's configurations use 200 ... 1000): blocked Cholesky + inverse, csrc/asb_smalldense.hip. */
int asb_qr_apply(asb_ctx* ctx, const double* G_dev);
/* deim, :797-860, device half: residual r = V[:, :k] coef - v_k per dimension and its arg-max over this
 * shard (global row index).  coef: host (3, k), NULL for k = 0.  The k x k interpolation solves stay with
 * the caller (numpy lstsq, as in the reference :829), fed by asb_deim_row. */
int asb_deim_step(asb_ctx* ctx, int64_t k, const double* coef, int64_t* idx_out, double* val_out);
/* deim (:797-860) entirely on the device, for a basis whose rows are all on this rank: Pt_out (K) = the interpolation rows in
 * order; maxabs_out (K) = the largest |residual| entry of each step (the reference stops with "zero residual" when
 * np.allclose(r, 0), i.e. <= 1e-8); the k x k systems are solved through a bordered inverse carried on the device and
 * verified -- *solve_failed != 0 means a check failed and the caller should fall back to asb_deim_step + lstsq (:829). */
int asb_deim_run(asb_ctx* ctx, int64_t* Pt_out, double* maxabs_out, int* solve_failed);
/* deim_blocksForm (:733-795) / geom_block_form_utilizing_differential_operator (:619-731, error in the constraint space):
 * residual of block k (p basis vectors) with the interpolation coefficients coef (host, 3 x (k p) x p; NULL at k = 0);
 * per-row energies into the context (arg-max over rows / constraints: asb_deflate_block_argmax with group 1 / p);
 * *maxabs_out = largest |r| entry (np.allclose(r, 0) test of :771). */
int asb_deim_block_residual(asb_ctx* ctx, int64_t k, int p, const double* coef, double* maxabs_out);
/* first maximum over this shard of the sums of p consecutive per-row energies held by the context (global block index) */
int asb_energy_block_argmax(asb_ctx* ctx, int p, int64_t* block_out, double* val_out);
/* V[gidx, :, :] -> row_out (K, 3); returns 1 (and writes nothing) when another rank owns gidx */
int asb_deim_row(asb_ctx* ctx, int64_t gidx, double* row_out);

/* ------------------------------------------------ SPLOCS refinement ----------- */
/* posComponents.splocs_glob_optimization, snapbases/posComponents.py:132-189.
 * State after a residual-mode deflation: C = comps, W = weigs, U = 0 (:135-139).  One outer
 * iteration = weights -> (host support maps) -> admm -> gram -> objective.  The F x 3N
 * residual is never formed (see csrc/asb_splocs.hip).  K <= 128. */
int asb_splocs_begin(asb_ctx* ctx);
/* P = X C^T (F x K) and M = C C^T (K x K) of this shard for the CURRENT C, into the
 * caller's device buffers (to be all-reduced over ranks) or into the context when NULL.
 * normX2_local (optional): |X|^2 of the shard. */
int asb_splocs_gram(asb_ctx* ctx, double* P_dev, double* M_dev, double* normX2_local);
/* :144-156 weight sweep with the (all-reduced) P, M (NULL: the context's own), G = W^T W;
 * centre_idx/val (K): per component this shard's vertex of largest |C_k[v]|^2 (:161). */
int asb_splocs_weights(asb_ctx* ctx, const double* P_dev, const double* M_dev,
                       int64_t* centre_idx, double* centre_val);
/* :167-181 ADMM with Lambda (host, K x n_loc) = splocs_lambda * support_map; C = Z at the end */
int asb_splocs_admm(asb_ctx* ctx, const double* Lambda, double rho, int n_iter);
/* the same step with Lambda built on the device: Lambda[k] = lambda * (clip(phi_k, dmin, dmax) - dmin) / (dmax - dmin)
 * (:162-165, utils/support.py:61-64) from the cached distance field in slot slots[k] (K slots, host) */
int asb_splocs_admm_fields(asb_ctx* ctx, const int64_t* slots, double lambda, double dmin, double dmax, double rho, int n_iter);
/* :183-186 pieces of the objective for the new C (call asb_splocs_gram first):
 * wp = <W, P>, gm = <W^T W, M>  =>  |X - W C|^2 = |X|^2 - 2 wp + gm;
 * sparsity_local = sum Lambda |C_v| over the shard. */
int asb_splocs_objective(asb_ctx* ctx, const double* P_dev, const double* M_dev, double* wp,
                         double* gm, double* sparsity_local);
/* refined components (K, n_loc, 3) and weights (F, K); either may be NULL */
int asb_splocs_results(asb_ctx* ctx, double* C_out, double* W_out);

/* ------------------------------------------------ host-side probes ----------- */
/* the small dense solvers of csrc/asb_smalldense.hip on host arrays (tests):
 * tridiagonal (d, e) -> all eigenvalues descending + k leading unit eigenvectors Z (n x k row-major);
 * one-sided Jacobi on the rows of A (nv x m): singular values descending, left vectors as COLUMNS of U (nv x nv, optional);
 * Tt[c][i] = (L^-1)[i][c] of the Cholesky factor G = L L^T (any K). */
int asb_test_tridiag_eig(asb_ctx* ctx, const double* d, const double* e, int64_t n, int64_t k, double* lam_desc, double* Z,
                         int64_t* n_bad);
int asb_test_jacobi_rows(asb_ctx* ctx, const double* A, int64_t nv, int64_t m, double* U, double* sig, int64_t* sweeps);
int asb_test_chol_tinv(asb_ctx* ctx, const double* G, int64_t K, double* Tt);
/* timing probe of the multi-tile projection kernel on the context's tensor in projection mode (component storage is
 * overwritten): nct tiles (2..4); mode 0 = as it runs, 1 = X operand from four cache-resident tiles, 2 = without the MFMAs;
 * best of `reps` launches in ms.  tools/probe_l2w.py. */
int asb_test_l2w_probe(asb_ctx* ctx, int nct, int mode, int reps, double* ms_out);
/* The 3x3 symmetric eigen-solver used by asb_deflate_pick, run on the HOST (unit test
 * without a GPU).  a6 = (a00,a01,a02,a11,a12,a22); out4 = (lambda_max, u0, u1, u2). */
void asb_test_eig3(const double* a6, double* out4);
/* tests: the sketch replay on host arrays -- cols (r x 3 n: column i, entry 3 v + d, the coefficient of vertex v's row d on the
 * unit direction i divided by sqrt(wn2[i])), wn2 (r), exact energies E (n) -> scores (n; max over the steps of energy / winner's
 * energy), the replay's winners pred (steps; -1 behind its end), *status = 1 (0: the exchange timed out, scores = energies) */
int asb_test_sketch_predict(asb_ctx* ctx, const double* cols, const double* wn2, const double* E, int64_t n, int r, int steps,
                            double* scores, int64_t* pred, int* status);
/* test hook: inverse of a host symmetric positive definite matrix (n x n) through the device's blocked
 * Gauss-Jordan / f64-MFMA GEMM path that the device geodesics use for their two SPD systems */
int asb_test_spd_inverse(asb_ctx* ctx, const double* A_host, int64_t n, double* Ainv_host);

#ifdef __cplusplus
}
#endif
#endif /* ASB_H */
