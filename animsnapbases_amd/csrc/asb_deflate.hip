// Residual-mode greedy deflation: host side + C ABI -- posComponents.extract_k_components,
// snapbases/posComponents.py:67-122 of the reference.  gfx950 (MI355X) only.
#include "asb_kernels.h"

// Host-callable probe of the eigen-solver (CPU unit test, no GPU needed).
// a6 = (a00, a01, a02, a11, a12, a22); out4 = (lambda_max, u0, u1, u2).
extern "C" void asb_test_eig3(const double* a6, double* out4) {
    eig3_top(a6[0], a6[1], a6[2], a6[3], a6[4], a6[5], out4[0], out4[1], out4[2], out4[3]);
}

// --------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------
// one streaming pass over the residual; leaves ctx->nblk partial records
static int stream_pass(asb_ctx* ctx, bool update, const double* wk, double* scal_k, const double* s,
                       double* ck) {
    StreamCfg c;
    if (!pick_cfg(ctx->Fp, c)) ASB_FAIL(ctx, ASB_ERR_LIMIT, "F = %lld too large (max 32768)", (long long)ctx->F);
    const int grid = stream_grid(ctx, c, ctx->n_loc);
    StreamArgs a{ctx->R, wk, scal_k, s, ck, ctx->energy, ctx->pmax, ctx->pidx, ctx->psum, (long long)ctx->n_loc, nullptr};
    size_t slot;
    int rc = prof_begin(ctx, slot);
    if (rc) return rc;
    launch_stream(ctx, c, update, grid, a);
    ASB_CHECK_LAUNCH(ctx);
    rc = prof_end(ctx, slot);
    if (rc) return rc;
    ctx->nblk = grid;
    return ASB_OK;
}

extern "C" int asb_deflate_begin(asb_ctx* ctx, int64_t K, int mode, int local_support) {
    if (!ctx) return ASB_ERR_ARG;
    if (!ctx->X) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_deflate_begin: no snapshots uploaded");
    if (K < 1) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_deflate_begin: K = %lld", (long long)K);
    if (mode != ASB_DEFLATE_RESIDUAL && mode != ASB_DEFLATE_PROJECT)
        ASB_FAIL(ctx, ASB_ERR_ARG, "asb_deflate_begin: unknown mode %d", mode);
    if (mode == ASB_DEFLATE_PROJECT && local_support)
        ASB_FAIL(ctx, ASB_ERR_ARG, "asb_deflate_begin: the projection mode is only valid for global support");
    ASB_HIP(ctx, hipSetDevice(ctx->dev));
    ctx->K = K;
    ctx->mode = mode;
    ctx->local = local_support;
    ctx->k_done = 0;
    ctx->n_panels = ctx->n_refresh = 0;
    ctx->n_spec_steps = ctx->n_spec_kept = 0;
    ctx->n_guess_panels = 0;
    ctx->n_sketch_runs = ctx->n_sketch_reads = 0;
    if (ctx->sk_counts) ASB_HIP(ctx, hipMemsetAsync(ctx->sk_counts, 0, 4 * sizeof(unsigned), ctx->stream));
    ctx->sketch_valid = false;
    ctx->sketch_run_off = false;
    ctx->read_by_score = ctx->last_by_score = false;
    ctx->rate_plain = ctx->rate_sketch = -1.0;
    ctx->mode_streak = 0;
    ctx->probe_after = 2;
    ctx->spec_budget = ASB_PANEL_COLS;
    ctx->fb_mark_reads = ctx->fb_mark_k = 0;
    ctx->k_switch = -1;
    {
        const int rcd = asb_dl_begin(ctx);          // overlapped download of the basis (asb_components_stream): a new run
        if (rcd) return rcd;
    }
    if (mode == ASB_DEFLATE_PROJECT) return asb_project_begin(ctx, K);
    const size_t rows = (size_t)ctx->n_loc * 3;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->R, rows * ctx->Fp))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->energy, (size_t)ctx->n_loc))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->W, (size_t)K * ctx->Fp))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->comps, (size_t)K * rows))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->scal, (size_t)(K + 1) * 4))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->xrec, (size_t)(2 + 3 * ctx->Fp)))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->s_dev, (size_t)ctx->n_loc))) return rc;
    ASB_HIP(ctx, hipMemsetAsync(ctx->scal, 0, (size_t)(K + 1) * 4 * sizeof(double), ctx->stream));
    ASB_HIP(ctx, hipMemcpyAsync(ctx->R, ctx->X, rows * ctx->Fp * sizeof(double), hipMemcpyDeviceToDevice,
                                ctx->stream));
    // initial energies (:78-80 for k = 0)
    return stream_pass(ctx, false, nullptr, nullptr, nullptr, nullptr);
}

// Residual mode: room for K_new components in all (W, comps, scal grow, what the run has produced so far is kept) -- for loops
// that end on a tolerance and cannot say beforehand how many components they take ('pca_blocks_with_St',
// constraintsComponents.py:179: `while norm(R) > tol`; the reference appends to Python lists).
template <typename T>
static int grow_keep(asb_ctx* ctx, T** p, size_t count_new, size_t count_keep) {
    const size_t want = count_new * sizeof(T);
    auto it = ctx->alloc_bytes.find((void*)p);
    if (*p && it != ctx->alloc_bytes.end() && it->second >= want) return ASB_OK;
    T* q = nullptr;
    ASB_HIP(ctx, hipMalloc((void**)&q, want));
    if (*p && count_keep) {
        hipError_t e = hipMemcpyAsync(q, *p, count_keep * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            (void)hipFree(q);
            ASB_FAIL(ctx, ASB_ERR_HIP, "asb_deflate_reserve: %s", hipGetErrorString(e));
        }
    }
    if (*p) (void)hipFree(*p);
    *p = q;
    ctx->alloc_bytes[(void*)p] = want;
    return ASB_OK;
}
extern "C" int asb_deflate_reserve(asb_ctx* ctx, int64_t K_new) {
    if (!ctx || !ctx->R || !ctx->W || !ctx->comps || !ctx->scal) return ASB_ERR_ARG;
    if (ctx->mode != ASB_DEFLATE_RESIDUAL) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_deflate_reserve needs the residual mode");
    if (K_new <= ctx->K) return ASB_OK;
    const size_t rows = (size_t)ctx->n_loc * 3, kd = (size_t)ctx->k_done, Kn = (size_t)K_new;
    int rc;
    if ((rc = grow_keep(ctx, &ctx->W, Kn * ctx->Fp, kd * ctx->Fp))) return rc;
    if ((rc = grow_keep(ctx, &ctx->comps, Kn * rows, kd * rows))) return rc;
    const size_t old_scal = (size_t)(ctx->K + 1) * 4;
    if ((rc = grow_keep(ctx, &ctx->scal, (Kn + 1) * 4, old_scal))) return rc;
    ASB_HIP(ctx, hipMemsetAsync(ctx->scal + old_scal, 0, ((Kn + 1) * 4 - old_scal) * sizeof(double), ctx->stream));
    ctx->K = K_new;
    if (ctx->dl_enabled) {          // (the streamed download is sized per run: a grown run copies at the end through the plain path)
        ctx->dl_done = 0;
    }
    return ASB_OK;
}

extern "C" int64_t asb_deflate_xchg_len(const asb_ctx* ctx) { return ctx ? 2 + 3 * ctx->Fp : 0; }

extern "C" int asb_deflate_local_best(asb_ctx* ctx, int64_t k, double* rec_dev) {
    if (!ctx || !ctx->R) return ASB_ERR_ARG;
    if (!rec_dev) rec_dev = ctx->xrec;
    hipLaunchKernelGGL(k_local_best, dim3(1), dim3(256), 0, ctx->stream, ctx->R, ctx->pmax, ctx->pidx, ctx->psum,
                       ctx->nblk, (long long)ctx->v0, (int)ctx->Fp, rec_dev, ctx->scal, (long long)k, (long long)ctx->forced_row,
                       (long long)ctx->n_loc);
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

extern "C" int asb_deflate_pick(asb_ctx* ctx, int64_t k, const double* recs_dev, int64_t n_rec) {
    if (!ctx || !ctx->R) return ASB_ERR_ARG;
    if (k < 0 || k > ctx->K) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_deflate_pick: k = %lld out of range", (long long)k);
    hipLaunchKernelGGL(k_pick, dim3(1), dim3(256), 0, ctx->stream, ctx->R, ctx->pmax, ctx->pidx, ctx->psum,
                       ctx->nblk, recs_dev, (int)n_rec, (long long)(2 + 3 * ctx->Fp), (long long)ctx->v0,
                       (int)ctx->F, (int)ctx->Fp, ctx->W, ctx->scal, (long long)k, (long long)ctx->K, ctx->local,
                       (PanelState*)nullptr, (const long long*)nullptr, (long long)0,
                       (long long)(recs_dev ? -1 : ctx->forced_row));
    ASB_CHECK_LAUNCH(ctx);
    ctx->forced_row = -1;          // a forced row holds for one pick
    return ASB_OK;
}

// 'pca_blocks' (constraintsComponents.py:324-412): the next asb_deflate_local_best / asb_deflate_pick takes the slab of
// global row `gidx` instead of the arg-max (residual mode).
extern "C" int asb_deflate_force_next(asb_ctx* ctx, int64_t gidx) {
    if (!ctx || !ctx->R) return ASB_ERR_ARG;
    if (ctx->mode != ASB_DEFLATE_RESIDUAL) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_deflate_force_next needs the residual mode");
    if (gidx < 0 || gidx >= ctx->N_glob) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_deflate_force_next: row %lld out of range", (long long)gidx);
    if (ctx->N_glob == ctx->n_loc && (gidx < ctx->v0 || gidx >= ctx->v0 + ctx->n_loc))
        ASB_FAIL(ctx, ASB_ERR_ARG, "asb_deflate_force_next: row %lld is not on this shard", (long long)gidx);
    ctx->forced_row = gidx;
    return ASB_OK;
}

static int energy_block_argmax(asb_ctx* ctx, int p, int64_t* block_out, double* val_out);
extern "C" int asb_deflate_block_argmax(asb_ctx* ctx, int p, int64_t* block_out, double* val_out) {
    if (!ctx || !ctx->R || !ctx->energy || !block_out || p < 1) return ASB_ERR_ARG;
    if (ctx->mode != ASB_DEFLATE_RESIDUAL) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_deflate_block_argmax needs the residual mode");
    return energy_block_argmax(ctx, p, block_out, val_out);
}
// the same on whatever per-row energies the context holds (asb_deim_block_residual)
extern "C" int asb_energy_block_argmax(asb_ctx* ctx, int p, int64_t* block_out, double* val_out) {
    if (!ctx || !ctx->energy || !block_out || p < 1) return ASB_ERR_ARG;
    return energy_block_argmax(ctx, p, block_out, val_out);
}
static int energy_block_argmax(asb_ctx* ctx, int p, int64_t* block_out, double* val_out) {
    if (ctx->v0 % p || ctx->n_loc % p)
        ASB_FAIL(ctx, ASB_ERR_ARG, "asb_deflate_block_argmax: the shard [%lld, +%lld) does not hold whole blocks of %d rows",
                 (long long)ctx->v0, (long long)ctx->n_loc, p);
    const long long nb = ctx->n_loc / p;
    const int grid = (int)((nb + 255) / 256 < 256 ? (nb + 255) / 256 : 256);
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->bam_val, (size_t)256))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->bam_idx, (size_t)256))) return rc;
    double hv[256];
    long long hi[256];
    if (nb > 0) {
        hipLaunchKernelGGL(k_block_argmax, dim3(grid), dim3(256), 0, ctx->stream, ctx->energy, nb, p, (long long)(ctx->v0 / p),
                           ctx->bam_val, ctx->bam_idx);
        ASB_CHECK_LAUNCH(ctx);
        ASB_HIP(ctx, hipMemcpyAsync(hv, ctx->bam_val, grid * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        ASB_HIP(ctx, hipMemcpyAsync(hi, ctx->bam_idx, grid * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    double be = -1.0;
    long long bi = 0x7fffffffffffffffLL;
    for (int b = 0; b < (nb > 0 ? grid : 0); ++b)
        if (hv[b] > be || (hv[b] == be && hi[b] < bi)) { be = hv[b]; bi = hi[b]; }
    *block_out = bi;
    if (val_out) *val_out = be;
    return ASB_OK;
}

extern "C" int asb_deflate_get_pick(asb_ctx* ctx, int64_t k, int64_t* idx, double* sigma) {
    if (!ctx || !ctx->scal || k < 0 || k >= ctx->K) return ASB_ERR_ARG;
    double h[4];
    ASB_HIP(ctx, hipMemcpyAsync(h, ctx->scal + k * 4, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (idx) memcpy(idx, &h[2], 8);
    if (sigma) *sigma = h[0];
    return ASB_OK;
}

extern "C" int asb_deflate_apply(asb_ctx* ctx, int64_t k, const double* s) {
    if (!ctx || !ctx->R) return ASB_ERR_ARG;
    if (k < 0 || k >= ctx->K) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_deflate_apply: k = %lld out of range", (long long)k);
    const double* s_dev = nullptr;
    if (s) {
        ASB_HIP(ctx, hipMemcpyAsync(ctx->s_dev, s, (size_t)ctx->n_loc * sizeof(double), hipMemcpyHostToDevice,
                                    ctx->stream));
        s_dev = ctx->s_dev;
    }
    int rc = stream_pass(ctx, true, ctx->W + k * ctx->Fp, ctx->scal + k * 4, s_dev,
                         ctx->comps + (size_t)k * 3 * ctx->n_loc);
    if (rc) return rc;
    ctx->k_done = k + 1;
    return ASB_OK;
}

// the same with the support map already on the device (asb_deflate_apply_geodesic)
int asb_deflate_apply_dev(asb_ctx* ctx, int64_t k, const double* s_dev) {
    if (!ctx || !ctx->R) return ASB_ERR_ARG;
    if (k < 0 || k >= ctx->K) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_deflate_apply: k = %lld out of range", (long long)k);
    int rc = stream_pass(ctx, true, ctx->W + k * ctx->Fp, ctx->scal + k * 4, s_dev, ctx->comps + (size_t)k * 3 * ctx->n_loc);
    if (rc) return rc;
    ctx->k_done = k + 1;
    return ASB_OK;
}

extern "C" int asb_deflate_run_global(asb_ctx* ctx, int64_t k0, int64_t k1) {
    if (!ctx || !ctx->W) return ASB_ERR_ARG;
    if (ctx->local) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_deflate_run_global: context is in local-support mode");
    if (k0 < 0 || k1 > ctx->K || k0 > k1) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_deflate_run_global: bad range");
    if (ctx->mode == ASB_DEFLATE_PROJECT) return asb_project_run(ctx, k0, k1);
    for (int64_t k = k0; k < k1; ++k) {
        int rc = asb_deflate_pick(ctx, k, nullptr, 0);
        if (rc) return rc;
        rc = asb_deflate_apply(ctx, k, nullptr);
        if (rc) return rc;
    }
    return ASB_OK;
}

extern "C" int asb_deflate_results(asb_ctx* ctx, double* comps, double* weigs, int64_t* idx, double* sigma,
                                   double* normR2_local) {
    if (!ctx || !ctx->W) return ASB_ERR_ARG;
    const int64_t K = ctx->K;
    if (ctx->mode == ASB_DEFLATE_PROJECT) return asb_project_results(ctx, comps, weigs, idx, sigma, normR2_local);
    // finalise the local norm of the last finished component (single-rank path reads partials)
    hipLaunchKernelGGL(k_pick, dim3(1), dim3(256), 0, ctx->stream, ctx->R, ctx->pmax, ctx->pidx, ctx->psum,
                       ctx->nblk, (const double*)nullptr, 0, (long long)(2 + 3 * ctx->Fp), (long long)ctx->v0,
                       (int)ctx->F, (int)ctx->Fp, ctx->W, ctx->scal, (long long)ctx->k_done, (long long)0, 0,
                       (PanelState*)nullptr, (const long long*)nullptr, (long long)0, (long long)-1);
    ASB_CHECK_LAUNCH(ctx);
    std::vector<double> h((size_t)(K + 1) * 4);
    ASB_HIP(ctx, hipMemcpyAsync(h.data(), ctx->scal, h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (comps)
        ASB_HIP(ctx, hipMemcpyAsync(comps, ctx->comps, (size_t)K * 3 * ctx->n_loc * sizeof(double),
                                    hipMemcpyDeviceToHost, ctx->stream));
    std::vector<double> hw;
    if (weigs) {
        hw.resize((size_t)K * ctx->Fp);
        ASB_HIP(ctx, hipMemcpyAsync(hw.data(), ctx->W, hw.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t k = 0; k < K; ++k) {
        if (sigma) sigma[k] = h[k * 4 + 0];
        if (idx) memcpy(&idx[k], &h[k * 4 + 2], 8);
        if (normR2_local) normR2_local[k] = h[k * 4 + 3];
    }
    if (weigs)
        for (int64_t f = 0; f < ctx->F; ++f)
            for (int64_t k = 0; k < K; ++k) weigs[f * K + k] = hw[(size_t)k * ctx->Fp + f];
    return ASB_OK;
}

// --------------------------------------------------------------------------------------
// element-wise post-processing of the components (posComponents.py:279-292)
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_comps_post(double* __restrict__ comps, long long K, long long rows,
                                                    int unscale, double psf, const double* __restrict__ mean,
                                                    const double* __restrict__ inv_mass) {
    const long long total = K * rows;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i % rows;
        double v = comps[i];
        if (unscale) {
            v /= psf;
            v += mean[r];
        }
        if (inv_mass) v *= inv_mass[r / 3];
        comps[i] = v;
    }
}

extern "C" int asb_components_post(asb_ctx* ctx, int unscale, double pre_scale_factor, const double* invMassL_loc,
                                   double* comps_out) {
    if (!ctx || !ctx->comps) return ASB_ERR_ARG;
    if (unscale && !ctx->have_mean) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_components_post: no mean on the device");
    const double* im = nullptr;
    if (invMassL_loc) {
        ASB_HIP(ctx, hipMemcpyAsync(ctx->s_dev, invMassL_loc, (size_t)ctx->n_loc * sizeof(double),
                                    hipMemcpyHostToDevice, ctx->stream));
        im = ctx->s_dev;
    }
    const long long rows = ctx->n_loc * 3, total = ctx->K * rows;
    long long want = (total + 255) / 256;
    const int grid = (int)(want < ctx->nblk_cap ? want : ctx->nblk_cap);
    hipLaunchKernelGGL(k_comps_post, dim3(grid), dim3(256), 0, ctx->stream, ctx->comps, (long long)ctx->K, rows,
                       unscale, pre_scale_factor, ctx->mean, im);
    ASB_CHECK_LAUNCH(ctx);
    if (comps_out) {
        ASB_HIP(ctx, hipMemcpyAsync(comps_out, ctx->comps, (size_t)total * sizeof(double), hipMemcpyDeviceToHost,
                                    ctx->stream));
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return ASB_OK;
}

extern "C" int asb_deflate_stats(asb_ctx* ctx, int64_t* n_panels, int64_t* n_refresh) {
    if (!ctx) return ASB_ERR_ARG;
    if (n_panels) *n_panels = ctx->n_panels;
    if (n_refresh) *n_refresh = ctx->n_refresh;
    return ASB_OK;
}

extern "C" int asb_deflate_energy_passes(asb_ctx* ctx, int64_t* n_passes) {
    if (!ctx || !n_passes) return ASB_ERR_ARG;
    *n_passes = (ctx->mode == ASB_DEFLATE_PROJECT || ctx->k_switch >= 0) ? ctx->n_energy_pass : 1;
    return ASB_OK;
}

extern "C" int asb_deflate_coop_fallbacks(asb_ctx* ctx, int64_t* n) {
    if (!ctx || !n) return ASB_ERR_ARG;
    *n = ctx->n_coop_fallbacks;
    return ASB_OK;
}

extern "C" int asb_deflate_guessed_panels(asb_ctx* ctx, int64_t* n) {
    if (!ctx || !n) return ASB_ERR_ARG;
    *n = ctx->n_guess_panels;
    return ASB_OK;
}

extern "C" int asb_deflate_sketch_stats(asb_ctx* ctx, int64_t* runs, int64_t* reads) {
    if (!ctx) return ASB_ERR_ARG;
    // (the kernel itself decides whether the sketch holds enough of the residual to replay: device counters)
    unsigned c[4] = {0, 0, 0, 0};
    if (ctx->sk_counts) {
        ASB_HIP(ctx, hipMemcpyAsync(c, ctx->sk_counts, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
        ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (runs) *runs = c[0];
    if (reads) *reads = ctx->n_sketch_reads > (int64_t)c[1] ? ctx->n_sketch_reads - (int64_t)c[1] : 0;
    return ASB_OK;
}

extern "C" int asb_deflate_spec_stats(asb_ctx* ctx, int64_t* tried, int64_t* kept) {
    if (!ctx) return ASB_ERR_ARG;
    if (tried) *tried = ctx->n_spec_steps;
    if (kept) *kept = ctx->n_spec_kept;
    return ASB_OK;
}
