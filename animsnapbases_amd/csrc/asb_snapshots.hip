// Context management + snapshot preparation kernels -- posSnapshots, the reference's
// snapbases/posSnapshots.py:64-105 (do_snapshots_precomputations) and :163-172 (standarize).
// gfx950 (MI355X) only.
#include "asb_common.h"

#include <cstdlib>
#include <cstring>

extern "C" int asb_snapshots_center(asb_ctx* ctx, int rest_shape, int subtract, double* local_sum);

// --------------------------------------------------------------------------------------
// Tiled transpose (F, C) <-> (C, Fp) through LDS; optional scaling of column c by
// colscale[c/3] (mass weighting, posSnapshots.py:82).  32x32 tiles, 256 threads.
//   out[c * ld_out + r] = in[r * ld_in + c] * scale(c)      r < rows_in, c < cols_in
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_transpose(const double* __restrict__ in, long long rows_in,
                                                   long long cols_in, long long ld_in,
                                                   double* __restrict__ out, long long ld_out,
                                                   const double* __restrict__ colscale, int scale_on_in_col) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const long long c0 = (long long)blockIdx.x * 32, r0 = (long long)blockIdx.y * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long r = r0 + ty + i * 8, c = c0 + tx;
        double v = 0.0;
        if (r < rows_in && c < cols_in) {
            v = in[r * ld_in + c];
            if (colscale && scale_on_in_col) v *= colscale[c / 3];
        }
        tile[ty + i * 8][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long c = c0 + ty + i * 8, r = r0 + tx;
        if (r < rows_in && c < cols_in) out[c * ld_out + r] = tile[tx][ty + i * 8];
    }
}

// --------------------------------------------------------------------------------------
// rest_shape "first" fused into the layout change: the rest row of column c is frame 0 of that column, known before
// anything else is read, so one sweep transposes, mass-weights, subtracts the rest shape (when standardising), writes the
// zero padding and leaves sum(x) and sum(x^2) of what it wrote -- what k_transpose + k_center + k_sqdev did in three
// sweeps.  One block per strip of 32 columns over all frames; per-block partials [sum, sum of squares], reduced in order.
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_transpose_first(const double* __restrict__ in, long long F, long long C, long long ld_in,
                                                         double* __restrict__ out, long long Fp, const double* __restrict__ colscale,
                                                         int subtract, double* __restrict__ mean, double* __restrict__ part) {
    __shared__ double tile[32][33];
    __shared__ double m_sh[32];
    __shared__ double red[8];
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    const long long c0 = (long long)blockIdx.x * 32;
    if (tid < 32) {
        const long long c = c0 + tid;
        double m = 0.0;
        if (c < C) {
            m = in[c];
            if (colscale) m *= colscale[c / 3];
            mean[c] = m;
        }
        m_sh[tid] = m;
    }
    __syncthreads();
    double v2[2] = {0.0, 0.0};
    for (long long r0 = 0; r0 < F; r0 += 32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long r = r0 + ty + i * 8, c = c0 + tx;
            double v = 0.0;
            if (r < F && c < C) {
                v = in[r * ld_in + c];
                if (colscale) v *= colscale[c / 3];
            }
            tile[ty + i * 8][tx] = v;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long c = c0 + ty + i * 8, r = r0 + tx;
            if (r < F && c < C) {
                double v = tile[tx][ty + i * 8];
                if (subtract) v -= m_sh[ty + i * 8];
                out[c * Fp + r] = v;
                v2[0] += v;
                v2[1] += v * v;
            }
        }
        __syncthreads();
    }
    const int pad = (int)(Fp - F);
    for (int q = tid; q < 32 * pad; q += 256) {
        const long long c = c0 + q / pad;
        if (c < C) out[c * Fp + F + q % pad] = 0.0;
    }
    block_sum<2>(v2, red);
    if (tid == 0) { part[2 * (long long)blockIdx.x] = v2[0]; part[2 * (long long)blockIdx.x + 1] = v2[1]; }
}

__global__ __launch_bounds__(1024) void k_sum_pairs(const double* __restrict__ part, long long n, double* __restrict__ out) {
    __shared__ double red[32];
    double v[2] = {0.0, 0.0};
    for (long long i = threadIdx.x; i < n; i += 1024) { v[0] += part[2 * i]; v[1] += part[2 * i + 1]; }
    block_sum<2>(v, red);
    if (threadIdx.x == 0) { out[0] = v[0]; out[1] = v[1]; }
}

// --------------------------------------------------------------------------------------
// One wave per row of the vertex-major tensor.
// k_center: mean[r] = row[0] (rest_shape 0) or mean_f row (1); optionally row -= mean;
//           block partial of sum(row) afterwards.
// k_sqdev : block partial of sum (row - mu)^2.
// k_scale : row *= a.
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_center(double* __restrict__ X, long long nrows, int F, int Fp,
                                                int rest_shape, int subtract, double* __restrict__ mean,
                                                double* __restrict__ psum) {
    __shared__ double sh[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double acc = 0.0;
    for (long long r = (long long)blockIdx.x * 4 + wid; r < nrows; r += (long long)gridDim.x * 4) {
        double* row = X + r * Fp;
        double m;
        if (rest_shape == 0) {
            m = row[0];
        } else {
            double s = 0.0;
            for (int f = lane; f < F; f += 64) s += row[f];
            m = wave_sum(s) / (double)F;
        }
        double s2 = 0.0;
        for (int f = lane; f < F; f += 64) {
            double v = row[f];
            if (subtract) {
                v -= m;
                row[f] = v;
            }
            s2 += v;
        }
        acc += wave_sum(s2);
        if (lane == 0) mean[r] = m;
    }
    if (lane == 0) sh[wid] = acc;
    __syncthreads();
    if (threadIdx.x == 0) psum[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void k_sqdev(const double* __restrict__ X, long long nrows, int F, int Fp,
                                               double mu, double* __restrict__ psum) {
    __shared__ double sh[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double acc = 0.0;
    for (long long r = (long long)blockIdx.x * 4 + wid; r < nrows; r += (long long)gridDim.x * 4) {
        const double* row = X + r * Fp;
        double s = 0.0;
        for (int f = lane; f < F; f += 64) {
            const double d = row[f] - mu;
            s += d * d;
        }
        acc += wave_sum(s);
    }
    if (lane == 0) sh[wid] = acc;
    __syncthreads();
    if (threadIdx.x == 0) psum[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void k_scale(double* __restrict__ X, long long n2, double a) {
    double2* p = reinterpret_cast<double2*>(X);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2;
         i += (long long)gridDim.x * blockDim.x) {
        double2 v = p[i];
        v.x *= a;
        v.y *= a;
        p[i] = v;
    }
}

// k_scale_energy: row *= a AND the per-vertex energies of the scaled tensor in the same sweep (one wave per vertex: its
// three rows are 3 Fp contiguous doubles; the zero padding stays zero).  E0[v] = sum of the scaled squares; EV[v] = E0[v]
// minus the energy along the constant-in-time direction (row sums squared / F); per-block (sum, max, sum of that energy)
// partials for |X|^2, the largest energy and the share of the constant direction -- what the projection path would
// otherwise re-read all of X for.
__global__ __launch_bounds__(256) void k_scale_energy(double* __restrict__ X, long long n_vert, int Fp, int F, double a,
                                                      double* __restrict__ E0, double* __restrict__ EV,
                                                      double* __restrict__ psum, double* __restrict__ pmax,
                                                      double* __restrict__ pmean) {
    __shared__ double sh[12];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int h2 = Fp / 2;
    const double inv_f = 1.0 / (double)F;
    double bsum = 0.0, bmax = 0.0, bmean = 0.0;
    for (long long v = (long long)blockIdx.x * 4 + wid; v < n_vert; v += (long long)gridDim.x * 4) {
        double e0 = 0.0, e1 = 0.0, m = 0.0;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            double2* p = reinterpret_cast<double2*>(X + (v * 3 + d) * Fp);
            double s = 0.0;
            for (int i = lane; i < h2; i += 64) {
                double2 q = p[i];
                q.x *= a;
                q.y *= a;
                p[i] = q;
                e0 += q.x * q.x;
                e1 += q.y * q.y;
                s += q.x + q.y;
            }
            s = wave_sum(s);
            m += s * s;
        }
        const double e = wave_sum(e0 + e1);
        m *= inv_f;
        if (lane == 0) {
            E0[v] = e;
            EV[v] = e > m ? e - m : 0.0;
        }
        bsum += e;
        bmean += m;
        bmax = fmax(bmax, e);
    }
    if (lane == 0) { sh[wid] = bsum; sh[4 + wid] = bmax; sh[8 + wid] = bmean; }
    __syncthreads();
    if (threadIdx.x == 0) {
        psum[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        pmax[blockIdx.x] = fmax(fmax(sh[4], sh[5]), fmax(sh[6], sh[7]));
        pmean[blockIdx.x] = (sh[8] + sh[9]) + (sh[10] + sh[11]);
    }
}

__global__ __launch_bounds__(256) void k_e0_finish(const double* __restrict__ psum, const double* __restrict__ pmax,
                                                   const double* __restrict__ pmean, int n, double* __restrict__ out) {
    __shared__ double sh[12];
    double s = 0.0, m = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { s += psum[i]; m = fmax(m, pmax[i]); c += pmean[i]; }
    s = wave_sum(s);
    m = wave_max(m);
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = s; sh[4 + (threadIdx.x >> 6)] = m; sh[8 + (threadIdx.x >> 6)] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        out[1] = fmax(fmax(sh[4], sh[5]), fmax(sh[6], sh[7]));
        out[2] = (sh[8] + sh[9]) + (sh[10] + sh[11]);
    }
}

// out[0] = sum EV, out[1] = sum EV^2 (one block: n values from L2): how unevenly the energy outside the constant direction is
// spread over the vertices -- localised data (a few regions carry it all) against global modes
__global__ __launch_bounds__(1024) void k_ev_moments(const double* __restrict__ EV, long long n, double* __restrict__ out) {
    __shared__ double sh[2 * 16];
    double v[2] = {0.0, 0.0};
    for (long long i = threadIdx.x; i < n; i += 1024) { const double e = EV[i]; v[0] += e; v[1] += e * e; }
    v[0] = wave_sum(v[0]);
    v[1] = wave_sum(v[1]);
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = v[0]; sh[16 + (threadIdx.x >> 6)] = v[1]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int q = 0; q < 16; ++q) { a += sh[q]; b += sh[16 + q]; }
        out[0] = a;
        out[1] = b;
    }
}

__global__ __launch_bounds__(256) void k_sum(const double* __restrict__ in, int n, double* __restrict__ out) {
    __shared__ double sh[4];
    double v[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += blockDim.x) v[0] += in[i];
    block_sum<1>(v, sh);
    if (threadIdx.x == 0) out[0] = v[0];
}

// --------------------------------------------------------------------------------------
// C ABI
// --------------------------------------------------------------------------------------
extern "C" int asb_abi_version(void) { return ASB_ABI_VERSION; }

extern "C" int asb_create(int device_id, void* hip_stream, asb_ctx** out) {
    if (!out) return ASB_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device_id || device_id < 0) return ASB_ERR_NODEV;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return ASB_ERR_NODEV;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return ASB_ERR_NODEV;   // kernels are built for gfx950 only
    asb_ctx* ctx = new asb_ctx();
    ctx->dev = device_id;
    *out = ctx;
    ASB_HIP(ctx, hipSetDevice(device_id));
    if (hip_stream == ASB_STREAM_DEFAULT) {
        ctx->stream = nullptr;                      // the null stream
    } else if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
    } else {
        ASB_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    ctx->n_cu = prop.multiProcessorCount;
    if (const char* pk = getenv("ASB_PROJECT_KERNEL")) ctx->project_kernel = atoi(pk);
    if (const char* pv = getenv("ASB_L2_VARIANT")) ctx->l2_variant = atoi(pv);
    if (const char* pc = getenv("ASB_PANEL_COOP")) ctx->panel_coop = atoi(pc);
    if (const char* ts = getenv("ASB_COOP_TEST_STALL")) ctx->coop_test_stall = atoi(ts);
    if (const char* er = getenv("ASB_E0_REUSE")) ctx->e0_reuse = atoi(er);
    if (const char* er = getenv("ASB_FIRST_PANEL_MEAN")) ctx->first_panel_mean = atoi(er);
    if (const char* hp = getenv("ASB_HOST_POLL")) ctx->host_poll = atoi(hp);
    if (const char* cr = getenv("ASB_CORRECT_ROWS")) ctx->correct_rows = atoi(cr);
    if (const char* sp = getenv("ASB_SUPER_PANELS")) ctx->super_panels = atoi(sp);
    if (const char* sp = getenv("ASB_SPEC_PANELS")) ctx->spec_panels = atoi(sp);
    if (const char* gc = getenv("ASB_GATHER_CPT")) ctx->gather_cpt = atoi(gc);
    if (const char* dp = getenv("ASB_DOUBLE_PANELS")) ctx->double_panels = atoi(dp);
    if (const char* dp = getenv("ASB_SUB_PANELS")) ctx->sub_panels = atoi(dp);
    if (const char* dp = getenv("ASB_PRE_ORTH")) ctx->pre_orth = atoi(dp);
    if (const char* dp = getenv("ASB_TILE_CHAIN")) ctx->tile_chain = atoi(dp);
    if (const char* dp = getenv("ASB_SUB_CHAIN")) ctx->sub_chain = atoi(dp);
    if (const char* dp = getenv("ASB_COOP_LAUNCH")) ctx->coop_launch = atoi(dp) ? 1 : 0;
    if (const char* dp = getenv("ASB_SPEC_PASS")) ctx->spec_pass = atoi(dp);
    if (const char* dp = getenv("ASB_SPEC_W_RANK")) ctx->spec_w_rank = atoi(dp);
    if (const char* sk = getenv("ASB_SKETCH")) ctx->sketch = atoi(sk);
    if (const char* sf = getenv("ASB_STALL_FALLBACK")) ctx->stall_fallback = atoi(sf);
    if (const char* dv = getenv("ASB_DIVERSE")) ctx->diverse = atoi(dv);
    if (const char* sk = getenv("ASB_SKETCH_TEST_STALL")) ctx->sk_test_stall = atoi(sk);
    if (const char* dp = getenv("ASB_SUB_FIRST")) ctx->sub_first = atoi(dp) < 1 ? 1 : atoi(dp);
    if (const char* bt = getenv("ASB_BAND_TARGET")) { ctx->band_target = atoll(bt); ctx->band_cap = ctx->band_target * 4 / 3; }
    ctx->nblk_cap = prop.multiProcessorCount * 8;   // grid cap for streaming passes (guide: G11)
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->pmax, (size_t)ctx->nblk_cap))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->pidx, (size_t)ctx->nblk_cap))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->psum, (size_t)ctx->nblk_cap))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->scalar_dev, (size_t)48))) return rc;
    return ASB_OK;
}

extern "C" void asb_destroy(asb_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->dev);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->X_original) {                    // a POD in levels that did not reach its end: the original snapshots own the slot
        ctx->X = ctx->X_original;
        ctx->X_original = nullptr;
    }
    if (ctx->X_deflated) (void)hipFree(ctx->X_deflated);
    if (ctx->pod_u1) (void)hipFree(ctx->pod_u1);
    for (auto& kv : ctx->alloc_bytes) {       // every context-owned buffer was registered by asb_alloc
        void** slot = (void**)kv.first;
        if (*slot) (void)hipFree(*slot);
        *slot = nullptr;
    }
    asb_splocs_free(ctx);
    asb_geo_free(ctx);
    if (ctx->host_pin) (void)hipHostFree(ctx->host_pin);
    if (ctx->res_pin) (void)hipHostFree(ctx->res_pin);
    if (ctx->dl_stream) { (void)hipStreamSynchronize(ctx->dl_stream); (void)hipStreamDestroy(ctx->dl_stream); }
    if (ctx->dl_event) (void)hipEventDestroy(ctx->dl_event);
    if (ctx->dl_host && ctx->dl_host_owned) (void)hipHostFree(ctx->dl_host);
    for (auto& e : ctx->ev_pool) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" const char* asb_last_error(const asb_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int asb_sync(asb_ctx* ctx) {
    if (!ctx) return ASB_ERR_ARG;
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ASB_OK;
}

extern "C" int asb_prof_reset(asb_ctx* ctx, int enable) {
    if (!ctx) return ASB_ERR_ARG;
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->ev_used = 0;
    ctx->prof = enable != 0;
    return ASB_OK;
}

extern "C" int asb_prof_get(asb_ctx* ctx, int64_t* launches, double* total_ms) {
    if (!ctx) return ASB_ERR_ARG;
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double tot = 0.0;
    for (size_t i = 0; i < ctx->ev_used; ++i) {
        float ms = 0.f;
        ASB_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_pool[i].first, ctx->ev_pool[i].second));
        tot += ms;
    }
    if (launches) *launches = (int64_t)ctx->ev_used;
    if (total_ms) *total_ms = tot;
    return ASB_OK;
}

static int set_shape(asb_ctx* ctx, int64_t F, int64_t N_glob, int64_t v0, int64_t n_loc, bool clear = true) {
    if (F < 1 || n_loc < 1 || v0 < 0 || v0 + n_loc > N_glob)
        ASB_FAIL(ctx, ASB_ERR_ARG, "bad snapshot shape F=%lld N=%lld v0=%lld n_loc=%lld", (long long)F,
                 (long long)N_glob, (long long)v0, (long long)n_loc);
    if (F > 32768) ASB_FAIL(ctx, ASB_ERR_LIMIT, "F = %lld exceeds the 32768-frame limit of the streaming kernels", (long long)F);
    ctx->F = F;
    ctx->Fp = (F + 15) / 16 * 16;   // rows are whole 128-byte lines; 16-frame chunks for the MFMA pass
    ctx->N_glob = N_glob;
    ctx->v0 = v0;
    ctx->n_loc = n_loc;
    ctx->have_mean = false;
    { ctx->e0_valid = false; ctx->ev_valid = false; }
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->X, (size_t)n_loc * 3 * ctx->Fp))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->mean, (size_t)n_loc * 3))) return rc;
    if (clear) ASB_HIP(ctx, hipMemsetAsync(ctx->X, 0, (size_t)n_loc * 3 * ctx->Fp * sizeof(double), ctx->stream));
    ASB_HIP(ctx, hipMemsetAsync(ctx->mean, 0, (size_t)n_loc * 3 * sizeof(double), ctx->stream));
    return ASB_OK;
}

static int transpose_in(asb_ctx* ctx, const double* stage_dev, const double* massL_dev) {
    const long long C = ctx->n_loc * 3;
    dim3 grid((unsigned)((C + 31) / 32), (unsigned)((ctx->F + 31) / 32));
    hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, ctx->stream, stage_dev, (long long)ctx->F, C, C, ctx->X,
                       (long long)ctx->Fp, massL_dev, 1);
    ASB_CHECK_LAUNCH(ctx);
    return ASB_OK;
}

extern "C" int asb_snapshots_upload(asb_ctx* ctx, const double* X, int64_t F, int64_t N_glob, int64_t v0,
                                    int64_t n_loc, const double* massL) {
    if (!ctx || !X) return ASB_ERR_ARG;
    ASB_HIP(ctx, hipSetDevice(ctx->dev));
    int rc = set_shape(ctx, F, N_glob, v0, n_loc);
    if (rc) return rc;
    double* stage = nullptr;
    double* mdev = nullptr;
    const size_t C = (size_t)n_loc * 3;
    ASB_HIP(ctx, hipMalloc((void**)&stage, (size_t)F * C * sizeof(double)));
    hipError_t e = hipMemcpy2DAsync(stage, C * sizeof(double), X + v0 * 3, (size_t)N_glob * 3 * sizeof(double),
                                    C * sizeof(double), (size_t)F, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && massL) {
        e = hipMalloc((void**)&mdev, (size_t)n_loc * sizeof(double));
        if (e == hipSuccess)
            e = hipMemcpyAsync(mdev, massL + v0, (size_t)n_loc * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess) {
        rc = transpose_in(ctx, stage, mdev);
        e = hipStreamSynchronize(ctx->stream);
    }
    (void)hipFree(stage);
    if (mdev) (void)hipFree(mdev);
    if (e != hipSuccess) ASB_FAIL(ctx, ASB_ERR_HIP, "asb_snapshots_upload: %s", hipGetErrorString(e));
    return rc;
}

extern "C" int asb_snapshots_adopt_dev(asb_ctx* ctx, const double* X_dev, int64_t F, int64_t n_loc,
                                       const double* massL_loc, int64_t v0, int64_t N_glob) {
    if (!ctx || !X_dev) return ASB_ERR_ARG;
    ASB_HIP(ctx, hipSetDevice(ctx->dev));
    int rc = set_shape(ctx, F, N_glob, v0, n_loc);
    if (rc) return rc;
    double* mdev = nullptr;
    if (massL_loc) {
        ASB_HIP(ctx, hipMalloc((void**)&mdev, (size_t)n_loc * sizeof(double)));
        ASB_HIP(ctx, hipMemcpyAsync(mdev, massL_loc, (size_t)n_loc * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    rc = transpose_in(ctx, X_dev, mdev);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (mdev) (void)hipFree(mdev);
    if (e != hipSuccess) ASB_FAIL(ctx, ASB_ERR_HIP, "asb_snapshots_adopt_dev: %s", hipGetErrorString(e));
    return rc;
}

// layout change + rest shape (+ its subtraction) in one go; sums_out[0] = sum(x), sums_out[1] = sum(x^2) of the shard as it
// stands afterwards (rest_shape 1, "average", needs the column means first: the separate sweeps, sums_out[1] = -1)
static int transpose_rest(asb_ctx* ctx, const double* stage_dev, const double* massL_dev, int rest_shape, int subtract, double* sums_out) {
    const long long C = ctx->n_loc * 3;
    if (rest_shape != 0) {
        int rc = transpose_in(ctx, stage_dev, massL_dev);
        if (rc) return rc;
        sums_out[1] = -1.0;
        return asb_snapshots_center(ctx, rest_shape, subtract, &sums_out[0]);
    }
    const long long nb = (C + 31) / 32;
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->tr_part, (size_t)2 * nb))) return rc;
    hipLaunchKernelGGL(k_transpose_first, dim3((unsigned)nb), dim3(256), 0, ctx->stream, stage_dev, (long long)ctx->F, C, C, ctx->X,
                       (long long)ctx->Fp, massL_dev, subtract, ctx->mean, ctx->tr_part);
    ASB_CHECK_LAUNCH(ctx);
    hipLaunchKernelGGL(k_sum_pairs, dim3(1), dim3(1024), 0, ctx->stream, ctx->tr_part, nb, ctx->scalar_dev);
    ASB_CHECK_LAUNCH(ctx);
    ctx->have_mean = true;
    { ctx->e0_valid = false; ctx->ev_valid = false; }
    ASB_HIP(ctx, hipMemcpyAsync(sums_out, ctx->scalar_dev, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ASB_OK;
}

extern "C" int asb_snapshots_upload_rest(asb_ctx* ctx, const double* X, int64_t F, int64_t N_glob, int64_t v0, int64_t n_loc,
                                         const double* massL, int rest_shape, int subtract, double* sums_out) {
    if (!ctx || !X || !sums_out) return ASB_ERR_ARG;
    if (rest_shape != 0 && rest_shape != 1) ASB_FAIL(ctx, ASB_ERR_ARG, "unknown rest shape code %d", rest_shape);
    ASB_HIP(ctx, hipSetDevice(ctx->dev));
    int rc = set_shape(ctx, F, N_glob, v0, n_loc, rest_shape != 0);
    if (rc) return rc;
    double* stage = nullptr;
    double* mdev = nullptr;
    const size_t C = (size_t)n_loc * 3;
    ASB_HIP(ctx, hipMalloc((void**)&stage, (size_t)F * C * sizeof(double)));
    hipError_t e = hipMemcpy2DAsync(stage, C * sizeof(double), X + v0 * 3, (size_t)N_glob * 3 * sizeof(double),
                                    C * sizeof(double), (size_t)F, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && massL) {
        e = hipMalloc((void**)&mdev, (size_t)n_loc * sizeof(double));
        if (e == hipSuccess)
            e = hipMemcpyAsync(mdev, massL + v0, (size_t)n_loc * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess) {
        rc = transpose_rest(ctx, stage, mdev, rest_shape, subtract, sums_out);
        e = hipStreamSynchronize(ctx->stream);
    }
    (void)hipFree(stage);
    if (mdev) (void)hipFree(mdev);
    if (e != hipSuccess) ASB_FAIL(ctx, ASB_ERR_HIP, "asb_snapshots_upload_rest: %s", hipGetErrorString(e));
    return rc;
}

extern "C" int asb_snapshots_adopt_dev_rest(asb_ctx* ctx, const double* X_dev, int64_t F, int64_t n_loc, const double* massL_loc,
                                            int64_t v0, int64_t N_glob, int rest_shape, int subtract, double* sums_out) {
    if (!ctx || !X_dev || !sums_out) return ASB_ERR_ARG;
    if (rest_shape != 0 && rest_shape != 1) ASB_FAIL(ctx, ASB_ERR_ARG, "unknown rest shape code %d", rest_shape);
    ASB_HIP(ctx, hipSetDevice(ctx->dev));
    int rc = set_shape(ctx, F, N_glob, v0, n_loc, rest_shape != 0);
    if (rc) return rc;
    double* mdev = nullptr;
    if (massL_loc) {
        ASB_HIP(ctx, hipMalloc((void**)&mdev, (size_t)n_loc * sizeof(double)));
        ASB_HIP(ctx, hipMemcpyAsync(mdev, massL_loc, (size_t)n_loc * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    rc = transpose_rest(ctx, X_dev, mdev, rest_shape, subtract, sums_out);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (mdev) (void)hipFree(mdev);
    if (e != hipSuccess) ASB_FAIL(ctx, ASB_ERR_HIP, "asb_snapshots_adopt_dev_rest: %s", hipGetErrorString(e));
    return rc;
}

static int finish_sum(asb_ctx* ctx, int nblk, double* out_host) {
    hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, ctx->stream, ctx->psum, nblk, ctx->scalar_dev);
    ASB_CHECK_LAUNCH(ctx);
    ASB_HIP(ctx, hipMemcpyAsync(out_host, ctx->scalar_dev, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ASB_OK;
}

static int row_grid(const asb_ctx* ctx) {
    long long want = (ctx->n_loc * 3 + 3) / 4;
    return (int)(want < ctx->nblk_cap ? want : ctx->nblk_cap);
}

extern "C" int asb_snapshots_center(asb_ctx* ctx, int rest_shape, int subtract, double* local_sum) {
    if (!ctx || !ctx->X) return ASB_ERR_ARG;
    if (rest_shape != 0 && rest_shape != 1) ASB_FAIL(ctx, ASB_ERR_ARG, "unknown rest shape code %d", rest_shape);
    const int grid = row_grid(ctx);
    hipLaunchKernelGGL(k_center, dim3(grid), dim3(256), 0, ctx->stream, ctx->X, (long long)ctx->n_loc * 3,
                       (int)ctx->F, (int)ctx->Fp, rest_shape, subtract, ctx->mean, ctx->psum);
    ASB_CHECK_LAUNCH(ctx);
    ctx->have_mean = true;
    if (subtract) { ctx->e0_valid = false; ctx->ev_valid = false; }
    double tmp;
    return finish_sum(ctx, grid, local_sum ? local_sum : &tmp);
}

extern "C" int asb_snapshots_sqdev(asb_ctx* ctx, double mu, double* local_sqdev) {
    if (!ctx || !ctx->X || !local_sqdev) return ASB_ERR_ARG;
    const int grid = row_grid(ctx);
    hipLaunchKernelGGL(k_sqdev, dim3(grid), dim3(256), 0, ctx->stream, ctx->X, (long long)ctx->n_loc * 3,
                       (int)ctx->F, (int)ctx->Fp, mu, ctx->psum);
    ASB_CHECK_LAUNCH(ctx);
    return finish_sum(ctx, grid, local_sqdev);
}

extern "C" int asb_snapshots_scale(asb_ctx* ctx, double a) {
    if (!ctx || !ctx->X) return ASB_ERR_ARG;
    // the scaling sweep also leaves the per-vertex energies of the scaled tensor (E0), |X|^2 and the largest energy:
    // asb_project_begin then needs no pass over X of its own
    int rc;
    if ((rc = asb_alloc(ctx, &ctx->E0, (size_t)ctx->n_loc))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->EV, (size_t)ctx->n_loc))) return rc;
    if ((rc = asb_alloc(ctx, &ctx->e0_sc, (size_t)8))) return rc;
    long long want = (ctx->n_loc + 3) / 4;
    const int grid = (int)(want < ctx->nblk_cap ? want : ctx->nblk_cap);
    if ((rc = asb_alloc(ctx, &ctx->mean_part, (size_t)ctx->nblk_cap))) return rc;
    hipLaunchKernelGGL(k_scale_energy, dim3(grid), dim3(256), 0, ctx->stream, ctx->X, (long long)ctx->n_loc, (int)ctx->Fp, (int)ctx->F,
                       a, ctx->E0, ctx->EV, ctx->psum, ctx->pmax, ctx->mean_part);
    ASB_CHECK_LAUNCH(ctx);
    hipLaunchKernelGGL(k_e0_finish, dim3(1), dim3(256), 0, ctx->stream, ctx->psum, ctx->pmax, ctx->mean_part, grid, ctx->e0_sc);
    ASB_CHECK_LAUNCH(ctx);
    ctx->e0_valid = true;
    ctx->ev_valid = true;
    // share of |X|^2 along the constant-in-time direction (this sweep is part of the preparation, not of a deflation step)
    hipLaunchKernelGGL(k_ev_moments, dim3(1), dim3(1024), 0, ctx->stream, ctx->EV, (long long)ctx->n_loc, ctx->e0_sc + 4);
    ASB_CHECK_LAUNCH(ctx);
    double h[6];
    ASB_HIP(ctx, hipMemcpyAsync(h, ctx->e0_sc, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // squared coefficient of variation of the per-vertex energies outside the constant direction
    ctx->ev_cv2 = (h[4] > 0.0 && ctx->n_loc > 1) ? (double)ctx->n_loc * h[5] / (h[4] * h[4]) - 1.0 : 0.0;
    if (getenv("ASB_DEBUG_PANELS")) fprintf(stderr, "[asb] energies outside the constant direction: squared coefficient of variation %.3f over %lld vertices\n", ctx->ev_cv2, (long long)ctx->n_loc);
    ctx->mean_frac = h[0] > 0.0 ? h[2] / h[0] : 0.0;
    ctx->mean_energy = h[2];
    ctx->prep_normx2 = h[0];
    return ASB_OK;
}

extern "C" int asb_snapshots_get_mean(asb_ctx* ctx, double* mean_out) {
    if (!ctx || !ctx->X || !mean_out) return ASB_ERR_ARG;
    if (!ctx->have_mean) ASB_FAIL(ctx, ASB_ERR_ARG, "asb_snapshots_get_mean before asb_snapshots_center");
    ASB_HIP(ctx, hipMemcpyAsync(mean_out, ctx->mean, (size_t)ctx->n_loc * 3 * sizeof(double), hipMemcpyDeviceToHost,
                                ctx->stream));
    ASB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ASB_OK;
}

// (C, Fp) vertex-major device tensor -> host (F, C)
int asb_download_vertex_major(asb_ctx* ctx, const double* src, double* out) {
    const long long C = ctx->n_loc * 3;
    double* stage = nullptr;
    ASB_HIP(ctx, hipMalloc((void**)&stage, (size_t)ctx->F * C * sizeof(double)));
    dim3 grid((unsigned)((ctx->F + 31) / 32), (unsigned)((C + 31) / 32));
    hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, ctx->stream, src, C, (long long)ctx->F, (long long)ctx->Fp,
                       stage, C, (const double*)nullptr, 0);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipMemcpyAsync(out, stage, (size_t)ctx->F * C * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(stage);
    if (e != hipSuccess) ASB_FAIL(ctx, ASB_ERR_HIP, "download: %s", hipGetErrorString(e));
    return ASB_OK;
}

extern "C" int asb_snapshots_download(asb_ctx* ctx, double* out) {
    if (!ctx || !ctx->X || !out) return ASB_ERR_ARG;
    return asb_download_vertex_major(ctx, ctx->X, out);
}

extern "C" int asb_deflate_download_residual(asb_ctx* ctx, double* out) {
    if (!ctx || !ctx->R || !out) return ASB_ERR_ARG;
    return asb_download_vertex_major(ctx, ctx->R, out);
}
