// Snapshot ingest: rigid Procrustes alignment of every frame to frame 0 -- utils/process.py:210-250
// (find_rbm_procrustes + transform inside align) of the reference.  gfx950 only.
//
// Frames arrive in the reference layout (F, N, 3).  One block per frame: centroid of the frame and of
// frame 0, the 3x3 cross-covariance M = (to - t1)^T (from - t0), its rotation R = U V^T (the orthogonal polar
// factor, from the Jacobi eigen-decomposition of M^T M; R *= -1 when det R < 0, as the reference does), then
// v' = R v + (t1 - R t0).  Arithmetic in f64 (the reference works in f32 on the h5 data and stores f32).
#include "asb_kernels.h"

__device__ inline void eig3_full(double a00, double a01, double a02, double a11, double a12, double a22, double lam[3],
                                 double V[3][3]) {
    double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = a01 * a01 + a02 * a02 + a12 * a12, dia = a00 * a00 + a11 * a11 + a22 * a22;
        if (off == 0.0 || off <= 1e-36 * dia) break;
        ASB_JROT(a00, a11, a01, a02, a12, v00, v01, v10, v11, v20, v21)
        ASB_JROT(a00, a22, a02, a01, a12, v00, v02, v10, v12, v20, v22)
        ASB_JROT(a11, a22, a12, a01, a02, v01, v02, v11, v12, v21, v22)
    }
    lam[0] = a00; lam[1] = a11; lam[2] = a22;
    V[0][0] = v00; V[0][1] = v01; V[0][2] = v02;
    V[1][0] = v10; V[1][1] = v11; V[1][2] = v12;
    V[2][0] = v20; V[2][1] = v21; V[2][2] = v22;
}

// T (F, 4, 4) row-major homogeneous matrices; frames (F, N, 3)
__global__ __launch_bounds__(256) void k_procrustes(const double* __restrict__ frames, long long N, int rigid,
                                                    double* __restrict__ T) {
    __shared__ double sh[15 * 4];
    const double* fr = frames + (long long)blockIdx.x * N * 3;
    const double* f0 = frames;
    // pass 1: centroids
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (long long v = threadIdx.x; v < N; v += blockDim.x) {
        s[0] += fr[3 * v]; s[1] += fr[3 * v + 1]; s[2] += fr[3 * v + 2];
        s[3] += f0[3 * v]; s[4] += f0[3 * v + 1]; s[5] += f0[3 * v + 2];
    }
    block_sum<6>(s, sh);
    double t0[3] = {s[0] / N, s[1] / N, s[2] / N}, t1[3] = {s[3] / N, s[4] / N, s[5] / N};
    // pass 2: M[a][b] = sum (to_a - t1_a)(from_b - t0_b)
    double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (long long v = threadIdx.x; v < N; v += blockDim.x) {
        const double p0 = fr[3 * v] - t0[0], p1 = fr[3 * v + 1] - t0[1], p2 = fr[3 * v + 2] - t0[2];
        const double q0 = f0[3 * v] - t1[0], q1 = f0[3 * v + 1] - t1[1], q2 = f0[3 * v + 2] - t1[2];
        m[0] += q0 * p0; m[1] += q0 * p1; m[2] += q0 * p2;
        m[3] += q1 * p0; m[4] += q1 * p1; m[5] += q1 * p2;
        m[6] += q2 * p0; m[7] += q2 * p1; m[8] += q2 * p2;
    }
    __syncthreads();
    block_sum<9>(m, sh);
    if (threadIdx.x != 0) return;
    // R = M (M^T M)^(-1/2)
    double B[6] = {0, 0, 0, 0, 0, 0};   // M^T M (sym): 00 01 02 11 12 22
    for (int k = 0; k < 3; ++k) {
        B[0] += m[3 * k] * m[3 * k]; B[1] += m[3 * k] * m[3 * k + 1]; B[2] += m[3 * k] * m[3 * k + 2];
        B[3] += m[3 * k + 1] * m[3 * k + 1]; B[4] += m[3 * k + 1] * m[3 * k + 2]; B[5] += m[3 * k + 2] * m[3 * k + 2];
    }
    double lam[3], V[3][3];
    eig3_full(B[0], B[1], B[2], B[3], B[4], B[5], lam, V);
    double S[3][3];     // (M^T M)^(-1/2) = V diag(1/sqrt(lam)) V^T
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) acc += V[a][k] * V[b][k] / sqrt(fmax(lam[k], 1e-300));
            S[a][b] = acc;
        }
    double R[3][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) R[a][b] = m[3 * a] * S[0][b] + m[3 * a + 1] * S[1][b] + m[3 * a + 2] * S[2][b];
    const double det = R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) - R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) +
                       R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
    if (det < 0)
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) R[a][b] = -R[a][b];          // the reference's `R *= -1` (process.py:226-227)
    double* Tm = T + (long long)blockIdx.x * 16;
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) Tm[4 * a + b] = rigid ? R[a][b] : (a == b ? 1.0 : 0.0);
        Tm[4 * a + 3] = t1[a] - (R[a][0] * t0[0] + R[a][1] * t0[1] + R[a][2] * t0[2]);     // (:232) uses R either way
    }
    Tm[12] = 0; Tm[13] = 0; Tm[14] = 0; Tm[15] = 1;
}

__global__ __launch_bounds__(256) void k_apply_rbm(double* __restrict__ frames, long long N, const double* __restrict__ T) {
    const double* Tm = T + (long long)blockIdx.y * 16;
    double* fr = frames + (long long)blockIdx.y * N * 3;
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < N; v += (long long)gridDim.x * blockDim.x) {
        const double x = fr[3 * v], y = fr[3 * v + 1], z = fr[3 * v + 2];
        fr[3 * v] = Tm[0] * x + Tm[1] * y + Tm[2] * z + Tm[3];
        fr[3 * v + 1] = Tm[4] * x + Tm[5] * y + Tm[6] * z + Tm[7];
        fr[3 * v + 2] = Tm[8] * x + Tm[9] * y + Tm[10] * z + Tm[11];
    }
}

// frames: host (F, N, 3) float64, aligned in place; T_out (optional, host F x 16): the rigid-body matrices
extern "C" int asb_align_frames(asb_ctx* ctx, double* frames, int64_t F, int64_t N, int rigid, double* T_out) {
    if (!ctx || !frames || F < 1 || N < 1) return ASB_ERR_ARG;
    ASB_HIP(ctx, hipSetDevice(ctx->dev));
    double *d = nullptr, *T = nullptr;
    const size_t bytes = (size_t)F * N * 3 * sizeof(double);
    ASB_HIP(ctx, hipMalloc((void**)&d, bytes));
    hipError_t e = hipMalloc((void**)&T, (size_t)F * 16 * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(d, frames, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_procrustes, dim3((unsigned)F), dim3(256), 0, ctx->stream, d, (long long)N, rigid, T);
        long long bx = (N + 255) / 256;
        hipLaunchKernelGGL(k_apply_rbm, dim3((unsigned)(bx < 1024 ? bx : 1024), (unsigned)F), dim3(256), 0, ctx->stream, d,
                           (long long)N, T);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(frames, d, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && T_out) e = hipMemcpyAsync(T_out, T, (size_t)F * 16 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (T) (void)hipFree(T);
    if (e != hipSuccess) ASB_FAIL(ctx, ASB_ERR_HIP, "asb_align_frames: %s", hipGetErrorString(e));
    return ASB_OK;
}
