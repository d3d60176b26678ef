// Fused gather + difference-wise attention for gfx950 -- the core of LocalTrans
// (reference modules/pointnet2_utils.py:518-569 == modules/repsurface_utils.py:488-535):
//
//   e_j = (q - k_j)/sqrt(C),  a = softmax_j(e),  w_j = a_j - sum_j a_j,  ctx = max_j (w_j * v_j)
//
// The reference materialises two gathered [B,S,K,C] tensors and ~10 elementwise passes over
// them.  Here one lane owns one (point, channel): the K gathered k/v values, the softmax, the
// offset and the max over K live in registers; consecutive lanes are consecutive channels, so
// every gathered row is read as contiguous 4-B-per-lane segments (C >= 64: 256 B per wave and
// row).  HBM/L2 algorithmic bytes per point: 4*(2*C + 2*K*C) + 8*K (+C for the saved arg-max).
// Backward uses the closed form of SURVEY.md Appendix A8 and recomputes the softmax; it keeps
// only the 1-byte arg-max from forward.
#include "mpa_common.h"

namespace {

constexpr int KMAX = 16;
constexpr int TPB = 256;

// Computes a_j (softmax over the K energies e_j), their sum o, and returns through w the
// reference's shifted weights.  Order of operations mirrors F.softmax (max-subtracted exp).
template <int K_>
__device__ __forceinline__ void softmax_offset(const float *e, int K, float *a, float &o)
{
    const int k_ = K_ > 0 ? K_ : K;
    float m = e[0];
#pragma unroll
    for (int j = 1; j < (K_ > 0 ? K_ : KMAX); ++j)
        if (j < k_) m = fmaxf(m, e[j]);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < (K_ > 0 ? K_ : KMAX); ++j)
        if (j < k_) {
            a[j] = __expf(e[j] - m);
            sum += a[j];
        }
    o = 0.f;
#pragma unroll
    for (int j = 0; j < (K_ > 0 ? K_ : KMAX); ++j)
        if (j < k_) {
            a[j] = a[j] / sum;
            o += a[j];
        }
}

// ------------------------------------------------------------------ feature branch
template <int K_>
__global__ __launch_bounds__(TPB) void diffattn_fwd_kernel(const float *__restrict__ q, const float *__restrict__ kk,
                                                           const float *__restrict__ vv, int ldkv,
                                                           const int64_t *__restrict__ idx, int N, int S, int K,
                                                           int C, float alpha, long long total,
                                                           float *__restrict__ ctx, uint8_t *__restrict__ argk)
{
    const int k_ = K_ > 0 ? K_ : K;
    for (long long i = blockIdx.x * (long long)TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        long long p = i / C;               // b*S + s
        int c = (int)(i - p * C);
        int b = (int)(p / S);
        const int64_t *nb = idx + p * k_;
        const float qv = q[i];
        float e[K_ > 0 ? K_ : KMAX], v[K_ > 0 ? K_ : KMAX], a[K_ > 0 ? K_ : KMAX];
#pragma unroll
        for (int j = 0; j < (K_ > 0 ? K_ : KMAX); ++j)
            if (j < k_) {
                long long row = ((long long)b * N + mpa_clamp_idx(nb[j], N)) * ldkv + c;
                e[j] = (qv - kk[row]) * alpha;
                v[j] = vv[row];
            }
        float o;
        softmax_offset<K_>(e, k_, a, o);
        float best = (a[0] - o) * v[0];
        int bj = 0;
#pragma unroll
        for (int j = 1; j < (K_ > 0 ? K_ : KMAX); ++j)
            if (j < k_) {
                float t = (a[j] - o) * v[j];
                if (t > best) { best = t; bj = j; }
            }
        ctx[i] = best;
        argk[i] = (uint8_t)bj;
    }
}

template <int K_>
__global__ __launch_bounds__(TPB) void diffattn_bwd_kernel(const float *__restrict__ q, const float *__restrict__ kk,
                                                           const float *__restrict__ vv, int ldkv,
                                                           const int64_t *__restrict__ idx,
                                                           const uint8_t *__restrict__ argk,
                                                           const float *__restrict__ gctx, int N, int S, int K, int C,
                                                           float alpha, long long total, float *__restrict__ gq,
                                                           float *__restrict__ gk, float *__restrict__ gv, int ldg)
{
    const int k_ = K_ > 0 ? K_ : K;
    for (long long i = blockIdx.x * (long long)TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        long long p = i / C;
        int c = (int)(i - p * C);
        int b = (int)(p / S);
        const int64_t *nb = idx + p * k_;
        const float qv = q[i];
        const int ks = argk[i];
        const float g = gctx[i];
        float e[K_ > 0 ? K_ : KMAX], a[K_ > 0 ? K_ : KMAX];
        long long rows[K_ > 0 ? K_ : KMAX];
        float vstar = 0.f;
        long long rstar = 0;
#pragma unroll
        for (int j = 0; j < (K_ > 0 ? K_ : KMAX); ++j)
            if (j < k_) {
                long long r = (long long)b * N + mpa_clamp_idx(nb[j], N);
                rows[j] = r;
                e[j] = (qv - kk[r * ldkv + c]) * alpha;
                if (j == ks) { vstar = vv[r * ldkv + c]; rstar = r; }
            }
        float o;
        softmax_offset<K_>(e, k_, a, o);
        float astar = 0.f;
#pragma unroll
        for (int j = 0; j < (K_ > 0 ? K_ : KMAX); ++j)
            if (j < k_ && j == ks) astar = a[j];
        const float h = g * vstar;
        const float common = -1.0f - astar + o;
        float dq = 0.f;
#pragma unroll
        for (int j = 0; j < (K_ > 0 ? K_ : KMAX); ++j)
            if (j < k_) {
                float de = a[j] * h * ((j == ks ? 1.0f : 0.0f) + common);
                dq += de;
                atomicAdd(gk + rows[j] * ldg + c, -alpha * de);
            }
        gq[i] = alpha * dq;
        atomicAdd(gv + rstar * ldg + c, g * (astar - o));
    }
}

// ------------------------------------------------------------------ xyz branch
// One lane = one output channel with its 12 projection weights in registers; the workgroup
// walks points, whose centre / neighbour coordinates are wave-uniform.
template <int K_>
__global__ __launch_bounds__(TPB) void diffattn_xyz_fwd_kernel(
    const float *__restrict__ xyz, const float *__restrict__ center, const int64_t *__restrict__ idx,
    const float *__restrict__ Wq, const float *__restrict__ bq, const float *__restrict__ Wk,
    const float *__restrict__ bk, const float *__restrict__ Wv, const float *__restrict__ bv, int N, int S, int K,
    int C, float alpha, long long npoints, float *__restrict__ ctx, uint8_t *__restrict__ argk)
{
    const int k_ = K_ > 0 ? K_ : K;
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float wq0 = Wq[c * 3], wq1 = Wq[c * 3 + 1], wq2 = Wq[c * 3 + 2], bqc = bq[c];
    const float wk0 = Wk[c * 3], wk1 = Wk[c * 3 + 1], wk2 = Wk[c * 3 + 2], bkc = bk[c];
    const float wv0 = Wv[c * 3], wv1 = Wv[c * 3 + 1], wv2 = Wv[c * 3 + 2], bvc = bv[c];
    for (long long p = blockIdx.x; p < npoints; p += gridDim.x) {
        int b = (int)(p / S);
        const float cx = center[p * 3], cy = center[p * 3 + 1], cz = center[p * 3 + 2];
        const float qv = fmaf(wq2, cz, fmaf(wq1, cy, fmaf(wq0, cx, bqc)));
        const int64_t *nb = idx + p * k_;
        float e[K_ > 0 ? K_ : KMAX], v[K_ > 0 ? K_ : KMAX], a[K_ > 0 ? K_ : KMAX];
#pragma unroll
        for (int j = 0; j < (K_ > 0 ? K_ : KMAX); ++j)
            if (j < k_) {
                const float *x = xyz + ((long long)b * N + mpa_clamp_idx(nb[j], N)) * 3;
                float rx = x[0] - cx, ry = x[1] - cy, rz = x[2] - cz;
                float kj = fmaf(wk2, rz, fmaf(wk1, ry, fmaf(wk0, rx, bkc)));
                v[j] = fmaf(wv2, rz, fmaf(wv1, ry, fmaf(wv0, rx, bvc)));
                e[j] = (qv - kj) * alpha;
            }
        float o;
        softmax_offset<K_>(e, k_, a, o);
        float best = (a[0] - o) * v[0];
        int bj = 0;
#pragma unroll
        for (int j = 1; j < (K_ > 0 ? K_ : KMAX); ++j)
            if (j < k_) {
                float t = (a[j] - o) * v[j];
                if (t > best) { best = t; bj = j; }
            }
        ctx[p * C + c] = best;
        argk[p * C + c] = (uint8_t)bj;
    }
}

template <int K_>
__global__ __launch_bounds__(TPB) void diffattn_xyz_bwd_kernel(
    const float *__restrict__ xyz, const float *__restrict__ center, const int64_t *__restrict__ idx,
    const float *__restrict__ Wq, const float *__restrict__ bq, const float *__restrict__ Wk,
    const float *__restrict__ bk, const float *__restrict__ Wv, const float *__restrict__ bv,
    const uint8_t *__restrict__ argk, const float *__restrict__ gctx, int N, int S, int K, int C, float alpha,
    long long npoints, float *__restrict__ gWq, float *__restrict__ gbq, float *__restrict__ gWk,
    float *__restrict__ gbk, float *__restrict__ gWv, float *__restrict__ gbv)
{
    const int k_ = K_ > 0 ? K_ : K;
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float wq0 = Wq[c * 3], wq1 = Wq[c * 3 + 1], wq2 = Wq[c * 3 + 2], bqc = bq[c];
    const float wk0 = Wk[c * 3], wk1 = Wk[c * 3 + 1], wk2 = Wk[c * 3 + 2], bkc = bk[c];
    const float wv0 = Wv[c * 3], wv1 = Wv[c * 3 + 1], wv2 = Wv[c * 3 + 2], bvc = bv[c];
    float aq[4] = {0, 0, 0, 0}, ak[4] = {0, 0, 0, 0}, av[4] = {0, 0, 0, 0};   // (dW[0..2], db)
    for (long long p = blockIdx.x; p < npoints; p += gridDim.x) {
        int b = (int)(p / S);
        const float cx = center[p * 3], cy = center[p * 3 + 1], cz = center[p * 3 + 2];
        const float qv = fmaf(wq2, cz, fmaf(wq1, cy, fmaf(wq0, cx, bqc)));
        const int64_t *nb = idx + p * k_;
        const int ks = argk[p * C + c];
        const float g = gctx[p * C + c];
        float e[K_ > 0 ? K_ : KMAX], a[K_ > 0 ? K_ : KMAX];
        float rx[K_ > 0 ? K_ : KMAX], ry[K_ > 0 ? K_ : KMAX], rz[K_ > 0 ? K_ : KMAX];
        float vstar = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
        for (int j = 0; j < (K_ > 0 ? K_ : KMAX); ++j)
            if (j < k_) {
                const float *x = xyz + ((long long)b * N + mpa_clamp_idx(nb[j], N)) * 3;
                rx[j] = x[0] - cx; ry[j] = x[1] - cy; rz[j] = x[2] - cz;
                float kj = fmaf(wk2, rz[j], fmaf(wk1, ry[j], fmaf(wk0, rx[j], bkc)));
                e[j] = (qv - kj) * alpha;
                if (j == ks) {
                    vstar = fmaf(wv2, rz[j], fmaf(wv1, ry[j], fmaf(wv0, rx[j], bvc)));
                    sx = rx[j]; sy = ry[j]; sz = rz[j];
                }
            }
        float o;
        softmax_offset<K_>(e, k_, a, o);
        float astar = 0.f;
#pragma unroll
        for (int j = 0; j < (K_ > 0 ? K_ : KMAX); ++j)
            if (j < k_ && j == ks) astar = a[j];
        const float h = g * vstar;
        const float common = -1.0f - astar + o;
        float dq = 0.f;
#pragma unroll
        for (int j = 0; j < (K_ > 0 ? K_ : KMAX); ++j)
            if (j < k_) {
                float de = a[j] * h * ((j == ks ? 1.0f : 0.0f) + common);
                dq += de;
                float dk = -alpha * de;
                ak[0] = fmaf(dk, rx[j], ak[0]); ak[1] = fmaf(dk, ry[j], ak[1]); ak[2] = fmaf(dk, rz[j], ak[2]);
                ak[3] += dk;
            }
        dq *= alpha;
        aq[0] = fmaf(dq, cx, aq[0]); aq[1] = fmaf(dq, cy, aq[1]); aq[2] = fmaf(dq, cz, aq[2]); aq[3] += dq;
        const float dv = g * (astar - o);
        av[0] = fmaf(dv, sx, av[0]); av[1] = fmaf(dv, sy, av[1]); av[2] = fmaf(dv, sz, av[2]); av[3] += dv;
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        atomicAdd(gWq + c * 3 + d, aq[d]);
        atomicAdd(gWk + c * 3 + d, ak[d]);
        atomicAdd(gWv + c * 3 + d, av[d]);
    }
    atomicAdd(gbq + c, aq[3]);
    atomicAdd(gbk + c, ak[3]);
    atomicAdd(gbv + c, av[3]);
}

inline int grid_for(long long total)
{
    long long g = (total + TPB - 1) / TPB;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int mpa_diffattn_fwd_f32(const float *q, const float *k, const float *v, int ldkv, const int64_t *idx,
                                    int B, int N, int S, int K, int C, float *ctx, uint8_t *argk, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!q || !k || !v || !idx || !ctx || !argk || B <= 0 || N <= 0 || S <= 0 || K <= 0 || C <= 0 || ldkv < C)
        return MPA_EINVAL;
    if (K > KMAX) return MPA_EUNSUPPORTED;
    long long total = (long long)B * S * C;
    float alpha = 1.0f / sqrtf((float)C);
    hipStream_t st = (hipStream_t)stream;
    if (K == 8)
        hipLaunchKernelGGL(diffattn_fwd_kernel<8>, dim3(grid_for(total)), dim3(TPB), 0, st, q, k, v, ldkv, idx, N, S, K,
                           C, alpha, total, ctx, argk);
    else
        hipLaunchKernelGGL(diffattn_fwd_kernel<0>, dim3(grid_for(total)), dim3(TPB), 0, st, q, k, v, ldkv, idx, N, S, K,
                           C, alpha, total, ctx, argk);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_diffattn_bwd_f32(const float *q, const float *k, const float *v, int ldkv, const int64_t *idx,
                                    const uint8_t *argk, const float *grad_ctx, int B, int N, int S, int K, int C,
                                    float *grad_q, float *grad_k, float *grad_v, int ldg, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!q || !k || !v || !idx || !argk || !grad_ctx || !grad_q || !grad_k || !grad_v || B <= 0 || N <= 0 || S <= 0 ||
        K <= 0 || C <= 0 || ldkv < C || ldg < C)
        return MPA_EINVAL;
    if (K > KMAX) return MPA_EUNSUPPORTED;
    long long total = (long long)B * S * C;
    float alpha = 1.0f / sqrtf((float)C);
    hipStream_t st = (hipStream_t)stream;
    if (K == 8)
        hipLaunchKernelGGL(diffattn_bwd_kernel<8>, dim3(grid_for(total)), dim3(TPB), 0, st, q, k, v, ldkv, idx, argk,
                           grad_ctx, N, S, K, C, alpha, total, grad_q, grad_k, grad_v, ldg);
    else
        hipLaunchKernelGGL(diffattn_bwd_kernel<0>, dim3(grid_for(total)), dim3(TPB), 0, st, q, k, v, ldkv, idx, argk,
                           grad_ctx, N, S, K, C, alpha, total, grad_q, grad_k, grad_v, ldg);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_diffattn_xyz_fwd_f32(const float *xyz, const float *center, const int64_t *idx, const float *Wq,
                                        const float *bq, const float *Wk, const float *bk, const float *Wv,
                                        const float *bv, int B, int N, int S, int K, int C, float *ctx, uint8_t *argk,
                                        void *stream)
{
    MPA_CLEAR_ERROR();
    if (!xyz || !center || !idx || !Wq || !bq || !Wk || !bk || !Wv || !bv || !ctx || !argk || B <= 0 || N <= 0 ||
        S <= 0 || K <= 0 || C <= 0)
        return MPA_EINVAL;
    if (K > KMAX) return MPA_EUNSUPPORTED;
    long long np = (long long)B * S;
    int bx = C >= TPB ? TPB : ((C + 63) / 64) * 64;
    dim3 grid((unsigned)(np > 8192 ? 8192 : np), mpa_ceil_div(C, bx));
    float alpha = 1.0f / sqrtf((float)C);
    hipStream_t st = (hipStream_t)stream;
    if (K == 8)
        hipLaunchKernelGGL(diffattn_xyz_fwd_kernel<8>, grid, dim3(bx), 0, st, xyz, center, idx, Wq, bq, Wk, bk, Wv, bv, N,
                           S, K, C, alpha, np, ctx, argk);
    else
        hipLaunchKernelGGL(diffattn_xyz_fwd_kernel<0>, grid, dim3(bx), 0, st, xyz, center, idx, Wq, bq, Wk, bk, Wv, bv, N,
                           S, K, C, alpha, np, ctx, argk);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_diffattn_xyz_bwd_f32(const float *xyz, const float *center, const int64_t *idx, const float *Wq,
                                        const float *bq, const float *Wk, const float *bk, const float *Wv,
                                        const float *bv, const uint8_t *argk, const float *grad_ctx, int B, int N,
                                        int S, int K, int C, float *gWq, float *gbq, float *gWk, float *gbk,
                                        float *gWv, float *gbv, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!xyz || !center || !idx || !Wq || !bq || !Wk || !bk || !Wv || !bv || !argk || !grad_ctx || !gWq || !gbq ||
        !gWk || !gbk || !gWv || !gbv || B <= 0 || N <= 0 || S <= 0 || K <= 0 || C <= 0)
        return MPA_EINVAL;
    if (K > KMAX) return MPA_EUNSUPPORTED;
    long long np = (long long)B * S;
    int bx = C >= TPB ? TPB : ((C + 63) / 64) * 64;
    dim3 grid((unsigned)(np > 1024 ? 1024 : np), mpa_ceil_div(C, bx));
    float alpha = 1.0f / sqrtf((float)C);
    hipStream_t st = (hipStream_t)stream;
    if (K == 8)
        hipLaunchKernelGGL(diffattn_xyz_bwd_kernel<8>, grid, dim3(bx), 0, st, xyz, center, idx, Wq, bq, Wk, bk, Wv, bv,
                           argk, grad_ctx, N, S, K, C, alpha, np, gWq, gbq, gWk, gbk, gWv, gbv);
    else
        hipLaunchKernelGGL(diffattn_xyz_bwd_kernel<0>, grid, dim3(bx), 0, st, xyz, center, idx, Wq, bq, Wk, bk, Wv, bv,
                           argk, grad_ctx, N, S, K, C, alpha, np, gWq, gbq, gWk, gbk, gWv, gbv);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}
