// The tails of the models' heads as single launches: log-softmax, the label-smoothed losses and the classification
// head's max | mean pooling over the points of the last state.
//
//   log_softmax over the class axis                    models/repsurf/repsurf_ssg_umb.py:67
//   label-smoothed NLL on log-probabilities            util/utils.py:74-88            (SmoothClsLoss)
//   label-smoothed cross entropy on logits             models/repsurf/pointnet2_part_seg_msg.py:159-180 (get_loss)
//   cat(max over points, mean over points)             modules/repsurface_utils.py:629-633
//
// In torch these are 8-16 elementwise / reduction launches of ~5 us each per pass (scatter of the one-hot, two
// multiplications, negation, two reductions, the softmax pair, concatenation, ...) on tensors of a few KB in the
// classification step, and whole passes over the [B*N, 50] logits in part segmentation.  One wave per row; the
// class axis lives on the lanes.
#include "mpa_common.h"
#include "mpa_bf16.h"

namespace {

__device__ __forceinline__ float wave_max_f32(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// rows of x [M][C]: m = max, lse = m + log(sum exp(x - m)); returns lse (wave-uniform)
__device__ __forceinline__ float row_lse(const float *__restrict__ row, int C, int lane)
{
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, row[c]);
    m = wave_max_f32(m);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(row[c] - m);
    s = wave_sum_f32(s);
    return m + logf(s);
}

__global__ __launch_bounds__(256) void log_softmax_fwd_kernel(const float *__restrict__ x, int M, int C,
                                                              float *__restrict__ y)
{
    const int lane = threadIdx.x & 63;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < M; r += gridDim.x * 4) {
        const float *row = x + (size_t)r * C;
        // (exact functions here: the log-probabilities are a model OUTPUT compared with the reference at 1e-4)
        float m = -INFINITY;
        for (int c = lane; c < C; c += 64) m = fmaxf(m, row[c]);
        m = wave_max_f32(m);
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += expf(row[c] - m);
        s = wave_sum_f32(s);
        const float lse = m + logf(s);
        for (int c = lane; c < C; c += 64) y[(size_t)r * C + c] = row[c] - lse;
    }
}

// gx = g - exp(y) * sum_c g
__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(const float *__restrict__ y, const float *__restrict__ g,
                                                              int M, int C, float *__restrict__ gx)
{
    const int lane = threadIdx.x & 63;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < M; r += gridDim.x * 4) {
        const float *yr = y + (size_t)r * C, *gr = g + (size_t)r * C;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += gr[c];
        s = wave_sum_f32(s);
        for (int c = lane; c < C; c += 64) gx[(size_t)r * C + c] = gr[c] - expf(yr[c]) * s;
    }
}

// Label-smoothed loss of one row: -( (1-eps) * lp[t] + eps/(C-1) * sum_{c != t} lp[c] ), lp = x - lse (LOGITS) or x.
// partial[block] = sum of the block's rows; rows are dealt to the blocks in contiguous chunks, so the two-stage sum has
// a fixed order (deterministic, unlike an atomic accumulator).
template <bool LOGITS>
__global__ __launch_bounds__(256) void smooth_loss_fwd_kernel(const float *__restrict__ x,
                                                              const int64_t *__restrict__ target, int M, int C,
                                                              float eps, int rows_per_block, float *__restrict__ lse_out,
                                                              float *__restrict__ partial)
{
    __shared__ float wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    const float off = eps / (float)(C - 1), on = 1.f - eps;
    float acc = 0.f;
    for (int r = r0 + wave; r < r1; r += 4) {
        const float *row = x + (size_t)r * C;
        float lse = 0.f;
        if (LOGITS) {
            lse = row_lse(row, C, lane);
            if (lane == 0) lse_out[r] = lse;
        }
        const int t = (int)mpa_clamp_idx(target[r], C);
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += (c == t ? on : off) * (row[c] - lse);
        acc -= wave_sum_f32(s);
    }
    if (lane == 0) wsum[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ __launch_bounds__(256) void smooth_loss_finish_kernel(const float *__restrict__ partial, int nblocks,
                                                                 float inv_M, float *__restrict__ loss)
{
    __shared__ float wsum[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[i];
    s = wave_sum_f32(s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) * inv_M;
}

// d loss / d x, scaled by the upstream gradient g[0]:  LOGITS: (softmax(x) - w) / M;  log-probabilities: -w / M
template <bool LOGITS>
__global__ __launch_bounds__(256) void smooth_loss_bwd_kernel(const float *__restrict__ x,
                                                              const int64_t *__restrict__ target,
                                                              const float *__restrict__ lse,
                                                              const float *__restrict__ g, int M, int C, float eps,
                                                              float *__restrict__ gx)
{
    const float scale = g[0] / (float)M;
    const float off = eps / (float)(C - 1), on = 1.f - eps;
    const long long total = (long long)M * C;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / C), c = (int)(i - (long long)r * C);
        const int t = (int)mpa_clamp_idx(target[r], C);
        const float w = c == t ? on : off;
        gx[i] = scale * (LOGITS ? expf(x[i] - lse[r]) - w : -w);
    }
}

// out[b][c] = max_p x[b][p][c], out[b][C + c] = mean_p x[b][p][c]; arg[b][c] = first row attaining the maximum.
// One lane per (cloud, channel), P <= a few dozen rows read coalesced across the channels.
__global__ __launch_bounds__(256) void pool_max_mean_fwd_kernel(const float *__restrict__ x, int B, int P, int C,
                                                                float *__restrict__ out, int *__restrict__ arg)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    const float *p = x + (size_t)b * P * C + c;
    float best = p[0], sum = p[0];
    int bi = 0;
    for (int r = 1; r < P; ++r) {
        const float v = p[(size_t)r * C];
        sum += v;
        if (v > best || (v != v && best == best)) { best = v; bi = r; }      // first maximum; a NaN is kept (torch.max)
    }
    out[(size_t)b * 2 * C + c] = best;
    out[(size_t)b * 2 * C + C + c] = sum / (float)P;
    arg[i] = bi;
}

__global__ __launch_bounds__(256) void pool_max_mean_bwd_kernel(const float *__restrict__ g, const int *__restrict__ arg,
                                                                int B, int P, int C, float *__restrict__ gx)
{
    const long long total = (long long)B * P * C;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long bp = i / C;
        const int p = (int)(bp % P), b = (int)(bp / P);
        const float gm = g[(size_t)b * 2 * C + c], ga = g[(size_t)b * 2 * C + C + c];
        gx[i] = (arg[(size_t)b * C + c] == p ? gm : 0.f) + ga / (float)P;
    }
}

inline int row_grid(int M) { const int g = (M + 3) / 4; return g > 4096 ? 4096 : (g < 1 ? 1 : g); }

}  // namespace

extern "C" int mpa_log_softmax_fwd_f32(const float *x, int M, int C, float *y, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!x || !y || M <= 0 || C <= 0) return MPA_EINVAL;
    hipLaunchKernelGGL(log_softmax_fwd_kernel, dim3(row_grid(M)), dim3(256), 0, (hipStream_t)stream, x, M, C, y);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_log_softmax_bwd_f32(const float *y, const float *grad_y, int M, int C, float *grad_x, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!y || !grad_y || !grad_x || M <= 0 || C <= 0) return MPA_EINVAL;
    hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3(row_grid(M)), dim3(256), 0, (hipStream_t)stream, y, grad_y, M, C,
                       grad_x);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

// rows per block: 16 (four per wave) keeps every CU busy from a few thousand rows on and a 64-row classification batch on
// four workgroups; the second kernel adds the <= M/16 block sums in a fixed order
static inline int smooth_loss_rows_per_block(int M) { return M > (1 << 20) ? 64 : 16; }

extern "C" int mpa_smooth_loss_workspace_floats(int M)
{
    const int rows_per_block = smooth_loss_rows_per_block(M);
    return (M + rows_per_block - 1) / rows_per_block;
}

extern "C" int mpa_smooth_loss_fwd_f32(const float *x, const int64_t *target, int M, int C, float eps, int from_logits,
                                       float *lse, float *partial, float *loss, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!x || !target || !partial || !loss || M <= 0 || C <= 1 || (from_logits && !lse)) return MPA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int rows_per_block = smooth_loss_rows_per_block(M);
    const int nblocks = (M + rows_per_block - 1) / rows_per_block;
    if (from_logits)
        hipLaunchKernelGGL(smooth_loss_fwd_kernel<true>, dim3(nblocks), dim3(256), 0, st, x, target, M, C, eps,
                           rows_per_block, lse, partial);
    else
        hipLaunchKernelGGL(smooth_loss_fwd_kernel<false>, dim3(nblocks), dim3(256), 0, st, x, target, M, C, eps,
                           rows_per_block, lse, partial);
    hipLaunchKernelGGL(smooth_loss_finish_kernel, dim3(1), dim3(256), 0, st, partial, nblocks, 1.f / (float)M, loss);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_smooth_loss_bwd_f32(const float *x, const int64_t *target, const float *lse, const float *grad_loss,
                                       int M, int C, float eps, int from_logits, float *grad_x, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!x || !target || !grad_loss || !grad_x || M <= 0 || C <= 1 || (from_logits && !lse)) return MPA_EINVAL;
    const long long total = (long long)M * C;
    long long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    if (from_logits)
        hipLaunchKernelGGL(smooth_loss_bwd_kernel<true>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, target,
                           lse, grad_loss, M, C, eps, grad_x);
    else
        hipLaunchKernelGGL(smooth_loss_bwd_kernel<false>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, target,
                           lse, grad_loss, M, C, eps, grad_x);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_pool_max_mean_fwd_f32(const float *x, int B, int P, int C, float *out, int *arg, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!x || !out || !arg || B <= 0 || P <= 0 || C <= 0) return MPA_EINVAL;
    hipLaunchKernelGGL(pool_max_mean_fwd_kernel, dim3(mpa_ceil_div((long long)B * C, 256)), dim3(256), 0,
                       (hipStream_t)stream, x, B, P, C, out, arg);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_pool_max_mean_bwd_f32(const float *grad_out, const int *arg, int B, int P, int C, float *grad_x,
                                         void *stream)
{
    MPA_CLEAR_ERROR();
    if (!grad_out || !arg || !grad_x || B <= 0 || P <= 0 || C <= 0) return MPA_EINVAL;
    const long long total = (long long)B * P * C;
    long long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(pool_max_mean_bwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, grad_out, arg, B, P,
                       C, grad_x);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

// ---- sum of n (<= 8) equally shaped tensors in one pass: the gradient of a tensor that several consumers read
// (ops.fanout): autograd would add the n contributions pairwise, n - 1 launches that re-read and re-write the running
// sum; here every contribution is read once and the sum written once (fp32 accumulation, one rounding for bf16 rows).
namespace {
struct AddNArgs { const void *src[8]; long long ld[8]; };

template <typename T>
__global__ __launch_bounds__(256) void add_n_kernel(const AddNArgs a, int n, long long rows, int c4, T *__restrict__ out)
{
    const long long count4 = rows * c4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < count4;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / c4;
        const int c = (int)(i - r * c4) * 4;
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (u < n) v[u] = mpa_ld4<T>(static_cast<const T *>(a.src[u]) + r * a.ld[u] + c);
        float4 s = v[0];
#pragma unroll
        for (int u = 1; u < 8; ++u)
            if (u < n) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        mpa_st4<T>(out + 4 * i, s);
    }
}

template <typename T>
int add_n_any(const void *const *srcs, const long long *lds, int n, long long rows, int C, T *out, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!srcs || !out || n < 1 || n > 8 || rows <= 0 || C <= 0) return MPA_EINVAL;
    if (C % 4 != 0 || ((uintptr_t)out & mpa_vec4_align<T>::mask) != 0) return MPA_EUNSUPPORTED;
    AddNArgs a;
    for (int u = 0; u < 8; ++u) {
        a.src[u] = u < n ? srcs[u] : nullptr;
        a.ld[u] = u < n ? (lds ? lds[u] : C) : 0;
        if (u < n && (!srcs[u] || ((uintptr_t)srcs[u] & mpa_vec4_align<T>::mask) != 0 || a.ld[u] < C || a.ld[u] % 4 != 0))
            return MPA_EUNSUPPORTED;
    }
    long long g = (rows * (C / 4) + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(add_n_kernel<T>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, a, n, rows, C / 4, out);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}
}  // namespace

extern "C" int mpa_add_n_f32(const float *const *srcs, const long long *lds, int n, long long rows, int C, float *out,
                             void *stream)
{
    return add_n_any<float>(reinterpret_cast<const void *const *>(srcs), lds, n, rows, C, out, stream);
}

extern "C" int mpa_add_n_bf16(const mpa_bf16 *const *srcs, const long long *lds, int n, long long rows, int C,
                              mpa_bf16 *out, void *stream)
{
    return add_n_any<bf16_t>(reinterpret_cast<const void *const *>(srcs), lds, n, rows, C, reinterpret_cast<bf16_t *>(out),
                             stream);
}
