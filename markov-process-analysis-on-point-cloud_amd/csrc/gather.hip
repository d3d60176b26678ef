// Row gathers / scatters for gfx950: index_points (reference modules/pointnet2_utils.py:64-81),
// upsample (:13-50) and the 3-NN interpolation of PointNetFeaturePropagation (:899-906).
// All are HBM/L2-bound row moves: one lane moves a float4 of a C-float row, so a wave covers
// 1 KiB of contiguous row bytes per instruction; scatters use no-return float atomics on whole
// rows (the shape that runs at the full atomic rate, MI355X guide "Global float atomics").
#include "mpa_common.h"
#include "mpa_bf16.h"
#include "csr_build.h"
#include <cstdlib>

namespace {

constexpr int TPB = 256;

// out[b,m,:] = points[b, idx[b,m], :]
template <typename V>
__global__ void gather_fwd_kernel(const float *__restrict__ points, const int64_t *__restrict__ idx, int N, int M,
                                  int CV, long long total, float *__restrict__ out)
{
    // total = B*M*CV vector elements; CV = C / lanes-per-vector
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long row = i / CV;
        int c = (int)(i - row * CV);
        int b = (int)(row / M);
        long long src = (long long)b * N + mpa_clamp_idx(idx[row], N);
        reinterpret_cast<V *>(out)[i] = reinterpret_cast<const V *>(points)[src * CV + c];
    }
}

// grad_points[b, idx[b,m], :] += grad_out[b,m,:]
template <typename T>
__global__ void gather_bwd_kernel(const T *__restrict__ grad_out, const int64_t *__restrict__ idx, int N, int M,
                                  int C, long long total, float *__restrict__ grad_points)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long row = i / C;
        int c = (int)(i - row * C);
        int b = (int)(row / M);
        long long dst = (long long)b * N + mpa_clamp_idx(idx[row], N);
        atomicAdd(grad_points + dst * C + c, mpa_ld1<T>(grad_out + i));
    }
}

// bf16 rows: grad_points[b, idx[b,m], :] += grad_out[b,m,:] directly on the bf16 destination, two channels (one
// dword) per lane with a compare-and-swap loop -- exact for rows listed once (the FPS index maps of the models: the
// result is the bf16 sum the separate zero-fill / fp32 scatter / cast / add produced), race-free for repeated rows
// (each addition is then rounded to bf16 in turn).
__global__ void gather_bwd_into_bf16_kernel(const unsigned *__restrict__ grad_out, const int64_t *__restrict__ idx, int N,
                                            int M, int C2, long long total, unsigned *__restrict__ grad_points)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / C2;
        const int c = (int)(i - row * C2);
        const int b = (int)(row / M);
        const long long dst = (long long)b * N + mpa_clamp_idx(idx[row], N);
        const unsigned g = grad_out[i];
        unsigned *p = grad_points + dst * C2 + c;
        unsigned old = *p, assumed;
        do {
            assumed = old;
            const unsigned sum = mpa_pack_bf16x2(mpa_bf16_lo(assumed) + mpa_bf16_lo(g), mpa_bf16_hi(assumed) + mpa_bf16_hi(g));
            old = atomicCAS(p, assumed, sum);
        } while (old != assumed);
    }
}

// upsample forward, scatter phase: every coarse row s adds itself to the fine rows it lists
// (once per distinct fine index: scatter_ semantics) and bumps their divisor if p[s][0] != 0.
__global__ void upsample_scatter_kernel(const float *__restrict__ points, const int64_t *__restrict__ knn, int S,
                                        int K, int Nf, int C, long long total, float *__restrict__ out,
                                        float *__restrict__ cnt)
{
    // one thread per (b, s, k, c)
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long e = i / C;            // (b*S + s)*K + k
        int c = (int)(i - e * C);
        long long bs = e / K;
        int k = (int)(e - bs * K);
        int b = (int)(bs / S);
        const int64_t *row = knn + bs * K;
        int64_t n = mpa_clamp_idx(row[k], Nf);
        bool dup = false;
        for (int j = 0; j < k; ++j) dup |= (mpa_clamp_idx(row[j], Nf) == n);
        if (dup) continue;
        float p = points[bs * C + c];
        long long dst = (long long)b * Nf + n;
        atomicAdd(out + dst * C + c, p);
        if (c == 0 && p != 0.0f) atomicAdd(cnt + dst, 1.0f);
    }
}

__global__ void upsample_divide_kernel(float *__restrict__ out, const float *__restrict__ cnt, int C, long long total)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        float n = cnt[i / C];
        n = n == 0.0f ? 1.0f : n;
        out[i] = out[i] / n;
    }
}

// upsample forward as a gather over the inverted neighbour table (csr_build.h): lane = (fine row n, V
// channels) sums the coarse rows that list n (once per coarse row: scatter_ semantics), counts those
// whose channel 0 is non-zero and divides -- no float atomics (B*S*K*C of them ran at the chip's atomic
// rate: 70 us per call in the part-seg decoder), no clearing pass, no separate divide.
template <int V, typename T>
__global__ __launch_bounds__(TPB) void upsample_gather_kernel(const T *__restrict__ points,
                                                              const int64_t *__restrict__ knn,
                                                              const int *__restrict__ rowptr,
                                                              const int *__restrict__ entries, int S, int K, int Nf,
                                                              int C, int lanes_per_row, T *__restrict__ out,
                                                              float *__restrict__ cnt)
{
    const int rl = threadIdx.x / lanes_per_row, cl = threadIdx.x % lanes_per_row;
    const int rpb = TPB / lanes_per_row;
    const int b = blockIdx.y;
    const int n = blockIdx.x * rpb + rl;
    if (n >= Nf) return;
    const int *rp = rowptr + (size_t)b * (Nf + 1);
    const int beg = rp[n], end = rp[n + 1];
    const int *en = entries + (size_t)b * S * K;
    const T *pb = points + (size_t)b * S * C;
    for (int c = cl * V; c < C; c += lanes_per_row * V) {
        float acc[V];
#pragma unroll
        for (int u = 0; u < V; ++u) acc[u] = 0.f;
        float div = 0.f;
        // four entries at a time, the entry words first, then their rows, all in flight together (one entry after the
        // other was two dependent round trips each, plus up to K-1 index loads for the repeat test).  A coarse row that
        // lists this fine row more than once counts once (scatter_ semantics): its entries share their coarse row, so
        // an entry is skipped when an earlier entry of the list has the same one -- compared in registers inside the
        // chunk, against the (L1-resident) earlier chunks for the few rows with more than four entries
        for (int e0 = beg; e0 < end; e0 += 4) {
            int srow[4];
            bool use[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                srow[u] = en[min(e0 + u, end - 1)] / K;
                use[u] = e0 + u < end;
            }
#pragma unroll
            for (int u = 1; u < 4; ++u)
#pragma unroll
                for (int w = 0; w < u; ++w) use[u] = use[u] && srow[w] != srow[u];
            for (int ep = beg; ep < e0; ++ep) {
                const int sp = en[ep] / K;
#pragma unroll
                for (int u = 0; u < 4; ++u) use[u] = use[u] && sp != srow[u];
            }
            float4 v[4];
            float first[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const T *row = pb + (size_t)srow[u] * C;
                if constexpr (V == 4) {
                    v[u] = use[u] ? mpa_ld4<T>(row + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                    v[u] = make_float4(use[u] ? mpa_ld1<T>(row + c) : 0.f, 0.f, 0.f, 0.f);
                }
                first[u] = use[u] ? mpa_ld1<T>(row) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[0] += v[u].x;
                if constexpr (V == 4) { acc[1] += v[u].y; acc[2] += v[u].z; acc[3] += v[u].w; }
                div += first[u] != 0.0f ? 1.f : 0.f;
            }
        }
        if (c == 0) cnt[(size_t)b * Nf + n] = div;
        const float d = div == 0.f ? 1.f : div;
        T *o = out + ((size_t)b * Nf + n) * C + c;
        if constexpr (V == 4) {
            mpa_st4<T>(o, make_float4(acc[0] / d, acc[1] / d, acc[2] / d, acc[3] / d));
        } else {
            mpa_st1<T>(o, acc[0] / d);
        }
    }
}

// grad_points[b,s,:] = sum over distinct n in knn[b,s,:] of grad_out[b,n,:] / max(cnt[b,n],1)
template <typename T>
__global__ void upsample_bwd_kernel(const T *__restrict__ grad_out, const int64_t *__restrict__ knn,
                                    const float *__restrict__ cnt, int S, int K, int Nf, int C, long long total,
                                    T *__restrict__ grad_points)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long bs = i / C;
        int c = (int)(i - bs * C);
        int b = (int)(bs / S);
        const int64_t *row = knn + bs * K;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) {
            int64_t n = mpa_clamp_idx(row[k], Nf);
            bool dup = false;
            for (int j = 0; j < k; ++j) dup |= (mpa_clamp_idx(row[j], Nf) == n);
            if (dup) continue;
            long long src = (long long)b * Nf + n;
            float d = cnt[src];
            d = d == 0.0f ? 1.0f : d;
            acc += mpa_ld1<T>(grad_out + src * C + c) / d;
        }
        mpa_st1<T>(grad_points + i, acc);
    }
}

// The same with a lane owning 4 consecutive channels of a coarse point: the K indices, their duplicate mask and
// the K divisors are read once per lane (the scalar form re-reads the index row per channel and per duplicate test),
// every gathered row arrives as one 8- / 16-byte load per lane, all K of them in flight together.
template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_v4_kernel(const T *__restrict__ grad_out,
                                                              const int64_t *__restrict__ knn,
                                                              const float *__restrict__ cnt, int S, int K, int Nf, int C,
                                                              long long total4, T *__restrict__ grad_points)
{
    constexpr int KM = 16;
    const int c4n = C >> 2;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const long long bs = i / c4n;
        const int c = (int)(i - bs * c4n) << 2;
        const int b = (int)(bs / S);
        const int64_t *row = knn + bs * K;
        int n[KM];
        float d[KM];
        float4 g[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) n[k] = k < K ? (int)mpa_clamp_idx(row[k], Nf) : -1;
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            bool dup = false;
#pragma unroll
            for (int j = 0; j < k; ++j) dup |= (n[j] == n[k]);
            if (dup) n[k] = -1;                          // a point listed twice contributes once (reference :44-46)
        }
#pragma unroll
        for (int k = 0; k < KM; ++k)
            if (n[k] >= 0) {
                const long long src = (long long)b * Nf + n[k];
                d[k] = cnt[src];
                g[k] = mpa_ld4<T>(grad_out + src * C + c);
            }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < KM; ++k)
            if (n[k] >= 0) {
                const float dd = d[k] == 0.0f ? 1.0f : d[k];
                acc.x += g[k].x / dd; acc.y += g[k].y / dd; acc.z += g[k].z / dd; acc.w += g[k].w / dd;
            }
        mpa_st4<T>(grad_points + bs * C + c, acc);
    }
}

__device__ __forceinline__ void interp_weights(const float *d, float w[3])
{
    float r0 = 1.0f / (d[0] + 1e-8f), r1 = 1.0f / (d[1] + 1e-8f), r2 = 1.0f / (d[2] + 1e-8f);
    float nrm = (r0 + r1) + r2;
    w[0] = r0 / nrm; w[1] = r1 / nrm; w[2] = r2 / nrm;
}

template <typename T>
__global__ void interp_fwd_kernel(const T *__restrict__ points2, const int64_t *__restrict__ idx,
                                  const float *__restrict__ dist, int Nq, int Nb, int C, long long total,
                                  T *__restrict__ out)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long row = i / C;
        int c = (int)(i - row * C);
        int b = (int)(row / Nq);
        float w[3];
        interp_weights(dist + row * 3, w);
        const int64_t *id = idx + row * 3;
        const T *p = points2 + (long long)b * Nb * C + c;
        float acc = mpa_ld1<T>(p + mpa_clamp_idx(id[0], Nb) * C) * w[0];
        acc += mpa_ld1<T>(p + mpa_clamp_idx(id[1], Nb) * C) * w[1];
        acc += mpa_ld1<T>(p + mpa_clamp_idx(id[2], Nb) * C) * w[2];
        mpa_st1<T>(out + i, acc);
    }
}

// gradients are summed in fp32 whatever the storage type of the incoming gradient (rows are listed by several
// query points: float atomics on a cleared fp32 buffer; the bf16 caller rounds once afterwards)
template <typename T>
__global__ void interp_bwd_kernel(const T *__restrict__ grad_out, const int64_t *__restrict__ idx,
                                  const float *__restrict__ dist, int Nq, int Nb, int C, long long total,
                                  float *__restrict__ grad_points2)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long row = i / C;
        int c = (int)(i - row * C);
        int b = (int)(row / Nq);
        float w[3];
        interp_weights(dist + row * 3, w);
        const int64_t *id = idx + row * 3;
        float *p = grad_points2 + (long long)b * Nb * C + c;
        float g = mpa_ld1<T>(grad_out + i);
        atomicAdd(p + mpa_clamp_idx(id[0], Nb) * C, g * w[0]);
        atomicAdd(p + mpa_clamp_idx(id[1], Nb) * C, g * w[1]);
        atomicAdd(p + mpa_clamp_idx(id[2], Nb) * C, g * w[2]);
    }
}

inline int grid_for(long long total)
{
    long long g = (total + TPB - 1) / TPB;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

// plain kernel instead of hipMemsetAsync: inside a captured HIP graph the clear is then an ordinary
// kernel node like its neighbours (no memset nodes anywhere on the path)
__global__ void clear_kernel(float *__restrict__ a, long long na, float *__restrict__ b, long long nb)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < na; i += stride) a[i] = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nb; i += stride) b[i] = 0.f;
}

}  // namespace

extern "C" int mpa_gather_fwd_f32(const float *points, const int64_t *idx, int B, int N, int M, int C, float *out,
                                  void *stream)
{
    MPA_CLEAR_ERROR();
    if (!points || !idx || !out || B <= 0 || N <= 0 || M <= 0 || C <= 0) return MPA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    bool al16 = (((uintptr_t)points | (uintptr_t)out) & 15) == 0;
    if ((C & 3) == 0 && al16) {
        long long total = (long long)B * M * (C / 4);
        hipLaunchKernelGGL(gather_fwd_kernel<float4>, dim3(grid_for(total)), dim3(TPB), 0, st, points, idx, N, M,
                           C / 4, total, out);
    } else {
        long long total = (long long)B * M * C;
        hipLaunchKernelGGL(gather_fwd_kernel<float>, dim3(grid_for(total)), dim3(TPB), 0, st, points, idx, N, M, C,
                           total, out);
    }
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_gather_bwd_f32(const float *grad_out, const int64_t *idx, int B, int N, int M, int C,
                                  float *grad_points, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!grad_out || !idx || !grad_points || B <= 0 || N <= 0 || M <= 0 || C <= 0) return MPA_EINVAL;
    long long total = (long long)B * M * C;
    hipLaunchKernelGGL(gather_bwd_kernel<float>, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, grad_out, idx, N,
                       M, C, total, grad_points);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_gather_fwd_bf16(const mpa_bf16 *points, const int64_t *idx, int B, int N, int M, int C, mpa_bf16 *out,
                                   void *stream)
{
    MPA_CLEAR_ERROR();
    if (!points || !idx || !out || B <= 0 || N <= 0 || M <= 0 || C <= 0) return MPA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // a row move: bf16 rows are copied as 16-B / 4-B pieces of their bytes
    const float *pf = reinterpret_cast<const float *>(points);
    float *of = reinterpret_cast<float *>(out);
    const bool al16 = (((uintptr_t)points | (uintptr_t)out) & 15) == 0;
    if ((C & 7) == 0 && al16) {
        long long total = (long long)B * M * (C / 8);
        hipLaunchKernelGGL(gather_fwd_kernel<float4>, dim3(grid_for(total)), dim3(TPB), 0, st, pf, idx, N, M, C / 8, total, of);
    } else if ((C & 1) == 0 && (((uintptr_t)points | (uintptr_t)out) & 3) == 0) {
        long long total = (long long)B * M * (C / 2);
        hipLaunchKernelGGL(gather_fwd_kernel<float>, dim3(grid_for(total)), dim3(TPB), 0, st, pf, idx, N, M, C / 2, total, of);
    } else {
        return MPA_EUNSUPPORTED;
    }
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_gather_bwd_bf16(const mpa_bf16 *grad_out, const int64_t *idx, int B, int N, int M, int C,
                                   float *grad_points, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!grad_out || !idx || !grad_points || B <= 0 || N <= 0 || M <= 0 || C <= 0) return MPA_EINVAL;
    long long total = (long long)B * M * C;
    hipLaunchKernelGGL(gather_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream,
                       reinterpret_cast<const bf16_t *>(grad_out), idx, N, M, C, total, grad_points);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_gather_bwd_into_bf16(const mpa_bf16 *grad_out, const int64_t *idx, int B, int N, int M, int C,
                                        mpa_bf16 *grad_points, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!grad_out || !idx || !grad_points || B <= 0 || N <= 0 || M <= 0 || C <= 0) return MPA_EINVAL;
    if ((C & 1) != 0 || ((((uintptr_t)grad_out | (uintptr_t)grad_points)) & 3) != 0) return MPA_EUNSUPPORTED;
    const long long total = (long long)B * M * (C / 2);
    hipLaunchKernelGGL(gather_bwd_into_bf16_kernel, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned *>(grad_out), idx, N, M, C / 2, total,
                       reinterpret_cast<unsigned *>(grad_points));
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" size_t mpa_upsample_workspace_bytes(int B, int S, int K, int Nf)
{
    if (B <= 0 || S <= 0 || K <= 0 || Nf <= 0 || Nf > CSR_MAX_N || B > 65535 || (long long)S * K > 0x7fffffffLL) return 0;
    return (((size_t)B * (Nf + 1) * 4 + 255) & ~(size_t)255) + (size_t)B * S * K * 4;
}

extern "C" int mpa_upsample_mean_fwd_f32(const float *points, const int64_t *knn_idx, int B, int S, int K, int Nf,
                                         int C, float *out, float *cnt, void *workspace, size_t workspace_bytes,
                                         void *stream)
{
    MPA_CLEAR_ERROR();
    if (!points || !knn_idx || !out || !cnt || B <= 0 || S <= 0 || K <= 0 || Nf <= 0 || C <= 0) return MPA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const size_t need = mpa_upsample_workspace_bytes(B, S, K, Nf);
    static const bool force_atomic = getenv("MPA_UPSAMPLE_ATOMIC") != nullptr;
    if (workspace && need && workspace_bytes >= need && ((uintptr_t)workspace & 15) == 0 && !force_atomic) {
        int *rowptr = reinterpret_cast<int *>(workspace);
        int *entries = reinterpret_cast<int *>((char *)workspace + (((size_t)B * (Nf + 1) * 4 + 255) & ~(size_t)255));
        launch_csr_build(knn_idx, B, Nf, S * K, rowptr, entries, st, nullptr, true);
        const bool v4 = (C & 3) == 0 && ((((uintptr_t)points | (uintptr_t)out)) & 15) == 0;
        const int per = v4 ? C / 4 : C;
        int lanes = 1;
        while (lanes < per && lanes < TPB) lanes <<= 1;
        const dim3 grid(mpa_ceil_div(Nf, TPB / lanes), B);
        if (v4)
            hipLaunchKernelGGL((upsample_gather_kernel<4, float>), grid, dim3(TPB), 0, st, points, knn_idx, rowptr, entries, S, K,
                               Nf, C, lanes, out, cnt);
        else
            hipLaunchKernelGGL((upsample_gather_kernel<1, float>), grid, dim3(TPB), 0, st, points, knn_idx, rowptr, entries, S, K,
                               Nf, C, lanes, out, cnt);
        MPA_LAUNCH_CHECK();
        return MPA_OK;
    }
    hipLaunchKernelGGL(clear_kernel, dim3(grid_for((long long)B * Nf * C)), dim3(TPB), 0, st, out, (long long)B * Nf * C,
                       cnt, (long long)B * Nf);
    long long total = (long long)B * S * K * C;
    hipLaunchKernelGGL(upsample_scatter_kernel, dim3(grid_for(total)), dim3(TPB), 0, st, points, knn_idx, S, K, Nf, C,
                       total, out, cnt);
    long long tot2 = (long long)B * Nf * C;
    hipLaunchKernelGGL(upsample_divide_kernel, dim3(grid_for(tot2)), dim3(TPB), 0, st, out, cnt, C, tot2);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_upsample_mean_bwd_f32(const float *grad_out, const int64_t *knn_idx, const float *cnt, int B,
                                         int S, int K, int Nf, int C, float *grad_points, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!grad_out || !knn_idx || !cnt || !grad_points || B <= 0 || S <= 0 || K <= 0 || Nf <= 0 || C <= 0)
        return MPA_EINVAL;
    long long total = (long long)B * S * C;
    if (K <= 16 && (C & 3) == 0 && ((((uintptr_t)grad_out | (uintptr_t)grad_points)) & 15) == 0) {
        hipLaunchKernelGGL(upsample_bwd_v4_kernel<float>, dim3(grid_for(total / 4)), dim3(256), 0, (hipStream_t)stream,
                           grad_out, knn_idx, cnt, S, K, Nf, C, total / 4, grad_points);
        MPA_LAUNCH_CHECK();
        return MPA_OK;
    }
    hipLaunchKernelGGL(upsample_bwd_kernel<float>, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, grad_out,
                       knn_idx, cnt, S, K, Nf, C, total, grad_points);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

// bf16 rows: the inverted-table gather only (the workspace is required: no atomic fallback on bf16)
extern "C" int mpa_upsample_mean_fwd_bf16(const mpa_bf16 *points, const int64_t *knn_idx, int B, int S, int K, int Nf,
                                          int C, mpa_bf16 *out, float *cnt, void *workspace, size_t workspace_bytes,
                                          void *stream)
{
    MPA_CLEAR_ERROR();
    if (!points || !knn_idx || !out || !cnt || B <= 0 || S <= 0 || K <= 0 || Nf <= 0 || C <= 0) return MPA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const size_t need = mpa_upsample_workspace_bytes(B, S, K, Nf);
    if (!workspace || !need || workspace_bytes < need || ((uintptr_t)workspace & 15) != 0) return MPA_EUNSUPPORTED;
    int *rowptr = reinterpret_cast<int *>(workspace);
    int *entries = reinterpret_cast<int *>((char *)workspace + (((size_t)B * (Nf + 1) * 4 + 255) & ~(size_t)255));
    launch_csr_build(knn_idx, B, Nf, S * K, rowptr, entries, st, nullptr, true);
    const bf16_t *pb = reinterpret_cast<const bf16_t *>(points);
    bf16_t *ob = reinterpret_cast<bf16_t *>(out);
    const bool v4 = (C & 3) == 0 && ((((uintptr_t)points | (uintptr_t)out)) & 7) == 0;
    const int per = v4 ? C / 4 : C;
    int lanes = 1;
    while (lanes < per && lanes < TPB) lanes <<= 1;
    const dim3 grid(mpa_ceil_div(Nf, TPB / lanes), B);
    if (v4)
        hipLaunchKernelGGL((upsample_gather_kernel<4, bf16_t>), grid, dim3(TPB), 0, st, pb, knn_idx, rowptr, entries, S, K,
                           Nf, C, lanes, ob, cnt);
    else
        hipLaunchKernelGGL((upsample_gather_kernel<1, bf16_t>), grid, dim3(TPB), 0, st, pb, knn_idx, rowptr, entries, S, K,
                           Nf, C, lanes, ob, cnt);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_upsample_mean_bwd_bf16(const mpa_bf16 *grad_out, const int64_t *knn_idx, const float *cnt, int B,
                                          int S, int K, int Nf, int C, mpa_bf16 *grad_points, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!grad_out || !knn_idx || !cnt || !grad_points || B <= 0 || S <= 0 || K <= 0 || Nf <= 0 || C <= 0)
        return MPA_EINVAL;
    long long total = (long long)B * S * C;
    if (K <= 16 && (C & 3) == 0 && ((((uintptr_t)grad_out | (uintptr_t)grad_points)) & 7) == 0) {
        hipLaunchKernelGGL(upsample_bwd_v4_kernel<bf16_t>, dim3(grid_for(total / 4)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const bf16_t *>(grad_out), knn_idx, cnt, S, K, Nf, C, total / 4,
                           reinterpret_cast<bf16_t *>(grad_points));
        MPA_LAUNCH_CHECK();
        return MPA_OK;
    }
    hipLaunchKernelGGL(upsample_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream,
                       reinterpret_cast<const bf16_t *>(grad_out), knn_idx, cnt, S, K, Nf, C, total,
                       reinterpret_cast<bf16_t *>(grad_points));
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

// ---- global max over the points of a state: out[b][c] = max_n x[b][n][c] with the first maximum's index
// (`x.max(dim=1)` of the part-seg head, reference modules/pointnet2_utils.py:846-850).  torch's own reduction is a
// multi-workgroup kernel that mis-replays under HIP-graph capture on this stack and had to be taken in two stages
// (20 us each on [32,2048,64]); here a workgroup of 64 row lanes x 16 channel lanes owns 16 channels of a cloud, every
// lane walks N/64 rows with 8 loads in flight and the 64 partial (value, row) pairs meet in LDS.  A NaN in a
// column is propagated (the first NaN row is the arg), as torch.max does: a diverged run shows up in the loss.
namespace {
constexpr int MAXP_RL = 64, MAXP_CL = 16;

template <typename T>
__global__ __launch_bounds__(MAXP_RL * MAXP_CL) void max_points_fwd_kernel(const T *__restrict__ x, int N, int C,
                                                                           T *__restrict__ out, int *__restrict__ arg)
{
    __shared__ float sv[MAXP_RL][MAXP_CL];
    __shared__ int si[MAXP_RL][MAXP_CL];
    const int cl = threadIdx.x & (MAXP_CL - 1), rl = threadIdx.x / MAXP_CL;
    const int b = blockIdx.y, c = blockIdx.x * MAXP_CL + cl;
    float best = -INFINITY;
    int bi = 0;
    if (c < C) {
        const T *p = x + (size_t)b * N * C + c;
        int n = rl;
        for (; n + 7 * MAXP_RL < N; n += 8 * MAXP_RL) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = mpa_ld1<T>(p + (size_t)(n + u * MAXP_RL) * C);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (v[u] > best || (v[u] != v[u] && best == best)) { best = v[u]; bi = n + u * MAXP_RL; }
        }
        for (; n < N; n += MAXP_RL) {
            const float v = mpa_ld1<T>(p + (size_t)n * C);
            if (v > best || (v != v && best == best)) { best = v; bi = n; }
        }
    }
    sv[rl][cl] = best;
    si[rl][cl] = bi;
    __syncthreads();
    if (rl == 0 && c < C) {
        for (int r = 1; r < MAXP_RL; ++r) {
            const float v = sv[r][cl];
            const int i2 = si[r][cl];
            // a NaN is sticky (torch.max propagates it, reference :846-850); among equals / NaNs the lowest row wins
            const bool vn = v != v, bn = best != best;
            if ((vn && (!bn || i2 < bi)) || (!vn && !bn && (v > best || (v == best && i2 < bi)))) { best = v; bi = i2; }
        }
        mpa_st1<T>(out + (size_t)b * C + c, best);
        arg[(size_t)b * C + c] = bi;
    }
}

// grad_x[b][n][c] = (n == arg[b][c]) ? grad_out[b][c] : 0 -- the zero fill and the scatter in one pass.
template <typename T>
__global__ __launch_bounds__(256) void max_points_bwd_kernel(const T *__restrict__ g, const int *__restrict__ arg, int N,
                                                             int C, long long total, T *__restrict__ gx)
{
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long bn = i / C;
        const int n = (int)(bn % N), b = (int)(bn / N);
        const int a = arg[(size_t)b * C + c];
        mpa_st1<T>(gx + i, a == n ? mpa_ld1<T>(g + (size_t)b * C + c) : 0.f);
    }
}

template <typename T>
int max_points_fwd_any(const T *x, int B, int N, int C, T *out, int *arg, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!x || !out || !arg || B <= 0 || N <= 0 || C <= 0 || B > 65535) return MPA_EINVAL;
    hipLaunchKernelGGL(max_points_fwd_kernel<T>, dim3(mpa_ceil_div(C, MAXP_CL), B), dim3(MAXP_RL * MAXP_CL), 0,
                       (hipStream_t)stream, x, N, C, out, arg);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

template <typename T>
int max_points_bwd_any(const T *g, const int *arg, int B, int N, int C, T *gx, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!g || !arg || !gx || B <= 0 || N <= 0 || C <= 0) return MPA_EINVAL;
    const long long total = (long long)B * N * C;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(max_points_bwd_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, arg, N, C, total,
                       gx);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}
}  // namespace

extern "C" int mpa_max_points_fwd_f32(const float *x, int B, int N, int C, float *out, int *arg, void *stream)
{
    return max_points_fwd_any<float>(x, B, N, C, out, arg, stream);
}
extern "C" int mpa_max_points_fwd_bf16(const mpa_bf16 *x, int B, int N, int C, mpa_bf16 *out, int *arg, void *stream)
{
    return max_points_fwd_any<bf16_t>(reinterpret_cast<const bf16_t *>(x), B, N, C, reinterpret_cast<bf16_t *>(out), arg,
                                      stream);
}
extern "C" int mpa_max_points_bwd_f32(const float *grad_out, const int *arg, int B, int N, int C, float *grad_x,
                                      void *stream)
{
    return max_points_bwd_any<float>(grad_out, arg, B, N, C, grad_x, stream);
}
extern "C" int mpa_max_points_bwd_bf16(const mpa_bf16 *grad_out, const int *arg, int B, int N, int C, mpa_bf16 *grad_x,
                                       void *stream)
{
    return max_points_bwd_any<bf16_t>(reinterpret_cast<const bf16_t *>(grad_out), arg, B, N, C,
                                      reinterpret_cast<bf16_t *>(grad_x), stream);
}

namespace {
template <typename T>
int three_interp_fwd_any(const T *points2, const int64_t *idx, const float *dist, int B, int Nq, int Nb, int C, T *out,
                         void *stream)
{
    MPA_CLEAR_ERROR();
    if (!points2 || !idx || !dist || !out || B <= 0 || Nq <= 0 || Nb <= 0 || C <= 0) return MPA_EINVAL;
    long long total = (long long)B * Nq * C;
    hipLaunchKernelGGL(interp_fwd_kernel<T>, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, points2, idx, dist,
                       Nq, Nb, C, total, out);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

template <typename T>
int three_interp_bwd_any(const T *grad_out, const int64_t *idx, const float *dist, int B, int Nq, int Nb, int C,
                         float *grad_points2, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!grad_out || !idx || !dist || !grad_points2 || B <= 0 || Nq <= 0 || Nb <= 0 || C <= 0) return MPA_EINVAL;
    long long total = (long long)B * Nq * C;
    hipLaunchKernelGGL(interp_bwd_kernel<T>, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, grad_out, idx,
                       dist, Nq, Nb, C, total, grad_points2);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}
}  // namespace

extern "C" int mpa_three_interp_fwd_f32(const float *points2, const int64_t *idx, const float *dist, int B, int Nq,
                                        int Nb, int C, float *out, void *stream)
{
    return three_interp_fwd_any<float>(points2, idx, dist, B, Nq, Nb, C, out, stream);
}

extern "C" int mpa_three_interp_bwd_f32(const float *grad_out, const int64_t *idx, const float *dist, int B, int Nq,
                                        int Nb, int C, float *grad_points2, void *stream)
{
    return three_interp_bwd_any<float>(grad_out, idx, dist, B, Nq, Nb, C, grad_points2, stream);
}

extern "C" int mpa_three_interp_fwd_bf16(const mpa_bf16 *points2, const int64_t *idx, const float *dist, int B, int Nq,
                                         int Nb, int C, mpa_bf16 *out, void *stream)
{
    return three_interp_fwd_any<bf16_t>(reinterpret_cast<const bf16_t *>(points2), idx, dist, B, Nq, Nb, C,
                                        reinterpret_cast<bf16_t *>(out), stream);
}

extern "C" int mpa_three_interp_bwd_bf16(const mpa_bf16 *grad_out, const int64_t *idx, const float *dist, int B, int Nq,
                                         int Nb, int C, float *grad_points2, void *stream)
{
    return three_interp_bwd_any<bf16_t>(reinterpret_cast<const bf16_t *>(grad_out), idx, dist, B, Nq, Nb, C, grad_points2,
                                        stream);
}
