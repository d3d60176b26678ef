// kNN / ball-query / square_distance for gfx950 -- replaces square_distance + topk
// (reference modules/pointnet2_utils.py:190-222) and query_ball_point (:112-134).
//
// The [B,S,N] distance matrix is never written.  One lane owns one query: its C channels and
// its sorted K-list live in registers; base rows stream through an LDS tile and are read as
// wave-wide broadcasts (every lane reads the same row), so LDS traffic is one row per 64
// queries.  The arithmetic is the reference CPU result bit for bit:
//   dot   : acc = q0*b0; acc = fmaf(qc, bc, acc) in channel order          (A1)
//   norms : separately rounded squares; C<8 sequential, C%8==0 the 8-lane x 4-accumulator
//           order of ATen's vectorised sum                                      (A2)
//   dist  : fl(fl(fl(-2*dot) + |q|^2) + |b|^2)                                 (A3)
// Ties keep the lower base index (strict < on an ascending scan), the order a stable sort gives.
#include "mpa_common.h"

namespace {

// A2: torch.sum(x**2, -1).  CT > 0: compile-time channel count (fully unrolled).
template <int CT, typename Ptr>
__device__ __forceinline__ float sum_sq_model(Ptr x, int C)
{
    const int c_ = CT > 0 ? CT : C;
    if (c_ < 8) {
        float s = x[0] * x[0];
        for (int c = 1; c < c_; ++c) {
            float sq = x[c] * x[c];
            s = s + sq;
        }
        return s;
    }
    const int nv = c_ / 8;
    const int A = nv < 4 ? nv : 4;
    float acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            float v = a < A ? x[a * 8 + l] : 0.f;
            acc[a][l] = v * v;
        }
    if (CT > 0) {
#pragma unroll
        for (int v = 4; v < (CT > 0 ? CT / 8 : 4); ++v) {
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                float e = x[v * 8 + l];
                float sq = e * e;
                acc[v & 3][l] = acc[v & 3][l] + sq;
            }
        }
    } else {
        for (int v = A; v < nv; ++v) {
            int a = v % A;
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                float e = x[v * 8 + l];
                float sq = e * e;
                // runtime accumulator select without dynamic register indexing
#pragma unroll
                for (int aa = 0; aa < 4; ++aa)
                    if (aa == a) acc[aa][l] = acc[aa][l] + sq;
            }
        }
    }
    float lane[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        float s = acc[0][l];
        if (A > 1) s = s + acc[1][l];
        if (A > 2) s = s + acc[2][l];
        if (A > 3) s = s + acc[3][l];
        lane[l] = s;
    }
    float s = lane[0];
#pragma unroll
    for (int l = 1; l < 8; ++l) s = s + lane[l];
    for (int c = nv * 8; c < c_; ++c) {
        float sq = x[c] * x[c];
        s = s + sq;
    }
    return s;
}

// LDS tile geometry: TB base rows of C floats, row stride C rounded up to a float4 multiple
// plus one float4 of padding (keeps 16-B alignment, spreads the norm pass over banks).
__host__ __device__ inline int tile_ld(int C) { return ((C + 3) / 4) * 4 + 4; }

template <int CT>
__device__ __forceinline__ void stage_tile(const float *__restrict__ base, int n0, int rows, int C, int LD,
                                           float *tile, float *tnorm, int tid, int nthreads)
{
    const int c_ = CT > 0 ? CT : C;
    // rows n0..n0+rows-1 are contiguous in global memory
    const float *src = base + (size_t)n0 * c_;
    if ((c_ & 3) == 0) {
        const int c4 = c_ >> 2;
        for (int i = tid; i < rows * c4; i += nthreads) {
            int r = i / c4, c = i - r * c4;
            float4 v = reinterpret_cast<const float4 *>(src)[i];
            *reinterpret_cast<float4 *>(tile + r * LD + 4 * c) = v;
        }
    } else {
        for (int i = tid; i < rows * c_; i += nthreads) {
            int r = i / c_, c = i - r * c_;
            tile[r * LD + c] = src[i];
        }
    }
    __syncthreads();
    for (int r = tid; r < rows; r += nthreads) tnorm[r] = sum_sq_model<CT>(tile + r * LD, c_);
    __syncthreads();
}

template <int CT>
__device__ __forceinline__ float dot_chain(const float *q, const float *row, int C)
{
    const int c_ = CT > 0 ? CT : C;
    float acc;
    if (CT > 0 && (CT & 3) == 0) {
        float4 r0 = *reinterpret_cast<const float4 *>(row);
        acc = q[0] * r0.x;
        acc = fmaf(q[1], r0.y, acc);
        acc = fmaf(q[2], r0.z, acc);
        acc = fmaf(q[3], r0.w, acc);
#pragma unroll
        for (int c = 4; c < CT; c += 4) {
            float4 r = *reinterpret_cast<const float4 *>(row + c);
            acc = fmaf(q[c], r.x, acc);
            acc = fmaf(q[c + 1], r.y, acc);
            acc = fmaf(q[c + 2], r.z, acc);
            acc = fmaf(q[c + 3], r.w, acc);
        }
    } else {
        acc = q[0] * row[0];
#pragma unroll
        for (int c = 1; c < (CT > 0 ? CT : 1); ++c) acc = fmaf(q[c], row[c], acc);
        if (CT == 0)
            for (int c = 1; c < c_; ++c) acc = fmaf(q[c], row[c], acc);
    }
    return acc;
}

__device__ __forceinline__ float finish_dist(float dot, float qn, float bn)
{
    float d = -2.0f * dot;
    d = d + qn;
    d = d + bn;
    return d;
}

constexpr int TQ = 64;   // queries per workgroup (one wave)

// CT = compile-time C (query row in registers) ; CT == 0: runtime C, query row in LDS.
template <int CT, int KMAX>
__global__ __launch_bounds__(TQ) void knn_kernel(const float *__restrict__ base, const float *__restrict__ query,
                                                 int N, int S, int C, int K, int TB,
                                                 float *__restrict__ out_dist, int64_t *__restrict__ out_idx)
{
    extern __shared__ float lds[];
    const int c_ = CT > 0 ? CT : C;
    const int LD = tile_ld(c_);
    float *tile = lds;                 // [TB][LD]
    float *tnorm = tile + TB * LD;     // [TB]
    float *qlds = tnorm + TB;          // [TQ][LD]   (CT == 0 only)

    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int s = blockIdx.x * TQ + tid;
    const bool live = s < S;
    const float *bp = base + (size_t)b * N * c_;
    const float *qp = query + ((size_t)b * S + (live ? s : 0)) * c_;

    float qreg[CT > 0 ? CT : 1];
    const float *q;
    if (CT > 0) {
#pragma unroll
        for (int c = 0; c < CT; ++c) qreg[c] = qp[c];
        q = qreg;
    } else {
        float *mine = qlds + tid * LD;
        for (int c = 0; c < c_; ++c) mine[c] = qp[c];
        q = mine;
    }
    const float qn = sum_sq_model<CT>(q, c_);

    float bd[KMAX];
    int bi[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { bd[k] = INFINITY; bi[k] = 0; }   // NaN inputs: in-range garbage, never -1

    for (int n0 = 0; n0 < N; n0 += TB) {
        const int rows = min(TB, N - n0);
        __syncthreads();
        stage_tile<CT>(bp, n0, rows, c_, LD, tile, tnorm, tid, TQ);
        for (int t = 0; t < rows; ++t) {
            float d = finish_dist(dot_chain<CT>(q, tile + t * LD, c_), qn, tnorm[t]);
            if (d < bd[KMAX - 1]) {
                int n = n0 + t;
#pragma unroll
                for (int k = KMAX - 1; k > 0; --k) {
                    bool up = d < bd[k - 1];
                    bool here = d < bd[k];
                    bd[k] = up ? bd[k - 1] : (here ? d : bd[k]);
                    bi[k] = up ? bi[k - 1] : (here ? n : bi[k]);
                }
                bool h0 = d < bd[0];
                bd[0] = h0 ? d : bd[0];
                bi[0] = h0 ? n : bi[0];
            }
        }
    }
    if (live) {
        size_t o = ((size_t)b * S + s) * K;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) {
                out_idx[o + k] = bi[k];
                if (out_dist) out_dist[o + k] = bd[k];
            }
    }
}

template <int CT>
__global__ __launch_bounds__(TQ) void ball_kernel(const float *__restrict__ base, const float *__restrict__ query,
                                                  int N, int S, int C, float r2, int nsample, int TB,
                                                  int64_t *__restrict__ out_idx)
{
    extern __shared__ float lds[];
    const int c_ = CT > 0 ? CT : C;
    const int LD = tile_ld(c_);
    float *tile = lds;
    float *tnorm = tile + TB * LD;
    float *qlds = tnorm + TB;

    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int s = blockIdx.x * TQ + tid;
    const bool live = s < S;
    const float *bp = base + (size_t)b * N * c_;
    const float *qp = query + ((size_t)b * S + (live ? s : 0)) * c_;
    float qreg[CT > 0 ? CT : 1];
    const float *q;
    if (CT > 0) {
#pragma unroll
        for (int c = 0; c < CT; ++c) qreg[c] = qp[c];
        q = qreg;
    } else {
        float *mine = qlds + tid * LD;
        for (int c = 0; c < c_; ++c) mine[c] = qp[c];
        q = mine;
    }
    const float qn = sum_sq_model<CT>(q, c_);
    int64_t *o = out_idx + ((size_t)b * S + (live ? s : 0)) * nsample;
    int cnt = 0;
    int firsthit = N;
    for (int n0 = 0; n0 < N; n0 += TB) {
        const int rows = min(TB, N - n0);
        __syncthreads();
        stage_tile<CT>(bp, n0, rows, c_, LD, tile, tnorm, tid, TQ);
        // wave-uniform early exit once every live lane has its nsample hits
        if (__ballot(live && cnt < nsample) == 0ull) continue;
        for (int t = 0; t < rows; ++t) {
            float d = finish_dist(dot_chain<CT>(q, tile + t * LD, c_), qn, tnorm[t]);
            if (live && cnt < nsample && !(d > r2)) {
                if (cnt == 0) firsthit = n0 + t;
                o[cnt++] = n0 + t;
            }
        }
    }
    if (live)
        for (int k = cnt; k < nsample; ++k) o[k] = firsthit;
}

// square_distance (API completeness; the fused path never materialises it): one lane per
// destination point so the [S,N] rows are written coalesced; the source row is wave-uniform.
__global__ void sqdist_kernel(const float *__restrict__ src, const float *__restrict__ dst, int S, int N, int C,
                              float *__restrict__ out)
{
    const int b = blockIdx.z;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int s0 = blockIdx.y * 16;
    if (n >= N) return;
    const float *d = dst + ((size_t)b * N + n) * C;
    const float dn = sum_sq_model<0>(d, C);
    for (int s = s0; s < min(S, s0 + 16); ++s) {
        const float *q = src + ((size_t)b * S + s) * C;
        const float qn = sum_sq_model<0>(q, C);
        float acc = q[0] * d[0];
        for (int c = 1; c < C; ++c) acc = fmaf(q[c], d[c], acc);
        out[((size_t)b * S + s) * N + n] = finish_dist(acc, qn, dn);
    }
}

int pick_tb(int C, int N, bool query_in_lds)
{
    int LD = tile_ld(C);
    long long budget = 48 * 1024;                       // keep the base tile under 48 KiB ...
    if (query_in_lds) {                                 // ... unless the query rows already eat the LDS
        long long left = 160 * 1024 - (long long)TQ * LD * 4 - 1024;
        if (left < budget) budget = left;
    }
    long long tb = budget / 4 / (LD + 1);
    tb = tb > 256 ? 256 : tb;
    if (tb > N) tb = N;
    return (int)tb;
}

template <int CT, int KMAX>
int launch_knn(const float *base, const float *query, int B, int N, int S, int C, int K, float *od, int64_t *oi,
               hipStream_t st)
{
    int TB = pick_tb(C, N, CT == 0);
    if (TB < 1) return MPA_EUNSUPPORTED;
    int LD = tile_ld(C);
    size_t lds = ((size_t)TB * LD + TB + (CT == 0 ? (size_t)TQ * LD : 0)) * sizeof(float);
    if (lds > 160 * 1024) return MPA_EUNSUPPORTED;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(&knn_kernel<CT, KMAX>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return MPA_EHIP;
    dim3 grid(mpa_ceil_div(S, TQ), B);
    hipLaunchKernelGGL((knn_kernel<CT, KMAX>), grid, dim3(TQ), lds, st, base, query, N, S, C, K, TB, od, oi);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

template <int CT>
int launch_knn_k(const float *base, const float *query, int B, int N, int S, int C, int K, float *od, int64_t *oi,
                 hipStream_t st)
{
    if (K <= 8) return launch_knn<CT, 8>(base, query, B, N, S, C, K, od, oi, st);
    if (K <= 16) return launch_knn<CT, 16>(base, query, B, N, S, C, K, od, oi, st);
    return launch_knn<CT, 32>(base, query, B, N, S, C, K, od, oi, st);
}

template <int CT>
int launch_ball(const float *base, const float *query, int B, int N, int S, int C, float r2, int ns, int64_t *oi,
                hipStream_t st)
{
    int TB = pick_tb(C, N, CT == 0);
    if (TB < 1) return MPA_EUNSUPPORTED;
    int LD = tile_ld(C);
    size_t lds = ((size_t)TB * LD + TB + (CT == 0 ? (size_t)TQ * LD : 0)) * sizeof(float);
    if (lds > 160 * 1024) return MPA_EUNSUPPORTED;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(&ball_kernel<CT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return MPA_EHIP;
    dim3 grid(mpa_ceil_div(S, TQ), B);
    hipLaunchKernelGGL((ball_kernel<CT>), grid, dim3(TQ), lds, st, base, query, N, S, C, r2, ns, TB, oi);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

}  // namespace

extern "C" int mpa_knn_f32(const float *base, const float *query, int B, int N, int S, int C, int K,
                           float *out_dist, int64_t *out_idx, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!base || !query || !out_idx || B <= 0 || N <= 0 || S <= 0 || C <= 0 || K <= 0) return MPA_EINVAL;
    if (K > 32 || K > N) return MPA_EUNSUPPORTED;
    if (C >= 8 && (C & 7)) return MPA_EUNSUPPORTED;   // norm rounding model only validated for C<8 or C%8==0
    hipStream_t st = (hipStream_t)stream;
    switch (C) {
    case 3: return launch_knn_k<3>(base, query, B, N, S, C, K, out_dist, out_idx, st);
    case 64: return launch_knn_k<64>(base, query, B, N, S, C, K, out_dist, out_idx, st);
    case 128: return launch_knn_k<128>(base, query, B, N, S, C, K, out_dist, out_idx, st);
    case 256: return launch_knn_k<256>(base, query, B, N, S, C, K, out_dist, out_idx, st);
    default: return launch_knn_k<0>(base, query, B, N, S, C, K, out_dist, out_idx, st);
    }
}

extern "C" int mpa_ball_query_f32(const float *base, const float *query, int B, int N, int S, int C,
                                  float radius2, int nsample, int64_t *out_idx, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!base || !query || !out_idx || B <= 0 || N <= 0 || S <= 0 || C <= 0 || nsample <= 0) return MPA_EINVAL;
    if (C >= 8 && (C & 7)) return MPA_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (C == 3) return launch_ball<3>(base, query, B, N, S, C, radius2, nsample, out_idx, st);
    return launch_ball<0>(base, query, B, N, S, C, radius2, nsample, out_idx, st);
}

extern "C" int mpa_square_distance_f32(const float *src, const float *dst, int B, int S, int N, int C,
                                       float *out, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!src || !dst || !out || B <= 0 || S <= 0 || N <= 0 || C <= 0) return MPA_EINVAL;
    if (C >= 8 && (C & 7)) return MPA_EUNSUPPORTED;
    dim3 grid(mpa_ceil_div(N, 256), mpa_ceil_div(S, 16), B);
    hipLaunchKernelGGL(sqdist_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, S, N, C, out);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}
