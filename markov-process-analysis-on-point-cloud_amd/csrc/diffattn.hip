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
#include "mpa_bf16.h"
#include "csr_build.h"
#include <cstdlib>
#include <type_traits>

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
    // one reciprocal (v_rcp_f32, 1 ulp) and K products instead of K IEEE divisions (10 instructions
    // each: a sixth of the VALU-bound xyz kernels); the weights move by <= 2 ulp, far inside 1e-4
    const float r = __builtin_amdgcn_rcpf(sum);
    o = 0.f;
#pragma unroll
    for (int j = 0; j < (K_ > 0 ? K_ : KMAX); ++j)
        if (j < k_) {
            a[j] = a[j] * r;
            o += a[j];
        }
}

// ------------------------------------------------------------------ feature branch
template <int K_>
__global__ __launch_bounds__(TPB) void diffattn_fwd_kernel(const float *__restrict__ q, const float *__restrict__ kk,
                                                           const float *__restrict__ vv, int ldkv, int ldq,
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
        const float qv = q[p * ldq + c];
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

// float4 lanes: one lane owns 4 consecutive channels of a point, so every gathered row is read as
// 16 B per lane (C % 4 == 0, 16-B aligned rows).
template <int K_, typename T>
__global__ __launch_bounds__(TPB) void diffattn_fwd_v4_kernel(const T *__restrict__ q, const T *__restrict__ kk,
                                                              const T *__restrict__ vv, int ldkv, int ldq,
                                                              const int64_t *__restrict__ idx, int N, int S, int K,
                                                              int C, float alpha, long long total4,
                                                              T *__restrict__ ctx, uint8_t *__restrict__ argk)
{
    constexpr int KK = K_ > 0 ? K_ : KMAX;
    const int k_ = K_ > 0 ? K_ : K;
    const int c4n = C >> 2;
    for (long long i = blockIdx.x * (long long)TPB + threadIdx.x; i < total4; i += (long long)gridDim.x * TPB) {
        const long long p = i / c4n;
        const int c = (int)(i - p * c4n) << 2;
        const int b = (int)(p / S);
        const int64_t *nb = idx + p * k_;
        const float4 q4 = mpa_ld4<T>(q + p * ldq + c);
        const float qv[4] = {q4.x, q4.y, q4.z, q4.w};
        float4 k4[KK], v4[KK];
#pragma unroll
        for (int j = 0; j < KK; ++j)
            if (j < k_) {
                const long long row = ((long long)b * N + mpa_clamp_idx(nb[j], N)) * ldkv + c;
                k4[j] = mpa_ld4<T>(kk + row);
                v4[j] = mpa_ld4<T>(vv + row);
            }
        float best4[4];
        uint8_t bj4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float e[KK], a[KK];
#pragma unroll
            for (int j = 0; j < KK; ++j)
                if (j < k_) e[j] = (qv[u] - (&k4[j].x)[u]) * alpha;
            float o;
            softmax_offset<K_>(e, k_, a, o);
            float best = (a[0] - o) * (&v4[0].x)[u];
            int bj = 0;
#pragma unroll
            for (int j = 1; j < KK; ++j)
                if (j < k_) {
                    float t = (a[j] - o) * (&v4[j].x)[u];
                    if (t > best) { best = t; bj = j; }
                }
            best4[u] = best;
            bj4[u] = (uint8_t)bj;
        }
        mpa_st4<T>(ctx + p * C + c, make_float4(best4[0], best4[1], best4[2], best4[3]));
        *reinterpret_cast<uchar4 *>(argk + p * C + c) = make_uchar4(bj4[0], bj4[1], bj4[2], bj4[3]);
    }
}

template <int K_>
__global__ __launch_bounds__(TPB) void diffattn_bwd_kernel(const float *__restrict__ q, const float *__restrict__ kk,
                                                           const float *__restrict__ vv, int ldkv, int ldq,
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
        const float qv = q[p * ldq + c];
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
        gq[p * ldq + c] = alpha * dq;
        atomicAdd(gv + rstar * ldg + c, g * (astar - o));
    }
}

// Backward without float atomics, in two passes over an inverted neighbour table.
//   csr_build_kernel     per cloud: counting sort of the S*K (point, slot) entries by base row
//                        -> rowptr [B][N+1], entries [B][S*K] (entry = s*K + j)
//   diffattn_bwd_p1      lane = (point, channel) as in forward: recomputes the softmax, writes
//                        grad_q and the per-slot key gradients T [B,S,K,C] and the value
//                        gradient Tv [B,S,C] with plain coalesced stores
//   diffattn_bwd_p2      lane = (base row, 4 channels): sums T over the row's entries (and Tv
//                        where the entry's slot is the arg-max) -> grad_k, grad_v, fully written
// (K+1) float atomics per (point, channel) -- 150 M per training step of the cls model -- ran at
// the chip's atomic rate (~0.9 TB/s); an LDS-resident accumulator per (cloud, channel slice)
// was no faster: ds_add_f32 retires about one lane per 2.5 clocks.
constexpr int BWD_TPB = 1024;

template <int K_>
__global__ __launch_bounds__(TPB) void diffattn_bwd_p1_kernel(
    const float *__restrict__ q, const float *__restrict__ kk, const float *__restrict__ vv, int ldkv, int ldq,
    const int64_t *__restrict__ idx, const uint8_t *__restrict__ argk, const float *__restrict__ gctx, int N, int S,
    int K, int C, float alpha, long long total, float *__restrict__ gq, float *__restrict__ T,
    float *__restrict__ Tv)
{
    const int k_ = K_ > 0 ? K_ : K;
    for (long long i = blockIdx.x * (long long)TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        long long p = i / C;
        int c = (int)(i - p * C);
        int b = (int)(p / S);
        const int64_t *nb = idx + p * k_;
        const float qv = q[p * ldq + c];
        const int ks = argk[i];
        const float g = gctx[i];
        float e[K_ > 0 ? K_ : KMAX], a[K_ > 0 ? K_ : KMAX];
        float vstar = 0.f;
#pragma unroll
        for (int j = 0; j < (K_ > 0 ? K_ : KMAX); ++j)
            if (j < k_) {
                long long r = ((long long)b * N + mpa_clamp_idx(nb[j], N)) * ldkv + c;
                e[j] = (qv - kk[r]) * alpha;
                if (j == ks) vstar = vv[r];
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
        float *Tp = T + p * k_ * C + c;
#pragma unroll
        for (int j = 0; j < (K_ > 0 ? K_ : KMAX); ++j)
            if (j < k_) {
                float de = a[j] * h * ((j == ks ? 1.0f : 0.0f) + common);
                dq += de;
                Tp[(long long)j * C] = -alpha * de;
            }
        gq[p * ldq + c] = alpha * dq;
        Tv[i] = g * (astar - o);
    }
}

template <int K_, typename TF>
__device__ __forceinline__ void diffattn_bwd_p1_v4_body(
    const TF *__restrict__ q, const TF *__restrict__ kk, const TF *__restrict__ vv, int ldkv, int ldq,
    const int64_t *__restrict__ idx, const uint8_t *__restrict__ argk, const TF *__restrict__ gctx, int N, int S,
    int K, int C, float alpha, long long total4, TF *__restrict__ gq, TF *__restrict__ T,
    TF *__restrict__ Tv, const int blk, const int nblk)
{
    constexpr int KK = K_ > 0 ? K_ : KMAX;
    const int k_ = K_ > 0 ? K_ : K;
    const int c4n = C >> 2;
    for (long long i = blk * (long long)TPB + threadIdx.x; i < total4; i += (long long)nblk * TPB) {
        const long long p = i / c4n;
        const int c = (int)(i - p * c4n) << 2;
        const int b = (int)(p / S);
        const int64_t *nb = idx + p * k_;
        const float4 q4 = mpa_ld4<TF>(q + p * ldq + c);
        const float4 g4 = mpa_ld4<TF>(gctx + p * C + c);
        const uchar4 ks4 = *reinterpret_cast<const uchar4 *>(argk + p * C + c);
        const float qv[4] = {q4.x, q4.y, q4.z, q4.w}, gv_[4] = {g4.x, g4.y, g4.z, g4.w};
        const int ks[4] = {ks4.x, ks4.y, ks4.z, ks4.w};
        float4 k4[KK];
        long long rows[KK];
#pragma unroll
        for (int j = 0; j < KK; ++j)
            if (j < k_) {
                rows[j] = ((long long)b * N + mpa_clamp_idx(nb[j], N)) * ldkv + c;
                k4[j] = mpa_ld4<TF>(kk + rows[j]);
            }
        float vstar[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            long long r = rows[0];
#pragma unroll
            for (int j = 1; j < KK; ++j)
                if (j < k_ && j == ks[u]) r = rows[j];
            vstar[u] = mpa_ld1<TF>(vv + r + u);
        }
        float de[KK][4], dq4[4], dv4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float e[KK], a[KK];
#pragma unroll
            for (int j = 0; j < KK; ++j)
                if (j < k_) e[j] = (qv[u] - (&k4[j].x)[u]) * alpha;
            float o;
            softmax_offset<K_>(e, k_, a, o);
            float astar = 0.f;
#pragma unroll
            for (int j = 0; j < KK; ++j)
                if (j < k_ && j == ks[u]) astar = a[j];
            const float h = gv_[u] * vstar[u];
            const float common = -1.0f - astar + o;
            float dq = 0.f;
#pragma unroll
            for (int j = 0; j < KK; ++j)
                if (j < k_) {
                    float d = a[j] * h * ((j == ks[u] ? 1.0f : 0.0f) + common);
                    dq += d;
                    de[j][u] = -alpha * d;
                }
            dq4[u] = alpha * dq;
            dv4[u] = gv_[u] * (astar - o);
        }
        TF *Tp = T + p * k_ * C + c;
#pragma unroll
        for (int j = 0; j < KK; ++j)
            if (j < k_) mpa_st4<TF>(Tp + (long long)j * C, make_float4(de[j][0], de[j][1], de[j][2], de[j][3]));
        mpa_st4<TF>(gq + p * ldq + c, make_float4(dq4[0], dq4[1], dq4[2], dq4[3]));
        mpa_st4<TF>(Tv + p * C + c, make_float4(dv4[0], dv4[1], dv4[2], dv4[3]));
    }
}

template <int K_, typename TF>
__global__ __launch_bounds__(TPB) void diffattn_bwd_p1_v4_kernel(
    const TF *__restrict__ q, const TF *__restrict__ kk, const TF *__restrict__ vv, int ldkv, int ldq,
    const int64_t *__restrict__ idx, const uint8_t *__restrict__ argk, const TF *__restrict__ gctx, int N, int S,
    int K, int C, float alpha, long long total4, TF *__restrict__ gq, TF *__restrict__ T,
    TF *__restrict__ Tv)
{
    diffattn_bwd_p1_v4_body<K_, TF>(q, kk, vv, ldkv, ldq, idx, argk, gctx, N, S, K, C, alpha, total4, gq, T, Tv, blockIdx.x,
                                    gridDim.x);
}

// The first pass and the inverted-table build in ONE launch: the table depends on idx only and the first pass does
// not read it (the second pass does), so the build's (ranges x B) workgroups ride along instead of costing a launch
// of their own in front (8-14 us each, 10 per cls step / 16 per part-seg step).
struct CsrArgs {
    int N, SK, range, ranges, B;
    int *rowptr, *entries, *queue;
};
template <int K_, typename TF>
__global__ __launch_bounds__(TPB) void diffattn_bwd_p1_csr_kernel(
    const TF *__restrict__ q, const TF *__restrict__ kk, const TF *__restrict__ vv, int ldkv, int ldq,
    const int64_t *__restrict__ idx, const uint8_t *__restrict__ argk, const TF *__restrict__ gctx, int N, int S,
    int K, int C, float alpha, long long total4, TF *__restrict__ gq, TF *__restrict__ T,
    TF *__restrict__ Tv, CsrArgs ca)
{
    extern __shared__ int p1_csr_lds[];
    const int csr_blocks = ca.ranges * ca.B;
    if ((int)blockIdx.x < csr_blocks) {
        csr_build_body(idx, ca.N, ca.SK, ca.range, ca.rowptr, ca.entries, ca.queue, blockIdx.x % ca.ranges,
                       blockIdx.x / ca.ranges, p1_csr_lds);
        return;
    }
    diffattn_bwd_p1_v4_body<K_, TF>(q, kk, vv, ldkv, ldq, idx, argk, gctx, N, S, K, C, alpha, total4, gq, T, Tv,
                                    blockIdx.x - csr_blocks, gridDim.x - csr_blocks);
}

// lane = (base row, V channels); rows_per_block = 256 / lanes_per_row; blockIdx.y = cloud, so the
// cloud's bases are scalar and the per-lane offsets 32-bit (S*K*C < 2^31 checked by the host).
// Rows with more than P2_LONG entries (hubs: e.g. the kNN of many identical feature rows all
// returns the same neighbours) are left to a second phase in which the whole workgroup shares one
// row's entries and combines through LDS; without it one lane group walks thousands of entries
// while the rest of the chip waits (420 us instead of 20 us per launch in the part-seg decoder).
constexpr int P2_LONG = 48;

template <int V, typename TF>
__global__ __launch_bounds__(TPB) void diffattn_bwd_p2_kernel(
    const TF *__restrict__ T, const TF *__restrict__ Tv, const uint8_t *__restrict__ argk,
    const int *__restrict__ rowptr, const int *__restrict__ entries, int N, int S, int K, int C, int lanes_per_row,
    TF *__restrict__ gk, TF *__restrict__ gv, int ldg, int *__restrict__ queue)
{
    __shared__ int long_rows[TPB];
    __shared__ int n_long;
    __shared__ int job;
    __shared__ float red[2 * V * TPB];
    const int rl = threadIdx.x / lanes_per_row, cl = threadIdx.x % lanes_per_row;
    const int rpb = TPB / lanes_per_row;
    const int b = blockIdx.y;
    const int r = blockIdx.x * rpb + rl;
    const int *rp = rowptr + (size_t)b * (N + 1);
    const int *en = entries + (size_t)b * S * K;
    const TF *Tb = T + (size_t)b * S * K * C;
    const TF *Tvb = Tv + (size_t)b * S * C;
    const uint8_t *ab = argk + (size_t)b * S * C;
    if (threadIdx.x == 0) n_long = 0;
    __syncthreads();

    // one entry's contribution to this lane's V channels starting at c
    auto add_entry = [&](unsigned ent, int c, float (&ak)[V], float (&av)[V]) {
        const unsigned s_ = ent / (unsigned)K, j = ent - s_ * (unsigned)K;
        if constexpr (V == 4) {
            const float4 t = mpa_ld4<TF>(Tb + (ent * (unsigned)C + (unsigned)c));
            const uchar4 am = *reinterpret_cast<const uchar4 *>(ab + (s_ * (unsigned)C + (unsigned)c));
            ak[0] += t.x; ak[1] += t.y; ak[2] += t.z; ak[3] += t.w;
            if (am.x == j || am.y == j || am.z == j || am.w == j) {
                const float4 d = mpa_ld4<TF>(Tvb + (s_ * (unsigned)C + (unsigned)c));
                if (am.x == j) av[0] += d.x;
                if (am.y == j) av[1] += d.y;
                if (am.z == j) av[2] += d.z;
                if (am.w == j) av[3] += d.w;
            }
        } else {
            ak[0] += mpa_ld1<TF>(Tb + (ent * (unsigned)C + (unsigned)c));
            if (ab[s_ * (unsigned)C + (unsigned)c] == j) av[0] += mpa_ld1<TF>(Tvb + (s_ * (unsigned)C + (unsigned)c));
        }
    };
    auto store_row = [&](int row, int c, const float (&ak)[V], const float (&av)[V]) {
        TF *ok = gk + ((size_t)b * N + row) * ldg + c, *ov = gv + ((size_t)b * N + row) * ldg + c;
        if constexpr (V == 4) {
            mpa_st4<TF>(ok, make_float4(ak[0], ak[1], ak[2], ak[3]));
            mpa_st4<TF>(ov, make_float4(av[0], av[1], av[2], av[3]));
        } else {
            mpa_st1<TF>(ok, ak[0]);
            mpa_st1<TF>(ov, av[0]);
        }
    };

    if (r < N) {
        const int beg = rp[r], end = rp[r + 1];
        if (end - beg > P2_LONG) {
            if (cl == 0) long_rows[atomicAdd(&n_long, 1)] = r;
        } else {
            for (int c = cl * V; c < C; c += lanes_per_row * V) {
                float ak[V], av[V];
#pragma unroll
                for (int u = 0; u < V; ++u) ak[u] = av[u] = 0.f;
                for (int e0 = beg; e0 < end; e0 += 4) {
                    unsigned ent[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) ent[u] = (unsigned)en[min(e0 + u, end - 1)];
                    if constexpr (V == 4) {
                        float4 t[4];
                        uchar4 am[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {                  // four rows of T in flight together
                            t[u] = mpa_ld4<TF>(Tb + (ent[u] * (unsigned)C + (unsigned)c));
                            am[u] = *reinterpret_cast<const uchar4 *>(ab + ((ent[u] / (unsigned)K) * (unsigned)C + (unsigned)c));
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (e0 + u >= end) break;
                            ak[0] += t[u].x; ak[1] += t[u].y; ak[2] += t[u].z; ak[3] += t[u].w;
                            const unsigned s_ = ent[u] / (unsigned)K, j = ent[u] - s_ * (unsigned)K;
                            if (am[u].x == j || am[u].y == j || am[u].z == j || am[u].w == j) {
                                const float4 d = mpa_ld4<TF>(Tvb + (s_ * (unsigned)C + (unsigned)c));
                                if (am[u].x == j) av[0] += d.x;
                                if (am[u].y == j) av[1] += d.y;
                                if (am[u].z == j) av[2] += d.z;
                                if (am[u].w == j) av[3] += d.w;
                            }
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (e0 + u >= end) break;
                            add_entry(ent[u], c, ak, av);
                        }
                    }
                }
                store_row(r, c, ak, av);
            }
        }
    }
    __syncthreads();
    // ---- hub rows.  They tend to sit next to each other (ties resolve to the lowest indices), i.e. in ONE
    // workgroup's row range: processed there one after the other they cost hundreds of microseconds while the rest
    // of the chip idles.  So every workgroup posts its hub rows to the cloud's queue and then drains the queue
    // together with the cloud's other workgroups (each job = one row, summed by a whole workgroup).  A job is
    // never lost: whoever posts drains afterwards until the queue is empty.
    int *qb = queue + (size_t)b * CSR_QUEUE_INTS;        // [0] posted, [1] taken, [2..] rows (-1 = not yet written)
    int nl = n_long;
    if (threadIdx.x == 0) {
        int kept = 0;
        for (int li = 0; li < nl; ++li) {
            const int slot = __hip_atomic_fetch_add(qb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (slot < CSR_QUEUE_MAX)
                __hip_atomic_store(qb + 2 + slot, long_rows[li], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                long_rows[kept++] = long_rows[li];           // queue full: stays with this workgroup
        }
        n_long = kept;
    }
    __syncthreads();
    nl = n_long;
    for (int li = 0;; ++li) {
        if (threadIdx.x == 0) {
            int row = -1;
            if (li < nl) {
                row = long_rows[li];
            } else {
                for (;;) {
                    const int posted = min(__hip_atomic_load(qb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), CSR_QUEUE_MAX);
                    int taken = __hip_atomic_load(qb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (taken >= posted) break;
                    if (__hip_atomic_compare_exchange_strong(qb + 1, &taken, taken + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT)) {
                        // the slot was reserved before `posted` was read; its row is written right after the
                        // reservation: wait for it (bounded)
                        for (int spin = 0; spin < (1 << 20); ++spin) {
                            row = __hip_atomic_load(qb + 2 + taken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (row >= 0) break;
                            __builtin_amdgcn_s_sleep(1);
                        }
                        break;
                    }
                }
            }
            job = row;
        }
        __syncthreads();
        const int row = job;
        if (row < 0) break;
        const int beg = rp[row], end = rp[row + 1];
        for (int c0 = 0; c0 < C; c0 += lanes_per_row * V) {
            const int c = c0 + cl * V;
            float ak[V], av[V];
#pragma unroll
            for (int u = 0; u < V; ++u) ak[u] = av[u] = 0.f;
            if (c < C) {
                for (int e0 = beg + rl; e0 < end; e0 += 4 * rpb) {         // four entries in flight per lane
                    unsigned ent[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) ent[u] = (unsigned)en[min(e0 + u * rpb, end - 1)];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (e0 + u * rpb < end) add_entry(ent[u], c, ak, av);
                }
            }
#pragma unroll
            for (int u = 0; u < V; ++u) {
                red[(u * 2) * TPB + threadIdx.x] = ak[u];
                red[(u * 2 + 1) * TPB + threadIdx.x] = av[u];
            }
            __syncthreads();
            if (rl == 0 && c < C) {
#pragma unroll
                for (int u = 0; u < V; ++u) {
                    float sk = 0.f, sv = 0.f;
                    for (int y = 0; y < rpb; ++y) {
                        sk += red[(u * 2) * TPB + y * lanes_per_row + cl];
                        sv += red[(u * 2 + 1) * TPB + y * lanes_per_row + cl];
                    }
                    ak[u] = sk;
                    av[u] = sv;
                }
                store_row(row, c, ak, av);
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------ xyz branch
// One lane = one output channel with its 12 projection weights in registers; the workgroup
// walks points, whose centre / neighbour coordinates are wave-uniform.
template <int K_, typename TF>
__global__ __launch_bounds__(TPB) void diffattn_xyz_fwd_kernel(
    const float *__restrict__ xyz, const float *__restrict__ center, const int64_t *__restrict__ idx,
    const float *__restrict__ Wq, const float *__restrict__ bq, const float *__restrict__ Wk,
    const float *__restrict__ bk, const float *__restrict__ Wv, const float *__restrict__ bv, int N, int S, int K,
    int C, int bx, float alpha, long long npoints, TF *__restrict__ ctx, uint8_t *__restrict__ argk)
{
    const int k_ = K_ > 0 ? K_ : K;
    // bx channel lanes (a multiple of 64: a wave shares its point) x PW point lanes per workgroup, as in the backward kernel
    const int pw = __builtin_amdgcn_readfirstlane(threadIdx.x / bx), PW = blockDim.x / bx;
    const int c = blockIdx.y * bx + threadIdx.x % bx;
    const bool live = c < C;                 // (every lane stays: lanes 0..3*KK+2 of each wave stage the point's operands)
    const int cc = live ? c : C - 1;
    const float wq0 = Wq[cc * 3], wq1 = Wq[cc * 3 + 1], wq2 = Wq[cc * 3 + 2], bqc = bq[cc];
    const float wk0 = Wk[cc * 3], wk1 = Wk[cc * 3 + 1], wk2 = Wk[cc * 3 + 2], bkc = bk[cc];
    const float wv0 = Wv[cc * 3], wv1 = Wv[cc * 3 + 1], wv2 = Wv[cc * 3 + 2], bvc = bv[cc];
    // centre + neighbour coordinates of a point: staged by one vector load per hop (lane 3j+c = component c of neighbour
    // j, lanes 3*KK..3*KK+2 = centre), one point ahead, and handed to the wave with v_readlane -- see the backward kernel
    constexpr int KK = K_ > 0 ? K_ : KMAX;
    const int lane = threadIdx.x & 63;
    const int sj = lane / 3, sc = lane - 3 * sj;
    const bool nb_lane = lane < 3 * k_, c_lane = lane >= 3 * KK && lane < 3 * KK + 3;
    auto hop1 = [&](long long pp) -> long long { return nb_lane ? (long long)idx[pp * k_ + sj] : 0ll; };
    auto hop2 = [&](long long pp, int bb, long long vi) -> float {
        const float *src = nb_lane ? xyz + ((long long)bb * N + mpa_clamp_idx(vi, N)) * 3 + sc
                                   : center + pp * 3 + (c_lane ? lane - 3 * KK : 0);
        return (nb_lane || c_lane) ? *src : 0.f;
    };
    const long long stride = (long long)gridDim.x * PW;
    long long p = (long long)blockIdx.x * PW + pw;
    int b = (int)(p / S), s = (int)(p - (long long)b * S);
    const int db = (int)(stride / S), ds = (int)(stride - (long long)db * S);
    float vx = 0.f;
    if (p < npoints) vx = hop2(p, b, hop1(p));
    while (p < npoints) {
        const long long pn = p + stride;
        int bn = b + db, sn = s + ds;
        if (sn >= S) { sn -= S; ++bn; }
        const bool more = pn < npoints;
        const long long pq = more ? pn : p;
        const int bq = more ? bn : b;
        const int vxi = __float_as_int(vx);
        const float cx = __int_as_float(__builtin_amdgcn_readlane(vxi, 3 * KK));
        const float cy = __int_as_float(__builtin_amdgcn_readlane(vxi, 3 * KK + 1));
        const float cz = __int_as_float(__builtin_amdgcn_readlane(vxi, 3 * KK + 2));
        float rx[KK], ry[KK], rz[KK];
#pragma unroll
        for (int j = 0; j < KK; ++j) {
            rx[j] = __int_as_float(__builtin_amdgcn_readlane(vxi, 3 * j)) - cx;
            ry[j] = __int_as_float(__builtin_amdgcn_readlane(vxi, 3 * j + 1)) - cy;
            rz[j] = __int_as_float(__builtin_amdgcn_readlane(vxi, 3 * j + 2)) - cz;
        }
        const long long vi2 = hop1(pq);
        __builtin_amdgcn_sched_barrier(0);
        const float qv = fmaf(wq2, cz, fmaf(wq1, cy, fmaf(wq0, cx, bqc)));
        float e[KK], v[KK], a[KK];
#pragma unroll
        for (int j = 0; j < KK; ++j)
            if (j < k_) {
                float kj = fmaf(wk2, rz[j], fmaf(wk1, ry[j], fmaf(wk0, rx[j], bkc)));
                v[j] = fmaf(wv2, rz[j], fmaf(wv1, ry[j], fmaf(wv0, rx[j], bvc)));
                e[j] = (qv - kj) * alpha;
            }
        __builtin_amdgcn_sched_barrier(0);
        vx = hop2(pq, bq, vi2);
        __builtin_amdgcn_sched_barrier(0);
        float o;
        softmax_offset<K_>(e, k_, a, o);
        float best = (a[0] - o) * v[0];
        int bj = 0;
#pragma unroll
        for (int j = 1; j < KK; ++j)
            if (j < k_) {
                float t = (a[j] - o) * v[j];
                if (t > best) { best = t; bj = j; }
            }
        if (live) {
            mpa_st1<TF>(ctx + p * C + c, best);
            argk[p * C + c] = (uint8_t)bj;
        }
        p = pn; b = bn; s = sn;
    }
}

// Workgroup = PW point-lanes x bx channel-lanes (1024 threads): each point-lane walks points, a
// lane keeps its channel's 18 partial gradients in registers; they meet in LDS (ds_add_f32) and
// leave as one global atomic per (workgroup, value).
template <int K_, typename TF>
__global__ __launch_bounds__(BWD_TPB) void diffattn_xyz_bwd_kernel(
    const float *__restrict__ xyz, const float *__restrict__ center, const int64_t *__restrict__ idx,
    const float *__restrict__ Wq, const float *__restrict__ bq, const float *__restrict__ Wk,
    const float *__restrict__ bk, const float *__restrict__ Wv, const float *__restrict__ bv,
    const uint8_t *__restrict__ argk, const TF *__restrict__ gctx, int N, int S, int K, int C, int bx,
    float alpha, long long npoints, float *__restrict__ gWq, float *__restrict__ gbq, float *__restrict__ gWk,
    float *__restrict__ gbk, float *__restrict__ gWv, float *__restrict__ gbv)
{
    __shared__ float red[12 * BWD_TPB];              // [PW][12][bx]: one row of partial sums per point lane
    const int k_ = K_ > 0 ? K_ : K;
    // bx is a multiple of 64, so a wave shares its point lane: telling the compiler (readfirstlane) turns the
    // point's centre / neighbour-index / coordinate loads (35 per point) into scalar loads
    const int cl = threadIdx.x % bx, pw = __builtin_amdgcn_readfirstlane(threadIdx.x / bx), PW = blockDim.x / bx;
    const int c = blockIdx.y * bx + cl;
    const bool live = c < C;
    const int cc = live ? c : C - 1;
    const float wq0 = Wq[cc * 3], wq1 = Wq[cc * 3 + 1], wq2 = Wq[cc * 3 + 2], bqc = bq[cc];
    const float wk0 = Wk[cc * 3], wk1 = Wk[cc * 3 + 1], wk2 = Wk[cc * 3 + 2], bkc = bk[cc];
    const float wv0 = Wv[cc * 3], wv1 = Wv[cc * 3 + 1], wv2 = Wv[cc * 3 + 2], bvc = bv[cc];
    float aq[4] = {0, 0, 0, 0}, ak[4] = {0, 0, 0, 0}, av[4] = {0, 0, 0, 0};   // (dW[0..2], db)
    // The point's centre, neighbour indices and neighbour coordinates are wave-uniform and a two-hop dependent chain
    // (indices, then rows).  As scalar loads they were ~20 scalar-cache misses per point and wave, and that miss path,
    // not the VALU, bounded the kernel (2.5x its instruction count).  Now ONE vector load per hop stages them: lane
    // 3j+c holds component c of neighbour j, lanes 3*KK..3*KK+2 the centre; v_readlane hands them to the whole wave.
    // Both hops run ONE POINT AHEAD (indices before the current point's softmax, rows after it).  The cloud of a
    // point (p / S, a 64-bit scalar division: ~150 SALU instructions) is carried along incrementally.
    constexpr int KK = K_ > 0 ? K_ : KMAX;
    const long long stride = (long long)gridDim.x * PW;
    long long p = (long long)blockIdx.x * PW + pw;
    int b = (int)(p / S), s = (int)(p - (long long)b * S);
    const int db = (int)(stride / S), ds = (int)(stride - (long long)db * S);
    const int lane = threadIdx.x & 63;
    const int sj = lane / 3, sc = lane - 3 * sj;
    const bool nb_lane = lane < 3 * k_, c_lane = lane >= 3 * KK && lane < 3 * KK + 3;
    auto hop1 = [&](long long pp) -> long long { return nb_lane ? (long long)idx[pp * k_ + sj] : 0ll; };
    auto hop2 = [&](long long pp, int bb, long long vi) -> float {
        const float *src = nb_lane ? xyz + ((long long)bb * N + mpa_clamp_idx(vi, N)) * 3 + sc
                                   : center + pp * 3 + (c_lane ? lane - 3 * KK : 0);
        return (nb_lane || c_lane) ? *src : 0.f;
    };
    float vx = 0.f;
    if (p < npoints) vx = hop2(p, b, hop1(p));
    int ks_n = 0;
    float g_n = 0.f;
    if (p < npoints) {
        ks_n = argk[p * C + cc];
        g_n = mpa_ld1<TF>(gctx + p * C + cc);
    }
    while (p < npoints) {
        const long long pn = p + stride;
        int bn = b + db, sn = s + ds;
        if (sn >= S) { sn -= S; ++bn; }
        const bool more = pn < npoints;
        const long long pq = more ? pn : p;
        const int bq = more ? bn : b;
        const int vxi = __float_as_int(vx);
        const float cx = __int_as_float(__builtin_amdgcn_readlane(vxi, 3 * KK));
        const float cy = __int_as_float(__builtin_amdgcn_readlane(vxi, 3 * KK + 1));
        const float cz = __int_as_float(__builtin_amdgcn_readlane(vxi, 3 * KK + 2));
        float rx[KK], ry[KK], rz[KK];
#pragma unroll
        for (int j = 0; j < KK; ++j) {
            rx[j] = __int_as_float(__builtin_amdgcn_readlane(vxi, 3 * j)) - cx;
            ry[j] = __int_as_float(__builtin_amdgcn_readlane(vxi, 3 * j + 1)) - cy;
            rz[j] = __int_as_float(__builtin_amdgcn_readlane(vxi, 3 * j + 2)) - cz;
        }
        const int ks = ks_n;
        const float g = g_n;
        ks_n = argk[pq * C + cc];                   // (the lane's own two loads, also one point ahead)
        g_n = mpa_ld1<TF>(gctx + pq * C + cc);
        const long long vi2 = hop1(pq);             // first hop for the next point
        __builtin_amdgcn_sched_barrier(0);
        const float qv = fmaf(wq2, cz, fmaf(wq1, cy, fmaf(wq0, cx, bqc)));
        float e[KK], a[KK];
        float vstar = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
        for (int j = 0; j < KK; ++j)
            if (j < k_) {
                float kj = fmaf(wk2, rz[j], fmaf(wk1, ry[j], fmaf(wk0, rx[j], bkc)));
                e[j] = (qv - kj) * alpha;
                if (j == ks) {
                    vstar = fmaf(wv2, rz[j], fmaf(wv1, ry[j], fmaf(wv0, rx[j], bvc)));
                    sx = rx[j]; sy = ry[j]; sz = rz[j];
                }
            }
        float o;
        softmax_offset<K_>(e, k_, a, o);
        __builtin_amdgcn_sched_barrier(0);
        vx = hop2(pq, bq, vi2);                     // second hop for the next point
        __builtin_amdgcn_sched_barrier(0);
        float astar = 0.f;
#pragma unroll
        for (int j = 0; j < KK; ++j)
            if (j < k_ && j == ks) astar = a[j];
        const float h = g * vstar;
        const float common = -1.0f - astar + o;
        float dq = 0.f;
#pragma unroll
        for (int j = 0; j < KK; ++j)
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
        p = pn; b = bn; s = sn;
    }
    // block reduction over the PW point lanes through LDS with plain stores (LDS float atomics retire
    // ~1 lane per 2.5 clocks: 12 of them per thread were a quarter of this kernel), then one global
    // atomic per (parameter, channel) and block
    float *mine = red + (size_t)pw * 12 * bx + cl;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        mine[d * bx] = aq[d];
        mine[(4 + d) * bx] = ak[d];
        mine[(8 + d) * bx] = av[d];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 12 * bx; t += blockDim.x) {
        const int d = t / bx, l = t - d * bx;
        const int ch = blockIdx.y * bx + l;
        if (ch >= C) continue;
        float sum = 0.f;
        for (int w = 0; w < PW; ++w) sum += red[(size_t)w * 12 * bx + t];
        const int which = d >> 2, comp = d & 3;                 // (q, k, v) x (dW[0..2], db)
        float *gw = which == 0 ? gWq : (which == 1 ? gWk : gWv);
        float *gb = which == 0 ? gbq : (which == 1 ? gbk : gbv);
        if (comp < 3) atomicAdd(gw + ch * 3 + comp, sum);
        else atomicAdd(gb + ch, sum);
    }
}

inline int grid_for(long long total)
{
    long long g = (total + TPB - 1) / TPB;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

// rows x C floats with leading dimension ld, two tensors at once.  A plain kernel instead of hipMemset2DAsync:
// inside a captured HIP graph the clear is an ordinary kernel node (no memset nodes anywhere on the path).
__global__ void clear2_strided_kernel(float *__restrict__ a, float *__restrict__ b, long long rows, int C, int ld)
{
    const long long total = rows * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C;
        const int c = (int)(i - r * C);
        a[r * ld + c] = 0.f;
        b[r * ld + c] = 0.f;
    }
}

}  // namespace

template <typename TF>
static int diffattn_fwd_any(const TF *q, int ldq, const TF *k, const TF *v, int ldkv, const int64_t *idx, int B, int N,
                            int S, int K, int C, TF *ctx, uint8_t *argk, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!q || !k || !v || !idx || !ctx || !argk || B <= 0 || N <= 0 || S <= 0 || K <= 0 || C <= 0 || ldkv < C ||
        ldq < C)
        return MPA_EINVAL;
    if (K > KMAX) return MPA_EUNSUPPORTED;
    constexpr bool F32 = std::is_same<TF, float>::value;
    long long total = (long long)B * S * C;
    float alpha = 1.0f / sqrtf((float)C);
    hipStream_t st = (hipStream_t)stream;
    static const bool scalar_only = getenv("MPA_DIFFATTN_SCALAR") != nullptr;
    const bool vec4 = (!scalar_only || !F32) && (C & 3) == 0 && (ldkv & 3) == 0 && (ldq & 3) == 0 &&
                      ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)ctx) & mpa_vec4_align<TF>::mask) == 0) &&
                      (((uintptr_t)argk & 3) == 0);
    if (vec4 && K == 8)
        hipLaunchKernelGGL((diffattn_fwd_v4_kernel<8, TF>), dim3(grid_for(total / 4)), dim3(TPB), 0, st, q, k, v, ldkv, ldq,
                           idx, N, S, K, C, alpha, total / 4, ctx, argk);
    else if (vec4)
        hipLaunchKernelGGL((diffattn_fwd_v4_kernel<0, TF>), dim3(grid_for(total / 4)), dim3(TPB), 0, st, q, k, v, ldkv, ldq,
                           idx, N, S, K, C, alpha, total / 4, ctx, argk);
    else if constexpr (F32) {
        if (K == 8)
            hipLaunchKernelGGL(diffattn_fwd_kernel<8>, dim3(grid_for(total)), dim3(TPB), 0, st, q, k, v, ldkv, ldq, idx, N,
                               S, K, C, alpha, total, ctx, argk);
        else
            hipLaunchKernelGGL(diffattn_fwd_kernel<0>, dim3(grid_for(total)), dim3(TPB), 0, st, q, k, v, ldkv, ldq, idx, N,
                               S, K, C, alpha, total, ctx, argk);
    } else {
        return MPA_EUNSUPPORTED;            // bf16 rows: 4-channel lanes only (C % 4 == 0, 8-byte aligned rows)
    }
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_diffattn_fwd_f32(const float *q, int ldq, const float *k, const float *v, int ldkv,
                                    const int64_t *idx, int B, int N, int S, int K, int C, float *ctx, uint8_t *argk,
                                    void *stream)
{
    return diffattn_fwd_any<float>(q, ldq, k, v, ldkv, idx, B, N, S, K, C, ctx, argk, stream);
}

extern "C" int mpa_diffattn_fwd_bf16(const mpa_bf16 *q, int ldq, const mpa_bf16 *k, const mpa_bf16 *v, int ldkv,
                                     const int64_t *idx, int B, int N, int S, int K, int C, mpa_bf16 *ctx,
                                     uint8_t *argk, void *stream)
{
    return diffattn_fwd_any<bf16_t>(reinterpret_cast<const bf16_t *>(q), ldq, reinterpret_cast<const bf16_t *>(k),
                                    reinterpret_cast<const bf16_t *>(v), ldkv, idx, B, N, S, K, C,
                                    reinterpret_cast<bf16_t *>(ctx), argk, stream);
}

namespace {
struct BwdWorkspace {
    size_t t_off, tv_off, rowptr_off, entries_off, queue_off, total;
};
inline BwdWorkspace bwd_workspace(int B, int N, int S, int K, int C, size_t esz)
{
    BwdWorkspace w;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    w.t_off = 0;
    w.tv_off = up((size_t)B * S * K * C * esz);
    w.rowptr_off = w.tv_off + up((size_t)B * S * C * esz);
    w.entries_off = w.rowptr_off + up((size_t)B * (N + 1) * 4);
    w.queue_off = w.entries_off + up((size_t)B * S * K * 4);
    w.total = w.queue_off + up((size_t)B * CSR_QUEUE_INTS * 4);
    return w;
}
inline bool bwd_workspace_ok(int B, int N, int S, int K, int C)
{
    return !(B <= 0 || N <= 0 || S <= 0 || K <= 0 || C <= 0 || N > CSR_MAX_N || (long long)S * K * C > 0x7fffffffLL ||
             B > 65535);
}
}  // namespace

extern "C" size_t mpa_diffattn_bwd_workspace_bytes(int B, int N, int S, int K, int C)
{
    return bwd_workspace_ok(B, N, S, K, C) ? bwd_workspace(B, N, S, K, C, 4).total : 0;
}

extern "C" size_t mpa_diffattn_bwd_workspace_bytes_bf16(int B, int N, int S, int K, int C)
{
    return bwd_workspace_ok(B, N, S, K, C) ? bwd_workspace(B, N, S, K, C, 2).total : 0;
}

template <typename TF>
static int diffattn_bwd_any(const TF *q, int ldq, const TF *k, const TF *v, int ldkv, const int64_t *idx,
                            const uint8_t *argk, const TF *grad_ctx, int B, int N, int S, int K, int C, TF *grad_q,
                            TF *grad_k, TF *grad_v, int ldg, void *workspace, size_t workspace_bytes, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!q || !k || !v || !idx || !argk || !grad_ctx || !grad_q || !grad_k || !grad_v || B <= 0 || N <= 0 || S <= 0 ||
        K <= 0 || C <= 0 || ldkv < C || ldg < C || ldq < C)
        return MPA_EINVAL;
    if (K > KMAX) return MPA_EUNSUPPORTED;
    constexpr bool F32 = std::is_same<TF, float>::value;
    long long total = (long long)B * S * C;
    float alpha = 1.0f / sqrtf((float)C);
    hipStream_t st = (hipStream_t)stream;
    const size_t need = bwd_workspace_ok(B, N, S, K, C) ? bwd_workspace(B, N, S, K, C, sizeof(TF)).total : 0;
    static const bool force_atomic = getenv("MPA_DIFFATTN_ATOMIC") != nullptr;
    if (workspace && need && workspace_bytes >= need && ((uintptr_t)workspace & 15) == 0 && !(force_atomic && F32)) {
        const BwdWorkspace w = bwd_workspace(B, N, S, K, C, sizeof(TF));
        TF *T = reinterpret_cast<TF *>((char *)workspace + w.t_off);
        TF *Tv = reinterpret_cast<TF *>((char *)workspace + w.tv_off);
        int *rowptr = reinterpret_cast<int *>((char *)workspace + w.rowptr_off);
        int *entries = reinterpret_cast<int *>((char *)workspace + w.entries_off);
        int *queue = reinterpret_cast<int *>((char *)workspace + w.queue_off);
        static const bool scalar_only = getenv("MPA_DIFFATTN_SCALAR") != nullptr;
        static const bool no_fuse = getenv("MPA_DIFFATTN_NO_CSR_FUSION") != nullptr;
        const bool p1v4 = (!scalar_only || !F32) && (C & 3) == 0 && (ldkv & 3) == 0 && (ldq & 3) == 0 &&
                          ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)grad_ctx | (uintptr_t)grad_q) &
                            mpa_vec4_align<TF>::mask) == 0) && (((uintptr_t)argk & 3) == 0);
        const bool fused = p1v4 && K == 8 && !no_fuse;
        if (!fused) launch_csr_build(idx, B, N, S * K, rowptr, entries, st, queue);
        if (fused) {
            CsrArgs ca;
            ca.N = N; ca.SK = S * K; ca.range = csr_range(B, N); ca.ranges = mpa_ceil_div(N, ca.range); ca.B = B;
            ca.rowptr = rowptr; ca.entries = entries; ca.queue = queue;
            const int csr_blocks = ca.ranges * B;
            hipLaunchKernelGGL((diffattn_bwd_p1_csr_kernel<8, TF>), dim3(csr_blocks + grid_for(total / 4)), dim3(TPB),
                               (size_t)2 * ca.range * sizeof(int), st, q, k, v, ldkv, ldq, idx, argk, grad_ctx, N, S, K, C,
                               alpha, total / 4, grad_q, T, Tv, ca);
        } else if (p1v4 && K == 8)
            hipLaunchKernelGGL((diffattn_bwd_p1_v4_kernel<8, TF>), dim3(grid_for(total / 4)), dim3(TPB), 0, st, q, k, v, ldkv,
                               ldq, idx, argk, grad_ctx, N, S, K, C, alpha, total / 4, grad_q, T, Tv);
        else if (p1v4)
            hipLaunchKernelGGL((diffattn_bwd_p1_v4_kernel<0, TF>), dim3(grid_for(total / 4)), dim3(TPB), 0, st, q, k, v, ldkv,
                               ldq, idx, argk, grad_ctx, N, S, K, C, alpha, total / 4, grad_q, T, Tv);
        else if constexpr (F32) {
            if (K == 8)
                hipLaunchKernelGGL(diffattn_bwd_p1_kernel<8>, dim3(grid_for(total)), dim3(TPB), 0, st, q, k, v, ldkv, ldq,
                                   idx, argk, grad_ctx, N, S, K, C, alpha, total, grad_q, T, Tv);
            else
                hipLaunchKernelGGL(diffattn_bwd_p1_kernel<0>, dim3(grid_for(total)), dim3(TPB), 0, st, q, k, v, ldkv, ldq,
                                   idx, argk, grad_ctx, N, S, K, C, alpha, total, grad_q, T, Tv);
        } else {
            return MPA_EUNSUPPORTED;
        }
        const bool v4 = (C & 3) == 0 && (ldg & 3) == 0 &&
                        ((((uintptr_t)grad_k | (uintptr_t)grad_v) & mpa_vec4_align<TF>::mask) == 0);
        const int per = v4 ? C / 4 : C;
        int lanes = 1;
        while (lanes < per && lanes < TPB) lanes <<= 1;                     // power of two: divides 256
        const dim3 grid2(mpa_ceil_div(N, TPB / lanes), B);
        if (v4)
            hipLaunchKernelGGL((diffattn_bwd_p2_kernel<4, TF>), grid2, dim3(TPB), 0, st, T, Tv, argk, rowptr, entries, N, S,
                               K, C, lanes, grad_k, grad_v, ldg, queue);
        else
            hipLaunchKernelGGL((diffattn_bwd_p2_kernel<1, TF>), grid2, dim3(TPB), 0, st, T, Tv, argk, rowptr, entries, N, S,
                               K, C, lanes, grad_k, grad_v, ldg, queue);
        MPA_LAUNCH_CHECK();
        return MPA_OK;
    }
    if constexpr (!F32) {
        return MPA_EUNSUPPORTED;            // bf16 gradients are never accumulated with atomics: the workspace is required
    } else {
        // global-atomic kernel: clears the outputs first (grad_k | grad_v rows of C floats, stride ldg)
        hipLaunchKernelGGL(clear2_strided_kernel, dim3(grid_for((long long)B * N * C)), dim3(TPB), 0, st, grad_k, grad_v,
                           (long long)B * N, C, ldg);
        if (K == 8)
            hipLaunchKernelGGL(diffattn_bwd_kernel<8>, dim3(grid_for(total)), dim3(TPB), 0, st, q, k, v, ldkv, ldq, idx, argk,
                               grad_ctx, N, S, K, C, alpha, total, grad_q, grad_k, grad_v, ldg);
        else
            hipLaunchKernelGGL(diffattn_bwd_kernel<0>, dim3(grid_for(total)), dim3(TPB), 0, st, q, k, v, ldkv, ldq, idx, argk,
                               grad_ctx, N, S, K, C, alpha, total, grad_q, grad_k, grad_v, ldg);
        MPA_LAUNCH_CHECK();
        return MPA_OK;
    }
}

extern "C" int mpa_diffattn_bwd_f32(const float *q, int ldq, const float *k, const float *v, int ldkv,
                                    const int64_t *idx, const uint8_t *argk, const float *grad_ctx, int B, int N, int S,
                                    int K, int C, float *grad_q, float *grad_k, float *grad_v, int ldg, void *workspace,
                                    size_t workspace_bytes, void *stream)
{
    return diffattn_bwd_any<float>(q, ldq, k, v, ldkv, idx, argk, grad_ctx, B, N, S, K, C, grad_q, grad_k, grad_v, ldg,
                                   workspace, workspace_bytes, stream);
}

extern "C" int mpa_diffattn_bwd_bf16(const mpa_bf16 *q, int ldq, const mpa_bf16 *k, const mpa_bf16 *v, int ldkv,
                                     const int64_t *idx, const uint8_t *argk, const mpa_bf16 *grad_ctx, int B, int N,
                                     int S, int K, int C, mpa_bf16 *grad_q, mpa_bf16 *grad_k, mpa_bf16 *grad_v, int ldg,
                                     void *workspace, size_t workspace_bytes, void *stream)
{
    return diffattn_bwd_any<bf16_t>(reinterpret_cast<const bf16_t *>(q), ldq, reinterpret_cast<const bf16_t *>(k),
                                    reinterpret_cast<const bf16_t *>(v), ldkv, idx, argk,
                                    reinterpret_cast<const bf16_t *>(grad_ctx), B, N, S, K, C,
                                    reinterpret_cast<bf16_t *>(grad_q), reinterpret_cast<bf16_t *>(grad_k),
                                    reinterpret_cast<bf16_t *>(grad_v), ldg, workspace, workspace_bytes, stream);
}

template <typename TF>
static int diffattn_xyz_fwd_any(const float *xyz, const float *center, const int64_t *idx, const float *Wq,
                                const float *bq, const float *Wk, const float *bk, const float *Wv, const float *bv,
                                int B, int N, int S, int K, int C, TF *ctx, uint8_t *argk, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!xyz || !center || !idx || !Wq || !bq || !Wk || !bk || !Wv || !bv || !ctx || !argk || B <= 0 || N <= 0 ||
        S <= 0 || K <= 0 || C <= 0)
        return MPA_EINVAL;
    if (K > KMAX) return MPA_EUNSUPPORTED;
    long long np = (long long)B * S;
    const int bx = C > 128 ? 256 : (C > 64 ? 128 : 64);      // channel lanes: a divisor of TPB
    const int pw = TPB / bx;
    static const int cap = getenv("MPA_XYZ_FWD_BLOCKS") ? atoi(getenv("MPA_XYZ_FWD_BLOCKS")) : 2048;
    const long long gx = (np + pw - 1) / pw;
    dim3 grid((unsigned)(gx > cap ? cap : gx), mpa_ceil_div(C, bx));
    float alpha = 1.0f / sqrtf((float)C);
    hipStream_t st = (hipStream_t)stream;
    if (K == 8)
        hipLaunchKernelGGL((diffattn_xyz_fwd_kernel<8, TF>), grid, dim3(TPB), 0, st, xyz, center, idx, Wq, bq, Wk, bk, Wv, bv,
                           N, S, K, C, bx, alpha, np, ctx, argk);
    else
        hipLaunchKernelGGL((diffattn_xyz_fwd_kernel<0, TF>), grid, dim3(TPB), 0, st, xyz, center, idx, Wq, bq, Wk, bk, Wv, bv,
                           N, S, K, C, bx, alpha, np, ctx, argk);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_diffattn_xyz_fwd_f32(const float *xyz, const float *center, const int64_t *idx, const float *Wq,
                                        const float *bq, const float *Wk, const float *bk, const float *Wv,
                                        const float *bv, int B, int N, int S, int K, int C, float *ctx, uint8_t *argk,
                                        void *stream)
{
    return diffattn_xyz_fwd_any<float>(xyz, center, idx, Wq, bq, Wk, bk, Wv, bv, B, N, S, K, C, ctx, argk, stream);
}

extern "C" int mpa_diffattn_xyz_fwd_bf16(const float *xyz, const float *center, const int64_t *idx, const float *Wq,
                                         const float *bq, const float *Wk, const float *bk, const float *Wv,
                                         const float *bv, int B, int N, int S, int K, int C, mpa_bf16 *ctx,
                                         uint8_t *argk, void *stream)
{
    return diffattn_xyz_fwd_any<bf16_t>(xyz, center, idx, Wq, bq, Wk, bk, Wv, bv, B, N, S, K, C,
                                        reinterpret_cast<bf16_t *>(ctx), argk, stream);
}

template <typename TF>
static int diffattn_xyz_bwd_any(const float *xyz, const float *center, const int64_t *idx, const float *Wq,
                                const float *bq, const float *Wk, const float *bk, const float *Wv, const float *bv,
                                const uint8_t *argk, const TF *grad_ctx, int B, int N, int S, int K, int C, float *gWq,
                                float *gbq, float *gWk, float *gbk, float *gWv, float *gbv, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!xyz || !center || !idx || !Wq || !bq || !Wk || !bk || !Wv || !bv || !argk || !grad_ctx || !gWq || !gbq ||
        !gWk || !gbk || !gWv || !gbv || B <= 0 || N <= 0 || S <= 0 || K <= 0 || C <= 0)
        return MPA_EINVAL;
    if (K > KMAX) return MPA_EUNSUPPORTED;
    long long np = (long long)B * S;
    int bx = C > 128 ? 256 : (C > 64 ? 128 : 64);           // channel lanes: a divisor of 1024
    const int pw = BWD_TPB / bx;
    long long gx = (np + pw - 1) / pw;
    static const int cap = getenv("MPA_XYZ_BWD_BLOCKS") ? atoi(getenv("MPA_XYZ_BWD_BLOCKS")) : 256;
    dim3 grid((unsigned)(gx > cap ? cap : gx), mpa_ceil_div(C, bx));
    float alpha = 1.0f / sqrtf((float)C);
    hipStream_t st = (hipStream_t)stream;
    if (K == 8)
        hipLaunchKernelGGL((diffattn_xyz_bwd_kernel<8, TF>), grid, dim3(BWD_TPB), 0, st, xyz, center, idx, Wq, bq, Wk, bk, Wv,
                           bv, argk, grad_ctx, N, S, K, C, bx, alpha, np, gWq, gbq, gWk, gbk, gWv, gbv);
    else
        hipLaunchKernelGGL((diffattn_xyz_bwd_kernel<0, TF>), grid, dim3(BWD_TPB), 0, st, xyz, center, idx, Wq, bq, Wk, bk, Wv,
                           bv, argk, grad_ctx, N, S, K, C, bx, alpha, np, gWq, gbq, gWk, gbk, gWv, gbv);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_diffattn_xyz_bwd_f32(const float *xyz, const float *center, const int64_t *idx, const float *Wq,
                                        const float *bq, const float *Wk, const float *bk, const float *Wv,
                                        const float *bv, const uint8_t *argk, const float *grad_ctx, int B, int N,
                                        int S, int K, int C, float *gWq, float *gbq, float *gWk, float *gbk,
                                        float *gWv, float *gbv, void *stream)
{
    return diffattn_xyz_bwd_any<float>(xyz, center, idx, Wq, bq, Wk, bk, Wv, bv, argk, grad_ctx, B, N, S, K, C, gWq, gbq,
                                       gWk, gbk, gWv, gbv, stream);
}

extern "C" int mpa_diffattn_xyz_bwd_bf16(const float *xyz, const float *center, const int64_t *idx, const float *Wq,
                                         const float *bq, const float *Wk, const float *bk, const float *Wv,
                                         const float *bv, const uint8_t *argk, const mpa_bf16 *grad_ctx, int B, int N,
                                         int S, int K, int C, float *gWq, float *gbq, float *gWk, float *gbk,
                                         float *gWv, float *gbv, void *stream)
{
    return diffattn_xyz_bwd_any<bf16_t>(xyz, center, idx, Wq, bq, Wk, bk, Wv, bv, argk,
                                        reinterpret_cast<const bf16_t *>(grad_ctx), B, N, S, K, C, gWq, gbq, gWk, gbk, gWv,
                                        gbv, stream);
}
