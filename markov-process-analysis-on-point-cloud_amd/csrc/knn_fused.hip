// Farthest point sampling of the NEXT state, the coordinate search of THIS state and its feature-space search in one
// launch.  FPS is S dependent iterations on one workgroup per cloud (32-64 of 256 CUs; 0.34 ms of a cls step, 0.79 ms
// of a part-seg step, not overlapped with anything under HIP-graph replay); the searches of the state before it
// depend on other data and fill the rest of the chip meanwhile -- the launch lasts as long as the longer of the two.
// Workgroups [0, B): fps_body; then the coordinate search's; then the feature search's (bodies of knn.hip).
#define MPA_KNN_BODIES_ONLY
#include "knn.hip"
#include "fps_body.h"
#include <cstdlib>

namespace {

struct FusedSearch {
    const float *base, *query, *norms;
    float *dist;
    int64_t *idx;
    int N, S, K, qb, blocks;               // qb: query workgroups per cloud
    int res;                               // coordinate searches: base cloud resident in LDS (both passes out of LDS)
};

// coordinate search of a fused launch: resident when the cloud fits the launch's LDS (a C = 3 tile visit out of LDS
// instead of a global round trip: the state-0 search of a 1024-point batch is 64 us resident, ~4x that otherwise)
__device__ __forceinline__ void fused_xyz_search(const FusedSearch &x, const int bid, float *lds)
{
    if (x.res)
        knn_mfma_body<3, 4, 8, 1, true, false>(x.base, x.query, x.N, x.S, x.K, x.dist, x.idx, bid % x.qb, bid / x.qb, lds);
    else
        knn_mfma_body<3, 4, 8, 1, false, false>(x.base, x.query, x.N, x.S, x.K, x.dist, x.idx, bid % x.qb, bid / x.qb, lds);
}

template <int P, int CT, int QG, bool GN>
__global__ __launch_bounds__(256) void fps_knn2_kernel(const float *__restrict__ fxyz, int fN, int fS,
                                                       const int64_t *__restrict__ start, int64_t *__restrict__ f_idx,
                                                       float *__restrict__ f_out_xyz, int B, FusedSearch x, FusedSearch y,
                                                       FusedSearch z)
{
    // z: a second coordinate search (round 3: the NEXT batch's state-0 search, carried beside this batch's state-1
    // searches together with the next batch's level-1 sampling: ops.GeometryPipeline)
    extern __shared__ float lds[];
    int bid = blockIdx.x;
    if (bid < B) {
        // up to 512 points a single wave samples faster than four (no barrier per iteration: 0.33 against 0.39 us,
        // fps.hip); the workgroup's other waves leave
        if constexpr (P <= 2) {
            if (threadIdx.x >= 64) return;
            fps_body<1, 4 * P>(fxyz, fN, fS, start, f_idx, f_out_xyz, bid, lds);
        } else {
            fps_body<4, P>(fxyz, fN, fS, start, f_idx, f_out_xyz, bid, lds);
        }
        return;
    }
    bid -= B;
    // (short coordinate-search workgroups first, the long feature-search workgroups last: the other order measured
    // slower in both forms of the carrying launch, 3.535 / 3.470 against 3.511 / 3.452 ms per classification step)
    if (bid < x.blocks) {
        fused_xyz_search(x, bid, lds);
        return;
    }
    bid -= x.blocks;
    if (bid < z.blocks) {
        fused_xyz_search(z, bid, lds);
        return;
    }
    bid -= z.blocks;
    knn_mfma_body<CT, 4, 8, QG, false, GN>(y.base, y.query, y.N, y.S, y.K, y.dist, y.idx, bid % y.qb, bid / y.qb, lds,
                                           y.norms);
}

template <int CT, int QG>
constexpr size_t feat_lds_bytes()
{
    constexpr int CP = (CT + 3) & ~3, NQ = 32 * QG;
    constexpr size_t work = ((size_t)4 * (32 * (CP + 4) + 32) + KNN_G * NQ + NQ + NQ + 32 + 2 * NQ * KNN_CAP) * sizeof(float);
    constexpr size_t merge = (size_t)32 * 2 * 4 * 8 * 8;
    return work > merge ? work : merge;
}

template <int P, int CT, int QG, bool GN>
int launch_fused(const float *fxyz, int B, int fN, int fS, const int64_t *start, int64_t *f_idx, float *f_out_xyz,
                 const FusedSearch &x_in, const FusedSearch &y, const FusedSearch &z_in, hipStream_t st)
{
    constexpr size_t xyz_lds = ((size_t)4 * (32 * (4 + 4) + 32) + KNN_G * 32 + 32 + 64 + 2 * 32 * KNN_CAP) * sizeof(float);
    constexpr size_t fps_lds = (size_t)3 * 256 * P * sizeof(float) + 2 * 4 * sizeof(uint2);
    constexpr size_t a = feat_lds_bytes<CT, QG>() > xyz_lds ? feat_lds_bytes<CT, QG>() : xyz_lds;
    constexpr size_t lds0 = a > fps_lds ? a : fps_lds;
    static_assert(lds0 <= 160 * 1024, "LDS of a gfx950 CU");
    // resident coordinate searches when the cloud fits what the launch allocates anyway
    FusedSearch x = x_in, z = z_in;
    auto resident = [&](FusedSearch &s) {
        const size_t need = xyz_lds + (size_t)mpa_ceil_div(s.N, 32) * 32 * 5 * sizeof(float);
        s.res = (s.blocks > 0 && s.N <= KNN_RES_MAX && need <= lds0) ? 1 : 0;
    };
    resident(x);
    resident(z);
    constexpr size_t lds = lds0;
    if (lds > 64 * 1024) {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&fps_knn2_kernel<P, CT, QG, GN>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (attr != hipSuccess) return MPA_EHIP;
    }
    const long long blocks = (long long)B + x.blocks + y.blocks + z.blocks;
    if (blocks >= 0x7fffffffLL) return MPA_EUNSUPPORTED;
    hipLaunchKernelGGL((fps_knn2_kernel<P, CT, QG, GN>), dim3((unsigned)blocks), dim3(256), lds, st, fxyz, fN, fS, start,
                       f_idx, f_out_xyz, B, x, y, z);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

template <int CT, int QG, bool GN>
int launch_fused_p(int P, const float *fxyz, int B, int fN, int fS, const int64_t *start, int64_t *f_idx,
                   float *f_out_xyz, const FusedSearch &x, const FusedSearch &y, const FusedSearch &z, hipStream_t st)
{
    switch (P) {
    case 1: return launch_fused<1, CT, QG, GN>(fxyz, B, fN, fS, start, f_idx, f_out_xyz, x, y, z, st);
    case 2: return launch_fused<2, CT, QG, GN>(fxyz, B, fN, fS, start, f_idx, f_out_xyz, x, y, z, st);
    case 4: return launch_fused<4, CT, QG, GN>(fxyz, B, fN, fS, start, f_idx, f_out_xyz, x, y, z, st);
    case 8: return launch_fused<8, CT, QG, GN>(fxyz, B, fN, fS, start, f_idx, f_out_xyz, x, y, z, st);
    default: return launch_fused<16, CT, QG, GN>(fxyz, B, fN, fS, start, f_idx, f_out_xyz, x, y, z, st);
    }
}

// ================================================================================= the coarse states
// The last states of a chain are a few dozen to a few hundred points: 128 -> 64 -> 32 in the classification model.
// There every launch of the general kernels is latency -- one or two workgroups per cloud walking 32-row tiles with
// 4-8 dependent global round trips each (C = 256: 50 us at 1.3 % matrix-pipe utilisation, C = 128: 32 us, C = 64:
// 29 us), a sampling launch of 8-15 us and a coordinate search of 12-15 us per state, 8 launches and 174 us for the
// three coarsest states of a classification step.  Here a state's whole geometry step is ONE launch of small
// workgroups: [0, B) sample the next state (one wave, cloud in registers), the next qx*B search coordinates, the rest
// search features -- and a search workgroup stages its cloud's WHOLE base (N <= 256 rows) with every load issued
// before the first use, computes all 32 x N distances in one pass (one or two 32-row tiles per wave, the same MFMA
// chain and rounding as knn_mfma_body: bit-identical distances), and selects by K rounds of a 64-bit (distance, index)
// minimum over 8 lanes per query.
constexpr int SMALL_MAX_N = 256;

template <int CT>
constexpr size_t small_lds_bytes(int N)
{
    const int NP = (N + 31) / 32 * 32;
    return ((size_t)(NP + 32) * (((CT + 3) & ~3) + 4) + NP + 32 * (size_t)(NP + 1)) * sizeof(float);
}

template <int CT>
__device__ __forceinline__ void knn_small_body(const float *__restrict__ base, const float *__restrict__ query, int N,
                                               int S, int K, float *__restrict__ out_dist, int64_t *__restrict__ out_idx,
                                               const int bx, const int b, float *lds)
{
    constexpr int CP = (CT + 3) & ~3, PITCH = CP + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int ntiles = (N + 31) / 32, NP = ntiles * 32;
    float *rows = lds;                               // [NP][PITCH] base rows
    float *qrows = rows + (size_t)NP * PITCH;        // [32][PITCH] this workgroup's queries
    float *snorm = qrows + 32 * PITCH;               // [NP]
    float *dmat = snorm + NP;                        // [32][NP + 1]
    const float *bp = base + (size_t)b * N * CT;
    const int q0 = bx * 32;
    if constexpr (CT == 3) {
        for (int i = tid; i < NP * 4; i += 256) {
            const int r = i >> 2, c = i & 3;
            rows[r * PITCH + c] = (r < N && c < 3) ? bp[r * 3 + c] : 0.f;
        }
    } else {
        constexpr int V = CT / 4;                    // float4 per row
        const float4 *src = reinterpret_cast<const float4 *>(bp);
        const int live = N * V, all = NP * V;
        for (int i0 = tid; i0 < all; i0 += 256 * 8) {
            float4 tmp[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 256 * u;
                tmp[u] = i < live ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 256 * u;
                if (i < all) {
                    const int r = i / V, c4 = i - r * V;
                    *reinterpret_cast<float4 *>(rows + (size_t)r * PITCH + 4 * c4) = tmp[u];
                }
            }
        }
    }
    // the 32 query rows next to them (rows past S repeat the last one; their results are not stored): the MFMA's
    // query operand q[j][2kk + half] is then one 16-byte LDS read per two products, like the base operand, instead of
    // CT/2 registers per lane
    if constexpr (CT == 3) {
        for (int i = tid; i < 32 * 4; i += 256) {
            const int r = i >> 2, c = i & 3;
            qrows[r * PITCH + c] = c < 3 ? query[((size_t)b * S + min(q0 + r, S - 1)) * 3 + c] : 0.f;
        }
    } else {
        constexpr int V = CT / 4;
        for (int i = tid; i < 32 * V; i += 256) {
            const int r = i / V, c4 = i - r * V;
            *reinterpret_cast<float4 *>(qrows + (size_t)r * PITCH + 4 * c4) =
                reinterpret_cast<const float4 *>(query + ((size_t)b * S + min(q0 + r, S - 1)) * CT)[c4];
        }
    }
    __syncthreads();
    for (int r = tid; r < NP; r += 256) snorm[r] = r < N ? tile_row_norm<CT>(rows + (size_t)r * PITCH) : INFINITY;
    const float *qrow = qrows + (size_t)l31 * PITCH;
    const float qn = tile_row_norm<CT>(qrow);          // |q|^2 by the A2 model (same order as sum_sq_model)
    __syncthreads();
    for (int t = wave; t < ntiles; t += 4) {
        const float *row = rows + (size_t)(t * 32 + l31) * PITCH;
        float4 sn[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) sn[q4] = *reinterpret_cast<const float4 *>(snorm + t * 32 + 8 * q4 + 4 * half);
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int m = 0; m < CP / 4; ++m) {
            const float4 a = *reinterpret_cast<const float4 *>(row + 4 * m);
            const float4 qq = *reinterpret_cast<const float4 *>(qrow + 4 * m);
            const float a0 = half ? a.y : a.x, a1 = half ? a.w : a.z;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, half ? qq.y : qq.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, half ? qq.w : qq.z, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = (-2.0f * acc[r] + qn) + (&sn[r >> 2].x)[r & 3];
            dmat[l31 * (NP + 1) + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = d;
        }
    }
    __syncthreads();
    // selection: 8 lanes per query, lane e owns the candidates n = e + 8 j; K rounds of a 64-bit key minimum
    // (monotone distance key << 32 | n: smaller distance first, lower index among equals -- the (distance, index) rank)
    const int q = tid >> 3, e = tid & 7;
    unsigned long long key[SMALL_MAX_N / 8];
#pragma unroll
    for (int j = 0; j < SMALL_MAX_N / 8; ++j) {
        const int n = e + 8 * j;
        key[j] = n < N ? ((unsigned long long)((unsigned)f2key(dmat[q * (NP + 1) + n]) ^ 0x80000000u) << 32) | (unsigned)n
                       : ~0ull;
    }
    const bool live = q0 + q < S;
    const size_t o = ((size_t)b * S + min(q0 + q, S - 1)) * K;
    for (int k = 0; k < K; ++k) {
        unsigned long long m = key[0];
#pragma unroll
        for (int j = 1; j < SMALL_MAX_N / 8; ++j) m = key[j] < m ? key[j] : m;
        unsigned long long w = m;
#pragma unroll
        for (int sh = 1; sh < 8; sh <<= 1) {
            const unsigned long long other = __shfl_xor(w, sh, 64);
            w = other < w ? other : w;
        }
        if (m == w && w != ~0ull) {                    // this lane owns the winner: retire it
#pragma unroll
            for (int j = 0; j < SMALL_MAX_N / 8; ++j)
                if (key[j] == w) key[j] = ~0ull;
        }
        if (e == 0 && live && w != ~0ull) {
            const int kb = (int)((unsigned)(w >> 32) ^ 0x80000000u);
            out_idx[o + k] = (int64_t)(unsigned)(w & 0xffffffffu);
            if (out_dist) out_dist[o + k] = __int_as_float(kb >= 0 ? kb : kb ^ 0x7fffffff);
        }
    }
}

struct SmallSearch {
    const float *base, *query;
    float *dist;
    int64_t *idx;
    int N, S, K, qb, blocks;
};

template <int CT>
__global__ __launch_bounds__(256) void coarse_level_kernel(const float *__restrict__ fxyz, int fN, int fS,
                                                           const int64_t *__restrict__ start, int64_t *__restrict__ f_idx,
                                                           float *__restrict__ f_out_xyz, int fB, SmallSearch x, SmallSearch y)
{
    extern __shared__ float lds[];
    int bid = blockIdx.x;
    if (bid < fB) {
        // one wave samples (the cloud in registers, no barrier inside the loop); the others leave
        float *sx = lds, *sy = lds + 256, *sz = lds + 512;
        const float *cloud = fxyz + (size_t)bid * fN * 3;
        for (int i = threadIdx.x; i < fN * 3; i += 256) {
            const int n = i / 3;
            lds[(i - 3 * n) * 256 + n] = cloud[i];
        }
        __syncthreads();
        if (threadIdx.x >= 64) return;
        int64_t *oi = f_idx + (size_t)bid * fS;
        float *ox = f_out_xyz + (size_t)bid * fS * 3;
        if (fN <= 64) fps_level<1, 1>(sx, sy, sz, fN, fS, (int)start[bid], oi, ox, nullptr, nullptr, nullptr, nullptr);
        else if (fN <= 128) fps_level<1, 2>(sx, sy, sz, fN, fS, (int)start[bid], oi, ox, nullptr, nullptr, nullptr, nullptr);
        else fps_level<1, 4>(sx, sy, sz, fN, fS, (int)start[bid], oi, ox, nullptr, nullptr, nullptr, nullptr);
        return;
    }
    bid -= fB;
    if (bid < x.blocks) {
        knn_small_body<3>(x.base, x.query, x.N, x.S, x.K, x.dist, x.idx, bid % x.qb, bid / x.qb, lds);
        return;
    }
    bid -= x.blocks;
    knn_small_body<CT>(y.base, y.query, y.N, y.S, y.K, y.dist, y.idx, bid % y.qb, bid / y.qb, lds);
}

template <int CT>
int launch_coarse(const float *fxyz, int fB, int fN, int fS, const int64_t *start, int64_t *f_idx, float *f_out_xyz,
                  const SmallSearch &x, const SmallSearch &y, hipStream_t st)
{
    size_t lds = small_lds_bytes<CT>(y.N);
    if (x.blocks > 0 && small_lds_bytes<3>(x.N) > lds) lds = small_lds_bytes<3>(x.N);
    if (lds < 3 * 256 * sizeof(float)) lds = 3 * 256 * sizeof(float);
    if (lds > 160 * 1024) return MPA_EUNSUPPORTED;
    if (lds > 64 * 1024) {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&coarse_level_kernel<CT>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (attr != hipSuccess) return MPA_EHIP;
    }
    hipLaunchKernelGGL((coarse_level_kernel<CT>), dim3(fB + x.blocks + y.blocks), dim3(256), lds, st, fxyz, fN, fS, start,
                       f_idx, f_out_xyz, fB, x, y);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

}  // namespace

// A coarse state's geometry step in one launch (see above): optional sampling of fps_S from fps_N <= 256 points,
// optional coordinate search (xN <= 256), feature search with N <= 256 rows of C in {32, 64, 128, 256} floats, K <= 8.
// MPA_EUNSUPPORTED outside those shapes.  Results are bit-identical to mpa_fps_f32 / mpa_knn_f32.
extern "C" int mpa_coarse_level_f32(const float *fps_xyz, int B, int fps_N, int fps_S, const int64_t *start_idx,
                                    int64_t *fps_idx, float *fps_out_xyz, const float *xyz_base, const float *xyz_query,
                                    int xN, int xS, int xK, float *xyz_dist, int64_t *xyz_idx, const float *feat_base,
                                    const float *feat_query, int N, int S, int C, int K, float *out_dist,
                                    int64_t *out_idx, void *stream)
{
    MPA_CLEAR_ERROR();
    const bool sample = fps_xyz != nullptr;
    if (!feat_base || !feat_query || !out_idx || B <= 0 || N <= 0 || S <= 0 || K <= 0) return MPA_EINVAL;
    if (sample && (!start_idx || !fps_idx || !fps_out_xyz || fps_N <= 0 || fps_S <= 0)) return MPA_EINVAL;
    if (xyz_base && (!xyz_query || !xyz_idx || xN <= 0 || xS <= 0 || xK <= 0)) return MPA_EINVAL;
    if (K > 8 || K > N || N > SMALL_MAX_N || (xyz_base && (xK > 8 || xK > xN || xN > SMALL_MAX_N)) ||
        (sample && fps_N > 256) || (((uintptr_t)feat_base | (uintptr_t)feat_query) & 15) != 0)
        return MPA_EUNSUPPORTED;
    SmallSearch x, y;
    x.base = xyz_base; x.query = xyz_query; x.dist = xyz_dist; x.idx = xyz_idx; x.N = xN; x.S = xS; x.K = xK;
    x.qb = xyz_base ? mpa_ceil_div(xS, 32) : 1; x.blocks = xyz_base ? x.qb * B : 0;
    y.base = feat_base; y.query = feat_query; y.dist = out_dist; y.idx = out_idx; y.N = N; y.S = S; y.K = K;
    y.qb = mpa_ceil_div(S, 32); y.blocks = y.qb * B;
    hipStream_t st = (hipStream_t)stream;
    const int fB = sample ? B : 0;
    switch (C) {
    case 32: return launch_coarse<32>(fps_xyz, fB, fps_N, fps_S, start_idx, fps_idx, fps_out_xyz, x, y, st);
    case 64: return launch_coarse<64>(fps_xyz, fB, fps_N, fps_S, start_idx, fps_idx, fps_out_xyz, x, y, st);
    case 128: return launch_coarse<128>(fps_xyz, fB, fps_N, fps_S, start_idx, fps_idx, fps_out_xyz, x, y, st);
    case 256: return launch_coarse<256>(fps_xyz, fB, fps_N, fps_S, start_idx, fps_idx, fps_out_xyz, x, y, st);
    default: return MPA_EUNSUPPORTED;
    }
}

extern "C" int mpa_fps_knn_feat_f32(const float *fps_xyz, int B, int fps_N, int fps_S, const int64_t *start_idx,
                                    int64_t *fps_idx, float *fps_out_xyz, const float *xyz_base,
                                    const float *xyz_query, int xN, int xS, int xK, float *xyz_dist, int64_t *xyz_idx,
                                    const float *feat_base, const float *feat_norms, const float *feat_query, int N,
                                    int S, int C, int K, float *out_dist, int64_t *out_idx, void *stream)
{
    return mpa_geo_level_f32(fps_xyz, B, fps_N, fps_S, start_idx, fps_idx, fps_out_xyz, xyz_base, xyz_query, xN, xS, xK,
                             xyz_dist, xyz_idx, nullptr, nullptr, 0, 0, 0, nullptr, nullptr, feat_base, feat_norms,
                             feat_query, N, S, C, K, out_dist, out_idx, stream);
}

extern "C" int mpa_geo_level_f32(const float *fps_xyz, int B, int fps_N, int fps_S, const int64_t *start_idx,
                                 int64_t *fps_idx, float *fps_out_xyz, const float *xyz_base, const float *xyz_query,
                                 int xN, int xS, int xK, float *xyz_dist, int64_t *xyz_idx, const float *xyz2_base,
                                 const float *xyz2_query, int zN, int zS, int zK, float *xyz2_dist, int64_t *xyz2_idx,
                                 const float *feat_base, const float *feat_norms, const float *feat_query, int N, int S,
                                 int C, int K, float *out_dist, int64_t *out_idx, void *stream)
{
    MPA_CLEAR_ERROR();
    if (xyz2_base && (!xyz2_query || !xyz2_idx || !xyz2_dist || zN <= 0 || zS <= 0 || zK <= 0)) return MPA_EINVAL;
    if (xyz2_base && (zK > 8 || zK > zN)) return MPA_EUNSUPPORTED;
    // fps_xyz == NULL: no sampling workgroups -- the two searches of a state alone in one launch (the sampling chain
    // of that pass was computed a step ahead by geometry riders, mpa_gemm_tn_grouped_rider_f32)
    const bool sample = fps_xyz != nullptr;
    if (!feat_base || !feat_query || !out_idx || B <= 0 || N <= 0 || S <= 0 || K <= 0) return MPA_EINVAL;
    if (sample && (!start_idx || !fps_idx || fps_N <= 0 || fps_S <= 0)) return MPA_EINVAL;
    if (xyz_base && (!xyz_query || !xyz_idx || xN <= 0 || xS <= 0 || xK <= 0)) return MPA_EINVAL;
    // the instantiated combinations: FPS of 129..4096 points, K <= 8, feature rows of 64 / 128 floats, 16-byte aligned
    if (K > 8 || K > N || (xyz_base && (xK > 8 || xK > xN)) || (sample && (fps_N <= 128 || fps_N > 4096)) ||
        (C != 64 && C != 128) || ((((uintptr_t)feat_base | (uintptr_t)feat_query | (uintptr_t)feat_norms) & 15) != 0))
        return MPA_EUNSUPPORTED;
    const int P = !sample || fps_N <= 256 ? 1 : (fps_N <= 512 ? 2 : (fps_N <= 1024 ? 4 : (fps_N <= 2048 ? 8 : 16)));
    const int fB = sample ? B : 0;               // sampling workgroups in front of the searches'
    FusedSearch x, y;
    x.base = xyz_base; x.query = xyz_query; x.norms = nullptr; x.dist = xyz_dist; x.idx = xyz_idx;
    x.N = xN; x.S = xS; x.K = xK; x.qb = xyz_base ? mpa_ceil_div(xS, 32) : 1; x.blocks = xyz_base ? x.qb * B : 0;
    y.base = feat_base; y.query = feat_query; y.norms = feat_norms; y.dist = out_dist; y.idx = out_idx;
    y.N = N; y.S = S; y.K = K;
    FusedSearch z;
    z.base = xyz2_base; z.query = xyz2_query; z.norms = nullptr; z.dist = xyz2_dist; z.idx = xyz2_idx;
    z.N = zN; z.S = zS; z.K = zK; z.qb = xyz2_base ? mpa_ceil_div(zS, 32) : 1; z.blocks = xyz2_base ? z.qb * B : 0;
    hipStream_t st = (hipStream_t)stream;
    static const int carry_qg = getenv("MPA_CARRY_QG") ? atoi(getenv("MPA_CARRY_QG")) : 1;
    if (C == 64 && feat_norms != nullptr && fB > 0 && xyz2_base != nullptr && carry_qg == 1) {
        // the launch that carries a whole sampling level of the next batch beside a second coordinate search (the
        // cross-step pipeline's state-1 launch): ONE query group per workgroup with the norms given -- shorter workgroups
        // and 48 KiB of LDS / fewer registers (three workgroups per CU), so the feature search packs around the B
        // sampling workgroups instead of leaving its last B workgroups to wait for their slots (324 us with two groups)
        y.qb = mpa_ceil_div(S, 32); y.blocks = y.qb * B;
        return launch_fused_p<64, 1, true>(P, fps_xyz, fB, fps_N, fps_S, start_idx, fps_idx, fps_out_xyz, x, y, z, st);
    }
    if (C == 64 && feat_norms != nullptr) {                      // two query groups per workgroup (needs the norms)
        y.qb = mpa_ceil_div(S, 64); y.blocks = y.qb * B;
        return launch_fused_p<64, 2, true>(P, fps_xyz, fB, fps_N, fps_S, start_idx, fps_idx, fps_out_xyz, x, y, z, st);
    }
    y.qb = mpa_ceil_div(S, 32); y.blocks = y.qb * B;
    y.norms = nullptr;
    if (C == 64) return launch_fused_p<64, 1, false>(P, fps_xyz, fB, fps_N, fps_S, start_idx, fps_idx, fps_out_xyz, x, y, z, st);
    return launch_fused_p<128, 1, false>(P, fps_xyz, fB, fps_N, fps_S, start_idx, fps_idx, fps_out_xyz, x, y, z, st);
}
