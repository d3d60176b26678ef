// Geometry riders: the sampling chain (and a coordinate search) of the NEXT batch, carried by the first workgroups
// of a long launch of the CURRENT step whose own workgroups are short -- the grouped weight-gradient launches that
// close a backward pass (linear.hip, linear_bf16.hip).
//
// Farthest point sampling is S dependent iterations on ONE workgroup per cloud: 0.39 ms of a 3.7 ms classification
// step during which 64 of 256 CUs work, and under HIP-graph replay nothing else runs beside it.  It depends on the
// coordinates only (reference modules/repsurface_utils.py:581-619, modules/pointnet2_utils.py:84-109), so the
// chain of batch t+1 can run while step t still computes: workgroups [0, B) of the carrying launch sample (one cloud
// each, all levels given to this rider in sequence, the sampled coordinates handed from level to level through LDS),
// the next qb*B workgroups run one coordinate search (knn_point of a state in the state before it, reference
// :190-222), the rest are the carrier's own.  Results are bit-identical to the stand-alone entry points
// (fps_level / knn_mfma_body are the same device bodies).
#pragma once
#include "mpa_common.h"
#include "fps_body.h"
#ifndef MPA_KNN_BODIES_ONLY
#define MPA_KNN_BODIES_ONLY
#endif
#include "knn.hip"
#include <cstdlib>

namespace {

constexpr int RIDER_LEVELS = 4;
constexpr int RIDER_MAX_N = 4096;

struct RiderArgs {
    const float *src;                       // [B][N][3] coordinates the first level samples from
    int B, N, nlev;                         // nlev == 0: no sampling workgroups
    int S[RIDER_LEVELS];
    const int64_t *start[RIDER_LEVELS];     // [B] first index per cloud and level
    int64_t *idx[RIDER_LEVELS];             // [B][S_j]
    float *xyz[RIDER_LEVELS];               // [B][S_j][3]
    const float *base, *query;              // coordinate search (base == nullptr: none): query [B][sS][3] in base [B][sN][3]
    float *dist;                            // [B][sS][sK]
    int64_t *kidx;
    int sN, sS, sK, qb;
    int res;                                // search with the base cloud resident in LDS (both passes out of LDS)
    int fps_blocks, blocks;                 // B (or 0); fps_blocks + qb*B
    int sblocks;                            // qb*B search workgroups (0: none)
    int *queue;                             // carried riders: word [8] counts finished sampling workgroups (zeroed by the caller)
};

// one level, width chosen by the number of points it samples FROM (as mpa_fps_f32 does: a single wave up to 512 points)
__device__ __forceinline__ void rider_level(const float *sx, const float *sy, const float *sz, int n, int S, int far,
                                            int64_t *oidx, float *oxyz, float *nx, float *ny, float *nz,
                                            unsigned long long *slot)
{
    const int wave = threadIdx.x >> 6;
#define RIDER_ONE(PP) do { if (wave == 0) fps_level<1, PP>(sx, sy, sz, n, S, far, oidx, oxyz, nx, ny, nz, slot); } while (0)
    if (n <= 64) RIDER_ONE(1);
    else if (n <= 128) RIDER_ONE(2);
    else if (n <= 256) RIDER_ONE(4);
    else if (n <= 512) RIDER_ONE(8);
    else if (n <= 1024) fps_level<4, 4>(sx, sy, sz, n, S, far, oidx, oxyz, nx, ny, nz, slot);
    else if (n <= 2048) fps_level<4, 8>(sx, sy, sz, n, S, far, oidx, oxyz, nx, ny, nz, slot);
    else fps_level<4, 16>(sx, sy, sz, n, S, far, oidx, oxyz, nx, ny, nz, slot);
#undef RIDER_ONE
}

// LDS of a sampling workgroup: region A = 3 * pad4(N) floats (source of levels 0, 2), region B = 3 * pad4(S[0]) floats
// (source of levels 1, 3; absent when nlev == 1), then 8 64-bit slots.
__device__ __forceinline__ int rider_pad4(int n) { return (n + 3) & ~3; }

__device__ __forceinline__ void rider_search(const RiderArgs &r, const int rb, float *lds)
{
    if (r.res)
        knn_mfma_body<3, 4, 8, 1, true, false>(r.base, r.query, r.sN, r.sS, r.sK, r.dist, r.kidx, rb % r.qb, rb / r.qb, lds);
    else
        knn_mfma_body<3, 4, 8, 1, false, false>(r.base, r.query, r.sN, r.sS, r.sK, r.dist, r.kidx, rb % r.qb, rb / r.qb, lds);
}

__device__ __forceinline__ void rider_sample(const RiderArgs &r, const int b, float *lds);

__device__ __forceinline__ void rider_body(const RiderArgs &r, const int bid, float *lds)
{
    if (bid >= r.fps_blocks) rider_search(r, bid - r.fps_blocks, lds);
    else rider_sample(r, bid, lds);
}

// Where a CARRIED rider's workgroups sit in the carrying launch.  A sampling workgroup that shares its CU with one of
// the carrier's MFMA workgroups runs 2.3x slower (431 us instead of 186 for the 1024 -> 512 level, measured; raising
// the sampling waves' priority changes nothing), which stretches the launch instead of hiding in it.  Workgroups b and b + 8 of a grid land on the same XCD (MI355X_MICROARCH.md, workgroup dispatch: observed,
// speed only), and an XCD holds `slots` = 32 CUs x (workgroups per CU of this kernel) of them at a time -- so the first
// `slots * nx` ids that are = 0..nx-1 (mod 8) are given to the rider: one per cloud samples, the rest PARK (thread 0
// sleeps until every sampling workgroup has finished, bounded) so that no carrier workgroup can be placed beside a
// sampling one.  All other ids are persistent workers that walk the launch's work items with a fixed stride (ids
// beyond one resident set would queue behind the rider's XCD lanes: the dispatcher places ids in order).  Nothing here
// is needed for correctness: any placement computes the same results.
// MEASURED (tools/rider_dw_bench.py, the 59 products of a classification step, 2 launches + reduces = 510 us):
// sampling riders placed this way 766 us, unplaced (beside MFMA workgroups) 794 us, as launches of their own before
// the products 845 us -- the lanes reserved for the chain stay idle for the rest of their launch, so carrying the
// chain saves ~80 us of its 340 us while the two searches it used to hide (64 + 30 us) come out in the open:
// runtime.GraphedTrainStep(prefetch_geometry=True) is therefore OFF by default.
struct RiderPlace {
    int region, nx, slots, per_lane;         // ids [0, region) with (id & 7) < nx are the rider's; the first per_lane
    int workers;                             // of each lane sample (cloud = lane * per_lane + position), the others park;
};                                           // workers: ids of the launch that are not the rider's

// -> cloud index (>= 0), -2: park, -1: not a rider id
__device__ __forceinline__ int rider_slot(const RiderPlace &pl, const int id)
{
    if (id >= pl.region || (id & 7) >= pl.nx) return -1;
    const int pos = id >> 3, lane = id & 7;
    return pos < pl.per_lane ? lane * pl.per_lane + pos : -2;
}

__device__ __forceinline__ void rider_sample_or_park(const RiderArgs &r, const int slot, float *lds)
{
    if (slot >= 0 && slot < r.fps_blocks) {
        rider_sample(r, slot, lds);
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(r.queue + 8, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (threadIdx.x != 0) return;
    for (int spins = 0; spins < 4000; ++spins) {          // ~1.7 us per turn: gives up after ~7 ms whatever happens
        if (__hip_atomic_load(r.queue + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= r.fps_blocks) break;
        __builtin_amdgcn_s_sleep(64);
    }
}

__device__ __forceinline__ void rider_sample(const RiderArgs &r, const int b, float *lds)
{
    const int tid = threadIdx.x;
    const int strideA = rider_pad4(r.N), strideB = rider_pad4(r.S[0]);
    float *regA = lds, *regB = lds + 3 * strideA;
    unsigned long long *slot = reinterpret_cast<unsigned long long *>(regB + (r.nlev > 1 ? 3 * strideB : 0));
    const float *cloud = r.src + (size_t)b * r.N * 3;
    for (int i = tid; i < r.N * 3; i += 256) {
        const int n = i / 3;
        regA[(i - 3 * n) * strideA + n] = cloud[i];
    }
    __syncthreads();
    int n = r.N;
    for (int j = 0; j < r.nlev; ++j) {
        const bool even = (j & 1) == 0;
        const float *s = even ? regA : regB;
        const int ss = even ? strideA : strideB;
        float *d = even ? regB : regA;
        const int ds = even ? strideB : strideA;
        const bool more = j + 1 < r.nlev;
        const int S = r.S[j];
        rider_level(s, s + ss, s + 2 * ss, n, S, (int)r.start[j][b], r.idx[j] + (size_t)b * S,
                    r.xyz[j] + (size_t)b * S * 3, more ? d : nullptr, more ? d + ds : nullptr, more ? d + 2 * ds : nullptr,
                    slot);
        __syncthreads();                      // the sampled coordinates (written by thread 0) are the next level's source
        n = S;
    }
}

__global__ __launch_bounds__(256) void geo_rider_kernel(const RiderArgs r)
{
    extern __shared__ float lds[];
    rider_body(r, blockIdx.x, lds);
}

// host: validate a rider of the C ABI and lay it out for the kernels.  -> MPA_OK / MPA_E*
inline int rider_prepare(const MpaGeoRider &in, RiderArgs &r, size_t &lds_bytes, size_t lds_budget = 64 * 1024)
{
    if (in.B <= 0 || in.nlev < 0 || in.nlev > RIDER_LEVELS) return MPA_EINVAL;
    r.src = in.src; r.B = in.B; r.N = in.N; r.nlev = in.nlev;
    size_t fps_lds = 0;
    if (in.nlev > 0) {
        if (!in.src || in.N <= 0) return MPA_EINVAL;
        if (in.N > RIDER_MAX_N) return MPA_EUNSUPPORTED;
        int n = in.N;
        for (int j = 0; j < RIDER_LEVELS; ++j) {
            r.S[j] = j < in.nlev ? in.S[j] : 0;
            r.start[j] = j < in.nlev ? in.start[j] : nullptr;
            r.idx[j] = j < in.nlev ? in.idx[j] : nullptr;
            r.xyz[j] = j < in.nlev ? in.xyz[j] : nullptr;
            if (j < in.nlev) {
                if (!in.start[j] || !in.idx[j] || !in.xyz[j] || in.S[j] <= 0) return MPA_EINVAL;
                if (j > 0 && in.S[j] > n) return MPA_EUNSUPPORTED;      // (a level may only repeat points when it is the first)
                n = in.S[j];
            }
        }
        const size_t a = 3 * (size_t)((in.N + 3) & ~3), b = in.nlev > 1 ? 3 * (size_t)((in.S[0] + 3) & ~3) : 0;
        if (in.nlev > 1 && in.S[0] > in.N) return MPA_EUNSUPPORTED;
        fps_lds = (a + b) * sizeof(float) + 8 * sizeof(unsigned long long);
    } else {
        for (int j = 0; j < RIDER_LEVELS; ++j) { r.S[j] = 0; r.start[j] = nullptr; r.idx[j] = nullptr; r.xyz[j] = nullptr; }
    }
    r.base = in.base; r.query = in.query; r.dist = in.dist; r.kidx = in.kidx;
    r.sN = in.sN; r.sS = in.sS; r.sK = in.sK; r.qb = 1;
    size_t knn_lds = 0;
    int sblocks = 0;
    if (in.base) {
        if (!in.query || !in.dist || !in.kidx || in.sN <= 0 || in.sS <= 0 || in.sK <= 0) return MPA_EINVAL;
        if (in.sK > 8 || in.sK > in.sN) return MPA_EUNSUPPORTED;
        r.qb = mpa_ceil_div(in.sS, 32);
        sblocks = r.qb * in.B;
        knn_lds = ((size_t)4 * (32 * (4 + 4) + 32) + KNN_G * 32 + 32 + 64 + 2 * 32 * KNN_CAP) * sizeof(float);
        const size_t merge = (size_t)32 * 2 * 4 * 8 * 8;
        if (merge > knn_lds) knn_lds = merge;
        // the resident form (a C = 3 tile visit out of LDS instead of a global round trip, knn.hip) when the cloud fits
        // what the carrying launch has anyway
        const size_t resident = knn_lds + (size_t)mpa_ceil_div(in.sN, 32) * 32 * 5 * sizeof(float);
        static const bool no_res = getenv("MPA_RIDER_NO_RESIDENT") != nullptr;
        r.res = (in.sN <= KNN_RES_MAX && resident <= lds_budget && !no_res) ? 1 : 0;
        if (r.res) knn_lds = resident;
    } else {
        r.res = 0;
    }
    r.fps_blocks = in.nlev > 0 ? in.B : 0;
    r.sblocks = sblocks;
    r.queue = in.queue;
    r.blocks = r.fps_blocks + sblocks;
    if (r.blocks == 0) return MPA_EINVAL;
    lds_bytes = fps_lds > knn_lds ? fps_lds : knn_lds;
    return MPA_OK;
}

inline int rider_launch_alone(const MpaGeoRider &in, hipStream_t st)
{
    RiderArgs r;
    size_t lds = 0;
    const int rc = rider_prepare(in, r, lds);
    if (rc != MPA_OK) return rc;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&geo_rider_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return MPA_EHIP;
    }
    hipLaunchKernelGGL(geo_rider_kernel, dim3(r.blocks), dim3(256), lds, st, r);
    return MPA_OK;
}

}  // namespace
