// Farthest point sampling for gfx950 -- replaces the python loop of
// farthest_point_sample (reference modules/pointnet2_utils.py:84-109).
//
// One workgroup per cloud, the whole cloud resident on chip for all S dependent iterations:
//   * coordinates staged once into LDS (SoA), then held in registers, P contiguous points
//     per lane, together with the running min-distance;
//   * per iteration each lane updates its P points (sub, 3 separately-rounded squares, 2 adds,
//     compare-select: the reference's rounding, no FMA contraction) and keeps its local first
//     maximum;
//   * wave argmax: the distances are non-negative, so their bit patterns order as unsigned
//     ints: 4 DPP max steps + 4 v_readlane give the wave maximum on the scalar unit, a ballot
//     + s_ff1 picks the lowest lane holding it (= lowest index, since lanes own ascending
//     contiguous index ranges): torch.max's "first maximum";
//   * waves are combined through a double-buffered LDS slot: one s_barrier per iteration.
// HBM traffic is 12*N + 8*S (+12*S) bytes per cloud; the kernel is latency bound by design
// (S serial iterations), see DESIGN.md.
#include "mpa_common.h"
#include "fps_body.h"

namespace {

template <int WAVES, int P>
__global__ __launch_bounds__(WAVES * 64) void fps_kernel(const float *__restrict__ xyz, int N, int S,
                                                         const int64_t *__restrict__ start,
                                                         int64_t *__restrict__ out_idx,
                                                         float *__restrict__ out_xyz)
{
    extern __shared__ float lds[];
    fps_body<WAVES, P>(xyz, N, S, start, out_idx, out_xyz, blockIdx.x, lds);
}


// ---- any channel count (reference :84-109 takes xyz [B,N,C] for any C: the ShapeNetPart reader samples on
// xyz|normal rows, an earlier part-seg variant in feature space).  Not on the models' path: one workgroup
// per cloud, rows read from global memory (L2 resident) every iteration, running minima in LDS, and the
// reference's summation order of torch.sum((x - c)**2, -1) written out (squares
// rounded on their own; C < 8 left to right; otherwise 8 lanes x min(4, C/8) accumulators, accumulators
// combined ((a0+a1)+a2)+a3, lanes left to right, then the C % 8 tail).
__device__ __forceinline__ float fps_dist2_generic(const float *__restrict__ p, const float *__restrict__ c, int C)
{
    if (C < 8) {
        float d = p[0] - c[0];
        float s = d * d;
        for (int k = 1; k < C; ++k) {
            d = p[k] - c[k];
            const float sq = d * d;
            s = s + sq;
        }
        return s;
    }
    const int nv = C >> 3;
    float acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int l = 0; l < 8; ++l) acc[a][l] = 0.f;          // 0 + x is exact: same as starting from the square
    const int A = nv < 4 ? nv : 4;
    for (int v0 = 0; v0 < nv; v0 += A) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (a < A && v0 + a < nv) {
#pragma unroll
                for (int l = 0; l < 8; ++l) {
                    const int k = (v0 + a) * 8 + l;
                    const float d = p[k] - c[k];
                    const float sq = d * d;
                    acc[a][l] = acc[a][l] + sq;
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        const float t = ((acc[0][l] + acc[1][l]) + acc[2][l]) + acc[3][l];     // absent accumulators are exact zeros
        s = l == 0 ? t : s + t;
    }
    for (int k = nv * 8; k < C; ++k) {
        const float d = p[k] - c[k];
        const float sq = d * d;
        s = s + sq;
    }
    return s;
}

constexpr int FPSG_T = 256;

__global__ __launch_bounds__(FPSG_T) void fps_generic_kernel(const float *__restrict__ pts, int N, int C, int S,
                                                             const int64_t *__restrict__ start,
                                                             int64_t *__restrict__ out_idx)
{
    extern __shared__ float lds[];            // running minima [N], then [2][4] winner keys, then the centre row [C]
    float *md = lds;
    unsigned long long *slot = reinterpret_cast<unsigned long long *>(lds + ((N + 1) & ~1));
    float *crow = reinterpret_cast<float *>(slot + 8);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const float *cloud = pts + (size_t)b * N * C;
    for (int n = tid; n < N; n += FPSG_T) md[n] = 1e10f;
    int far = (int)start[b];
    int par = 0;
    for (int it = 0; it < S; ++it) {
        __syncthreads();                       // previous iteration's reads of crow are done
        for (int k = tid; k < C; k += FPSG_T) crow[k] = cloud[(size_t)far * C + k];
        if (tid == 0) out_idx[(size_t)b * S + it] = far;
        __syncthreads();
        // key = distance bits << 32 | ~index: the largest key is the largest distance, lowest index first
        unsigned long long best = 0ull;
        for (int n = tid; n < N; n += FPSG_T) {
            const float dd = fps_dist2_generic(cloud + (size_t)n * C, crow, C);
            float m = md[n];
            m = dd < m ? dd : m;
            md[n] = m;
            const unsigned long long key = ((unsigned long long)__float_as_uint(m) << 32) | (unsigned)~n;
            best = key > best ? key : best;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(best, o, 64);
            best = other > best ? other : best;
        }
        if (lane == 0) slot[par * 4 + wave] = best;
        __syncthreads();
        unsigned long long bst = slot[par * 4];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const unsigned long long s2 = slot[par * 4 + w];
            bst = s2 > bst ? s2 : bst;
        }
        far = (int)~(unsigned)bst;
        par ^= 1;
    }
}

template <int WAVES, int P>
int launch(const float *xyz, int B, int N, int S, const int64_t *start, int64_t *out_idx, float *out_xyz,
           hipStream_t st)
{
    size_t lds = (size_t)3 * WAVES * 64 * P * sizeof(float) + 2 * WAVES * sizeof(uint2);
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&fps_kernel<WAVES, P>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return MPA_EHIP;
    }
    hipLaunchKernelGGL((fps_kernel<WAVES, P>), dim3(B), dim3(WAVES * 64), lds, st, xyz, N, S, start, out_idx,
                       out_xyz);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

}  // namespace

extern "C" int mpa_fps_f32(const float *xyz, int B, int N, int S, const int64_t *start_idx, int64_t *out_idx,
                           float *out_xyz, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!xyz || !start_idx || !out_idx || B <= 0 || N <= 0 || S <= 0) return MPA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
#define FPS_CASE(MAXN, W, P) \
    if (N <= (MAXN)) return launch<W, P>(xyz, B, N, S, start_idx, out_idx, out_xyz, st);
    // (waves, points per lane) by measurement: an iteration costs ~0.36-0.45 us whatever the split --
    // it is the chain LDS read -> update -> DPP arg-max -> LDS slot -> barrier -> LDS read -- and a
    // single wave (no barrier) wins up to N = 512; at N = 1024 two waves cost 0.42, four 0.39, eight 0.60
    // and sixteen 0.70 us per iteration (the barrier gets expensive beyond one wave per SIMD).
    FPS_CASE(64, 1, 1)
    FPS_CASE(128, 1, 2)
    FPS_CASE(256, 1, 4)
    FPS_CASE(512, 1, 8)
    FPS_CASE(1024, 4, 4)
    FPS_CASE(2048, 4, 8)
    FPS_CASE(4096, 16, 4)
    FPS_CASE(8192, 16, 8)
    FPS_CASE(12288, 16, 12)     // 144 KB of LDS: ModelNet's 10000-point shapes (dataset/ModelNetDataLoader.py:47)
#undef FPS_CASE
    return MPA_EUNSUPPORTED;
}

extern "C" int mpa_fps_generic_f32(const float *points, int B, int N, int C, int S, const int64_t *start_idx,
                                   int64_t *out_idx, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!points || !start_idx || !out_idx || B <= 0 || N <= 0 || C <= 0 || S <= 0) return MPA_EINVAL;
    const size_t lds = sizeof(float) * (size_t)((N + 1) & ~1) + 8 * sizeof(unsigned long long) + sizeof(float) * (size_t)C;
    if (lds > 160 * 1024) return MPA_EUNSUPPORTED;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(&fps_generic_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return MPA_EHIP;
    hipLaunchKernelGGL(fps_generic_kernel, dim3(B), dim3(FPSG_T), lds, (hipStream_t)stream, points, N, C, S, start_idx,
                       out_idx);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}
