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
