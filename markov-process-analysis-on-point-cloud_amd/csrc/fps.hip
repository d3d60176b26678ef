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

namespace {

template <int WAVES, int P>
__global__ __launch_bounds__(WAVES * 64) void fps_kernel(const float *__restrict__ xyz, int N, int S,
                                                         const int64_t *__restrict__ start,
                                                         int64_t *__restrict__ out_idx,
                                                         float *__restrict__ out_xyz)
{
    constexpr int T = WAVES * 64;
    constexpr int NP = T * P;
    extern __shared__ float lds[];
    float *sx = lds, *sy = lds + NP, *sz = lds + 2 * NP;
    uint2 *slot = reinterpret_cast<uint2 *>(lds + 3 * NP);   // [2][WAVES]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int b = blockIdx.x;
    const float *cloud = xyz + (size_t)b * N * 3;

    for (int i = tid; i < N * 3; i += T) {
        int n = i / 3;
        lds[(i - 3 * n) * NP + n] = cloud[i];
    }
    __syncthreads();

    float px[P], py[P], pz[P], md[P];
    const int first = tid * P;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int n = first + p;
        bool ok = n < N;
        px[p] = ok ? sx[n] : 0.f;
        py[p] = ok ? sy[n] : 0.f;
        pz[p] = ok ? sz[n] : 0.f;
        md[p] = ok ? 1e10f : 0.f;   // padding points stay at 0 and can never win a maximum
    }

    int far = (int)start[b];
    int par = 0;
    for (int it = 0; it < S; ++it) {
        float cx = sx[far], cy = sy[far], cz = sz[far];
        if (tid == 0) {
            out_idx[(size_t)b * S + it] = far;
            if (out_xyz) {
                float *o = out_xyz + ((size_t)b * S + it) * 3;
                o[0] = cx; o[1] = cy; o[2] = cz;
            }
        }
        unsigned best = 0;
        int bestp = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            float dx = px[p] - cx, dy = py[p] - cy, dz = pz[p] - cz;
            float dd = (dx * dx + dy * dy) + dz * dz;
            float m = md[p];
            m = dd < m ? dd : m;
            md[p] = m;
            unsigned ub = __float_as_uint(m);
            if (ub > best) { best = ub; bestp = p; }
        }
        unsigned wmax = wave_max_u32(best);
        unsigned long long hit = __ballot(best == wmax);
        int wl = __ffsll((long long)hit) - 1;
        int widx = __builtin_amdgcn_readlane(first + bestp, wl);
        if (WAVES == 1) {
            far = widx;
        } else {
            if (lane == 0) slot[par * WAVES + wave] = make_uint2(wmax, (unsigned)widx);
            __syncthreads();
            uint2 bst = slot[par * WAVES];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) {
                uint2 s = slot[par * WAVES + w];
                if (s.x > bst.x) bst = s;
            }
            far = (int)bst.y;
            par ^= 1;
        }
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
    // single wave (no barrier) wins up to N = 512; 16 waves were 1.8x slower at N = 1024.
    FPS_CASE(64, 1, 1)
    FPS_CASE(128, 1, 2)
    FPS_CASE(256, 1, 4)
    FPS_CASE(512, 1, 8)
    FPS_CASE(1024, 4, 4)
    FPS_CASE(2048, 4, 8)
    FPS_CASE(4096, 16, 4)
    FPS_CASE(8192, 16, 8)
#undef FPS_CASE
    return MPA_EUNSUPPORTED;
}
