// Farthest point sampling: the per-cloud workgroup body (see fps.hip for the description), shared by
// fps_kernel, the fused FPS + kNN launches of knn.hip / knn_fused.hip and the geometry riders of the grouped
// weight-gradient launches (linear.hip).
#pragma once
#include "mpa_common.h"

namespace {

// One sampling level on a cloud whose coordinates are already in LDS (SoA: sx, sy, sz with at least WAVES*64*P
// readable entries, of which the first N are points).  Threads [0, WAVES*64) of the workgroup take part (for
// WAVES == 1: wave 0 only, no barrier inside).  `slot` = 2*WAVES 64-bit words of LDS (WAVES > 1 only).  The sampled
// coordinates go to out_xyz (global, optional) and to nx/ny/nz (LDS SoA, optional: the source of a following level).
template <int WAVES, int P>
__device__ __forceinline__ void fps_level(const float *sx, const float *sy, const float *sz, int N, int S, int far,
                                          int64_t *__restrict__ out_idx, float *__restrict__ out_xyz, float *nx,
                                          float *ny, float *nz, unsigned long long *slot)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
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

    int par = 0;
    for (int it = 0; it < S; ++it) {
        float cx = sx[far], cy = sy[far], cz = sz[far];
        if (tid == 0) {
            out_idx[it] = far;
            if (out_xyz) {
                float *o = out_xyz + (size_t)it * 3;
                o[0] = cx; o[1] = cy; o[2] = cz;
            }
            if (nx) { nx[it] = cx; ny[it] = cy; nz[it] = cz; }
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
            // [2][WAVES] wave winners as ONE 64-bit key each: distance bits << 32 | ~index.  The largest key is the
            // largest distance and, among equal distances, the lowest index, so the waves are combined with
            // plain 64-bit maxima of one b64 read per slot (a (distance, index) pair made the compiler read
            // the distances, pick, and go back to LDS for the winner's index: one more dependent round trip
            // per iteration).
            if (lane == 0) slot[par * WAVES + wave] = ((unsigned long long)wmax << 32) | (unsigned)~widx;
            __syncthreads();
            unsigned long long bst = slot[par * WAVES];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) {
                const unsigned long long s = slot[par * WAVES + w];
                bst = s > bst ? s : bst;
            }
            far = (int)~(unsigned)bst;
            par ^= 1;
        }
    }
}

template <int WAVES, int P>
__device__ __forceinline__ void fps_body(const float *__restrict__ xyz, int N, int S,
                                         const int64_t *__restrict__ start, int64_t *__restrict__ out_idx,
                                         float *__restrict__ out_xyz, const int b, float *lds)
{
    constexpr int T = WAVES * 64;
    constexpr int NP = T * P;
    float *sx = lds, *sy = lds + NP, *sz = lds + 2 * NP;
    unsigned long long *slot = reinterpret_cast<unsigned long long *>(lds + 3 * NP);
    const int tid = threadIdx.x;
    const float *cloud = xyz + (size_t)b * N * 3;
    for (int i = tid; i < N * 3; i += T) {
        int n = i / 3;
        lds[(i - 3 * n) * NP + n] = cloud[i];
    }
    __syncthreads();
    fps_level<WAVES, P>(sx, sy, sz, N, S, (int)start[b], out_idx + (size_t)b * S,
                        out_xyz ? out_xyz + (size_t)b * S * 3 : nullptr, nullptr, nullptr, nullptr, slot);
}

}  // namespace
