// Shared helpers for the gfx950 kernels of libmpa_hip.so.
// All translation units are built with -ffp-contract=off: products and sums are separate
// roundings unless fmaf() is written, which is what the bit-exact index kernels rely on.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mpa_hip.h"

#define MPA_WAVE 64

// torch's own runtime calls (event/stream queries) leave benign errors such as hipErrorNotReady
// in the per-thread "last error" slot; drop them before launching so that MPA_LAUNCH_CHECK
// reports only this library's launches.
#define MPA_CLEAR_ERROR() (void)hipGetLastError()

// defined in api.hip: remembers the hipError_t behind the last MPA_EHIP of this thread
void mpa_note_hip_error(int hip_error);

#define MPA_LAUNCH_CHECK()                          \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) {                    \
            mpa_note_hip_error((int)e__);           \
            return MPA_EHIP;                        \
        }                                           \
    } while (0)

// Indices arriving through the ABI are trusted to be in range, but a stray value (e.g. computed
// from NaN features upstream) must never become a wild address: clamp before dereferencing.
__device__ __forceinline__ long long mpa_clamp_idx(long long i, int n)
{
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

static inline int mpa_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- DPP helpers (wave64, gfx9 DPP controls) ------------------------------------------
// dpp_ctrl encodings: quad_perm = 0x00..0xFF, row_shr:n = 0x110+n, row_ror:n = 0x120+n,
// row_mirror = 0x140, row_half_mirror = 0x141.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

// max over the 64 lanes of an unsigned value, returned wave-uniform.
// 4 DPP steps give every lane its 16-lane row maximum; the 4 rows are combined on the
// scalar unit through v_readlane.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    v = max(v, dpp_u32<0xB1>(v));    // quad_perm [1,0,3,2]
    v = max(v, dpp_u32<0x4E>(v));    // quad_perm [2,3,0,1]
    v = max(v, dpp_u32<0x141>(v));   // row_half_mirror
    v = max(v, dpp_u32<0x140>(v));   // row_mirror
    unsigned a = __builtin_amdgcn_readlane(v, 0);
    unsigned b = __builtin_amdgcn_readlane(v, 16);
    unsigned c = __builtin_amdgcn_readlane(v, 32);
    unsigned d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}

__device__ __forceinline__ float wave_sum_f32(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
