// Split-K reduction shared by the fp32 and bf16 grouped weight-gradient launches
// (linear.hip, linear_bf16.hip): partial fp32 slabs -> the real outputs.
#pragma once
#include "mpa_common.h"

namespace {

constexpr int GROUP_MAX = 40;

// out_p[M*N] += sum_z slab_p[z][M*N] for all problems of a group (out_p cleared by the z = 0 tiles above)
struct GroupedReduceArgs {
    int count;
    int block_start[GROUP_MAX + 1];      // prefix sum of ceil(mn/256)*gy
    struct { const float *slab; float *out; int mn, splits, gx, gy; } p[GROUP_MAX];
};

__global__ void splitk_reduce_grouped_kernel(const GroupedReduceArgs args)
{
    int b = blockIdx.x, i = 0;
    while (i + 1 < args.count && b >= args.block_start[i + 1]) ++i;
    const auto &q = args.p[i];
    const int local = b - args.block_start[i];
    const int bx = local % q.gx, by = local / q.gx;
    const long long e = bx * 256LL + threadIdx.x;
    if (e >= q.mn) return;
    const int zper = (q.splits + q.gy - 1) / q.gy;
    const int z0 = by * zper, z1 = min(q.splits, z0 + zper);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int z = z0;
    for (; z + 3 < z1; z += 4) {
        a0 += q.slab[(size_t)z * q.mn + e];
        a1 += q.slab[(size_t)(z + 1) * q.mn + e];
        a2 += q.slab[(size_t)(z + 2) * q.mn + e];
        a3 += q.slab[(size_t)(z + 3) * q.mn + e];
    }
    for (; z < z1; ++z) a0 += q.slab[(size_t)z * q.mn + e];
    if (z0 < z1) atomicAdd(q.out + e, (a0 + a1) + (a2 + a3));
}


}  // namespace
