// Inverted neighbour table (CSR) of an index tensor idx [B, SK] with values in [0, N): for every cloud,
// rowptr [B][N+1] and entries [B][SK] (entry = position in the cloud's idx row), rows in ascending
// order, entries of a row in no particular order (sort_rows: ascending, for rows of up to CSR_SORT_MAX entries).
// Shared by the difference-attention backward
// (diffattn.hip) and the upsample forward (gather.hip): both turn a scatter-add over idx into a gather.
#pragma once
#include "mpa_common.h"

namespace {

constexpr int CSR_MAX_N = 65536;       // rows per cloud; a workgroup's row range is <= max(256, N/64): <= 8 KB of LDS

// One workgroup per (cloud, range of base rows): it scans all S*K entries of the cloud, counts
// those of its rows in LDS (integer LDS atomics are as slow as the float ones, so the rows are
// split over CSR_RANGES workgroups per cloud), counts the entries below its range for the global
// offset, scans and fills its rows' lists.
constexpr int CSR_TPB = 256;
constexpr int CSR_SORT_MAX = 32;       // rows up to this many entries are sorted after the fill (see csr_build_body)

__device__ __forceinline__ int block_exclusive_scan(int v, int *wave_tot, int &total)
{
    // inclusive scan inside the wave (DPP-free shuffles; 4 waves)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    if (lane == 63) wave_tot[w] = x;
    __syncthreads();
    int base = 0;
    total = 0;
#pragma unroll
    for (int i = 0; i < CSR_TPB / 64; ++i) {
        if (i < w) base += wave_tot[i];
        total += wave_tot[i];
    }
    __syncthreads();
    return base + x - v;
}

// Per-cloud queue of "hub" rows (rows listed by very many entries): [B][CSR_QUEUE_INTS] ints = reserved count,
// head, CSR_QUEUE_MAX row slots.  Cleared here (optional), filled and drained by the consumer kernel.
constexpr int CSR_QUEUE_MAX = 254;
constexpr int CSR_QUEUE_INTS = CSR_QUEUE_MAX + 2;

// The workgroup body (bx = range index, by = cloud), shared by csr_build_kernel and by launches that carry the table
// build next to independent work (the attention backward's first pass, diffattn.hip).
__device__ __forceinline__ void csr_build_body(const int64_t *__restrict__ idx, int N, int SK, int range,
                                               int *__restrict__ rowptr, int *__restrict__ entries,
                                               int *__restrict__ queue, const int bx, const int by, int *csr_lds,
                                               const bool sort_rows = false)
{
    __shared__ int wave_tot[CSR_TPB / 64];
    constexpr int EPT = 32;                // entries per thread and chunk, all loads in flight at once
    int *cnt = csr_lds, *pos = csr_lds + range;          // cnt[range] | pos[range]
    const int b = by, tid = threadIdx.x;
    const int r0 = bx * range, r1 = min(N, r0 + range);
    const int64_t *nb = idx + (size_t)b * SK;
    int *rp = rowptr + (size_t)b * (N + 1);
    int *en = entries + (size_t)b * SK;
    for (int r = tid; r < range; r += CSR_TPB) cnt[r] = 0;
    if (queue != nullptr && bx == 0)
        for (int i = tid; i < CSR_QUEUE_INTS; i += CSR_TPB) queue[(size_t)b * CSR_QUEUE_INTS + i] = i < 2 ? 0 : -1;
    __syncthreads();
    int rr[EPT];
    const bool one_chunk = SK <= EPT * CSR_TPB;
    int below = 0;
    for (int base = 0; base < SK; base += EPT * CSR_TPB) {
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const int e = base + u * CSR_TPB + tid;
            rr[u] = e < SK ? (int)mpa_clamp_idx(nb[e], N) : 0x7fffffff;
        }
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            below += rr[u] < r0;
            if (rr[u] >= r0 && rr[u] < r1) atomicAdd(&cnt[rr[u] - r0], 1);
        }
    }
    int nbelow;
    block_exclusive_scan(below, wave_tot, nbelow);          // (syncs: cnt is complete afterwards)
    // exclusive scan of cnt: thread t owns the contiguous chunk [t*per, (t+1)*per) of the range
    const int per = (range + CSR_TPB - 1) / CSR_TPB;
    int local = 0;
    for (int r = tid * per; r < min(r1 - r0, (tid + 1) * per); ++r) local += cnt[r];
    int tot;
    int run = nbelow + block_exclusive_scan(local, wave_tot, tot);
    for (int r = tid * per; r < min(r1 - r0, (tid + 1) * per); ++r) {
        pos[r] = run;
        rp[r0 + r] = run;
        run += cnt[r];
    }
    if (r1 == N && tid == 0) rp[N] = SK;
    __syncthreads();
    // fill: entries of one row land in the order the LDS atomics retire (any order is a valid table)
    for (int base = 0; base < SK; base += EPT * CSR_TPB) {
        if (!one_chunk) {
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const int e = base + u * CSR_TPB + tid;
                rr[u] = e < SK ? (int)mpa_clamp_idx(nb[e], N) : 0x7fffffff;
            }
        }
#pragma unroll
        for (int u = 0; u < EPT; ++u)
            if (rr[u] >= r0 && rr[u] < r1) en[atomicAdd(&pos[rr[u] - r0], 1)] = base + u * CSR_TPB + tid;
    }
    // sort_rows (the upsample forward's tables): rows of up to CSR_SORT_MAX entries are put in ascending entry order -- the
    // consumer sums a row's entries in list order, so its sums no longer depend on the order the cursors' atomics
    // retired and the part-seg forward is reproducible run to run (tools/determinism_probe.py); longer rows -- hubs --
    // keep the order they got.  The attention backward's tables skip it: 0.5 % of a cls step, and its sums meet float
    // atomics further down anyway.
    if (!sort_rows) return;
    __syncthreads();
    for (int r = tid * per; r < min(r1 - r0, (tid + 1) * per); ++r) {
        const int n = cnt[r];
        if (n < 2 || n > CSR_SORT_MAX) continue;
        int *row = en + rp[r0 + r];
        for (int i = 1; i < n; ++i) {
            const int v = row[i];
            int j = i - 1;
            while (j >= 0 && row[j] > v) {
                row[j + 1] = row[j];
                --j;
            }
            row[j + 1] = v;
        }
    }
}

__global__ __launch_bounds__(CSR_TPB) void csr_build_kernel(const int64_t *__restrict__ idx, int N, int SK, int range,
                                                             int *__restrict__ rowptr, int *__restrict__ entries,
                                                             int *__restrict__ queue, int sort_rows)
{
    extern __shared__ int csr_dyn_lds[];
    csr_build_body(idx, N, SK, range, rowptr, entries, queue, blockIdx.x, blockIdx.y, csr_dyn_lds, sort_rows != 0);
}

// rows per workgroup of a build over B clouds of N rows: ~256 rows each, >= 256 workgroups in all
inline int csr_range(int B, int N)
{
    int ranges = 1;
    while (ranges < 64 && (N / ranges > 256 || B * ranges < 256) && N / (2 * ranges) >= 32) ranges <<= 1;
    return (N + ranges - 1) / ranges;
}

inline void launch_csr_build(const int64_t *idx, int B, int N, int SK, int *rowptr, int *entries, hipStream_t st,
                             int *queue = nullptr, bool sort_rows = false)
{
    const int range = csr_range(B, N);
    hipLaunchKernelGGL(csr_build_kernel, dim3(mpa_ceil_div(N, range), B), dim3(CSR_TPB),
                       (size_t)2 * range * sizeof(int), st, idx, N, SK, range, rowptr, entries, queue, sort_rows ? 1 : 0);
}

}  // namespace
