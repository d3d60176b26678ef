// kNN / ball-query / square_distance for gfx950 -- replaces square_distance + topk
// (reference modules/pointnet2_utils.py:190-222) and query_ball_point (:112-134).
//
// The [B,S,N] distance matrix is never written.  One lane owns one query: its C channels and
// its sorted K-list live in registers; base rows stream through an LDS tile and are read as
// wave-wide broadcasts (every lane reads the same row), so LDS traffic is one row per 64
// queries.  The arithmetic is the reference CPU result bit for bit:
//   dot   : acc = q0*b0; acc = fmaf(qc, bc, acc) in channel order          (A1)
//   norms : separately rounded squares; C<8 sequential, C%8==0 the 8-lane x 4-accumulator
//           order of ATen's vectorised sum                                      (A2)
//   dist  : fl(fl(fl(-2*dot) + |q|^2) + |b|^2)                                 (A3)
// Ties keep the lower base index (strict < on an ascending scan), the order a stable sort gives.
#include "mpa_common.h"
#include "fps_body.h"
#include <cstdlib>
#include <type_traits>

namespace {

// A2: torch.sum(x**2, -1).  CT > 0: compile-time channel count (fully unrolled).
template <int CT, typename Ptr>
__device__ __forceinline__ float sum_sq_model(Ptr x, int C)
{
    const int c_ = CT > 0 ? CT : C;
    if (c_ < 8) {
        float s = x[0] * x[0];
        for (int c = 1; c < c_; ++c) {
            float sq = x[c] * x[c];
            s = s + sq;
        }
        return s;
    }
    const int nv = c_ / 8;
    const int A = nv < 4 ? nv : 4;
    float acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            float v = a < A ? x[a * 8 + l] : 0.f;
            acc[a][l] = v * v;
        }
    if (CT > 0) {
#pragma unroll
        for (int v = 4; v < (CT > 0 ? CT / 8 : 4); ++v) {
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                float e = x[v * 8 + l];
                float sq = e * e;
                acc[v & 3][l] = acc[v & 3][l] + sq;
            }
        }
    } else {
        for (int v = A; v < nv; ++v) {
            int a = v % A;
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                float e = x[v * 8 + l];
                float sq = e * e;
                // runtime accumulator select without dynamic register indexing
#pragma unroll
                for (int aa = 0; aa < 4; ++aa)
                    if (aa == a) acc[aa][l] = acc[aa][l] + sq;
            }
        }
    }
    float lane[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        float s = acc[0][l];
        if (A > 1) s = s + acc[1][l];
        if (A > 2) s = s + acc[2][l];
        if (A > 3) s = s + acc[3][l];
        lane[l] = s;
    }
    float s = lane[0];
#pragma unroll
    for (int l = 1; l < 8; ++l) s = s + lane[l];
    for (int c = nv * 8; c < c_; ++c) {
        float sq = x[c] * x[c];
        s = s + sq;
    }
    return s;
}

// A2 for an LDS row read as float4 (conflict-free 16-B row reads instead of CT strided scalar
// reads); same accumulation order as sum_sq_model for CT a multiple of 8.
template <int CT>
__device__ __forceinline__ float sum_sq_row4(const float *row)
{
    static_assert(CT % 8 == 0, "vector form of the A2 model");
    constexpr int NVEC = CT / 8;
    constexpr int A = NVEC < 4 ? NVEC : 4;
    float acc[4][8];
#pragma unroll
    for (int m = 0; m < CT / 4; ++m) {
        const float4 v = *reinterpret_cast<const float4 *>(row + 4 * m);
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = 4 * m + j, vv = idx / 8, l = idx % 8;
            const float sq = e[j] * e[j];
            if (vv < A) acc[vv][l] = sq;
            else acc[vv % A][l] = acc[vv % A][l] + sq;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        float t = acc[0][l];
        if (A > 1) t = t + acc[1][l];
        if (A > 2) t = t + acc[2][l];
        if (A > 3) t = t + acc[3][l];
        s = l == 0 ? t : s + t;
    }
    return s;
}

template <int CT>
__device__ __forceinline__ float tile_row_norm(const float *row)
{
    if constexpr (CT % 8 == 0) return sum_sq_row4<CT>(row);
    else return sum_sq_model<CT>(row, CT);
}

// LDS tile geometry: TB base rows of C floats, row stride C rounded up to a float4 multiple
// plus one float4 of padding (keeps 16-B alignment, spreads the norm pass over banks).
__host__ __device__ inline int tile_ld(int C) { return ((C + 3) / 4) * 4 + 4; }

template <int CT>
__device__ __forceinline__ void stage_tile(const float *__restrict__ base, int n0, int rows, int C, int LD,
                                           float *tile, float *tnorm, int tid, int nthreads)
{
    const int c_ = CT > 0 ? CT : C;
    // rows n0..n0+rows-1 are contiguous in global memory
    const float *src = base + (size_t)n0 * c_;
    if ((c_ & 3) == 0) {
        const int c4 = c_ >> 2;
        for (int i = tid; i < rows * c4; i += nthreads) {
            int r = i / c4, c = i - r * c4;
            float4 v = reinterpret_cast<const float4 *>(src)[i];
            *reinterpret_cast<float4 *>(tile + r * LD + 4 * c) = v;
        }
    } else {
        for (int i = tid; i < rows * c_; i += nthreads) {
            int r = i / c_, c = i - r * c_;
            tile[r * LD + c] = src[i];
        }
    }
    __syncthreads();
    for (int r = tid; r < rows; r += nthreads) tnorm[r] = sum_sq_model<CT>(tile + r * LD, c_);
    __syncthreads();
}

template <int CT>
__device__ __forceinline__ float dot_chain(const float *q, const float *row, int C)
{
    const int c_ = CT > 0 ? CT : C;
    float acc;
    if (CT > 0 && (CT & 3) == 0) {
        float4 r0 = *reinterpret_cast<const float4 *>(row);
        acc = q[0] * r0.x;
        acc = fmaf(q[1], r0.y, acc);
        acc = fmaf(q[2], r0.z, acc);
        acc = fmaf(q[3], r0.w, acc);
#pragma unroll
        for (int c = 4; c < CT; c += 4) {
            float4 r = *reinterpret_cast<const float4 *>(row + c);
            acc = fmaf(q[c], r.x, acc);
            acc = fmaf(q[c + 1], r.y, acc);
            acc = fmaf(q[c + 2], r.z, acc);
            acc = fmaf(q[c + 3], r.w, acc);
        }
    } else {
        acc = q[0] * row[0];
#pragma unroll
        for (int c = 1; c < (CT > 0 ? CT : 1); ++c) acc = fmaf(q[c], row[c], acc);
        if (CT == 0)
            for (int c = 1; c < c_; ++c) acc = fmaf(q[c], row[c], acc);
    }
    return acc;
}

__device__ __forceinline__ float finish_dist(float dot, float qn, float bn)
{
    float d = -2.0f * dot;
    d = d + qn;
    d = d + bn;
    return d;
}

constexpr int TQ = 64;   // queries per workgroup (one wave)

// CT = compile-time C (query row in registers) ; CT == 0: runtime C, query row in LDS.
template <int CT, int KMAX>
__global__ __launch_bounds__(TQ) void knn_kernel(const float *__restrict__ base, const float *__restrict__ query,
                                                 int N, int S, int C, int K, int TB,
                                                 float *__restrict__ out_dist, int64_t *__restrict__ out_idx)
{
    extern __shared__ float lds[];
    const int c_ = CT > 0 ? CT : C;
    const int LD = tile_ld(c_);
    float *tile = lds;                 // [TB][LD]
    float *tnorm = tile + TB * LD;     // [TB]
    float *qlds = tnorm + TB;          // [TQ][LD]   (CT == 0 only)

    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int s = blockIdx.x * TQ + tid;
    const bool live = s < S;
    const float *bp = base + (size_t)b * N * c_;
    const float *qp = query + ((size_t)b * S + (live ? s : 0)) * c_;

    float qreg[CT > 0 ? CT : 1];
    const float *q;
    if (CT > 0) {
#pragma unroll
        for (int c = 0; c < CT; ++c) qreg[c] = qp[c];
        q = qreg;
    } else {
        float *mine = qlds + tid * LD;
        for (int c = 0; c < c_; ++c) mine[c] = qp[c];
        q = mine;
    }
    const float qn = sum_sq_model<CT>(q, c_);

    float bd[KMAX];
    int bi[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { bd[k] = INFINITY; bi[k] = 0; }   // NaN inputs: in-range garbage, never -1

    for (int n0 = 0; n0 < N; n0 += TB) {
        const int rows = min(TB, N - n0);
        __syncthreads();
        stage_tile<CT>(bp, n0, rows, c_, LD, tile, tnorm, tid, TQ);
        for (int t = 0; t < rows; ++t) {
            float d = finish_dist(dot_chain<CT>(q, tile + t * LD, c_), qn, tnorm[t]);
            if (d < bd[KMAX - 1]) {
                int n = n0 + t;
#pragma unroll
                for (int k = KMAX - 1; k > 0; --k) {
                    bool up = d < bd[k - 1];
                    bool here = d < bd[k];
                    bd[k] = up ? bd[k - 1] : (here ? d : bd[k]);
                    bi[k] = up ? bi[k - 1] : (here ? n : bi[k]);
                }
                bool h0 = d < bd[0];
                bd[0] = h0 ? d : bd[0];
                bi[0] = h0 ? n : bi[0];
            }
        }
    }
    if (live) {
        size_t o = ((size_t)b * S + s) * K;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) {
                out_idx[o + k] = bi[k];
                if (out_dist) out_dist[o + k] = bd[k];
            }
    }
}

template <int CT>
__global__ __launch_bounds__(TQ) void ball_kernel(const float *__restrict__ base, const float *__restrict__ query,
                                                  int N, int S, int C, float r2, int nsample, int TB,
                                                  int64_t *__restrict__ out_idx)
{
    extern __shared__ float lds[];
    const int c_ = CT > 0 ? CT : C;
    const int LD = tile_ld(c_);
    float *tile = lds;
    float *tnorm = tile + TB * LD;
    float *qlds = tnorm + TB;

    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int s = blockIdx.x * TQ + tid;
    const bool live = s < S;
    const float *bp = base + (size_t)b * N * c_;
    const float *qp = query + ((size_t)b * S + (live ? s : 0)) * c_;
    float qreg[CT > 0 ? CT : 1];
    const float *q;
    if (CT > 0) {
#pragma unroll
        for (int c = 0; c < CT; ++c) qreg[c] = qp[c];
        q = qreg;
    } else {
        float *mine = qlds + tid * LD;
        for (int c = 0; c < c_; ++c) mine[c] = qp[c];
        q = mine;
    }
    const float qn = sum_sq_model<CT>(q, c_);
    int64_t *o = out_idx + ((size_t)b * S + (live ? s : 0)) * nsample;
    int cnt = 0;
    int firsthit = N;
    for (int n0 = 0; n0 < N; n0 += TB) {
        const int rows = min(TB, N - n0);
        __syncthreads();
        stage_tile<CT>(bp, n0, rows, c_, LD, tile, tnorm, tid, TQ);
        // wave-uniform early exit once every live lane has its nsample hits
        if (__ballot(live && cnt < nsample) == 0ull) continue;
        for (int t = 0; t < rows; ++t) {
            float d = finish_dist(dot_chain<CT>(q, tile + t * LD, c_), qn, tnorm[t]);
            if (live && cnt < nsample && !(d > r2)) {
                if (cnt == 0) firsthit = n0 + t;
                o[cnt++] = n0 + t;
            }
        }
    }
    if (live)
        for (int k = cnt; k < nsample; ++k) o[k] = firsthit;
}

// square_distance (API completeness; the fused path never materialises it): one lane per
// destination point so the [S,N] rows are written coalesced; the source row is wave-uniform.
__global__ void sqdist_kernel(const float *__restrict__ src, const float *__restrict__ dst, int S, int N, int C,
                              float *__restrict__ out)
{
    const int b = blockIdx.z;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int s0 = blockIdx.y * 16;
    if (n >= N) return;
    const float *d = dst + ((size_t)b * N + n) * C;
    const float dn = sum_sq_model<0>(d, C);
    for (int s = s0; s < min(S, s0 + 16); ++s) {
        const float *q = src + ((size_t)b * S + s) * C;
        const float qn = sum_sq_model<0>(q, C);
        float acc = q[0] * d[0];
        for (int c = 1; c < C; ++c) acc = fmaf(q[c], d[c], acc);
        out[((size_t)b * S + s) * N + n] = finish_dist(acc, qn, dn);
    }
}

// ------------------------------------------------------------------------------------------
// MFMA kNN: the distance matrix tile by tile on the matrix cores, exact selection by
// "bound, then filter".
//
// v_mfma_f32_32x32x2_f32 is bit-equal to a k-ordered fmaf chain (acc = fma(a_k, b_k, acc), one
// rounding per product), which is exactly the reference's bmm dot (A1), so a 32x32 tile of dot
// products comes out of C/2 MFMAs bit-identical to the scalar kernel above.  Operand orientation
// puts the QUERY on the lane: A = 32 base rows, B = 32 queries, so lane l holds, for query
// (l & 31), the 16 dots against base rows (r&3) + 8*(r>>2) + 4*(l>>5) of the tile.
//
// A workgroup owns 32 queries; its WAVES waves take the base tiles round-robin, each staging its
// tile in a private LDS slab (no workgroup barrier inside a pass).  Keeping a sorted K-list per
// lane costs ~6*K VALU ops per hit and SIMD execution pays it whenever ANY of 64 lanes hits, which
// made selection 3x more expensive than the MFMAs.  Instead the distances are produced twice
// (the matrix cores are otherwise idle):
//   pass A  folds every candidate into one of 32 disjoint groups per query and keeps the group
//           minima (4 v_min + one LDS atomic-min per 4 candidates).  The K-th smallest group
//           minimum tau is an upper bound of the true K-th distance (the K smallest minima are K
//           distinct points) -- and a tight one: ~1.2*K candidates pass it on random data.
//   pass B  recomputes the tiles and appends the few candidates with d <= tau to a per-query LDS
//           list (one compare per candidate, no sorting), which is then ranked by (distance,
//           index): the order a stable ascending sort gives, i.e. the scalar kernel's tie rule.
// If a list overflows (massive ties, e.g. duplicated points) the workgroup falls back to per-lane
// sorted lists, merged by the same ranking.
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int KMAX>
__device__ __forceinline__ void list_insert(float (&bd)[KMAX], int (&bi)[KMAX], float d, int n)
{
#pragma unroll
    for (int k = KMAX - 1; k > 0; --k) {
        bool up = d < bd[k - 1];
        bool here = d < bd[k];
        bd[k] = up ? bd[k - 1] : (here ? d : bd[k]);
        bi[k] = up ? bi[k - 1] : (here ? n : bi[k]);
    }
    bool h0 = d < bd[0];
    bd[0] = h0 ? d : bd[0];
    bi[0] = h0 ? n : bi[0];
}

// monotone float -> int key (handles the slightly negative distances the A3 rounding can give)
__device__ __forceinline__ int f2key(float d)
{
    int b = __float_as_int(d);
    return b >= 0 ? b : b ^ 0x7fffffff;
}
__device__ __forceinline__ bool lex_less(float d1, int n1, float d2, int n2)
{
    return d1 < d2 || (d1 == d2 && n1 < n2);
}

constexpr int KNN_CAP = 32;      // candidates kept per query in pass B
constexpr int KNN_G = 32;        // groups per query in pass A

// QG: 32-query groups per workgroup.  With QG = 2 every staged base tile (LDS rows, norms) feeds two independent MFMA
// chains: half the staging / norm / LDS-read work per product, and a tile's MFMA phase (2 x C/2 instructions) is long
// enough to cover the next tile's global load -- with one group the load of the following tile (an L2 hit, 0.6-0.9 us)
// is as long as the 0.85 us of MFMAs it hides behind at C = 64, and the matrix cores idle about half the time.
// RES (C = 3 searches, N <= KNN_RES_MAX): the whole base cloud and its norms are staged ONCE per workgroup
// ([N][4] floats + [N] norms, <= 20 bytes per point) and both passes run out of LDS.  A C = 3 tile is two MFMAs: with
// per-tile staging every visit costs a global-load latency (~1 us for 128 cycles of MFMA); resident, a visit is the
// two MFMAs plus the selection arithmetic.
constexpr int KNN_RES_MAX = 4096;

template <int CT, int WAVES, int KMAX, int QG = 1, bool RES = false, bool GN = false>
__device__ __forceinline__ void knn_mfma_body(const float *__restrict__ base, const float *__restrict__ query, int N,
                                              int S, int K, float *__restrict__ out_dist,
                                              int64_t *__restrict__ out_idx, const int bx, const int by, float *lds,
                                              const float *__restrict__ bnorm = nullptr)
{
    // GN: bnorm holds the base rows' squared norms [B][ceil32(N)] from mpa_row_norms_f32 (+inf past N).  Without it
    // every workgroup recomputes the norms of every tile in both passes: ~140 dependent VALU instructions on 32 of a
    // wave's lanes per tile, a quarter to a half of the tile's MFMA time at C = 64.  (A template parameter, not a
    // run-time test: with both forms alive the two-group kernel needs 280 registers and loses its second wave per SIMD.)
    static_assert(!(GN && RES), "resident clouds compute their norms once per workgroup");
    constexpr int CP = (CT + 3) & ~3;            // channels padded to a float4 multiple (C=3 -> 4)
    constexpr int PITCH = CP + 4;                // LDS row pitch: (PITCH/4) odd -> conflict-free b128 rows
    constexpr int NV = 32 * CP / 4 / 64;         // float4 staged per lane per tile (CP >= 8)
    constexpr int NT_ = WAVES * 64;
    constexpr int NQ = 32 * QG;                  // queries per workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    static_assert(!RES || CT == 3, "resident clouds: coordinate searches only");
    float *slab = lds + wave * (32 * PITCH + 32);          // [32][PITCH] tile + [32] norms
    float *snorm = slab + 32 * PITCH;
    int *gmin = reinterpret_cast<int *>(lds + WAVES * (32 * PITCH + 32));   // [KNN_G][NQ] keys
    float *tau = reinterpret_cast<float *>(gmin + KNN_G * NQ);              // [NQ]
    int *cnt = reinterpret_cast<int *>(tau + NQ);                           // [NQ] (+ overflow flag at [NQ])
    float *cand_d = reinterpret_cast<float *>(cnt + NQ + 32);               // [NQ][KNN_CAP]
    int *cand_i = reinterpret_cast<int *>(cand_d + NQ * KNN_CAP);           // [NQ][KNN_CAP]

    const int b = by;
    const int q0 = bx * NQ;
    const float *bp = base + (size_t)b * N * CT;

    // query operands: lane (j, half) holds q[j][2kk + half] of every group; |q|^2 by the A2 model
    float qv[QG][CP / 2];
    float qn[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const int qrow_i = min(q0 + g * 32 + l31, S - 1);
        const float *qp = query + ((size_t)b * S + qrow_i) * CT;
#pragma unroll
        for (int kk = 0; kk < CP / 2; ++kk) qv[g][kk] = (2 * kk + half < CT) ? qp[2 * kk + half] : 0.f;
        qn[g] = sum_sq_model<CT>(qp, CT);
    }

    const int ntiles = (N + 31) / 32;
    float *cloud = reinterpret_cast<float *>(cand_i + NQ * KNN_CAP);        // RES: [ntiles*32][4] | norms [ntiles*32]
    float *cnorm = cloud + (size_t)ntiles * 32 * 4;
    if constexpr (RES) {
        // coalesced read of the packed [N][3] rows, 8 loads in flight per lane (one at a time, the 24 rounds of a
        // 2048-point cloud are 24 global round trips: longer than both passes over the resident cloud)
        const int total = ntiles * 32 * 3;
        for (int j0 = tid; j0 < total; j0 += 8 * NT_) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u * NT_;
                v[u] = j < 3 * N ? bp[j] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u * NT_;
                if (j < total) cloud[j + j / 3] = v[u];             // 4 * (j / 3) + j % 3
            }
        }
        for (int r = tid; r < ntiles * 32; r += NT_) cloud[4 * r + 3] = 0.f;
    }
    for (int i = tid; i < KNN_G * NQ; i += NT_) gmin[i] = 0x7f800000;      // +inf key
    for (int i = tid; i < NQ + 1; i += NT_) cnt[i] = 0;
    __syncthreads();
    if constexpr (RES) {
        for (int r = tid; r < ntiles * 32; r += NT_) cnorm[r] = r < N ? tile_row_norm<CT>(cloud + 4 * r) : INFINITY;
        __syncthreads();
    }

    float4 stg[CP >= 8 ? NV : 1];
    float stg3[2];
    constexpr bool PREFETCH = CT <= 64;          // wide rows: the staging registers are needed for qv
    auto load_tile = [&](int t) {
        if (CT == 3) {
            const float *src = bp + (size_t)t * 32 * 3;
            const int lim = (min(32, N - t * 32)) * 3;
            stg3[0] = lane < lim ? src[lane] : 0.f;
            stg3[1] = lane + 64 < lim ? src[lane + 64] : 0.f;
        } else {
            const float4 *src = reinterpret_cast<const float4 *>(bp + (size_t)t * 32 * CT);
            const int lim = min(32, N - t * 32) * (CP / 4);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int i = lane + 64 * v;
                stg[v] = i < lim ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto store_tile = [&]() {
        if (CT == 3) {
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int i = lane + 64 * v;
                if (i < 96) slab[(i / 3) * PITCH + (i % 3)] = stg3[v];
            }
            if (lane < 32) slab[lane * PITCH + 3] = 0.f;
        } else {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int i = lane + 64 * v;
                const int r = i / (CP / 4), c4 = i - r * (CP / 4);
                *reinterpret_cast<float4 *>(slab + r * PITCH + 4 * c4) = stg[v];
            }
        }
    };
    auto copy_tile = [&](int t) {                // global -> LDS without holding the whole tile
        const float4 *src = reinterpret_cast<const float4 *>(bp + (size_t)t * 32 * CT);
        const int lim = min(32, N - t * 32) * (CP / 4);
        // 8 loads in flight per lane (4 made a C = 256 tile 8 dependent global round trips: the coarsest states'
        // searches are one or two workgroups per cloud and see every one of them)
#pragma unroll
        for (int v0 = 0; v0 < NV; v0 += 8) {
            float4 tmp[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = lane + 64 * (v0 + u);
                tmp[u] = (v0 + u < NV && i < lim) ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = lane + 64 * (v0 + u);
                const int r = i / (CP / 4), c4 = i - r * (CP / 4);
                if (v0 + u < NV) *reinterpret_cast<float4 *>(slab + r * PITCH + 4 * c4) = tmp[u];
            }
        }
    };
    // one tile: stage, norms (A2 model; rows beyond N get +inf so they never qualify), QG interleaved MFMA chains
    floatx16 acc[QG];
    float4 sn[4];
    auto tile_dots = [&](int t, bool first) {
        const float *row;
        if constexpr (RES) {
            row = cloud + (t * 32 + l31) * 4;
            snorm = cnorm + t * 32;
        } else {
            if (PREFETCH) {
                if (first) load_tile(t);
                store_tile();
                if (t + WAVES < ntiles) load_tile(t + WAVES);      // in flight during this tile's MFMAs
            } else {
                copy_tile(t);
            }
            __builtin_amdgcn_wave_barrier();
            if constexpr (!GN) {
                if (lane < 32) snorm[lane] = (t * 32 + lane < N) ? tile_row_norm<CT>(slab + lane * PITCH) : INFINITY;
            }
            row = slab + l31 * PITCH;
        }
        // this lane's 16 base-row norms (rows (r & 3) + 8 (r >> 2) + 4 half of the tile) as four 16-byte reads issued
        // BEFORE the MFMA chain: read one by one where they are used, every distance waited for its own LDS round
        // trip (16 per tile and pass, ~100 clocks each -- more than the MFMAs of a C = 64 tile)
        __builtin_amdgcn_wave_barrier();
        if constexpr (GN) {
            const float *gn = bnorm + (size_t)b * ntiles * 32 + t * 32 + 4 * half;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) sn[q4] = *reinterpret_cast<const float4 *>(gn + 8 * q4);
        } else {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) sn[q4] = *reinterpret_cast<const float4 *>(snorm + 8 * q4 + 4 * half);
        }
#pragma unroll
        for (int g = 0; g < QG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
#pragma unroll
        for (int m = 0; m < CP / 4; ++m) {
            const float4 a = *reinterpret_cast<const float4 *>(row + 4 * m);
            const float a0 = half ? a.y : a.x, a1 = half ? a.w : a.z;
#pragma unroll
            for (int g = 0; g < QG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, qv[g][2 * m], acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < QG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, qv[g][2 * m + 1], acc[g], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    };
#define KNN_DIST(g, r) ((-2.0f * acc[g][r] + qn[g]) + (&sn[(r) >> 2].x)[(r) & 3])

    bool fast = (ntiles * 8 >= K) && K <= KNN_CAP;
    if (fast) {
        // ---------------- pass A: group minima.  Group g = (8t + 2qd + half) & 31 of a query: with
        // tiles dealt t = wave, wave + WAVES, ... every group belongs to exactly one (wave, lane, qd,
        // t & 3), so the minima live in registers (LDS atomics retire ~1 lane per 2.5 clocks: 640
        // cycles per tile, against 2048 MFMA cycles at C = 64) and are stored once at the end.
        constexpr int TPH = 4 / WAVES;                      // distinct t & 3 values a wave sees
        float gm[QG][TPH][4];
#pragma unroll
        for (int g = 0; g < QG; ++g)
#pragma unroll
            for (int a = 0; a < TPH; ++a)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) gm[g][a][qd] = INFINITY;
        for (int t0 = wave, it = 0; t0 < ntiles; t0 += 4) {
#pragma unroll
            for (int a = 0; a < TPH; ++a) {
                const int t = t0 + a * WAVES;               // t & 3 == (wave + a * WAVES) & 3
                if (t < ntiles) {
                    tile_dots(t, it == 0);
                    ++it;
#pragma unroll
                    for (int g = 0; g < QG; ++g)
#pragma unroll
                        for (int qd = 0; qd < 4; ++qd) {
                            const float m = fminf(fminf(KNN_DIST(g, 4 * qd), KNN_DIST(g, 4 * qd + 1)),
                                                  fminf(KNN_DIST(g, 4 * qd + 2), KNN_DIST(g, 4 * qd + 3)));
                            gm[g][a][qd] = fminf(gm[g][a][qd], m);
                        }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
#pragma unroll
        for (int g = 0; g < QG; ++g)
#pragma unroll
            for (int a = 0; a < TPH; ++a)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int grp = (((wave + a * WAVES) & 3) * 8 + qd * 2 + half) & (KNN_G - 1);
                    gmin[grp * NQ + g * 32 + l31] = f2key(gm[g][a][qd]);
                }
        __syncthreads();
        // tau[q] = K-th smallest group minimum (rank by (key, group))
        for (int item = tid; item < NQ * KNN_G; item += NT_) {
            const int q = item % NQ, grp = item / NQ;
            const int v = gmin[grp * NQ + q];
            int rank = 0;
            for (int o = 0; o < KNN_G; ++o) {
                const int u = gmin[o * NQ + q];
                rank += (u < v || (u == v && o < grp)) ? 1 : 0;
            }
            if (rank == K - 1) tau[q] = __int_as_float(v >= 0 ? v : v ^ 0x7fffffff);
        }
        __syncthreads();
        // ---------------- pass B: collect d <= tau
        float tq[QG];
#pragma unroll
        for (int g = 0; g < QG; ++g) tq[g] = tau[g * 32 + l31];
        for (int t = wave, it = 0; t < ntiles; t += WAVES, ++it) {
            tile_dots(t, it == 0);
#pragma unroll
            for (int g = 0; g < QG; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d = KNN_DIST(g, r);
                    const int n = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (d <= tq[g] && n < N) {
                        const int ql = g * 32 + l31;
                        const int slot = atomicAdd(cnt + ql, 1);
                        if (slot < KNN_CAP) {
                            cand_d[ql * KNN_CAP + slot] = d;
                            cand_i[ql * KNN_CAP + slot] = n;
                        } else {
                            cnt[NQ] = 1;                 // overflow: redo this workgroup the slow way
                        }
                    }
                }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        fast = cnt[NQ] == 0;
        if (fast) {
            for (int item = tid; item < NQ * KNN_CAP; item += NT_) {
                const int q = item / KNN_CAP, sl = item - q * KNN_CAP;
                const int n_q = cnt[q];
                if (sl < n_q && q0 + q < S) {
                    const float d = cand_d[q * KNN_CAP + sl];
                    const int n = cand_i[q * KNN_CAP + sl];
                    int rank = 0;
                    for (int o = 0; o < n_q; ++o)
                        rank += lex_less(cand_d[q * KNN_CAP + o], cand_i[q * KNN_CAP + o], d, n) ? 1 : 0;
                    if (rank < K) {
                        const size_t o_ = ((size_t)b * S + q0 + q) * K + rank;
                        out_idx[o_] = n;
                        if (out_dist) out_dist[o_] = d;
                    }
                }
            }
            return;
        }
        __syncthreads();
    }

    // ---------------- slow path: per-lane sorted lists (two half-lists per query and wave), one query group at a
    // time (the tiles are recomputed per group: rare -- massive ties or K beyond the candidate lists)
    constexpr int NL = 2 * WAVES;
    float *cd = lds;                                              // [32][NL][KMAX]
    int *ci = reinterpret_cast<int *>(lds + 32 * NL * KMAX);
    const int list = wave * 2 + half;
    auto slow_group = [&](auto gc) {
        constexpr int g = decltype(gc)::value;
        float bd[KMAX];
        int bi[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { bd[k] = INFINITY; bi[k] = 0; }
        for (int t = wave, it = 0; t < ntiles; t += WAVES, ++it) {
            tile_dots(t, it == 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = KNN_DIST(g, r);
                if (d < bd[KMAX - 1]) list_insert<KMAX>(bd, bi, d, t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half);
            }
            __builtin_amdgcn_wave_barrier();
        }
        // merge the 2*WAVES sorted partial lists of every query: every lane ranks its own elements in
        // the union (own position + lexicographically smaller pairs in each other list)
        __syncthreads();
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            cd[(l31 * NL + list) * KMAX + k] = bd[k];
            ci[(l31 * NL + list) * KMAX + k] = bi[k];
        }
        __syncthreads();
        if (q0 + g * 32 + l31 < S) {
            const size_t o = ((size_t)b * S + q0 + g * 32 + l31) * K;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const float d = bd[k];
                const int n = bi[k];
                int rank = k;
                for (int ol = 0; ol < NL; ++ol) {
                    if (ol == list) continue;
                    for (int e = 0; e < KMAX; ++e)
                        rank += lex_less(cd[(l31 * NL + ol) * KMAX + e], ci[(l31 * NL + ol) * KMAX + e], d, n) ? 1 : 0;
                }
                if (rank < K && d < INFINITY) {
                    out_idx[o + rank] = n;
                    if (out_dist) out_dist[o + rank] = d;
                }
            }
        }
        __syncthreads();                                          // the lists alias the slabs of the next group's tiles
    };
    slow_group(std::integral_constant<int, 0>{});
    if constexpr (QG > 1) slow_group(std::integral_constant<int, 1>{});
    if constexpr (QG > 2) {
        slow_group(std::integral_constant<int, 2>{});
        slow_group(std::integral_constant<int, 3>{});
    }
    static_assert(QG == 1 || QG == 2 || QG == 4, "query groups per workgroup");
#undef KNN_DIST
}

template <int CT, int WAVES, int KMAX, int QG, bool RES, bool GN>
__global__ __launch_bounds__(WAVES * 64) void knn_mfma_kernel(const float *__restrict__ base,
                                                              const float *__restrict__ query, int N, int S, int K,
                                                              float *__restrict__ out_dist,
                                                              int64_t *__restrict__ out_idx,
                                                              const float *__restrict__ bnorm)
{
    extern __shared__ float lds[];
    knn_mfma_body<CT, WAVES, KMAX, QG, RES, GN>(base, query, N, S, K, out_dist, out_idx, blockIdx.x, blockIdx.y, lds, bnorm);
}

// Squared norms of rows [B][N][C] by the A2 rounding model -> out [B][ceil32(N)], +inf in the padding: computed once
// per search instead of once per (query workgroup, pass, tile).
template <int CT>
__global__ __launch_bounds__(256) void row_norms_kernel(const float *__restrict__ x, int N, int npad, int C,
                                                        float *__restrict__ out)
{
    const int b = blockIdx.y;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= npad) return;
    const int c_ = CT > 0 ? CT : C;
    out[(size_t)b * npad + r] = r < N ? sum_sq_model<CT>(x + ((size_t)b * N + r) * c_, c_) : INFINITY;
}

// Farthest point sampling of one point-set state and the xyz-space kNN of the state before it in ONE
// launch: FPS keeps 1 workgroup per cloud busy for S serial iterations (64 of 256 CUs at batch 64),
// the kNN's (S/32)*B workgroups fill the rest of the chip meanwhile.  Both read coordinates only
// and are independent of each other: workgroups [0, B) sample, the others search.
template <int P>
__global__ __launch_bounds__(256) void fps_knn3_kernel(const float *__restrict__ fxyz, int fN, int fS,
                                                       const int64_t *__restrict__ start,
                                                       int64_t *__restrict__ f_idx, float *__restrict__ f_out_xyz,
                                                       int B, const float *__restrict__ base,
                                                       const float *__restrict__ query, int N, int S, int K,
                                                       float *__restrict__ out_dist, int64_t *__restrict__ out_idx)
{
    extern __shared__ float lds[];
    if ((int)blockIdx.x < B) {
        fps_body<4, P>(fxyz, fN, fS, start, f_idx, f_out_xyz, blockIdx.x, lds);
    } else {
        const int r = blockIdx.x - B, qb = (S + 31) / 32;
        knn_mfma_body<3, 4, 8, 1, false, false>(base, query, N, S, K, out_dist, out_idx, r % qb, r / qb, lds);
    }
}

template <int CT, int WAVES, int KMAX, int QG, bool RES = false, bool GN = false>
int launch_knn_mfma(const float *base, const float *query, int B, int N, int S, int K, float *od, int64_t *oi,
                    hipStream_t st, const float *bnorm = nullptr)
{
    constexpr int CP = (CT + 3) & ~3;
    constexpr int NQ = 32 * QG;
    constexpr size_t work = ((size_t)WAVES * (32 * (CP + 4) + 32) + KNN_G * NQ + NQ + NQ + 32 + 2 * NQ * KNN_CAP) *
                            sizeof(float);
    constexpr size_t merge = (size_t)32 * 2 * WAVES * KMAX * 8;
    static_assert(!RES || merge <= work, "the slow path's lists must not reach the resident cloud");
    constexpr size_t fixed = work > merge ? work : merge;
    constexpr size_t most = fixed + (RES ? (size_t)KNN_RES_MAX * 5 * sizeof(float) : 0);
    static_assert(most <= 160 * 1024, "LDS of a gfx950 CU");
    if (RES && N > KNN_RES_MAX) return MPA_EINVAL;
    const size_t lds = fixed + (RES ? (size_t)mpa_ceil_div(N, 32) * 32 * 5 * sizeof(float) : 0);
    if (most > 64 * 1024) {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&knn_mfma_kernel<CT, WAVES, KMAX, QG, RES, GN>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)most);
        if (attr != hipSuccess) return MPA_EHIP;
    }
    dim3 grid(mpa_ceil_div(S, NQ), B);
    hipLaunchKernelGGL((knn_mfma_kernel<CT, WAVES, KMAX, QG, RES, GN>), grid, dim3(WAVES * 64), lds, st, base, query, N, S, K,
                       od, oi, bnorm);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

// QGMAX: the widest query grouping the channel count leaves registers for; used when the launch still has >= 4
// workgroups per CU with it (the fine states), otherwise one group per workgroup keeps more waves in flight.
template <int CT, int WAVES, int QGMAX = 1>
int launch_knn_mfma_k(const float *base, const float *query, int B, int N, int S, int K, float *od, int64_t *oi,
                      hipStream_t st, const float *bnorm = nullptr)
{
    if constexpr (CT == 3) {
        static const bool no_res = getenv("MPA_KNN_NO_RESIDENT") != nullptr;
        if (K <= 8 && N <= KNN_RES_MAX && !no_res)
            return launch_knn_mfma<CT, WAVES, 8, 1, true>(base, query, B, N, S, K, od, oi, st);
    } else {
        // with the base norms given (K <= 8: every search of the models): QGMAX query groups per workgroup when the
        // launch still has >= 2 workgroups per CU with them (the fine states), one group otherwise
        if (bnorm != nullptr && K <= 8) {
            static const int force_qg = getenv("MPA_KNN_QG") ? atoi(getenv("MPA_KNN_QG")) : 0;
            const long long wgs1 = (long long)mpa_ceil_div(S, 32) * B;
            bool wide = QGMAX > 1 && wgs1 >= 1024LL * QGMAX / 2;
            if (force_qg) wide = QGMAX > 1 && force_qg > 1;
            if constexpr (QGMAX > 1) {
                if (wide) return launch_knn_mfma<CT, WAVES, 8, QGMAX, false, true>(base, query, B, N, S, K, od, oi, st, bnorm);
            }
            return launch_knn_mfma<CT, WAVES, 8, 1, false, true>(base, query, B, N, S, K, od, oi, st, bnorm);
        }
    }
    if (K <= 8) return launch_knn_mfma<CT, WAVES, 8, 1>(base, query, B, N, S, K, od, oi, st);
    return launch_knn_mfma<CT, WAVES, 32, 1>(base, query, B, N, S, K, od, oi, st);
}

int pick_tb(int C, int N, bool query_in_lds)
{
    int LD = tile_ld(C);
    long long budget = 48 * 1024;                       // keep the base tile under 48 KiB ...
    if (query_in_lds) {                                 // ... unless the query rows already eat the LDS
        long long left = 160 * 1024 - (long long)TQ * LD * 4 - 1024;
        if (left < budget) budget = left;
    }
    long long tb = budget / 4 / (LD + 1);
    tb = tb > 256 ? 256 : tb;
    if (tb > N) tb = N;
    return (int)tb;
}

template <int CT, int KMAX>
int launch_knn(const float *base, const float *query, int B, int N, int S, int C, int K, float *od, int64_t *oi,
               hipStream_t st)
{
    int TB = pick_tb(C, N, CT == 0);
    if (TB < 1) return MPA_EUNSUPPORTED;
    int LD = tile_ld(C);
    size_t lds = ((size_t)TB * LD + TB + (CT == 0 ? (size_t)TQ * LD : 0)) * sizeof(float);
    if (lds > 160 * 1024) return MPA_EUNSUPPORTED;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(&knn_kernel<CT, KMAX>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return MPA_EHIP;
    dim3 grid(mpa_ceil_div(S, TQ), B);
    hipLaunchKernelGGL((knn_kernel<CT, KMAX>), grid, dim3(TQ), lds, st, base, query, N, S, C, K, TB, od, oi);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

template <int CT>
int launch_knn_k(const float *base, const float *query, int B, int N, int S, int C, int K, float *od, int64_t *oi,
                 hipStream_t st)
{
    if (K <= 8) return launch_knn<CT, 8>(base, query, B, N, S, C, K, od, oi, st);
    if (K <= 16) return launch_knn<CT, 16>(base, query, B, N, S, C, K, od, oi, st);
    return launch_knn<CT, 32>(base, query, B, N, S, C, K, od, oi, st);
}

template <int CT>
int launch_ball(const float *base, const float *query, int B, int N, int S, int C, float r2, int ns, int64_t *oi,
                hipStream_t st)
{
    int TB = pick_tb(C, N, CT == 0);
    if (TB < 1) return MPA_EUNSUPPORTED;
    int LD = tile_ld(C);
    size_t lds = ((size_t)TB * LD + TB + (CT == 0 ? (size_t)TQ * LD : 0)) * sizeof(float);
    if (lds > 160 * 1024) return MPA_EUNSUPPORTED;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(&ball_kernel<CT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return MPA_EHIP;
    dim3 grid(mpa_ceil_div(S, TQ), B);
    hipLaunchKernelGGL((ball_kernel<CT>), grid, dim3(TQ), lds, st, base, query, N, S, C, r2, ns, TB, oi);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

}  // namespace

#ifndef MPA_KNN_BODIES_ONLY       // (knn_fused.hip includes this file for the device bodies above only)
static int knn_any(const float *base, const float *base_norms, const float *query, int B, int N, int S, int C, int K,
                   float *out_dist, int64_t *out_idx, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!base || !query || !out_idx || B <= 0 || N <= 0 || S <= 0 || C <= 0 || K <= 0) return MPA_EINVAL;
    if (K > 32 || K > N) return MPA_EUNSUPPORTED;
    if (C >= 8 && (C & 7)) return MPA_EUNSUPPORTED;   // norm rounding model only validated for C<8 or C%8==0
    hipStream_t st = (hipStream_t)stream;
    const bool al16 = (((uintptr_t)base | (uintptr_t)query) & 15) == 0;
    const bool scalar_path = getenv("MPA_KNN_SCALAR") != nullptr;       // development: the VALU kernel
    if (!scalar_path && (C == 3 || al16)) {
        switch (C) {                                                    // matrix-core kernel
        case 3: return launch_knn_mfma_k<3, 4, 1>(base, query, B, N, S, K, out_dist, out_idx, st);
        case 32: return launch_knn_mfma_k<32, 4, 2>(base, query, B, N, S, K, out_dist, out_idx, st, base_norms);
        case 64: return launch_knn_mfma_k<64, 4, 2>(base, query, B, N, S, K, out_dist, out_idx, st, base_norms);
        // wide rows: the few (S/32)*B workgroups are latency bound on staging their base tiles, so the
        // tiles are dealt to more waves (one 33 KB / 17 KB slab each: past the 64 KB default)
        case 128: return launch_knn_mfma_k<128, 4>(base, query, B, N, S, K, out_dist, out_idx, st, base_norms);
        case 256: return launch_knn_mfma_k<256, 2>(base, query, B, N, S, K, out_dist, out_idx, st, base_norms);
        default: break;
        }
    }
    switch (C) {                                                        // generic VALU kernel
    case 3: return launch_knn_k<3>(base, query, B, N, S, C, K, out_dist, out_idx, st);
    case 64: return launch_knn_k<64>(base, query, B, N, S, C, K, out_dist, out_idx, st);
    default: return launch_knn_k<0>(base, query, B, N, S, C, K, out_dist, out_idx, st);
    }
}

extern "C" int mpa_knn_f32(const float *base, const float *query, int B, int N, int S, int C, int K,
                           float *out_dist, int64_t *out_idx, void *stream)
{
    return knn_any(base, nullptr, query, B, N, S, C, K, out_dist, out_idx, stream);
}

extern "C" int mpa_row_norms_f32(const float *x, int B, int N, int C, float *norms, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!x || !norms || B <= 0 || N <= 0 || C <= 0 || B > 65535) return MPA_EINVAL;
    if (C >= 8 && (C & 7)) return MPA_EUNSUPPORTED;
    const int npad = mpa_ceil_div(N, 32) * 32;
    const dim3 grid(mpa_ceil_div(npad, 256), B);
    hipStream_t st = (hipStream_t)stream;
    switch (C) {
    case 32: hipLaunchKernelGGL(row_norms_kernel<32>, grid, dim3(256), 0, st, x, N, npad, C, norms); break;
    case 64: hipLaunchKernelGGL(row_norms_kernel<64>, grid, dim3(256), 0, st, x, N, npad, C, norms); break;
    case 128: hipLaunchKernelGGL(row_norms_kernel<128>, grid, dim3(256), 0, st, x, N, npad, C, norms); break;
    case 256: hipLaunchKernelGGL(row_norms_kernel<256>, grid, dim3(256), 0, st, x, N, npad, C, norms); break;
    default: hipLaunchKernelGGL(row_norms_kernel<0>, grid, dim3(256), 0, st, x, N, npad, C, norms); break;
    }
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_knn_norms_f32(const float *base, const float *base_norms, const float *query, int B, int N, int S,
                                 int C, int K, float *out_dist, int64_t *out_idx, void *stream)
{
    if (!base_norms || ((uintptr_t)base_norms & 15) != 0) return MPA_EINVAL;
    return knn_any(base, base_norms, query, B, N, S, C, K, out_dist, out_idx, stream);
}

extern "C" int mpa_fps_f32(const float *xyz, int B, int N, int S, const int64_t *start_idx, int64_t *out_idx,
                           float *out_xyz, void *stream);

extern "C" int mpa_fps_knn_xyz_f32(const float *fps_xyz, int B, int fps_N, int fps_S, const int64_t *start_idx,
                                   int64_t *fps_idx, float *fps_out_xyz, const float *knn_base,
                                   const float *knn_query, int N, int S, int K, float *out_dist, int64_t *out_idx,
                                   void *stream)
{
    MPA_CLEAR_ERROR();
    if (!fps_xyz || !start_idx || !fps_idx || !knn_base || !knn_query || !out_idx || B <= 0 || fps_N <= 0 ||
        fps_S <= 0 || N <= 0 || S <= 0 || K <= 0)
        return MPA_EINVAL;
    if (K > 32 || K > N) return MPA_EUNSUPPORTED;
    static const bool fused_on = getenv("MPA_NO_FPS_KNN_FUSION") == nullptr;
    const long long blocks = (long long)B + (long long)mpa_ceil_div(S, 32) * B;
    if (fused_on && K <= 8 && fps_N > 128 && fps_N <= 2048 && blocks < 0x7fffffffLL) {
        constexpr size_t knn_work = ((size_t)4 * (32 * (4 + 4) + 32) + KNN_G * 32 + 32 + 64 + 2 * 32 * KNN_CAP) * sizeof(float);
        constexpr size_t knn_merge = (size_t)32 * 2 * 4 * 8 * 8;
        constexpr size_t knn_lds = knn_work > knn_merge ? knn_work : knn_merge;
        const int P = fps_N <= 256 ? 1 : (fps_N <= 512 ? 2 : (fps_N <= 1024 ? 4 : 8));
        const size_t fps_lds = (size_t)3 * 256 * P * sizeof(float) + 2 * 4 * sizeof(uint2);
        const size_t lds = fps_lds > knn_lds ? fps_lds : knn_lds;
        hipStream_t st = (hipStream_t)stream;
#define MPA_FUSED_CASE(PP)                                                                                           \
    hipLaunchKernelGGL(fps_knn3_kernel<PP>, dim3((unsigned)blocks), dim3(256), lds, st, fps_xyz, fps_N, fps_S, start_idx, \
                       fps_idx, fps_out_xyz, B, knn_base, knn_query, N, S, K, out_dist, out_idx)
        if (P == 1) MPA_FUSED_CASE(1);
        else if (P == 2) MPA_FUSED_CASE(2);
        else if (P == 4) MPA_FUSED_CASE(4);
        else MPA_FUSED_CASE(8);
#undef MPA_FUSED_CASE
        MPA_LAUNCH_CHECK();
        return MPA_OK;
    }
    int rc = mpa_fps_f32(fps_xyz, B, fps_N, fps_S, start_idx, fps_idx, fps_out_xyz, stream);
    if (rc != MPA_OK) return rc;
    return mpa_knn_f32(knn_base, knn_query, B, N, S, 3, K, out_dist, out_idx, stream);
}

extern "C" int mpa_ball_query_f32(const float *base, const float *query, int B, int N, int S, int C,
                                  float radius2, int nsample, int64_t *out_idx, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!base || !query || !out_idx || B <= 0 || N <= 0 || S <= 0 || C <= 0 || nsample <= 0) return MPA_EINVAL;
    if (C >= 8 && (C & 7)) return MPA_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (C == 3) return launch_ball<3>(base, query, B, N, S, C, radius2, nsample, out_idx, st);
    return launch_ball<0>(base, query, B, N, S, C, radius2, nsample, out_idx, st);
}

extern "C" int mpa_square_distance_f32(const float *src, const float *dst, int B, int S, int N, int C,
                                       float *out, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!src || !dst || !out || B <= 0 || S <= 0 || N <= 0 || C <= 0) return MPA_EINVAL;
    if (C >= 8 && (C & 7)) return MPA_EUNSUPPORTED;
    dim3 grid(mpa_ceil_div(N, 256), mpa_ceil_div(S, 16), B);
    hipLaunchKernelGGL(sqdist_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, S, N, C, out);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}
#endif  // MPA_KNN_BODIES_ONLY
