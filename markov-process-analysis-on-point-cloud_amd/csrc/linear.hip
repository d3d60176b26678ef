// The transition / pointwise MLP unit for gfx950 -- reference Linear
// (modules/pointnet2_utils.py:401-425): nn.Linear -> BatchNorm1d over the B*S rows -> LeakyReLU.
//
// GEMM: LDS-tiled fp32 MFMA (v_mfma_f32_32x32x2_f32).  The f32 MFMA is bit-equal to an fmaf
// chain over k, so results do not depend on the tiling.  Both LDS tiles are kept k-major
// ([k][m] and [k][n]): a lane's MFMA operand is then As[k0 + (lane>>5)][m0 + (lane&31)], i.e.
// each half-wave reads 32 consecutive floats -- conflict-free ds_read_b32.  Row-major operands
// (activations [M,K], nn.Linear weights [N,K]) are transposed while being staged; k-major
// operands (dY^T, X in the weight-gradient product) are staged with straight float4 copies.
// The epilogue adds the bias, optionally accumulates with float atomics (split-K weight
// gradients) and optionally produces the per-column sum / sum of squares that BatchNorm needs,
// so the activation tensor is not re-read for statistics.
//
// Most layers here are K,N in {64,128}: arithmetic intensity ~16 FLOP/B against a machine
// balance of ~25, i.e. HBM-bound; the wide layers (512..2048) are MFMA-bound.
#include "mpa_common.h"
#include "mpa_bf16.h"
#include "splitk_reduce.h"
#include "geo_rider.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int NT = 256;   // threads per workgroup (4 waves)
constexpr int TS = 64;    // output tile: 64 x 64 per workgroup
constexpr int KS = 64;    // K slab staged per iteration; each of the 4 waves owns 16 of its k

// One workgroup = one 64x64 output tile, its 4 waves splitting K ("intra-block split-K"):
// every wave holds the whole tile in 4 MFMA accumulators (64 AGPRs) and consumes its own
// quarter of each 64-deep K slab, so the LDS operand reads are 4 ds_read_b32 per 4 MFMAs
// (256 MFMA cycles) and the small outputs of this model (M*N/4096 tiles) still put 4 waves on
// every CU.  Slabs are double-buffered in LDS with the next slab's global loads issued before
// the current slab's MFMAs (one barrier per slab).  The four partial tiles are summed through
// LDS, which also turns the epilogue into coalesced float4 row stores with the bias, the
// BatchNorm column statistics and (for split-K across workgroups) float atomics.
template <bool TA, bool TB>
__device__ __forceinline__ void gemm_body(const int block_x, const int block_z, const float *__restrict__ A, int lda,
                                          const float *__restrict__ B, int ldb, const float *__restrict__ bias,
                                          float *__restrict__ C, int ldc, int M, int N, int K, int kchunk,
                                          int out_mode, int vecA, int vecB, float *__restrict__ tile_stats,
                                          float *__restrict__ zero_c, float *__restrict__ a_col_sum, int tn_stream,
                                          int stats_acc = 0)
{
    // out_mode 0: C = result (+bias);  1: atomicAdd into C;  2: split-K partial slab
    //             C + blockIdx.z*M*ldc (plain stores, summed by splitk_reduce_kernel).
    // tile_stats [tiles_m][2][N] (optional): per 64-row tile and column, the sum and the sum of
    // squared deviations from the TILE mean (Chan's pairwise form).  Plain stores, no atomics:
    // BatchNorm statistics stay deterministic, and mpa_bn_finalize_f32 combines the tiles without
    // the catastrophic cancellation of E[y^2] - E[y]^2.
    // k-major LDS tiles [k][m] / [k][n], unpadded (2 x 2 x 16 KiB = exactly 64 KiB, so no
    // large-LDS opt-in is needed and two workgroups fit a CU).  Row-major global operands are
    // transposed while staged (4 strided ds_write_b32 per float4); the column index is XOR-swizzled
    // with SWZ(k) = 4*((k>>2)&7) so those writes spread over the banks (2-way at worst) while a
    // half-wave's MFMA operand read (32 consecutive columns of one k row) stays a permutation of
    // 32 banks.  k-major operands are copied with ds_write_b128 (the swizzle is a multiple of 4).
    constexpr int LDA = TS, LDB = TS;
    constexpr int BUF = KS * (LDA + LDB);
    extern __shared__ float lds[];   // 2 * BUF floats = 4*64*64: reused by the final reduction
#define SWZ(k) ((((k) >> 2) & 7) << 2)
#define PIN() __builtin_amdgcn_sched_barrier(0)   // nothing moves across: requests stay in front of the multiply phase

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;

    // XCD-aware tile order: consecutive workgroups are dealt round-robin to the 8 XCDs, so give
    // each XCD a contiguous run of tiles (tiles sharing A rows / B columns then share an L2).
    const int tiles_n = (N + TS - 1) / TS, tiles_m = (M + TS - 1) / TS;
    const int ntiles = tiles_m * tiles_n;
    int id = block_x;
    if ((ntiles & 7) == 0) id = (id & 7) * (ntiles >> 3) + (id >> 3);
    const int m0 = (id / tiles_n) * TS, n0 = (id % tiles_n) * TS;
    const int kbeg = block_z * kchunk;
    const int kend = min(K, kbeg + kchunk);
    const int nslab = (kend - kbeg + KS - 1) / KS;

    // two register sets: slab s+2 is requested while slab s is multiplied and slab s+1 (requested one
    // iteration earlier) is written to LDS, so a request has two MFMA phases (~1.7 us) to land
    float4 ra0[4], rb0[4], ra1[4], rb1[4];
    // ---- global -> registers for slab s (64 x 64 floats per operand = 4 float4 per lane)
    const bool m_full = m0 + TS <= M, n_full = n0 + TS <= N;
    auto load_slab = [&](auto interior, int s, float4 (&ra)[4], float4 (&rb)[4]) {
        constexpr bool INTERIOR = decltype(interior)::value;
        const int k0 = kbeg + s * KS;
        // interior tiles (the usual case): 8 unconditional 16-B loads, all in flight together.  The
        // edge path below is per-element predicated and the compiler serialises its loads.
        if constexpr (INTERIOR) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = tid + NT * q, r = i >> 4, c = (i & 15) * 4;
                const float *p = TA ? A + (size_t)(k0 + r) * lda + m0 + c : A + (size_t)(m0 + r) * lda + k0 + c;
                ra[q] = *reinterpret_cast<const float4 *>(p);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = tid + NT * q;
                int r, c;            // r: row of the global matrix walked by this lane, c: 4-wide column start
                int gr, gc;
                bool rok;
                if (!TA) { r = i >> 4; c = (i & 15) * 4; gr = m0 + r; gc = k0 + c; rok = gr < M; }
                else { r = i >> 4; c = (i & 15) * 4; gr = k0 + r; gc = m0 + c; rok = gr < kend; }
                const int lim = TA ? M : kend;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rok) {
                    const float *p = A + (size_t)gr * lda + gc;
                    if (vecA && gc + 3 < lim) v = *reinterpret_cast<const float4 *>(p);
                    else {
                        if (gc < lim) v.x = p[0];
                        if (gc + 1 < lim) v.y = p[1];
                        if (gc + 2 < lim) v.z = p[2];
                        if (gc + 3 < lim) v.w = p[3];
                    }
                }
                ra[q] = v;
            }
        }
        if constexpr (INTERIOR) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = tid + NT * q, r = i >> 4, c = (i & 15) * 4;
                const float *p = TB ? B + (size_t)(n0 + r) * ldb + k0 + c : B + (size_t)(k0 + r) * ldb + n0 + c;
                rb[q] = *reinterpret_cast<const float4 *>(p);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = tid + NT * q;
                int r, c, gr, gc;
                bool rok;
                if (TB) { r = i >> 4; c = (i & 15) * 4; gr = n0 + r; gc = k0 + c; rok = gr < N; }
                else { r = i >> 4; c = (i & 15) * 4; gr = k0 + r; gc = n0 + c; rok = gr < kend; }
                const int lim = TB ? kend : N;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rok) {
                    const float *p = B + (size_t)gr * ldb + gc;
                    if (vecB && gc + 3 < lim) v = *reinterpret_cast<const float4 *>(p);
                    else {
                        if (gc < lim) v.x = p[0];
                        if (gc + 1 < lim) v.y = p[1];
                        if (gc + 2 < lim) v.z = p[2];
                        if (gc + 3 < lim) v.w = p[3];
                    }
                }
                rb[q] = v;
            }
        }
    };
    // ---- registers -> LDS buffer
    auto store_slab = [&](float *buf, const float4 (&ra)[4], const float4 (&rb)[4]) {
        float *As = buf, *Bs = buf + KS * LDA;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + NT * q;
            const int r = i >> 4, c = (i & 15) * 4;
            if (!TA) {            // lane holds A[m = r][k = c..c+3] -> As[k][m ^ SWZ(k)]  (SWZ(c..c+3) equal)
                float *d = As + c * LDA + (r ^ SWZ(c));
                d[0] = ra[q].x; d[LDA] = ra[q].y; d[2 * LDA] = ra[q].z; d[3 * LDA] = ra[q].w;
            } else {              // lane holds A^T[k = r][m = c..c+3]
                *reinterpret_cast<float4 *>(As + r * LDA + (c ^ SWZ(r))) = ra[q];
            }
            if (TB) {             // lane holds B[n = r][k = c..c+3] -> Bs[k][n ^ SWZ(k)]
                float *d = Bs + c * LDB + (r ^ SWZ(c));
                d[0] = rb[q].x; d[LDB] = rb[q].y; d[2 * LDB] = rb[q].z; d[3 * LDB] = rb[q].w;
            } else {
                *reinterpret_cast<float4 *>(Bs + r * LDB + (c ^ SWZ(r))) = rb[q];
            }
        }
    };

    float asum0 = 0.f, asum1 = 0.f;          // sum over k of this lane's A operands (a_col_sum)
    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    bool streamed = false;
    if constexpr (TA && !TB) {
      if (tn_stream) {
        streamed = true;
        // ---- k-major x k-major (weight gradients: dY^T [K][M] times X [K][N], K = B*S rows):
        // both MFMA operands are already laid out the way the instruction wants them -- for a
        // fixed k the 32 lanes of a half-wave read 32 consecutive floats (one 128-B line) -- so
        // each wave streams its quarter of the K range straight from global memory into
        // registers, 8 k-pairs (32 loads of 256 B) in flight ahead of the MFMAs.  No LDS staging,
        // no workgroup barrier in the loop: this product is a pure HBM stream with a tiny output.
        const int kspan = kend - kbeg;
        int kq = ((kspan + 3) / 4 + 1) & ~1;                 // per-wave share, even
        const int k0w = kbeg + wave * kq;
        const int k1w = min(kend, k0w + kq);
        const int ma0 = min(m0 + l31, M - 1), ma1 = min(m0 + 32 + l31, M - 1);     // clamped: rows/cols
        const int nb0 = min(n0 + l31, N - 1), nb1 = min(n0 + 32 + l31, N - 1);     // beyond M,N are dropped
        constexpr int U = 8;
        float pa0[U], pa1[U], pb0[U], pb1[U];
        auto issue = [&](int kb) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = kb + 2 * u + half;
                const bool ok = k < k1w;
                const float *ar = A + (size_t)(ok ? k : k0w) * lda;
                const float *br = B + (size_t)(ok ? k : k0w) * ldb;
                const float x0 = ar[ma0], x1 = ar[ma1], y0 = br[nb0], y1 = br[nb1];
                pa0[u] = ok ? x0 : 0.f; pa1[u] = ok ? x1 : 0.f;
                pb0[u] = ok ? y0 : 0.f; pb1[u] = ok ? y1 : 0.f;
            }
        };
        if (k0w < k1w) {
            issue(k0w);
            for (int kb = k0w; kb < k1w; kb += 2 * U) {
                float ca0[U], ca1[U], cb0[U], cb1[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { ca0[u] = pa0[u]; ca1[u] = pa1[u]; cb0[u] = pb0[u]; cb1[u] = pb1[u]; }
                if (kb + 2 * U < k1w) issue(kb + 2 * U);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca0[u], cb0[u], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca0[u], cb1[u], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca1[u], cb0[u], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca1[u], cb1[u], acc[1][1], 0, 0, 0);
                    asum0 += ca0[u];
                    asum1 += ca1[u];
                }
            }
        }
      }
    }
    if (!streamed) {
    auto multiply = [&](const float *buf) {
        const float *As = buf + (wave * 16 + half) * LDA;
        const float *Bs = buf + KS * LDA + (wave * 16 + half) * LDB;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            // k = wave*16 + 2t + half: (k>>2)&7 = (wave*4 + (t>>1)) & 7, the same for both halves
            const int cs = l31 ^ ((((wave << 2) + (t >> 1)) & 7) << 2);
            const float a0 = As[2 * t * LDA + cs], a1 = As[2 * t * LDA + 32 + cs];
            const float b0 = Bs[2 * t * LDB + cs], b1 = Bs[2 * t * LDB + 32 + cs];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            asum0 += a0;
            asum1 += a1;
        }
    };
    // multiply(buf) with the operands of step t+1 read from LDS before step t's MFMAs are issued, and -- optionally -- the
    // other register set written to `nbuf` a quarter at a time between the MFMA groups, so the LDS traffic of a slab runs
    // under its MFMAs instead of in front of the barrier
    auto store_quarter = [&](float *buf, const float4 (&ra)[4], const float4 (&rb)[4], int q) {
        float *As = buf, *Bs = buf + KS * LDA;
        const int i = tid + NT * q;
        const int r = i >> 4, c = (i & 15) * 4;
        if (!TA) {
            float *d = As + c * LDA + (r ^ SWZ(c));
            d[0] = ra[q].x; d[LDA] = ra[q].y; d[2 * LDA] = ra[q].z; d[3 * LDA] = ra[q].w;
        } else {
            *reinterpret_cast<float4 *>(As + r * LDA + (c ^ SWZ(r))) = ra[q];
        }
        if (TB) {
            float *d = Bs + c * LDB + (r ^ SWZ(c));
            d[0] = rb[q].x; d[LDB] = rb[q].y; d[2 * LDB] = rb[q].z; d[3 * LDB] = rb[q].w;
        } else {
            *reinterpret_cast<float4 *>(Bs + r * LDB + (c ^ SWZ(r))) = rb[q];
        }
    };
    auto multiply_store = [&](const float *buf, float *nbuf, const float4 (&ra)[4], const float4 (&rb)[4], auto with_store) {
        constexpr bool STORE = decltype(with_store)::value;
        const float *As = buf + (wave * 16 + half) * LDA;
        const float *Bs = buf + KS * LDA + (wave * 16 + half) * LDB;
        float a0[2], a1[2], b0[2], b1[2];
        auto fetch = [&](int t, int slot) {
            const int cs = l31 ^ ((((wave << 2) + (t >> 1)) & 7) << 2);
            a0[slot] = As[2 * t * LDA + cs]; a1[slot] = As[2 * t * LDA + 32 + cs];
            b0[slot] = Bs[2 * t * LDB + cs]; b1[slot] = Bs[2 * t * LDB + 32 + cs];
        };
        fetch(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int sl = t & 1;
            if (t < 7) fetch(t + 1, sl ^ 1);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[sl], b0[sl], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[sl], b1[sl], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[sl], b0[sl], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[sl], b1[sl], acc[1][1], 0, 0, 0);
            asum0 += a0[sl];
            asum1 += a1[sl];
            if (STORE && (t & 1)) store_quarter(nbuf, ra, rb, t >> 1);
            if (t < 7) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // the two operand reads of step t+1
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                 // step t's MFMAs
            if (STORE && (t & 1)) __builtin_amdgcn_sched_group_barrier(0x200, 4, 0);
        }
    };
    auto main_loop = [&](auto interior) {
        constexpr bool INTERIOR = decltype(interior)::value;
        if (INTERIOR && nslab >= 4) {
            // Straight-line prologue, steady loop and tail: with nothing conditional between a request and its use the
            // compiler's wait counts are exact (vmcnt(8): the older register set has landed, the newer is in flight).
            // Behind a conditional request they were merged to vmcnt(0) at the loop head, which left a request less
            // than one multiply phase to land instead of two.
            load_slab(interior, 0, ra0, rb0);
            load_slab(interior, 1, ra1, rb1);
            store_slab(lds, ra0, rb0);
            __syncthreads();
            int s = 0;
            for (; s + 3 < nslab; s += 2) {
                load_slab(interior, s + 2, ra0, rb0);
                PIN();
                multiply_store(lds, lds + BUF, ra1, rb1, std::true_type{});
                __syncthreads();
                load_slab(interior, s + 3, ra1, rb1);
                PIN();
                multiply_store(lds + BUF, lds, ra0, rb0, std::true_type{});
                __syncthreads();
            }
            if (nslab - s == 3) {
                load_slab(interior, s + 2, ra0, rb0);
                PIN();
                multiply_store(lds, lds + BUF, ra1, rb1, std::true_type{});
                __syncthreads();
                multiply_store(lds + BUF, lds, ra0, rb0, std::true_type{});
                __syncthreads();
                multiply_store(lds, nullptr, ra0, rb0, std::false_type{});
            } else {
                multiply_store(lds, lds + BUF, ra1, rb1, std::true_type{});
                __syncthreads();
                multiply_store(lds + BUF, nullptr, ra0, rb0, std::false_type{});
            }
            __syncthreads();          // the epilogue reuses the slab buffers: every wave's operand reads are done
            return;
        }
        if (nslab > 0) {
            load_slab(interior, 0, ra0, rb0);
            if (nslab > 1) load_slab(interior, 1, ra1, rb1);
            store_slab(lds, ra0, rb0);
        }
        __syncthreads();
        for (int s = 0; s < nslab; s += 2) {        // short or ragged products: every step conditional
            if (s + 2 < nslab) load_slab(interior, s + 2, ra0, rb0);
            multiply(lds);
            if (s + 1 < nslab) store_slab(lds + BUF, ra1, rb1);
            __syncthreads();
            if (s + 1 < nslab) {
                if (s + 3 < nslab) load_slab(interior, s + 3, ra1, rb1);
                multiply(lds + BUF);
                if (s + 2 < nslab) store_slab(lds, ra0, rb0);
                __syncthreads();
            }
        }
    };
    // interior tiles (the usual case) run a loop whose 8 loads per slab are unconditional 16-B
    // loads, all in flight together; the edge variant is per-element predicated
    if (vecA && vecB && m_full && n_full && ((kend - kbeg) % KS) == 0)
        main_loop(std::true_type{});
    else
        main_loop(std::false_type{});

    }   // LDS-staged main loop

    // ---- optional: column sums of op(A) over this workgroup's K range, i.e. sum_k A^T[k][m] -- the
    // bias gradient when A = dY^T in the weight-gradient product.  Only the n-tile 0 workgroups
    // contribute (every n-tile sees the same A rows); one float atomic per (workgroup, m).
    if (a_col_sum != nullptr && n0 == 0) {
        asum0 += __shfl_xor(asum0, 32, 64);
        asum1 += __shfl_xor(asum1, 32, 64);
        if (half == 0) {
            if (m0 + l31 < M) atomicAdd(a_col_sum + m0 + l31, asum0);
            if (m0 + 32 + l31 < M) atomicAdd(a_col_sum + m0 + 32 + l31, asum1);
        }
    }

    // ---- sum the 4 waves' partial tiles through LDS: red[wave][row][col], 64 x 64 each
    float *red = lds + wave * (TS * TS);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                red[row * TS + j * 32 + l31] = acc[i][j][r];
            }
    __syncthreads();

    if (out_mode == 1) {
        // float atomics want 256 contiguous bytes per wave instruction: lane = column
        const int cl = tid & 63, colA = n0 + cl;
        const float bA = (bias != nullptr && block_z == 0 && colA < N) ? bias[colA] : 0.f;
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int rloc = (tid >> 6) + 4 * q;
            const float *p = lds + rloc * TS + cl;
            const float v = ((p[0] + p[TS * TS]) + (p[2 * TS * TS] + p[3 * TS * TS])) + bA;
            if (m0 + rloc < M && colA < N) atomicAdd(C + (size_t)(m0 + rloc) * ldc + colA, v);
        }
        return;
    }
    if (out_mode == 2) {
        // split-K slabs are summed into the real output by splitk_reduce_kernel (atomics): the z = 0
        // workgroups clear their tile of it here, saving a separate memset launch
        if (zero_c != nullptr && block_z == 0) {
            const int cz = n0 + (tid & 15) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rz = m0 + (tid >> 4) + 16 * q;
                if (rz < M)
                    for (int e = 0; e < 4; ++e)
                        if (cz + e < N) zero_c[(size_t)rz * ldc + cz + e] = 0.f;
            }
        }
        C += (size_t)block_z * M * ldc;
    }

    const int c4 = (tid & 15) * 4;            // this lane's 4 columns (same for its 4 rows)
    const int col = n0 + c4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias != nullptr && block_z == 0) {
        if (col < N) bv.x = bias[col];
        if (col + 1 < N) bv.y = bias[col + 1];
        if (col + 2 < N) bv.z = bias[col + 2];
        if (col + 3 < N) bv.w = bias[col + 3];
    }
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 vkeep[4];
    const bool vecC = ((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rloc = (tid >> 4) + 16 * q;
        const int row = m0 + rloc;
        const float *p = lds + rloc * TS + c4;
        float4 v = *reinterpret_cast<const float4 *>(p);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 u = *reinterpret_cast<const float4 *>(p + w * TS * TS);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        vkeep[q] = v;
        if (row < M) {
            float *o = C + (size_t)row * ldc + col;
            if (vecC && col + 3 < N) {
                *reinterpret_cast<float4 *>(o) = v;
            } else {
                if (col < N) o[0] = v.x;
                if (col + 1 < N) o[1] = v.y;
                if (col + 2 < N) o[2] = v.z;
                if (col + 3 < N) o[3] = v.w;
            }
            s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
        }
    }
    if (tile_stats != nullptr) {
        // column sums over the tile's rows: lanes t, t+16, t+32, t+48 of a wave share the columns,
        // then the 4 waves through LDS; the tile mean goes back to every lane for the second pass
        const int tile_m = id / tiles_n;
        const int nrows = min(TS, M - m0);
        float *st = lds;                       // [4 waves][64] scratch, then [64] tile means
        float vs[4] = {s1.x, s1.y, s1.z, s1.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            vs[e] += __shfl_xor(vs[e], 16, 64);
            vs[e] += __shfl_xor(vs[e], 32, 64);
        }
        __syncthreads();                       // everyone is done reading the partial tiles
        if (lane < 16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) st[wave * TS + c4 + e] = vs[e];
        }
        __syncthreads();
        float tsum = 0.f;
        if (tid < TS) {
            tsum = (st[tid] + st[TS + tid]) + (st[2 * TS + tid] + st[3 * TS + tid]);
            st[4 * TS + tid] = tsum / (float)nrows;
        }
        __syncthreads();
        const float4 mu = *reinterpret_cast<const float4 *>(st + 4 * TS + c4);
        float4 d2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (m0 + (tid >> 4) + 16 * q < M) {
                float dx = vkeep[q].x - mu.x, dy = vkeep[q].y - mu.y, dz = vkeep[q].z - mu.z, dw = vkeep[q].w - mu.w;
                d2.x = fmaf(dx, dx, d2.x); d2.y = fmaf(dy, dy, d2.y);
                d2.z = fmaf(dz, dz, d2.z); d2.w = fmaf(dw, dw, d2.w);
            }
        }
        float vq[4] = {d2.x, d2.y, d2.z, d2.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            vq[e] += __shfl_xor(vq[e], 16, 64);
            vq[e] += __shfl_xor(vq[e], 32, 64);
        }
        if (lane < 16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) st[(5 + wave) * TS + c4 + e] = vq[e];
        }
        __syncthreads();
        if (tid < TS && n0 + tid < N) {
            const float m2 = (st[5 * TS + tid] + st[6 * TS + tid]) + (st[7 * TS + tid] + st[8 * TS + tid]);
            if (stats_acc > 0) {
                // accumulate form [R][3][N]: sum, within-tile M2, sum^2/rows -- mpa_bn_stats_act_fwd finishes them
                float *dst = tile_stats + (size_t)(tile_m % stats_acc) * 3 * N + n0 + tid;
                atomicAdd(dst, tsum);
                atomicAdd(dst + N, m2);
                atomicAdd(dst + 2 * N, tsum * tsum / (float)nrows);
            } else {
                float *dst = tile_stats + (size_t)tile_m * 2 * N + n0 + tid;
                dst[0] = tsum;
                dst[N] = m2;
            }
        }
    }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(NT, 2) void gemm_kernel(const float *__restrict__ A, int lda,
                                                     const float *__restrict__ B, int ldb,
                                                     const float *__restrict__ bias, float *__restrict__ C, int ldc,
                                                     int M, int N, int K, int kchunk, int out_mode, int vecA,
                                                     int vecB, float *__restrict__ tile_stats, float *__restrict__ zero_c,
                                                     float *__restrict__ a_col_sum, int tn_stream, int stats_acc)
{
    gemm_body<TA, TB>(blockIdx.x, blockIdx.z, A, lda, B, ldb, bias, C, ldc, M, N, K, kchunk, out_mode, vecA, vecB,
                      tile_stats, zero_c, a_col_sum, tn_stream, stats_acc);
}

// ---- grouped weight-gradient products: many independent small  out_p[M_p,N_p] = A_p^T B_p  (A_p, B_p
// k-major with K_p = B*S rows) in ONE launch.  Each is a latency-bound HBM stream with a tiny
// output; issued one by one they cost ~17 us apiece with most of the chip idle, together their
// streams overlap.  Problem descriptors travel in the kernel arguments (no device table).
struct GroupedProblem {
    const float *A, *B;
    float *out, *a_col_sum, *slab;       // slab: split-K partial tiles [splits][M*N] (or nullptr: direct)
    int lda, ldb, M, N, K, kchunk, splits, tiles, vec, stream;
};
struct GroupedArgs {
    int count;
    int block_start[GROUP_MAX + 1];      // prefix sum of tiles*splits
    GroupedProblem p[GROUP_MAX];
};

__device__ __forceinline__ void tn_grouped_block(const GroupedArgs &args, const int bid)
{
    __shared__ int which;
    if (threadIdx.x == 0) {
        int b = bid, i = 0;
        while (i + 1 < args.count && b >= args.block_start[i + 1]) ++i;
        which = i;
    }
    __syncthreads();
    const GroupedProblem &q = args.p[which];
    const int local = bid - args.block_start[which];
    const int tile = local % q.tiles, z = local / q.tiles;
    const int vecA = q.vec & 1, vecB = (q.vec >> 1) & 1;
    if (q.slab != nullptr)
        gemm_body<true, false>(tile, z, q.A, q.lda, q.B, q.ldb, nullptr, q.slab, q.N, q.M, q.N, q.K, q.kchunk, 2, vecA,
                               vecB, nullptr, q.out, q.a_col_sum, q.stream);
    else
        gemm_body<true, false>(tile, z, q.A, q.lda, q.B, q.ldb, nullptr, q.out, q.N, q.M, q.N, q.K, q.kchunk, 0, vecA,
                               vecB, nullptr, nullptr, q.a_col_sum, q.stream);
}

__global__ __launch_bounds__(NT, 2) void gemm_tn_grouped_kernel(const GroupedArgs args)
{
    tn_grouped_block(args, blockIdx.x);
}

// The same launch carrying a geometry rider (geo_rider.h): its first r.blocks workgroups sample / search the NEXT
// batch's coordinates -- they are dispatched first and run for the length of the launch on one CU slot per cloud --
// the others are the weight-gradient products.
// sampling only (no search items): the workers' loop touches nothing of the rider, which keeps its scalar registers
// out of the tile body (the general kernel below spills 129 of them)
__global__ __launch_bounds__(NT, 2) void gemm_tn_grouped_fps_kernel(const GroupedArgs args, const RiderArgs r,
                                                                    const RiderPlace pl, const int gemm_blocks)
{
    extern __shared__ float lds[];
    const int id = blockIdx.x;
    if (id < pl.region && (id & 7) < pl.nx) {
        rider_sample_or_park(r, rider_slot(pl, id), lds);
        return;
    }
    const int rank = id < pl.region ? (id >> 3) * (8 - pl.nx) + ((id & 7) - pl.nx) : id - pl.slots * pl.nx;
    const int stride = pl.workers;
    for (int item = rank; item < gemm_blocks; item += stride) {
        tn_grouped_block(args, item);
        __syncthreads();                          // the next tile reuses this one's LDS
    }
}

__global__ __launch_bounds__(NT, 2) void gemm_tn_grouped_rider_kernel(const GroupedArgs args, const RiderArgs r,
                                                                      const RiderPlace pl, const int gemm_blocks)
{
    extern __shared__ float lds[];
    const int slot = rider_slot(pl, blockIdx.x);
    if (slot != -1) {
        rider_sample_or_park(r, slot, lds);          // (see geo_rider.h: the chain gets CUs of its own)
        return;
    }
    // a worker: its rank among the launch's worker ids, then every `workers`-th item from there (workers is a multiple
    // of 8 and ranks keep id % 8 apart, so a worker -- and with it an XCD -- stays inside one residue class of the items,
    // which is what the carrier's XCD-aware tile order assumes)
    const int id = blockIdx.x;
    const int lanes_free = 8 - pl.nx;
    const int rank = id < pl.region ? (id >> 3) * lanes_free + ((id & 7) - pl.nx) : id - pl.slots * pl.nx;
    const int total = r.sblocks + gemm_blocks;
    for (int item = rank; item < total; item += pl.workers) {
        if (item < r.sblocks) rider_search(r, item, lds);
        else tn_grouped_block(args, item - r.sblocks);
        __syncthreads();                          // the next item reuses this one's LDS
    }
}

// ---- grouped forward / dX products: a few INDEPENDENT  C_p = A_p op(B_p) (+ bias)  in one launch -- the Linear units
// of LocalMerge's parallel attention streams (ffn | ffn, conv_res | conv_res) and of Fuse's four source
// states.  Each is a ~1 GFLOP problem of 256-512 tiles that leaves the chip half empty and pays its own ramp-up
// and tail; together their phases overlap.  Same tile body as gemm_kernel (no split-K: the group is the
// parallelism); descriptors travel in the kernel arguments.
constexpr int GROUP_NT_MAX = 8;
struct NtProblem {
    const float *A, *B, *bias;
    float *C, *stats;
    int lda, ldb, ldc, M, N, K, tiles, stats_acc, vec;
};
struct NtArgs {
    int count;
    int block_start[GROUP_NT_MAX + 1];
    NtProblem p[GROUP_NT_MAX];
};

template <bool TB>
__global__ __launch_bounds__(NT, 2) void gemm_nt_grouped_kernel(const NtArgs args)
{
    __shared__ int which;
    if (threadIdx.x == 0) {
        int b = blockIdx.x, i = 0;
        while (i + 1 < args.count && b >= args.block_start[i + 1]) ++i;
        which = i;
    }
    __syncthreads();
    const NtProblem &q = args.p[which];
    const int kchunk = ((q.K + KS - 1) / KS) * KS;
    gemm_body<false, TB>(blockIdx.x - args.block_start[which], 0, q.A, q.lda, q.B, q.ldb, q.bias, q.C, q.ldc, q.M, q.N, q.K,
                         kchunk, 0, q.vec & 1, (q.vec >> 1) & 1, q.stats, nullptr, nullptr, 0, q.stats_acc);
}

// ---- short-K products (K = 64 or 128, whole tiles): the layers of the fine point-set states,
// M = B*S up to 65536 rows against 64..256 channels.  They are HBM-bound (16 FLOP/B) and, in the
// general kernel above, latency-bound: one slab means no pipelining inside a workgroup, and its
// 192 registers + 64 KiB of LDS allow two workgroups per CU.  Here a workgroup stages the whole
// K extent of its 64 x 64 tile at once (32 KiB per 64 of K), every wave multiplies its own 32 x 32
// quadrant over all of K (no partial tiles to sum through LDS) and stores from the accumulator;
// <= 64 registers, so four (K = 64) or two (K = 128) workgroups per CU overlap their load, MFMA
// and store phases.
template <bool TB, int NS>
__global__ __launch_bounds__(NT, 4) void gemm_shortk_kernel(const float *__restrict__ A, int lda,
                                                            const float *__restrict__ B, int ldb,
                                                            const float *__restrict__ bias, float *__restrict__ C,
                                                            int ldc, int M, int N, float *__restrict__ tile_stats,
                                                            int stats_acc)
{
    constexpr int LDA = TS, LDB = TS;
    constexpr int BUF = KS * (LDA + LDB);
    extern __shared__ float lds[];                 // NS slabs of [k][m] | [k][n], swizzled as in gemm_body
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int tiles_n = N / TS, ntiles = (M / TS) * tiles_n;
    int id = blockIdx.x;
    if ((ntiles & 7) == 0) id = (id & 7) * (ntiles >> 3) + (id >> 3);
    const int tile_m = id / tiles_n;
    const int m0 = tile_m * TS, n0 = (id % tiles_n) * TS;

    float4 ra[NS][4], rb[NS][4];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + NT * q, r = i >> 4, c = (i & 15) * 4;
            ra[s][q] = *reinterpret_cast<const float4 *>(A + (size_t)(m0 + r) * lda + s * KS + c);
            rb[s][q] = *reinterpret_cast<const float4 *>(TB ? B + (size_t)(n0 + r) * ldb + s * KS + c
                                                            : B + (size_t)(s * KS + r) * ldb + n0 + c);
        }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        float *As = lds + s * BUF, *Bs = As + KS * LDA;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + NT * q, r = i >> 4, c = (i & 15) * 4;
            {   // lane holds A[m = r][k = c..c+3] -> As[k][m ^ SWZ(k)]
                float *d = As + c * LDA + (r ^ SWZ(c));
                d[0] = ra[s][q].x; d[LDA] = ra[s][q].y; d[2 * LDA] = ra[s][q].z; d[3 * LDA] = ra[s][q].w;
            }
            if (TB) {
                float *d = Bs + c * LDB + (r ^ SWZ(c));
                d[0] = rb[s][q].x; d[LDB] = rb[s][q].y; d[2 * LDB] = rb[s][q].z; d[3 * LDB] = rb[s][q].w;
            } else {
                *reinterpret_cast<float4 *>(Bs + r * LDB + (c ^ SWZ(r))) = rb[s][q];
            }
        }
    }
    __syncthreads();

    const int mq = (wave & 1) * 32, nq = (wave >> 1) * 32;
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float *As = lds + s * BUF + half * LDA, *Bs = lds + s * BUF + KS * LDA + half * LDB;
#pragma unroll 8
        for (int t = 0; t < KS / 2; ++t) {
            const int cs = l31 ^ (((t >> 1) & 7) << 2);          // SWZ(2t + half): (k >> 2) & 7 = (t >> 1) & 7
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[2 * t * LDA + mq + cs], Bs[2 * t * LDB + nq + cs], acc, 0, 0, 0);
        }
    }

    // ---- epilogue straight from the accumulator: lane = column, 16 rows ((r&3) + 8(r>>2) + 4 half)
    const int col = n0 + nq + l31;
    const float bv = bias != nullptr ? bias[col] : 0.f;
    float csum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float v = acc[r] + bv;
        acc[r] = v;
        C[(size_t)(m0 + mq + (r & 3) + 8 * (r >> 2) + 4 * half) * ldc + col] = v;
        csum += v;
    }
    if (tile_stats != nullptr) {
        // per-column sum and M2 (about the tile mean) over the tile's 64 rows: 16 rows per lane,
        // the other half-wave's 16, the partner wave's 32 (through LDS: everyone is done reading it)
        float *st = lds;                             // [4 waves][32] sums, then [4][32] M2 at +128
        __syncthreads();
        csum += __shfl_xor(csum, 32, 64);
        if (half == 0) st[wave * 32 + l31] = csum;
        __syncthreads();
        const float tsum = st[wave * 32 + l31] + st[(wave ^ 1) * 32 + l31];
        const float mu = tsum * (1.0f / TS);
        float m2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = acc[r] - mu;
            m2 = fmaf(d, d, m2);
        }
        m2 += __shfl_xor(m2, 32, 64);
        if (half == 0) st[128 + wave * 32 + l31] = m2;
        __syncthreads();
        if (mq == 0 && half == 0) {
            const float m2t = st[128 + wave * 32 + l31] + st[128 + (wave ^ 1) * 32 + l31];
            if (stats_acc > 0) {
                float *dst = tile_stats + (size_t)(tile_m % stats_acc) * 3 * N + col;
                atomicAdd(dst, tsum);
                atomicAdd(dst + N, m2t);
                atomicAdd(dst + 2 * N, tsum * tsum * (1.0f / TS));
            } else {
                float *dst = tile_stats + (size_t)tile_m * 2 * N + col;
                dst[0] = tsum;
                dst[N] = m2t;
            }
        }
    }
}

constexpr size_t gemm_lds_bytes(bool, bool) { return sizeof(float) * 2 * KS * (TS + TS); }   // 64 KiB

template <bool TA, bool TB>
int launch_gemm(const float *A, int lda, const float *B, int ldb, const float *bias, float *C, int ldc, int M, int N,
                int K, int splits, int kchunk, int out_mode, int vecA, int vecB, float *stats, float *zero_c,
                float *a_col_sum, int stats_acc, hipStream_t st)
{
    static const int tn_stream = getenv("MPA_TN_STREAM") ? atoi(getenv("MPA_TN_STREAM")) : 1;
    constexpr size_t lds = gemm_lds_bytes(TA, TB);
    static_assert(lds >= sizeof(float) * 4 * TS * TS && lds <= 64 * 1024, "reduction region fits, no opt-in");
    dim3 grid(mpa_ceil_div(M, TS) * mpa_ceil_div(N, TS), 1, splits);
    hipLaunchKernelGGL((gemm_kernel<TA, TB>), grid, dim3(NT), lds, st, A, lda, B, ldb, bias, C, ldc, M, N, K, kchunk,
                       out_mode, vecA, vecB, stats, zero_c, a_col_sum, tn_stream, stats_acc);
    return MPA_OK;
}

// out[M,N] += sum_z part[z][M,N]   (split-K partial slabs written with plain stores).
// The sum over z is itself split over gridDim.y workgroups (a 64x64 weight gradient has 4096
// outputs but hundreds of slabs), each adding its share with one float atomic per element: at
// most gridDim.y (<= 32) adders per address.  `out` is zeroed by the caller unless accumulating.
__global__ void splitk_reduce_kernel(const float *__restrict__ part, int splits, long long mn,
                                     float *__restrict__ out)
{
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= mn) return;
    const int zper = (splits + gridDim.y - 1) / gridDim.y;
    const int z0 = blockIdx.y * zper, z1 = min(splits, z0 + zper);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int z = z0;
    for (; z + 3 < z1; z += 4) {
        a0 += part[(size_t)z * mn + i];
        a1 += part[(size_t)(z + 1) * mn + i];
        a2 += part[(size_t)(z + 2) * mn + i];
        a3 += part[(size_t)(z + 3) * mn + i];
    }
    for (; z < z1; ++z) a0 += part[(size_t)z * mn + i];
    if (z0 < z1) atomicAdd(out + i, (a0 + a1) + (a2 + a3));
}

// ------------------------------------------------------------------ BatchNorm + LeakyReLU pieces
constexpr int EW_TPB = 256;

// Per-tile statistics of an existing [M,C] tensor in the GEMM epilogue's format
// (tile_stats [ceil(M/64)][2][C]: sum, sum of squared deviations from the tile mean).
__global__ __launch_bounds__(256) void tile_stats_kernel(const float *__restrict__ x, int M, int C,
                                                         float *__restrict__ tile_stats, int stats_acc)
{
    __shared__ float red[4][64];
    __shared__ float mean[64];
    const int tid = threadIdx.x, cl = tid & 63, g = tid >> 6;
    const int c = blockIdx.x * 64 + cl, r0 = blockIdx.y * TS;
    const int nrows = min(TS, M - r0);
    float v[16];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int r = g + 4 * q;
        v[q] = (r < nrows && c < C) ? x[(size_t)(r0 + r) * C + c] : 0.f;
        s += v[q];
    }
    red[g][cl] = s;
    __syncthreads();
    if (g == 0) {
        s = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        mean[cl] = s / (float)nrows;
    }
    __syncthreads();
    const float mu = mean[cl];
    float m2 = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q)
        if (g + 4 * q < nrows) { const float d = v[q] - mu; m2 = fmaf(d, d, m2); }
    __syncthreads();
    red[g][cl] = m2;
    __syncthreads();
    if (g == 0 && c < C) {
        const float m2t = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        if (stats_acc > 0) {
            float *dst = tile_stats + (size_t)(blockIdx.y % stats_acc) * 3 * C + c;
            atomicAdd(dst, s);
            atomicAdd(dst + C, m2t);
            atomicAdd(dst + 2 * C, s * s / (float)nrows);
        } else {
            float *dst = tile_stats + (size_t)blockIdx.y * 2 * C + c;
            dst[0] = s;
            dst[C] = m2t;
        }
    }
}

// mean / invstd per channel.  Training: combine the tiles' (sum, M2) pairwise-exactly --
// mean = sum(sum_t)/M, M2 = sum(M2_t + n_t*(sum_t/n_t - mean)^2), biased variance M2/M -- in a
// fixed order (deterministic), and update the running statistics (momentum, unbiased variance)
// as nn.BatchNorm1d does.  Eval: from the running statistics.  out = save [2][C] (mean, invstd).
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float *__restrict__ tile_stats, int M, int C,
                                                           float *__restrict__ running_mean,
                                                           float *__restrict__ running_var, int training,
                                                           float momentum, float eps, float *__restrict__ save,
                                                           float *__restrict__ zero_me, int zero_n,
                                                           long long *__restrict__ num_batches_tracked)
{
    // 1024 lanes = 16 tile-groups x 64 channels.  Each lane folds its tiles (t = g, g+16, ...) into
    // one (count, mean, M2) triple with Chan's pairwise update -- one pass, 16 tiles' (sum, M2) pairs
    // loaded together -- and the 16 triples of a channel are merged in LDS in a fixed order:
    // deterministic, and no E[y^2] - E[y]^2 cancellation anywhere.
    __shared__ float red_n[16][64], red_mu[16][64], red_m2[16][64];
    const int tid = threadIdx.x, cl = tid & 63, g = tid >> 6;
    const int c = blockIdx.x * 64 + cl;
    if (blockIdx.x == 0) {
        for (int i = tid; i < zero_n; i += 1024) zero_me[i] = 0.f;     // scratch the backward pass accumulates into
        if (tid == 0 && training && num_batches_tracked != nullptr) *num_batches_tracked += 1;
    }
    if (!training) {
        if (g == 0 && c < C) {
            save[c] = running_mean[c];
            save[C + c] = 1.0f / sqrtf(running_var[c] + eps);
        }
        return;
    }
    const int tiles = (M + TS - 1) / TS;
    const size_t st = (size_t)2 * C;
    float an = 0.f, amu = 0.f, am2 = 0.f;
    auto merge = [&](float bn, float bmu, float bm2) {
        const float n = an + bn;
        const float d = bmu - amu;
        const float w = __fdividef(bn, n);          // a weight in (0,1]: 2-ulp division is ample
        amu = fmaf(d, w, amu);
        am2 = am2 + bm2 + d * d * an * w;
        an = n;
    };
    if (c < C) {
        constexpr int U = 16;
        for (int base = g; base < tiles; base += 16 * U) {
            float sm[U], sq[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = min(base + 16 * u, tiles - 1);       // clamped: unconditional loads
                sm[u] = tile_stats[(size_t)t * st + c];
                sq[u] = tile_stats[(size_t)t * st + C + c];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = base + 16 * u;
                if (t < tiles) {
                    const int rows = min(TS, M - t * TS);
                    const float nt = (float)rows;
                    // full tiles: * 2^-6 is exact; only the last, partial tile divides
                    merge(nt, rows == TS ? sm[u] * (1.0f / TS) : sm[u] / nt, sq[u]);
                }
            }
        }
    }
    red_n[g][cl] = an; red_mu[g][cl] = amu; red_m2[g][cl] = am2;
    __syncthreads();
    if (g == 0 && c < C) {
        an = 0.f; amu = 0.f; am2 = 0.f;
#pragma unroll
        for (int y = 0; y < 16; ++y)
            if (red_n[y][cl] > 0.f) merge(red_n[y][cl], red_mu[y][cl], red_m2[y][cl]);
        const float mean = amu;
        const float var = am2 / (float)M;
        save[c] = mean;
        save[C + c] = 1.0f / sqrtf(var + eps);
        if (running_mean) {
            const float unb = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
        }
    }
}

// y = lrelu((x - mean[c]) * invstd[c] * gamma[c] + beta[c]); every workgroup stages the
// per-channel scale/shift in LDS, then streams its share of the rows with float4 accesses.
template <typename T>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T *__restrict__ x, const float *__restrict__ save,
                                                         const float *__restrict__ gamma,
                                                         const float *__restrict__ beta,
                                                         const T *__restrict__ residual, float slope, int M,
                                                         int C, T *__restrict__ y)
{
    extern __shared__ float ss[];          // scale[C], shift[C]
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float sc = gamma[c] * save[C + c];
        ss[c] = sc;
        ss[C + c] = beta[c] - save[c] * sc;
    }
    __syncthreads();
    if ((C & 3) == 0) {
        const long long total4 = (long long)M * C / 4;
        const int c4n = C / 4;
        for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4;
             i += (long long)gridDim.x * blockDim.x) {
            const int c = (int)(i % c4n) * 4;
            const float4 v = mpa_ld4<T>(x + 4 * i);
            const float4 sc = *reinterpret_cast<const float4 *>(ss + c);
            const float4 sh = *reinterpret_cast<const float4 *>(ss + C + c);
            float4 o;
            o.x = fmaf(v.x, sc.x, sh.x); o.y = fmaf(v.y, sc.y, sh.y);
            o.z = fmaf(v.z, sc.z, sh.z); o.w = fmaf(v.w, sc.w, sh.w);
            o.x = o.x > 0.f ? o.x : o.x * slope; o.y = o.y > 0.f ? o.y : o.y * slope;
            o.z = o.z > 0.f ? o.z : o.z * slope; o.w = o.w > 0.f ? o.w : o.w * slope;
            if (residual != nullptr) {
                const float4 r = mpa_ld4<T>(residual + 4 * i);
                o.x = r.x + o.x; o.y = r.y + o.y; o.z = r.z + o.z; o.w = r.w + o.w;
            }
            mpa_st4<T>(y + 4 * i, o);
        }
    } else {
        const long long total = (long long)M * C;
        for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
             i += (long long)gridDim.x * blockDim.x) {
            const int c = (int)(i % C);
            float t = fmaf(mpa_ld1<T>(x + i), ss[c], ss[C + c]);
            t = t > 0.f ? t : t * slope;
            mpa_st1<T>(y + i, residual != nullptr ? mpa_ld1<T>(residual + i) + t : t);
        }
    }
}


// y = residual + lrelu(bn(x)) with the BatchNorm statistics FINISHED IN THE PROLOGUE of every workgroup from the
// sums the GEMM epilogues accumulated (stats [R][3][C]: sum, within-tile M2, sum^2/rows per 64-row tile):
//   mean = S1/M,   M2 = S2 + (S3 - S1*mean)   (within tiles exactly, Chan's form; between tiles from the tile sums),
// biased variance M2/M -- no separate finalize launch.  Workgroup 0 also stores (mean, invstd) for the backward
// pass and updates the running statistics (momentum, unbiased variance) as nn.BatchNorm1d does.  Eval mode
// (training == 0) normalises with the running statistics.
// Prologue shared by the single and grouped forms: scale[C] | shift[C] of one unit into ss; workgroup `bx` == 0 stores
// (mean, invstd) and updates the running statistics.
__device__ __forceinline__ void bn_scale_shift(float *ss, const float *__restrict__ stats, int R, int training,
                                               float *__restrict__ running_mean, float *__restrict__ running_var,
                                               float momentum, float eps, long long *__restrict__ num_batches_tracked,
                                               const float *__restrict__ gamma, const float *__restrict__ beta, int M, int C,
                                               float *__restrict__ save, bool first)
{
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float mean, var;
        if (training) {
            float s1 = 0.f, s2 = 0.f, s3 = 0.f;
            for (int r = 0; r < R; ++r) {
                const float *p = stats + (size_t)r * 3 * C + c;
                s1 += p[0]; s2 += p[C]; s3 += p[2 * C];
            }
            mean = s1 / (float)M;
            var = fmaxf(0.f, (s2 + (s3 - s1 * mean)) / (float)M);
        } else {
            mean = running_mean[c];
            var = running_var[c];
        }
        const float invstd = 1.0f / sqrtf(var + eps);
        const float sc = gamma[c] * invstd;
        ss[c] = sc;
        ss[C + c] = beta[c] - mean * sc;
        if (first) {
            save[c] = mean;
            save[C + c] = invstd;
            if (training && running_mean != nullptr) {
                const float unb = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
            }
        }
    }
    if (first && threadIdx.x == 0 && training && num_batches_tracked != nullptr) *num_batches_tracked += 1;
}

// Grouped forms (mpa_bn_group_*): up to BN_GROUP_MAX independent units in one launch.
constexpr int BN_GROUP_MAX = 8;
struct BnUnit {
    const void *x;                 // [M][C] pre-normalisation rows (the GEMM's output)
    const float *stats;            // forward: [R][3][C] accumulated tile statistics
    float *running_mean, *running_var;
    long long *nbt;
    const float *gamma, *beta;
    const void *residual;          // forward (optional)
    void *y;                       // forward: output rows
    float *save;                   // [2][C] mean | invstd
    const void *gy;                // backward: upstream gradient rows (leading dimension ldg)
    float *partial;                // backward: [replicas][2][C] channel sums
    void *gx;                      // backward: gradient of x
    float *dgamma, *dbeta;
    int M, C, R, ldg, ldy, training, replicas;
    float momentum, eps, slope;
};
struct BnGroupArgs {
    BnUnit u[BN_GROUP_MAX];
    int count;
};

template <typename T>
__device__ __forceinline__ void bn_apply_rows(const float *ss, const T *__restrict__ x, const T *__restrict__ residual,
                                              float slope, int M, int C, T *__restrict__ y, int ldy, int bx, int gx)
{
    // y rows have leading dimension ldy (a column block of a wider tensor: the units of a group write side by side)
    if ((C & 3) == 0) {
        const long long total4 = (long long)M * C / 4;
        const int c4n = C / 4;
        for (long long i = bx * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gx * blockDim.x) {
            const long long row = i / c4n;
            const int c = (int)(i - row * c4n) * 4;
            const float4 v = mpa_ld4<T>(x + 4 * i);
            const float4 sc = *reinterpret_cast<const float4 *>(ss + c);
            const float4 sh = *reinterpret_cast<const float4 *>(ss + C + c);
            float4 o;
            o.x = fmaf(v.x, sc.x, sh.x); o.y = fmaf(v.y, sc.y, sh.y);
            o.z = fmaf(v.z, sc.z, sh.z); o.w = fmaf(v.w, sc.w, sh.w);
            o.x = o.x > 0.f ? o.x : o.x * slope; o.y = o.y > 0.f ? o.y : o.y * slope;
            o.z = o.z > 0.f ? o.z : o.z * slope; o.w = o.w > 0.f ? o.w : o.w * slope;
            if (residual != nullptr) {
                const float4 r = mpa_ld4<T>(residual + 4 * i);
                o.x = r.x + o.x; o.y = r.y + o.y; o.z = r.z + o.z; o.w = r.w + o.w;
            }
            mpa_st4<T>(y + row * ldy + c, o);
        }
    } else {
        const long long total = (long long)M * C;
        for (long long i = bx * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gx * blockDim.x) {
            const long long row = i / C;
            const int c = (int)(i - row * C);
            float t = fmaf(mpa_ld1<T>(x + i), ss[c], ss[C + c]);
            t = t > 0.f ? t : t * slope;
            mpa_st1<T>(y + row * ldy + c, residual != nullptr ? mpa_ld1<T>(residual + i) + t : t);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_stats_act_fwd_kernel(
    const T *__restrict__ x, const float *__restrict__ stats, int R, int training, float *__restrict__ running_mean,
    float *__restrict__ running_var, float momentum, float eps, long long *__restrict__ num_batches_tracked,
    const float *__restrict__ gamma, const float *__restrict__ beta, const T *__restrict__ residual, float slope, int M,
    int C, T *__restrict__ y, float *__restrict__ save)
{
    extern __shared__ float ss[];          // scale[C], shift[C]
    bn_scale_shift(ss, stats, R, training, running_mean, running_var, momentum, eps, num_batches_tracked, gamma, beta, M, C,
                   save, blockIdx.x == 0);
    __syncthreads();
    bn_apply_rows<T>(ss, x, residual, slope, M, C, y, C, blockIdx.x, gridDim.x);
}

// Grouped forward.  sum_mode == 0: blockIdx.y picks the unit, y_u = residual_u + lrelu(bn_u(x_u)).
// sum_mode != 0 (all units [M][C]): y_last = residual_0 + sum_u lrelu(bn_u(x_u)), added in unit order in fp32 --
// Fuse's accumulation over its source states in one pass over the rows (intermediate sums never reach memory).
template <typename T>
__global__ __launch_bounds__(256) void bn_group_fwd_kernel(BnGroupArgs a, int sum_mode)
{
    extern __shared__ float ss[];
    if (!sum_mode) {
        const BnUnit &u = a.u[blockIdx.y];
        bn_scale_shift(ss, u.stats, u.R, u.training, u.running_mean, u.running_var, u.momentum, u.eps, u.nbt, u.gamma,
                       u.beta, u.M, u.C, u.save, blockIdx.x == 0);
        __syncthreads();
        bn_apply_rows<T>(ss, static_cast<const T *>(u.x), static_cast<const T *>(u.residual), u.slope, u.M, u.C,
                         static_cast<T *>(u.y), u.ldy, blockIdx.x, gridDim.x);
        return;
    }
    const int n = a.count, C = a.u[0].C, M = a.u[0].M;
    for (int k = 0; k < n; ++k) {
        const BnUnit &u = a.u[k];
        bn_scale_shift(ss + (size_t)k * 2 * C, u.stats, u.R, u.training, u.running_mean, u.running_var, u.momentum, u.eps,
                       u.nbt, u.gamma, u.beta, M, C, u.save, blockIdx.x == 0);
    }
    __syncthreads();
    const T *res = static_cast<const T *>(a.u[0].residual);
    T *y = static_cast<T *>(a.u[n - 1].y);
    const long long total4 = (long long)M * C / 4;           // (host: C % 4 == 0)
    const int c4n = C / 4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        float4 acc = res != nullptr ? mpa_ld4<T>(res + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < n; ++k) {
            const float slope = a.u[k].slope;
            const float4 v = mpa_ld4<T>(static_cast<const T *>(a.u[k].x) + 4 * i);
            const float4 sc = *reinterpret_cast<const float4 *>(ss + (size_t)k * 2 * C + c);
            const float4 sh = *reinterpret_cast<const float4 *>(ss + (size_t)k * 2 * C + C + c);
            float4 o;
            o.x = fmaf(v.x, sc.x, sh.x); o.y = fmaf(v.y, sc.y, sh.y);
            o.z = fmaf(v.z, sc.z, sh.z); o.w = fmaf(v.w, sc.w, sh.w);
            o.x = o.x > 0.f ? o.x : o.x * slope; o.y = o.y > 0.f ? o.y : o.y * slope;
            o.z = o.z > 0.f ? o.z : o.z * slope; o.w = o.w > 0.f ? o.w : o.w * slope;
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
        mpa_st4<T>(y + 4 * i, acc);
    }
}

// out[c] += sum over the M rows of x[r*ld + c]   (bias gradients; `out` cleared by the caller)
__global__ __launch_bounds__(256) void col_sum_kernel(const float *__restrict__ x, int M, int C, int ld, int cpb,
                                                      int rows_per_block, float *__restrict__ out)
{
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int ry = tid / cpb, cl = tid - ry * cpb, RY = blockDim.x / cpb;
    const int c = blockIdx.y * cpb + cl;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < C) {
        int r = r0 + ry;
        for (; r + 3 * RY < r1; r += 4 * RY) {
            a0 += x[(size_t)r * ld + c];
            a1 += x[(size_t)(r + RY) * ld + c];
            a2 += x[(size_t)(r + 2 * RY) * ld + c];
            a3 += x[(size_t)(r + 3 * RY) * ld + c];
        }
        for (; r < r1; r += RY) a0 += x[(size_t)r * ld + c];
    }
    red[tid] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ry == 0 && c < C) {
        float s = red[cl];
        for (int y = 1; y < RY; ++y) s += red[y * cpb + cl];
        atomicAdd(out + c, s);
    }
}


// out[g][c] = sum over the R rows of group g of x[(g*R + r)*ld + c]: the gradient of a per-cloud row that was
// broadcast over the cloud's points (the global-feature / label-embedding columns of the part-seg head,
// reference modules/pointnet2_utils.py:846-856).  One workgroup per (group, 64 columns): 4 row lanes x 64
// column lanes, fp32 accumulation, combined through LDS, plain stores -- no atomics, nothing to clear, and no
// multi-workgroup reduction (torch's own, which autograd would use for expand(), is what goes wrong under
// HIP-graph replay on this stack: DESIGN.md section 5).
template <typename T>
__global__ __launch_bounds__(256) void group_col_sum_kernel(const T *__restrict__ x, int R, int C, int ld,
                                                            float *__restrict__ out)
{
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int g = blockIdx.y, c = blockIdx.x * 64 + cl;
    const T *base = x + (size_t)g * R * ld;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < C) {
        int r = rl;
        // 16 rows in flight per lane (4 made the 2048 rows of a part-seg cloud 128 dependent round trips: 68 us)
        for (; r + 60 < R; r += 64) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = mpa_ld1<T>(base + (size_t)(r + 4 * u) * ld + c);
#pragma unroll
            for (int u = 0; u < 16; u += 4) { a0 += v[u]; a1 += v[u + 1]; a2 += v[u + 2]; a3 += v[u + 3]; }
        }
        for (; r < R; r += 4) a0 += mpa_ld1<T>(base + (size_t)r * ld + c);
    }
    red[rl][cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (rl == 0 && c < C) out[(size_t)g * C + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

// Column reductions over the M rows of an [M,C] tensor.  A workgroup owns a slab of rows and
// a group of <= 256 channels; its 256 lanes are arranged [RY][CPB] (channel fastest: coalesced
// rows), each lane walks its rows 4 at a time (independent loads in flight), the RY partials
// are combined through LDS and one float atomic per (workgroup, channel) goes to HBM.
template <typename F>
__device__ __forceinline__ void slab_reduce2(int M, int C, int cpb, int rows_per_block, float *out0, float *out1, F f)
{
    __shared__ float red[2][256];
    const int tid = threadIdx.x;
    const int ry = tid / cpb, cl = tid - ry * cpb, RY = blockDim.x / cpb;
    const int c = blockIdx.y * cpb + cl;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    float a0 = 0.f, a1 = 0.f;
    if (c < C && ry < RY) {
        int r = r0 + ry;
        for (; r + 3 * RY < r1; r += 4 * RY) {
            float p0 = 0.f, p1 = 0.f, q0 = 0.f, q1 = 0.f, s0 = 0.f, s1 = 0.f, t0 = 0.f, t1 = 0.f;
            f(r, c, p0, p1); f(r + RY, c, q0, q1); f(r + 2 * RY, c, s0, s1); f(r + 3 * RY, c, t0, t1);
            a0 += (p0 + q0) + (s0 + t0);
            a1 += (p1 + q1) + (s1 + t1);
        }
        for (; r < r1; r += RY) f(r, c, a0, a1);
    }
    red[0][tid] = a0;
    red[1][tid] = a1;
    __syncthreads();
    if (ry == 0 && c < C) {
        for (int y = 1; y < RY; ++y) { a0 += red[0][y * cpb + cl]; a1 += red[1][y * cpb + cl]; }
        atomicAdd(out0 + c, a0);
        atomicAdd(out1 + c, a1);
    }
}

// Backward pass 1: g = grad_y * lrelu'(bn(x)); per channel sum(g) and sum(g*xhat).
template <typename T>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(
    const T *__restrict__ x, const T *__restrict__ gy, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ gamma, const float *__restrict__ beta, float slope,
    int M, int C, int ldg, int cpb, int rows_per_block, float *__restrict__ partial, int replicas)
{
    float *sum_g = partial + (size_t)(blockIdx.x % replicas) * 2 * C;
    float *sum_gx = sum_g + C;
    slab_reduce2(M, C, cpb, rows_per_block, sum_g, sum_gx, [&](int r, int c, float &sg, float &sgx) {
        const float xh = (mpa_ld1<T>(x + (size_t)r * C + c) - mean[c]) * invstd[c];
        const float t = xh * gamma[c] + beta[c];
        float g = mpa_ld1<T>(gy + (size_t)r * ldg + c);
        g = t > 0.f ? g : g * slope;
        sg += g;
        sgx = fmaf(g, xh, sgx);
    });
}

// float4 form (C % 4 == 0): a lane owns 4 adjacent channels, 256 lanes = RY rows x C/4 lanes, two
// rows (4 x 16-B loads) in flight per lane; RY partials combined through LDS, one atomic per
// (workgroup, channel).
template <typename T>
__device__ __forceinline__ void bn_bwd_reduce4_body(
    const T *__restrict__ x, const T *__restrict__ gy, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ gamma, const float *__restrict__ beta, float slope,
    int M, int C, int ldg, int lanes_per_row, int rows_per_block, float *__restrict__ partial, int replicas, int bx,
    int by)
{
    // partial [replicas][2][C] (pre-zeroed): workgroups spread their atomics over the replicas so
    // that at most gridDim.x/replicas of them add into one address; the apply pass sums them.
    float *__restrict__ sum_g = partial + (size_t)(bx % replicas) * 2 * C;
    float *__restrict__ sum_gx = sum_g + C;
    __shared__ float4 red[2][256];
    const int tid = threadIdx.x;
    const int ry = tid / lanes_per_row, cl = tid - ry * lanes_per_row, RY = 256 / lanes_per_row;
    const int c = (by * lanes_per_row + cl) * 4;
    const int r0 = bx * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sx = sg;
    if (c < C && ry < RY) {
        const float4 mu = *reinterpret_cast<const float4 *>(mean + c), is = *reinterpret_cast<const float4 *>(invstd + c);
        const float4 ga = *reinterpret_cast<const float4 *>(gamma + c), be = *reinterpret_cast<const float4 *>(beta + c);
        auto acc1 = [&](const float4 xv, const float4 gv) {
            float xh, t, g;
            xh = (xv.x - mu.x) * is.x; t = xh * ga.x + be.x; g = t > 0.f ? gv.x : gv.x * slope; sg.x += g; sx.x = fmaf(g, xh, sx.x);
            xh = (xv.y - mu.y) * is.y; t = xh * ga.y + be.y; g = t > 0.f ? gv.y : gv.y * slope; sg.y += g; sx.y = fmaf(g, xh, sx.y);
            xh = (xv.z - mu.z) * is.z; t = xh * ga.z + be.z; g = t > 0.f ? gv.z : gv.z * slope; sg.z += g; sx.z = fmaf(g, xh, sx.z);
            xh = (xv.w - mu.w) * is.w; t = xh * ga.w + be.w; g = t > 0.f ? gv.w : gv.w * slope; sg.w += g; sx.w = fmaf(g, xh, sx.w);
        };
        int r = r0 + ry;
        for (; r + RY < r1; r += 2 * RY) {
            const float4 x0 = mpa_ld4<T>(x + (size_t)r * C + c);
            const float4 g0 = mpa_ld4<T>(gy + (size_t)r * ldg + c);
            const float4 x1 = mpa_ld4<T>(x + (size_t)(r + RY) * C + c);
            const float4 g1 = mpa_ld4<T>(gy + (size_t)(r + RY) * ldg + c);
            acc1(x0, g0);
            acc1(x1, g1);
        }
        for (; r < r1; r += RY)
            acc1(mpa_ld4<T>(x + (size_t)r * C + c), mpa_ld4<T>(gy + (size_t)r * ldg + c));
    }
    red[0][tid] = sg;
    red[1][tid] = sx;
    __syncthreads();
    // consecutive lanes add consecutive channels: float atomics want contiguous 256-B segments
    const float *rf = reinterpret_cast<const float *>(&red[0][0]);
    const int cw = lanes_per_row * 4;                    // channels covered by this workgroup
    for (int t = tid; t < 2 * cw; t += 256) {
        const int which = t / cw, ch = t - which * cw;
        const int cg = by * cw + ch;
        if (cg < C) {
            float a = 0.f;
            for (int y = 0; y < RY; ++y) a += rf[(size_t)which * 1024 + (y * lanes_per_row + ch / 4) * 4 + (ch & 3)];
            atomicAdd((which ? sum_gx : sum_g) + cg, a);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce4_kernel(
    const T *__restrict__ x, const T *__restrict__ gy, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ gamma, const float *__restrict__ beta, float slope,
    int M, int C, int ldg, int lanes_per_row, int rows_per_block, float *__restrict__ partial, int replicas)
{
    bn_bwd_reduce4_body<T>(x, gy, mean, invstd, gamma, beta, slope, M, C, ldg, lanes_per_row, rows_per_block, partial,
                           replicas, blockIdx.x, blockIdx.y);
}

// host-side geometry of the float4 reduce: lanes per row, channel blocks, rows per workgroup, row blocks
struct BnReduceGeom { int lanes, gy, rpb, gx; };
inline BnReduceGeom bn_reduce_geom(int M, int C)
{
    BnReduceGeom g;
    int lanes = C / 4;                                   // lanes per row: a divisor of 256
    lanes = lanes >= 256 ? 256 : (lanes > 128 ? 256 : (lanes > 64 ? 128 : (lanes > 32 ? 64 : (lanes > 16 ? 32 : 16))));
    g.lanes = lanes;
    g.gy = mpa_ceil_div(C / 4, lanes);
    const int ry = 256 / lanes;
    int want = 1024 / g.gy;
    g.rpb = mpa_ceil_div(M, want < 1 ? 1 : want);
    if (g.rpb < 2 * ry) g.rpb = 2 * ry;
    g.gx = mpa_ceil_div(M, g.rpb);
    return g;
}

// Grouped backward pass 1: blockIdx.z picks the unit; workgroups outside that unit's own grid leave.
struct BnGroupGeom { int lanes[BN_GROUP_MAX], gy[BN_GROUP_MAX], rpb[BN_GROUP_MAX], gx[BN_GROUP_MAX]; };
template <typename T>
__global__ __launch_bounds__(256) void bn_group_bwd_reduce_kernel(BnGroupArgs a, BnGroupGeom g)
{
    const int k = blockIdx.z;
    if ((int)blockIdx.x >= g.gx[k] || (int)blockIdx.y >= g.gy[k]) return;
    const BnUnit &u = a.u[k];
    bn_bwd_reduce4_body<T>(static_cast<const T *>(u.x), static_cast<const T *>(u.gy), u.save, u.save + u.C, u.gamma, u.beta,
                           u.slope, u.M, u.C, u.ldg, g.lanes[k], g.rpb[k], u.partial, u.replicas, blockIdx.x, blockIdx.y);
}

// Backward pass 2: grad_x = gamma*invstd*(g - sum_g/M - xhat*sum_gx/M)   (batch statistics)
//                  grad_x = gamma*invstd*g                                (running statistics)
template <typename T>
__device__ __forceinline__ void bn_bwd_apply_body(
    float *cs, const T *__restrict__ x, const T *__restrict__ gy, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ gamma, const float *__restrict__ beta,
    const float *__restrict__ partial, int replicas, float slope, int use_batch_stats, int M, int C, int ldg,
    long long total, T *__restrict__ gx, float *__restrict__ dgamma, float *__restrict__ dbeta, int bx, int ngx)
{
    // per-channel constants in LDS: k1 = gamma*invstd, then grad_x = k1*(g - a - xhat*b) with
    // a = sum_g/M, b = sum_gxhat/M (zero in eval mode); totals of the replicas also go out as
    // dbeta / dgamma (workgroup 0).
    // cs [6][C]: mean, invstd, gamma, beta, a, b
    const float invM = 1.0f / (float)M;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float sg = 0.f, sx = 0.f;
        if (replicas == 8) {            // the host side's default: all 16 loads in flight at once
            float pg[8], px[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                pg[r] = partial[(size_t)r * 2 * C + c];
                px[r] = partial[(size_t)r * 2 * C + C + c];
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) { sg += pg[r]; sx += px[r]; }
        } else {
            for (int r = 0; r < replicas; ++r) {
                sg += partial[(size_t)r * 2 * C + c];
                sx += partial[(size_t)r * 2 * C + C + c];
            }
        }
        cs[c] = mean[c]; cs[C + c] = invstd[c]; cs[2 * C + c] = gamma[c]; cs[3 * C + c] = beta[c];
        cs[4 * C + c] = use_batch_stats ? sg * invM : 0.f;
        cs[5 * C + c] = use_batch_stats ? sx * invM : 0.f;
        if (bx == 0) {
            if (dbeta) dbeta[c] = sg;
            if (dgamma) dgamma[c] = sx;
        }
    }
    __syncthreads();
    const bool v4 = ((C & 3) == 0) && ((ldg & 3) == 0) &&
                    (((((uintptr_t)x | (uintptr_t)gy | (uintptr_t)gx)) & mpa_vec4_align<T>::mask) == 0);
    if (v4) {
        const long long total4 = total / 4;
        const int c4n = C / 4;
        for (long long i = bx * (long long)blockDim.x + threadIdx.x; i < total4;
             i += (long long)ngx * blockDim.x) {
            const long long row = i / c4n;
            const int c = (int)(i - row * c4n) * 4;
            const float4 xv = mpa_ld4<T>(x + 4 * i);
            const float4 gv = mpa_ld4<T>(gy + row * ldg + c);
            const float xin[4] = {xv.x, xv.y, xv.z, xv.w}, gin[4] = {gv.x, gv.y, gv.z, gv.w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float is = cs[C + c + j], ga = cs[2 * C + c + j];
                const float xh = (xin[j] - cs[c + j]) * is;
                const float t = xh * ga + cs[3 * C + c + j];
                const float g = t > 0.f ? gin[j] : gin[j] * slope;
                o[j] = ga * is * (g - cs[4 * C + c + j] - xh * cs[5 * C + c + j]);
            }
            mpa_st4<T>(gx + 4 * i, make_float4(o[0], o[1], o[2], o[3]));
        }
        return;
    }
    for (long long i = bx * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)ngx * blockDim.x) {
        const long long row = i / C;
        const int c = (int)(i - row * C);
        const float is = cs[C + c], ga = cs[2 * C + c];
        const float xh = (mpa_ld1<T>(x + i) - cs[c]) * is;
        const float t = xh * ga + cs[3 * C + c];
        const float g0 = mpa_ld1<T>(gy + row * ldg + c);
        const float g = t > 0.f ? g0 : g0 * slope;
        mpa_st1<T>(gx + i, ga * is * (g - cs[4 * C + c] - xh * cs[5 * C + c]));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(
    const T *__restrict__ x, const T *__restrict__ gy, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ gamma, const float *__restrict__ beta,
    const float *__restrict__ partial, int replicas, float slope, int use_batch_stats, int M, int C, int ldg,
    long long total, T *__restrict__ gx, float *__restrict__ dgamma, float *__restrict__ dbeta)
{
    extern __shared__ float cs[];
    bn_bwd_apply_body<T>(cs, x, gy, mean, invstd, gamma, beta, partial, replicas, slope, use_batch_stats, M, C, ldg, total,
                         gx, dgamma, dbeta, blockIdx.x, gridDim.x);
}

// Grouped backward pass 2: blockIdx.y picks the unit.
template <typename T>
__global__ __launch_bounds__(256) void bn_group_bwd_apply_kernel(BnGroupArgs a)
{
    extern __shared__ float cs[];
    const BnUnit &u = a.u[blockIdx.y];
    bn_bwd_apply_body<T>(cs, static_cast<const T *>(u.x), static_cast<const T *>(u.gy), u.save, u.save + u.C, u.gamma, u.beta,
                         u.partial, u.replicas, u.slope, u.training, u.M, u.C, u.ldg, (long long)u.M * u.C,
                         static_cast<T *>(u.gx), u.dgamma, u.dbeta, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256) void col_stats_kernel(const float *__restrict__ x, int M, int C, int cpb,
                                                        int rows_per_block, float *__restrict__ col_sum,
                                                        float *__restrict__ col_sumsq)
{
    slab_reduce2(M, C, cpb, rows_per_block, col_sum, col_sumsq, [&](int r, int c, float &s, float &q) {
        const float v = x[(size_t)r * C + c];
        s += v;
        q = fmaf(v, v, q);
    });
}

inline int ew_grid(long long total)
{
    long long g = (total + EW_TPB - 1) / EW_TPB;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

inline void slab_grid(int M, int C, dim3 &grid, int &cpb, int &rows_per_block)
{
    cpb = C >= 256 ? 256 : (C > 128 ? 256 : (C > 64 ? 128 : 64));   // channels per workgroup (divides 256)
    int gy = mpa_ceil_div(C, cpb);
    int ry = 256 / cpb;
    // ~1024 workgroups over the chip, every lane walking >= 8 rows
    int want = 1024 / gy;
    rows_per_block = mpa_ceil_div(M, want < 1 ? 1 : want);
    if (rows_per_block < 8 * ry) rows_per_block = 8 * ry;
    grid = dim3(mpa_ceil_div(M, rows_per_block), gy);
}

}  // namespace

// forward declaration (defined with the other column reductions below)
static int launch_col_stats(const float *x, int M, int C, float *col_sum, float *col_sumsq, hipStream_t st);

namespace {
__global__ __launch_bounds__(256) void clear_f32_kernel(float *__restrict__ p, long long n)
{
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = 0.f;
}
}  // namespace

extern "C" int mpa_gemm_f32(const float *A, int lda, int transA, const float *B, int ldb, int transB,
                            const float *bias, float *C, int ldc, int M, int N, int K, int accumulate,
                            float *tile_stats, int stats_replicas, float *a_col_sum, float *workspace,
                            size_t workspace_bytes, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || lda <= 0 || ldb <= 0 || ldc < N) return MPA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int vecA = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (lda % 4 == 0);
    const int vecB = ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && (ldb % 4 == 0);

    // short-K whole-tile products (see gemm_shortk_kernel)
    static const bool shortk_on = getenv("MPA_GEMM_NO_SHORTK") == nullptr;
    if (shortk_on && !transA && !accumulate && a_col_sum == nullptr && vecA && vecB && (M % TS) == 0 && (N % TS) == 0 &&
        (K == KS || K == 2 * KS)) {
        const dim3 grid((M / TS) * (N / TS));
        if (K == KS) {
            if (transB)
                hipLaunchKernelGGL((gemm_shortk_kernel<true, 1>), grid, dim3(NT), 32768, st, A, lda, B, ldb, bias, C, ldc,
                                   M, N, tile_stats, stats_replicas);
            else
                hipLaunchKernelGGL((gemm_shortk_kernel<false, 1>), grid, dim3(NT), 32768, st, A, lda, B, ldb, bias, C, ldc,
                                   M, N, tile_stats, stats_replicas);
        } else {
            if (transB)
                hipLaunchKernelGGL((gemm_shortk_kernel<true, 2>), grid, dim3(NT), 65536, st, A, lda, B, ldb, bias, C, ldc,
                                   M, N, tile_stats, stats_replicas);
            else
                hipLaunchKernelGGL((gemm_shortk_kernel<false, 2>), grid, dim3(NT), 65536, st, A, lda, B, ldb, bias, C, ldc,
                                   M, N, tile_stats, stats_replicas);
        }
        MPA_LAUNCH_CHECK();
        return MPA_OK;
    }

    // split K across workgroups when the output alone cannot fill the chip (weight gradients:
    // K = B*S rows; the narrow head layers): aim at ~512 workgroups, >= 256 k per workgroup.
    const long long ntiles = (long long)mpa_ceil_div(M, TS) * mpa_ceil_div(N, TS);
    const size_t mn = (size_t)M * N;
    int splits = 1;
    if (ntiles < 256 && K >= 512 && ldc == N) {
        splits = (int)((512 + ntiles - 1) / ntiles);
        if (splits > K / 256) splits = K / 256;
        const bool ws_ok = workspace != nullptr && ((reinterpret_cast<uintptr_t>(workspace) & 15) == 0);
        if (ws_ok) {
            size_t fit = workspace_bytes / (mn * sizeof(float));
            if ((size_t)splits > fit) splits = (int)fit;
        } else if (splits > 16) {
            splits = 16;          // atomic fallback: bound the adders per address
        }
        if (splits < 1) splits = 1;
    }
    if (const char *e = getenv("MPA_GEMM_SPLITS")) splits = atoi(e);      // development override
    int kchunk = mpa_ceil_div(mpa_ceil_div(K, splits), KS) * KS;
    splits = mpa_ceil_div(K, kchunk);

    int out_mode = accumulate ? 1 : 0;
    float *dst = C;
    bool reduce_after = false;
    if (splits > 1) {
        if (workspace != nullptr && workspace_bytes >= (size_t)splits * mn * sizeof(float)) {
            out_mode = 2;
            dst = workspace;
            reduce_after = true;
        } else {
            if (!accumulate) {
                // a plain kernel, not hipMemsetAsync: no memset nodes inside a captured HIP graph (as gather.hip, diffattn.hip)
                const long long n = (long long)mn;
                long long g = (n + 255) / 256;
                hipLaunchKernelGGL(clear_f32_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, st, C, n);
                MPA_LAUNCH_CHECK();
            }
            out_mode = 1;
        }
    }
    float *zero_c = (reduce_after && !accumulate) ? C : nullptr;
    const bool stats_after = splits > 1 && tile_stats != nullptr;   // partial sums carry no statistics
    float *stats = stats_after ? nullptr : tile_stats;
    int rc;
    if (transA && transB)
        rc = launch_gemm<true, true>(A, lda, B, ldb, bias, dst, ldc, M, N, K, splits, kchunk, out_mode, vecA, vecB,
                                     stats, zero_c, a_col_sum, stats_replicas, st);
    else if (transA)
        rc = launch_gemm<true, false>(A, lda, B, ldb, bias, dst, ldc, M, N, K, splits, kchunk, out_mode, vecA, vecB,
                                      stats, zero_c, a_col_sum, stats_replicas, st);
    else if (transB)
        rc = launch_gemm<false, true>(A, lda, B, ldb, bias, dst, ldc, M, N, K, splits, kchunk, out_mode, vecA, vecB,
                                      stats, zero_c, a_col_sum, stats_replicas, st);
    else
        rc = launch_gemm<false, false>(A, lda, B, ldb, bias, dst, ldc, M, N, K, splits, kchunk, out_mode, vecA, vecB,
                                       stats, zero_c, a_col_sum, stats_replicas, st);
    if (rc != MPA_OK) return rc;
    if (reduce_after) {
        const int gx = mpa_ceil_div((long long)mn, 256);
        int gy = mpa_ceil_div(1024, gx);                 // ~1024 workgroups in total
        gy = gy > 32 ? 32 : gy;
        gy = gy > splits ? splits : gy;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(gx, gy), dim3(256), 0, st, workspace, splits, (long long)mn, C);
    }
    if (stats_after) {
        if (accumulate) return MPA_EUNSUPPORTED;
        hipLaunchKernelGGL(tile_stats_kernel, dim3(mpa_ceil_div(N, 64), mpa_ceil_div(M, TS)), dim3(256), 0, st, C, M, N,
                           tile_stats, stats_replicas);
    }
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_geo_rider_f32(const MpaGeoRider *rider, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!rider) return MPA_EINVAL;
    const int rc = rider_launch_alone(*rider, (hipStream_t)stream);
    if (rc != MPA_OK) return rc;
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_gemm_tn_grouped_f32(const MpaGemmTnProblem *problems, int count, float *workspace,
                                       size_t workspace_bytes, void *stream)
{
    return mpa_gemm_tn_grouped_rider_f32(problems, count, workspace, workspace_bytes, nullptr, 0, stream);
}

extern "C" int mpa_gemm_tn_grouped_rider_f32(const MpaGemmTnProblem *problems, int count, float *workspace,
                                             size_t workspace_bytes, const MpaGeoRider *riders, int nriders,
                                             void *stream)
{
    MPA_CLEAR_ERROR();
    if (!problems || count <= 0 || nriders < 0 || (nriders > 0 && !riders)) return MPA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int launch_no = 0;
    size_t ws_used = 0;
    int done = 0;
    // more problems than one launch's argument block holds: equal-sized launches, in queue order (dealing the
    // fine states' HBM-bound streams out over the launches measured 1-2 % slower: they run best back to back)
    const int launches = mpa_ceil_div(count, GROUP_MAX), chunk = mpa_ceil_div(count, launches);
    // A queue of a few long, thin products (bf16 features: only the fp32 coordinate units' gradients come here -- rows
    // of 3 floats, up to 65,536 of them) would stream on a few dozen workgroups at 2048-row chunks (107 us for 17 MB):
    // when the whole call stays under four workgroups per CU that way, its thin products are cut four times finer.
    long long est = 0;
    for (int i = 0; i < count; ++i) {
        const long long tiles = (long long)mpa_ceil_div(problems[i].M, TS) * mpa_ceil_div(problems[i].N, TS);
        long long sp = tiles < 128 && problems[i].K >= 4096 ? (256 + tiles - 1) / tiles : 1;
        if (sp > problems[i].K / 2048) sp = problems[i].K / 2048 > 0 ? problems[i].K / 2048 : 1;
        est += tiles * sp;
    }
    const bool small_launch = est < 1024;
    while (done < count) {
        GroupedArgs ga;
        GroupedReduceArgs ra;
        int n = 0, nr = 0, blocks = 0, rblocks = 0;
        ga.block_start[0] = 0;
        ra.block_start[0] = 0;
        for (; done < count && n < chunk; ++done) {
            const MpaGemmTnProblem &in = problems[done];
            if (!in.A || !in.B || !in.out || in.M <= 0 || in.N <= 0 || in.K <= 0 || in.lda < in.M || in.ldb < in.N)
                return MPA_EINVAL;
            GroupedProblem &q = ga.p[n];
            q.A = in.A; q.B = in.B; q.out = in.out; q.a_col_sum = in.a_col_sum;
            q.lda = in.lda; q.ldb = in.ldb; q.M = in.M; q.N = in.N; q.K = in.K;
            q.tiles = mpa_ceil_div(in.M, TS) * mpa_ceil_div(in.N, TS);
            const size_t mn = (size_t)in.M * in.N;
            static const int tn_stream = getenv("MPA_TN_STREAM") ? atoi(getenv("MPA_TN_STREAM")) : 2;
            static const int target_wgs = getenv("MPA_TN_WGS") ? atoi(getenv("MPA_TN_WGS")) : 256;
            static const int min_kchunk_env = getenv("MPA_TN_KCHUNK") ? atoi(getenv("MPA_TN_KCHUNK")) : 2048;
            const int min_kchunk = small_launch ? 512 : min_kchunk_env;
            // tn_stream: 1 = stream every problem, 0 = LDS-staged, 2 = stream the single-/few-tile
            // problems (pure HBM streams) and stage the wide ones (MFMA-bound) through LDS
            q.stream = tn_stream == 2 ? (q.tiles <= 4 ? 1 : 0) : tn_stream;
            q.vec = ((in.lda & 3) == 0 && (reinterpret_cast<uintptr_t>(in.A) & 15) == 0 ? 1 : 0) |
                    ((in.ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(in.B) & 15) == 0 ? 2 : 0);
            int splits = 1;
            if (q.tiles < target_wgs / 2 && in.K >= 2 * min_kchunk) {
                splits = (target_wgs + q.tiles - 1) / q.tiles;
                if (splits > in.K / min_kchunk) splits = in.K / min_kchunk;
                const size_t room = workspace ? (workspace_bytes - ws_used) / (mn * sizeof(float)) : 0;
                if ((size_t)splits > room) splits = (int)room;
                if (splits < 1) splits = 1;
            }
            q.kchunk = mpa_ceil_div(mpa_ceil_div(in.K, splits), KS) * KS;
            q.splits = mpa_ceil_div(in.K, q.kchunk);
            q.slab = nullptr;
            if (q.splits > 1) {
                q.slab = workspace + ws_used / sizeof(float);
                ws_used += (size_t)q.splits * mn * sizeof(float);
                ws_used = (ws_used + 255) & ~(size_t)255;
                auto &r = ra.p[nr];
                r.slab = q.slab; r.out = q.out; r.mn = (int)mn; r.splits = q.splits;
                r.gx = mpa_ceil_div((long long)mn, 256);
                int gy = mpa_ceil_div(1024, r.gx);
                gy = gy > 32 ? 32 : gy;
                r.gy = gy > q.splits ? q.splits : gy;
                rblocks += r.gx * r.gy;
                ra.block_start[++nr] = rblocks;
            } else {
                q.kchunk = mpa_ceil_div(in.K, KS) * KS;
            }
            blocks += q.tiles * q.splits;
            ga.block_start[++n] = blocks;
        }
        ga.count = n;
        ra.count = nr;
        if (launch_no < nriders && riders[launch_no].queue == nullptr) {
            const int rc = rider_launch_alone(riders[launch_no], st);         // no queue words: not carried
            if (rc != MPA_OK) return rc;
            hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3(blocks), dim3(NT), gemm_lds_bytes(true, false), st, ga);
        } else if (launch_no < nriders) {
            // this launch carries rider `launch_no` (riders are in dependency order: one per launch, in stream order)
            RiderArgs rd;
            size_t rlds = 0;
            const int rc = rider_prepare(riders[launch_no], rd, rlds);
            if (rc != MPA_OK) return rc;
            size_t lds = gemm_lds_bytes(true, false);
            if (rlds > lds) {
                lds = rlds;
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tn_grouped_rider_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
                    hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tn_grouped_fps_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                    return MPA_EHIP;
            }
            // 2 workgroups of this kernel per CU (200 registers, >= 64 KiB of LDS): 64 per XCD
            RiderPlace pl;
            pl.slots = 64;
            // sampling workgroups per XCD lane: `MPA_RIDER_PER_LANE` (development) or one per CU (32), the lane's
            // other ids park
            static const int per_lane_env = getenv("MPA_RIDER_PER_LANE") ? atoi(getenv("MPA_RIDER_PER_LANE")) : 32;
            pl.per_lane = per_lane_env < 1 ? 1 : (per_lane_env > pl.slots ? pl.slots : per_lane_env);
            pl.nx = rd.fps_blocks > 0 ? mpa_ceil_div(rd.fps_blocks, pl.per_lane) : 0;
            pl.region = 8 * pl.slots;
            const int work = rd.sblocks + blocks;
            int grid;
            if (pl.nx == 0 || pl.nx > 7) {           // no sampling (or more clouds than 7 lanes hold): plain workers
                pl.nx = 0; pl.region = 0;
                if (rd.fps_blocks > 0) {
                    const int rc2 = rider_launch_alone(riders[launch_no], st);
                    if (rc2 != MPA_OK) return rc2;
                    rd.sblocks = 0;
                }
                grid = rd.sblocks + blocks;
            } else {
                // exactly one resident set of workgroups (2 per CU): the rider's on its XCD lanes, persistent workers on
                // the others, each walking the items with a fixed stride.  (Workgroup ids beyond the resident set would
                // queue behind the rider's lanes: the dispatcher places ids in order and id % 8 picks the XCD, so the
                // first id that must wait for a sampling workgroup to leave holds up every id after it -- 758 us per
                // launch, measured.)
                grid = pl.region;
            }
            pl.workers = grid - pl.slots * pl.nx;
            if (rd.sblocks == 0 && pl.nx > 0)
                hipLaunchKernelGGL(gemm_tn_grouped_fps_kernel, dim3(grid), dim3(NT), lds, st, ga, rd, pl, blocks);
            else
                hipLaunchKernelGGL(gemm_tn_grouped_rider_kernel, dim3(grid), dim3(NT), lds, st, ga, rd, pl, blocks);
        } else {
            hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3(blocks), dim3(NT), gemm_lds_bytes(true, false), st, ga);
        }
        ++launch_no;
        if (nr > 0) hipLaunchKernelGGL(splitk_reduce_grouped_kernel, dim3(rblocks), dim3(256), 0, st, ra);
    }
    for (; launch_no < nriders; ++launch_no) {            // more riders than launches: the rest go out on their own
        const int rc = rider_launch_alone(riders[launch_no], st);
        if (rc != MPA_OK) return rc;
    }
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

static int launch_col_stats(const float *x, int M, int C, float *col_sum, float *col_sumsq, hipStream_t st)
{
    dim3 grid;
    int cpb, rpb;
    slab_grid(M, C, grid, cpb, rpb);
    hipLaunchKernelGGL(col_stats_kernel, grid, dim3(256), 0, st, x, M, C, cpb, rpb, col_sum, col_sumsq);
    return MPA_OK;
}

extern "C" int mpa_col_stats_f32(const float *x, int M, int C, float *col_sum, float *col_sumsq, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!x || !col_sum || !col_sumsq || M <= 0 || C <= 0) return MPA_EINVAL;
    launch_col_stats(x, M, C, col_sum, col_sumsq, (hipStream_t)stream);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_tile_stats_f32(const float *x, int M, int C, float *tile_stats, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!x || !tile_stats || M <= 0 || C <= 0) return MPA_EINVAL;
    hipLaunchKernelGGL(tile_stats_kernel, dim3(mpa_ceil_div(C, 64), mpa_ceil_div(M, TS)), dim3(256), 0,
                       (hipStream_t)stream, x, M, C, tile_stats, 0);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_bn_finalize_f32(const float *tile_stats, int M, int C, float *running_mean, float *running_var,
                                   int training, float momentum, float eps, float *save_mean_invstd, float *zero_buf,
                                   int zero_count, int64_t *num_batches_tracked, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!save_mean_invstd || M <= 0 || C <= 0) return MPA_EINVAL;
    if (training && !tile_stats) return MPA_EINVAL;
    if (!training && (!running_mean || !running_var)) return MPA_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(mpa_ceil_div(C, 64)), dim3(1024), 0, (hipStream_t)stream, tile_stats, M,
                       C, running_mean, running_var, training, momentum, eps, save_mean_invstd, zero_buf,
                       zero_buf ? zero_count : 0, reinterpret_cast<long long *>(num_batches_tracked));
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_col_sum_f32(const float *x, int M, int C, int ld, float *out, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!x || !out || M <= 0 || C <= 0 || ld < C) return MPA_EINVAL;
    dim3 grid;
    int cpb, rpb;
    slab_grid(M, C, grid, cpb, rpb);
    hipLaunchKernelGGL(col_sum_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, M, C, ld, cpb, rpb, out);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

template <typename T>
static int bn_act_fwd_any(const T *x, const float *save_mean_invstd, const float *gamma, const float *beta,
                          const T *residual, float slope, int M, int C, T *y, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!x || !save_mean_invstd || !gamma || !beta || !y || M <= 0 || C <= 0) return MPA_EINVAL;
    if (C > 8192) return MPA_EUNSUPPORTED;
    const bool al = (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & mpa_vec4_align<T>::mask) == 0;
    if ((C & 3) == 0 && !al) return MPA_EUNSUPPORTED;
    long long total = (long long)M * C;
    hipLaunchKernelGGL(bn_act_fwd_kernel<T>, dim3(ew_grid(total / 4 + 1)), dim3(EW_TPB), 2 * C * sizeof(float),
                       (hipStream_t)stream, x, save_mean_invstd, gamma, beta, residual, slope, M, C, y);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_bn_act_fwd_f32(const float *x, const float *save_mean_invstd, const float *gamma,
                                  const float *beta, const float *residual, float slope, int M, int C, float *y,
                                  void *stream)
{
    return bn_act_fwd_any<float>(x, save_mean_invstd, gamma, beta, residual, slope, M, C, y, stream);
}

extern "C" int mpa_bn_act_fwd_bf16(const mpa_bf16 *x, const float *save_mean_invstd, const float *gamma,
                                   const float *beta, const mpa_bf16 *residual, float slope, int M, int C, mpa_bf16 *y,
                                   void *stream)
{
    return bn_act_fwd_any<bf16_t>(reinterpret_cast<const bf16_t *>(x), save_mean_invstd, gamma, beta,
                                  reinterpret_cast<const bf16_t *>(residual), slope, M, C,
                                  reinterpret_cast<bf16_t *>(y), stream);
}

template <typename T>
static int bn_act_bwd_reduce_any(const T *x, const T *grad_y, const float *mean, const float *invstd,
                                 const float *gamma, const float *beta, float slope, int M, int C, int ldg,
                                 float *partial, int replicas, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!x || !grad_y || !mean || !invstd || !gamma || !beta || !partial || replicas <= 0 || M <= 0 || C <= 0 ||
        ldg < C)
        return MPA_EINVAL;
    const bool al = ((((uintptr_t)x | (uintptr_t)grad_y) & mpa_vec4_align<T>::mask) == 0) &&
                    (((((uintptr_t)mean | (uintptr_t)invstd | (uintptr_t)gamma | (uintptr_t)beta)) & 15) == 0);
    if ((C & 3) == 0 && (ldg & 3) == 0 && al) {
        const BnReduceGeom g = bn_reduce_geom(M, C);
        const int lanes = g.lanes, gy_ = g.gy, rpb = g.rpb;
        hipLaunchKernelGGL(bn_act_bwd_reduce4_kernel<T>, dim3(mpa_ceil_div(M, rpb), gy_), dim3(256), 0,
                           (hipStream_t)stream, x, grad_y, mean, invstd, gamma, beta, slope, M, C, ldg, lanes, rpb, partial,
                           replicas);
    } else {
        dim3 grid;
        int cpb, rpb;
        slab_grid(M, C, grid, cpb, rpb);
        hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, x, grad_y, mean, invstd,
                           gamma, beta, slope, M, C, ldg, cpb, rpb, partial, replicas);
    }
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_bn_act_bwd_reduce_f32(const float *x, const float *grad_y, const float *mean, const float *invstd,
                                         const float *gamma, const float *beta, float slope, int M, int C,
                                         int ldg, float *partial, int replicas, void *stream)
{
    return bn_act_bwd_reduce_any<float>(x, grad_y, mean, invstd, gamma, beta, slope, M, C, ldg, partial, replicas, stream);
}

extern "C" int mpa_bn_act_bwd_reduce_bf16(const mpa_bf16 *x, const mpa_bf16 *grad_y, const float *mean,
                                          const float *invstd, const float *gamma, const float *beta, float slope, int M,
                                          int C, int ldg, float *partial, int replicas, void *stream)
{
    return bn_act_bwd_reduce_any<bf16_t>(reinterpret_cast<const bf16_t *>(x), reinterpret_cast<const bf16_t *>(grad_y), mean,
                                         invstd, gamma, beta, slope, M, C, ldg, partial, replicas, stream);
}

template <typename T>
static int bn_act_bwd_apply_any(const T *x, const T *grad_y, const float *mean, const float *invstd,
                                const float *gamma, const float *beta, const float *partial, int replicas, float slope,
                                int use_batch_stats, int M, int C, int ldg, T *grad_x, float *dgamma, float *dbeta,
                                void *stream)
{
    MPA_CLEAR_ERROR();
    if (!x || !grad_y || !mean || !invstd || !gamma || !beta || !grad_x || !partial || replicas <= 0 || M <= 0 ||
        C <= 0 || ldg < C)
        return MPA_EINVAL;
    if (C > 8192) return MPA_EUNSUPPORTED;
    long long total = (long long)M * C;
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel<T>, dim3(ew_grid(total / 4 + 1)), dim3(EW_TPB), 6 * C * sizeof(float),
                       (hipStream_t)stream, x, grad_y, mean, invstd, gamma, beta, partial, replicas, slope,
                       use_batch_stats, M, C, ldg, total, grad_x, dgamma, dbeta);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_bn_act_bwd_apply_f32(const float *x, const float *grad_y, const float *mean, const float *invstd,
                                        const float *gamma, const float *beta, const float *partial, int replicas,
                                        float slope, int use_batch_stats, int M, int C, int ldg, float *grad_x,
                                        float *dgamma, float *dbeta, void *stream)
{
    return bn_act_bwd_apply_any<float>(x, grad_y, mean, invstd, gamma, beta, partial, replicas, slope, use_batch_stats, M,
                                       C, ldg, grad_x, dgamma, dbeta, stream);
}

extern "C" int mpa_bn_act_bwd_apply_bf16(const mpa_bf16 *x, const mpa_bf16 *grad_y, const float *mean,
                                         const float *invstd, const float *gamma, const float *beta, const float *partial,
                                         int replicas, float slope, int use_batch_stats, int M, int C, int ldg,
                                         mpa_bf16 *grad_x, float *dgamma, float *dbeta, void *stream)
{
    return bn_act_bwd_apply_any<bf16_t>(reinterpret_cast<const bf16_t *>(x), reinterpret_cast<const bf16_t *>(grad_y), mean,
                                        invstd, gamma, beta, partial, replicas, slope, use_batch_stats, M, C, ldg,
                                        reinterpret_cast<bf16_t *>(grad_x), dgamma, dbeta, stream);
}

template <typename T>
static int group_col_sum_any(const T *x, int G, int R, int C, int ld, float *out, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!x || !out || G <= 0 || R <= 0 || C <= 0 || ld < C || G > 65535) return MPA_EINVAL;
    hipLaunchKernelGGL(group_col_sum_kernel<T>, dim3(mpa_ceil_div(C, 64), G), dim3(256), 0, (hipStream_t)stream, x, R, C,
                       ld, out);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_group_col_sum_f32(const float *x, int G, int R, int C, int ld, float *out, void *stream)
{
    return group_col_sum_any<float>(x, G, R, C, ld, out, stream);
}

extern "C" int mpa_group_col_sum_bf16(const mpa_bf16 *x, int G, int R, int C, int ld, float *out, void *stream)
{
    return group_col_sum_any<bf16_t>(reinterpret_cast<const bf16_t *>(x), G, R, C, ld, out, stream);
}

template <typename T>
static int bn_stats_act_fwd_any(const T *x, const float *stats, int replicas, int M, int C, float *running_mean,
                                float *running_var, int training, float momentum, float eps,
                                int64_t *num_batches_tracked, const float *gamma, const float *beta, const T *residual,
                                float slope, T *y, float *save_mean_invstd, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!x || !gamma || !beta || !y || !save_mean_invstd || M <= 0 || C <= 0) return MPA_EINVAL;
    if (training && (!stats || replicas <= 0)) return MPA_EINVAL;
    if (!training && (!running_mean || !running_var)) return MPA_EINVAL;
    if (C > 8192) return MPA_EUNSUPPORTED;
    const bool al = (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & mpa_vec4_align<T>::mask) == 0;
    if ((C & 3) == 0 && !al) return MPA_EUNSUPPORTED;
    long long total = (long long)M * C;
    hipLaunchKernelGGL(bn_stats_act_fwd_kernel<T>, dim3(ew_grid(total / 4 + 1)), dim3(EW_TPB), 2 * C * sizeof(float),
                       (hipStream_t)stream, x, stats, replicas, training, running_mean, running_var, momentum, eps,
                       reinterpret_cast<long long *>(num_batches_tracked), gamma, beta, residual, slope, M, C, y,
                       save_mean_invstd);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_bn_stats_act_fwd_f32(const float *x, const float *stats, int replicas, int M, int C,
                                        float *running_mean, float *running_var, int training, float momentum,
                                        float eps, int64_t *num_batches_tracked, const float *gamma, const float *beta,
                                        const float *residual, float slope, float *y, float *save_mean_invstd,
                                        void *stream)
{
    return bn_stats_act_fwd_any<float>(x, stats, replicas, M, C, running_mean, running_var, training, momentum, eps,
                                       num_batches_tracked, gamma, beta, residual, slope, y, save_mean_invstd, stream);
}

extern "C" int mpa_bn_stats_act_fwd_bf16(const mpa_bf16 *x, const float *stats, int replicas, int M, int C,
                                         float *running_mean, float *running_var, int training, float momentum,
                                         float eps, int64_t *num_batches_tracked, const float *gamma, const float *beta,
                                         const mpa_bf16 *residual, float slope, mpa_bf16 *y, float *save_mean_invstd,
                                         void *stream)
{
    return bn_stats_act_fwd_any<bf16_t>(reinterpret_cast<const bf16_t *>(x), stats, replicas, M, C, running_mean,
                                        running_var, training, momentum, eps, num_batches_tracked, gamma, beta,
                                        reinterpret_cast<const bf16_t *>(residual), slope, reinterpret_cast<bf16_t *>(y),
                                        save_mean_invstd, stream);
}

// ---- grouped BatchNorm launches (see BnGroupArgs): host structs -> kernel arguments ----------------------------------
static int bn_group_pack(const MpaBnUnit *units, int count, BnGroupArgs &a, bool bf16)
{
    if (!units || count <= 0 || count > BN_GROUP_MAX) return MPA_EINVAL;
    a.count = count;
    const uintptr_t mask = bf16 ? 7 : 15;
    for (int i = 0; i < count; ++i) {
        const MpaBnUnit &m = units[i];
        BnUnit &u = a.u[i];
        if (!m.x || !m.gamma || !m.beta || !m.save || m.M <= 0 || m.C <= 0) return MPA_EINVAL;
        if ((m.C & 3) != 0 || m.C > 8192) return MPA_EUNSUPPORTED;
        if ((((uintptr_t)m.x | (uintptr_t)m.y | (uintptr_t)m.residual | (uintptr_t)m.grad_y | (uintptr_t)m.grad_x) & mask) != 0)
            return MPA_EUNSUPPORTED;
        u.x = m.x; u.stats = m.stats; u.running_mean = m.running_mean; u.running_var = m.running_var;
        u.nbt = reinterpret_cast<long long *>(m.num_batches_tracked);
        u.gamma = m.gamma; u.beta = m.beta; u.residual = m.residual; u.y = m.y; u.save = m.save;
        u.gy = m.grad_y; u.partial = m.partial; u.gx = m.grad_x; u.dgamma = m.dgamma; u.dbeta = m.dbeta;
        u.M = m.M; u.C = m.C; u.R = m.stats_replicas; u.ldg = m.ldg; u.training = m.training; u.replicas = m.replicas;
        u.ldy = m.ldy > 0 ? m.ldy : m.C;
        if (u.ldy < m.C || (u.ldy & 3) != 0) return MPA_EINVAL;
        u.momentum = m.momentum; u.eps = m.eps; u.slope = m.slope;
    }
    return MPA_OK;
}

template <typename T>
static int bn_group_fwd_any(const MpaBnUnit *units, int count, int sum_mode, void *stream)
{
    MPA_CLEAR_ERROR();
    BnGroupArgs a;
    int rc = bn_group_pack(units, count, a, sizeof(T) == 2);
    if (rc != MPA_OK) return rc;
    long long most = 0;
    int cmax = 0;
    for (int i = 0; i < count; ++i) {
        const BnUnit &u = a.u[i];
        if (!u.y && !(sum_mode && i + 1 < count)) return MPA_EINVAL;
        if (u.training ? (!u.stats || u.R <= 0) : (!u.running_mean || !u.running_var)) return MPA_EINVAL;
        if (sum_mode && (u.M != a.u[0].M || u.C != a.u[0].C)) return MPA_EINVAL;
        most = most > (long long)u.M * u.C ? most : (long long)u.M * u.C;
        cmax = cmax > u.C ? cmax : u.C;
    }
    const size_t lds = (size_t)(sum_mode ? count : 1) * 2 * cmax * sizeof(float);
    if (lds > 64 * 1024) return MPA_EUNSUPPORTED;
    hipLaunchKernelGGL(bn_group_fwd_kernel<T>, dim3(ew_grid(most / 4 + 1), sum_mode ? 1 : count), dim3(EW_TPB), lds,
                       (hipStream_t)stream, a, sum_mode);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

template <typename T>
static int bn_group_bwd_reduce_any(const MpaBnUnit *units, int count, void *stream)
{
    MPA_CLEAR_ERROR();
    BnGroupArgs a;
    int rc = bn_group_pack(units, count, a, sizeof(T) == 2);
    if (rc != MPA_OK) return rc;
    BnGroupGeom g;
    int gx = 0, gy = 0;
    for (int i = 0; i < count; ++i) {
        const BnUnit &u = a.u[i];
        if (!u.gy || !u.partial || u.replicas <= 0 || u.ldg < u.C) return MPA_EINVAL;
        if ((u.ldg & 3) != 0 || ((uintptr_t)u.gamma | (uintptr_t)u.beta | (uintptr_t)u.save) & 15) return MPA_EUNSUPPORTED;
        const BnReduceGeom r = bn_reduce_geom(u.M, u.C);
        g.lanes[i] = r.lanes; g.gy[i] = r.gy; g.rpb[i] = r.rpb; g.gx[i] = r.gx;
        gx = gx > r.gx ? gx : r.gx;
        gy = gy > r.gy ? gy : r.gy;
    }
    hipLaunchKernelGGL(bn_group_bwd_reduce_kernel<T>, dim3(gx, gy, count), dim3(256), 0, (hipStream_t)stream, a, g);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

template <typename T>
static int bn_group_bwd_apply_any(const MpaBnUnit *units, int count, void *stream)
{
    MPA_CLEAR_ERROR();
    BnGroupArgs a;
    int rc = bn_group_pack(units, count, a, sizeof(T) == 2);
    if (rc != MPA_OK) return rc;
    long long most = 0;
    int cmax = 0;
    for (int i = 0; i < count; ++i) {
        const BnUnit &u = a.u[i];
        if (!u.gy || !u.partial || !u.gx || u.replicas <= 0 || u.ldg < u.C) return MPA_EINVAL;
        most = most > (long long)u.M * u.C ? most : (long long)u.M * u.C;
        cmax = cmax > u.C ? cmax : u.C;
    }
    hipLaunchKernelGGL(bn_group_bwd_apply_kernel<T>, dim3(ew_grid(most / 4 + 1), count), dim3(EW_TPB),
                       6 * cmax * sizeof(float), (hipStream_t)stream, a);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_bn_group_fwd_f32(const MpaBnUnit *units, int count, int sum_mode, void *stream)
{
    return bn_group_fwd_any<float>(units, count, sum_mode, stream);
}
extern "C" int mpa_bn_group_fwd_bf16(const MpaBnUnit *units, int count, int sum_mode, void *stream)
{
    return bn_group_fwd_any<bf16_t>(units, count, sum_mode, stream);
}
extern "C" int mpa_bn_group_bwd_reduce_f32(const MpaBnUnit *units, int count, void *stream)
{
    return bn_group_bwd_reduce_any<float>(units, count, stream);
}
extern "C" int mpa_bn_group_bwd_reduce_bf16(const MpaBnUnit *units, int count, void *stream)
{
    return bn_group_bwd_reduce_any<bf16_t>(units, count, stream);
}
extern "C" int mpa_bn_group_bwd_apply_f32(const MpaBnUnit *units, int count, void *stream)
{
    return bn_group_bwd_apply_any<float>(units, count, stream);
}
extern "C" int mpa_bn_group_bwd_apply_bf16(const MpaBnUnit *units, int count, void *stream)
{
    return bn_group_bwd_apply_any<bf16_t>(units, count, stream);
}

extern "C" int mpa_gemm_grouped_f32(const MpaGemmProblem *problems, int count, int transB, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!problems || count <= 0 || count > GROUP_NT_MAX) return MPA_EINVAL;
    NtArgs ga;
    int blocks = 0;
    ga.block_start[0] = 0;
    for (int i = 0; i < count; ++i) {
        const MpaGemmProblem &in = problems[i];
        if (!in.A || !in.B || !in.C || in.M <= 0 || in.N <= 0 || in.K <= 0 || in.lda < in.K || in.ldc < in.N ||
            in.ldb < (transB ? in.K : in.N))
            return MPA_EINVAL;
        NtProblem &q = ga.p[i];
        q.A = reinterpret_cast<const float *>(in.A); q.B = reinterpret_cast<const float *>(in.B); q.bias = in.bias;
        q.C = reinterpret_cast<float *>(in.C); q.stats = in.tile_stats;
        q.lda = in.lda; q.ldb = in.ldb; q.ldc = in.ldc; q.M = in.M; q.N = in.N; q.K = in.K;
        q.stats_acc = in.stats_replicas;
        q.tiles = mpa_ceil_div(in.M, TS) * mpa_ceil_div(in.N, TS);
        q.vec = ((reinterpret_cast<uintptr_t>(in.A) & 15) == 0 && (in.lda % 4) == 0 ? 1 : 0) |
                ((reinterpret_cast<uintptr_t>(in.B) & 15) == 0 && (in.ldb % 4) == 0 ? 2 : 0);
        blocks += q.tiles;
        ga.block_start[i + 1] = blocks;
    }
    ga.count = count;
    constexpr size_t lds = gemm_lds_bytes(false, true);
    if (transB)
        hipLaunchKernelGGL(gemm_nt_grouped_kernel<true>, dim3(blocks), dim3(NT), lds, (hipStream_t)stream, ga);
    else
        hipLaunchKernelGGL(gemm_nt_grouped_kernel<false>, dim3(blocks), dim3(NT), lds, (hipStream_t)stream, ga);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}
