// The transition / pointwise MLP unit on bf16 features for gfx950 -- reference Linear
// (modules/pointnet2_utils.py:401-425) in the precision BASELINE configs 3 and 5 name: features and
// their gradients are bf16 in HBM, products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation,
// parameters stay fp32 (converted while their tile is staged), BatchNorm statistics come from the fp32
// accumulators, weight gradients are produced in fp32.
//
// Shapes on the path are M = B*S rows (4,096 .. 65,536) against 64 .. 896 channels: 32 .. 200 FLOP per
// byte against a machine balance of ~310 (2.5 PFLOP/s : 8 TB/s), so every product here is HBM-bound and
// the kernels are built to stream: 16-B global loads for every operand, whole rows written back in 16-B
// pieces, operands staged once per workgroup through LDS, fp32 statistics taken from the accumulators
// so the output is never re-read.
//
//   gemm_bf16_kernel          C[M,N] = A[M,K] * op(B) (+ bias): A bf16 row-major; B either [N][K]
//                             (nn.Linear.weight: the forward product) or [K][N] (the same weight read along
//                             its rows: dX = dY * W), bf16 or fp32; C bf16 (or fp32: the logits).
//   gemm_bf16_tn_grouped      out_p[M,N] = A_p^T B_p, A_p [K][M], B_p [K][N] bf16, out fp32: all weight
//                             gradients dW = dY^T X of a backward pass in one launch (split-K slabs).
//
// LDS images.  An operand whose reduction index is contiguous in memory (A; B = [N][K]) is kept as
// [row][64 k] with its eight 16-B chunks XOR-swizzled by (row >> 1) & 7: a lane's MFMA fragment (8
// consecutive k of its row) is one ds_read_b128 and 16 rows x 1 chunk fall on 16 different 16-B slots.
// An operand whose reduction index is the ROW index (B = [K][N]; both operands of dW) is kept as
// [k/4][c/32][k%4][c%32]: the hardware transpose read ds_read_b64_tr_b16 hands a lane 4 consecutive k of
// its column, a half-wave's read is one contiguous 256-B block (conflict-free), and the image is filled
// with 16-B stores of what was loaded (no transposition in registers).
#include "mpa_common.h"
#include "mpa_bf16.h"
#include "splitk_reduce.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) bf16x4_t lds_bf16x4;

constexpr int NT = 256;    // 4 waves
constexpr int BM = 128;    // rows per workgroup tile (forward / dX)
constexpr int BK = 64;     // reduction depth per staged slab

// ---- LDS image offsets (bytes)
__device__ __forceinline__ int kc_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
template <int W>   // W columns per k row
__device__ __forceinline__ int km_off(int k, int c)
{
    return ((((k >> 2) * (W >> 5)) + (c >> 5)) << 8) + ((k & 3) << 6) + ((c & 31) << 1);
}

// ---- 8 consecutive elements -> 8 bf16 (one uint4)
__device__ __forceinline__ uint4 pack8(const float (&f)[8])
{
    return make_uint4(mpa_pack_bf16x2(f[0], f[1]), mpa_pack_bf16x2(f[2], f[3]), mpa_pack_bf16x2(f[4], f[5]),
                      mpa_pack_bf16x2(f[6], f[7]));
}
__device__ __forceinline__ uint4 load8(const bf16_t *p) { return *reinterpret_cast<const uint4 *>(p); }
__device__ __forceinline__ uint4 load8(const float *p)
{
    const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
    const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return pack8(f);
}
// the first `nvalid` (0..8) elements, the rest zero; element-wise (edges, unaligned rows)
template <typename T>
__device__ __forceinline__ uint4 load8_guard(const T *p, int nvalid)
{
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = e < nvalid ? mpa_ld1<T>(p + e) : 0.f;
    return pack8(f);
}

// the MFMA fragment (8 consecutive k of one column) of a k-major image: two transposed reads
template <int W>
__device__ __forceinline__ bf16x8_t tr_frag(const char *img, int kb, int cbase, int lane)
{
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    const int off = (g1 << 5) + ((i16 >> 2) << 6) + ((i16 & 3) << 3);
    const char *p0 = img + km_off<W>(kb, cbase) + off;
    const char *p1 = img + km_off<W>(kb + 4, cbase) + off;
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)p0);
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)p1);
    bf16x8_t r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// =============================================================================== forward / dX
// One workgroup = a 128 x BN output tile, its 4 waves 2 (rows) x 2 (columns): a wave owns 64 rows x BN/2
// columns = 2 x (BN/64) accumulator tiles of 32 x 32, and its 64 rows are exactly one 64-row tile of the
// BatchNorm statistics format (tile_stats [ceil(M/64)][2][N]: sum, sum of squared deviations from the tile
// mean), which therefore come out of the accumulators with one cross-half shuffle and no LDS.
//
// (Staging helpers are free functions taking the register arrays by reference: as lambdas capturing the
//  arrays, two instantiations kept them in scratch memory -- 320 B per lane, every slab through it -- and ran
//  3.7x slower than their twins.)
template <int BN> struct GemmCfg {
    static constexpr int NJ = BN / 64;                    // accumulator tiles per wave along N
    static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, BUF = A_BYTES + B_BYTES;
    static constexpr int BCH = BN * BK / 8 / NT;          // 16-B chunks of a B slab per lane
};

template <bool INTERIOR, bool TB, int BN, typename TBm>
__device__ __forceinline__ void gemm_load_slab(const bf16_t *__restrict__ A, int lda, const TBm *__restrict__ Bm,
                                               int ldb, int m0, int n0, int k0, int M, int N, int K, int vecA,
                                               int vecB, int tid, uint4 (&ra)[4], uint4 (&rb)[GemmCfg<BN>::BCH])
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {                                            // A: 128 rows x 8 chunks
        const int i = tid + NT * q, row = i >> 3, ch = i & 7;
        const bf16_t *p = A + (size_t)(m0 + row) * lda + k0 + ch * 8;
        if constexpr (INTERIOR) ra[q] = load8(p);
        else {
            const int nv = (m0 + row < M) ? min(8, max(0, K - (k0 + ch * 8))) : 0;
            ra[q] = (vecA && nv == 8) ? load8(p) : load8_guard(p, nv);
        }
    }
#pragma unroll
    for (int q = 0; q < GemmCfg<BN>::BCH; ++q) {
        const int i = tid + NT * q;
        if constexpr (TB) {                                                  // B [N][K]: BN rows x 8 chunks
            const int row = i >> 3, ch = i & 7;
            const TBm *p = Bm + (size_t)(n0 + row) * ldb + k0 + ch * 8;
            if constexpr (INTERIOR) rb[q] = load8(p);
            else {
                const int nv = (n0 + row < N) ? min(8, max(0, K - (k0 + ch * 8))) : 0;
                rb[q] = (vecB && nv == 8) ? load8(p) : load8_guard(p, nv);
            }
        } else {                                                             // B [K][N]: 64 k rows x BN/8 chunks
            const int kr = i / (BN / 8), cc = i % (BN / 8);
            const TBm *p = Bm + (size_t)(k0 + kr) * ldb + n0 + cc * 8;
            if constexpr (INTERIOR) rb[q] = load8(p);
            else {
                const int nv = (k0 + kr < K) ? min(8, max(0, N - (n0 + cc * 8))) : 0;
                rb[q] = (vecB && nv == 8) ? load8(p) : load8_guard(p, nv);
            }
        }
    }
}

template <bool TB, int BN>
__device__ __forceinline__ void gemm_store_slab(char *buf, int tid, const uint4 (&ra)[4],
                                                const uint4 (&rb)[GemmCfg<BN>::BCH])
{
    char *As = buf, *Bs = buf + GemmCfg<BN>::A_BYTES;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = tid + NT * q;
        *reinterpret_cast<uint4 *>(As + kc_off(i >> 3, i & 7)) = ra[q];
    }
#pragma unroll
    for (int q = 0; q < GemmCfg<BN>::BCH; ++q) {
        const int i = tid + NT * q;
        if constexpr (TB) *reinterpret_cast<uint4 *>(Bs + kc_off(i >> 3, i & 7)) = rb[q];
        else *reinterpret_cast<uint4 *>(Bs + km_off<BN>(i / (BN / 8), (i % (BN / 8)) * 8)) = rb[q];
    }
}

template <bool TB, int BN>
__device__ __forceinline__ void gemm_multiply(const char *buf, int wm, int wn, int lane,
                                              floatx16 (&acc)[2][GemmCfg<BN>::NJ])
{
    constexpr int NJ = GemmCfg<BN>::NJ;
    const int half = lane >> 5, r31 = lane & 31;
    const char *As = buf, *Bs = buf + GemmCfg<BN>::A_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
        bf16x8_t a[2], b[NJ];
#pragma unroll
        for (int i = 0; i < 2; ++i)
            a[i] = *reinterpret_cast<const bf16x8_t *>(As + kc_off(wm * 64 + i * 32 + r31, 2 * ks + half));
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if constexpr (TB)
                b[j] = *reinterpret_cast<const bf16x8_t *>(Bs + kc_off(wn * (BN / 2) + j * 32 + r31, 2 * ks + half));
            else
                b[j] = tr_frag<BN>(Bs, 16 * ks + 8 * half, wn * (BN / 2) + j * 32, lane);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}

template <bool INTERIOR, bool TB, int BN, typename TBm>
__device__ __forceinline__ void gemm_main_loop(const bf16_t *__restrict__ A, int lda, const TBm *__restrict__ Bm,
                                               int ldb, int m0, int n0, int M, int N, int K, int vecA, int vecB,
                                               char *smem, floatx16 (&acc)[2][GemmCfg<BN>::NJ])
{
    constexpr int BUF = GemmCfg<BN>::BUF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int nslab = (K + BK - 1) / BK;
    uint4 ra[4], rb[GemmCfg<BN>::BCH];
    gemm_load_slab<INTERIOR, TB, BN, TBm>(A, lda, Bm, ldb, m0, n0, 0, M, N, K, vecA, vecB, tid, ra, rb);
    gemm_store_slab<TB, BN>(smem, tid, ra, rb);
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        char *cur = smem + (s & 1) * BUF, *nxt = smem + ((s + 1) & 1) * BUF;
        if (s + 1 < nslab)                                    // next slab's loads fly during the MFMAs
            gemm_load_slab<INTERIOR, TB, BN, TBm>(A, lda, Bm, ldb, m0, n0, (s + 1) * BK, M, N, K, vecA, vecB, tid, ra, rb);
        gemm_multiply<TB, BN>(cur, wm, wn, lane, acc);
        if (s + 1 < nslab) gemm_store_slab<TB, BN>(nxt, tid, ra, rb);
        __syncthreads();
    }
}

template <bool TB, int BN, typename TBm, bool OUT_F32>
__device__ __forceinline__ void gemm_bf16_body(const int block_id, const bf16_t *__restrict__ A, int lda,
                                               const TBm *__restrict__ Bm, int ldb, const float *__restrict__ bias,
                                               void *__restrict__ Cv, int ldc, int M, int N, int K,
                                               float *__restrict__ tile_stats, int stats_acc, int vecA, int vecB,
                                               int vecC, char *smem)
{
    constexpr int NJ = GemmCfg<BN>::NJ;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, r31 = lane & 31;
    const int wm = wave & 1, wn = wave >> 1;

    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    const int ntiles = tiles_m * tiles_n;
    int id = block_id;
    if ((ntiles & 7) == 0) id = (id & 7) * (ntiles >> 3) + (id >> 3);       // XCD-aware: an XCD gets a run of tiles
    const int m0 = (id / tiles_n) * BM, n0 = (id % tiles_n) * BN;

    floatx16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (vecA && vecB && m0 + BM <= M && n0 + BN <= N && (K % BK) == 0)
        gemm_main_loop<true, TB, BN, TBm>(A, lda, Bm, ldb, m0, n0, M, N, K, vecA, vecB, smem, acc);
    else
        gemm_main_loop<false, TB, BN, TBm>(A, lda, Bm, ldb, m0, n0, M, N, K, vecA, vecB, smem, acc);

    // ---- epilogue: bias, BatchNorm tile statistics (from the fp32 accumulators), coalesced row stores
    float bv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int col = n0 + wn * (BN / 2) + j * 32 + r31;
        bv[j] = (bias != nullptr && col < N) ? bias[col] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] += bv[j];

    const int trow0 = m0 + wm * 64;                 // this wave's 64 rows = one statistics tile
    if (tile_stats != nullptr && trow0 < M) {
        const int nrows = min(64, M - trow0);
        const float inv = 1.0f / (float)nrows;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                    if (row < nrows) s += acc[i][j][e];
                }
            s += __shfl_xor(s, 32, 64);
            const float mu = s * inv;
            float m2 = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                    if (row < nrows) { const float d = acc[i][j][e] - mu; m2 = fmaf(d, d, m2); }
                }
            m2 += __shfl_xor(m2, 32, 64);
            const int col = n0 + wn * (BN / 2) + j * 32 + r31;
            if (half == 0 && col < N) {
                if (stats_acc > 0) {       // accumulate form [R][3][N] (see mpa_bn_stats_act_fwd)
                    float *dst = tile_stats + (size_t)((trow0 >> 6) % stats_acc) * 3 * N + col;
                    atomicAdd(dst, s);
                    atomicAdd(dst + N, m2);
                    atomicAdd(dst + 2 * N, s * s * inv);
                } else {
                    float *dst = tile_stats + (size_t)(trow0 >> 6) * 2 * N + col;
                    dst[0] = s;
                    dst[N] = m2;
                }
            }
        }
    }

    // each wave's 64 x BN/2 sub-tile goes through its own LDS region ([row][BN/2 + 4] floats) so that the
    // stores are whole-row pieces of 16 B per lane
    constexpr int WCOLS = BN / 2, LDW = WCOLS + 4;
    float *wreg = reinterpret_cast<float *>(smem) + wave * (64 * LDW);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                wreg[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half) * LDW + j * 32 + r31] = acc[i][j][e];
    __syncthreads();
    constexpr int LPR = WCOLS / 8;                  // lanes per row (8 columns each)
    constexpr int RPP = 64 / LPR;                   // rows per pass
    const int lr = lane / LPR, lc = (lane % LPR) * 8;
#pragma unroll
    for (int pass = 0; pass < 64 / RPP; ++pass) {
        const int rloc = pass * RPP + lr;
        const int row = trow0 + rloc, col = n0 + wn * WCOLS + lc;
        if (row >= M || col >= N) continue;
        const float4 v0 = *reinterpret_cast<const float4 *>(wreg + rloc * LDW + lc);
        const float4 v1 = *reinterpret_cast<const float4 *>(wreg + rloc * LDW + lc + 4);
        const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        if constexpr (OUT_F32) {
            float *o = reinterpret_cast<float *>(Cv) + (size_t)row * ldc + col;
            if (vecC && col + 7 < N) {
                *reinterpret_cast<float4 *>(o) = v0;
                *reinterpret_cast<float4 *>(o + 4) = v1;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (col + e < N) o[e] = f[e];
            }
        } else {
            bf16_t *o = reinterpret_cast<bf16_t *>(Cv) + (size_t)row * ldc + col;
            if (vecC && col + 7 < N) *reinterpret_cast<uint4 *>(o) = pack8(f);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (col + e < N) o[e] = (bf16_t)f[e];
            }
        }
    }
}

template <bool TB, int BN, typename TBm, bool OUT_F32>
__global__ __launch_bounds__(NT, 2) void gemm_bf16_kernel(const bf16_t *__restrict__ A, int lda,
                                                          const TBm *__restrict__ Bm, int ldb,
                                                          const float *__restrict__ bias, void *__restrict__ Cv, int ldc,
                                                          int M, int N, int K, float *__restrict__ tile_stats,
                                                          int stats_acc, int vecA, int vecB, int vecC)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_bf16_body<TB, BN, TBm, OUT_F32>(blockIdx.x, A, lda, Bm, ldb, bias, Cv, ldc, M, N, K, tile_stats, stats_acc, vecA,
                                         vecB, vecC, smem);
}

// ---- grouped forward / dX products: a few independent problems in one launch (the Linear units of parallel
// attention streams, Fuse's four source states): see gemm_nt_grouped_kernel in linear.hip.
constexpr int GROUP_NT_MAX = 8;
struct NtProblemB {
    const bf16_t *A;
    const void *B;
    const float *bias;
    void *C;
    float *stats;
    int lda, ldb, ldc, M, N, K, stats_acc, vec;
};
struct NtArgsB {
    int count;
    int block_start[GROUP_NT_MAX + 1];
    NtProblemB p[GROUP_NT_MAX];
};

template <bool TB, int BN, typename TBm>
__global__ __launch_bounds__(NT, 2) void gemm_bf16_grouped_kernel(const NtArgsB args)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int which = 0;
    while (which + 1 < args.count && (int)blockIdx.x >= args.block_start[which + 1]) ++which;     // (wave-uniform)
    const NtProblemB &q = args.p[which];
    gemm_bf16_body<TB, BN, TBm, false>(blockIdx.x - args.block_start[which], q.A, q.lda,
                                       reinterpret_cast<const TBm *>(q.B), q.ldb, q.bias, q.C, q.ldc, q.M, q.N, q.K, q.stats,
                                       q.stats_acc, q.vec & 1, (q.vec >> 1) & 1, (q.vec >> 2) & 1, smem);
}

template <int BN> constexpr size_t gemm_bf16_lds()
{
    constexpr size_t stage = 2 * (size_t)(BM * BK * 2 + BN * BK * 2);
    constexpr size_t epi = 4 * (size_t)64 * (BN / 2 + 4) * 4;
    return stage > epi ? stage : epi;
}

template <bool TB, int BN, typename TBm, bool OUT_F32>
int launch_gemm_bf16(const bf16_t *A, int lda, const TBm *B, int ldb, const float *bias, void *C, int ldc, int M, int N,
                     int K, float *stats, int stats_acc, int vecA, int vecB, int vecC, hipStream_t st)
{
    constexpr size_t lds = gemm_bf16_lds<BN>();
    auto kern = gemm_bf16_kernel<TB, BN, TBm, OUT_F32>;
    if (lds > 64 * 1024) {
        static bool once = false;       // (per instantiation)
        if (!once) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess)
                return MPA_EHIP;
            once = true;
        }
    }
    const dim3 grid(mpa_ceil_div(M, BM) * mpa_ceil_div(N, BN));
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, st, A, lda, B, ldb, bias, C, ldc, M, N, K, stats, stats_acc, vecA, vecB, vecC);
    return MPA_OK;
}

template <bool TB, typename TBm, bool OUT_F32>
int dispatch_bn(const bf16_t *A, int lda, const TBm *B, int ldb, const float *bias, void *C, int ldc, int M, int N, int K,
                float *stats, int stats_acc, int vecA, int vecB, int vecC, hipStream_t st)
{
    // 128-column tiles halve the re-reads of A through L2; narrow outputs, small grids and widths that 128 does
    // not divide (a partial tile runs the element-wise edge loop) take 64
    const long long t128 = (long long)mpa_ceil_div(M, BM) * mpa_ceil_div(N, 128);
    if (N > 64 && t128 >= 192 && (N % 128 == 0 || N % 64 != 0))
        return launch_gemm_bf16<TB, 128, TBm, OUT_F32>(A, lda, B, ldb, bias, C, ldc, M, N, K, stats, stats_acc, vecA, vecB, vecC, st);
    return launch_gemm_bf16<TB, 64, TBm, OUT_F32>(A, lda, B, ldb, bias, C, ldc, M, N, K, stats, stats_acc, vecA, vecB, vecC, st);
}

// =============================================================================== weight gradients
struct TnProblem {
    const bf16_t *A, *B;
    float *out, *a_col_sum, *slab;       // slab: split-K partial tiles [splits][M*N] (or nullptr: direct)
    int lda, ldb, M, N, K, kchunk, splits, tiles, vec;
};
struct TnArgs {
    int count;
    int block_start[GROUP_MAX + 1];      // prefix sum of tiles*splits
    TnProblem p[GROUP_MAX];
};

// One workgroup = a 64 x 64 tile of one problem's out[M,N] = A^T B over a chunk of the K = B*S rows; its 4
// waves are 2 x 2 quadrants of 32 x 32 (one accumulator tile each, no cross-wave sum).  Both operands are
// row = reduction index, so both LDS images are the transposed-read kind.  The product is a pure stream
// (16 KB of operands per 64 rows against 128 MFMA cycles): what matters is bytes in flight, so the next
// slab's 16-B loads are issued before the current slab's MFMAs and several workgroups share a CU.
__global__ __launch_bounds__(NT, 4) void gemm_bf16_tn_grouped_kernel(const TnArgs args)
{
    constexpr int TS = 64, IMG = TS * BK * 2;            // 8 KB per operand image
    __shared__ __attribute__((aligned(16))) char smem[4 * IMG];
    __shared__ int which;
    if (threadIdx.x == 0) {
        int b = blockIdx.x, i = 0;
        while (i + 1 < args.count && b >= args.block_start[i + 1]) ++i;
        which = i;
    }
    __syncthreads();
    const TnProblem &q = args.p[which];
    const int local = blockIdx.x - args.block_start[which];
    const int tile = local % q.tiles, z = local / q.tiles;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, r31 = lane & 31;
    const int wm = wave & 1, wn = wave >> 1;
    const int M = q.M, N = q.N, lda = q.lda, ldb = q.ldb;
    const int tiles_n = (N + TS - 1) / TS;
    const int m0 = (tile / tiles_n) * TS, n0 = (tile % tiles_n) * TS;
    const int kbeg = z * q.kchunk, kend = min(q.K, kbeg + q.kchunk);
    const int nslab = (kend - kbeg + BK - 1) / BK;
    const bf16_t *A = q.A, *B = q.B;
    const bool vecA = q.vec & 1, vecB = (q.vec >> 1) & 1;

    uint4 ra[2], rb[2];
    auto load_slab = [&](auto interior, int s) {
        constexpr bool INTERIOR = decltype(interior)::value;
        const int k0 = kbeg + s * BK;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + NT * u, kr = i >> 3, cc = (i & 7) * 8;
            const bf16_t *pa = A + (size_t)(k0 + kr) * lda + m0 + cc;
            const bf16_t *pb = B + (size_t)(k0 + kr) * ldb + n0 + cc;
            if constexpr (INTERIOR) {
                ra[u] = load8(pa);
                rb[u] = load8(pb);
            } else {
                const bool kok = k0 + kr < kend;
                const int na = kok ? min(8, max(0, M - (m0 + cc))) : 0, nb = kok ? min(8, max(0, N - (n0 + cc))) : 0;
                ra[u] = (vecA && na == 8) ? load8(pa) : load8_guard(pa, na);
                rb[u] = (vecB && nb == 8) ? load8(pb) : load8_guard(pb, nb);
            }
        }
    };
    auto store_slab = [&](char *buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + NT * u, kr = i >> 3, cc = (i & 7) * 8;
            *reinterpret_cast<uint4 *>(buf + km_off<TS>(kr, cc)) = ra[u];
            *reinterpret_cast<uint4 *>(buf + IMG + km_off<TS>(kr, cc)) = rb[u];
        }
    };
    floatx16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float asum = 0.f;                                  // sum over k of A[k][this lane's m] (bias gradient)
    const bool want_sum = q.a_col_sum != nullptr && n0 == 0 && wn == 0;
    auto multiply = [&](const char *buf) {
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const bf16x8_t a = tr_frag<TS>(buf, 16 * ks + 8 * half, wm * 32, lane);
            const bf16x8_t b = tr_frag<TS>(buf + IMG, 16 * ks + 8 * half, wn * 32, lane);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            if (want_sum) {
#pragma unroll
                for (int e = 0; e < 8; ++e) asum += (float)a[e];
            }
        }
    };
    auto main_loop = [&](auto interior) {
        if (nslab <= 0) return;
        load_slab(interior, 0);
        store_slab(smem);
        __syncthreads();
        for (int s = 0; s < nslab; ++s) {
            char *cur = smem + (s & 1) * 2 * IMG, *nxt = smem + ((s + 1) & 1) * 2 * IMG;
            if (s + 1 < nslab) load_slab(interior, s + 1);
            multiply(cur);
            if (s + 1 < nslab) store_slab(nxt);
            __syncthreads();
        }
    };
    if (vecA && vecB && m0 + TS <= M && n0 + TS <= N && ((kend - kbeg) % BK) == 0) main_loop(std::true_type{});
    else main_loop(std::false_type{});

    if (want_sum) {
        asum += __shfl_xor(asum, 32, 64);
        const int m = m0 + wm * 32 + r31;
        if (half == 0 && m < M) atomicAdd(q.a_col_sum + m, asum);
    }
    float *dst = q.out;
    if (q.slab != nullptr) {
        // split-K slabs are summed into the real output by splitk_reduce_grouped_kernel (atomics): the z = 0
        // workgroups clear their tile of it here
        if (z == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * half, col = n0 + wn * 32 + r31;
                if (row < M && col < N) q.out[(size_t)row * N + col] = 0.f;
            }
        }
        dst = q.slab + (size_t)z * M * N;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * half, col = n0 + wn * 32 + r31;
        if (row < M && col < N) dst[(size_t)row * N + col] = acc[e];
    }
}

}  // namespace

extern "C" int mpa_gemm_bf16(const mpa_bf16 *A, int lda, const void *B, int ldb, int transB, int b_is_f32,
                             const float *bias, void *C, int ldc, int c_is_f32, int M, int N, int K,
                             float *tile_stats, int stats_replicas, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || lda < K || ldc < N || ldb < (transB ? K : N)) return MPA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const bf16_t *Ab = reinterpret_cast<const bf16_t *>(A);
    const int vecA = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (lda % 8 == 0);
    const int vecB = b_is_f32 ? (((reinterpret_cast<uintptr_t>(B) & 15) == 0) && (ldb % 4 == 0))
                              : (((reinterpret_cast<uintptr_t>(B) & 15) == 0) && (ldb % 8 == 0));
    const int vecC = c_is_f32 ? (((reinterpret_cast<uintptr_t>(C) & 15) == 0) && (ldc % 4 == 0))
                              : (((reinterpret_cast<uintptr_t>(C) & 15) == 0) && (ldc % 8 == 0));
    int rc;
#define MPA_BF16_CASE(TB_, TBM_, OF32_)                                                                              \
    rc = dispatch_bn<TB_, TBM_, OF32_>(Ab, lda, reinterpret_cast<const TBM_ *>(B), ldb, bias, C, ldc, M, N, K,       \
                                       tile_stats, stats_replicas, vecA, vecB, vecC, st)
    if (transB) {
        if (b_is_f32) { if (c_is_f32) MPA_BF16_CASE(true, float, true); else MPA_BF16_CASE(true, float, false); }
        else { if (c_is_f32) MPA_BF16_CASE(true, bf16_t, true); else MPA_BF16_CASE(true, bf16_t, false); }
    } else {
        if (b_is_f32) { if (c_is_f32) MPA_BF16_CASE(false, float, true); else MPA_BF16_CASE(false, float, false); }
        else { if (c_is_f32) MPA_BF16_CASE(false, bf16_t, true); else MPA_BF16_CASE(false, bf16_t, false); }
    }
#undef MPA_BF16_CASE
    if (rc != MPA_OK) return rc;
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_gemm_tn_grouped_bf16(const MpaGemmTnProblemBf16 *problems, int count, float *workspace,
                                        size_t workspace_bytes, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!problems || count <= 0) return MPA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    static const int target_wgs = getenv("MPA_TN_BF16_WGS") ? atoi(getenv("MPA_TN_BF16_WGS")) : 512;
    static const int min_kchunk = getenv("MPA_TN_BF16_KCHUNK") ? atoi(getenv("MPA_TN_BF16_KCHUNK")) : 1024;
    size_t ws_used = 0;
    int done = 0;
    // more problems than one launch's argument block holds: equal-sized launches, in queue order (dealing the
    // fine states' HBM-bound streams out over the launches measured 1-2 % slower: they run best back to back)
    const int launches = mpa_ceil_div(count, GROUP_MAX), chunk = mpa_ceil_div(count, launches);
    while (done < count) {
        TnArgs ga;
        GroupedReduceArgs ra;
        int n = 0, nr = 0, blocks = 0, rblocks = 0;
        ga.block_start[0] = 0;
        ra.block_start[0] = 0;
        for (; done < count && n < chunk; ++done) {
            const MpaGemmTnProblemBf16 &in = problems[done];
            if (!in.A || !in.B || !in.out || in.M <= 0 || in.N <= 0 || in.K <= 0 || in.lda < in.M || in.ldb < in.N)
                return MPA_EINVAL;
            TnProblem &q = ga.p[n];
            q.A = reinterpret_cast<const bf16_t *>(in.A);
            q.B = reinterpret_cast<const bf16_t *>(in.B);
            q.out = in.out; q.a_col_sum = in.a_col_sum;
            q.lda = in.lda; q.ldb = in.ldb; q.M = in.M; q.N = in.N; q.K = in.K;
            q.tiles = mpa_ceil_div(in.M, 64) * mpa_ceil_div(in.N, 64);
            const size_t mn = (size_t)in.M * in.N;
            q.vec = ((in.lda & 7) == 0 && (reinterpret_cast<uintptr_t>(in.A) & 15) == 0 ? 1 : 0) |
                    ((in.ldb & 7) == 0 && (reinterpret_cast<uintptr_t>(in.B) & 15) == 0 ? 2 : 0);
            int splits = 1;
            if (q.tiles < target_wgs / 2 && in.K >= 2 * min_kchunk) {
                splits = (target_wgs + q.tiles - 1) / q.tiles;
                if (splits > in.K / min_kchunk) splits = in.K / min_kchunk;
                const size_t room = workspace ? (workspace_bytes - ws_used) / (mn * sizeof(float)) : 0;
                if ((size_t)splits > room) splits = (int)room;
                if (splits < 1) splits = 1;
            }
            q.kchunk = mpa_ceil_div(mpa_ceil_div(in.K, splits), BK) * BK;
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
                q.kchunk = mpa_ceil_div(in.K, BK) * BK;
            }
            blocks += q.tiles * q.splits;
            ga.block_start[++n] = blocks;
        }
        ga.count = n;
        ra.count = nr;
        hipLaunchKernelGGL(gemm_bf16_tn_grouped_kernel, dim3(blocks), dim3(NT), 0, st, ga);
        if (nr > 0) hipLaunchKernelGGL(splitk_reduce_grouped_kernel, dim3(rblocks), dim3(256), 0, st, ra);
    }
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

template <bool TB, int BN, typename TBm>
static int launch_grouped_bf16(const NtArgsB &ga, int blocks, hipStream_t st)
{
    constexpr size_t lds = gemm_bf16_lds<BN>();
    auto kern = gemm_bf16_grouped_kernel<TB, BN, TBm>;
    if (lds > 64 * 1024) {
        static bool once = false;
        if (!once) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess)
                return MPA_EHIP;
            once = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), lds, st, ga);
    return MPA_OK;
}

extern "C" int mpa_gemm_grouped_bf16(const MpaGemmProblem *problems, int count, int transB, int b_is_f32, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!problems || count <= 0 || count > GROUP_NT_MAX) return MPA_EINVAL;
    // one tile width for the whole launch: 128 columns when every problem is a multiple of it and the grid is large
    long long t128 = 0;
    bool all128 = true;
    for (int i = 0; i < count; ++i) {
        const MpaGemmProblem &in = problems[i];
        if (!in.A || !in.B || !in.C || in.M <= 0 || in.N <= 0 || in.K <= 0 || in.lda < in.K || in.ldc < in.N ||
            in.ldb < (transB ? in.K : in.N))
            return MPA_EINVAL;
        all128 = all128 && (in.N % 128 == 0);
        t128 += (long long)mpa_ceil_div(in.M, BM) * mpa_ceil_div(in.N, 128);
    }
    const int bn = (all128 && t128 >= 192) ? 128 : 64;
    NtArgsB ga;
    int blocks = 0;
    ga.block_start[0] = 0;
    for (int i = 0; i < count; ++i) {
        const MpaGemmProblem &in = problems[i];
        NtProblemB &q = ga.p[i];
        q.A = reinterpret_cast<const bf16_t *>(in.A); q.B = in.B; q.bias = in.bias; q.C = in.C; q.stats = in.tile_stats;
        q.lda = in.lda; q.ldb = in.ldb; q.ldc = in.ldc; q.M = in.M; q.N = in.N; q.K = in.K;
        q.stats_acc = in.stats_replicas;
        const int vecA = ((reinterpret_cast<uintptr_t>(in.A) & 15) == 0) && (in.lda % 8 == 0);
        const int vecB = ((reinterpret_cast<uintptr_t>(in.B) & 15) == 0) && (in.ldb % (b_is_f32 ? 4 : 8) == 0);
        const int vecC = ((reinterpret_cast<uintptr_t>(in.C) & 15) == 0) && (in.ldc % 8 == 0);
        q.vec = vecA | (vecB << 1) | (vecC << 2);
        blocks += mpa_ceil_div(in.M, BM) * mpa_ceil_div(in.N, bn);
        ga.block_start[i + 1] = blocks;
    }
    ga.count = count;
    hipStream_t st = (hipStream_t)stream;
    int rc;
#define MPA_GRP(TB_, BN_, T_) rc = launch_grouped_bf16<TB_, BN_, T_>(ga, blocks, st)
    if (transB) {
        if (b_is_f32) { if (bn == 128) MPA_GRP(true, 128, float); else MPA_GRP(true, 64, float); }
        else { if (bn == 128) MPA_GRP(true, 128, bf16_t); else MPA_GRP(true, 64, bf16_t); }
    } else {
        if (b_is_f32) { if (bn == 128) MPA_GRP(false, 128, float); else MPA_GRP(false, 64, float); }
        else { if (bn == 128) MPA_GRP(false, 128, bf16_t); else MPA_GRP(false, 64, bf16_t); }
    }
#undef MPA_GRP
    if (rc != MPA_OK) return rc;
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}
