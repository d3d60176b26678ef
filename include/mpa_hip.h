/*
 * mpa_hip.h -- C ABI of libmpa_hip.so: the MI355X (gfx950) implementation of the
 * reference's Markov set-abstraction hot path.
 *
 * The reference (ssr0512/Markov-Process-Analysis-on-Point-Cloud) has no FFI: its boundary is
 * the Python function / nn.Module API of modules/ (SURVEY.md section 8b).  Each entry point
 * below replaces the device work of one of those functions; the Python mirror in
 * markov-process-analysis-on-point-cloud_amd/modules/ keeps the reference names and signatures and calls
 * these through ctypes.  Paths below are relative to
 * Markov_Process_Analysis_on_Point_Cloud/ in the reference tree.
 *
 * Conventions (all entry points):
 *   - plain device pointers + sizes; no torch types; no allocation, no ownership transfer,
 *     no global state; re-entrant; asynchronous on the caller's `stream` (a hipStream_t
 *     passed as void*), never synchronises the device;
 *   - tensors are dense row-major, channel-last [B, N, C] float32; indices are int64
 *     (the reference uses torch.long everywhere);
 *   - return 0 on success, a negative MPA_E* code on a rejected argument, or
 *     MPA_EHIP when the launch itself failed (hipGetLastError is consumed).
 */
#ifndef MPA_HIP_H
#define MPA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPA_OK 0
#define MPA_EINVAL (-1)   /* bad size / null pointer */
#define MPA_EUNSUPPORTED (-2) /* shape outside what the kernels are built for */
#define MPA_EHIP (-3)     /* HIP launch error */

/* bfloat16 storage (the upper 16 bits of an IEEE fp32, round-to-nearest-even on conversion) */
typedef uint16_t mpa_bf16;

/* ABI version: bumped whenever an exported signature changes incompatibly.  100 = round 1; 200 = round 2 (mpa_gemm_f32
 * gained `stats_replicas`, mpa_adam_step_f32 gained `hyper`); 300 = round 3, 301 = + mpa_add_n_*.  A binding must compare mpa_version()
 * with the MPA_ABI_VERSION of the header it was written against and refuse a mismatch (the Python binding does,
 * markov-process-analysis-on-point-cloud_amd/_lib.py). */
#define MPA_ABI_VERSION 301
int mpa_version(void);
const char *mpa_error_string(int code);
/* the hipError_t (and its text) behind the calling thread's most recent MPA_EHIP */
int mpa_last_hip_error(void);
const char *mpa_last_hip_error_string(void);

/* ---- farthest_point_sample: modules/pointnet2_utils.py:84-109 (== repsurface_utils.py:150-172)
 * xyz [B,N,3]; start_idx [B] is the first sample of every cloud (the reference draws it with
 * torch.randint on the CPU generator, :96 -- the host wrapper does the same and passes it in).
 * out_idx [B,S] int64; out_xyz [B,S,3] (optional, may be NULL) receives xyz[out_idx], i.e.
 * the index_points() call that follows every FPS in the models (repsurface_utils.py:583).
 * Distances follow the reference's rounding exactly (no FMA contraction), ties -> first max.
 * One workgroup per cloud with the cloud resident on chip: N <= 12288 (MPA_EUNSUPPORTED beyond). */
int mpa_fps_f32(const float *xyz, int B, int N, int S, const int64_t *start_idx,
                int64_t *out_idx, float *out_xyz, void *stream);

/* farthest_point_sample on rows of any width C (the reference function takes xyz [B,N,C] for any C and sums
 * the squared differences over all channels, :103-104; dataset/ShapeNetDataLoader.py:127-133 samples on
 * xyz|normal rows).  points [B,N,C]; same start_idx / out_idx contract and the same tie rule as mpa_fps_f32;
 * the channel sum follows torch.sum's order on the reference's CPU path, so indices
 * are bit-exact for every C.  Not on the models' hot path: rows are re-read from L2 each iteration. */
int mpa_fps_generic_f32(const float *points, int B, int N, int C, int S, const int64_t *start_idx,
                        int64_t *out_idx, void *stream);

/* ---- square_distance: modules/pointnet2_utils.py:190-209.  src [B,S,C], dst [B,N,C] -> out [B,S,N];
 * bit-exact with the reference CPU result (FMA chain dot, separately rounded norms). */
int mpa_square_distance_f32(const float *src, const float *dst, int B, int S, int N, int C,
                            float *out, void *stream);

/* ---- knn_point: modules/pointnet2_utils.py:211-222.  base = `xyz` [B,N,C], query = `new_xyz`
 * [B,S,C]; K nearest base rows of every query, ascending by (distance, index); the [B,S,N]
 * matrix is never materialised.  out_dist [B,S,K] (may be NULL), out_idx [B,S,K].
 * C in {1..8, 16, 32, 64, 128, 256, 512}; K <= 32; K <= N. */
int mpa_knn_f32(const float *base, const float *query, int B, int N, int S, int C, int K,
                float *out_dist, int64_t *out_idx, void *stream);
/* FPS of the NEXT state (fps_* as in mpa_fps_knn_xyz_f32), the coordinate search of THIS state (xyz_*: optional, pass
 * xyz_base = NULL to skip) and its feature-space search (feat_*: as mpa_knn_f32 / mpa_knn_norms_f32, feat_norms optional)
 * in ONE launch: the sampling keeps one workgroup per cloud busy for fps_S dependent iterations, the searches use the
 * rest of the chip.  Results are those of the separate entry points bit for bit.  MPA_EUNSUPPORTED for shapes outside
 * the instantiated set (fps_N in 129..4096, K <= 8, C in {64, 128}, 16-byte aligned rows): use the separate calls. */
int mpa_fps_knn_feat_f32(const float *fps_xyz, int B, int fps_N, int fps_S, const int64_t *start_idx,
                         int64_t *fps_idx, float *fps_out_xyz, const float *xyz_base, const float *xyz_query,
                         int xN, int xS, int xK, float *xyz_dist, int64_t *xyz_idx, const float *feat_base,
                         const float *feat_norms, const float *feat_query, int N, int S, int C, int K,
                         float *out_dist, int64_t *out_idx, void *stream);
/* The same search with the base rows' squared norms given: norms [B][ceil32(N)] from mpa_row_norms_f32 (the reference's
 * `torch.sum(dst ** 2, -1)`, modules/pointnet2_utils.py:207, rounded as there; +inf in the padding).  One small launch
 * computes them once per search instead of once per (32-query workgroup, pass, tile) inside it.  Results are identical
 * to mpa_knn_f32's bit for bit.  C in {32, 64, 128, 256} use the norms; other widths ignore them. */
int mpa_row_norms_f32(const float *x, int B, int N, int C, float *norms, void *stream);
int mpa_knn_norms_f32(const float *base, const float *base_norms, const float *query, int B, int N, int S, int C,
                      int K, float *out_dist, int64_t *out_idx, void *stream);

/* FPS of one point-set state and the xyz-space (C = 3) kNN of the previous state in one launch:
 * mpa_fps_f32(fps_xyz [B,fps_N,3] -> fps_idx [B,fps_S], fps_out_xyz) and
 * mpa_knn_f32(knn_base [B,N,3], knn_query [B,S,3], K) -> (out_dist, out_idx); the two are independent
 * (neither reads what the other writes).  FPS occupies one workgroup per cloud for fps_S serial
 * iterations, the search fills the rest of the chip meanwhile.  Shapes outside the fused kernel's
 * range (fps_N in (128, 2048], K <= 8) are issued as the two separate launches. */
int mpa_fps_knn_xyz_f32(const float *fps_xyz, int B, int fps_N, int fps_S, const int64_t *start_idx,
                        int64_t *fps_idx, float *fps_out_xyz, const float *knn_base,
                        const float *knn_query, int N, int S, int K, float *out_dist, int64_t *out_idx,
                        void *stream);

/* ---- query_ball_point: modules/pointnet2_utils.py:112-134.  First `nsample` base indices with
 * not (d > radius2), padded with the first hit; a row without hits is filled with N. */
int mpa_ball_query_f32(const float *base, const float *query, int B, int N, int S, int C,
                       float radius2, int nsample, int64_t *out_idx, void *stream);

/* ---- index_points: modules/pointnet2_utils.py:64-81.  points [B,N,C], idx [B,M] (M = S or S*K)
 * -> out [B,M,C].  Backward scatter-adds grad_out rows into grad_points (which the caller
 * has zeroed); duplicate indices are summed with float atomics. */
int mpa_gather_fwd_f32(const float *points, const int64_t *idx, int B, int N, int M, int C,
                       float *out, void *stream);
int mpa_gather_bwd_f32(const float *grad_out, const int64_t *idx, int B, int N, int M, int C,
                       float *grad_points, void *stream);

/* ---- LocalTrans, feature branch: the difference-wise attention core,
 * modules/pointnet2_utils.py:548-569 (== repsurface_utils.py:515-535).
 * q [B,S,C] row view with leading dimension ldq floats (a column block of stacked projections
 * is read in place); k and v are row views with leading dimension ldkv floats (k, v = two column
 * blocks of one projected [B,N,ldkv] tensor, or separate tensors with ldkv = C);
 * idx [B,S,K] neighbours into the N base rows.  Per (b,s,c):
 *   e_j = (q - k[idx_j]) / sqrt(C);  a = softmax_j(e);  w_j = a_j - sum_j a_j;
 *   ctx = max_j w_j * v[idx_j]
 * ctx [B,S,C]; argk [B,S,C] uint8 receives the arg-max j (saved for backward). K <= 16. */
int mpa_diffattn_fwd_f32(const float *q, int ldq, const float *k, const float *v, int ldkv,
                         const int64_t *idx, int B, int N, int S, int K, int C,
                         float *ctx, uint8_t *argk, void *stream);
/* backward (closed form): recomputes the softmax; grad_q ([B,S,C] row view with
 * q's leading dimension ldq) and
 * grad_k / grad_v ([B,N,C] row views, leading dimension ldg) are all overwritten -- no clearing
 * by the caller.  With a workspace of mpa_diffattn_bwd_workspace_bytes() the scatter through idx
 * is done without float atomics (inverted neighbour table + per-slot gradients, summed per base
 * row); with workspace == NULL (or too small, or a shape for which the size
 * query returns 0) it falls back to global float atomics after clearing grad_k / grad_v. */
size_t mpa_diffattn_bwd_workspace_bytes(int B, int N, int S, int K, int C);
int mpa_diffattn_bwd_f32(const float *q, int ldq, const float *k, const float *v, int ldkv,
                         const int64_t *idx, const uint8_t *argk, const float *grad_ctx,
                         int B, int N, int S, int K, int C,
                         float *grad_q, float *grad_k, float *grad_v, int ldg,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ---- LocalTrans, xyz branch: modules/pointnet2_utils.py:518-544.  k and v are Linear(3->C)
 * applied to neighbour offsets, so the whole branch is one kernel on raw coordinates:
 *   q = Wq c + bq;  k_j = Wk (x[idx_j] - c) + bk;  v_j = Wv (x[idx_j] - c) + bv;  then as above.
 * xyz [B,N,3] base coordinates, center [B,S,3]; W* [C,3] row-major (nn.Linear.weight), b* [C]. */
int mpa_diffattn_xyz_fwd_f32(const float *xyz, const float *center, const int64_t *idx,
                             const float *Wq, const float *bq, const float *Wk, const float *bk,
                             const float *Wv, const float *bv,
                             int B, int N, int S, int K, int C,
                             float *ctx, uint8_t *argk, void *stream);
/* backward: accumulates (atomics; caller zeroes) gWq,gWk,gWv [C,3] and gbq,gbk,gbv [C].
 * Coordinates are inputs of the network and receive no gradient. */
int mpa_diffattn_xyz_bwd_f32(const float *xyz, const float *center, const int64_t *idx,
                             const float *Wq, const float *bq, const float *Wk, const float *bk,
                             const float *Wv, const float *bv, const uint8_t *argk,
                             const float *grad_ctx, int B, int N, int S, int K, int C,
                             float *gWq, float *gbq, float *gWk, float *gbk, float *gWv, float *gbv,
                             void *stream);

/* ---- Linear: the transition / pointwise MLP unit, modules/pointnet2_utils.py:401-425
 * (nn.Linear -> BatchNorm1d over the B*S rows -> LeakyReLU(0.2)).
 * GEMMs run on fp32 MFMA (v_mfma_f32_32x32x2_f32: bit-equal to an fmaf chain over k).
 *   mpa_gemm_f32:  C[M,N] = op(A) * op(B) (+ bias[N]) (+ C if accumulate)
 *       transA = 0: A is [M,K] (lda);  transA = 1: A is stored [K,M] (lda)
 *       transB = 0: B is [K,N] (ldb);  transB = 1: B is stored [N,K] (ldb)  (nn.Linear.weight)
 *   tile_stats (optional): [ceil(M/64)][2][N] floats, fully written by the epilogue: for every
 *   64-row tile and column the sum and the sum of squared deviations from the tile mean -- the
 *   BatchNorm batch statistics in Chan's pairwise form (no atomics: deterministic; no
 *   E[y^2]-E[y]^2 cancellation).  mpa_bn_finalize_f32 combines them.
 *   stats_replicas = R > 0 selects the ACCUMULATE form instead: tile_stats is [R][3][N] floats, pre-zeroed by the
 *   caller, and every 64-row tile adds (float atomics, replica = tile index mod R) its sum, its within-tile M2 and
 *   sum^2/rows per column; mpa_bn_stats_act_fwd_f32 finishes mean / variance from them in its own prologue, so
 *   the Linear unit needs no separate finalize launch (the between-tile part of the variance is then taken from
 *   the tile sums: exact within tiles, ~1e-7 * mean^2/var relative between them).
 *   a_col_sum (optional, [M], cleared by the caller): receives sum_k op(A)[m][k] with float atomics
 *   -- the bias gradient, for free, when op(A) = dY^T in the weight-gradient product.
 *   workspace (optional, 16-B aligned): scratch for split-K partial tiles (weight gradients have
 *   K = B*S rows and a tiny output); without it split-K falls back to float atomics into C.
 *   accumulate != 0 adds into C. */
int mpa_gemm_f32(const float *A, int lda, int transA, const float *B, int ldb, int transB,
                 const float *bias, float *C, int ldc, int M, int N, int K, int accumulate,
                 float *tile_stats, int stats_replicas, float *a_col_sum, float *workspace, size_t workspace_bytes,
                 void *stream);
/* Grouped forward / dX products: `count` (<= 8) INDEPENDENT  C_p[M,N] = A_p[M,K] op(B_p) (+ bias_p)  in one launch
 * (the Linear units of LocalMerge's parallel attention streams, modules/pointnet2_utils.py:465-470, and of Fuse's
 * four source states, :617-705: small problems that each leave the chip half empty).  All problems share transB;
 * tile_stats / stats_replicas as in mpa_gemm_f32 (accumulate form only when stats_replicas > 0).  No split-K.
 * `problems` is a HOST array (copied into the kernel arguments).  The _bf16 form takes bf16 A and C (B bf16 or
 * fp32 as b_is_f32 says). */
typedef struct MpaGemmProblem {
    const void *A;
    const void *B;
    const float *bias;
    void *C;
    float *tile_stats;
    int lda, ldb, ldc, M, N, K, stats_replicas;
} MpaGemmProblem;
int mpa_gemm_grouped_f32(const MpaGemmProblem *problems, int count, int transB, void *stream);
/* Grouped BatchNorm launches for `count` (<= 8) INDEPENDENT Linear units (the units mpa_gemm_grouped_* computed):
 * forward  y_u = residual_u + lrelu(bn_u(x_u)) exactly as mpa_bn_stats_act_fwd_*, one launch for all units
 * (sum_mode != 0, all units [M][C]: units[count-1].y = units[0].residual + sum_u lrelu(bn_u(x_u)), added in unit order --
 * Fuse's sum over its source states, modules/pointnet2_utils.py:617-705); backward reduce / apply as
 * mpa_bn_act_bwd_reduce_* / _apply_* with mean | invstd taken from `save`.  C % 4 == 0, 16-byte aligned rows
 * (MPA_EUNSUPPORTED otherwise: use the single-unit entries).  `units` is a HOST array. */
typedef struct MpaBnUnit {
    const void *x;                 /* [M][C] rows before normalisation (fp32 or bf16 as the entry says) */
    const float *stats;            /* forward, training: [stats_replicas][3][C] accumulated tile statistics */
    float *running_mean, *running_var;
    int64_t *num_batches_tracked;
    const float *gamma, *beta;
    const void *residual;          /* forward, optional */
    void *y;                       /* forward output */
    float *save;                   /* [2][C] mean | invstd: written by forward, read by backward */
    const void *grad_y;            /* backward: upstream gradient, leading dimension ldg */
    float *partial;                /* backward: [replicas][2][C], zero on entry of the reduce pass */
    void *grad_x;                  /* backward output */
    float *dgamma, *dbeta;         /* backward outputs (optional) */
    int M, C, stats_replicas, ldg, training, replicas;
    float momentum, eps, slope;
    int ldy;                       /* forward: leading dimension of y (0 = C): units may write column blocks of one tensor */
} MpaBnUnit;
int mpa_bn_group_fwd_f32(const MpaBnUnit *units, int count, int sum_mode, void *stream);
int mpa_bn_group_bwd_reduce_f32(const MpaBnUnit *units, int count, void *stream);
int mpa_bn_group_bwd_apply_f32(const MpaBnUnit *units, int count, void *stream);
/* Grouped weight gradients: out_p[M,N] = A_p^T B_p for `count` independent problems in one launch
 * (+ one reduce launch), A_p stored [K][M] (lda), B_p stored [K][N] (ldb): dW = dY^T X of every
 * Linear of a backward pass.  Each is a latency-bound stream with a tiny output, so they are
 * overlapped instead of launched one by one.  a_col_sum (optional, [M], cleared by the caller)
 * receives the column sums of A (the bias gradient).  `problems` is a HOST array (copied into
 * the kernel arguments); workspace holds the split-K partial tiles. */
typedef struct MpaGemmTnProblem {
    const float *A;
    const float *B;
    float *out;
    float *a_col_sum;
    int lda, ldb, M, N, K;
} MpaGemmTnProblem;
int mpa_gemm_tn_grouped_f32(const MpaGemmTnProblem *problems, int count, float *workspace,
                            size_t workspace_bytes, void *stream);
/* mpa_fps_knn_feat_f32 with a SECOND coordinate search (xyz2_*: base [B,zN,3], query [B,zS,3], zK <= 8; NULL: none) in
 * the same launch -- round 3: the sampling and the state-0 search of the NEXT batch ride beside this batch's state-1
 * searches (cross-step pipelining of the geometry, ops.GeometryPipeline; reference modules/repsurface_utils.py:581-619).
 * fps_xyz == NULL: no sampling workgroups.  Every result equals the stand-alone entry point's bit for bit. */
int mpa_geo_level_f32(const float *fps_xyz, int B, int fps_N, int fps_S, const int64_t *start_idx,
                      int64_t *fps_idx, float *fps_out_xyz, const float *xyz_base, const float *xyz_query,
                      int xN, int xS, int xK, float *xyz_dist, int64_t *xyz_idx, const float *xyz2_base,
                      const float *xyz2_query, int zN, int zS, int zK, float *xyz2_dist, int64_t *xyz2_idx,
                      const float *feat_base, const float *feat_norms, const float *feat_query, int N, int S, int C,
                      int K, float *out_dist, int64_t *out_idx, void *stream);
/* A COARSE state's geometry step in one launch (csrc/knn_fused.hip): the sampling of the next state (optional:
 * fps_xyz != NULL; fps_S from fps_N <= 256 points), the coordinate search (optional: xyz_base != NULL; xN <= 256) and
 * the feature-space search (N <= 256 base rows of C in {32,64,128,256} floats, 16-byte aligned), all K <= 8 -- the
 * last states of a chain (128 -> 64 -> 32 points in the classification model), where every launch of the general
 * kernels is latency.  Same outputs, bit for bit, as mpa_fps_f32 + mpa_knn_f32 x 2.  MPA_EUNSUPPORTED outside these
 * shapes. */
int mpa_coarse_level_f32(const float *fps_xyz, int B, int fps_N, int fps_S, const int64_t *start_idx,
                         int64_t *fps_idx, float *fps_out_xyz, const float *xyz_base, const float *xyz_query,
                         int xN, int xS, int xK, float *xyz_dist, int64_t *xyz_idx, const float *feat_base,
                         const float *feat_query, int N, int S, int C, int K, float *out_dist, int64_t *out_idx,
                         void *stream);
/* Geometry rider: sampling levels and one coordinate search of the NEXT batch, carried by the first workgroups of a
 * long launch of the current step (cross-step co-scheduling of the FPS chain, which depends on coordinates only:
 * modules/repsurface_utils.py:581-619, modules/pointnet2_utils.py:84-109, :211-222).
 *   sampling (nlev in 0..4 levels, one workgroup per cloud runs them all in sequence):
 *     level 0 samples S[0] points from src [B,N,3] (N <= 4096), level j > 0 samples S[j] <= S[j-1] points from level
 *     j-1's result; start[j] [B], idx[j] [B,S[j]] int64, xyz[j] [B,S[j],3] as in mpa_fps_f32;
 *   search (optional, base != NULL): knn_point(sK <= 8, base [B,sN,3], query [B,sS,3]) -> dist / kidx [B,sS,sK],
 *     as mpa_knn_f32 with C = 3 (it must not depend on this rider's own sampling results).
 * Results are bit-identical to the stand-alone entry points. */
typedef struct MpaGeoRider {
    const float *src;
    int B, N, nlev;
    int S[4];
    const int64_t *start[4];
    int64_t *idx[4];
    float *xyz[4];
    const float *base;
    const float *query;
    int sN, sS, sK;
    float *dist;
    int64_t *kidx;
    int *queue;     /* 16 ints, ZEROED by the caller before every call that carries the rider (work counters, finished
                     * sampling workgroups); NULL: the rider is never carried, it goes out as a launch of its own */
} MpaGeoRider;
/* mpa_gemm_tn_grouped_f32 with riders: rider i travels in the i-th weight-gradient launch of the call (40 problems per
 * launch), riders beyond the number of launches are issued afterwards as launches of their own, in order -- so a
 * rider may consume the results of the riders before it.  `riders` is a HOST array. */
int mpa_gemm_tn_grouped_rider_f32(const MpaGemmTnProblem *problems, int count, float *workspace,
                                  size_t workspace_bytes, const MpaGeoRider *riders, int nriders, void *stream);
/* one rider as a launch of its own (the first batch of a run, which has no previous step to ride in) */
int mpa_geo_rider_f32(const MpaGeoRider *rider, void *stream);
/* tile statistics (same format as the GEMM epilogue's) of an existing tensor x [M,C]. */
int mpa_tile_stats_f32(const float *x, int M, int C, float *tile_stats, void *stream);
/* per-column sum and sum of squares of x [M,C] -> col_sum, col_sumsq [C] (caller zeroes). */
int mpa_col_stats_f32(const float *x, int M, int C, float *col_sum, float *col_sumsq, void *stream);
/* BatchNorm1d statistics: save_mean_invstd [2][C] <- (mean, 1/sqrt(var+eps)).  training != 0:
 * from tile_stats over the M rows (biased variance), and running_mean/var (may be NULL) are
 * updated with `momentum` (unbiased variance) as nn.BatchNorm1d does; training == 0: from the
 * running statistics (tile_stats ignored).  zero_buf (optional): zero_count floats cleared by the
 * same launch -- the [2][C] accumulator that mpa_bn_act_bwd_reduce_f32 later adds into.
 * num_batches_tracked (optional): nn.BatchNorm1d's int64 counter, incremented when training. */
int mpa_bn_finalize_f32(const float *tile_stats, int M, int C, float *running_mean, float *running_var,
                        int training, float momentum, float eps, float *save_mean_invstd, float *zero_buf,
                        int zero_count, int64_t *num_batches_tracked, void *stream);
/* y = residual + leaky_relu((x - mean[c]) * invstd[c] * gamma[c] + beta[c], slope) over [M,C]
 * (slope = 1: no activation; residual may be NULL -- it is LocalTrans' `residual + ffn(context)`,
 * modules/pointnet2_utils.py:572, fused into the same pass).  In place allowed (y == x). */
int mpa_bn_act_fwd_f32(const float *x, const float *save_mean_invstd, const float *gamma, const float *beta,
                       const float *residual, float slope, int M, int C, float *y, void *stream);
/* The same with the statistics finished inside the launch from the accumulate form of the GEMM epilogue
 * (stats [replicas][3][C], see mpa_gemm_f32): every workgroup derives mean / 1/sqrt(var+eps) in its prologue,
 * workgroup 0 also stores them to save_mean_invstd [2][C] (for the backward pass) and updates running_mean /
 * running_var / num_batches_tracked as nn.BatchNorm1d does.  training == 0: running statistics (stats unused). */
int mpa_bn_stats_act_fwd_f32(const float *x, const float *stats, int replicas, int M, int C,
                             float *running_mean, float *running_var, int training, float momentum, float eps,
                             int64_t *num_batches_tracked, const float *gamma, const float *beta,
                             const float *residual, float slope, float *y, float *save_mean_invstd, void *stream);
/* out[c] += sum over the M rows of x[r*ld + c]  (bias gradients; out [C] cleared by the caller). */
int mpa_col_sum_f32(const float *x, int M, int C, int ld, float *out, void *stream);
/* out[g][c] = sum over the R rows of group g of x[(g*R + r)*ld + c], x [G*R, >= C] (leading dimension ld), out [G,C]
 * fp32, fully written (no atomics, no clearing): the gradient of a per-cloud row broadcast over the cloud's
 * points -- the global-feature and label-embedding columns the part-seg head concatenates to every point
 * (modules/pointnet2_utils.py:846-856). */
int mpa_group_col_sum_f32(const float *x, int G, int R, int C, int ld, float *out, void *stream);
/* backward of y = lrelu(bn(x)).  Pass 1 accumulates, with float atomics spread over `replicas`
 * copies, partial[r][0][c] += sum g and partial[r][1][c] += sum g*xhat (g = grad_y * lrelu'(.);
 * partial [replicas][2][C] pre-zeroed, e.g. by mpa_bn_finalize_f32's zero_buf).  Pass 2 sums the
 * replicas, writes grad_x = gamma*invstd*(g - sum_g/M - xhat*sum_gxhat/M) (use_batch_stats != 0)
 * or gamma*invstd*g (running statistics), and stores dbeta = sum g, dgamma = sum g*xhat
 * (either may be NULL).  grad_y rows have leading dimension ldg floats (a column block of a wider
 * gradient, e.g. one input of a concatenation, is read in place); x and grad_x are dense [M,C]. */
int mpa_bn_act_bwd_reduce_f32(const float *x, const float *grad_y, const float *mean, const float *invstd,
                              const float *gamma, const float *beta, float slope, int M, int C,
                              int ldg, float *partial, int replicas, void *stream);
int mpa_bn_act_bwd_apply_f32(const float *x, const float *grad_y, const float *mean, const float *invstd,
                             const float *gamma, const float *beta, const float *partial, int replicas,
                             float slope, int use_batch_stats, int M, int C, int ldg, float *grad_x,
                             float *dgamma, float *dbeta, void *stream);

/* ---- umbrella surface features (RepSurf front-end; reference modules/repsurface_utils.py:106-126,
 * :321-376, modules/recons_utils.py:27-57,82-90,108-124,152-176, modules/polar_utils.py:10-31).
 * xyz [B,N,3]; knn_idx [B,N,K] = knn_point(K, xyz, xyz) (entry 0, the point itself, is dropped).
 * Per point the K-1 neighbour offsets are sorted by azimuth and paired cyclically into triangles
 * (centre, p_i, p_i+1); out [B,N,K-1,CH] = centre of gravity (3) | its spherical coordinates
 * rho, theta/pi, phi/2pi+0.5 (3) | unit normal, flipped so that the first triangle's x is positive
 * and by cloud_sign[b] (+-1, the reference's per-cloud random inversion; NULL = +1) (3) | and, if
 * return_dist, normal.centre/sqrt(3) (1): CH = 10 or 9.  Degenerate triangles (NaN normal) take
 * normal, centre and constant of the point's first valid triangle.  K - 1 <= 16.  No backward. */
int mpa_umbrella_features_f32(const float *xyz, const int64_t *knn_idx, int B, int N, int K,
                              const float *cloud_sign, int return_dist, float *out, void *stream);

/* ---- upsample: the decoder's coarse->fine transition, modules/pointnet2_utils.py:13-50.
 * points [B,S,C], knn_idx [B,S,K] with values < Nf (= S*scale_ratio).  out [B,Nf,C] is the
 * mean, over the coarse rows s that list fine point n, of points[s]; the divisor counts only
 * contributors whose channel-0 value is non-zero (0 -> 1), uncovered fine points stay 0
 * (the reference's quirks, :44-46).  The dense [B,S,Nf,C] tensor is never formed.
 * cnt [B,Nf] float (output, kept for backward).  out and cnt are fully written by the callee.
 * With a workspace of mpa_upsample_workspace_bytes() bytes (16-byte aligned device memory, scratch
 * for the call) the forward is atomic-free: the neighbour table is inverted on the device and each
 * fine point gathers its coarse rows.  workspace NULL / too small (or the size query returning 0:
 * Nf beyond the inverter's per-cloud limit) selects the scatter path with float atomics. */
size_t mpa_upsample_workspace_bytes(int B, int S, int K, int Nf);
int mpa_upsample_mean_fwd_f32(const float *points, const int64_t *knn_idx, int B, int S, int K,
                              int Nf, int C, float *out, float *cnt, void *workspace,
                              size_t workspace_bytes, void *stream);
int mpa_upsample_mean_bwd_f32(const float *grad_out, const int64_t *knn_idx, const float *cnt,
                              int B, int S, int K, int Nf, int C, float *grad_points,
                              void *stream);

/* Global max over the points of a state: out[b][c] = max_n x[b][n][c], arg[b][c] = the first row attaining it
 * (`t.max(dim=1, keepdim=True)[0]` of the part-seg head, modules/pointnet2_utils.py:846-850), and its backward
 * grad_x[b][n][c] = (n == arg[b][c]) ? grad_out[b][c] : 0 (fully written).  A NaN in a column is propagated (arg = its
 * first NaN row), as torch.max does. */
int mpa_max_points_fwd_f32(const float *x, int B, int N, int C, float *out, int *arg, void *stream);
int mpa_max_points_bwd_f32(const float *grad_out, const int *arg, int B, int N, int C, float *grad_x, void *stream);
/* ---- PointNetFeaturePropagation interpolation: modules/pointnet2_utils.py:899-906.
 * three_nn = mpa_knn_f32 with K = 3 (query = xyz1, base = xyz2).
 * out[b,n,:] = sum_j w_j * points2[b, idx[b,n,j], :],  w_j = (1/(d_j+1e-8)) / sum_j (1/(d_j+1e-8)). */
int mpa_three_interp_fwd_f32(const float *points2, const int64_t *idx, const float *dist,
                             int B, int Nq, int Nb, int C, float *out, void *stream);
int mpa_three_interp_bwd_f32(const float *grad_out, const int64_t *idx, const float *dist,
                             int B, int Nq, int Nb, int C, float *grad_points2, void *stream);
/* the same on bf16 features (north_star: "the matching upsample/interpolate decoder" on the bf16 feature stream):
 * weights and sums in fp32, one rounding per output; the backward adds into a CLEARED fp32 grad_points2 (rows are
 * listed by several query points), which the caller rounds once. */
int mpa_three_interp_fwd_bf16(const mpa_bf16 *points2, const int64_t *idx, const float *dist,
                              int B, int Nq, int Nb, int C, mpa_bf16 *out, void *stream);
int mpa_three_interp_bwd_bf16(const mpa_bf16 *grad_out, const int64_t *idx, const float *dist,
                              int B, int Nq, int Nb, int C, float *grad_points2, void *stream);

/* ---- the tails of the models' heads, one launch each way (csrc/head.hip).
 * log_softmax over the class axis of x [M,C] (models/repsurf/repsurf_ssg_umb.py:67) and its backward
 * grad_x = grad_y - exp(y) * sum_c grad_y. */
int mpa_log_softmax_fwd_f32(const float *x, int M, int C, float *y, void *stream);
int mpa_log_softmax_bwd_f32(const float *y, const float *grad_y, int M, int C, float *grad_x, void *stream);
/* Label-smoothed loss, mean over the M rows of -sum_c w_c lp_c with w = 1-eps at the target class and eps/(C-1)
 * elsewhere: from_logits = 0: x are log-probabilities (SmoothClsLoss, util/utils.py:74-88); from_logits = 1: x are
 * logits, lp = log_softmax(x) (get_loss, models/repsurf/pointnet2_part_seg_msg.py:159-180; lse [M] receives the rows'
 * log-sum-exp for the backward).  partial: mpa_smooth_loss_workspace_floats(M) floats of scratch (per-block sums,
 * combined in a fixed order: the loss is bit-reproducible).  loss [1].  Backward: grad_x [M,C] = grad_loss[0] *
 * d loss / d x. */
int mpa_smooth_loss_workspace_floats(int M);
int mpa_smooth_loss_fwd_f32(const float *x, const int64_t *target, int M, int C, float eps, int from_logits,
                            float *lse, float *partial, float *loss, void *stream);
int mpa_smooth_loss_bwd_f32(const float *x, const int64_t *target, const float *lse, const float *grad_loss,
                            int M, int C, float eps, int from_logits, float *grad_x, void *stream);
/* cat(max over the P points, mean over the P points) of x [B,P,C] -> out [B,2C] (the classification head's pooling,
 * modules/repsurface_utils.py:629-633); arg [B,C] = first row attaining the maximum (a NaN is kept, as torch.max).
 * Backward: grad_x [B,P,C] fully written. */
int mpa_pool_max_mean_fwd_f32(const float *x, int B, int P, int C, float *out, int *arg, void *stream);
int mpa_pool_max_mean_bwd_f32(const float *grad_out, const int *arg, int B, int P, int C, float *grad_x, void *stream);

/* out[r][c] = sum_u srcs[u][r * lds[u] + c] for n <= 8 sources of `rows` rows of C elements (host-side arrays of n
 * pointers and n row strides in elements, lds == NULL: C; C % 4 == 0, rows 16-byte (fp32) / 8-byte (bf16) aligned; out
 * is dense): the gradient of a tensor with several consumers in ONE pass.  Replaces the n - 1 pairwise torch.add
 * launches of autograd's gradient accumulation (reference: implicit in every tensor the modules read more than once,
 * e.g. modules/pointnet2_utils.py:548-569 q / residual, :798-837 Fuse inputs).  fp32 accumulation. */
int mpa_add_n_f32(const float *const *srcs, const long long *lds, int n, long long rows, int C, float *out, void *stream);
int mpa_add_n_bf16(const mpa_bf16 *const *srcs, const long long *lds, int n, long long rows, int C, mpa_bf16 *out,
                   void *stream);

/* ---- optimizer step over flat buckets (the training loop of tool/train_cls_scanobjectnn.py:205-216
 * uses torch.optim.Adam; gradients here live in a few flat buffers, so one elementwise pass per
 * bucket replaces ~300 per-parameter launches).  torch.optim.Adam arithmetic, no amsgrad.
 * `step` is a device scalar holding the (already advanced) step count t.  `hyper` (optional) is a device
 * array [lr, weight_decay] that overrides the by-value `lr` / `weight_decay`: values passed by value are
 * frozen into a captured HIP graph, device scalars can be rewritten between replays (learning-rate
 * schedules: tool/train_cls_scanobjectnn.py:219-238, tool/train_partseg.py:152-221). */
int mpa_scalar_add_f32(float *x, float a, void *stream);
int mpa_adam_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n,
                      float lr, float beta1, float beta2, float eps, float weight_decay,
                      const float *step, const float *hyper, void *stream);

/* ==================================================================================================
 * bf16 feature path (BASELINE configs 3 and 5): the same entry points for FEATURES stored
 * as bf16 -- activations and their gradients are bf16 in HBM (the path is HBM-bound: half the bytes),
 * every kernel computes in fp32 registers, and coordinates, distances, indices, BatchNorm statistics,
 * parameters, parameter gradients and optimizer state stay fp32.  The reference has no reduced-precision
 * mode (no autocast / GradScaler anywhere), so these follow the SAME formulas as the fp32
 * entry points above (same reference file:line), rounding only where a feature tensor is stored.
 * Shapes must allow the vector paths (channel counts multiples of 4, 8-byte aligned rows) unless stated.
 * ================================================================================================== */

/* Linear (modules/pointnet2_utils.py:413-418) on v_mfma_f32_32x32x16_bf16, fp32 accumulation:
 *   C[M,N] = A[M,K] * op(B) (+ bias[N]);  A bf16 row-major (lda);
 *   transB = 1: B stored [N][K] (nn.Linear.weight -- the forward product x W^T);
 *   transB = 0: B stored [K][N] (the same weight walked along its rows: dX = dY W);
 *   b_is_f32: B holds fp32 (the master parameters, converted while staged) instead of bf16;
 *   c_is_f32: C receives fp32 (the logits handed to the loss) instead of bf16.
 * tile_stats (optional): [ceil(M/64)][2][N] floats, the BatchNorm batch statistics of the UNROUNDED fp32
 * results in the format mpa_bn_finalize_f32 consumes (see mpa_gemm_f32).  Any M, N, K; rows that are not
 * 16-byte aligned take an element-wise path. */
int mpa_gemm_bf16(const mpa_bf16 *A, int lda, const void *B, int ldb, int transB, int b_is_f32,
                  const float *bias, void *C, int ldc, int c_is_f32, int M, int N, int K,
                  float *tile_stats, int stats_replicas, void *stream);
/* Grouped weight gradients on bf16 operands: out_p[M,N] (fp32) = A_p^T B_p, A_p [K][M] (lda), B_p [K][N] (ldb)
 * bf16 -- dW = dY^T X of every Linear of a backward pass in one launch (+ one reduce launch); same contract
 * as mpa_gemm_tn_grouped_f32 (a_col_sum optional, fp32, cleared by the caller; problems is a HOST array). */
typedef struct MpaGemmTnProblemBf16 {
    const mpa_bf16 *A;
    const mpa_bf16 *B;
    float *out;
    float *a_col_sum;
    int lda, ldb, M, N, K;
} MpaGemmTnProblemBf16;
int mpa_gemm_tn_grouped_bf16(const MpaGemmTnProblemBf16 *problems, int count, float *workspace,
                             size_t workspace_bytes, void *stream);
/* BatchNorm1d + LeakyReLU (+ residual) over bf16 rows; statistics, gamma / beta and their gradients fp32
 * (mpa_bn_finalize_f32 is shared).  Same semantics as the _f32 entry points. */
int mpa_bn_act_fwd_bf16(const mpa_bf16 *x, const float *save_mean_invstd, const float *gamma, const float *beta,
                        const mpa_bf16 *residual, float slope, int M, int C, mpa_bf16 *y, void *stream);
int mpa_bn_stats_act_fwd_bf16(const mpa_bf16 *x, const float *stats, int replicas, int M, int C,
                              float *running_mean, float *running_var, int training, float momentum, float eps,
                              int64_t *num_batches_tracked, const float *gamma, const float *beta,
                              const mpa_bf16 *residual, float slope, mpa_bf16 *y, float *save_mean_invstd,
                              void *stream);
int mpa_bn_act_bwd_reduce_bf16(const mpa_bf16 *x, const mpa_bf16 *grad_y, const float *mean, const float *invstd,
                               const float *gamma, const float *beta, float slope, int M, int C,
                               int ldg, float *partial, int replicas, void *stream);
int mpa_bn_act_bwd_apply_bf16(const mpa_bf16 *x, const mpa_bf16 *grad_y, const float *mean, const float *invstd,
                              const float *gamma, const float *beta, const float *partial, int replicas,
                              float slope, int use_batch_stats, int M, int C, int ldg, mpa_bf16 *grad_x,
                              float *dgamma, float *dbeta, void *stream);
/* index_points (:64-81) on bf16 rows.  Backward scatter-adds the bf16 gradient rows into an fp32
 * accumulator (grad_points fp32, cleared by the caller): sums over duplicated indices stay fp32. */
int mpa_gather_fwd_bf16(const mpa_bf16 *points, const int64_t *idx, int B, int N, int M, int C,
                        mpa_bf16 *out, void *stream);
int mpa_gather_bwd_bf16(const mpa_bf16 *grad_out, const int64_t *idx, int B, int N, int M, int C,
                        float *grad_points, void *stream);
/* The same scatter-add INTO a bf16 destination (not cleared: pass zeros for the plain backward of index_points, or a
 * gradient to add to): two channels per lane, compare-and-swap on the dword.  Exact for rows listed once; rows listed
 * several times are added one by one, each sum rounded to bf16.  C even, 4-byte aligned rows. */
int mpa_gather_bwd_into_bf16(const mpa_bf16 *grad_out, const int64_t *idx, int B, int N, int M, int C,
                             mpa_bf16 *grad_points, void *stream);
/* difference-wise attention (:548-569) on bf16 q / k / v / ctx (softmax, offset and max in fp32 registers).
 * Backward requires the workspace of mpa_diffattn_bwd_workspace_bytes_bf16() (per-slot key gradients are
 * kept in bf16, summed per base row in fp32, stored bf16; never float atomics on bf16). */
int mpa_diffattn_fwd_bf16(const mpa_bf16 *q, int ldq, const mpa_bf16 *k, const mpa_bf16 *v, int ldkv,
                          const int64_t *idx, int B, int N, int S, int K, int C,
                          mpa_bf16 *ctx, uint8_t *argk, void *stream);
size_t mpa_diffattn_bwd_workspace_bytes_bf16(int B, int N, int S, int K, int C);
int mpa_diffattn_bwd_bf16(const mpa_bf16 *q, int ldq, const mpa_bf16 *k, const mpa_bf16 *v, int ldkv,
                          const int64_t *idx, const uint8_t *argk, const mpa_bf16 *grad_ctx,
                          int B, int N, int S, int K, int C,
                          mpa_bf16 *grad_q, mpa_bf16 *grad_k, mpa_bf16 *grad_v, int ldg,
                          void *workspace, size_t workspace_bytes, void *stream);
/* xyz branch (:518-544): coordinates and projection weights fp32, ctx (and its gradient) bf16 */
int mpa_diffattn_xyz_fwd_bf16(const float *xyz, const float *center, const int64_t *idx,
                              const float *Wq, const float *bq, const float *Wk, const float *bk,
                              const float *Wv, const float *bv,
                              int B, int N, int S, int K, int C,
                              mpa_bf16 *ctx, uint8_t *argk, void *stream);
int mpa_diffattn_xyz_bwd_bf16(const float *xyz, const float *center, const int64_t *idx,
                              const float *Wq, const float *bq, const float *Wk, const float *bk,
                              const float *Wv, const float *bv, const uint8_t *argk,
                              const mpa_bf16 *grad_ctx, int B, int N, int S, int K, int C,
                              float *gWq, float *gbq, float *gWk, float *gbk, float *gWv, float *gbv,
                              void *stream);
/* upsample (:13-50) on bf16 rows; the workspace of mpa_upsample_workspace_bytes() is required (the
 * inverted-table gather; no atomic path on bf16).  cnt stays fp32. */
int mpa_upsample_mean_fwd_bf16(const mpa_bf16 *points, const int64_t *knn_idx, int B, int S, int K,
                               int Nf, int C, mpa_bf16 *out, float *cnt, void *workspace,
                               size_t workspace_bytes, void *stream);
int mpa_upsample_mean_bwd_bf16(const mpa_bf16 *grad_out, const int64_t *knn_idx, const float *cnt,
                               int B, int S, int K, int Nf, int C, mpa_bf16 *grad_points,
                               void *stream);
int mpa_group_col_sum_bf16(const mpa_bf16 *x, int G, int R, int C, int ld, float *out, void *stream);
int mpa_gemm_grouped_bf16(const MpaGemmProblem *problems, int count, int transB, int b_is_f32, void *stream);
int mpa_max_points_fwd_bf16(const mpa_bf16 *x, int B, int N, int C, mpa_bf16 *out, int *arg, void *stream);
int mpa_max_points_bwd_bf16(const mpa_bf16 *grad_out, const int *arg, int B, int N, int C, mpa_bf16 *grad_x, void *stream);
int mpa_bn_group_fwd_bf16(const MpaBnUnit *units, int count, int sum_mode, void *stream);
int mpa_bn_group_bwd_reduce_bf16(const MpaBnUnit *units, int count, void *stream);
int mpa_bn_group_bwd_apply_bf16(const MpaBnUnit *units, int count, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MPA_HIP_H */
