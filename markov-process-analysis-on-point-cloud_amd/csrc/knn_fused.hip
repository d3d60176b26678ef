// Farthest point sampling of the NEXT state, the coordinate search of THIS state and its feature-space search in one
// launch.  FPS is S dependent iterations on one workgroup per cloud (32-64 of 256 CUs; 0.34 ms of a cls step, 0.79 ms
// of a part-seg step, not overlapped with anything under HIP-graph replay); the searches of the state before it
// depend on other data and fill the rest of the chip meanwhile -- the launch lasts as long as the longer of the two.
// Workgroups [0, B): fps_body; then the coordinate search's; then the feature search's (bodies of knn.hip).
#define MPA_KNN_BODIES_ONLY
#include "knn.hip"

namespace {

struct FusedSearch {
    const float *base, *query, *norms;
    float *dist;
    int64_t *idx;
    int N, S, K, qb, blocks;               // qb: query workgroups per cloud
};

template <int P, int CT, int QG, bool GN>
__global__ __launch_bounds__(256) void fps_knn2_kernel(const float *__restrict__ fxyz, int fN, int fS,
                                                       const int64_t *__restrict__ start, int64_t *__restrict__ f_idx,
                                                       float *__restrict__ f_out_xyz, int B, FusedSearch x, FusedSearch y)
{
    extern __shared__ float lds[];
    int bid = blockIdx.x;
    if (bid < B) {
        fps_body<4, P>(fxyz, fN, fS, start, f_idx, f_out_xyz, bid, lds);
        return;
    }
    bid -= B;
    if (bid < x.blocks) {
        knn_mfma_body<3, 4, 8, 1, false, false>(x.base, x.query, x.N, x.S, x.K, x.dist, x.idx, bid % x.qb, bid / x.qb, lds);
        return;
    }
    bid -= x.blocks;
    knn_mfma_body<CT, 4, 8, QG, false, GN>(y.base, y.query, y.N, y.S, y.K, y.dist, y.idx, bid % y.qb, bid / y.qb, lds,
                                           y.norms);
}

template <int CT, int QG>
constexpr size_t feat_lds_bytes()
{
    constexpr int CP = (CT + 3) & ~3, NQ = 32 * QG;
    constexpr size_t work = ((size_t)4 * (32 * (CP + 4) + 32) + KNN_G * NQ + NQ + NQ + 32 + 2 * NQ * KNN_CAP) * sizeof(float);
    constexpr size_t merge = (size_t)32 * 2 * 4 * 8 * 8;
    return work > merge ? work : merge;
}

template <int P, int CT, int QG, bool GN>
int launch_fused(const float *fxyz, int B, int fN, int fS, const int64_t *start, int64_t *f_idx, float *f_out_xyz,
                 const FusedSearch &x, const FusedSearch &y, hipStream_t st)
{
    constexpr size_t xyz_lds = ((size_t)4 * (32 * (4 + 4) + 32) + KNN_G * 32 + 32 + 64 + 2 * 32 * KNN_CAP) * sizeof(float);
    constexpr size_t fps_lds = (size_t)3 * 256 * P * sizeof(float) + 2 * 4 * sizeof(uint2);
    constexpr size_t a = feat_lds_bytes<CT, QG>() > xyz_lds ? feat_lds_bytes<CT, QG>() : xyz_lds;
    constexpr size_t lds = a > fps_lds ? a : fps_lds;
    static_assert(lds <= 160 * 1024, "LDS of a gfx950 CU");
    if (lds > 64 * 1024) {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(&fps_knn2_kernel<P, CT, QG, GN>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (attr != hipSuccess) return MPA_EHIP;
    }
    const long long blocks = (long long)B + x.blocks + y.blocks;
    if (blocks >= 0x7fffffffLL) return MPA_EUNSUPPORTED;
    hipLaunchKernelGGL((fps_knn2_kernel<P, CT, QG, GN>), dim3((unsigned)blocks), dim3(256), lds, st, fxyz, fN, fS, start,
                       f_idx, f_out_xyz, B, x, y);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

template <int CT, int QG, bool GN>
int launch_fused_p(int P, const float *fxyz, int B, int fN, int fS, const int64_t *start, int64_t *f_idx,
                   float *f_out_xyz, const FusedSearch &x, const FusedSearch &y, hipStream_t st)
{
    switch (P) {
    case 1: return launch_fused<1, CT, QG, GN>(fxyz, B, fN, fS, start, f_idx, f_out_xyz, x, y, st);
    case 2: return launch_fused<2, CT, QG, GN>(fxyz, B, fN, fS, start, f_idx, f_out_xyz, x, y, st);
    case 4: return launch_fused<4, CT, QG, GN>(fxyz, B, fN, fS, start, f_idx, f_out_xyz, x, y, st);
    default: return launch_fused<8, CT, QG, GN>(fxyz, B, fN, fS, start, f_idx, f_out_xyz, x, y, st);
    }
}

}  // namespace

extern "C" int mpa_fps_knn_feat_f32(const float *fps_xyz, int B, int fps_N, int fps_S, const int64_t *start_idx,
                                    int64_t *fps_idx, float *fps_out_xyz, const float *xyz_base,
                                    const float *xyz_query, int xN, int xS, int xK, float *xyz_dist, int64_t *xyz_idx,
                                    const float *feat_base, const float *feat_norms, const float *feat_query, int N,
                                    int S, int C, int K, float *out_dist, int64_t *out_idx, void *stream)
{
    MPA_CLEAR_ERROR();
    // fps_xyz == NULL: no sampling workgroups -- the two searches of a state alone in one launch (the sampling chain
    // of that pass was computed a step ahead by geometry riders, mpa_gemm_tn_grouped_rider_f32)
    const bool sample = fps_xyz != nullptr;
    if (!feat_base || !feat_query || !out_idx || B <= 0 || N <= 0 || S <= 0 || K <= 0) return MPA_EINVAL;
    if (sample && (!start_idx || !fps_idx || fps_N <= 0 || fps_S <= 0)) return MPA_EINVAL;
    if (xyz_base && (!xyz_query || !xyz_idx || xN <= 0 || xS <= 0 || xK <= 0)) return MPA_EINVAL;
    // the instantiated combinations: FPS of 129..2048 points, K <= 8, feature rows of 64 / 128 floats, 16-byte aligned
    if (K > 8 || K > N || (xyz_base && (xK > 8 || xK > xN)) || (sample && (fps_N <= 128 || fps_N > 2048)) ||
        (C != 64 && C != 128) || ((((uintptr_t)feat_base | (uintptr_t)feat_query | (uintptr_t)feat_norms) & 15) != 0))
        return MPA_EUNSUPPORTED;
    const int P = !sample || fps_N <= 256 ? 1 : (fps_N <= 512 ? 2 : (fps_N <= 1024 ? 4 : 8));
    const int fB = sample ? B : 0;               // sampling workgroups in front of the searches'
    FusedSearch x, y;
    x.base = xyz_base; x.query = xyz_query; x.norms = nullptr; x.dist = xyz_dist; x.idx = xyz_idx;
    x.N = xN; x.S = xS; x.K = xK; x.qb = xyz_base ? mpa_ceil_div(xS, 32) : 1; x.blocks = xyz_base ? x.qb * B : 0;
    y.base = feat_base; y.query = feat_query; y.norms = feat_norms; y.dist = out_dist; y.idx = out_idx;
    y.N = N; y.S = S; y.K = K;
    hipStream_t st = (hipStream_t)stream;
    if (C == 64 && feat_norms != nullptr) {                      // two query groups per workgroup (needs the norms)
        y.qb = mpa_ceil_div(S, 64); y.blocks = y.qb * B;
        return launch_fused_p<64, 2, true>(P, fps_xyz, fB, fps_N, fps_S, start_idx, fps_idx, fps_out_xyz, x, y, st);
    }
    y.qb = mpa_ceil_div(S, 32); y.blocks = y.qb * B;
    y.norms = nullptr;
    if (C == 64) return launch_fused_p<64, 1, false>(P, fps_xyz, fB, fps_N, fps_S, start_idx, fps_idx, fps_out_xyz, x, y, st);
    return launch_fused_p<128, 1, false>(P, fps_xyz, fB, fps_N, fps_S, start_idx, fps_idx, fps_out_xyz, x, y, st);
}
