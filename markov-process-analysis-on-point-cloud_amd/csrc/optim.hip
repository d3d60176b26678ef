// Adam over flat parameter / gradient buckets for gfx950.
// The training step keeps gradients in a few large flat buffers (distributed.GradReducer), so the
// optimizer is one HBM-bound elementwise pass per bucket over (param, grad, exp_avg, exp_avg_sq)
// instead of ~300 per-parameter launches.  Same arithmetic as torch.optim.Adam (no amsgrad):
//   g' = g + wd*p;  m = lerp(m, g', 1-b1);  v = b2*v + (1-b2)*g'^2;
//   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// The step count t lives on the device (so a captured HIP graph replays with the right bias
// correction); mpa_scalar_add_f32 advances it.  The learning rate and the weight decay can live on the
// device too (`hyper` = [lr, weight_decay]): by-value kernel arguments are frozen into a captured graph,
// and the reference's training loops change the rate every epoch (StepLR / CosineAnnealingLR,
// tool/train_cls_scanobjectnn.py:219-238, tool/train_partseg.py:152-221).
#include "mpa_common.h"

namespace {

__global__ void scalar_add_kernel(float *x, float a) { *x += a; }

__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v, long long n,
                                                   float lr, float b1, float b2, float eps, float wd,
                                                   const float *__restrict__ step, const float *__restrict__ hyper)
{
    if (hyper != nullptr) { lr = hyper[0]; wd = hyper[1]; }
    const float t = *step;
    const float bc1 = 1.0f - powf(b1, t);
    const float bc2_sqrt = sqrtf(1.0f - powf(b2, t));
    const float step_size = lr / bc1;
    const long long n4 = n / 4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4 *>(p)[i];
        const float4 gg = reinterpret_cast<const float4 *>(g)[i];
        float4 mm = reinterpret_cast<float4 *>(m)[i];
        float4 vv = reinterpret_cast<float4 *>(v)[i];
        float pa[4] = {pp.x, pp.y, pp.z, pp.w}, ga[4] = {gg.x, gg.y, gg.z, gg.w};
        float ma[4] = {mm.x, mm.y, mm.z, mm.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gr = ga[j] + wd * pa[j];
            ma[j] = ma[j] + (1.0f - b1) * (gr - ma[j]);
            va[j] = b2 * va[j] + (1.0f - b2) * gr * gr;
            const float denom = sqrtf(va[j]) / bc2_sqrt + eps;
            pa[j] = pa[j] - step_size * (ma[j] / denom);
        }
        reinterpret_cast<float4 *>(p)[i] = make_float4(pa[0], pa[1], pa[2], pa[3]);
        reinterpret_cast<float4 *>(m)[i] = make_float4(ma[0], ma[1], ma[2], ma[3]);
        reinterpret_cast<float4 *>(v)[i] = make_float4(va[0], va[1], va[2], va[3]);
    }
    for (long long i = n4 * 4 + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const float gr = g[i] + wd * p[i];
        const float mi = m[i] + (1.0f - b1) * (gr - m[i]);
        const float vi = b2 * v[i] + (1.0f - b2) * gr * gr;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    }
}

}  // namespace

extern "C" int mpa_scalar_add_f32(float *x, float a, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!x) return MPA_EINVAL;
    hipLaunchKernelGGL(scalar_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, x, a);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}

extern "C" int mpa_adam_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n,
                                 float lr, float beta1, float beta2, float eps, float weight_decay,
                                 const float *step, const float *hyper, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step || n <= 0) return MPA_EINVAL;
    if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0)
        return MPA_EUNSUPPORTED;
    long long g = (n / 4 + 255) / 256;
    g = g > 2048 ? 2048 : (g < 1 ? 1 : g);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, hyper);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}
