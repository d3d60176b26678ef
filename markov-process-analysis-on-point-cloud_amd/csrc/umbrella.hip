// Umbrella surface features for gfx950 -- the RepSurf front-end of the reference
// (modules/repsurface_utils.py:106-126 group_by_umbrella, :321-376 UmbrellaSurfaceConstructor;
//  modules/recons_utils.py:27-57 cal_normal, :82-90 cal_center, :108-124 cal_const, :152-176
//  check_nan_umb; modules/polar_utils.py:10-31 xyz2sphere).
//
// The reference gathers [B,N,K,3] neighbours, converts them to spherical coordinates, argsorts,
// re-gathers, rolls, concatenates into [B,N,G,3,3] triangles and then runs six more elementwise
// passes over them.  Here one lane owns one point: its K-1 neighbour offsets, the azimuth sort
// (a stable insertion sort of <= 16 keys in registers), the G triangles' normals / centres /
// polar coordinates / plane constants and the NaN replacement all stay in registers, and the
// [B,N,G,10] feature block the 1x1-convolution MLP consumes is written once.  Coordinates carry no
// gradient, so there is no backward.
#include "mpa_common.h"

namespace {

constexpr int GMAX_ALL = 16;      // K - 1 <= 16; the usual K = 9 runs the 8-wide instantiation (no spills)

__device__ __forceinline__ void sphere(float x, float y, float z, float &rho, float &theta, float &phi)
{
    // rho, theta / pi (0 where rho == 0), phi / (2 pi) + 0.5  -- polar_utils.py:19-29
    rho = sqrtf((x * x + y * y) + z * z);
    theta = rho == 0.f ? 0.f : acosf(z / rho) / 3.14159265358979323846f;
    phi = atan2f(y, x) / 6.28318530717958647692f + 0.5f;
}

template <int GMAX>
__global__ __launch_bounds__(256) void umbrella_features_kernel(const float *__restrict__ xyz,
                                                                const int64_t *__restrict__ knn, int N, int K,
                                                                const float *__restrict__ cloud_sign,
                                                                int return_dist, long long npoints,
                                                                float *__restrict__ out)
{
    const int G = K - 1;
    const int CH = return_dist ? 10 : 9;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npoints;
         p += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(p / N);
        const float *cloud = xyz + (size_t)b * N * 3;
        const float cx = xyz[p * 3], cy = xyz[p * 3 + 1], cz = xyz[p * 3 + 2];
        float rx[GMAX], ry[GMAX], rz[GMAX], key[GMAX];
        // neighbours 1..K-1 (the nearest one -- the point itself -- is dropped), relative to the point
#pragma unroll
        for (int j = 0; j < GMAX; ++j)
            if (j < G) {
                const float *q = cloud + mpa_clamp_idx(knn[p * K + 1 + j], N) * 3;
                rx[j] = q[0] - cx; ry[j] = q[1] - cy; rz[j] = q[2] - cz;
                key[j] = atan2f(ry[j], rx[j]) / 6.28318530717958647692f + 0.5f;
            }
        // stable insertion sort by azimuth (ties keep the distance order)
#pragma unroll
        for (int i = 1; i < GMAX; ++i)
            if (i < G) {
#pragma unroll
                for (int j = GMAX - 1; j > 0; --j)
                    if (j <= i && key[j] < key[j - 1]) {
                        float t;
                        t = key[j]; key[j] = key[j - 1]; key[j - 1] = t;
                        t = rx[j]; rx[j] = rx[j - 1]; rx[j - 1] = t;
                        t = ry[j]; ry[j] = ry[j - 1]; ry[j - 1] = t;
                        t = rz[j]; rz[j] = rz[j - 1]; rz[j - 1] = t;
                    }
            }
        float nx[GMAX], ny[GMAX], nz[GMAX], mx[GMAX], my[GMAX], mz[GMAX], ps[GMAX];
        int first_ok = -1;
        float flip = 1.f;
#pragma unroll
        for (int i = 0; i < GMAX; ++i)
            if (i < G) {
                // triangle (centre, p_i, p_i+1): normal = p_i x p_i+1 (unit), centre of gravity
                float ax = rx[i], ay = ry[i], az = rz[i];
                float bx = rx[0], by = ry[0], bz = rz[0];
#pragma unroll
                for (int j = 1; j < GMAX; ++j)
                    if (j == i + 1 && j < G) { bx = rx[j]; by = ry[j]; bz = rz[j]; }
                // torch.cross on the host evaluates a1*b2 - a2*b1 as fma(a1, b2, -fl(a2*b1)): for the two
                // identical offsets of a duplicated point the result is the product's rounding error,
                // not 0, and the reference's normal there is that noise, normalised (NOT the NaN path).
                // Same form here, so such triangles agree with the reference instead of being replaced.
                const float vx = fmaf(ay, bz, -(az * by)), vy = fmaf(az, bx, -(ax * bz)), vz = fmaf(ax, by, -(ay * bx));
                const float nn = sqrtf((vx * vx + vy * vy) + vz * vz);
                nx[i] = vx / nn; ny[i] = vy / nn; nz[i] = vz / nn;
                mx[i] = (ax + bx) / 3.0f; my[i] = (ay + by) / 3.0f; mz[i] = (az + bz) / 3.0f;
                if (i == 0) flip = nx[0] > 0.f ? 1.f : -1.f;      // keep the FIRST triangle's x positive
            }
        const float sgn = flip * (cloud_sign ? cloud_sign[b] : 1.f);
#pragma unroll
        for (int i = 0; i < GMAX; ++i)
            if (i < G) {
                nx[i] *= sgn; ny[i] *= sgn; nz[i] *= sgn;
                ps[i] = ((nx[i] * mx[i] + ny[i] * my[i]) + nz[i] * mz[i]) / 1.7320508075688772f;
                const bool bad = (nx[i] != nx[i]) || (ny[i] != ny[i]) || (nz[i] != nz[i]);
                if (!bad && first_ok < 0) first_ok = i;
            }
        if (first_ok < 0) first_ok = 0;
        float fnx = 0.f, fny = 0.f, fnz = 0.f, fmx = 0.f, fmy = 0.f, fmz = 0.f, fps = 0.f;
#pragma unroll
        for (int i = 0; i < GMAX; ++i)
            if (i < G && i == first_ok) { fnx = nx[i]; fny = ny[i]; fnz = nz[i]; fmx = mx[i]; fmy = my[i]; fmz = mz[i]; fps = ps[i]; }
        float *o = out + (size_t)p * G * CH;
#pragma unroll
        for (int i = 0; i < GMAX; ++i)
            if (i < G) {
                float rho, theta, phi;
                sphere(mx[i], my[i], mz[i], rho, theta, phi);          // polar of the ORIGINAL centre (not replaced)
                const bool bad = (nx[i] != nx[i]) || (ny[i] != ny[i]) || (nz[i] != nz[i]);
                float *q = o + i * CH;
                q[0] = bad ? fmx : mx[i]; q[1] = bad ? fmy : my[i]; q[2] = bad ? fmz : mz[i];
                q[3] = rho; q[4] = theta; q[5] = phi;
                q[6] = bad ? fnx : nx[i]; q[7] = bad ? fny : ny[i]; q[8] = bad ? fnz : nz[i];
                if (return_dist) q[9] = bad ? fps : ps[i];
            }
    }
}

}  // namespace

extern "C" int mpa_umbrella_features_f32(const float *xyz, const int64_t *knn_idx, int B, int N, int K,
                                         const float *cloud_sign, int return_dist, float *out, void *stream)
{
    MPA_CLEAR_ERROR();
    if (!xyz || !knn_idx || !out || B <= 0 || N <= 0 || K < 3) return MPA_EINVAL;
    if (K - 1 > GMAX_ALL) return MPA_EUNSUPPORTED;
    const long long np = (long long)B * N;
    long long g = (np + 255) / 256;
    g = g > 16384 ? 16384 : g;
    if (K - 1 <= 8)
        hipLaunchKernelGGL(umbrella_features_kernel<8>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, xyz, knn_idx,
                           N, K, cloud_sign, return_dist, np, out);
    else
        hipLaunchKernelGGL(umbrella_features_kernel<GMAX_ALL>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, xyz,
                           knn_idx, N, K, cloud_sign, return_dist, np, out);
    MPA_LAUNCH_CHECK();
    return MPA_OK;
}
