// Version / error strings of libmpa_hip.so.
#include "mpa_common.h"

static thread_local int g_last_hip_error = 0;

void mpa_note_hip_error(int hip_error) { g_last_hip_error = hip_error; }

// MPA_ABI_VERSION of include/mpa_hip.h: bumped whenever an exported signature changes incompatibly (200: round 2
// added `stats_replicas` to mpa_gemm_f32 and `hyper` to mpa_adam_step_f32; 300: round 3's entry points)
extern "C" int mpa_version(void) { return MPA_ABI_VERSION; }

extern "C" int mpa_last_hip_error(void) { return g_last_hip_error; }

extern "C" const char *mpa_last_hip_error_string(void)
{
    return hipGetErrorString((hipError_t)g_last_hip_error);
}

extern "C" const char *mpa_error_string(int code)
{
    switch (code) {
    case MPA_OK: return "ok";
    case MPA_EINVAL: return "invalid argument (null pointer or non-positive size)";
    case MPA_EUNSUPPORTED: return "shape not supported by the gfx950 kernels";
    case MPA_EHIP: return "HIP launch failed";
    default: return "unknown mpa error";
    }
}
