// Version / error strings of libmpa_hip.so.
#include "mpa_common.h"

static thread_local int g_last_hip_error = 0;

void mpa_note_hip_error(int hip_error) { g_last_hip_error = hip_error; }

extern "C" int mpa_version(void) { return 100; }

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
