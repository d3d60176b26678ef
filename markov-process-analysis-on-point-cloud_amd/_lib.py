"""ctypes binding of libmpa_hip.so (C ABI declared in include/mpa_hip.h).

The product path has no CPU fallback: if the shared library is missing this module raises
at import time, and every op raises on non-CUDA tensors.
"""
import ctypes
import os

# torch must be loaded BEFORE libmpa_hip.so: the wheel bundles its own libamdhip64.so.7 and the
# library's DT_NEEDED entry has the same soname, so with torch first the dynamic loader binds
# both to ONE HIP runtime (shared device context, streams and allocations).  Loaded the other
# way round the process ends up with two runtimes and launches fail with hipErrorNoDevice.
import torch  # noqa: F401  (ordering, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmpa_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libmpa_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C markov-process-analysis-on-point-cloud_amd/csrc`). There is no CPU fallback." % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH)

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float

# name -> argtypes (restype is int unless noted); order must match include/mpa_hip.h
SIGNATURES = {
    "mpa_fps_f32": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "mpa_fps_generic_f32": [_vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "mpa_square_distance_f32": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "mpa_knn_f32": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "mpa_row_norms_f32": [_vp, _i, _i, _i, _vp, _vp],
    "mpa_fps_knn_feat_f32": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i,
                             _vp, _vp, _vp],
    "mpa_knn_norms_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "mpa_ball_query_f32": [_vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp],
    "mpa_gather_fwd_f32": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "mpa_gather_bwd_f32": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "mpa_fps_knn_xyz_f32": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "mpa_umbrella_features_f32": [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp],
    "mpa_diffattn_fwd_f32": [_vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "mpa_diffattn_bwd_f32": [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp,
                             ctypes.c_size_t, _vp],
    "mpa_diffattn_bwd_workspace_bytes": [_i, _i, _i, _i, _i],
    "mpa_diffattn_xyz_fwd_f32": [_vp] * 9 + [_i] * 5 + [_vp, _vp, _vp],
    "mpa_diffattn_xyz_bwd_f32": [_vp] * 11 + [_i] * 5 + [_vp] * 6 + [_vp],
    "mpa_gemm_f32": [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, ctypes.c_size_t, _vp],
    "mpa_bn_stats_act_fwd_f32": [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _f, _f, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp],
    "mpa_tile_stats_f32": [_vp, _i, _i, _vp, _vp],
    "mpa_bn_finalize_f32": [_vp, _i, _i, _vp, _vp, _i, _f, _f, _vp, _vp, _i, _vp, _vp],
    "mpa_col_stats_f32": [_vp, _i, _i, _vp, _vp, _vp],
    "mpa_bn_act_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _f, _i, _i, _vp, _vp],
    "mpa_col_sum_f32": [_vp, _i, _i, _i, _vp, _vp],
    "mpa_group_col_sum_f32": [_vp, _i, _i, _i, _i, _vp, _vp],
    "mpa_bn_act_bwd_reduce_f32": [_vp] * 6 + [_f, _i, _i, _i, _vp, _i, _vp],
    "mpa_bn_act_bwd_apply_f32": [_vp] * 7 + [_i, _f, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "mpa_scalar_add_f32": [_vp, _f, _vp],
    "mpa_adam_step_f32": [_vp, _vp, _vp, _vp, ctypes.c_longlong, _f, _f, _f, _f, _f, _vp, _vp, _vp],
    "mpa_upsample_workspace_bytes": [_i, _i, _i, _i],
    "mpa_upsample_mean_fwd_f32": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, ctypes.c_size_t, _vp],
    "mpa_upsample_mean_bwd_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "mpa_three_interp_fwd_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "mpa_three_interp_bwd_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
}

class GemmTnProblem(ctypes.Structure):
    """struct MpaGemmTnProblem of include/mpa_hip.h"""
    _fields_ = [("A", _vp), ("B", _vp), ("out", _vp), ("a_col_sum", _vp),
                ("lda", _i), ("ldb", _i), ("M", _i), ("N", _i), ("K", _i)]


SIGNATURES["mpa_gemm_tn_grouped_f32"] = [ctypes.POINTER(GemmTnProblem), _i, _vp, ctypes.c_size_t, _vp]


class GeoRider(ctypes.Structure):
    """struct MpaGeoRider of include/mpa_hip.h"""
    _fields_ = [("src", _vp), ("B", _i), ("N", _i), ("nlev", _i), ("S", _i * 4), ("start", _vp * 4), ("idx", _vp * 4),
                ("xyz", _vp * 4), ("base", _vp), ("query", _vp), ("sN", _i), ("sS", _i), ("sK", _i), ("dist", _vp),
                ("kidx", _vp), ("queue", _vp)]


SIGNATURES["mpa_gemm_tn_grouped_rider_f32"] = [ctypes.POINTER(GemmTnProblem), _i, _vp, ctypes.c_size_t,
                                               ctypes.POINTER(GeoRider), _i, _vp]
SIGNATURES["mpa_geo_rider_f32"] = [ctypes.POINTER(GeoRider), _vp]


class GemmProblem(ctypes.Structure):
    """struct MpaGemmProblem of include/mpa_hip.h"""
    _fields_ = [("A", _vp), ("B", _vp), ("bias", _vp), ("C", _vp), ("tile_stats", _vp),
                ("lda", _i), ("ldb", _i), ("ldc", _i), ("M", _i), ("N", _i), ("K", _i), ("stats_replicas", _i)]


SIGNATURES["mpa_gemm_grouped_f32"] = [ctypes.POINTER(GemmProblem), _i, _i, _vp]


class BnUnit(ctypes.Structure):
    """struct MpaBnUnit of include/mpa_hip.h"""
    _fields_ = [("x", _vp), ("stats", _vp), ("running_mean", _vp), ("running_var", _vp), ("num_batches_tracked", _vp),
                ("gamma", _vp), ("beta", _vp), ("residual", _vp), ("y", _vp), ("save", _vp), ("grad_y", _vp),
                ("partial", _vp), ("grad_x", _vp), ("dgamma", _vp), ("dbeta", _vp),
                ("M", _i), ("C", _i), ("stats_replicas", _i), ("ldg", _i), ("training", _i), ("replicas", _i),
                ("momentum", ctypes.c_float), ("eps", ctypes.c_float), ("slope", ctypes.c_float), ("ldy", _i)]


for _sfx in ("f32", "bf16"):
    SIGNATURES["mpa_max_points_fwd_" + _sfx] = [_vp, _i, _i, _i, _vp, _vp, _vp]
    SIGNATURES["mpa_max_points_bwd_" + _sfx] = [_vp, _vp, _i, _i, _i, _vp, _vp]
    SIGNATURES["mpa_bn_group_fwd_" + _sfx] = [ctypes.POINTER(BnUnit), _i, _i, _vp]
    SIGNATURES["mpa_bn_group_bwd_reduce_" + _sfx] = [ctypes.POINTER(BnUnit), _i, _vp]
    SIGNATURES["mpa_bn_group_bwd_apply_" + _sfx] = [ctypes.POINTER(BnUnit), _i, _vp]


class GemmTnProblemBf16(ctypes.Structure):
    """struct MpaGemmTnProblemBf16 of include/mpa_hip.h"""
    _fields_ = [("A", _vp), ("B", _vp), ("out", _vp), ("a_col_sum", _vp),
                ("lda", _i), ("ldb", _i), ("M", _i), ("N", _i), ("K", _i)]


# ---- bf16 feature path (same argument order as the _f32 entry points unless noted)
SIGNATURES.update({
    "mpa_gemm_bf16": [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp],
    "mpa_bn_stats_act_fwd_bf16": SIGNATURES["mpa_bn_stats_act_fwd_f32"],
    "mpa_gemm_tn_grouped_bf16": [ctypes.POINTER(GemmTnProblemBf16), _i, _vp, ctypes.c_size_t, _vp],
    "mpa_gemm_grouped_bf16": [ctypes.POINTER(GemmProblem), _i, _i, _i, _vp],
    "mpa_gather_bwd_into_bf16": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "mpa_bn_act_fwd_bf16": SIGNATURES["mpa_bn_act_fwd_f32"],
    "mpa_bn_act_bwd_reduce_bf16": SIGNATURES["mpa_bn_act_bwd_reduce_f32"],
    "mpa_bn_act_bwd_apply_bf16": SIGNATURES["mpa_bn_act_bwd_apply_f32"],
    "mpa_gather_fwd_bf16": SIGNATURES["mpa_gather_fwd_f32"],
    "mpa_gather_bwd_bf16": SIGNATURES["mpa_gather_bwd_f32"],
    "mpa_diffattn_fwd_bf16": SIGNATURES["mpa_diffattn_fwd_f32"],
    "mpa_diffattn_bwd_bf16": SIGNATURES["mpa_diffattn_bwd_f32"],
    "mpa_diffattn_bwd_workspace_bytes_bf16": SIGNATURES["mpa_diffattn_bwd_workspace_bytes"],
    "mpa_diffattn_xyz_fwd_bf16": SIGNATURES["mpa_diffattn_xyz_fwd_f32"],
    "mpa_diffattn_xyz_bwd_bf16": SIGNATURES["mpa_diffattn_xyz_bwd_f32"],
    "mpa_upsample_mean_fwd_bf16": SIGNATURES["mpa_upsample_mean_fwd_f32"],
    "mpa_upsample_mean_bwd_bf16": SIGNATURES["mpa_upsample_mean_bwd_f32"],
    "mpa_group_col_sum_bf16": SIGNATURES["mpa_group_col_sum_f32"],
    "mpa_three_interp_fwd_bf16": SIGNATURES["mpa_three_interp_fwd_f32"],
    "mpa_three_interp_bwd_bf16": SIGNATURES["mpa_three_interp_bwd_f32"],
})

SIGNATURES["mpa_geo_level_f32"] = ([_vp, _i, _i, _i, _vp, _vp, _vp] + [_vp, _vp, _i, _i, _i, _vp, _vp] * 2 +
                                   [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp])
SIGNATURES["mpa_coarse_level_f32"] = [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i,
                                      _i, _vp, _vp, _vp]
SIGNATURES.update({
    "mpa_log_softmax_fwd_f32": [_vp, _i, _i, _vp, _vp],
    "mpa_log_softmax_bwd_f32": [_vp, _vp, _i, _i, _vp, _vp],
    "mpa_smooth_loss_workspace_floats": [_i],
    "mpa_smooth_loss_fwd_f32": [_vp, _vp, _i, _i, _f, _i, _vp, _vp, _vp, _vp],
    "mpa_smooth_loss_bwd_f32": [_vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp, _vp],
    "mpa_pool_max_mean_fwd_f32": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "mpa_pool_max_mean_bwd_f32": [_vp, _vp, _i, _i, _i, _vp, _vp],
    "mpa_add_n_f32": [_vp, _vp, _i, ctypes.c_longlong, _i, _vp, _vp],
    "mpa_add_n_bf16": [_vp, _vp, _i, ctypes.c_longlong, _i, _vp, _vp],
})

for _name, _args in SIGNATURES.items():
    _fn = getattr(lib, _name)
    _fn.argtypes = _args
    _fn.restype = ctypes.c_int
lib.mpa_diffattn_bwd_workspace_bytes.restype = ctypes.c_size_t
lib.mpa_diffattn_bwd_workspace_bytes_bf16.restype = ctypes.c_size_t
lib.mpa_upsample_workspace_bytes.restype = ctypes.c_size_t
lib.mpa_version.restype = ctypes.c_int
# the argument lists above are written against this ABI version of include/mpa_hip.h (MPA_ABI_VERSION): a stale
# library would take shifted arguments (a stream pointer read as a device array), so refuse it at load time
ABI_VERSION = 301
if lib.mpa_version() != ABI_VERSION:
    raise ImportError("libmpa_hip.so reports ABI version %d, this binding expects %d -- rebuild it "
                      "(make -C markov-process-analysis-on-point-cloud_amd/csrc)" % (lib.mpa_version(), ABI_VERSION))
lib.mpa_error_string.restype = ctypes.c_char_p
lib.mpa_error_string.argtypes = [ctypes.c_int]
lib.mpa_last_hip_error.restype = ctypes.c_int
lib.mpa_last_hip_error_string.restype = ctypes.c_char_p


class MpaError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        detail = ""
        if rc == -3:
            detail = " [hipError %d: %s]" % (lib.mpa_last_hip_error(), lib.mpa_last_hip_error_string().decode())
        raise MpaError("%s failed: %s (code %d)%s" % (what, lib.mpa_error_string(rc).decode(), rc, detail))
