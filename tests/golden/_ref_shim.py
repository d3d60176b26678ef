"""Import shim for the upstream reference (runs ONLY in the build container).

The reference tree as shipped is not importable: its modules
import `models.polar_utils` / `models.recons_utils` / `models.pointnet2_utils`, which
live under `modules/`, and two helpers (`query_knn_point`, `sample`) are defined
nowhere.  This shim only aliases module names in `sys.modules` and supplies the two
undefined helpers from the reference's own `knn_point`; it does not alter any reference
arithmetic.  Nothing here travels to the GPU box: the golden vectors it produces do.
"""
import os
import sys
import types

REF_ROOT = "/root/reference/Markov_Process_Analysis_on_Point_Cloud"


def load_reference():
    """Returns (p2, rs, cls_model_mod, seg_model_mod): the reference's
    modules/pointnet2_utils, modules/repsurface_utils, models/repsurf/repsurf_ssg_umb,
    models/repsurf/pointnet2_part_seg_msg."""
    import torch

    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present (expected only in the build container)")
    sys.dont_write_bytecode = True
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

    import importlib
    import modules  # the reference's package

    polar = importlib.import_module("modules.polar_utils")
    sys.modules["models.polar_utils"] = polar

    # break the circular import pointnet2_utils -> recons_utils -> pointnet2_utils
    stub = types.ModuleType("modules.pointnet2_utils")
    stub.query_knn_point = None
    stub.index_points = None
    sys.modules["modules.pointnet2_utils"] = stub
    recons = importlib.import_module("modules.recons_utils")
    sys.modules["models.recons_utils"] = recons
    del sys.modules["modules.pointnet2_utils"]
    if hasattr(modules, "pointnet2_utils"):
        delattr(modules, "pointnet2_utils")

    p2 = importlib.import_module("modules.pointnet2_utils")
    p2.query_knn_point = lambda k, xyz, new_xyz, cuda=False: p2.knn_point(k, xyz, new_xyz)[1]
    _qb = p2.query_ball_point

    def _query_ball_point(radius, nsample, xyz, new_xyz, cuda=False):
        return _qb(radius, nsample, xyz, new_xyz)

    p2.query_ball_point = _query_ball_point
    recons.query_knn_point = p2.query_knn_point
    recons.index_points = p2.index_points
    sys.modules["models.pointnet2_utils"] = p2
    # upsample() hard-codes torch.cuda.FloatTensor (pointnet2_utils.py:36); CPU stand-in
    torch.cuda.FloatTensor = torch.FloatTensor

    rs = importlib.import_module("modules.repsurface_utils")
    cls_mod = importlib.import_module("models.repsurf.repsurf_ssg_umb")
    seg_mod = importlib.import_module("models.repsurf.pointnet2_part_seg_msg")
    return p2, rs, cls_mod, seg_mod
