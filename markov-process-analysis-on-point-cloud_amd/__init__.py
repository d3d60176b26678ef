"""mpa_amd -- MI355X (gfx950) implementation of the Markov set-abstraction hot path of
ssr0512/Markov-Process-Analysis-on-Point-Cloud, behind the reference's own modules/ API.

Layout (mirrors the reference tree for the path only):
    csrc/            hand-written HIP kernels + the C ABI (include/mpa_hip.h) -> libmpa_hip.so
    _lib.py          ctypes binding of libmpa_hip.so (fails loudly if the library is absent)
    ops.py           device ops + autograd Functions over the C ABI
    modules/         pointnet2_utils.py / repsurface_utils.py : the reference's operator and
                     nn.Module names, signatures and state-dict keys
    models/repsurf/  the classification / part-seg wiring
    distributed.py   one-process-per-GPU data-parallel helpers (RCCL all-reduce of gradients)

The directory name carries hyphens (the project name); import it as `mpa_amd` through the
loader module at the repository root (mpa_amd.py).
"""
from . import _lib  # noqa: F401  (binds the shared library; raises if it is missing)
from . import ops  # noqa: F401

__version__ = "0.1.0"
