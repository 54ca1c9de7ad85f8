"""radnet_hip -- host-side binding + scheduler for libradnet_hip.so (gfx950 kernels, C ABI)."""
import os as _os

# The engine's lanes (three compute streams + the upload stream) and RCCL's per-communicator streams exceed the 4 hardware
# queues HIP multiplexes streams over by default; busy streams sharing a queue serialise.  Effective only if
# this import precedes the first HIP call of the process (bench.py sets it before importing torch as well).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .lib import ConvDesc, Context, RadnetError, declared_symbols, load_library  # noqa: F401


def make_engine(C_cfg, **kw):
    """Engine for Config.network ('resnet50' | 'vgg16')."""
    from .engine_vgg import make_engine as _mk
    return _mk(C_cfg, **kw)
