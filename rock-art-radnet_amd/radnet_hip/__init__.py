"""radnet_hip -- host-side binding + scheduler for libradnet_hip.so (gfx950 kernels, C ABI)."""
from .lib import ConvDesc, Context, RadnetError, declared_symbols, load_library  # noqa: F401


def make_engine(C_cfg, **kw):
    """Engine for Config.network ('resnet50' | 'vgg16')."""
    from .engine_vgg import make_engine as _mk
    return _mk(C_cfg, **kw)
