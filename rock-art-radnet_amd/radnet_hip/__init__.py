"""radnet_hip -- host-side binding + scheduler for libradnet_hip.so (gfx950 kernels, C ABI)."""
from .lib import ConvDesc, Context, RadnetError, declared_symbols, load_library  # noqa: F401
