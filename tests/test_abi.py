"""The C-ABI library loads and exports every symbol include/radnet_hip.h declares (no compute calls: no GPU here)."""
import ctypes
import os

import pytest

from radnet_hip import lib as L


def test_library_exports_every_declared_symbol():
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = L.load_library()
    names = L.declared_symbols()
    assert len(names) >= 34
    for n in names:
        assert hasattr(lib, n), "header declares %s but the library does not export it" % n
    assert lib.radnet_version() >= 100


def test_conv_desc_layout_matches_header():
    # field order/types are mirrored by hand: catch drift by size (8-byte pointers, 4-byte ints, natural alignment)
    n_ptr = sum(1 for _, t in L.ConvDesc._fields_ if t is ctypes.c_void_p)
    n_i32 = sum(1 for _, t in L.ConvDesc._fields_ if t is ctypes.c_int32)
    assert (n_ptr, n_i32) == (13, 22)
    assert ctypes.sizeof(L.ConvDesc) >= n_ptr * 8 + n_i32 * 4


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(L.RadnetError):
        L.Context(0)
    h = ctypes.c_void_p()
    assert L.load_library().radnet_create(0, None, ctypes.byref(h)) != 0
