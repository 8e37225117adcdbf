"""CPU-side checks of the C-ABI: the library builds for gfx950, loads, and exports every
symbol include/fdet.h declares (no compute calls without a GPU)."""
import ctypes
import os

import pytest


def test_library_exports_every_declared_symbol():
    import fdet_amd
    from fdet_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        _native.build()
    L = _native.lib()
    declared = _native.header_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/fdet.h but not exported"
    assert sorted(_native.SIGNATURES) == declared, "ctypes table and header disagree"
    assert L.fdet_version() == 100


def test_argument_validation_needs_no_gpu():
    """Bad arguments are rejected on the host before anything is enqueued."""
    import fdet_amd
    from fdet_amd import _native
    L = _native.lib()
    rc = L.fdet_nms(None, None, None, 1, 5000, 0.5, None, None, None)
    assert rc == -1
    assert b"nms" in L.fdet_last_error()
    assert L.fdet_stem_ws_bytes(1, 3, 64, 480, 480, 7, 3, 1) == 0          # unsupported stem geometry
    assert L.fdet_conv3x3_wgrad_ws_bytes(256, 64, 64, 60, 60) > 0


def test_product_path_refuses_cpu_tensors():
    import torch
    import fdet_amd
    from fdet_amd import hotpath, FdetError
    with pytest.raises(FdetError):
        hotpath.u8_to_f32_norm(torch.zeros(16, dtype=torch.uint8))
