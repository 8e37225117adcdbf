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


def test_pre_split_layout_plans_need_no_gpu():
    """Host-side geometry of the pre-split (PS) activation layout (csrc/fdet_ps.h): plain maps up to 63 columns, column strips
    beyond (round 4: config 3's 320 / 160 / 80-column levels), nothing for odd wide maps."""
    import fdet_amd
    from fdet_amd import _native
    L = _native.lib()
    assert [L.fdet_ps_strips(w) for w in (15, 60, 63, 64, 80, 160, 320, 65)] == [1, 1, 1, 2, 2, 3, 6, 0]
    # 60x60, 64 channels: units of 16 bytes = (N + 2 guard images) x 2 planes x 8 groups x 62 rows x 64 slots
    assert L.fdet_ps_bytes(256, 64, 60, 60) == (256 + 2) * 2 * 8 * 62 * 64 * 16
    assert L.fdet_ps_image0_offset(256, 64, 60, 60) == 2 * 8 * 62 * 64 * 16
    # 320 columns = 6 strips of 54 (the last holds 50): every strip is an image of its own
    assert L.fdet_ps_bytes(32, 64, 320, 320) == (32 * 6 + 2) * 2 * 8 * 322 * 64 * 16
    assert L.fdet_ps_bytes(2, 64, 60, 65) == 0 and L.fdet_ps_bytes(2, 60, 60, 60) == 0     # odd wide map; channels % 8
    assert L.fdet_conv3x3_wgrad_ps_ws_bytes(1, 32, 64, 320, 320) > 0
    assert L.fdet_conv3x3_wgrad_ps_ws_bytes(1, 32, 32, 320, 320) == 0                       # 64 channels only


def test_product_path_refuses_cpu_tensors():
    import torch
    import fdet_amd
    from fdet_amd import hotpath, FdetError
    with pytest.raises(FdetError):
        hotpath.u8_to_f32_norm(torch.zeros(16, dtype=torch.uint8))
