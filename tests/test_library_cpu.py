"""CPU-only checks of the product library and host logic (no GPU needed):
the C-ABI library loads, exports every symbol include/vs_stab.h declares, its
PODs match the ctypes mirrors, and - with no GPU - compute entry points fail
loudly instead of falling back to anything."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from vsamd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "vs_stab.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vs_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(vs):
    names = declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(vs.lib, n)]
    assert not missing, missing


def test_abi_version_and_build_info(vs):
    assert vs.lib.vs_abi_version() == 2
    info = vs.lib.vs_build_info().decode()
    assert "gfx950" in info


def test_params_default_matches_reference_header(vs, oracle):
    """vs_params_default == Stabilizer.h:76-175 defaults (as restated by the oracle)."""
    a, b = vs.params(), oracle.params()
    assert a.struct_size == C.sizeof(capi.VsParams)
    for name, _ in capi.VsParams._fields_:
        if name == "reserved":
            continue
        assert getattr(a, name) == getattr(b, name), name
    assert (a.smoothing_radius, a.max_corners, a.quality_level, a.min_distance, a.block_size) == (30, 200, 0.01, 30.0, 3)
    assert (a.lk_win_size, a.lk_max_level, a.lk_max_iters, a.lk_epsilon) == (15, 2, 20, 0.03)
    assert (a.ransac_max_iters, a.ransac_threshold) == (500, 5.0)


def test_status_strings(vs):
    assert vs.lib.vs_status_string(0) == b"ok"
    assert b"device" in vs.lib.vs_status_string(2)


def test_bad_params_rejected_before_touching_the_gpu(vs):
    p = vs.params()
    p.struct_size = 4
    h = C.c_void_p()
    assert vs.lib.vs_stab_create(C.byref(p), 0, C.byref(h)) == 1       # VS_ERR_INVALID_ARG
    p = vs.params(enable_virtual_canvas=1, canvas_blend_weight=1.5)
    assert vs.lib.vs_stab_create(C.byref(p), 0, C.byref(h)) == 4       # VS_ERR_UNSUPPORTED (uchar cast out of range in the reference)
    p = vs.params(enable_virtual_canvas=1, temporal_buffer_size=-1)
    assert vs.lib.vs_stab_create(C.byref(p), 0, C.byref(h)) == 1
    p = vs.params(enable_virtual_canvas=1, temporal_buffer_size=-1, crop_n_zoom=1)   # never reached behind crop-and-zoom
    assert vs.lib.vs_stab_create(C.byref(p), 0, C.byref(h)) != 1
    p = vs.params(smoothing_method=capi.SMOOTH_GAUSSIAN, gaussian_sigma=10.6)     # 65 taps: more than the device kernel holds
    assert vs.lib.vs_stab_create(C.byref(p), 0, C.byref(h)) == 4
    p = vs.params(smoothing_method=capi.SMOOTH_GAUSSIAN, gaussian_sigma=0.0)
    assert vs.lib.vs_stab_create(C.byref(p), 0, C.byref(h)) == 1
    p = vs.params(smoothing_method=capi.SMOOTH_GAUSSIAN, gaussian_sigma=10.4)     # 63 taps
    assert vs.lib.vs_stab_create(C.byref(p), 0, C.byref(h)) not in (1, 4)
    p = vs.params(gaussian_sigma=50.0)                                            # not read by box smoothing
    assert vs.lib.vs_stab_create(C.byref(p), 0, C.byref(h)) not in (1, 4)
    p = vs.params(border_type=capi.BORDER_FADE + 1, border_size=8)
    assert vs.lib.vs_stab_create(C.byref(p), 0, C.byref(h)) == 1


def test_no_gpu_means_loud_failure_not_fallback(vs):
    if vs.lib.vs_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    p = vs.params()
    assert vs.lib.vs_stab_create(C.byref(p), 0, C.byref(h)) == 2       # VS_ERR_NO_DEVICE
    assert b"no CPU fallback" in vs.lib.vs_last_error()
    ptr = C.c_void_p()
    assert vs.lib.vs_dev_malloc(C.byref(ptr), 1024) == 2
    with pytest.raises(capi.VsError):
        vs.warp_affine(np.zeros((8, 8, 3), np.uint8), [1, 0, 0, 0, 1, 0])


def test_product_does_not_link_or_load_the_oracle(vs):
    """The oracle is test infrastructure: the shipped library must not depend on it."""
    import subprocess
    out = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "vso_oracle" not in out
    syms = subprocess.run(["nm", "-D", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "vso_" not in syms
    for root, _, files in os.walk(os.path.join(ROOT, "video-stab_amd")):
        for f in files:
            if f.endswith((".hip", ".cpp", ".h", ".py")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "oracle_lib" not in txt and "libvso_oracle" not in txt and "vso.h" not in txt, f
