"""CPU-side checks of the drop-in boundary: the C ABI library builds for gfx950, loads,
exports every symbol include/dcmt.h declares, and the argument validation that needs no GPU."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from depth_completion_mt_amd import _lib as L
from depth_completion_mt_amd import api


def test_library_builds_and_exports_every_declared_symbol():
    path = L.build()
    assert os.path.exists(path)
    hdr = open(os.path.join(ROOT, "include", "dcmt.h")).read()
    declared = set(re.findall(r"\b(dcmt_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    nm = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (dcmt_[a-z0-9_]+)", nm))
    assert declared <= exported, declared - exported
    lib = L.lib()
    for name in L.EXPORTS:
        assert getattr(lib, name) is not None


def test_code_object_targets_gfx950_only():
    out = subprocess.run(["strings", "-n", "6", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    archs = set(re.findall(r"gfx[0-9a-f]{3,4}", out))
    assert archs == {"gfx950"}, archs


def test_params_struct_layout_matches_header():
    assert ctypes.sizeof(L.Params) == 4 + 4 + 25 + 3 + 6 * 4 + 2 * 4
    p = api.make_params()
    assert p.max_depth == 100.0
    assert np.float32(p.valid_thresh) == np.float32(0.1)
    k = np.frombuffer(bytes(p.k0), dtype=np.uint8).reshape(5, 5)
    want = np.zeros((5, 5), np.uint8)
    want[1, 3] = want[4, 4] = 1
    assert np.array_equal(k, want)                      # as-compiled element (img_completion.cpp:71-77)
    assert p.blur == L.BLUR_GAUSSIAN and p.stop_after == L.STAGE_FINAL and p.max_fill_iters == 64
    assert p.flags == 0 and (p.norm_lo, p.norm_hi) == (0.0, 100.0)      # SL/main_sl.cpp:370's range; off unless the flag is set
    n = api.make_params(normalize=(0, 80))
    assert n.flags == L.FLAG_NORMALIZE and (n.norm_lo, n.norm_hi) == (0.0, 80.0)
    d = api.make_params(k0="diamond")
    assert sum(bytes(d.k0)) == 13
    assert api.make_params(blur_type="bilateral").blur == L.BLUR_BILATERAL
    assert api.make_params(blur_type="whatever").blur == L.BLUR_NONE     # reference: any other string = no blur


def test_status_strings_and_version():
    assert L.lib().dcmt_version() == 120
    for s in range(0, -7, -1):
        assert L.strerror(s) and L.strerror(s) != "unknown status"
    assert L.strerror(-99) == "unknown status"


def test_null_and_range_validation_without_gpu():
    lib = L.lib()
    h = ctypes.c_void_p()
    assert lib.dcmt_create(0, 0, 10, 1, ctypes.byref(h)) == L.E_INVALID
    assert lib.dcmt_create(0, 10, 10, 1, None) == L.E_INVALID
    assert lib.dcmt_create(0, 1 << 15, 1 << 15, 1, ctypes.byref(h)) == L.E_INVALID     # 2^30 pixels: beyond 32-bit byte offsets
    # the "store that writes nothing" offset (kDropOffset = 0x7ffffff0, dcmt_kernels_fused.h) must lie beyond every admitted frame:
    # 0x1ffffff1 pixels = 0x7fffffc4 bytes is refused (the limit is 0x1ffffff0 pixels)
    assert lib.dcmt_create(0, 1, 0x1ffffff1, 1, ctypes.byref(h)) == L.E_INVALID
    assert lib.dcmt_slic_num_centers(352, 1216, 18) == 67 * 19 and lib.dcmt_slic_num_centers(375, 1242, 68) == 17 * 5
    assert lib.dcmt_slic_labels_dev(None, None, 8, 8, 1, 6, 1, None, None, None) == L.E_INVALID
    assert lib.dcmt_project_points_dev(None, None, None, 0, 1, None, None, None, 8, 8, None) == L.E_INVALID
    assert lib.dcmt_stereo_refine_dev(None, None, None, None, None, 8, 8, 1, None, None) == L.E_INVALID
    sp = L.StereoParams()
    lib.dcmt_default_stereo_params(ctypes.byref(sp))
    assert (round(sp.baseline, 2), round(sp.focal, 3), sp.damp, sp.max_depth, sp.iterations) == (0.54, 959.791, 500.0, 100.0, 4)
    p = api.make_params()
    assert lib.dcmt_complete_f32_dev(None, None, None, 1, 1, 1, ctypes.byref(p), None) == L.E_INVALID
    assert lib.dcmt_last_fill_iters(None, None, 0) == L.E_INVALID
    lib.dcmt_destroy(None)      # no-op


@pytest.mark.skipif(L.lib().dcmt_device_count() > 0, reason="a GPU is present")
def test_product_path_fails_loudly_without_a_gpu():
    """No CPU fallback: without a gfx950 device the product API raises."""
    with pytest.raises(api.DcmtError) as e:
        api.img_completion(np.zeros((8, 8), np.float32))
    assert e.value.status == L.E_NO_DEVICE


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "depth_completion_mt_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle|#include\s+\"dcmt_oracle", txt, re.M), f
    for f in ("include/dcmt.h", "include/img_completion.h"):
        pth = os.path.join(ROOT, f)
        if os.path.exists(pth):
            assert "dcmt_oracle" not in open(pth).read()
