"""The C-ABI library loads on a CPU-only box and exports every symbol include/ibgs_rast.h declares.
No compute calls here (no GPU): only argument validation paths that return before any HIP call."""
import ctypes
import os
import re

import pytest

from ibgs_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "ibgs_rast.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(ibgs_[a-z_0-9]+)\s*\(", text)
    return sorted(set(n for n in names if not n.endswith("_fn")))


def test_header_symbols_exported(built_lib):
    names = declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(built_lib, n), "libibgs_rast.so does not export %s" % n
    assert sorted(_lib.EXPORTS) == names, "ibgs_amd/_lib.py EXPORTS out of sync with the header"


def test_struct_sizes_match(built_lib):
    assert built_lib.ibgs_sizeof_forward_args() == ctypes.sizeof(_lib.ForwardArgs)
    assert built_lib.ibgs_sizeof_backward_args() == ctypes.sizeof(_lib.BackwardArgs)


def test_arena_sizes(built_lib):
    g1, g2 = built_lib.ibgs_required_geom(1000), built_lib.ibgs_required_geom(2000)
    assert 0 < g1 < g2
    assert built_lib.ibgs_required_geom(0) > 0
    assert built_lib.ibgs_required_img(1920, 1080) > 1920 * 1080 * 8
    assert built_lib.ibgs_required_binning(10**6, 1920, 1080) >= 16 * 10**6
    assert built_lib.ibgs_required_tex(4, 1920, 1080) >= 4 * 1920 * 1080 * 16
    for nm in (b"rec", b"depths", b"cov3D", b"tiles", b"clamped", b"order", b"offsets"):
        assert built_lib.ibgs_geom_offset(1000, nm) >= 0
    assert built_lib.ibgs_geom_offset(1000, b"nope") == -1
    assert built_lib.ibgs_img_offset(64, 64, b"final_T") > 0


def test_validation_before_any_gpu_work(built_lib):
    a = _lib.ForwardArgs()
    a.P, a.W, a.H = 0, 64, 64
    assert built_lib.ibgs_forward(ctypes.byref(a)) == 0            # P == 0 short-circuits (rasterize_points.cu:101)
    a.P = -1
    assert built_lib.ibgs_forward(ctypes.byref(a)) == -1
    a.P = 10                                                        # required pointers missing
    assert built_lib.ibgs_forward(ctypes.byref(a)) == -1
    assert b"pointer" in built_lib.ibgs_last_error()
    b = _lib.BackwardArgs()
    b.P = 0
    assert built_lib.ibgs_backward(ctypes.byref(b)) == 0
    assert built_lib.ibgs_mark_visible(None, 0, None, None, None, None) == 0
    assert built_lib.ibgs_mark_visible(None, 5, None, None, None, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.RasterizerLibraryError):
        _lib.load()
