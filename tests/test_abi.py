"""The C-ABI library loads without a GPU and exports every symbol include/crt.h declares."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "crt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(crt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import cudaraytracer_amd as crt
    from cudaraytracer_amd import api
    syms = declared_symbols()
    assert len(syms) >= 35
    lib = ctypes.CDLL(crt.lib_path)
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in crt.h but not exported: {missing}"
    assert sorted(api.ABI_SYMBOLS) == syms, "cudaraytracer_amd.api.ABI_SYMBOLS is out of date with include/crt.h"


def test_header_cites_the_reference_interfaces():
    text = open(os.path.join(ROOT, "include", "crt.h")).read()
    for cite in ("Ray.hpp:35", "Ray.cu:665", "bvh.cu:12-273", "bvh.cu:288-400", "bvh.cu:274-286", "objloader.cpp:20-112",
                 "SceneObjects.hpp:30-52", "SceneObjects.hpp:16-28", "main.cu:545-573"):
        assert cite in text, f"crt.h no longer cites {cite}"


def test_struct_layouts_match_the_header():
    from cudaraytracer_amd import api
    assert api.PRIM_DTYPE.itemsize == 164 and api.LIGHT_DTYPE.itemsize == 60
    assert ctypes.sizeof(api.Params) == 64 and ctypes.sizeof(api.Camera) == 24
    p = api.default_params()
    # defaults equal the reference's literals: 720x360 (main.cu:141-142), numBounces 2 (Ray.cu:27)
    assert (p.width, p.height, p.bounces, p.spp, p.tile_w, p.tile_h, p.rank, p.nranks) == (720, 360, 2, 1, 32, 32, 0, 1)
    cam, tgt = api.default_camera()
    assert list(cam) == [0, 50, 250] and list(tgt) == [0, 0, 0]


def test_error_codes_without_a_device_or_with_bad_arguments():
    from cudaraytracer_amd import api
    # no compute call is made here: argument validation only
    assert api.lib.crt_scene_set(None, None, 3) == -1
    assert api.lib.crt_bvh_build(None, 0) == -1
    assert api.lib.crt_load_obj(b"/nonexistent/file.obj", None, None, 0) == -5      # CRT_ERR_IO, no getchar() block
    bad = api.default_params(tile_w=12)
    with pytest.raises(api.CrtError):
        api.tiles_owned(bad)
    prims = np.zeros(3, api.PRIM_DTYPE)
    prims["type"] = 1
    r = api.lib.crt_host_bvh(prims.ctypes.data_as(ctypes.c_void_p), 3, 7, None, None, None, None, None, None, None, None, 0)
    assert r == -1
