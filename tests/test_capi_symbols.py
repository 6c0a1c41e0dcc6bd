"""The C-ABI libraries load and export every symbol the headers declare; argument validation and the
no-device failure path return status codes (never crash, never fall back to a CPU renderer)."""
import ctypes as C
import re
from pathlib import Path

import pytest

import homework_18_graphics_raytracer_amd as rt
from homework_18_graphics_raytracer_amd import _capi

INCLUDE = Path(__file__).resolve().parent.parent / "include"


def _declared(header: str):
    text = (INCLUDE / header).read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_amd_library_exports_every_declared_symbol():
    lib = _capi.amd_lib()
    names = _declared("rt_amd.h")
    assert set(names) == set(_capi.AMD_SYMBOLS), set(names) ^ set(_capi.AMD_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.rt_abi_version() == 1


def test_host_library_exports_every_declared_symbol():
    lib = _capi.host_lib()
    names = _declared("rt_host.h")
    assert set(names) == set(_capi.HOST_SYMBOLS), set(names) ^ set(_capi.HOST_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n


def test_argument_validation_returns_status():
    lib = _capi.amd_lib()
    assert lib.rt_scene_create(None, None) == -1
    assert b"null" in lib.rt_last_error()
    assert lib.rt_render_whitted(None, None, None, None, None, None) == -1
    bad = rt.Frame(10, 10, 5, 0, 0, 11, 10, 1)  # x1 > width
    assert lib.rt_frame_pixels(C.byref(bad)) == 0
    good = rt.Frame.rows_of_rank(1920, 1080, 8, 3, 8)
    assert lib.rt_frame_rows(C.byref(good)) == 135 and lib.rt_frame_pixels(C.byref(good)) == 135 * 1920
    assert lib.rt_set_variant(99) == -1 and lib.rt_set_variant(0) == -1 and lib.rt_set_variant(6) == -1  # 2, 3, 18, 19 only
    # a tile of 2^32 pixels or more is refused, not wrapped (checked before any device work)
    huge = rt.Frame.full(65536, 65536, 5)
    h = C.c_void_p()
    assert lib.rt_rng_create(C.byref(huge), C.byref(h)) == -5 and b"2^32" in lib.rt_last_error()
    assert lib.rt_rng_create(None, None) == -1 and lib.rt_render_distributed(None, None, None, 3.0, 0.04, None, 1, None, None, None, None, None) == -1


def test_scene_validation():
    lib = _capi.amd_lib()
    world = rt.reference_world()
    d = world.desc()
    h = C.c_void_p()
    # object index out of range is rejected before any device work
    tri = (_capi.Triangle * 1)()
    tri[0].object_index = 99
    bad = _capi.SceneDesc(tri, 1, d.spheres, d.n_spheres, d.materials, d.n_materials, d.lights, d.n_lights)
    assert lib.rt_scene_create(C.byref(bad), C.byref(h)) == -1
    assert b"object_index" in lib.rt_last_error()


def test_no_device_fails_loudly_without_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    world = rt.reference_world()
    with pytest.raises(rt.RtError) as ei:
        rt.Scene(world)
    assert ei.value.code in (-2, -3)
