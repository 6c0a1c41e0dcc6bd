"""Several devices from ONE process through the C ABI (include/rt_amd.h rt_multi_*): the sharding of dist.py — interleaved row
bands, replicated scene, no data-path collective — for a host that is neither Python nor MPI.  The GPU box has one GPU, so
the device list repeats device 0: three bands rendered concurrently on three streams, three scene copies, three sets of
per-pixel generators — everything but the second physical device.  Results must equal the single-device entry points and
the oracle bit for bit."""
import ctypes as C

import numpy as np
import pytest

import homework_18_graphics_raytracer_amd as rt
from homework_18_graphics_raytracer_amd import _capi
import _oracle

pytestmark = pytest.mark.gpu


def _multi(desc, devices):
    lib = _capi.amd_lib()
    h = C.c_void_p()
    arr = (C.c_int * len(devices))(*devices)
    _capi.check(lib.rt_multi_create(C.byref(desc), arr, len(devices), C.byref(h)))
    return h


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
@pytest.mark.parametrize("w,h,depth", [(160, 99, 5), (64, 2, 3)])
def test_whitted_over_a_device_list_equals_the_oracle(devices, w, h, depth):
    lib = _capi.amd_lib()
    world, cam = rt.reference_world(), rt.reference_camera()
    desc = world.desc()
    m = _multi(desc, devices)
    try:
        for frame in (rt.Frame.full(w, h, depth), rt.Frame(w, h, depth, 3, 0, w - 5, h, 1)):
            img = np.empty((frame.rows, frame.cols, 3), dtype=np.float32)
            casts = C.c_ulonglong(0)
            _capi.check(lib.rt_multi_render_whitted_host(m, C.byref(cam), C.byref(frame), img.ctypes.data_as(C.c_void_p), C.byref(casts)))
            want, wcasts = _oracle.render_whitted(desc, cam, frame)
            assert np.array_equal(img.view(np.uint32), want.view(np.uint32)) and casts.value == wcasts
    finally:
        lib.rt_multi_destroy(m)


def test_stochastic_loop_over_a_device_list_is_the_reference_loop():
    """main()'s progressive loop (main.rs:1129-1173) with the image on the host and the epochs on three bands: one epoch per call,
    post_process in between — against the oracle's loop, and continuing streams across calls."""
    lib = _capi.amd_lib()
    world, cam = rt.reference_world(), rt.reference_camera()
    desc = world.desc()
    frame = rt.Frame.full(96, 70, 5)
    m = _multi(desc, [0, 0, 0])
    try:
        img = np.empty((70, 96, 3), dtype=np.float32)
        casts = C.c_ulonglong(0)
        _capi.check(lib.rt_multi_render_whitted_host(m, C.byref(cam), C.byref(frame), img.ctypes.data_as(C.c_void_p), C.byref(casts)))
        rt.post_process(img)
        want, _ = _oracle.render_whitted(desc, cam, frame)
        _oracle.post_process(want)
        st = _oracle.rng_init(frame)
        total = 0
        for epochs in (1, 1, 3):
            _capi.check(lib.rt_multi_render_distributed_host(m, C.byref(cam), C.byref(frame), 3.0, 0.04, epochs, img.ctypes.data_as(C.c_void_p), C.byref(casts)))
            s, v, c = _oracle.render_distributed(desc, cam, frame, st, epochs)
            for e in range(epochs):
                want += np.where(v[e][..., None] != 0, s[e], np.float32(0))
            assert casts.value == c
            total += c
            rt.post_process(img)
            _oracle.post_process(want)
            assert np.array_equal(img.view(np.uint32), want.view(np.uint32))
    finally:
        lib.rt_multi_destroy(m)


@pytest.mark.parametrize("stage", [False, True])
@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_device_resident_frame_over_a_device_list(devices, stage):
    """rt_multi_render_whitted / rt_multi_render_distributed: the frame is assembled in device memory (no host bounce), so that
    rt_post_process_device can follow on the same stream — main()'s whole loop without the image leaving the GPUs.  `stage`
    forces every band through the staging buffer + hipMemcpyPeerAsync route that a second physical device would take."""
    import torch

    rt.set_option("RT_AMD_MULTI_FORCE_STAGE", 1 if stage else None)
    lib = _capi.amd_lib()
    world, cam = rt.reference_world(), rt.reference_camera()
    desc = world.desc()
    frame = rt.Frame.full(96, 70, 5)
    m = _multi(desc, devices)
    try:
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            img = torch.zeros((70, 96, 3), dtype=torch.float32, device="cuda")
            cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
        stream.synchronize()
        sp = C.c_void_p(stream.cuda_stream)
        _capi.check(lib.rt_multi_render_whitted(m, C.byref(cam), C.byref(frame), C.c_void_p(img.data_ptr()), C.c_void_p(cnt.data_ptr()), sp))
        _capi.check(lib.rt_post_process_device(C.c_void_p(img.data_ptr()), 70 * 96, None, sp))
        want, wcasts = _oracle.render_whitted(desc, cam, frame)
        _oracle.post_process(want)
        st = _oracle.rng_init(frame)
        total = wcasts
        for epochs in (1, 2):
            _capi.check(lib.rt_multi_render_distributed(m, C.byref(cam), C.byref(frame), 3.0, 0.04, epochs, C.c_void_p(img.data_ptr()), C.c_void_p(cnt.data_ptr()), sp))
            _capi.check(lib.rt_post_process_device(C.c_void_p(img.data_ptr()), 70 * 96, None, sp))
            s, v, c = _oracle.render_distributed(desc, cam, frame, st, epochs)
            for e in range(epochs):
                want += np.where(v[e][..., None] != 0, s[e], np.float32(0))
            _oracle.post_process(want)
            total += c
        stream.synchronize()
        assert np.array_equal(img.cpu().numpy().view(np.uint32), want.view(np.uint32))
        assert int(cnt.item()) == total
        assert torch.cuda.current_device() == 0
    finally:
        lib.rt_multi_destroy(m)
        rt.set_option("RT_AMD_MULTI_FORCE_STAGE", None)


def test_argument_validation():
    lib = _capi.amd_lib()
    desc = rt.reference_world().desc()
    h = C.c_void_p()
    assert lib.rt_multi_create(C.byref(desc), (C.c_int * 1)(7), 1, C.byref(h)) == -1 and b"device" in lib.rt_last_error()
    assert lib.rt_multi_create(C.byref(desc), None, 0, C.byref(h)) == -1
    assert lib.rt_multi_render_whitted_host(None, None, None, None, None) == -1
    assert lib.rt_multi_render_whitted(None, None, None, None, None, None) == -1
    assert lib.rt_multi_render_distributed(None, None, None, 3.0, 0.04, 1, None, None, None) == -1
