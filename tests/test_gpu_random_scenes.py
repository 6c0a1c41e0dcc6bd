"""GPU parity on scenes the reference does not have: random geometry/materials/lights, primitive counts that
exercise the intersection loop's edges (0, 1, odd, > 64 triangles; no spheres), deep recursion (the
RT_MAX_DEPTH template), degenerate inputs.  Bit-identical radiance and equal cast counts, as everywhere."""
import ctypes as C

import numpy as np
import pytest

import homework_18_graphics_raytracer_amd as rt
from homework_18_graphics_raytracer_amd import _capi
import _oracle
import _scenes

pytestmark = pytest.mark.gpu


def _check(world, cam, frame, variant=2):
    scene = rt.Scene(world)
    _capi.check(_capi.amd_lib().rt_set_variant(variant))
    try:
        got, casts = rt.render_whitted_numpy(scene, cam, frame)
    finally:
        _capi.check(_capi.amd_lib().rt_set_variant(_capi.DEFAULT_VARIANT))
    want, wcasts = _oracle.render_whitted(world.desc(), cam, frame)
    g, w = got.view(np.uint32), want.view(np.uint32)
    same = (g == w) | (np.isnan(got) & np.isnan(want))  # NaN payload/sign may differ between x86 and gfx950
    assert same.all(), f"{(~same).sum()} channels differ; first {np.argwhere(~same)[:3].tolist()}"
    assert casts == wcasts
    return got


@pytest.mark.parametrize("seed,nt,ns", [(1, 0, 3), (2, 1, 0), (3, 2, 1), (4, 7, 2), (5, 33, 4), (6, 64, 0), (7, 65, 5), (8, 131, 3), (9, 200, 9)])
def test_random_scene_whitted(seed, nt, ns):
    world = _scenes.random_world(seed, nt, ns)
    _check(world, _scenes.camera(seed), rt.Frame.full(96, 64, 5))


@pytest.mark.parametrize("variant", [3, 18, 19])
def test_random_scene_every_render_path(variant):
    world = _scenes.random_world(21, 40, 4)
    _check(world, _scenes.camera(21), rt.Frame.full(120, 90, 6), variant)


@pytest.mark.parametrize("depth", [9, 12, 20, 32])
def test_deep_recursion_uses_the_large_stack_template(depth):
    world = rt.reference_world()
    _check(world, rt.reference_camera(), rt.Frame.full(80, 60, depth))


def test_nan_face_normal_is_accepted_as_the_reference_accepts_it():
    """A degenerate (collinear) triangle has a NaN face normal: t = NaN passes `t <= 0`, NaN areas pass `< 0`, and NaN as the
    nearest distance lets every later triangle through (main.rs:205, 224, 229-233) — the scan's history matters."""
    rng = np.random.default_rng(9)
    w = rt.World()
    o = w.push_object(_scenes.material(rng, "plain"))
    o.push_flat_triangle([(0, 0, 0), (1, 0, 0), (2, 0, 0)], [(0, 0), (1, 0), (0, 1)])  # NaN face normal: accepted with t = NaN
    for k in range(9):
        o.push_flat_triangle([(k * 0.3 - 1, 0.2, -1), (k * 0.3 - 1, 0.2, 1), (k * 0.3 - 0.8, 0.9, 0)], [(0, 0), (1, 0), (0, 1)])
    w.push_light(_scenes.light(rng, 2))
    for variant in (2, 18):
        _check(w, rt.reference_camera(), rt.Frame.full(64, 48, 3), variant)


def test_empty_scene_renders_black():
    w = rt.World()
    img = _check(w, _scenes.camera(0), rt.Frame.full(40, 30, 5))
    assert not img.any()


def test_no_lights_and_single_pixel_frame():
    rng = np.random.default_rng(5)
    w = rt.World()
    w.push_object(_scenes.material(rng, "plain")).push_sphere((0, 0, 0), 1.0)
    cam = _scenes.camera(3)
    _check(w, cam, rt.Frame.full(33, 17, 5))
    _check(w, cam, rt.Frame.full(1, 1, 5))
    _check(w, cam, rt.Frame(64, 64, 3, 63, 63, 64, 64, 1))


def test_degenerate_triangle_and_coplanar_rays():
    """A zero-area triangle has a NaN face normal (0/0 in normalize): every comparison with NaN is false, so
    the reference ACCEPTS it as a hit with t = NaN (main.rs:205,224,229); the kernel must reproduce that."""
    rng = np.random.default_rng(9)
    w = rt.World()
    o = w.push_object(_scenes.material(rng, "plain"))
    o.push_flat_triangle([(0, 0, 0), (1, 0, 0), (2, 0, 0)], [(0, 0), (1, 0), (0, 1)])      # collinear -> NaN normal
    o.push_square([(-2, 0, -2), (-2, 0, 2), (2, 0, 2), (2, 0, -2)], [(0, 0), (0, 1), (1, 0), (0, 1)])
    w.push_light(_scenes.light(rng, 2))
    cam = rt.reference_camera()
    _check(w, cam, rt.Frame.full(48, 36, 3))
    # camera IN the floor plane looking along it: n.d == 0 and n.o == d exactly -> t = 0/0
    cam.center = (3.0, 0.0, 0.0)
    cam.toward = (-1.0, 0.0, 0.0)
    cam.near = 0.0
    _check(w, cam, rt.Frame.full(48, 36, 3))


@pytest.mark.parametrize("split", [1, 0], ids=["split", "fused"])
@pytest.mark.parametrize("seed,nt,ns,depth,epochs", [(11, 9, 2, 4, 2), (12, 70, 3, 5, 2), (13, 0, 2, 3, 3), (14, 140, 5, 9, 2)])
def test_random_scene_distributed(seed, nt, ns, depth, epochs, split):
    import torch
    from homework_18_graphics_raytracer_amd import _capi

    _capi.amd_lib().rt_set_distributed_split(split)

    world = _scenes.random_world(seed, nt, ns)
    cam = _scenes.camera(seed)
    frame = rt.Frame.full(64, 48, depth)
    scene = rt.Scene(world)
    rng = rt.Rng(frame)
    samples = torch.empty((epochs, frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
    valid = torch.empty((epochs, frame.rows, frame.cols), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    try:
        rt.render_distributed(scene, cam, frame, rng, epochs, samples=samples, valid=valid, ray_count=cnt)
        torch.cuda.synchronize()
    finally:
        _capi.amd_lib().rt_set_distributed_split(-1)
    st = _oracle.rng_init(frame)
    ws, wv, wc = _oracle.render_distributed(world.desc(), cam, frame, st, epochs)
    s = samples.cpu().numpy()
    same = (s.view(np.uint32) == ws.view(np.uint32)) | (np.isnan(s) & np.isnan(ws))
    assert same.all(), f"{(~same).sum()} channels differ"
    assert np.array_equal(valid.cpu().numpy(), wv) and int(cnt.item()) == wc
    assert np.array_equal(rng.download(), st)
