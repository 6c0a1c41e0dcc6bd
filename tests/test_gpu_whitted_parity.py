"""GPU parity proper: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

Integer/bit-exact bar: the f32 radiance buffer must be IDENTICAL (compared as u32 bit patterns), and
the number of World::cast evaluations must be equal.  Sizes are chosen so the oracle finishes in seconds.
"""
import ctypes as C

import numpy as np
import pytest

import homework_18_graphics_raytracer_amd as rt
import _oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch

    assert torch.cuda.is_available()
    world = rt.reference_world()
    return world, rt.reference_camera(), rt.Scene(world)


def _compare(world, camera, scene, frame, variant=18):
    from homework_18_graphics_raytracer_amd import _capi

    _capi.check(_capi.amd_lib().rt_set_variant(variant))
    try:
        got, casts = rt.render_whitted_numpy(scene, camera, frame)
    finally:
        _capi.check(_capi.amd_lib().rt_set_variant(_capi.DEFAULT_VARIANT))
    want, want_casts = _oracle.render_whitted(world.desc(), camera, frame)
    diff = got.view(np.uint32) != want.view(np.uint32)
    if diff.any():
        idx = np.argwhere(diff)
        msg = [f"{len(idx)} of {diff.size} channels differ; first: {idx[:5].tolist()}"]
        for y, x, c in idx[:5]:
            msg.append(f"  ({y},{x},{c}) gpu={got[y, x, c]!r} oracle={want[y, x, c]!r}")
        pytest.fail("\n".join(msg))
    assert casts == want_casts
    return got, casts


@pytest.mark.parametrize("variant", [2, 3, 18, 19])
@pytest.mark.parametrize("w,h,depth", [(256, 256, 1), (320, 240, 5), (200, 150, 8), (97, 61, 0), (64, 64, 3), (1003, 597, 3)])
def test_whitted_bit_exact(ctx, w, h, depth, variant):
    """Every selectable render path: the per-pixel kernel with scalar (2) / LDS-staged (3) triangle records, the persistent
    wavefront kernel (18, the default; 19 = with the LDS per-pixel kernel as its fallback)."""
    world, camera, scene = ctx
    _compare(world, camera, scene, rt.Frame.full(w, h, depth), variant)


def test_whitted_reference_size_bit_exact(ctx):
    """The reference's own configuration: 1280x960, depth 5 (main.rs:1084-1085,1098)."""
    world, camera, scene = ctx
    _compare(world, camera, scene, rt.Frame.full(1280, 960, 5))


def test_whitted_tiles_equal_full_frame(ctx):
    """Interleaved row tiles (multi-GPU sharding) reassemble to the full frame exactly."""
    world, camera, scene = ctx
    w, h, depth, n = 192, 108, 8, 4
    full, casts = rt.render_whitted_numpy(scene, camera, rt.Frame.full(w, h, depth))
    asm = np.zeros_like(full)
    total = 0
    for r in range(n):
        fr = rt.Frame.rows_of_rank(w, h, depth, r, n)
        tile, c = rt.render_whitted_numpy(scene, camera, fr)
        asm[r::n] = tile
        total += c
    assert np.array_equal(asm.view(np.uint32), full.view(np.uint32))
    assert total == casts


def test_whitted_sub_rectangle(ctx):
    world, camera, scene = ctx
    w, h, depth = 160, 120, 5
    full, _ = rt.render_whitted_numpy(scene, camera, rt.Frame.full(w, h, depth))
    fr = rt.Frame(w, h, depth, 37, 11, 101, 83, 1)
    tile, _ = rt.render_whitted_numpy(scene, camera, fr)
    assert np.array_equal(tile.view(np.uint32), full[11:83, 37:101].view(np.uint32))


def test_device_output_and_ray_counter(ctx):
    import torch

    world, camera, scene = ctx
    frame = rt.Frame.full(128, 96, 5)
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    out = rt.render_whitted(scene, camera, frame, ray_count=cnt)
    out2 = rt.render_whitted(scene, camera, frame, ray_count=cnt)
    torch.cuda.synchronize()
    want, want_casts = _oracle.render_whitted(world.desc(), camera, frame)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert torch.equal(out, out2)
    assert int(cnt.item()) == 2 * want_casts  # the counter accumulates


@pytest.mark.parametrize("axis,scale", [((1.0, 2.0, 3.0), 0.37), ((2.0, 1.0, -1.0), 2.5), ((0.3, -1.0, 0.2), 1.0)])
def test_points_behind_a_spot_light_on_its_axis(axis, scale):
    """lights.rs:57-60 for a point behind the light, on its axis: the angle's cosine rounds to just below -1 for some pixels, acos is
    NaN, `NaN > spread` is false — the light asks and its colour is NaN.  The shortcut that answers "outside the cone" from the
    cosine alone (rt_shade.h light_asks) must leave those to the reference's expression.  Whitted pass (both kernels) and
    the depth-of-field pass's shade kernel; NaN pixels must sit where the oracle's do, everything else bit for bit."""
    import torch

    import _scenes
    from homework_18_graphics_raytracer_amd import _capi

    world, cam = _scenes.behind_spot_world(axis, scale)
    scene = rt.Scene(world)
    frame = rt.Frame.full(96, 96, 3)
    want, want_casts = _oracle.render_whitted(world.desc(), cam, frame)
    assert np.isnan(want).any(), "the configuration no longer reaches the case it was built for"
    for variant in (18, 2):
        _capi.check(_capi.amd_lib().rt_set_variant(variant))
        try:
            got, casts = rt.render_whitted_numpy(scene, cam, frame)
        finally:
            _capi.check(_capi.amd_lib().rt_set_variant(_capi.DEFAULT_VARIANT))
        assert casts == want_casts
        assert ((got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))).all()
    tile = rt.Frame.full(96, 96, 3)
    rng = rt.Rng(tile)
    samples = torch.empty((2, 96, 96, 3), dtype=torch.float32, device="cuda")
    valid = torch.empty((2, 96, 96), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    rt.render_distributed(scene, cam, tile, rng, 2, blur=0.0, samples=samples, valid=valid, ray_count=cnt)
    torch.cuda.synchronize()
    st = _oracle.rng_init(tile)
    ws, wv, wc = _oracle.render_distributed(world.desc(), cam, tile, st, 2, blur=0.0)
    s = samples.cpu().numpy()
    assert np.isnan(ws).any()
    assert ((s.view(np.uint32) == ws.view(np.uint32)) | (np.isnan(s) & np.isnan(ws))).all()
    assert np.array_equal(valid.cpu().numpy(), wv) and int(cnt.item()) == wc and np.array_equal(rng.download(), st)
