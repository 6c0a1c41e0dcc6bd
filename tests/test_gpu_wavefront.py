"""GPU parity of the persistent-wavefront path (variant bit 4, csrc/rt_pwf.hip: one persistent kernel whose workgroups
keep queues of single-cast work items), the default render path.  Same bar as the per-pixel kernel: radiance identical
bit for bit to the CPU oracle and equal World::cast counts — also when a frame does not fit the arenas and the
per-pixel kernel takes it over."""
import numpy as np
import pytest

import homework_18_graphics_raytracer_amd as rt
from homework_18_graphics_raytracer_amd import _capi
import _oracle
import _scenes

pytestmark = pytest.mark.gpu

PWF = 16
PATHS = [PWF | 2]


def _check(world, cam, frame, budget=None, scene=None, variant=PWF | 2):
    lib = _capi.amd_lib()
    scene = scene or rt.Scene(world)
    _capi.check(lib.rt_set_variant(variant))
    if budget is not None:
        _capi.check(lib.rt_set_wavefront_budget(budget))
    try:
        got, casts = rt.render_whitted_numpy(scene, cam, frame)
    finally:
        _capi.check(lib.rt_set_variant(_capi.DEFAULT_VARIANT))
        _capi.check(lib.rt_set_wavefront_budget(6))
    want, wcasts = _oracle.render_whitted(world.desc(), cam, frame)
    g, w = got.view(np.uint32), want.view(np.uint32)
    same = (g == w) | (np.isnan(got) & np.isnan(want))  # NaN payload/sign may differ between x86 and gfx950
    assert same.all(), f"{(~same).sum()} channels differ; first {np.argwhere(~same)[:3].tolist()}"
    assert casts == wcasts
    return got


@pytest.fixture(scope="module")
def ref():
    world = rt.reference_world()
    return world, rt.reference_camera(), rt.Scene(world)


@pytest.mark.parametrize("variant", PATHS)
@pytest.mark.parametrize("w,h,depth", [(256, 256, 1), (320, 240, 5), (200, 150, 8), (97, 61, 0), (64, 64, 3), (1003, 597, 3)])
def test_wavefront_bit_exact_on_the_reference_scene(ref, w, h, depth, variant):
    world, cam, scene = ref
    _check(world, cam, rt.Frame.full(w, h, depth), scene=scene, variant=variant)


@pytest.mark.parametrize("variant", PATHS)
def test_wavefront_reference_size(ref, variant):
    """The reference's own configuration: 1280x960, depth 5 (main.rs:1084-1085,1098)."""
    world, cam, scene = ref
    _check(world, cam, rt.Frame.full(1280, 960, 5), scene=scene, variant=variant)


HEADLINE_CASTS = 17756787  # World::cast evaluations of the 1920x1080 depth-8 frame (oracle; BENCH_r01's ray count)


def test_headline_configuration_device_path(ref):
    """BASELINE.json configs[1]/[2], the configuration the metric is quoted on: 1920x1080, depth 8, the Whitted loop
    (main.rs:1087-1109), through the stream-ordered device entry point bench.py times — radiance as u32 and the cast
    count against the oracle."""
    import torch

    world, cam, scene = ref
    frame = rt.Frame.full(1920, 1080, 8)
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    got = rt.render_whitted(scene, cam, frame, ray_count=count).cpu().numpy()
    want, wcasts = _oracle.render_whitted(world.desc(), cam, frame)
    assert wcasts == HEADLINE_CASTS and int(count.item()) == HEADLINE_CASTS
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("variant", PATHS)
@pytest.mark.parametrize("budget", [1, 2, 3])
def test_budget_overflow_falls_back_to_the_per_pixel_kernel(ref, budget, variant):
    """1 node per pixel cannot even hold the first reflections: the overflow flag must hand the whole frame (and
    its cast count) to the per-pixel kernel; 3 per pixel overflows somewhere in the deeper levels."""
    world, cam, scene = ref
    _check(world, cam, rt.Frame.full(200, 150, 8), budget=budget, scene=scene, variant=variant)


@pytest.mark.parametrize("variant", PATHS)
def test_generous_budget(ref, variant):
    world, cam, scene = ref
    _check(world, cam, rt.Frame.full(160, 120, 8), budget=64, scene=scene, variant=variant)


@pytest.mark.parametrize("seed,nt,ns", [(1, 0, 3), (2, 1, 0), (3, 2, 1), (4, 7, 2), (5, 33, 4), (6, 64, 0), (7, 65, 5), (8, 131, 3), (9, 200, 9)])
@pytest.mark.parametrize("variant", PATHS)
def test_wavefront_random_scenes(seed, nt, ns, variant):
    world = _scenes.random_world(seed, nt, ns)
    _check(world, _scenes.camera(seed), rt.Frame.full(96, 64, 5), budget=16, variant=variant)


@pytest.mark.parametrize("depth", [9, 12, 20, 32])
@pytest.mark.parametrize("variant", PATHS)
def test_wavefront_deep_recursion(depth, variant):
    world = rt.reference_world()
    _check(world, rt.reference_camera(), rt.Frame.full(80, 60, depth), budget=32, variant=variant)


@pytest.mark.parametrize("variant", PATHS)
def test_wavefront_row_tiles_equal_the_full_frame(ref, variant):
    """Interleaved row tiles (the multi-GPU sharding) through the wavefront paths."""
    world, cam, scene = ref
    full = _check(world, cam, rt.Frame.full(150, 101, 5), scene=scene, variant=variant)
    lib = _capi.amd_lib()
    _capi.check(lib.rt_set_variant(variant))
    try:
        for rank in range(3):
            band, _ = rt.render_whitted_numpy(scene, cam, rt.Frame.rows_of_rank(150, 101, 5, rank, 3))
            assert (band.view(np.uint32) == full[rank::3].view(np.uint32)).all()
    finally:
        _capi.check(lib.rt_set_variant(_capi.DEFAULT_VARIANT))


@pytest.mark.parametrize("w,h,depth", [(256, 256, 5), (250, 203, 3)])
def test_oversized_tiles_are_rendered_as_row_bands(ref, w, h, depth):
    """An arena holds 64 K ring slots at most; a tile whose budget needs more is split into bands of rows, one launch
    each.  A huge budget forces that on a small frame (two or three bands, the last one ragged)."""
    world, cam, scene = ref
    _check(world, cam, rt.Frame.full(w, h, depth), budget=1024, scene=scene)
    # ... also for an interleaved (multi-GPU) share of the frame
    lib = _capi.amd_lib()
    full, _ = _oracle.render_whitted(world.desc(), cam, rt.Frame.full(w, h, depth))
    _capi.check(lib.rt_set_wavefront_budget(2048))
    try:
        band, _ = rt.render_whitted_numpy(scene, cam, rt.Frame.rows_of_rank(w, h, depth, 1, 2))
    finally:
        _capi.check(lib.rt_set_wavefront_budget(6))
    assert (band.view(np.uint32) == full[1::2].view(np.uint32)).all()


@pytest.mark.parametrize("variant", [PWF | 2, 2])
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_clustered_objects_random(seed, variant):
    """Scenes whose objects become cluster segments (12-triangle boxes): whole-object skipping must not change a bit."""
    world = _scenes.clustered_world(seed, n_boxes=3 + seed % 3)
    _check(world, _scenes.camera(seed), rt.Frame.full(96, 64, 6), budget=16, variant=variant)


@pytest.mark.parametrize("variant", [PWF | 2, 2])
@pytest.mark.parametrize("seed,eye", [(11, (0.0, 0.5, 3.0)), (12, (0.5, 0.5, 3.0)), (13, (0.25, 1.0, 2.5)), (14, (0.0, -0.5, 3.0)), (15, (1.0, 0.0, 3.0))])
def test_clustered_objects_degenerate_rays(seed, eye, variant):
    """Axis-aligned boxes on a half-unit grid seen by an axis-aligned camera from a grid point: rays parallel to faces
    (n.d == 0: t = +-inf or NaN, NaN areas, which the reference ACCEPTS) and origins on face planes.  A cluster may only
    be skipped when no lane can produce such a hit."""
    world = _scenes.clustered_world(seed, n_boxes=5, axis_aligned=True)
    for w, h in ((64, 64), (65, 33)):  # even sizes put clip_x == 0 / clip_y == 0 on pixel centres
        _check(world, _scenes.axis_camera(eye), rt.Frame.full(w, h, 5), budget=16, variant=variant)


@pytest.mark.parametrize("variant", [PWF | 2, 2])
@pytest.mark.parametrize("seed,eye,toward", [(1, (0.0, 0.5, 3.0), (0.0, 0.0, -1.0)), (2, (0.5, 1.0, 3.0), (0.0, 0.0, -1.0)),
                                             (3, (3.0, 0.5, 0.0), (-1.0, 0.0, 0.0)), (4, (0.0, 3.0, 0.5), (0.0, -1.0, 0.0)),
                                             (5, (-0.5, 0.0, 2.5), (0.0, 0.0, -1.0)), (6, (1.0, 1.5, 2.0), (0.0, 0.0, -1.0))])
def test_plane_sharing_on_squares_with_degenerate_rays(seed, eye, toward, variant):
    """square() pairs share one plane evaluation in the intersection loop (rt_device_scene.h RT_TRI_FOLLOWS / _WEAK).  Their
    normals may differ in the signs of zero components, which reaches a result only through n.d == +-0 — exactly what an
    axis-aligned camera on a grid point produces (rays parallel to planes, origins ON planes: t = +-inf, NaN, +-0)."""
    world = _scenes.squares_world(seed)
    cam = _scenes.axis_camera(eye, toward)
    if toward[1] != 0.0:
        cam.up = (0.0, 0.0, 1.0)
    for w, h in ((64, 64), (65, 33)):
        _check(world, cam, rt.Frame.full(w, h, 5), budget=16, variant=variant)
    _check(world, _scenes.camera(seed), rt.Frame.full(96, 64, 6), budget=16, variant=variant)


def test_two_streams_render_concurrently_with_their_own_workspaces(ref):
    """A scene may be rendered from several streams at once; each stream has its own arenas."""
    import torch

    world, cam, scene = ref
    frames = [rt.Frame.full(320, 240, 6), rt.Frame.full(257, 199, 8)]
    want = [rt.render_whitted(scene, cam, f).clone() for f in frames]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [torch.zeros_like(w) for w in want]
    for _ in range(5):
        for o in outs:
            o.zero_()
        torch.cuda.synchronize()
        for k in range(2):
            for _rep in range(3):  # back to back on each stream, interleaved between the streams
                rt.render_whitted(scene, cam, frames[k], out=outs[k], stream=streams[k])
                rt.render_whitted(scene, cam, frames[1 - k], out=outs[1 - k], stream=streams[1 - k])
        torch.cuda.synchronize()
        for k in range(2):
            assert torch.equal(outs[k].view(torch.int32), want[k].view(torch.int32))


@pytest.mark.parametrize("share", [2, 4, 1000])
def test_frames_in_flight_side_by_side_on_parts_of_the_device(ref, share):
    """RT_AMD_WF_SHARE = n: a launch of the persistent kernel takes 1/n of the workgroups the device holds (one workgroup at the
    least), so that n frames in flight on n streams run side by side (dist.FramePipeline(in_flight=n), bench.py for N > 1).  The same
    pixels and cast counts as the oracle's, whatever the share."""
    import torch

    world, cam, scene = ref
    frames = [rt.Frame.full(320, 240, 8), rt.Frame.rows_of_rank(640, 360, 6, 0, 4)]
    want = [_oracle.render_whitted(world.desc(), cam, f) for f in frames]
    streams = [torch.cuda.Stream() for _ in range(4)]
    with rt.options(RT_AMD_WF_SHARE=share):
        for k, f in enumerate(frames):
            outs = [torch.zeros((f.rows, f.cols, 3), dtype=torch.float32, device="cuda") for _ in streams]
            cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
            for rep in range(2):
                for o, st in zip(outs, streams):
                    rt.render_whitted(scene, cam, f, out=o, ray_count=cnt, stream=st)
            torch.cuda.synchronize()
            for o in outs:
                assert np.array_equal(o.cpu().numpy().view(np.uint32), want[k][0].view(np.uint32))
            assert int(cnt.item()) == 2 * len(streams) * want[k][1]


def test_choose_streams_returns_streams_that_render_the_same_frame(ref):
    """dist.choose_streams (bench.py for N > 1): a few sets of streams timed with frames in flight, the best kept; whatever set it is,
    the frames rendered on it are the oracle's."""
    import torch

    from homework_18_graphics_raytracer_amd import dist as rtdist

    world, cam, scene = ref
    f = rt.Frame.rows_of_rank(480, 270, 8, 0, 2)
    bands = [torch.zeros((f.rows, f.cols, 3), dtype=torch.float32, device="cuda") for _ in range(3)]
    with rt.options(RT_AMD_WF_SHARE=3):
        streams, tried = rtdist.choose_streams(lambda i: rt.render_whitted(scene, cam, f, out=bands[i]), 3, attempts=2, frames_per_stream=2)
        assert len(streams) == 3 and len(tried) == 2 and all(t > 0 for t in tried)
        for b in bands:
            b.zero_()
        for i, st in enumerate(streams):
            rt.render_whitted(scene, cam, f, out=bands[i], stream=st)
        torch.cuda.synchronize()
    want, _ = _oracle.render_whitted(world.desc(), cam, f)
    for b in bands:
        assert np.array_equal(b.cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_two_host_threads_render_concurrently_with_profiling_on(ref):
    """rt_render_whitted from two HOST threads, each on its own stream, with the profiling hooks enabled: the settings
    are atomics, the workspaces are created under the scene's lock and a call's event pair is thread-local, so both
    threads get bit-identical frames and every launch is timed exactly once (include/rt_amd.h "Threading")."""
    import ctypes as C
    import threading

    import torch

    world, cam, scene = ref
    lib = _capi.amd_lib()
    frames = [rt.Frame.full(320, 240, 6), rt.Frame.full(257, 199, 8)]
    want = [rt.render_whitted(scene, cam, f).clone() for f in frames]
    torch.cuda.synchronize()
    reps, errors = 12, []
    outs = [[torch.zeros_like(want[k]) for _ in range(reps)] for k in range(2)]
    _capi.check(lib.rt_profile_enable(1))
    start = threading.Barrier(2)

    def worker(k):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            start.wait()
            for r in range(reps):
                rt.render_whitted(scene, cam, frames[k], out=outs[k][r], stream=stream)
            stream.synchronize()
        except Exception as exc:  # surfaced below: an exception in a thread must fail the test
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    ms, n = C.c_double(0.0), C.c_uint(0)
    _capi.check(lib.rt_profile_read(C.byref(ms), C.byref(n)))
    _capi.check(lib.rt_profile_enable(0))
    assert not errors, errors
    assert n.value == 2 * reps and ms.value > 0.0
    for k in range(2):
        for r in range(reps):
            assert torch.equal(outs[k][r].view(torch.int32), want[k].view(torch.int32)), (k, r)


def test_no_memory_for_the_arenas_means_the_per_pixel_kernel(ref):
    """If the arenas cannot be allocated the frame is rendered by the per-pixel kernel in the same call (the hook makes
    the allocation fail; a scene of its own, because the workspace is kept per scene and stream)."""
    import os

    world, cam, _ = ref
    rt.set_option("RT_AMD_DIAG_WS_REFUSE", "1")
    try:
        _check(world, cam, rt.Frame.full(200, 150, 6), scene=rt.Scene(world))
    finally:
        rt.set_option("RT_AMD_DIAG_WS_REFUSE", None)


def _mismatches(got, want):
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))  # as in _check
    return int((~same).sum())


def test_state_between_calls_on_one_stream(ref):
    """A workspace keeps the frame description and a zeroed block of counters from one call to the next (rt_api.hip: the
    kernel's last workgroup closes the frame and prepares the next launch).  Alternate frames, let one overflow into the
    per-pixel kernel in between, and come back: every image and every cast count as on a fresh stream."""
    import torch

    world, cam, scene = ref
    lib = _capi.amd_lib()
    frames = [rt.Frame.full(320, 200, 8), rt.Frame.full(160, 120, 5), rt.Frame(640, 400, 8, 64, 32, 64 + 320, 32 + 200, 1)]
    want = []
    for f in frames:
        w, c = _oracle.render_whitted(world.desc(), cam, f)
        want.append((w, c))
    stream = torch.cuda.Stream()
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    order = [0, 0, 1, 0, 2, 2, 1, 1, 0]
    with torch.cuda.stream(stream):
        for step, k in enumerate(order):
            if step == 4:  # an overflowing frame in between: budget 1 hands it to the per-pixel kernel
                _capi.check(lib.rt_set_wavefront_budget(1))
            count.zero_()
            got = rt.render_whitted(scene, cam, frames[k], ray_count=count, stream=stream)
            if step == 4:
                _capi.check(lib.rt_set_wavefront_budget(6))
            stream.synchronize()
            bad = _mismatches(got.cpu().numpy(), want[k][0])
            assert bad == 0, f"step {step}: frame {k}: {bad} channels differ"
            assert int(count.item()) == want[k][1], f"step {step}: frame {k}"


def test_captured_call_replays_between_direct_calls(ref):
    """include/rt_amd.h "State between calls": a render call captured into a graph prepares its own state, and so does
    every later launch on that stream — replays and direct calls with other frames may alternate."""
    import torch

    world, cam, scene = ref
    fa, fb = rt.Frame.full(256, 160, 8), rt.Frame.full(200, 96, 4)
    wa, ca = _oracle.render_whitted(world.desc(), cam, fa)
    wb, cb = _oracle.render_whitted(world.desc(), cam, fb)
    stream = torch.cuda.Stream()
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    out_a = torch.empty((fa.rows, fa.cols, 3), dtype=torch.float32, device="cuda")
    with torch.cuda.stream(stream):
        rt.render_whitted(scene, cam, fa, out=out_a, stream=stream)  # the workspace exists before the capture
        rt.render_whitted(scene, cam, fa, out=out_a, stream=stream)  # ... and is in its steady state
    stream.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        rt.render_whitted(scene, cam, fa, out=out_a, ray_count=count, stream=torch.cuda.current_stream())
    for step in range(6):
        if step % 2 == 0:
            out_a.zero_()
            count.zero_()
            graph.replay()
            torch.cuda.synchronize()
            assert _mismatches(out_a.cpu().numpy(), wa) == 0, f"replay at step {step}"
            assert int(count.item()) == ca
        else:
            with torch.cuda.stream(stream):
                count.zero_()
                got = rt.render_whitted(scene, cam, fb if step != 3 else fa, ray_count=count, stream=stream)
            stream.synchronize()
            w, c = (wb, cb) if step != 3 else (wa, ca)
            assert _mismatches(got.cpu().numpy(), w) == 0, f"direct call at step {step}"
            assert int(count.item()) == c


def test_frame_large_enough_for_packed_page_counters(ref):
    """Beyond ~4 Mpixel the arena queues' page counters are packed two to a word so that three workgroups per CU still fit
    in LDS (rt_pwf_common.h pa_ready_packed: the kernel's second instantiation): 3840x2160 against the oracle."""
    world, cam, scene = ref
    frame = rt.Frame.full(3840, 2160, 3)
    got, casts = rt.render_whitted_numpy(scene, cam, frame)
    want, wcasts = _oracle.render_whitted(world.desc(), cam, frame)
    assert _mismatches(got, want) == 0
    assert casts == wcasts


@pytest.mark.parametrize("n_lights", [0, 1, 2, 4, 5, 9])
@pytest.mark.parametrize("variant", PATHS + [2])
def test_other_numbers_of_lights(n_lights, variant):
    """get_shade's loop (main.rs:413-461) over none, fewer and more lights than the reference scene's three — the search
    for an item's next light runs with the wave in step, and a build with SHADE queues by light (PA_LQ) files lights beyond
    its queues together — on the wavefront path and the per-pixel kernel."""
    world = _scenes.random_world(40 + n_lights, 30, 3, n_lights=n_lights)
    _check(world, _scenes.camera(7), rt.Frame.full(120, 80, 5), budget=16, variant=variant)


@pytest.mark.parametrize("n_lights", [(1 << 14) - 1, 1 << 14])
def test_as_many_lights_as_a_shade_item_can_name(n_lights):
    """A SHADE item names its next light in 14 bits (rt_pwf.hip): a scene with 2^14 - 1 lights still renders on the wavefront path,
    one with 2^14 is handed to the per-pixel kernel — the oracle's frame and cast count either way."""
    world = _scenes.random_world(5, 12, 1, n_lights=n_lights)
    _check(world, _scenes.camera(3), rt.Frame.full(16, 8, 1), variant=PWF | 2)


def test_arenas_that_fill_up_with_own_tiles_left_fall_back(ref):
    """Half of a workgroup's even share of the tiles are its own (rt_pwf.hip); a workgroup whose arena fills up stops taking tiles, its
    own ones included, and nobody else will take those: the frame's tile count then falls short, the last workgroup raises the
    overflow flag and the per-pixel kernel renders the frame.  1280x720 at one node per pixel: 18 tiles per workgroup, arenas a third
    of what the frame needs — the oracle's frame and cast count all the same."""
    world, cam, scene = ref
    _check(world, cam, rt.Frame.full(1280, 720, 8), budget=1, scene=scene)
