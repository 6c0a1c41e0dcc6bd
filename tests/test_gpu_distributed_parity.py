"""GPU parity of the distributed (stochastic / DoF) pass: samples, filter flags, RNG states and cast counts
of the HIP kernels against the oracle, bit for bit, for the same seeds (y*2^33 + x) — for both organisations of the pass
(rt_set_distributed_split: 1 = the persistent-lane chain kernel (pair-wise cast) + shade + unwind, the default; 0 = the single
fused kernel)."""
import os

import numpy as np
import pytest

import homework_18_graphics_raytracer_amd as rt
from homework_18_graphics_raytracer_amd import _capi
import _oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[1, 0], ids=["split", "fused"])
def organisation(request):
    lib = _capi.amd_lib()
    lib.rt_set_distributed_split(request.param)
    yield request.param
    lib.rt_set_distributed_split(-1)


@pytest.fixture(scope="module")
def ctx():
    world = rt.reference_world()
    return world, rt.reference_camera(), rt.Scene(world)


def _run_gpu(scene, camera, frame, n_epochs, rng=None, accum=None):
    import torch

    rng = rng or rt.Rng(frame)
    samples = torch.empty((n_epochs, frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
    valid = torch.empty((n_epochs, frame.rows, frame.cols), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    rt.render_distributed(scene, camera, frame, rng, n_epochs, accum=accum, samples=samples, valid=valid, ray_count=cnt)
    torch.cuda.synchronize()
    return rng, samples.cpu().numpy(), valid.cpu().numpy(), int(cnt.item())


def test_rng_seeding_matches_oracle(ctx):
    frame = rt.Frame(64, 48, 5, 3, 5, 40, 41, 2)  # a sub-tile with a row step: seeds use IMAGE coordinates
    rng = rt.Rng(frame)
    assert np.array_equal(rng.download(), _oracle.rng_init(frame))


@pytest.mark.parametrize("w,h,depth,epochs", [(96, 72, 5, 3), (64, 48, 8, 2), (80, 60, 0, 2), (33, 21, 2, 14)])
def test_distributed_bit_exact(ctx, w, h, depth, epochs):
    world, camera, scene = ctx
    frame = rt.Frame.full(w, h, depth)
    rng, s, v, casts = _run_gpu(scene, camera, frame, epochs)
    st = _oracle.rng_init(frame)
    ws, wv, wcasts = _oracle.render_distributed(world.desc(), camera, frame, st, epochs)
    assert np.array_equal(v, wv)
    bad = s.view(np.uint32) != ws.view(np.uint32)
    assert not bad.any(), f"{bad.sum()} channels differ, first at {np.argwhere(bad)[:3].tolist()}"
    assert casts == wcasts
    assert np.array_equal(rng.download(), st)  # same number of draws everywhere: the streams stay in step


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_other_scenes_squares_clusters_and_random_triangles(seed):
    """Scenes other than the reference's: axis-aligned squares (triangle pairs that share a plane evaluation in the
    intersection loop), clustered boxes and random triangles with generative materials — incoherent scattered rays through
    the same loop; blur 0 makes every primary ray of the axis-aligned camera exactly axis-parallel in some component."""
    import _scenes

    for world, cam, blur in ((_scenes.squares_world(seed), _scenes.axis_camera((0.5, 0.5, 3.0)), 0.0),
                             (_scenes.squares_world(seed + 10), _scenes.camera(seed), 0.04),
                             (_scenes.clustered_world(seed, n_boxes=3), _scenes.camera(seed), 0.02),
                             (_scenes.random_world(seed, 40, 3), _scenes.camera(seed), 0.04)):
        scene = rt.Scene(world)
        frame = rt.Frame.full(64, 48, 5)
        import torch

        rng = rt.Rng(frame)
        samples = torch.empty((3, frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
        valid = torch.empty((3, frame.rows, frame.cols), dtype=torch.uint8, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
        rt.render_distributed(scene, cam, frame, rng, 3, focus=3.0, blur=blur, samples=samples, valid=valid, ray_count=cnt)
        torch.cuda.synchronize()
        st = _oracle.rng_init(frame)
        ws, wv, wcasts = _oracle.render_distributed(world.desc(), cam, frame, st, 3, focus=3.0, blur=blur)
        s = samples.cpu().numpy()
        same = (s.view(np.uint32) == ws.view(np.uint32)) | (np.isnan(s) & np.isnan(ws))
        assert same.all(), f"{(~same).sum()} channels differ"
        assert np.array_equal(valid.cpu().numpy(), wv) and int(cnt.item()) == wcasts
        assert np.array_equal(rng.download(), st)


@pytest.mark.parametrize("pipeline", ["1", "0"])
@pytest.mark.parametrize("cap_mb,epochs", [("1", 3), ("8", 3), ("8", 5), ("16", 7)])
def test_split_pass_in_batches_of_epochs(ctx, organisation, cap_mb, epochs, pipeline):
    """A small workspace cap makes the split pass run the call in batches (1 epoch; 2 + a short last one) — over two workspaces
    used in turn, batch k's shade and unwind kernels beside batch k+1's chain kernel (the default), or in line over one."""
    if not organisation:
        pytest.skip("the fused kernel has no workspace")
    import torch

    world, camera, scene = ctx
    frame = rt.Frame.full(96, 72, 5)
    accum = torch.zeros((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
    rt.set_option("RT_AMD_DIST_WS_MB", cap_mb)
    rt.set_option("RT_AMD_DIST_PIPELINE", pipeline)
    try:
        rng, s, v, casts = _run_gpu(scene, camera, frame, epochs, accum=accum)
    finally:
        rt.set_option("RT_AMD_DIST_WS_MB", None)
        rt.set_option("RT_AMD_DIST_PIPELINE", None)
    st = _oracle.rng_init(frame)
    ws, wv, wcasts = _oracle.render_distributed(world.desc(), camera, frame, st, epochs)
    assert np.array_equal(v, wv) and np.array_equal(s.view(np.uint32), ws.view(np.uint32)) and casts == wcasts
    assert np.array_equal(rng.download(), st)
    want = np.zeros((frame.rows, frame.cols, 3), dtype=np.float32)
    for e in range(epochs):
        want = np.where(wv[e][..., None] != 0, want + ws[e], want)
    assert np.array_equal(accum.cpu().numpy().view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("lookahead", ["1", "0"])
@pytest.mark.parametrize("calls", [(70,), (9, 1, 30, 30), (1,) * 12 + (58,)])
def test_long_streams_cross_many_isaac_blocks(ctx, lookahead, calls):
    """~10 words per sample: 70 epochs use two to three 256-word blocks per pixel.  The device keeps the next block
    prepared ahead of time (look-ahead pass before every batch); a pixel that needs a second block within one visit
    generates it in the render kernel.  Every mix of the two must leave the reference's stream and the reference's record."""
    world, camera, scene = ctx
    frame = rt.Frame.full(24, 17, 8)
    rt.set_option("RT_AMD_RNG_LOOKAHEAD", lookahead)
    try:
        rng, got, flags, casts = None, [], [], 0
        for n in calls:
            rng, s, v, c = _run_gpu(scene, camera, frame, n, rng=rng)
            got.append(s)
            flags.append(v)
            casts += c
    finally:
        rt.set_option("RT_AMD_RNG_LOOKAHEAD", None)
    st = _oracle.rng_init(frame)
    ws, wv, wcasts = _oracle.render_distributed(world.desc(), camera, frame, st, sum(calls))
    assert np.array_equal(np.concatenate(got).view(np.uint32), ws.view(np.uint32))
    assert np.array_equal(np.concatenate(flags), wv) and casts == wcasts
    assert np.array_equal(rng.download(), st)


@pytest.mark.parametrize("focus,blur", [(3.0, 0.0), (1.5, 0.3), (8.0, 0.004), (0.5, 2.0)])
def test_other_lens_settings_and_a_sub_rectangle(ctx, focus, blur):
    """shoot_focus with other focus distances and aperture blurs (0: every sample through the pinhole; 2.0: origins far
    off the lens axis), on a tile that is a sub-rectangle with a row step — seeds and clip coordinates use IMAGE x / y."""
    import torch

    world, camera, scene = ctx
    frame = rt.Frame(120, 90, 6, 17, 5, 83, 77, 3)
    rng = rt.Rng(frame)
    samples = torch.empty((4, frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
    valid = torch.empty((4, frame.rows, frame.cols), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    rt.render_distributed(scene, camera, frame, rng, 4, focus=focus, blur=blur, samples=samples, valid=valid, ray_count=cnt)
    torch.cuda.synchronize()
    st = _oracle.rng_init(frame)
    ws, wv, wcasts = _oracle.render_distributed(world.desc(), camera, frame, st, 4, focus=focus, blur=blur)
    s = samples.cpu().numpy()
    same = (s.view(np.uint32) == ws.view(np.uint32)) | (np.isnan(s) & np.isnan(ws))
    assert same.all() and np.array_equal(valid.cpu().numpy(), wv) and int(cnt.item()) == wcasts
    assert np.array_equal(rng.download(), st)


def test_organisations_can_alternate_on_one_rng(ctx, organisation):
    """The organisations share the RNG records (banks, look-ahead flags): switching between calls continues the stream."""
    world, camera, scene = ctx
    frame = rt.Frame.full(48, 36, 7)
    lib = _capi.amd_lib()
    rng, got = None, []
    for k, n in enumerate((3, 12, 2, 30, 5)):
        lib.rt_set_distributed_split((organisation + k) % 2)
        rng, s, _, _ = _run_gpu(scene, camera, frame, n, rng=rng)
        got.append(s)
    st = _oracle.rng_init(frame)
    ws, _, _ = _oracle.render_distributed(world.desc(), camera, frame, st, 52)
    assert np.array_equal(np.concatenate(got).view(np.uint32), ws.view(np.uint32))
    assert np.array_equal(rng.download(), st)


@pytest.mark.parametrize("depth", [13, 32])
def test_deep_chains(ctx, depth):
    """More request slots than the shade kernel's default tile can list in LDS: it shrinks its tile (33 slots at depth 32)."""
    world, camera, scene = ctx
    frame = rt.Frame.full(40, 30, depth)
    rng, s, v, casts = _run_gpu(scene, camera, frame, 3)
    st = _oracle.rng_init(frame)
    ws, wv, wcasts = _oracle.render_distributed(world.desc(), camera, frame, st, 3)
    assert np.array_equal(s.view(np.uint32), ws.view(np.uint32)) and np.array_equal(v, wv) and casts == wcasts
    assert np.array_equal(rng.download(), st)


def test_two_tiles_on_two_streams(ctx):
    """Two tiles of one scene, each with its own rt_rng, rendered from two streams at once (per-stream workspaces, each
    rt_rng's own look-ahead stream): same samples as one after the other."""
    import torch

    world, camera, scene = ctx
    frames = [rt.Frame.rows_of_rank(96, 64, 6, 0, 2), rt.Frame.rows_of_rank(96, 64, 6, 1, 2)]
    want = []
    for f in frames:
        _, s, v, _ = _run_gpu(scene, camera, f, 5)
        want.append((s, v))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    rngs = [rt.Rng(f) for f in frames]
    outs = [torch.empty((5, f.rows, f.cols, 3), dtype=torch.float32, device="cuda") for f in frames]
    flags = [torch.empty((5, f.rows, f.cols), dtype=torch.uint8, device="cuda") for f in frames]
    torch.cuda.synchronize()
    for first, n in ((0, 2), (2, 3)):  # two calls per stream, interleaved between the streams
        for k in range(2):
            rt.render_distributed(scene, camera, frames[k], rngs[k], n, samples=outs[k][first:first + n], valid=flags[k][first:first + n], stream=streams[k])
    torch.cuda.synchronize()
    for k in range(2):
        assert np.array_equal(outs[k].cpu().numpy().view(np.uint32), want[k][0].view(np.uint32))
        assert np.array_equal(flags[k].cpu().numpy(), want[k][1])


def test_accumulate_only_call_equals_the_sample_outputs(ctx):
    """d_samples / d_valid / d_ray_count are optional: the accumulator alone must see the same sums."""
    import torch

    world, camera, scene = ctx
    frame = rt.Frame.full(70, 50, 6)
    a1 = torch.zeros((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
    a2 = torch.zeros_like(a1)
    rng1 = rt.Rng(frame)
    rt.render_distributed(scene, camera, frame, rng1, 4, accum=a1)
    rng2, _, _, _ = _run_gpu(scene, camera, frame, 4, accum=a2)
    torch.cuda.synchronize()
    assert torch.equal(a1.view(torch.int32), a2.view(torch.int32))
    assert np.array_equal(rng1.download(), rng2.download())


def test_stream_continues_across_calls_and_accumulates_in_epoch_order(ctx):
    import torch

    world, camera, scene = ctx
    frame = rt.Frame.full(72, 54, 5)
    accum = torch.zeros((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
    rng, s1, v1, _ = _run_gpu(scene, camera, frame, 2, accum=accum)
    _, s2, v2, _ = _run_gpu(scene, camera, frame, 3, rng=rng, accum=accum)
    st = _oracle.rng_init(frame)
    ws, wv, _ = _oracle.render_distributed(world.desc(), camera, frame, st, 5)
    assert np.array_equal(np.concatenate([s1, s2]).view(np.uint32), ws.view(np.uint32))
    want = np.zeros((frame.rows, frame.cols, 3), dtype=np.float32)
    for e in range(5):  # img[at] = img[at] + photon for the surviving samples, epoch after epoch (main.rs:1163-1167)
        want = np.where(wv[e][..., None] != 0, want + ws[e], want)
    assert np.array_equal(accum.cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_tile_with_row_step_matches_full_frame_rows(ctx):
    world, camera, scene = ctx
    full = rt.Frame.full(64, 48, 5)
    _, s_full, v_full, _ = _run_gpu(scene, camera, full, 2)
    band = rt.Frame.rows_of_rank(64, 48, 5, 1, 4)
    _, s_band, v_band, _ = _run_gpu(scene, camera, band, 2)
    assert np.array_equal(s_band.view(np.uint32), s_full[:, 1::4].view(np.uint32)) and np.array_equal(v_band, v_full[:, 1::4])


def test_full_size_frame_of_the_headline_configuration(ctx, organisation):
    """BASELINE.json's size for the stochastic configuration — 1920x1080, depth 8 — two epochs, every sample against the
    oracle (all host threads); the split organisation also GPU against GPU with a workspace cap that forces one-epoch batches."""
    world, camera, scene = ctx
    frame = rt.Frame.full(1920, 1080, 8)
    rng, s, v, casts = _run_gpu(scene, camera, frame, 2)
    st = _oracle.rng_init(frame)
    ws, wv, wcasts = _oracle.render_distributed(world.desc(), camera, frame, st, 2)
    assert np.array_equal(s.view(np.uint32), ws.view(np.uint32)) and np.array_equal(v, wv) and casts == wcasts
    assert np.array_equal(rng.download(), st)
    if organisation:
        rt.set_option("RT_AMD_DIST_WS_MB", "2048")
        try:
            rng2, s2, v2, casts2 = _run_gpu(scene, camera, frame, 2)
        finally:
            rt.set_option("RT_AMD_DIST_WS_MB", None)
        assert np.array_equal(s2.view(np.uint32), s.view(np.uint32)) and np.array_equal(v2, v) and casts2 == casts
        assert np.array_equal(rng2.download(), st)


def test_scatter_job_of_configs4_equals_the_oracle(ctx):
    """BASELINE.json configs[4] as SURVEY §8(d) states it ("Config 5"): 10 368 000 = 1920 x 1080 x 5 (pixel, epoch) samples of
    distributed_ray_trace (main.rs:521-614, loop 1129-1161) from freshly seeded streams, depth 8, in ONE call of the default
    organisation — the job bench.py times as `scatter_pass`.  Every sample, every filter flag, every generator record (516 words
    per pixel after five epochs) and the cast count against the oracle; and the accumulated image against the oracle's sum."""
    import torch

    world, camera, scene = ctx
    frame = rt.Frame.full(1920, 1080, 8)
    epochs = 5
    assert frame.rows * frame.cols * epochs == 10_368_000
    rng = rt.Rng(frame)
    samples = torch.empty((epochs, frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
    valid = torch.empty((epochs, frame.rows, frame.cols), dtype=torch.uint8, device="cuda")
    accum = torch.zeros((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    rt.render_distributed(scene, camera, frame, rng, epochs, accum=accum, samples=samples, valid=valid, ray_count=count)
    torch.cuda.synchronize()
    st = _oracle.rng_init(frame)
    ws, wv, wcasts = _oracle.render_distributed(world.desc(), camera, frame, st, epochs)
    s, v = samples.cpu().numpy(), valid.cpu().numpy()
    assert np.array_equal(v, wv) and int(count.item()) == wcasts
    assert ((s.view(np.uint32) == ws.view(np.uint32)) | (np.isnan(s) & np.isnan(ws))).all()
    assert np.array_equal(rng.download(), st)
    want = np.zeros((frame.rows, frame.cols, 3), dtype=np.float32)
    for e in range(epochs):  # main.rs:1157-1167: the filter, then img += photon, epoch by epoch
        want += np.where(wv[e][..., None] != 0, ws[e], np.float32(0))
    assert np.array_equal(accum.cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_configs3_job_as_stated_equals_the_oracle(ctx, organisation):
    """BASELINE.json configs[3] exactly as bench.py times it (`stochastic_pass`): 64 depth-of-field samples per pixel of the
    1920 x 1080 depth-8 frame from freshly seeded streams in ONE rt_render_distributed call of the default organisation — at this
    size eight batches of eight epochs pipelined over two workspaces (batch k's shade and unwind kernels beside batch k+1's chain
    kernel, the look-ahead on its own stream).  Checked against the oracle: the accumulator (the epoch-ordered, filtered sum of
    main.rs:1157-1167), EVERY generator record after the 64 epochs (516 words per pixel), and the cast count.  The oracle runs the
    same loop eight epochs at a time (its samples are folded into the expected sum and dropped)."""
    if not organisation:
        pytest.skip("configs[3] is stated for the default (split) organisation")
    import torch

    world, camera, scene = ctx
    frame = rt.Frame.full(1920, 1080, 8)
    epochs = 64
    rng = rt.Rng(frame)
    accum = torch.zeros((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    rt.render_distributed(scene, camera, frame, rng, epochs, accum=accum, ray_count=count)  # asynchronous: the oracle runs beside it
    st = _oracle.rng_init(frame)
    want = np.zeros((frame.rows, frame.cols, 3), dtype=np.float32)
    wcasts = 0
    desc = world.desc()
    for _ in range(epochs // 8):
        ws, wv, c = _oracle.render_distributed(desc, camera, frame, st, 8)
        wcasts += c
        for e in range(8):  # the filter, then img += photon, epoch by epoch
            want += np.where(wv[e][..., None] != 0, ws[e], np.float32(0))
        del ws, wv
    torch.cuda.synchronize()
    assert int(count.item()) == wcasts
    got = accum.cpu().numpy()
    bad = got.view(np.uint32) != want.view(np.uint32)
    assert not bad.any(), f"{bad.sum()} accumulator channels differ, first at {np.argwhere(bad)[:3].tolist()}"
    assert np.array_equal(rng.download(), st)


_EIGHTH = {}


@pytest.mark.parametrize("value", [0, 1])
@pytest.mark.parametrize("switch", ["RT_AMD_DIST_BY_COST", "RT_AMD_DIST_OWN_FIRST", "RT_AMD_DIST_PREP_FIRST"])
def test_a_real_eighth_share_with_each_small_share_switch_forced(ctx, organisation, switch, value):
    """What a rank of an 8-GPU job renders: rows 3, 11, 19, ... of the 1920 x 1080 depth-8 frame (259 200 pixels — the size the
    small-share organisation exists for: the chain kernel's pixels grouped by cost, a wave's first chunk its own, the look-ahead
    ahead of the shade kernel; include/rt_amd.h rt_set_option).  24 epochs, so that the grouping by cost has two batches of history
    behind it; each switch forced off and on.  Samples, flags, generator records, casts and the accumulator against the oracle."""
    if not organisation:
        pytest.skip("the switches belong to the split organisation")
    import torch

    world, camera, scene = ctx
    frame = rt.Frame.rows_of_rank(1920, 1080, 8, 3, 8)
    epochs = 24
    assert frame.rows * frame.cols == 259_200
    if not _EIGHTH:
        st = _oracle.rng_init(frame)
        ws, wv, wcasts = _oracle.render_distributed(world.desc(), camera, frame, st, epochs)
        want = np.zeros((frame.rows, frame.cols, 3), dtype=np.float32)
        for e in range(epochs):
            want += np.where(wv[e][..., None] != 0, ws[e], np.float32(0))
        _EIGHTH.update(st=st, ws=ws, wv=wv, wcasts=wcasts, want=want)
    accum = torch.zeros((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
    with rt.options(**{switch: value}):
        rng, s, v, casts = _run_gpu(scene, camera, frame, epochs, accum=accum)
    o = _EIGHTH
    assert np.array_equal(v, o["wv"]) and casts == o["wcasts"]
    assert ((s.view(np.uint32) == o["ws"].view(np.uint32)) | (np.isnan(s) & np.isnan(o["ws"]))).all()
    assert np.array_equal(accum.cpu().numpy().view(np.uint32), o["want"].view(np.uint32))
    assert np.array_equal(rng.download(), o["st"])


@pytest.mark.parametrize("refuse", ["1", "3", "99"])
def test_workspace_that_cannot_be_allocated_means_smaller_batches_then_one_kernel(ctx, organisation, refuse):
    """No device memory for the batch the cap allows: the batch is halved until it fits; not even one epoch fits: the
    one-kernel organisation renders the call.  Same samples either way (the hook makes the first n allocations fail)."""
    if not organisation:
        pytest.skip("the one-kernel organisation has no workspace")
    world, camera, _ = ctx
    scene = rt.Scene(world)  # a scene of its own: the workspace is kept per scene and stream, this one must be new
    frame = rt.Frame.full(50, 38, 6)
    rt.set_option("RT_AMD_DIAG_WS_REFUSE", refuse)
    try:
        rng, s, v, casts = _run_gpu(scene, camera, frame, 7)
    finally:
        rt.set_option("RT_AMD_DIAG_WS_REFUSE", None)
    st = _oracle.rng_init(frame)
    ws, wv, wcasts = _oracle.render_distributed(world.desc(), camera, frame, st, 7)
    assert np.array_equal(s.view(np.uint32), ws.view(np.uint32)) and np.array_equal(v, wv) and casts == wcasts
    assert np.array_equal(rng.download(), st)


def test_host_image_entry_point_is_the_reference_loop(ctx):
    """rt_render_distributed_host: img[at] = img[at] + photon for the surviving samples, epoch after epoch, continuing
    from the caller's image and the rt_rng's streams — the reference's loop with a host-resident `img`."""
    world, camera, scene = ctx
    frame = rt.Frame.full(64, 40, 5)
    rng = rt.Rng(frame)
    img = np.zeros((frame.rows, frame.cols, 3), dtype=np.float32)
    casts = rt.render_distributed_numpy(scene, camera, frame, rng, 3, img)
    casts += rt.render_distributed_numpy(scene, camera, frame, rng, 2, img)
    st = _oracle.rng_init(frame)
    ws, wv, wcasts = _oracle.render_distributed(world.desc(), camera, frame, st, 5)
    want = np.zeros_like(img)
    for e in range(5):
        want = np.where(wv[e][..., None] != 0, want + ws[e], want)
    assert np.array_equal(img.view(np.uint32), want.view(np.uint32)) and casts == wcasts
    assert np.array_equal(rng.download(), st)


def test_argument_validation(ctx):
    world, camera, scene = ctx
    frame = rt.Frame.full(32, 32, 5)
    rng = rt.Rng(frame)
    with pytest.raises(rt.RtError):
        rt.render_distributed(scene, camera, frame, rng, 1)  # neither accum nor samples
    import torch

    other = rt.Frame.full(16, 16, 5)
    acc = torch.zeros((16, 16, 3), dtype=torch.float32, device="cuda")
    with pytest.raises(rt.RtError):
        rt.render_distributed(scene, camera, other, rng, 1, accum=acc)  # RNG built for a different tile


@pytest.mark.parametrize("n_lights", [0, 1, 5])
def test_other_numbers_of_lights(n_lights):
    """The stochastic pass with none, one and five lights (the shade kernel keys its request lists on the first three)."""
    import torch
    import _scenes

    world, cam = _scenes.random_world(50 + n_lights, 30, 3, n_lights=n_lights), _scenes.camera(5)
    scene = rt.Scene(world)
    frame = rt.Frame.full(64, 48, 5)
    rng = rt.Rng(frame)
    samples = torch.empty((3, frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
    valid = torch.empty((3, frame.rows, frame.cols), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    rt.render_distributed(scene, cam, frame, rng, 3, focus=3.0, blur=0.04, samples=samples, valid=valid, ray_count=cnt)
    torch.cuda.synchronize()
    st = _oracle.rng_init(frame)
    ws, wv, wcasts = _oracle.render_distributed(world.desc(), cam, frame, st, 3, focus=3.0, blur=0.04)
    s = samples.cpu().numpy()
    same = (s.view(np.uint32) == ws.view(np.uint32)) | (np.isnan(s) & np.isnan(ws))
    assert same.all(), f"{(~same).sum()} channels differ"
    assert np.array_equal(valid.cpu().numpy(), wv) and int(cnt.item()) == wcasts
    assert np.array_equal(rng.download(), st)


def test_two_host_threads_run_pipelined_calls_on_their_own_streams(ctx):
    """rt_render_distributed from two HOST threads at once, each with its own generator and stream, calls of several batches
    each (two workspaces per stream used in turn, shade + unwind kernels on the generator's own second stream, the look-ahead on
    its third): both leave the accumulators, cast counts and generator records of the same calls made one after the other."""
    import threading

    import torch

    world, camera, scene = ctx
    frames = [rt.Frame.full(96, 72, 5), rt.Frame.full(80, 60, 6)]
    calls = (4, 3)
    rt.set_option("RT_AMD_DIST_WS_MB", 16)  # two epochs per batch at these sizes
    try:
        def run(frame, stream=None):
            rng = rt.Rng(frame)
            accum = torch.zeros((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
            cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
            if stream is not None:
                stream.wait_stream(torch.cuda.current_stream())
            for n in calls:
                rt.render_distributed(scene, camera, frame, rng, n, accum=accum, ray_count=cnt, stream=stream)
            (stream or torch.cuda.current_stream()).synchronize()
            torch.cuda.synchronize()
            return accum.cpu().numpy(), int(cnt.item()), rng.download()

        want = [run(f) for f in frames]
        got, errors = [None, None], []
        start = threading.Barrier(2)

        def worker(k):
            try:
                torch.cuda.set_device(0)
                stream = torch.cuda.Stream()
                start.wait()
                for _ in range(3):
                    got[k] = run(frames[k], stream)
            except Exception as exc:  # an exception in a thread must fail the test
                errors.append(exc)

        threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        rt.set_option("RT_AMD_DIST_WS_MB", None)
    assert not errors, errors
    for k in range(2):
        assert np.array_equal(got[k][0].view(np.uint32), want[k][0].view(np.uint32)) and got[k][1] == want[k][1]
        assert np.array_equal(got[k][2], want[k][2])
