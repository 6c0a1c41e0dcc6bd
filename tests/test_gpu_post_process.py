"""Device post_process + sRGB encode against the host restatement and the oracle, bit for bit; and the whole
device pipeline (render -> post_process -> encode) against the reference's own output image."""
import numpy as np
import pytest
from PIL import Image

import homework_18_graphics_raytracer_amd as rt
import _oracle

pytestmark = pytest.mark.gpu


def _dev(img_np):
    import torch

    t = torch.from_numpy(img_np.copy()).cuda()
    div = torch.zeros(1, dtype=torch.float32, device="cuda")
    rt.post_process_device(t, divisor=div)
    u8 = rt.encode_srgb8_device(t)
    torch.cuda.synchronize()
    return t.cpu().numpy(), float(div.item()), u8.cpu().numpy()


def test_matches_oracle_on_a_rendered_frame():
    world = rt.reference_world()
    img, _ = _oracle.render_whitted(world.desc(), rt.reference_camera(), rt.Frame.full(320, 240, 5))
    got, div, u8 = _dev(img)
    want = img.copy()
    wdiv = _oracle.post_process(want, 0)
    assert div == wdiv and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(u8, _oracle.encode_srgb8(want))


def test_edge_cases_match_host():
    rng = np.random.default_rng(3)
    cases = [
        np.zeros((8, 8, 3), np.float32),                                  # no normal luma: untouched (reference panics)
        np.full((8, 8, 3), 1e-9, np.float32),                             # percentile <= EPSILON: untouched
        rng.uniform(0, 4, (61, 47, 3)).astype(np.float32),               # ragged size
        np.where(rng.random((50, 50, 3)) < 0.2, np.nan, rng.uniform(0, 2, (50, 50, 3))).astype(np.float32),  # NaN lumas dropped
        (rng.uniform(-1, 1, (40, 40, 3))).astype(np.float32),            # negative lumas take part in the ranking
        np.repeat(rng.uniform(0, 1, (1, 1, 3)), 100, axis=0).reshape(10, 10, 3).astype(np.float32),  # all equal
    ]
    for img in cases:
        got, div, u8 = _dev(img)
        want = img.copy()
        wdiv = rt.post_process(want)
        assert div == wdiv
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all()
        assert np.array_equal(u8, rt.encode_srgb8(want))


def test_device_pipeline_reproduces_the_reference_image():
    """render (HIP) -> post_process (HIP) -> sRGB/u8 (HIP) at the reference's 1280x960 depth 5 vs report/out_single_epoch.png."""
    import torch

    world = rt.reference_world()
    scene = rt.Scene(world)
    frame = rt.Frame.full(1280, 960, 5)
    img = rt.render_whitted(scene, rt.reference_camera(), frame)
    rt.post_process_device(img)
    u8 = rt.encode_srgb8_device(img)
    torch.cuda.synchronize()
    ref = np.asarray(Image.open(_oracle.GOLDEN / "ref_out_single_epoch.png").convert("RGB")).astype(np.int32)
    diff = np.abs(u8.cpu().numpy().astype(np.int32) - ref)
    assert diff.max() <= 1 and np.mean(diff == 0) >= 0.9999


def test_passes_over_two_bands_equal_the_single_device_select():
    """rt_post_keys/hist/pick/scale_device (include/rt_amd.h): a frame cut into two interleaved row bands, each with its own keys
    and state, the states summed between the passes the way dist.post_process_sharded's all-reduces do — every band ends with the
    divisor rt_post_process_device finds for the whole frame, and the same pixels.  Edge cases: NaN lumas, an empty band."""
    import torch

    from homework_18_graphics_raytracer_amd import dist as rtdist

    rng = np.random.default_rng(11)
    world = rt.reference_world()
    rendered, _ = _oracle.render_whitted(world.desc(), rt.reference_camera(), rt.Frame.full(200, 150, 5))
    noisy = np.where(rng.random((90, 70, 3)) < 0.1, np.nan, rng.uniform(-0.5, 3, (90, 70, 3))).astype(np.float32)
    for img, parts in ((rendered, 2), (noisy, 3), (rendered[:1], 2)):  # the last: one row over two "ranks" — the second band is empty
        want, wdiv, _ = _dev(img)
        bands = [torch.from_numpy(np.ascontiguousarray(img[r::parts])).cuda() for r in range(parts)]
        passes = [rtdist._PostPassesHip(b) for b in bands]
        for p in passes:
            p.do_keys()
        total = sum(p.count for p in passes)
        for p in passes:
            p.count.copy_(total)
        for k in range(4):
            for p in passes:
                p.do_hist(k)
            hist = sum(p.hist for p in passes)
            for p in passes:
                p.hist.copy_(hist)
                p.do_pick(k)
        divs = [float(p.do_scale().item()) for p in passes]
        torch.cuda.synchronize()
        assert all(d == wdiv for d in divs), (divs, wdiv)
        for r, b in enumerate(bands):
            got, ref = b.cpu().numpy(), want[r::parts]
            assert ((got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))).all()


def test_sharded_post_process_on_the_device_does_not_wait_for_the_host():
    """dist.post_process_sharded(sync=False) on a CUDA band: same divisor and pixels as rt_post_process_device, the divisor left
    in device memory (one process, no process group: the passes without the sums)."""
    import torch

    from homework_18_graphics_raytracer_amd import dist as rtdist

    world = rt.reference_world()
    img, _ = _oracle.render_whitted(world.desc(), rt.reference_camera(), rt.Frame.full(320, 240, 5))
    want, wdiv, wu8 = _dev(img)
    band = torch.from_numpy(img.copy()).cuda()
    u8, d = rtdist.finish_frame_sharded(band, 240, 0, 1, sync=False)
    assert torch.is_tensor(d) and d.is_cuda
    torch.cuda.synchronize()
    assert float(d.item()) == wdiv and np.array_equal(band.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert np.array_equal(u8.cpu().numpy(), wu8)
