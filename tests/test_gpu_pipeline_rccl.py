"""The N > 1 plumbing of bench.py on the GPU box's one GPU: dist.FramePipeline over RCCL with a single rank — the
asynchronous gather, the two alternating band buffers and the assembly on rank 0 are the code the 8-GPU run uses."""
import os
import socket

import numpy as np
import pytest

import homework_18_graphics_raytracer_amd as rt
from homework_18_graphics_raytracer_amd import dist as rtdist

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("in_flight", [1, 2, 4])
def test_pipelined_gather_over_rccl_single_rank(in_flight):
    import torch
    import torch.distributed as dist

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        world = rt.reference_world(); cam = rt.reference_camera(); scene = rt.Scene(world)
        w, h = 320, 203
        depths = [0, 2, 5, 1, 3, 8]
        pipe = rtdist.FramePipeline(w, h, max(depths), 0, 1, in_flight=in_flight)  # frame k on stream k % in_flight
        got = []
        for k, depth in enumerate(depths):
            with pipe.stream(k):
                band = pipe.band(k)
                rt.render_whitted(scene, cam, rt.Frame.full(w, h, depth), out=band)
                prev = pipe.submit(k)
                if prev is not None:
                    got.append(prev.clone())
        rest = []
        pipe.finish(into=rest)
        got += [f.clone() for f in rest]
        torch.cuda.synchronize()
        assert len(got) == len(depths)
        for k, depth in enumerate(depths):
            want = rt.render_whitted(scene, cam, rt.Frame.full(w, h, depth))
            torch.cuda.synchronize()
            assert torch.equal(got[k].view(torch.int32), want.view(torch.int32)), f"frame {k}"
    finally:
        dist.destroy_process_group()


def test_sharded_stochastic_pass_over_rccl_single_rank():
    """configs[3]'s plumbing (bench.py's stochastic_pass for N > 1) on one rank: accumulate_epochs_sharded — the rank's
    epochs on its row band, then the RCCL gather — against the oracle's sum of the same epochs, and the `y_step` bands
    of a 3-way split (rendered here one after the other on the one GPU) against the full frame's pixels."""
    import torch
    import torch.distributed as dist

    import _oracle

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        world = rt.reference_world(); cam = rt.reference_camera(); scene = rt.Scene(world)
        w, h, depth, epochs = 160, 99, 5, 5

        def render_epochs(frame):
            rng = rt.Rng(frame)
            acc = torch.zeros((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
            rt.render_distributed(scene, cam, frame, rng, epochs, accum=acc)
            return acc

        full = rtdist.accumulate_epochs_sharded(render_epochs, w, h, depth, 0, 1)
        torch.cuda.synchronize()
        frame = rt.Frame.full(w, h, depth)
        s_, v_, _ = _oracle.render_distributed(world.desc(), cam, frame, _oracle.rng_init(frame), epochs)
        want = np.zeros((h, w, 3), dtype=np.float32)
        for e in range(epochs):  # img[at] = img[at] + photon in epoch order (main.rs:1163-1167)
            want += np.where(v_[e][..., None] != 0, s_[e], np.float32(0))
        got = full.cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        # a pixel's stream is seeded by IMAGE coordinates: the bands of a 3-way split hold the same sums
        for r in range(3):
            band = render_epochs(rtdist.shard_frame(w, h, depth, r, 3)).cpu().numpy()
            assert np.array_equal(band.view(np.uint32), want[r::3].view(np.uint32)), f"band {r}"
    finally:
        dist.destroy_process_group()


def test_post_process_sharded_over_rccl_and_the_reference_pins():
    """dist.finish_frame_sharded on the GPU over RCCL (one rank): main()'s progressive loop with the p99 luma taken by the
    cross-rank radix select, each band normalised and encoded on the device, u8 rows gathered — (i) bit-identical to the
    single-device pipeline (rt_post_process_device + rt_encode_srgb8_device) after every epoch, and (ii) at 7 epochs the
    reference's own report/out.png, to the bar of tests/test_gpu_reference_pins.py (max |diff| 1, >= 99.999 % identical)."""
    import torch
    import torch.distributed as dist
    from PIL import Image
    from pathlib import Path

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        world = rt.reference_world(); cam = rt.reference_camera(); scene = rt.Scene(world)
        w, h, depth = 1280, 960, 5  # main.rs:1084-1085, 1098
        frame = rtdist.shard_frame(w, h, depth, 0, 1)
        img = rt.render_whitted(scene, cam, frame)
        ref = img.clone()
        u8, d = rtdist.finish_frame_sharded(img, h, 0, 1)
        rt.post_process_device(ref)
        assert torch.equal(img.view(torch.int32), ref.view(torch.int32)) and torch.equal(u8, rt.encode_srgb8_device(ref))
        rng, rng_ref = rt.Rng(frame), rt.Rng(frame)
        for k in range(1, 8):
            rt.render_distributed(scene, cam, frame, rng, 1, focus=3.0, blur=0.04, accum=img)
            rt.render_distributed(scene, cam, frame, rng_ref, 1, focus=3.0, blur=0.04, accum=ref)
            u8, d = rtdist.finish_frame_sharded(img, h, 0, 1)
            rt.post_process_device(ref)
            assert d > 0.0 and torch.equal(img.view(torch.int32), ref.view(torch.int32)), f"epoch {k}"
        torch.cuda.synchronize()
        assert torch.equal(u8, rt.encode_srgb8_device(ref))
        want = np.asarray(Image.open(Path(__file__).parent / "golden" / "ref_out_distributed.png").convert("RGB")).astype(np.int32)
        got = u8.cpu().numpy().astype(np.int32)
        assert got.shape == want.shape
        diff = np.abs(got - want)
        assert diff.max() <= 1 and np.mean(diff == 0) >= 0.99999, (diff.max(), float(np.mean(diff == 0)))
    finally:
        dist.destroy_process_group()
