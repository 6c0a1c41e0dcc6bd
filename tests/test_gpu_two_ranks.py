"""HIP on TWO ranks (the gloo tests inject the oracle as the renderer — they test the sharding, not the kernels on several
ranks).  The GPU box has one GPU, and RCCL refuses two ranks on one device, so here two PROCESSES share cuda:0: each renders its
interleaved row band with the real kernels through the C ABI (Whitted, then depth-of-field epochs with the generators of its own
rows), the collectives — the f32 gather, and the five all-reduces + u8 gather of dist.finish_frame_sharded — run over gloo on host
copies of the bands.  The assembled frames must equal the oracle's single-process frame bit for bit, and main()'s progressive loop
over the two ranks (post_process between the epochs) must equal the oracle's."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, w, h, depth, epochs, out_path):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import homework_18_graphics_raytracer_amd as rt
    from homework_18_graphics_raytracer_amd import dist as rtdist

    torch.cuda.set_device(0)
    scene = rt.Scene(rt.reference_world())
    cam = rt.reference_camera()
    frame = rtdist.shard_frame(w, h, depth, rank, world)
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    band = rt.render_whitted(scene, cam, frame, ray_count=count)  # the HIP kernel, this rank's rows
    torch.cuda.synchronize()
    whitted = rtdist.gather_frame(band.cpu(), h, rank, world)  # f32 bands -> rank 0
    # main()'s progressive loop over the ranks: post_process of the sharded frame, then epoch by epoch accumulate + post_process
    img = band.cpu().clone()
    frames_u8, divisors = [], []
    u8, d = rtdist.finish_frame_sharded(img, h, rank, world)
    frames_u8.append(u8)
    divisors.append(d)
    rng = rt.Rng(frame)
    for _ in range(epochs):
        dev = img.cuda()
        rt.render_distributed(scene, cam, frame, rng, 1, accum=dev, ray_count=count)  # img += photon on the device
        torch.cuda.synchronize()
        img = dev.cpu()
        u8, d = rtdist.finish_frame_sharded(img, h, rank, world)
        frames_u8.append(u8)
        divisors.append(d)
    casts = count.cpu()
    dist.all_reduce(casts)
    if rank == 0:
        np.savez(out_path, whitted=whitted.numpy(), u8=np.stack([f.numpy() for f in frames_u8]), divisors=np.array(divisors, dtype=np.float32),
                 casts=casts.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h", [(2, 96, 70), (3, 64, 47)])
def test_hip_kernels_on_several_ranks_equal_the_single_process_loop(tmp_path, world, w, h):
    sys.path.insert(0, str(ROOT / "tests"))
    import homework_18_graphics_raytracer_amd as rt
    import _oracle

    depth, epochs = 5, 2
    out = tmp_path / "ranks.npz"
    mp.spawn(_worker, args=(world, _free_port(), w, h, depth, epochs, str(out)), nprocs=world, join=True)
    got = np.load(out)
    desc, cam, frame = rt.reference_world().desc(), rt.reference_camera(), rt.Frame.full(w, h, depth)
    want, wcasts = _oracle.render_whitted(desc, cam, frame)
    assert np.array_equal(got["whitted"].view(np.uint32), want.view(np.uint32))
    img = want.copy()
    st = _oracle.rng_init(frame)
    d = _oracle.post_process(img)
    assert got["divisors"][0] == np.float32(d) and np.array_equal(got["u8"][0], _oracle.encode_srgb8(img))
    for e in range(epochs):
        s, v, c = _oracle.render_distributed(desc, cam, frame, st, 1)
        img += np.where(v[0][..., None] != 0, s[0], np.float32(0))
        wcasts += c
        d = _oracle.post_process(img)
        assert got["divisors"][1 + e] == np.float32(d)
        assert np.array_equal(got["u8"][1 + e], _oracle.encode_srgb8(img))
    assert int(got["casts"][0]) == wcasts
