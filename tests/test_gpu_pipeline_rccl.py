"""The N > 1 plumbing of bench.py on the GPU box's one GPU: dist.FramePipeline over RCCL with a single rank — the
asynchronous gather, the two alternating band buffers and the assembly on rank 0 are the code the 8-GPU run uses."""
import os
import socket

import numpy as np
import pytest

import homework_18_graphics_raytracer_amd as rt
from homework_18_graphics_raytracer_amd import dist as rtdist

pytestmark = pytest.mark.gpu


def test_pipelined_gather_over_rccl_single_rank():
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
        pipe = rtdist.FramePipeline(w, h, max(depths), 0, 1)
        got = []
        for k, depth in enumerate(depths):
            band = pipe.band(k)
            rt.render_whitted(scene, cam, rt.Frame.full(w, h, depth), out=band)
            prev = pipe.submit(k)
            if k >= 1:
                got.append(prev.clone())
        got.append(pipe.finish().clone())
        torch.cuda.synchronize()
        for k, depth in enumerate(depths):
            want = rt.render_whitted(scene, cam, rt.Frame.full(w, h, depth))
            torch.cuda.synchronize()
            assert torch.equal(got[k].view(torch.int32), want.view(torch.int32)), f"frame {k}"
    finally:
        dist.destroy_process_group()
