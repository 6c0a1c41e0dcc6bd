#!/usr/bin/env python3
"""Experiment: does it matter which HIP streams the frames in flight run on?  (A process has GPU_MAX_HW_QUEUES hardware queues, 4 by
default; streams share them.)  Rank 0's 1/2 and 1/8 share, four in flight on quarters of the device, after creating `skip` other
streams first."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import homework_18_graphics_raytracer_amd as rt  # noqa: E402

scene = rt.Scene(rt.reference_world())
cam = rt.reference_camera()
keep = []
for skip in (0, 1, 2, 3, 5):
    while len(keep) < skip:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            torch.zeros(1, device="cuda")
        keep.append(s)
    for world in (2, 8):
        frame = rt.Frame.rows_of_rank(1920, 1080, 8, 0, world)
        S = 4
        outs = [torch.empty((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda") for _ in range(S)]
        rt.set_option("RT_AMD_WF_SHARE", S)
        tried = None
        if len(sys.argv) > 1 and sys.argv[1] == "--choose":  # dist.choose_streams: the best of four sets, by a short timing
            from homework_18_graphics_raytracer_amd import dist as rtdist
            streams, tried = rtdist.choose_streams(lambda i: rt.render_whitted(scene, cam, frame, out=outs[i]), S)
        else:
            streams = [torch.cuda.Stream() for _ in range(S)]

        def run(n):
            for k in range(n):
                with torch.cuda.stream(streams[k % S]):
                    rt.render_whitted(scene, cam, frame, out=outs[k % S])
            torch.cuda.synchronize()

        run(8)
        res = []
        for n in (20, 200):
            t0 = time.perf_counter()
            run(n)
            res.append((time.perf_counter() - t0) * 1e3 / n)
        rt.set_option("RT_AMD_WF_SHARE", None)
        print(f"{skip} streams made before: share 1/{world}, four in flight: {res[0]:.4f} ms per frame over 20 frames, {res[1]:.4f} over 200" + (f"; sets tried: {tried}" if tried else ""), flush=True)
