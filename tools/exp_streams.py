#!/usr/bin/env python3
"""Experiment: frames of a share in flight on several streams.  Rank 0's share of an N-rank job renders in far less than 1/N of the
frame's time because so small a share leaves the GPU underfilled and ends on the critical path of its deepest pixels;
consecutive frames of a sequence are independent, so a rank can have S of them in flight, one per stream.

    python tools/exp_streams.py [--worlds 1 2 4 8] [--streams 1 2 3 4] [--frames 240]
"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import homework_18_graphics_raytracer_amd as rt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--worlds", type=int, nargs="+", default=[1, 2, 4, 8])
ap.add_argument("--streams", type=int, nargs="+", default=[1, 2, 3, 4])
ap.add_argument("--frames", type=int, default=240)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--depth", type=int, default=8)
ap.add_argument("--split-device", action="store_true", help="RT_AMD_WF_SHARE = S: each launch takes 1/S of the workgroups the device holds — the S kernels in flight side by side instead of one behind the tail of the other")
ap.add_argument("--graph", action="store_true", help="each stream's render call captured once in a HIP graph (torch.cuda.CUDAGraph) and replayed: what the host pays per frame is one graph launch")
a = ap.parse_args()
scene = rt.Scene(rt.reference_world())
cam = rt.reference_camera()
for world in a.worlds:
    frame = rt.Frame.full(a.width, a.height, a.depth) if world == 1 else rt.Frame.rows_of_rank(a.width, a.height, a.depth, 0, world)
    base = None
    for S in a.streams:
        streams = [torch.cuda.Stream() for _ in range(S)]
        rt.set_option("RT_AMD_WF_SHARE", S if a.split_device and S > 1 else None)
        outs = [torch.empty((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda") for _ in range(S)]

        graphs = None
        if a.graph:
            graphs = []
            for i in range(S):
                with torch.cuda.stream(streams[i]):
                    rt.render_whitted(scene, cam, frame, out=outs[i])  # the stream's workspace exists before the capture
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=streams[i]):
                    rt.render_whitted(scene, cam, frame, out=outs[i])
                graphs.append(g)

        def run(n):
            for k in range(n):
                with torch.cuda.stream(streams[k % S]):
                    if graphs is not None:
                        graphs[k % S].replay()
                    else:
                        rt.render_whitted(scene, cam, frame, out=outs[k % S])
            torch.cuda.synchronize()

        run(4 * S)
        t0 = time.perf_counter()
        run(a.frames)
        ms = (time.perf_counter() - t0) * 1e3 / a.frames
        if S == a.streams[0]:
            base = ms
        same = all(torch.equal(outs[0], o) for o in outs[1:])
        print(f"share 1/{world}{', graphs' if a.graph else ''}{', device split' if a.split_device and S > 1 else ''}: {S} frame(s) in flight: {ms:.4f} ms per frame ({base / ms:.2f}x one in flight), frames identical: {same}", flush=True)
