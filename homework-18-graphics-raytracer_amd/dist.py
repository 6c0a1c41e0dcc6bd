"""Image-tile sharding over the GPUs of one node.

Pixels are independent (the reference's par_iter over (y, x), src/main.rs:1090), the scene is a few KB
and is replicated on every rank, so the path shards with NO data-path collective: rank r renders the
interleaved row band {r, r+N, r+2N, ...} (the cost per pixel is spatially non-uniform — background
pixels cost one cast, glass pixels dozens — and interleaving rows balances it).  The only exchange is
assembling the framebuffer on rank 0: one gather of the f32 RGB bands (RCCL over xGMI on GPUs, gloo in
the CPU tests), then a de-interleave.

One process per GPU; `backend="nccl"` is RCCL on ROCm.
"""
from __future__ import annotations

from typing import Callable, Optional

from ._capi import Frame


def shard_frame(width: int, height: int, max_depth: int, rank: int, world: int) -> Frame:
    """The tile of `rank`: all columns, rows rank, rank+world, ... (an rt_frame with y0=rank, y_step=world)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    if world > height:
        raise ValueError("more ranks than image rows")
    return Frame.rows_of_rank(width, height, max_depth, rank, world)


def band_rows(height: int, rank: int, world: int) -> int:
    return (height - rank + world - 1) // world


def gather_frame(band, height: int, rank: int, world: int, dst: int = 0, group=None, staging=None):
    """Assemble the full (height, width, 3) image on `dst` from every rank's row band.

    `band`: this rank's (band_rows, width, 3) float32 tensor (CPU for gloo, CUDA for RCCL).
    Returns the full image on `dst`, None elsewhere.  `staging` (optional, dst only) is a reusable
    (world, max_rows, width, 3) buffer so that steady-state frames allocate nothing.
    """
    import torch
    import torch.distributed as dist

    if world == 1 and not dist.is_initialized():
        return band
    width = band.shape[1]
    max_rows = band_rows(height, 0, world)
    if band.shape[0] != max_rows:  # ragged last bands: pad to the common size for the collective
        padded = band.new_zeros((max_rows, width, 3))
        padded[: band.shape[0]] = band
        band = padded
    if rank == dst:
        if staging is None:
            staging = band.new_empty((world, max_rows, width, 3))
        dist.gather(band, [staging[r] for r in range(world)], dst=dst, group=group)
        # de-interleave: image row y = k*world + r  <-  staging[r, k]
        full = staging.permute(1, 0, 2, 3).reshape(max_rows * world, width, 3)
        return full[:height].contiguous() if max_rows * world != height else full.contiguous()
    dist.gather(band, None, dst=dst, group=group)
    return None


def render_frame_sharded(render_band: Callable[[Frame], "object"], width: int, height: int, max_depth: int, rank: int,
                         world: int, dst: int = 0, group=None, staging=None):
    """Render this rank's band with `render_band(frame) -> tensor` and gather the frame on `dst`.

    `render_band` is the HIP path in production (homework_18_graphics_raytracer_amd.render_whitted);
    the CPU/gloo tests inject the oracle here, which is the only reason it is a parameter.
    """
    frame = shard_frame(width, height, max_depth, rank, world)
    band = render_band(frame)
    return gather_frame(band, height, rank, world, dst=dst, group=group, staging=staging)


def accumulate_epochs_sharded(render_epochs: Callable[[Frame], "object"], width: int, height: int, max_depth: int, rank: int,
                              world: int, dst: int = 0, group=None, staging=None):
    """The distributed (stochastic) pass over N ranks — BASELINE.json configs[3]/[4].

    Pixels own their RNG streams (seed y*2^33 + x, src/main.rs:1119), so a rank keeps the IsaacRng states of
    its row band for all epochs and nothing but the accumulated band ever moves: `render_epochs(frame)` runs
    any number of epochs on this rank's band (rt.render_distributed with accum=..., RNG created for `frame`)
    and returns the (band_rows, width, 3) sum; the bands are gathered to `dst` exactly like a Whitted frame.
    The result equals the single-process sum bit for bit because every pixel's samples are added in the same
    epoch order regardless of which rank owns it.
    """
    frame = shard_frame(width, height, max_depth, rank, world)
    return gather_frame(render_epochs(frame), height, rank, world, dst=dst, group=group, staging=staging)


class _PostPassesHip:
    """The passes of post_process on THIS rank's band as stream-ordered device work: rt_post_keys_device / rt_post_hist_device /
    rt_post_pick_device / rt_post_scale_device (include/rt_amd.h, csrc/rt_post.hip) on the current stream.  `count` and `hist`
    are views of the device-resident state, to be summed over the ranks in place."""

    def __init__(self, band):
        import ctypes as C

        import torch

        from . import _capi

        assert band.is_cuda and band.dtype == torch.float32 and band.is_contiguous() and band.shape[-1] == 3
        self._C, self._lib, self._check = C, _capi.amd_lib(), _capi.check
        self.band, self.n = band, band.numel() // 3
        self.keys = torch.empty(max(self.n, 1), dtype=torch.int32, device=band.device)
        self.state = torch.empty(260, dtype=torch.int32, device=band.device)  # RT_POST_STATE_WORDS
        self.divisor = torch.empty(1, dtype=torch.float32, device=band.device)
        self.count, self.hist = self.state[0:1], self.state[4:260]

    def _stream(self):
        import torch

        return self._C.c_void_p(torch.cuda.current_stream(self.band.device).cuda_stream)

    def _p(self, t):
        return self._C.c_void_p(t.data_ptr())

    def do_keys(self):
        self._check(self._lib.rt_post_keys_device(self._p(self.band), self.n, self._p(self.keys), self._p(self.state), self._stream()))

    def do_hist(self, pass_):
        self._check(self._lib.rt_post_hist_device(self._p(self.keys), self.n, pass_, self._p(self.state), self._stream()))

    def do_pick(self, pass_):
        self._check(self._lib.rt_post_pick_device(pass_, self._p(self.state), self._stream()))

    def do_scale(self):
        self._check(self._lib.rt_post_scale_device(self._p(self.band), self.n, self._p(self.state), self._p(self.divisor), self._stream()))
        return self.divisor


class _PostPassesCpu:
    """The CPU twin of the same passes for a host-resident band (the gloo tests; a host that keeps its image in memory): the state
    words mean what csrc/rt_post.hip's mean — [0] count, [1] rank wanted, [2] digits chosen, [4..259] histogram."""

    def __init__(self, band):
        import torch

        from . import luma_row

        assert not band.is_cuda and band.dtype == torch.float32 and band.shape[-1] == 3
        self.band = band
        self.row = [torch.tensor(v, dtype=torch.float32) for v in luma_row()]
        self.state = torch.zeros(260, dtype=torch.int64)
        self.count, self.hist = self.state[0:1], self.state[4:260]
        self.keys = None

    def do_keys(self):
        import torch

        b, (w0, w1, w2) = self.band, self.row
        luma = (b[..., 0] * w0 + b[..., 1] * w1) + b[..., 2] * w2  # three separate roundings, in the reference's order
        bits = luma.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        mag = bits & 0x7FFFFFFF
        normal = (mag >= 0x00800000) & (mag < 0x7F800000)  # f32::is_normal (main.rs:751)
        self.keys = torch.where((bits & 0x80000000) != 0, (~bits) & 0xFFFFFFFF, bits | 0x80000000)[normal]  # monotone in the float's value
        self.state.zero_()
        self.state[0] = self.keys.numel()

    def do_hist(self, pass_):
        import numpy as np
        import torch

        if pass_ == 0:  # (len as f32 * 0.99) as usize, main.rs:754 — of the count over ALL bands
            n = int(self.state[0])
            self.state[1] = min(int(np.float32(n) * np.float32(0.99)), max(n - 1, 0))
            self.state[2] = 0
        shift = 24 - 8 * pass_
        prefix = int(self.state[2])
        sel = self.keys if pass_ == 0 else self.keys[(self.keys >> (shift + 8)) == (prefix >> (shift + 8))]
        self.hist += torch.bincount((sel >> shift) & 0xFF, minlength=256)

    def do_pick(self, pass_):
        if int(self.state[0]) == 0:
            return
        k, digit = int(self.state[1]), 255
        for d, c in enumerate(self.hist.tolist()):
            if k < c:
                digit = d
                break
            k -= c
        self.state[1] = k
        self.state[2] = int(self.state[2]) | (digit << (24 - 8 * pass_))
        self.hist.zero_()

    def do_scale(self):
        import numpy as np
        import torch

        zero = torch.zeros(1, dtype=torch.float32)
        if int(self.state[0]) == 0:
            return zero
        key = int(self.state[2])
        raw = (key & 0x7FFFFFFF) if key & 0x80000000 else (~key) & 0xFFFFFFFF
        p98 = np.array([raw], dtype=np.uint32).view(np.float32)[0]
        if not (p98 > np.float32(1.1920928955078125e-7)):  # main.rs:755
            return zero
        # a one-element TENSOR divisor: a Python scalar would let torch multiply by the reciprocal (one rounding more than the division)
        self.band.div_(torch.full((1, 1, 1), float(p98), dtype=torch.float32))
        return torch.full((1,), float(p98), dtype=torch.float32)


def post_process_sharded(band, group=None, sync: bool = True):
    """`post_process` (src/main.rs:748-762) of a frame whose row bands live on the ranks of `group`: divides THIS rank's `band`
    (rows, width, 3 float32; CPU for gloo, CUDA for RCCL) in place by the 99th-percentile luma of ALL ranks' normal lumas and
    returns the divisor (0.0: the frame was left untouched, as the reference does when the percentile is <= f32::EPSILON or no
    luma is normal).  SURVEY §8(f-1)'s purpose: the frame never has to be assembled as f32 — after this every rank encodes its
    own band (encode_srgb8_band) and the gather moves u8, 6.2 MB at 1080p instead of 24.9.

    The reference sorts the lumas and indexes one element; the k-th smallest does not depend on the order, so it is found by an
    exact radix select over order-preserving 32-bit keys whose four 256-bin histograms (and, before them, the count) are summed
    over the ranks.  Every rank sees the same sums, picks the same digits, ends with the same key: the value main.rs:754 indexes.

    A CUDA band never leaves the device and nothing here waits for it: keys -> all_reduce(count) -> 4 x (hist -> all_reduce ->
    pick) -> scale are kernels of csrc/rt_post.hip on the current stream with RCCL's all-reduces ordered between them, the count,
    the histogram and the chosen digits in device memory throughout.  `sync=False` returns the divisor as a one-element tensor
    where the band lives (no host round trip at all); the default reads it (the one synchronisation, when the divisor is wanted)."""
    import torch.distributed as dist

    on = dist.is_available() and dist.is_initialized()
    passes = _PostPassesHip(band) if band.is_cuda else _PostPassesCpu(band)
    passes.do_keys()
    if on:
        dist.all_reduce(passes.count, group=group)
    for p in range(4):
        passes.do_hist(p)
        if on:
            dist.all_reduce(passes.hist, group=group)
        passes.do_pick(p)
    divisor = passes.do_scale()
    return float(divisor.item()) if sync else divisor


def encode_srgb8_band(band):
    """Linear f32 -> sRGB u8 (src/image.rs:55-66) of a band, where it lives: the HIP kernel for a CUDA tensor
    (rt_encode_srgb8_device), librt_host's loop for a CPU tensor — the same function of one value either way."""
    import torch

    from . import encode_srgb8, encode_srgb8_device

    if band.is_cuda:
        return encode_srgb8_device(band.contiguous())
    return torch.from_numpy(encode_srgb8(band.contiguous().numpy()))


def finish_frame_sharded(band, height: int, rank: int, world: int, dst: int = 0, group=None, staging=None, sync: bool = True):
    """What main() does with a frame after rendering it (main.rs:1113-1114, 1171-1172), sharded: post_process over the ranks'
    bands, sRGB/u8 encode of each band where it is, then ONE gather of u8 rows to `dst`.  Returns (u8 frame on dst | None, divisor);
    `band` is left normalised, as the reference leaves `img` — the next epoch accumulates into it.  `sync=False`: the divisor stays
    a one-element tensor and, for a CUDA band, the whole step is enqueued without the host waiting for any of it."""
    divisor = post_process_sharded(band, group=group, sync=sync)
    return gather_frame(encode_srgb8_band(band), height, rank, world, dst=dst, group=group, staging=staging), divisor


def choose_streams(render, in_flight: int, attempts: int = 4, frames_per_stream: int = 6):
    """`in_flight` torch streams on which frames in flight actually run side by side.

    A process's HIP streams share a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and which queues run concurrently is the
    driver's business, not the caller's: of the sets of four streams a process makes, about one in five renders four frames in flight
    SLOWER than one after the other (profiles/r04_frames_in_flight.txt, tools/exp_queue_mapping.py) — costly when each launch has been
    given a quarter of the device.  So: make a set, time a few frames on it (`render(i)` issues one frame on the current stream into
    buffer i of `in_flight`), keep the best of `attempts` sets.  Returns (streams, ms_per_frame of every attempt)."""
    import time

    import torch

    best, best_ms, seen = None, None, []
    for _ in range(max(1, attempts)):
        streams = [torch.cuda.Stream() for _ in range(in_flight)]

        def run(n):
            for k in range(n):
                with torch.cuda.stream(streams[k % in_flight]):
                    render(k % in_flight)
            torch.cuda.synchronize()

        run(2 * in_flight)  # each stream's workspace exists
        n = frames_per_stream * in_flight
        t0 = time.perf_counter()
        run(n)
        ms = (time.perf_counter() - t0) * 1e3 / n
        seen.append(round(ms, 4))
        if best_ms is None or ms < best_ms:
            best, best_ms = streams, ms
    return best, seen


class FramePipeline:
    """A sequence of frames with the gather of frame k overlapped with the rendering of the frames behind it.

    A rank's share of a 1080p frame renders in ~0.5 ms; the gather to rank 0 (RCCL over xGMI, which runs on its own
    stream) and the de-interleave cost a comparable time, so doing them back to back halves the frame rate.  Here a
    rank renders into one of its band buffers; the gather of that band is started asynchronously, and only before
    the buffer is written again — or when the frame is collected on rank 0 — does a compute stream wait for it.
    Nothing is allocated per frame.

    in_flight = S > 1: the frames of a sequence are independent, and a 1/N share of a frame leaves the GPU mostly idle
    while it ends on the critical path of its deepest pixels (a 1/8 share of the 1080p frame renders in
    0.33 ms alone, 0.17 ms per frame with four in flight).  Frame k is then rendered on stream k % S — `with
    pipe.stream(k):` around the render and the submit — into one of 2 S band buffers, and submit(k) hands back frame
    k - S, whose render and gather were issued on the same stream S frames ago.

        pipe = FramePipeline(width, height, max_depth, rank, world, in_flight=S)
        for k in range(n):
            with pipe.stream(k):                    # the current stream when S == 1
                band = pipe.band(k)                 # (rows, width, 3) view to render frame k's share into
                render(pipe.frame, band)            # stream-ordered on the current stream
                full = pipe.submit(k)               # starts frame k's gather; returns frame k-S assembled (rank 0), or None
        last = pipe.finish()                        # frame n-1 assembled (rank 0); the ones before it have been assembled too
    """

    def __init__(self, width: int, height: int, max_depth: int, rank: int, world: int, dst: int = 0, group=None, device="cuda", in_flight: int = 1,
                 streams=None):
        import torch

        self.frame = shard_frame(width, height, max_depth, rank, world)
        self.height, self.rank, self.world, self.dst, self.group = height, rank, world, dst, group
        self.in_flight = max(1, int(in_flight))
        self.slots = 2 * self.in_flight
        self.max_rows = band_rows(height, 0, world)
        # padded to the common band size so that ragged last bands need no per-frame copy
        self._bands = [torch.zeros((self.max_rows, width, 3), dtype=torch.float32, device=device) for _ in range(self.slots)]
        self._staging = [torch.empty((world, self.max_rows, width, 3), dtype=torch.float32, device=device) for _ in range(self.slots)] if rank == dst else None
        self._work = [None] * self.slots
        on_gpu = torch.device(device).type == "cuda"
        self._streams = None
        if self.in_flight > 1 and on_gpu:  # the caller's (choose_streams), or fresh ones
            self._streams = list(streams) if streams is not None else [torch.cuda.Stream() for _ in range(self.in_flight)]
            assert len(self._streams) == self.in_flight
        self._last = -1        # the last frame submitted
        self._assembled = -1   # the last frame assembled

    def stream(self, k: int):
        """The context frame k is rendered and submitted in: its stream of the S, or nothing to enter."""
        import contextlib

        import torch

        return torch.cuda.stream(self._streams[k % self.in_flight]) if self._streams is not None else contextlib.nullcontext()

    def band(self, k: int):
        slot = k % self.slots
        if self._work[slot] is not None:  # the gather that read this buffer 2 S frames ago
            self._work[slot].wait()
            self._work[slot] = None
        return self._bands[slot][: self.frame.rows]

    def _assemble(self, k: int):
        slot = k % self.slots
        if self._work[slot] is not None:
            self._work[slot].wait()
            self._work[slot] = None
        self._assembled = k
        if self.rank != self.dst:
            return None
        st = self._staging[slot]
        full = st.permute(1, 0, 2, 3).reshape(self.max_rows * self.world, st.shape[2], 3)  # image row y = k*world + r
        return full[: self.height].contiguous() if self.max_rows * self.world != self.height else full.contiguous()

    def submit(self, k: int):
        import torch.distributed as dist

        slot = k % self.slots
        if self.rank == self.dst:
            self._work[slot] = dist.gather(self._bands[slot], [self._staging[slot][r] for r in range(self.world)], dst=self.dst,
                                           group=self.group, async_op=True)
        else:
            self._work[slot] = dist.gather(self._bands[slot], None, dst=self.dst, group=self.group, async_op=True)
        prev = self._assemble(k - self.in_flight) if k - self.in_flight > self._assembled else None  # (not again after a finish())
        self._last = k
        return prev

    def finish(self, into=None):
        """Every frame submitted and not yet handed back, assembled in order; returns the last one (rank 0) and appends all
        of them to `into` if that is a list."""
        out = None
        for k in range(self._assembled + 1, self._last + 1):  # each on the stream it was rendered on
            with self.stream(k):
                out = self._assemble(k)
                if into is not None:
                    into.append(out)
        if self._streams is not None:
            import torch

            for st in self._streams:  # whoever called us goes on on its own stream
                torch.cuda.current_stream().wait_stream(st)
        for slot in range(self.slots):
            if self._work[slot] is not None:
                self._work[slot].wait()
                self._work[slot] = None
        return out
