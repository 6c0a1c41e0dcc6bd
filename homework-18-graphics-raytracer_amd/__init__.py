"""MI355X-native render path for the homework-18 raytracer — Python host mirror.

The product is the C-ABI library ``librt_amd.so`` (hand-written HIP for gfx950,
``csrc/rt_kernels.hip``) plus ``librt_host.so`` (scene build / OBJ import /
post_process / PNG, ``csrc/host/rt_host.cpp``).  This package only mirrors the
reference's host-side names on top of them:

    World / ObjectProxy            src/main.rs:130-178, 700-728
    Camera                         src/main.rs:43-49
    render (the Whitted par_iter)  src/main.rs:1087-1104
    post_process / write_to_file   src/main.rs:748-776

PyTorch is used only for device memory, streams and torch.distributed.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

from . import _capi
from ._capi import Camera, Frame, Light, Material, RtError, SceneDesc, Sphere, Triangle, Vertex

__all__ = [
    "World", "ObjectProxy", "Scene", "Camera", "Frame", "Material", "Light", "RtError", "reference_world",
    "reference_camera", "render_whitted", "render_whitted_numpy", "Rng", "render_distributed", "render_distributed_numpy", "set_option", "options", "post_process_device", "encode_srgb8_device", "post_process", "encode_srgb8", "write_to_file",
    "DEFAULT_OBJ",
]

DEFAULT_OBJ = str(_capi.REPO_ROOT / "tests" / "golden" / "dodecahedron.obj")


def _f3(v: Sequence[float]):
    return (C.c_float * 3)(*[float(x) for x in v])


class ObjectProxy:
    """src/main.rs:700-728."""

    def __init__(self, world: "World", object_index: int):
        self.world = world
        self.object_index = object_index

    def push_triangle(self, vertices: Sequence[Vertex]) -> "ObjectProxy":
        arr = (Vertex * 3)(*vertices)
        _capi.check_host(_capi.host_lib().rt_world_push_triangle(self.world._h, self.object_index, arr))
        return self

    def push_triangles(self, triangles: Sequence[Sequence[Vertex]]) -> "ObjectProxy":
        for t in triangles:
            self.push_triangle(t)
        return self

    def push_sphere(self, center: Sequence[float], radius: float) -> "ObjectProxy":
        _capi.check_host(_capi.host_lib().rt_world_push_sphere(self.world._h, self.object_index, _f3(center), float(radius)))
        return self

    def push_flat_triangle(self, positions: Sequence[Sequence[float]], uvs: Sequence[Sequence[float]]) -> "ObjectProxy":
        """triangle(), src/main.rs:730-739."""
        p = (C.c_float * 9)(*[float(x) for v in positions for x in v])
        uv = (C.c_float * 6)(*[float(x) for v in uvs for x in v])
        _capi.check_host(_capi.host_lib().rt_world_push_flat_triangle(self.world._h, self.object_index, p, uv))
        return self

    def push_square(self, positions: Sequence[Sequence[float]], uvs: Sequence[Sequence[float]]) -> "ObjectProxy":
        """square(), src/main.rs:741-746."""
        p = (C.c_float * 12)(*[float(x) for v in positions for x in v])
        uv = (C.c_float * 8)(*[float(x) for v in uvs for x in v])
        _capi.check_host(_capi.host_lib().rt_world_push_square(self.world._h, self.object_index, p, uv))
        return self

    def load_obj(self, path: str, divisor: float = 3.0, offset: Sequence[float] = (0.7, 1.0, -0.5)) -> int:
        """load_obj, src/main.rs:778-807.  Returns the number of triangles pushed."""
        return _capi.check_host(
            _capi.host_lib().rt_world_load_obj(self.world._h, self.object_index, str(path).encode(), float(divisor), _f3(offset))
        )


class World:
    """Host-side scene under construction; src/main.rs:130-178."""

    def __init__(self):
        lib = _capi.host_lib()
        self._free = lib.rt_world_free  # bound now: module globals may be gone at interpreter shutdown
        self._h = lib.rt_world_new()
        if not self._h:
            raise MemoryError("rt_world_new failed")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._free(h)

    def push_object(self, material: Material) -> ObjectProxy:
        idx = _capi.check_host(_capi.host_lib().rt_world_push_object(self._h, C.byref(material)))
        return ObjectProxy(self, idx)

    def push_light(self, light: Light) -> None:
        _capi.check_host(_capi.host_lib().rt_world_push_light(self._h, C.byref(light)))

    def save_scene(self, path: str, camera: Optional[Camera] = None) -> None:
        """Write the world (and optionally a camera) as a flat scene file: rt_world_save_scene, include/rt_host.h."""
        _capi.check_host(_capi.host_lib().rt_world_save_scene(self._h, C.byref(camera) if camera is not None else None, str(path).encode()))

    @classmethod
    def load_scene(cls, path: str):
        """Read a scene file: returns (World, Camera or None).  rt_world_load_scene, include/rt_host.h."""
        w = cls()
        cam, has = Camera(), C.c_int(0)
        _capi.check_host(_capi.host_lib().rt_world_load_scene(w._h, str(path).encode(), C.byref(cam), C.byref(has)))
        return w, (cam if has.value else None)

    def desc(self) -> SceneDesc:
        d = SceneDesc()
        _capi.host_lib().rt_world_desc(self._h, C.byref(d))
        d._keepalive = self  # the arrays belong to the world
        return d


def reference_world(obj_path: Optional[str] = None) -> World:
    """The literal scene of main(), src/main.rs:810-1075."""
    w = World()
    _capi.check_host(_capi.host_lib().rt_world_build_reference_scene(w._h, str(obj_path or DEFAULT_OBJ).encode()))
    return w


def reference_camera() -> Camera:
    """src/main.rs:1077-1083."""
    cam = Camera()
    _capi.host_lib().rt_reference_camera(C.byref(cam))
    return cam


class Scene:
    """Device-resident, immutable scene (rt_scene_create / rt_scene_destroy)."""

    def __init__(self, world_or_desc):
        desc = world_or_desc.desc() if isinstance(world_or_desc, World) else world_or_desc
        self._desc = desc
        self._h = C.c_void_p()
        _capi.check(_capi.amd_lib().rt_scene_create(C.byref(desc), C.byref(self._h)))

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            _capi.amd_lib().rt_scene_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def render_whitted(scene: Scene, camera: Camera, frame: Frame, out=None, ray_count=None, stream=None):
    """Whitted pass over one tile into device memory (src/main.rs:1090-1104).

    ``out``: torch float32 CUDA tensor of shape (rows, cols, 3) (allocated if None).
    ``ray_count``: torch int64 CUDA tensor with one element that the cast count is added to.
    Stream-ordered on ``stream`` (default: torch's current stream); returns ``out``.
    """
    import torch

    rows, cols = frame.rows, frame.cols
    if out is None:
        out = torch.empty((rows, cols, 3), dtype=torch.float32, device="cuda")
    if not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.numel() == rows * cols * 3):
        raise ValueError("out must be a contiguous float32 CUDA tensor with rows*cols*3 elements")
    cnt_ptr = None
    if ray_count is not None:
        if not (ray_count.is_cuda and ray_count.dtype == torch.int64 and ray_count.numel() == 1):
            raise ValueError("ray_count must be a 1-element int64 CUDA tensor")
        cnt_ptr = C.c_void_p(ray_count.data_ptr())
    s = stream if stream is not None else torch.cuda.current_stream()
    _capi.check(
        _capi.amd_lib().rt_render_whitted(
            scene._h, C.byref(camera), C.byref(frame), C.c_void_p(out.data_ptr()), cnt_ptr, C.c_void_p(s.cuda_stream)
        )
    )
    return out


def render_whitted_numpy(scene: Scene, camera: Camera, frame: Frame):
    """Host-buffer convenience (rt_render_whitted_host): returns (rgb[rows, cols, 3] float32, casts)."""
    rows, cols = frame.rows, frame.cols
    img = np.empty((rows, cols, 3), dtype=np.float32)
    casts = C.c_ulonglong(0)
    _capi.check(
        _capi.amd_lib().rt_render_whitted_host(scene._h, C.byref(camera), C.byref(frame), img.ctypes.data_as(C.c_void_p), C.byref(casts))
    )
    return img, int(casts.value)


class Rng:
    """Device-resident per-pixel IsaacRng states of one tile (src/main.rs:1117-1127); rt_rng_create/destroy."""

    def __init__(self, frame: Frame):
        self.frame = frame
        self._h = C.c_void_p()
        _capi.check(_capi.amd_lib().rt_rng_create(C.byref(frame), C.byref(self._h)))

    def download(self) -> np.ndarray:
        words = _capi.amd_lib().rt_rng_state_words()
        st = np.empty((self.frame.rows * self.frame.cols, words), dtype=np.uint32)
        _capi.check(_capi.amd_lib().rt_rng_download(self._h, st.ctypes.data_as(C.c_void_p)))
        return st

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            _capi.amd_lib().rt_rng_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def render_distributed(scene: Scene, camera: Camera, frame: Frame, rng: Rng, n_epochs: int = 1, focus: float = 3.0,
                       blur: float = 0.04, accum=None, samples=None, valid=None, ray_count=None, stream=None):
    """`n_epochs` passes of the distributed/DoF closure (src/main.rs:1131-1161) over one tile, on the device.

    accum   (rows, cols, 3) f32 CUDA tensor or None: surviving samples are added in epoch order.
    samples (n_epochs, rows, cols, 3) f32 / valid (n_epochs, rows, cols) u8 CUDA tensors or None: raw samples + filter.
    """
    import torch

    def ptr(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    rows, cols = frame.rows, frame.cols
    for t, shape, dt in ((accum, (rows, cols, 3), torch.float32), (samples, (n_epochs, rows, cols, 3), torch.float32),
                         (valid, (n_epochs, rows, cols), torch.uint8)):
        if t is not None and not (t.is_cuda and t.dtype == dt and t.is_contiguous() and tuple(t.shape) == shape):
            raise ValueError(f"expected a contiguous CUDA {dt} tensor of shape {shape}")
    s = stream if stream is not None else torch.cuda.current_stream()
    _capi.check(
        _capi.amd_lib().rt_render_distributed(scene._h, C.byref(camera), C.byref(frame), float(focus), float(blur), rng._h,
                                              int(n_epochs), ptr(accum), ptr(samples), ptr(valid), ptr(ray_count), C.c_void_p(s.cuda_stream))
    )
    return accum if accum is not None else samples


def render_distributed_numpy(scene: Scene, camera: Camera, frame: Frame, rng: Rng, n_epochs: int, img: np.ndarray,
                             focus: float = 3.0, blur: float = 0.04) -> int:
    """`n_epochs` epochs of the stochastic loop added into the host image `img` ((rows, cols, 3) f32, in place):
    rt_render_distributed_host, the form a host-resident `img` binds (src/main.rs:1131-1167).  Returns the cast count."""
    if not (img.dtype == np.float32 and img.flags.c_contiguous and img.shape == (frame.rows, frame.cols, 3)):
        raise ValueError("expected a contiguous (rows, cols, 3) float32 array")
    casts = C.c_ulonglong(0)
    _capi.check(_capi.amd_lib().rt_render_distributed_host(scene._h, C.byref(camera), C.byref(frame), float(focus), float(blur), rng._h,
                                                           int(n_epochs), img.ctypes.data_as(C.c_void_p), C.byref(casts)))
    return int(casts.value)


def set_option(name: str, value=None) -> None:
    """A process-wide switch of librt_amd.so (include/rt_amd.h rt_set_option): an integer named like the environment variable that
    seeds it (the environment is read once per process); None unsets it.  None of them changes a result."""
    _capi.check(_capi.amd_lib().rt_set_option(name.encode(), None if value is None else str(int(value)).encode()))


class options:
    """`with rt.options(RT_AMD_DIST_PIPELINE=0, RT_AMD_DIST_WS_MB=16): ...` — switches set for the block, unset after it."""

    def __init__(self, **switches):
        self._switches = switches

    def __enter__(self):
        for k, v in self._switches.items():
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k in self._switches:
            set_option(k, None)
        return False


def post_process_device(img, divisor=None, stream=None):
    """In-place p99-luma normalisation of a (rows, cols, 3) f32 CUDA tensor (src/main.rs:748-762), on the device."""
    import torch

    assert img.is_cuda and img.dtype == torch.float32 and img.is_contiguous() and img.shape[-1] == 3
    s = stream if stream is not None else torch.cuda.current_stream()
    _capi.check(_capi.amd_lib().rt_post_process_device(C.c_void_p(img.data_ptr()), img.numel() // 3,
                                                      None if divisor is None else C.c_void_p(divisor.data_ptr()), C.c_void_p(s.cuda_stream)))
    return img


def encode_srgb8_device(img, out=None, stream=None):
    """Linear f32 -> sRGB u8 on the device (src/image.rs:55-66)."""
    import torch

    assert img.is_cuda and img.dtype == torch.float32 and img.is_contiguous()
    if out is None:
        out = torch.empty(img.shape, dtype=torch.uint8, device=img.device)
    s = stream if stream is not None else torch.cuda.current_stream()
    _capi.check(_capi.amd_lib().rt_encode_srgb8_device(C.c_void_p(img.data_ptr()), img.numel(), C.c_void_p(out.data_ptr()), C.c_void_p(s.cuda_stream)))
    return out


class PhotonAccumulator:
    """src/photon.rs:9-34 (defined but unused by the reference's main()): per-pixel running sum and weight,
    resolved to sum / weight — a true average over the epochs of the stochastic pass, as the alternative to main()'s
    sum-and-renormalise.  Works on numpy arrays (librt_host.so) or CUDA tensors (librt_amd.so), bit-identically."""

    def __init__(self, rows: int, cols: int, device: str = "cpu"):
        self.rows, self.cols, self.device = rows, cols, device
        if device == "cpu":
            self.sum = np.zeros((rows, cols, 3), dtype=np.float32)
            self.weight = np.zeros((rows, cols), dtype=np.float32)
        else:
            import torch

            self.sum = torch.zeros((rows, cols, 3), dtype=torch.float32, device=device)
            self.weight = torch.zeros((rows, cols), dtype=torch.float32, device=device)

    def accumulate(self, samples, valid, stream=None) -> None:
        """accumulate() for every sample whose filter flag is set: samples (n_epochs, rows, cols, 3) f32, valid
        (n_epochs, rows, cols) u8 — the `samples` / `valid` outputs of render_distributed — in epoch order."""
        n_epochs = int(samples.shape[0])
        assert tuple(samples.shape) == (n_epochs, self.rows, self.cols, 3) and tuple(valid.shape) == (n_epochs, self.rows, self.cols)
        n_pixels = self.rows * self.cols
        if self.device == "cpu":
            assert samples.dtype == np.float32 and valid.dtype == np.uint8 and samples.flags.c_contiguous and valid.flags.c_contiguous
            _capi.host_lib().rt_accumulate(samples.ctypes.data_as(C.c_void_p), valid.ctypes.data_as(C.c_void_p), n_epochs, n_pixels,
                                           self.sum.ctypes.data_as(C.c_void_p), self.weight.ctypes.data_as(C.c_void_p))
        else:
            import torch

            assert samples.is_cuda and samples.dtype == torch.float32 and samples.is_contiguous()
            assert valid.is_cuda and valid.dtype == torch.uint8 and valid.is_contiguous()
            s = stream if stream is not None else torch.cuda.current_stream()
            _capi.check(_capi.amd_lib().rt_accumulate_device(C.c_void_p(samples.data_ptr()), C.c_void_p(valid.data_ptr()), n_epochs, n_pixels,
                                                            C.c_void_p(self.sum.data_ptr()), C.c_void_p(self.weight.data_ptr()),
                                                            C.c_void_p(s.cuda_stream)))

    def resolve(self, stream=None):
        """into_rgb_internal: sum / weight, black where nothing was accumulated."""
        n_pixels = self.rows * self.cols
        if self.device == "cpu":
            out = np.empty((self.rows, self.cols, 3), dtype=np.float32)
            _capi.host_lib().rt_accumulator_resolve(self.sum.ctypes.data_as(C.c_void_p), self.weight.ctypes.data_as(C.c_void_p), n_pixels,
                                                    out.ctypes.data_as(C.c_void_p))
            return out
        import torch

        out = torch.empty((self.rows, self.cols, 3), dtype=torch.float32, device=self.device)
        s = stream if stream is not None else torch.cuda.current_stream()
        _capi.check(_capi.amd_lib().rt_accumulator_resolve_device(C.c_void_p(self.sum.data_ptr()), C.c_void_p(self.weight.data_ptr()), n_pixels,
                                                                 C.c_void_p(out.data_ptr()), C.c_void_p(s.cuda_stream)))
        return out


def post_process(img: np.ndarray) -> float:
    """In-place p99-luma normalisation, src/main.rs:748-762.  Returns the divisor (0 = untouched)."""
    assert img.dtype == np.float32 and img.flags.c_contiguous and img.shape[-1] == 3
    return float(_capi.host_lib().rt_post_process(img.ctypes.data_as(C.c_void_p), img.size // 3))


def luma_row() -> tuple:
    """The three f32 luma weights of post_process: luma = (w0 * r + w1 * g) + w2 * b."""
    row = (C.c_float * 3)()
    _capi.host_lib().rt_luma_row(row)
    return (float(row[0]), float(row[1]), float(row[2]))


def encode_srgb8(img: np.ndarray) -> np.ndarray:
    """Linear f32 -> sRGB u8, src/image.rs:55-66."""
    assert img.dtype == np.float32 and img.flags.c_contiguous
    out = np.empty(img.shape, dtype=np.uint8)
    _capi.host_lib().rt_encode_srgb8(img.ctypes.data_as(C.c_void_p), img.size, out.ctypes.data_as(C.c_void_p))
    return out


def write_to_file(path: str, rgb8: np.ndarray) -> None:
    """RGB8 PNG via a temporary file + rename, src/main.rs:764-776."""
    assert rgb8.dtype == np.uint8 and rgb8.ndim == 3 and rgb8.shape[2] == 3 and rgb8.flags.c_contiguous
    _capi.check_host(_capi.host_lib().rt_write_png(str(Path(path)).encode(), rgb8.ctypes.data_as(C.c_void_p), rgb8.shape[1], rgb8.shape[0]))
