"""ctypes bindings for the C ABI in include/rt_amd.h and include/rt_host.h.

This is plumbing: struct mirrors, library loading and error translation.  The
render path itself is librt_amd.so (hand-written HIP for gfx950); there is no
Python or CPU fallback — if the library or a device is missing, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
REPO_ROOT = PKG_DIR.parent

RT_OK = 0
RT_MAX_DEPTH = 32


class RtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"rt_amd error {code}: {message}")
        self.code = code


class Vertex(C.Structure):  # geometric.rs:42-47
    _fields_ = [("position", C.c_float * 3), ("normal", C.c_float * 3), ("uv", C.c_float * 2)]


class Triangle(C.Structure):  # primitives.rs:26-29
    _fields_ = [("object_index", C.c_uint32), ("vertices", Vertex * 3)]


class Sphere(C.Structure):  # primitives.rs:15-24
    _fields_ = [("object_index", C.c_uint32), ("center", C.c_float * 3), ("radius", C.c_float)]


class Material(C.Structure):  # materials.rs:21-31 + enumerated closures
    _fields_ = [
        ("diffuse_fn", C.c_uint32),
        ("normal_fn", C.c_uint32),
        ("normal", C.c_float * 3),
        ("diffuse_color", C.c_float * 3),
        ("shiness", C.c_float),
        ("specular_color", C.c_float * 3),
        ("smoothness", C.c_float),
        ("transparency", C.c_float),
        ("refraction_index", C.c_float),
        ("opaque_decay", C.c_float),
        ("tex_color_a", C.c_float * 3),
        ("tex_color_b", C.c_float * 3),
        ("tex_frequency", C.c_float),
        ("normal_frequency", C.c_float),
    ]


class Light(C.Structure):  # lights.rs:6-30
    _fields_ = [
        ("kind", C.c_uint32),
        ("has_origin", C.c_uint32),
        ("origin", C.c_float * 3),
        ("direction", C.c_float * 3),
        ("angle", C.c_float),
        ("softness", C.c_float),
        ("color", C.c_float * 3),
    ]


class SceneDesc(C.Structure):  # main.rs:130-137
    _fields_ = [
        ("triangles", C.POINTER(Triangle)),
        ("n_triangles", C.c_uint32),
        ("spheres", C.POINTER(Sphere)),
        ("n_spheres", C.c_uint32),
        ("materials", C.POINTER(Material)),
        ("n_materials", C.c_uint32),
        ("lights", C.POINTER(Light)),
        ("n_lights", C.c_uint32),
    ]


class Camera(C.Structure):  # main.rs:43-49
    _fields_ = [
        ("fovy", C.c_float),
        ("center", C.c_float * 3),
        ("toward", C.c_float * 3),
        ("up", C.c_float * 3),
        ("near", C.c_float),
    ]


class Frame(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("max_depth", C.c_int32),
        ("x0", C.c_uint32),
        ("y0", C.c_uint32),
        ("x1", C.c_uint32),
        ("y1", C.c_uint32),
        ("y_step", C.c_uint32),
    ]

    @classmethod
    def full(cls, width: int, height: int, max_depth: int) -> "Frame":
        return cls(width, height, max_depth, 0, 0, width, height, 1)

    @classmethod
    def rows_of_rank(cls, width: int, height: int, max_depth: int, rank: int, world: int) -> "Frame":
        """Interleaved row bands: rank r renders rows r, r+world, ...."""
        return cls(width, height, max_depth, 0, rank, width, height, world)

    @property
    def cols(self) -> int:
        return self.x1 - self.x0

    @property
    def rows(self) -> int:
        return (self.y1 - self.y0 + self.y_step - 1) // self.y_step


# every symbol include/rt_amd.h declares (checked by tests/test_capi_symbols.py)
DEFAULT_VARIANT = 18  # RT_VARIANT_PWF | RT_VARIANT_STATIC (csrc/rt_kernels.h)

def sources_sha256() -> str:
    """One hash over the sources librt_amd.so is built from (csrc/*.hip, *.h, *.inc, the Makefile), in name order.  The committed
    counter profiles (profiles/traffic*.json, tools/make_traffic.py) carry the hash of the sources they were taken with; bench.py
    reports counter-based figures only while it equals this one — a kernel change without a fresh profile reads as `null`, not as a
    stale number."""
    import hashlib

    h = hashlib.sha256()
    src = PKG_DIR / "csrc"
    for f in sorted(list(src.glob("*.hip")) + list(src.glob("*.h")) + list(src.glob("*.inc")) + [src / "Makefile"]):
        h.update(f.name.encode() + b"\0")
        h.update(f.read_bytes())
    return h.hexdigest()


AMD_SYMBOLS = [
    "rt_abi_version", "rt_last_error", "rt_device_count", "rt_set_device", "rt_frame_rows", "rt_frame_pixels",
    "rt_scene_create", "rt_scene_destroy", "rt_render_whitted", "rt_render_whitted_host", "rt_set_option", "rt_set_variant",
    "rt_get_variant", "rt_set_wavefront_budget", "rt_set_distributed_split", "rt_profile_enable", "rt_profile_read", "rt_profile_read_distributed", "rt_math_eval_host", "rt_math_eval_device", "rt_scene_describe_nodes", "rt_rng_state_words", "rt_rng_create",
    "rt_rng_destroy", "rt_rng_download", "rt_render_distributed", "rt_render_distributed_host", "rt_multi_create", "rt_multi_destroy", "rt_multi_render_whitted_host", "rt_multi_render_distributed_host", "rt_multi_render_whitted", "rt_multi_render_distributed", "rt_post_process_device", "rt_post_keys_device", "rt_post_hist_device", "rt_post_pick_device", "rt_post_scale_device", "rt_post_release", "rt_encode_srgb8_device", "rt_accumulate_device", "rt_accumulator_resolve_device",
]
HOST_SYMBOLS = [
    "rt_world_new", "rt_world_free", "rt_world_push_object", "rt_world_push_triangle", "rt_world_push_sphere",
    "rt_world_push_light", "rt_world_push_flat_triangle", "rt_world_push_square", "rt_world_load_obj",
    "rt_world_build_reference_scene", "rt_world_save_scene", "rt_world_load_scene", "rt_reference_camera", "rt_world_desc", "rt_frame_full", "rt_post_process", "rt_luma_row",
    "rt_encode_srgb8", "rt_accumulate", "rt_accumulator_resolve", "rt_write_png", "rt_host_last_error",
]

_amd = None
_host = None


def _load(name: str) -> C.CDLL:
    path = PKG_DIR / name
    if not path.exists():
        raise RtError(-2, f"{path} is not built; run __graft_entry__.build() (make -C {PKG_DIR / 'csrc'})")
    return C.CDLL(str(path), mode=getattr(os, "RTLD_NOW", 2))


def host_lib() -> C.CDLL:
    global _host
    if _host is None:
        lib = _load("librt_host.so")
        lib.rt_world_new.restype = C.c_void_p
        lib.rt_world_free.argtypes = [C.c_void_p]
        lib.rt_world_free.restype = None
        lib.rt_world_push_object.argtypes = [C.c_void_p, C.POINTER(Material)]
        lib.rt_world_push_triangle.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Vertex)]
        lib.rt_world_push_sphere.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float), C.c_float]
        lib.rt_world_push_light.argtypes = [C.c_void_p, C.POINTER(Light)]
        lib.rt_world_push_flat_triangle.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.rt_world_push_square.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.rt_world_load_obj.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_float, C.POINTER(C.c_float)]
        lib.rt_world_build_reference_scene.argtypes = [C.c_void_p, C.c_char_p]
        lib.rt_world_save_scene.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_char_p]
        lib.rt_world_load_scene.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(Camera), C.POINTER(C.c_int)]
        lib.rt_reference_camera.argtypes = [C.POINTER(Camera)]
        lib.rt_reference_camera.restype = None
        lib.rt_world_desc.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        lib.rt_world_desc.restype = None
        lib.rt_frame_full.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.POINTER(Frame)]
        lib.rt_frame_full.restype = None
        lib.rt_post_process.argtypes = [C.c_void_p, C.c_size_t]
        lib.rt_post_process.restype = C.c_float
        lib.rt_encode_srgb8.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        lib.rt_encode_srgb8.restype = None
        lib.rt_accumulate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_size_t, C.c_void_p, C.c_void_p]
        lib.rt_accumulate.restype = None
        lib.rt_accumulator_resolve.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.rt_accumulator_resolve.restype = None
        lib.rt_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        lib.rt_host_last_error.restype = C.c_char_p
        _host = lib
    return _host


def amd_lib() -> C.CDLL:
    """The HIP library.  Loading needs libamdhip64 but no GPU; calls need a GPU."""
    global _amd
    if _amd is None:
        # One HIP runtime per process: PyTorch ships its own libamdhip64.so (soname libamdhip64.so.7) and a
        # process that initialises two copies of the runtime loses the device in the second one ("no
        # ROCm-capable device is detected").  Import torch FIRST so that librt_amd.so's DT_NEEDED
        # libamdhip64.so.7 binds to the copy torch already loaded; without torch it binds to /opt/rocm's.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = _load("librt_amd.so")
        lib.rt_last_error.restype = C.c_char_p
        lib.rt_frame_rows.argtypes = [C.POINTER(Frame)]
        lib.rt_frame_rows.restype = C.c_uint32
        lib.rt_frame_pixels.argtypes = [C.POINTER(Frame)]
        lib.rt_frame_pixels.restype = C.c_uint64
        lib.rt_scene_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
        lib.rt_scene_destroy.argtypes = [C.c_void_p]
        lib.rt_render_whitted.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Frame), C.c_void_p, C.c_void_p, C.c_void_p]
        lib.rt_render_whitted_host.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Frame), C.c_void_p, C.POINTER(C.c_ulonglong)]
        lib.rt_set_variant.argtypes = [C.c_int]
        lib.rt_set_option.argtypes = [C.c_char_p, C.c_char_p]
        lib.rt_multi_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
        lib.rt_multi_destroy.argtypes = [C.c_void_p]
        lib.rt_multi_render_whitted_host.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Frame), C.c_void_p, C.POINTER(C.c_ulonglong)]
        lib.rt_multi_render_distributed_host.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Frame), C.c_float, C.c_float, C.c_uint32,
                                                         C.c_void_p, C.POINTER(C.c_ulonglong)]
        lib.rt_multi_render_whitted.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Frame), C.c_void_p, C.c_void_p, C.c_void_p]
        lib.rt_multi_render_distributed.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Frame), C.c_float, C.c_float, C.c_uint32,
                                                    C.c_void_p, C.c_void_p, C.c_void_p]
        lib.rt_post_process_device.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        lib.rt_post_keys_device.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.rt_post_hist_device.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        lib.rt_post_pick_device.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        lib.rt_post_scale_device.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.rt_encode_srgb8_device.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        lib.rt_accumulate_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.rt_accumulator_resolve_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        lib.rt_rng_create.argtypes = [C.POINTER(Frame), C.POINTER(C.c_void_p)]
        lib.rt_rng_destroy.argtypes = [C.c_void_p]
        lib.rt_rng_download.argtypes = [C.c_void_p, C.c_void_p]
        lib.rt_render_distributed.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Frame), C.c_float, C.c_float, C.c_void_p,
                                              C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.rt_render_distributed_host.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Frame), C.c_float, C.c_float, C.c_void_p,
                                                   C.c_uint32, C.c_void_p, C.POINTER(C.c_ulonglong)]
        lib.rt_set_wavefront_budget.argtypes = [C.c_uint]
        lib.rt_set_distributed_split.argtypes = [C.c_int]
        lib.rt_profile_enable.argtypes = [C.c_int]
        lib.rt_profile_read.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_uint)]
        lib.rt_math_eval_host.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        lib.rt_math_eval_device.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        lib.rt_scene_describe_nodes.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32)]
        _amd = lib
    return _amd


def check(rc: int) -> int:
    if rc < 0:
        raise RtError(rc, amd_lib().rt_last_error().decode("utf-8", "replace"))
    return rc


def check_host(rc: int) -> int:
    if rc < 0:
        raise RtError(rc, host_lib().rt_host_last_error().decode("utf-8", "replace"))
    return rc
