"""Hand-derived known answers for the oracle's building blocks (the reference has no tests, SURVEY §4;
these are the analytic cases SURVEY §8c lists, plus the quirks §8a says must be reproduced)."""
import ctypes as C
import math

import numpy as np
import pytest

import homework_18_graphics_raytracer_amd as rt
from homework_18_graphics_raytracer_amd import _capi
import _oracle
from _oracle import BACK, BOTH, FRONT, SPHERE, TRIANGLE, OrcHit, OrcRay, f3, ray

L = _oracle.lib()


def scene(triangles=(), spheres=(), materials=1, lights=()):
    """Small hand-built scenes through the product's World builder (flat triangles)."""
    w = rt.World()
    base = rt.reference_world().desc().materials
    proxies = [w.push_object(base[1]) for _ in range(materials)]
    for obj, pos in triangles:
        proxies[obj].push_flat_triangle(pos, [(0, 0), (1, 0), (0, 1)])
    for obj, c, r in spheres:
        proxies[obj].push_sphere(c, r)
    for l in lights:
        w.push_light(l)
    return w


def cast(world, r):
    h = OrcHit()
    ok = L.orc_cast(C.byref(world.desc()), C.byref(r), C.byref(h))
    return h if ok else None


def test_clip_mapping_divides_both_axes_by_height():
    out = (C.c_float * 2)()
    L.orc_clip(1280, 960, 0, 0, out)  # main.rs:1094-1095
    assert (out[0], out[1]) == (np.float32(-640.0 / 960.0), np.float32(0.5))
    L.orc_clip(1280, 960, 640, 480, out)
    assert (out[0], out[1]) == (0.0, 0.0)


def test_centre_pixel_ray_is_normalised_toward_and_origin_is_behind_center():
    cam = rt.reference_camera()
    r = OrcRay()
    L.orc_shoot(C.byref(cam), (C.c_float * 2)(0.0, 0.0), C.byref(r))
    t = np.array(list(cam.toward), dtype=np.float32)
    assert np.allclose(list(r.direction), t / np.linalg.norm(t), atol=1e-7)
    # near = -0.1 puts the origin BEHIND the centre (main.rs:1082): center + toward * near
    assert np.allclose(list(r.origin), np.array(list(cam.center)) - 0.1 * t, atol=1e-6)
    assert r.face_direction == FRONT and r.has_exclude == 0


def test_sphere_front_and_back_hits():
    w = scene(spheres=[(0, (0, 0, 0), 1.0)])
    h = cast(w, ray((0, 0, 5), (0, 0, -1), FRONT))
    assert h.kind == SPHERE and h.distance == 4.0 and h.face_direction == FRONT and list(h.normal) == [0, 0, 1]
    h = cast(w, ray((0, 0, 5), (0, 0, -1), BACK))
    assert h.distance == 6.0 and h.face_direction == BACK and list(h.normal) == [0, 0, 1]  # flipped inward normal
    # Both: tc < k picks the far (back) root only when the origin is inside
    assert cast(w, ray((0, 0, 5), (0, 0, -1), BOTH)).distance == 4.0
    assert cast(w, ray((0, 0, 0), (0, 0, -1), BOTH)).face_direction == BACK
    assert cast(w, ray((0, 0, 5), (0, 0, 1), FRONT)) is None  # behind the ray: t <= 0
    assert cast(w, ray((2, 0, 5), (0, 0, -1), FRONT)) is None  # line-sphere distance 2 > r


def test_sphere_uv():
    w = scene(spheres=[(0, (0, 0, 0), 1.0)])
    h = cast(w, ray((0, 5, 0), (0, -1, 0), FRONT))  # normal (0,1,0): u = acos(1)/pi = 0, v = atan2(0,0)/(2pi)+0.5
    assert list(h.uv) == [0.0, 0.5]
    h = cast(w, ray((5, 0, 0), (-1, 0, 0), FRONT))  # normal (1,0,0): u = 0.5, v = 0.5
    assert list(h.uv) == [0.5, 0.5]


TRI = [(0, 0, 0), (1, 0, 0), (0, 1, 0)]  # face normal +z


def test_triangle_culling_modes():
    w = scene(triangles=[(0, TRI)])
    down = ray((0.25, 0.25, 1), (0, 0, -1), FRONT)
    h = cast(w, down)
    assert h.kind == TRIANGLE and h.distance == 1.0 and h.face_direction == FRONT and list(h.normal) == [0, 0, 1]
    assert list(h.position) == [0.25, 0.25, 0.0]
    assert cast(w, ray((0.25, 0.25, 1), (0, 0, -1), BACK)) is None      # front face culled in Back mode
    up = cast(w, ray((0.25, 0.25, -1), (0, 0, 1), BACK))
    assert up.face_direction == BACK and list(up.normal) == [0, 0, -1]  # normal negated on a backface
    assert cast(w, ray((0.25, 0.25, -1), (0, 0, 1), FRONT)) is None
    assert cast(w, ray((0.25, 0.25, -1), (0, 0, 1), BOTH)).face_direction == BACK
    assert cast(w, ray((0.75, 0.75, 1), (0, 0, -1), FRONT)) is None      # outside: an area is negative


def test_barycentric_uv_interpolation():
    w = scene(triangles=[(0, TRI)])
    h = cast(w, ray((0.25, 0.5, 1), (0, 0, -1), FRONT))
    assert np.allclose(list(h.uv), [0.25, 0.5], atol=1e-7)  # uvs (0,0),(1,0),(0,1) -> uv = (x, y)


def test_exclusion_only_skips_the_matching_face_direction():
    w = scene(triangles=[(0, TRI)])
    o, d = (0.25, 0.25, 1), (0, 0, -1)
    assert cast(w, ray(o, d, FRONT, exclude=(TRIANGLE, 0, FRONT))) is None     # Front exclusion skips a front hit
    assert cast(w, ray(o, d, FRONT, exclude=(TRIANGLE, 0, BACK))) is not None  # Back exclusion does not
    assert cast(w, ray(o, d, FRONT, exclude=(TRIANGLE, 0, BOTH))) is None
    assert cast(w, ray(o, d, FRONT, exclude=(SPHERE, 0, FRONT))) is not None   # other primitive kind


def test_equal_distance_later_primitive_wins_and_spheres_beat_triangles():
    # two coincident triangles: the nearest test is `nearest < t -> skip`, so a tie REPLACES (main.rs:229-233)
    w = scene(triangles=[(0, TRI), (1, TRI)], materials=2)
    h = cast(w, ray((0.25, 0.25, 1), (0, 0, -1), FRONT))
    assert h.index == 1 and h.object_index == 1
    # a sphere touching the same point at the same t wins over the triangle (sphere loop runs second)
    w = scene(triangles=[(0, TRI)], spheres=[(1, (0.25, 0.25, -1.0), 1.0)], materials=2)
    h = cast(w, ray((0.25, 0.25, 1), (0, 0, -1), FRONT))
    assert h.kind == SPHERE and h.distance == 1.0


def test_refract_closure():
    out = (C.c_float * 3)()
    n, l = f3(0, 0, 1), f3(0.6, 0, -0.8)
    assert L.orc_refract_dir(n, l, 1.0, out) == 1 and np.allclose(list(out), [0.6, 0, -0.8], atol=1e-7)  # k=1: straight through
    assert L.orc_refract_dir(n, f3(0, 0, -1), 1.6, out) == 1 and list(out) == [0, 0, -1]                 # normal incidence
    # total internal reflection: k^2 < 1 - cos^2  (k = 1/1.6 = 0.625 < sin = 0.8)
    assert L.orc_refract_dir(n, f3(0.8, 0, -0.6), 1.0 / 1.6, out) == 0
    assert L.orc_refract_dir(n, l, 1.0 / 1.6, out) == 1  # sin = 0.6 < 0.625 still escapes
    # Snell: sin_out = sin_in / k
    assert L.orc_refract_dir(n, l, 1.6, out) == 1
    assert abs(out[0] - 0.6 / 1.6) < 1e-6 and abs(np.linalg.norm(list(out)) - 1) < 1e-6


def test_reflect_ray_inherits_mode_and_excludes_inverted_face():
    hit = OrcHit()
    hit.kind, hit.index, hit.face_direction = TRIANGLE, 7, FRONT
    hit.position, hit.normal = f3(1, 2, 3), f3(0, 0, 1)
    inc = ray((0, 0, 0), (0.6, 0, -0.8), BACK)
    out = OrcRay()
    L.orc_reflect(C.byref(hit), C.byref(inc), C.byref(out))
    assert np.allclose(list(out.direction), [0.6, 0, 0.8], atol=1e-7) and list(out.origin) == [1, 2, 3]
    assert out.face_direction == BACK  # the INCOMING ray's mode (main.rs:334)
    assert (out.has_exclude, out.exclude_kind, out.exclude_index, out.exclude_face) == (1, TRIANGLE, 7, BACK)


def _light(kind, **kw):
    l = _capi.Light()
    l.kind = kind
    l.has_origin = 0 if kind == 0 else 1
    for k, v in kw.items():
        setattr(l, k, f3(*v) if isinstance(v, (tuple, list)) else v)
    return l


def _directional(light, pos):
    d, c, o, ho = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)(), C.c_int()
    ok = L.orc_light_directional(C.byref(light), f3(*pos), d, c, o, C.byref(ho))
    return (list(d), list(c), list(o), ho.value) if ok else None


def test_point_light_attenuates_with_inverse_distance_not_squared():
    l = _light(2, origin=(0, 0, 0), color=(1, 1, 1))
    d, c, o, ho = _directional(l, (0, 4, 0))
    assert d == [0, 1, 0] and ho == 1
    assert np.allclose(c, [0.25] * 3, rtol=1e-6)  # 1/(4+eps), lights.rs:76


def test_spot_light_cone_and_softness():
    l = _light(1, origin=(0, 10, 0), direction=(0, -1, 0), angle=math.radians(60), softness=1.0, color=(1, 0.5, 0.9))
    d, c, o, ho = _directional(l, (0, 0, 0))  # on the axis: angle 0 -> angular 1; distance 10
    assert d == [0, -1, 0] and np.allclose(c, [0.1, 0.05, 0.09], rtol=1e-6)
    assert _directional(l, (100, 9, 0)) is None  # ~89.4 deg off-axis > 60 deg: None
    d, c, o, ho = _directional(l, (10, 0, 0))    # 45 deg off axis: angular = (1 - 45/60)^(1+eps) = 0.25
    assert np.allclose(c[0], 0.25 / math.sqrt(200), rtol=1e-5)


def test_generative_materials():
    d = rt.reference_world().desc()
    out = (C.c_float * 14)()
    wall, checker = d.materials[2], d.materials[7]
    L.orc_material_approx(C.byref(wall), (C.c_float * 2)(0.0, 0.04), out)   # (0.04*20) as i32 = 0 -> even -> white
    assert list(out[3:6]) == [1, 1, 1] and list(out[0:3]) == [0, 0, 1]       # angle 0: (sin 0, 0, cos 0)
    L.orc_material_approx(C.byref(wall), (C.c_float * 2)(0.0, 0.06), out)   # 1.2 -> 1 -> odd
    assert list(out[3:6]) == [0.5, 0.5, 1]
    L.orc_material_approx(C.byref(wall), (C.c_float * 2)(0.05, 0.0), out)   # angle pi: v = (~0,0,-1) -> flipped to +z
    assert out[2] == 1.0
    L.orc_material_approx(C.byref(checker), (C.c_float * 2)(0.05, 0.06), out)  # (0.11*10) as i32 = 1 -> odd
    assert np.allclose(list(out[3:6]), [0.1, 0.1, 1.0])
    # Rust's % keeps the sign: a negative odd cell has remainder -1 != 0 -> colour b; negative even -> colour a
    L.orc_material_approx(C.byref(checker), (C.c_float * 2)(-0.15, 0.0), out)  # -1.5 -> -1
    assert np.allclose(list(out[3:6]), [0.1, 0.1, 1.0])
    L.orc_material_approx(C.byref(checker), (C.c_float * 2)(-0.25, 0.0), out)  # -2.5 -> -2
    assert np.allclose(list(out[3:6]), [1.0, 0.1, 0.1])


def test_adjust_normal_is_close_to_but_computed_from_the_arc():
    out = (C.c_float * 3)()
    L.orc_adjust_normal(f3(0, 0, 1), f3(0, 0, 1), out)      # identity branch (dot ~= mag)
    assert list(out) == [0, 0, 1]
    L.orc_adjust_normal(f3(0, 0, 1), f3(0, 1, 0), out)      # general branch
    assert np.allclose(list(out), [0, 1, 0], atol=2e-7)
    L.orc_adjust_normal(f3(0, 0, 1), f3(0, 0, -1), out)     # antiparallel: 180 deg about normalize(x cross z) = -y
    assert np.allclose(list(out), [0, 0, -1], atol=2e-7)
    n = np.array([0.3, -0.5, 0.81], dtype=np.float32)
    n /= np.linalg.norm(n)
    L.orc_adjust_normal(f3(0.6, 0, 0.8), f3(*n), out)       # a tilted bump normal stays unit length and 0.8 along n
    assert abs(np.linalg.norm(list(out)) - 1) < 1e-6 and abs(np.dot(list(out), n) - 0.8) < 1e-6


def test_diffuse_and_specular():
    d = rt.reference_world().desc()
    m = d.materials[5]  # red sphere: diffuse (1,.2,.2), specular yellow, smoothness 0.2
    dif, spe = (C.c_float * 3)(), (C.c_float * 3)()
    n, v = f3(0, 0, 1), f3(0, 0, 1)
    L.orc_diffuse_specular(C.byref(m), (C.c_float * 2)(0, 0), n, v, f3(0, 0, 1), dif, spe)
    assert np.allclose(list(dif), [1, 0.2, 0.2])
    s = 1.0 / (0.2 + 1.1920929e-7)
    assert np.allclose(list(spe), [(s + 8) / (8 * math.pi)] * 2 + [0.0], rtol=1e-6)  # r.v = 1 -> 1^s * e
    L.orc_diffuse_specular(C.byref(m), (C.c_float * 2)(0, 0), n, v, f3(0, 0, -1), dif, spe)
    assert list(dif) == [0, 0, 0] and list(spe) == [0, 0, 0]  # light below the surface


def test_depth_zero_returns_unscaled_shade_and_threshold_prunes():
    """main.rs:488-490: at depth <= 0 the shade is returned WITHOUT the (1-shiness)(1-transparency) factor."""
    world = rt.reference_world()
    d = world.desc()
    r = ray((1.5, 2.0, 1.5), (0.0, -1.0, 0.0), FRONT)  # straight down onto the floor (shiness 0.5)
    h = cast(world, r)
    assert h is not None and h.object_index == 1
    shade = (C.c_float * 3)()
    L.orc_get_shade(C.byref(d), C.byref(h), C.byref(r), shade, None)
    rgb0 = (C.c_float * 3)()
    casts = C.c_uint64()
    L.orc_ray_trace(C.byref(d), C.byref(r), 0, 1.0, rgb0, C.byref(casts))
    assert list(rgb0) == list(shade)
    # contribution below THRESHOLD at entry -> black, no cast at all (main.rs:469-471)
    L.orc_ray_trace(C.byref(d), C.byref(r), 5, 0.0009, rgb0, C.byref(casts))
    assert list(rgb0) == [0, 0, 0] and casts.value == 0


def test_get_refract_through_a_glass_slab_escapes_parallel():
    """A ray through the first glass slab (index 1.6, 0.1 thick) leaves parallel to how it entered."""
    world = rt.reference_world()
    d = world.desc()
    r = ray((0.1, 1.25, 2.0), (0.0, 0.0, -1.0), FRONT)
    h = cast(world, r)
    assert h.object_index == 4 and h.kind == TRIANGLE  # the nearer (second) slab, z = 0.81
    travel = C.c_float()
    esc = OrcRay()
    kind = L.orc_get_refract(C.byref(d), C.byref(h), C.byref(r), 100.0, C.byref(travel), C.byref(esc))
    assert kind == 0  # Escaped
    assert abs(travel.value - 0.1) < 1e-6 and np.allclose(list(esc.direction), [0, 0, -1], atol=1e-6)
    assert esc.face_direction == FRONT and (esc.exclude_kind, esc.exclude_face) == (TRIANGLE, BACK)
