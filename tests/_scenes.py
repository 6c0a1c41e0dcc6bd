"""Random and hand-built scenes for parity tests (through the product's World builder)."""
import numpy as np

import homework_18_graphics_raytracer_amd as rt
from homework_18_graphics_raytracer_amd._capi import Camera, Light, Material


def material(rng, kind="random"):
    m = Material()
    m.diffuse_fn = 0
    m.normal_fn = 0
    m.normal = (0.0, 0.0, 1.0)
    m.diffuse_color = tuple(rng.uniform(0.1, 1.0, 3))
    m.specular_color = tuple(rng.uniform(0.0, 1.0, 3))
    m.shiness = float(rng.choice([0.0, 0.1, 0.5, 1.0]))
    m.smoothness = float(rng.choice([1.0, 0.2, 0.01, 0.00001]))
    m.transparency = float(rng.choice([0.0, 0.0, 0.5, 0.96, 1.0]))
    m.refraction_index = float(rng.choice([1.0, 1.12, 1.6]))
    m.opaque_decay = float(rng.choice([0.0, 0.1, 0.3, 1.0]))
    if kind == "random" and rng.random() < 0.3:
        m.diffuse_fn = int(rng.integers(1, 3))
        m.normal_fn = int(rng.integers(0, 2))
        m.tex_color_a = tuple(rng.uniform(0, 1, 3))
        m.tex_color_b = tuple(rng.uniform(0, 1, 3))
        m.tex_frequency = float(rng.choice([3.0, 10.0, 20.0]))
        m.normal_frequency = float(rng.choice([1.0, 10.0]))
        m.normal = (float(rng.uniform(-0.3, 0.3)), 0.0, 1.0)
    return m


def light(rng, kind):
    l = Light()
    l.kind = kind
    l.color = tuple(rng.uniform(0.3, 1.0, 3))
    if kind == 0:
        d = rng.normal(0, 1, 3)
        d[1] = -abs(d[1]) - 0.2
        l.direction = tuple(d / np.linalg.norm(d))
        l.has_origin = int(rng.random() < 0.3)
        l.origin = tuple(rng.uniform(-3, 3, 3) + np.array([0, 6, 0]))
    elif kind == 1:
        l.has_origin = 1
        l.origin = tuple(rng.uniform(-2, 2, 3) + np.array([0, 8, 0]))
        l.direction = (0.0, -1.0, -0.0)
        l.angle = float(rng.uniform(0.4, 1.2))
        l.softness = float(rng.choice([0.5, 1.0, 2.0]))
    else:
        l.has_origin = 1
        l.origin = tuple(rng.uniform(-2, 2, 3) + np.array([0, 2.5, 0]))
    return l


def random_world(seed, n_triangles, n_spheres, n_materials=5, n_lights=3):
    rng = np.random.default_rng(seed)
    w = rt.World()
    proxies = [w.push_object(material(rng)) for _ in range(max(1, n_materials))]
    # a floor so that most rays hit something, then random triangles
    if n_triangles >= 2:
        proxies[0].push_square([(-4, -0.5, -4), (-4, -0.5, 4), (4, -0.5, 4), (4, -0.5, -4)], [(0, 0), (0, 1), (1, 0), (0, 1)])
    k = 2 if n_triangles >= 2 else 0
    while k < n_triangles:
        c = rng.uniform(-2, 2, 3) + np.array([0, 0.8, 0])
        p = c + rng.normal(0, 0.6, (3, 3))
        uv = rng.uniform(0, 1, (3, 2))
        proxies[int(rng.integers(0, len(proxies)))].push_flat_triangle(p.tolist(), uv.tolist())
        k += 1
    for _ in range(n_spheres):
        proxies[int(rng.integers(0, len(proxies)))].push_sphere(tuple(rng.uniform(-1.5, 1.5, 3) + np.array([0, 0.6, 0])), float(rng.uniform(0.2, 0.6)))
    for i in range(n_lights):
        w.push_light(light(rng, i % 3))
    return w


def camera(seed=0):
    rng = np.random.default_rng(1000 + seed)
    cam = Camera()
    cam.fovy = float(np.float32(np.radians(rng.uniform(40, 75))))
    eye = np.array([rng.uniform(2, 3.5), rng.uniform(1.5, 3.0), rng.uniform(2, 3.5)])
    cam.center = tuple(eye)
    t = -eye + rng.normal(0, 0.2, 3)
    cam.toward = tuple(t / np.linalg.norm(t))
    cam.up = (0.0, 1.0, 0.0)
    cam.near = float(rng.choice([-0.1, 0.0, 0.2]))
    return cam


def _box(center, half, rot):
    """The 12 triangles of a box (outward winding): corners c + rot @ (+-hx, +-hy, +-hz)."""
    c = np.asarray(center, dtype=np.float64)
    h = np.asarray(half, dtype=np.float64)
    corner = lambda sx, sy, sz: (c + rot @ (h * np.array([sx, sy, sz]))).tolist()
    quads = [  # each face as a quad in counter-clockwise order seen from outside
        [(1, -1, -1), (1, 1, -1), (1, 1, 1), (1, -1, 1)], [(-1, -1, 1), (-1, 1, 1), (-1, 1, -1), (-1, -1, -1)],
        [(-1, 1, -1), (-1, 1, 1), (1, 1, 1), (1, 1, -1)], [(-1, -1, 1), (-1, -1, -1), (1, -1, -1), (1, -1, 1)],
        [(-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1)], [(1, -1, -1), (-1, -1, -1), (-1, 1, -1), (1, 1, -1)],
    ]
    tris = []
    for q in quads:
        p = [corner(*v) for v in q]
        tris.append([p[0], p[1], p[2]])
        tris.append([p[0], p[2], p[3]])
    return tris


def _rotation(rng):
    q = rng.normal(0, 1, 4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def clustered_world(seed, n_boxes=4, axis_aligned=False, n_spheres=2):
    """Objects whose triangles are consecutive (so that rt_scene_create clusters them: csrc/rt_device_scene.h "segments"):
    a floor, then boxes of 12 triangles each — randomly rotated, or axis-aligned with round coordinates so that rays
    parallel to faces and origins on face planes occur exactly."""
    rng = np.random.default_rng(seed)
    w = rt.World()
    floor = w.push_object(material(rng, "plain"))
    floor.push_square([(-4, -0.5, -4), (-4, -0.5, 4), (4, -0.5, 4), (4, -0.5, -4)], [(0, 0), (0, 1), (1, 0), (0, 1)])
    for b in range(n_boxes):
        obj = w.push_object(material(rng, "plain"))
        if axis_aligned:
            centre = np.array([float(rng.integers(-3, 4)) * 0.5, float(rng.integers(0, 3)) * 0.5, float(rng.integers(-3, 4)) * 0.5])
            half = np.array([0.25, 0.5, 0.25]) * float(rng.integers(1, 3))
            rot = np.eye(3)
        else:
            centre = rng.uniform(-1.5, 1.5, 3) + np.array([0, 0.6, 0])
            half = rng.uniform(0.15, 0.5, 3)
            rot = _rotation(rng)
        for tri in _box(centre, half, rot):
            obj.push_flat_triangle(tri, rng.uniform(0, 1, (3, 2)).tolist())
    for _ in range(n_spheres):
        w.push_object(material(rng, "plain")).push_sphere(tuple(rng.uniform(-1.5, 1.5, 3) + np.array([0, 0.6, 0])), float(rng.uniform(0.2, 0.5)))
    for i in range(3):
        w.push_light(light(rng, i % 3))
    return w


def axis_camera(eye, toward=(0.0, 0.0, -1.0)):
    """A camera whose centre column / row rays have exact zeros in their directions."""
    cam = Camera()
    cam.fovy = float(np.float32(np.radians(60.0)))
    cam.center = tuple(float(x) for x in eye)
    cam.toward = tuple(float(x) for x in toward)
    cam.up = (0.0, 1.0, 0.0)
    cam.near = 0.0
    return cam


def squares_world(seed, n_squares=10):
    """Axis-aligned square()s with round coordinates, a few per object (so they stay plain runs, not clusters): each is two
    triangles on one plane whose face normals differ at most in the signs of zero components — the pairs the intersection
    loop shares a plane evaluation between (rt_device_scene.h RT_TRI_FOLLOWS / _WEAK) — plus a random triangle between
    some of them, so that followers, non-followers and leaders alternate.  With axis_camera() from a grid point, rays
    parallel to the planes (n.d == +-0: t = +-inf / NaN) and origins on them occur exactly."""
    rng = np.random.default_rng(4000 + seed)
    w = rt.World()
    obj = None
    for k in range(n_squares):
        if k % 3 == 0:
            obj = w.push_object(material(rng, "plain"))
        axis = int(rng.integers(0, 3))
        level = float(rng.integers(-2, 4)) * 0.5
        lo = [float(rng.integers(-4, 2)) * 0.5 for _ in range(2)]
        size = [float(rng.integers(1, 5)) * 0.5 for _ in range(2)]
        u, v = [a for a in range(3) if a != axis]
        corners = []
        for du, dv in ((0, 0), (1, 0), (1, 1), (0, 1)) if rng.integers(0, 2) else ((0, 0), (0, 1), (1, 1), (1, 0)):
            p = [0.0, 0.0, 0.0]
            p[axis] = level
            p[u] = lo[0] + du * size[0]
            p[v] = lo[1] + dv * size[1]
            corners.append(tuple(p))
        obj.push_square(corners, [(0, 0), (0, 1), (1, 1), (1, 0)])
        if rng.integers(0, 3) == 0:
            c = rng.uniform(-1.5, 1.5, 3)
            obj.push_flat_triangle((c + rng.normal(0, 0.5, (3, 3))).tolist(), rng.uniform(0, 1, (3, 2)).tolist())
    w.push_object(material(rng, "plain")).push_sphere((0.3, 0.4, -0.2), 0.35)
    for i in range(3):
        w.push_light(light(rng, i % 3))
    return w


def behind_spot_world(axis=(1.0, 2.0, 3.0), scale=0.37):
    """A square facing a spot light from BEHIND, seen through a very narrow camera aimed at the point where the light's axis,
    extended backwards, meets it: for some of those pixels cgmath's `angle` argument dot / (|a| |b|) rounds to just below -1,
    acos gives NaN, `NaN > spread` is false and the light — pointing the other way — DOES ask, with a NaN colour
    (lights.rs:57-60).  ADVICE r3: the shortcut of rt_shade.h light_asks must not turn those into None."""
    w = rt.World()
    m = Material()
    m.diffuse_fn = 0
    m.normal_fn = 0
    m.normal = (0.0, 0.0, 1.0)
    m.diffuse_color = (0.8, 0.7, 0.6)
    m.specular_color = (0.5, 0.5, 0.5)
    m.shiness, m.smoothness, m.transparency, m.refraction_index, m.opaque_decay = 0.3, 0.2, 0.0, 1.0, 0.0
    ob = w.push_object(m)
    o = np.array([0.3, 1.0, -0.2])
    dn = np.array(axis, float)
    dn /= np.linalg.norm(dn)
    c = o - 1.5 * dn
    a = np.cross(dn, [0, 1, 0])
    a /= np.linalg.norm(a)
    b = np.cross(dn, a)
    corners = [c - 2 * a - 2 * b, c + 2 * a - 2 * b, c + 2 * a + 2 * b, c - 2 * a + 2 * b]
    ob.push_square([tuple(p) for p in corners], [(0, 0), (0, 1), (1, 0), (0, 1)])
    spot = Light()
    spot.kind, spot.has_origin = 1, 1
    spot.origin = tuple(o)
    spot.direction = tuple(np.array(axis, float) * scale)
    spot.angle, spot.softness = 0.9, 1.0
    spot.color = (1.0, 0.9, 0.8)
    w.push_light(spot)
    point = Light()
    point.kind, point.has_origin = 2, 1
    point.origin = tuple(o + np.array([0.5, 0.3, 0.1]))
    point.color = (0.3, 0.3, 0.4)
    w.push_light(point)
    cam = Camera()
    eye = o + 0.4 * a + 0.3 * b + 0.2 * dn
    t = c - eye
    cam.center = tuple(eye)
    cam.toward = tuple(t / np.linalg.norm(t))
    cam.up = tuple(b)
    cam.near = -0.1
    cam.fovy = float(np.float32(2 * np.arctan(2e-3 / np.linalg.norm(t))))
    return w, cam
