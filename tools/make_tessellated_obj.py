#!/usr/bin/env python3
"""Write a tessellated copy of an OBJ's first model: every triangle split into 4^levels by edge midpoints.

    python tools/make_tessellated_obj.py tests/golden/dodecahedron.obj out.obj --levels 3 [--spherize]

The reference imports one OBJ through load_obj (main.rs:778-807: positions and triangular faces of the first model
only, no normals / uvs).  Feeding a tessellated dodecahedron to the same import path — rt_world_load_obj, or
rt_world_build_reference_scene(obj_path) for the whole literal scene around it — is how the scene-size sweep
(tools/scene_sweep.py) gets 64 -> 172 -> 604 -> ... -> 589 852 triangles without inventing geometry:
36 * 4^levels + 28.  --spherize pushes the new vertices out to the circumscribed sphere (a geodesic solid: every
triangle then has its own plane, the harder case for any per-object rejection); without it the solid keeps its 12 flat
faces and the image is that of the original up to which coplanar piece a ray meets.
"""
import argparse
import math


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--levels", type=int, default=1)
    ap.add_argument("--spherize", action="store_true")
    a = ap.parse_args()
    verts, faces = [], []
    for line in open(a.src):
        t = line.split()
        if t[:1] == ["v"]:
            verts.append(tuple(float(x) for x in t[1:4]))
        elif t[:1] == ["f"]:
            idx = [int(x.split("/")[0]) for x in t[1:4]]
            faces.append(tuple(i - 1 if i > 0 else len(verts) + i for i in idx))
    n0 = len(verts)
    cx = sum(v[0] for v in verts) / n0
    cy = sum(v[1] for v in verts) / n0
    cz = sum(v[2] for v in verts) / n0
    radius = max(math.dist(v, (cx, cy, cz)) for v in verts)
    for _ in range(a.levels):
        mid = {}

        def midpoint(i, j):
            key = (i, j) if i < j else (j, i)
            if key not in mid:
                p = tuple((verts[i][k] + verts[j][k]) * 0.5 for k in range(3))
                if a.spherize:
                    d = math.dist(p, (cx, cy, cz))
                    if d > 0:
                        p = tuple((cx, cy, cz)[k] + (p[k] - (cx, cy, cz)[k]) * radius / d for k in range(3))
                mid[key] = len(verts)
                verts.append(p)
            return mid[key]

        out = []
        for (i, j, k) in faces:  # same winding as the parent: normals keep pointing outwards
            ij, jk, ki = midpoint(i, j), midpoint(j, k), midpoint(k, i)
            out += [(i, ij, ki), (ij, j, jk), (ki, jk, k), (ij, jk, ki)]
        faces = out
    with open(a.dst, "w") as f:
        f.write(f"# {a.src} tessellated {a.levels} level(s){' and pushed out to its circumscribed sphere' if a.spherize else ''}: {len(faces)} triangles\n")
        for v in verts:
            f.write("v %.9g %.9g %.9g\n" % v)
        for (i, j, k) in faces:
            f.write(f"f {i + 1} {j + 1} {k + 1}\n")
    print(f"{a.dst}: {len(verts)} vertices, {len(faces)} triangles")


if __name__ == "__main__":
    main()
