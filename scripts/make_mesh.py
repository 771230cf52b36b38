#!/usr/bin/env python3
"""Synthetic ≈100k-triangle mesh for BASELINE config C5 (the reference's dragon.obj is
absent from its checkout, .MISSING_LARGE_BLOBS): n×n barycentric subdivision of every
triangle of the suzanne fixture (n=10 → 96,800 triangles).  Writes a `v`/`f` OBJ.

    python scripts/make_mesh.py OUT.obj [n]
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent


def load(path):
    v, f = [], []
    for line in Path(path).read_text().splitlines():
        t = line.split()
        if t and t[0] == "v":
            v.append([float(x) for x in t[1:4]])
        elif t and t[0] == "f":
            f.append([int(x.split("/")[0]) - 1 for x in t[1:4]])
    return np.array(v), np.array(f)


def subdivide(v, f, n):
    """Each triangle -> n*n congruent sub-triangles, same winding (front face kept)."""
    tris = []
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]

    def p(i, j):  # barycentric grid point
        return a + (b - a) * (i / n) + (c - a) * (j / n)

    for i in range(n):
        for j in range(n - i):
            tris.append(np.stack([p(i, j), p(i + 1, j), p(i, j + 1)], axis=1))
            if j < n - i - 1:
                tris.append(np.stack([p(i + 1, j), p(i + 1, j + 1), p(i, j + 1)], axis=1))
    return np.concatenate(tris, axis=0)  # [ntri, 3, 3]


def write_obj(tris, out):
    with open(out, "w") as fh:
        for t in tris.reshape(-1, 3):
            fh.write("v %.17g %.17g %.17g\n" % tuple(t))
        for k in range(len(tris)):
            fh.write("f %d %d %d\n" % (3 * k + 1, 3 * k + 2, 3 * k + 3))


if __name__ == "__main__":
    out = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    v, f = load(ROOT / "tests" / "golden" / "suzanne.obj")
    tris = subdivide(v, f, n)
    write_obj(tris, out)
    print(f"{out}: {len(tris)} triangles")
