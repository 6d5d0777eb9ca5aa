"""Fixture generator: values of the reference's ``func_emodulus`` (cracks.cc:1539-1544: ``BitmapFunction`` over
``$SRC/test.pgm``, range [E, 10 E], E = 1e4) at the cell centres of the tests/hetero_3d_1.prm mesh
-> tests/golden/hetero_3d_emod.json.  Run in the build container, where /root/reference exists:

    python tests/golden/make_hetero_emod.py

``BitmapFile`` / ``BitmapFunction`` are restated from cracks.cc:118-241, including the quirk that the bilinear
weights are clamped with min(max(.,1),0) and therefore vanish (the value is that of the lower-left pixel)."""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from cracks_amd import mesh as M  # noqa: E402


class BitmapFile:  # cracks.cc:137-207
    def __init__(self, name):
        with open(name) as f:
            tok = f.read().split("\n", 1)[1]  # first line: magic number
        lines = [ln for ln in tok.split("\n")]
        if lines[0].startswith("#"):
            lines = lines[1:]
        vals = " ".join(lines).split()
        self.nx, self.ny = int(vals[0]), int(vals[1])
        data = np.array(vals[2:2 + self.nx * self.ny + 1], dtype=np.int64)
        # the reference reads nx*ny numbers after the size line; the first of them is the file's maxval token
        self.image = data[:self.nx * self.ny] / 255.0
        self.hx, self.hy = 1.0 / (self.nx - 1), 1.0 / (self.ny - 1)

    def pixel(self, i, j):
        return self.image[self.nx * (self.ny - 1 - j) + i]

    def value(self, x, y):
        ix = min(max(int(x / self.hx), 0), self.nx - 2)
        iy = min(max(int(y / self.hy), 0), self.ny - 2)
        xi = min(max((x - ix * self.hx) / self.hx, 1.0), 0.0)
        eta = min(max((y - iy * self.hy) / self.hy, 1.0), 0.0)
        return ((1 - xi) * (1 - eta) * self.pixel(ix, iy) + xi * (1 - eta) * self.pixel(ix + 1, iy) +
                (1 - xi) * eta * self.pixel(ix, iy + 1) + xi * eta * self.pixel(ix + 1, iy + 1))


def bitmap_function_3d(f, p, x1, x2, y1, y2, vmin, vmax):  # cracks.cc:219-236
    x = (p[0] - x1) / (x2 - x1)
    y = (p[1] - y1) / (y2 - y1)
    z = (p[2] - y1) / (y2 - y1)
    return vmin + (f.value(x / 10.0, (y - z) / 10.0) + 0.5 * f.value((x + y) / 2.0, (z + x) / 2.0) +
                   0.25 * f.value(math.fmod(z + x - y, 10.0), math.fmod(y + x, 10.0))) * (vmax - vmin) / 2.25


def main():
    src = os.environ.get("CRACKS_SRC", "/root/reference")
    f = BitmapFile(os.path.join(src, "test.pgm"))
    mesh = M.hetero_3d_prerefined_mesh()
    E = 1.0e4
    cells = []
    for c in mesh.coords[mesh.cells].mean(axis=1):
        cells.append([float(c[0]), float(c[1]), float(c[2]), float(bitmap_function_3d(f, c, 0, 10, 0, 10, E, 10.0 * E))])
    out = {"source": "BitmapFunction(test.pgm, 0, 10, 0, 10, 1e4, 1e5) at the cell centres of tests/hetero_3d_1.prm",
           "cells": cells}
    with open(os.path.join(HERE, "hetero_3d_emod.json"), "w") as g:
        json.dump(out, g)
    print(len(cells), "cells; E range", min(c[3] for c in cells), max(c[3] for c in cells))


if __name__ == "__main__":
    main()
