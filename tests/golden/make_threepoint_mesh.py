"""Fixture generator: the reference's gmsh mesh of the three-point bending test (``meshes/threepoint.msh``, read by
``GridIn`` at cracks.cc:1222-1225 / 1262-1285) -> tests/golden/threepoint_mesh.json (node coordinates and the 280
quadrilaterals in deal.II vertex order).  Run in the build container, where /root/reference exists:

    python tests/golden/make_threepoint_mesh.py
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    src = os.environ.get("CRACKS_SRC", "/root/reference")
    lines = open(os.path.join(src, "meshes", "threepoint.msh")).read().split("\n")
    i = lines.index("$Nodes")
    n = int(lines[i + 1])
    ids, coords = {}, []
    for ln in lines[i + 2:i + 2 + n]:
        t = ln.split()
        ids[int(t[0])] = len(coords)
        coords.append([float(t[1]), float(t[2])])
    i = lines.index("$Elements")
    m = int(lines[i + 1])
    cells = []
    for ln in lines[i + 2:i + 2 + m]:
        t = ln.split()
        if int(t[1]) != 3:  # 4-node quadrangles only (type 1 = boundary lines)
            continue
        ntags = int(t[2])
        a, b, c, d = (ids[int(k)] for k in t[3 + ntags:3 + ntags + 4])
        # gmsh numbers a quadrangle counter-clockwise, deal.II lexicographically: (a, b, d, c)
        x = [coords[k] for k in (a, b, c, d)]
        area2 = sum(x[k][0] * x[(k + 1) % 4][1] - x[(k + 1) % 4][0] * x[k][1] for k in range(4))
        cells.append([a, b, d, c] if area2 > 0 else [a, d, b, c])
    out = {"source": "meshes/threepoint.msh (gmsh 2.2): 325 nodes, 280 quadrangles", "coords": coords, "cells": cells}
    with open(os.path.join(HERE, "threepoint_mesh.json"), "w") as f:
        json.dump(out, f)
    print(len(coords), "nodes,", len(cells), "cells")


if __name__ == "__main__":
    main()
