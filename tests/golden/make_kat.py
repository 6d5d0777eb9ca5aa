#!/usr/bin/env python3
"""Extract the known-answer fixtures the reference's regression goldens hold for the
assembly hot path into tests/golden/kat.json.

Run in the build container only (needs /root/reference, which does not travel):
    python tests/golden/make_kat.py

Data only: the numbers printed in tests/*.output (parameter block, DoF/cell counts, every
Newton table) and the tests/*.statistics tables.  The line ``0\t\t\t<res>`` of each Newton
table is ||system_pde_residual||_2 after one assemble_nl_residual() + set_zero
(cracks.cc:2790-2799); at time step 0 its input is closed-form, so it pins the hot path
without a linear solve.
"""
import json
import os
import re
import sys

REF = "/root/reference/tests"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat.json")


def parse_output(path):
    rec = {"source": os.path.relpath(path, "/root/reference"), "params": {}, "timesteps": []}
    cur = None
    with open(path) as f:
        for lineno, line in enumerate(f, 1):
            line = line.rstrip("\n")
            m = re.match(r"Problem dimension: (\d+)", line)
            if m:
                rec["dim"] = int(m.group(1))
            m = re.match(r"Cells:\t(\d+)", line)
            if m:
                rec["cells_initial"] = int(m.group(1))
            m = re.match(r"Running on (\d+) cores", line)
            if m:
                rec["ranks"] = int(m.group(1))
            m = re.match(r"DoFs: (\d+) solid \+ (\d+) phase = (\d+)", line)
            if m:
                rec.setdefault("dofs_history", []).append([int(m.group(k)) for k in (1, 2, 3)])
            m = re.match(r"(h \(min\)|k|eps|G_c|gamma penal|Poisson nu|E modulus|Lame mu|Lame lambda):\s+(\S+)", line)
            if m and not rec["timesteps"]:
                rec["params"][m.group(1)] = float(m.group(2))
            m = re.match(r"Timestep (\d+): (\S+) \((\S+)\)\s+Cells: (\d+)\s+DoFs: (\d+)", line)
            if m:
                cur = {"timestep": int(m.group(1)), "time_before": float(m.group(2)),
                       "dt": float(m.group(3)), "cells": int(m.group(4)), "dofs": int(m.group(5)),
                       "line": lineno, "newton": []}
                rec["timesteps"].append(cur)
            m = re.match(r"0\t\t\t(\S+)$", line)
            if m and cur is not None:
                cur["residual0"] = float(m.group(1))
                cur["residual0_line"] = lineno
            m = re.match(r"(\d+)\t(\d+)\t(\d+)\t(\S+)\t(\S+)\t(\d+)\t(\d+)$", line)
            if m and cur is not None:
                cur["newton"].append({"it": int(m.group(1)), "active_set": int(m.group(2)),
                                      "cycling": int(m.group(3)), "residual": float(m.group(4)),
                                      "reduction": float(m.group(5)), "line_search": int(m.group(6)),
                                      "lin_its": int(m.group(7))})
            m = re.match(r"No (\d+) time (\S+) bulk energy: (\S+) crack energy: (\S+)", line)
            if m and cur is not None:
                cur["bulk_energy"] = float(m.group(3))
                cur["crack_energy"] = float(m.group(4))
    return rec


def parse_statistics(path):
    rows = []
    with open(path) as f:
        for line in f:
            if line.startswith("#") or not line.strip():
                continue
            rows.append(line.split())
    return rows


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not present; kat.json is committed, nothing to do")
    out = {}
    for name in sorted(os.listdir(REF)):
        if name.endswith(".output"):
            key = name[:-len(".output")]
            out[key] = parse_output(os.path.join(REF, name))
            st = os.path.join(REF, key + ".statistics")
            if os.path.exists(st):
                out[key]["statistics"] = parse_statistics(st)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", OUT, "with", len(out), "cases")


if __name__ == "__main__":
    main()
