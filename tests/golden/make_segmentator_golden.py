#!/usr/bin/env python3
"""Regenerates tests/golden/segmentator_golden.json by running the REFERENCE Segmentator binary
(oracle/_ref/segmentator_ref = g++ on /root/reference/Segmentator/{segmentator,tinyply}.cpp, see oracle/Makefile)
on the procedural meshes of tests/meshes.py.  Run in the build container (needs /root/reference):

    python tests/golden/make_segmentator_golden.py
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import meshes  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "segmentator_ref")


def nan_grid():
    v, f = meshes.grid(4)
    return v, np.concatenate([f, np.array([(0, 0, 1), (0, 5, 10)], np.uint32)])


CASES = [("bent_strip", meshes.bent_strip, 0.01, 1), ("grid", meshes.grid, 0.01, 1), ("grid", meshes.grid, 0.01, 20), ("grid", meshes.grid, 0.5, 1),
         ("l_shape", meshes.l_shape, 0.01, 1), ("l_shape", meshes.l_shape, 0.01, 20), ("l_shape", meshes.l_shape, 0.5, 1),
         ("nan_grid", nan_grid, 0.01, 1),
         ("bumpy_60_5", lambda: meshes.bumpy(60, 5), 0.01, 20), ("bumpy_60_5", lambda: meshes.bumpy(60, 5), 0.001, 1),
         ("bumpy_200_9", lambda: meshes.bumpy(200, 9), 0.01, 20), ("bumpy_200_9", lambda: meshes.bumpy(200, 9), 0.05, 100)]

out = {"generator": "oracle/_ref/segmentator_ref (reference Segmentator, -std=c++11 -O2)", "cases": []}
for name, fn, k, mv in CASES:
    v, f = fn()
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "m.ply")
        meshes.write_ply(p, v, f, "le")
        subprocess.run([REF, p, repr(k), str(mv)], check=True, capture_output=True)
        js = [n for n in os.listdir(d) if n.endswith(".json")]
        seg = json.load(open(os.path.join(d, js[0])))["segIndices"]
    case = {"mesh": name, "k": k, "min_verts": mv}
    if len(seg) <= 64:
        case["seg"] = seg
    else:
        case["sha256"] = hashlib.sha256(np.array(seg, "<i4").tobytes()).hexdigest()
        case["num_segments"] = len(set(seg))
    out["cases"].append(case)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "segmentator_golden.json"), "w"), indent=1)
print("wrote", len(out["cases"]), "cases")
