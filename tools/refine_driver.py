#!/usr/bin/env python3
"""Dev helper: the whole refine pass (lsnRefine, 8 x 512x424 scene clouds, 2 x 10 iterations) timed like bench.py's refine leg."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from livescan3d_amd import native, synth
from oracle import orc

S, w, h = 8, 512, 424
rig = synth.make_rig("scene", S, w, h, seed=4, perturb=True)
v, counts = orc.generate_mesh_vertices(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds, n_threads=8)
e = np.concatenate([[0], np.cumsum(counts)])
xyz = np.stack([v["X"], v["Y"], v["Z"]], axis=1).astype(np.float32)
clouds = [np.ascontiguousarray(xyz[e[i]:e[i + 1]]) for i in range(S)]
wR = np.stack([rig.wt[12 * i + 3:12 * i + 12].reshape(3, 3) for i in range(S)])
wt = np.stack([rig.wt[12 * i:12 * i + 3] for i in range(S)])
native.refine(clouds, wR, wt, 1, 1)
ts = []
for rep in range(int(os.environ.get("REFINE_REPS", "4"))):
    t0 = time.perf_counter()
    out = native.refine(clouds, wR, wt, 2, 10)
    ts.append(1e3 * (time.perf_counter() - t0))
import hashlib
print("refine total_ms", " ".join(f"{t:.2f}" for t in ts), "digest", hashlib.sha256(b"".join(np.ascontiguousarray(c).tobytes() for c in out[0])).hexdigest()[:12])
