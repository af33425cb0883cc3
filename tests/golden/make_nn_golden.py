#!/usr/bin/env python3
"""Generates tests/golden/nn_*.npz: point sets + the nearest-neighbour indices / squared distances produced by the
REFERENCE's own NN step (vendored nanoflann 1.1.9 + PointCloud adaptor, src/NativeUtils/icp.cpp:18-32), compiled from
/root/reference by oracle/Makefile into oracle/_ref/.  Run in the build container (needs /root/reference):

    python tests/golden/make_nn_golden.py

The fixtures are data only (inputs + expected outputs); queries whose two nearest targets are at exactly the same f32
distance are dropped so the expected index does not depend on nanoflann's traversal order.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from livescan3d_amd import synth  # noqa: E402
from oracle import orc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def xyz(v):
    return np.stack([v["X"], v["Y"], v["Z"]], axis=1).astype(np.float32)


def tie_free(t, q):
    d = ((q[:, None, :].astype(np.float32) - t[None, :, :]) ** 2)
    d2 = (d[..., 0] + d[..., 1]) + d[..., 2]
    part = np.partition(d2, 1, axis=1) if t.shape[0] > 1 else np.stack([d2[:, 0], d2[:, 0] + 1], 1)
    return part[:, 0] < part[:, 1]


def main():
    assert orc.have_ref_nn(), "oracle/_ref/libref_nn.so missing: run `make -C oracle` where /root/reference exists"
    rng = np.random.default_rng(20261003)
    rig = synth.make_rig("scene", 3, 96, 80, seed=6, perturb=True)
    v, counts = orc.generate_mesh_vertices(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    e = np.concatenate([[0], np.cumsum(counts)])
    pts = xyz(v)
    sets = {
        "nn_scene_pair": (pts[e[0]:e[1]], pts[e[1]:e[2]]),
        "nn_scene_merged": (np.concatenate([pts[e[0]:e[1]], pts[e[2]:e[3]]]), pts[e[1]:e[2]]),
        "nn_gauss_outliers": (rng.normal(size=(3000, 3)).astype(np.float32), (rng.normal(size=(1500, 3)) * 3 + 1).astype(np.float32)),
    }
    for name, (t, q) in sets.items():
        t, q = np.ascontiguousarray(t[:4000]), np.ascontiguousarray(q[:2500])
        keep = tie_free(t, q)
        q = np.ascontiguousarray(q[keep])
        idx, dist = orc.ref_nn(t, q)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), targets=t, queries=q, idx=idx.astype(np.int32), dist2=dist)
        print(name, t.shape, q.shape, "dropped", int((~keep).sum()))


if __name__ == "__main__":
    main()
