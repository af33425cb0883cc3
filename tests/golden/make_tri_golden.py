#!/usr/bin/env python3
"""Generates tests/golden/tri_*.npz: depth maps + pixel->vertex maps and the triangle list produced by the
REFERENCE's own MeshGenerator::generateTrianglesGradients (src/NativeUtils/meshGenerator.cpp, compiled where it lies
under /root/reference by oracle/Makefile into oracle/_ref/libref_tri.so).  Run in the build container:

    python tests/golden/make_tri_golden.py

The fixtures are data only (inputs + expected outputs).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from livescan3d_amd import synth  # noqa: E402
from oracle import orc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def cases():
    """(name, depth (h,w) u16, pix_to_vert int32[h*w]) -- also used by tests/test_oracle_triangles.py for a wider sweep."""
    rng = np.random.default_rng(20261004)
    out = []
    for (w, h) in ((64, 48), (33, 21), (8, 5)):
        yy, xx = np.mgrid[0:h, 0:w]
        shapes = {
            "ramp": 1500 + 4 * xx + 3 * yy,
            "steps": 1500 + 4 * xx + 35 * ((xx // 5) % 2) + 28 * ((yy // 4) % 2),
            "noise": 1500 + rng.integers(-14, 15, size=(h, w)),
            "holes": np.where(rng.random((h, w)) < 0.1, 0, 1400 + 9 * xx + 7 * yy),
            "far": 60000 + rng.integers(-200, 200, size=(h, w)),
            "near": 1 + rng.integers(0, 12, size=(h, w)),
        }
        for name, d in shapes.items():
            depth = np.clip(d, 0, 65535).astype(np.uint16)
            # vertex map: raster-order ids of the non-zero pixels; ~5% of the pixels get zero depth (sensor dropout) and a
            # further 3% keep their depth but have no vertex (cropped away: createVertices hands the triangulation the
            # unmodified depth map, depthprocessing.cpp:181, and -1 in depth_to_vertices_map, :128,165)
            drop = rng.random((h, w)) < 0.05
            depth = np.where(drop, 0, depth).astype(np.uint16)
            valid = (depth != 0) & ~(rng.random((h, w)) < 0.03)
            p2v = np.where(valid.ravel(), np.cumsum(valid.ravel()) - 1, -1).astype(np.int32)
            out.append((f"{name}_{w}x{h}", depth, p2v))
    rig = synth.make_rig("scene", 2, 96, 80, seed=9, perturb=False)
    for i in range(2):
        depth = np.frombuffer(rig.depth_maps, dtype=np.uint16)[i * 96 * 80:(i + 1) * 96 * 80].reshape(80, 96)
        rgb = np.frombuffer(rig.depth_colors, dtype=np.uint8)[i * 96 * 80 * 3:(i + 1) * 96 * 80 * 3].reshape(80, 96, 3)
        v, v2p, p2v = orc.create_vertices(depth, rgb, rig.intr[i * 7:(i + 1) * 7], rig.wt[i * 12:(i + 1) * 12], rig.bounds, want_maps=True)
        out.append((f"scene{i}_96x80", depth.copy(), p2v.astype(np.int32)))      # the full depth map, as the reference passes it
    return out


def main():
    assert orc.have_ref_tri(), "oracle/_ref/libref_tri.so missing: run `make -C oracle` where /root/reference exists"
    blob = {}
    names = []
    for name, depth, p2v in cases():
        tri = orc.ref_triangles(depth, p2v)
        blob[name + "_depth"], blob[name + "_p2v"], blob[name + "_tri"] = depth, p2v, tri
        names.append(name)
        print(f"{name}: {len(tri)} triangles")
    blob["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "tri_reference.npz"), **blob)
    print("wrote tri_reference.npz", os.path.getsize(os.path.join(OUT, "tri_reference.npz")), "bytes")


if __name__ == "__main__":
    main()
