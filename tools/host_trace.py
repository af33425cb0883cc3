#!/usr/bin/env python3
"""A few dozen merge calls through the export, for a timeline: run under `rocprofv3 --kernel-trace --memory-copy-trace` or with
LSN_HOST_TRACE=1 (the library prints the wall-clock phases of three calls).  usage: host_trace.py [noise|scene|tick] [calls]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from livescan3d_amd import native, synth  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "noise"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 30
S, w, h = 8, 512, 424
native.require_gpu()
L = native.lib()
vp = C.c_void_p
rig = synth.make_rig("noise", S, w, h, seed=1, bounds=synth.CROP_BOUNDS) if kind == "noise" else synth.make_rig("scene", S, w, h, seed=4, perturb=True)
raw_d, raw_c = rig.depth_maps.copy(), rig.depth_colors.copy()
argv = [S, rig.depth_maps.ctypes.data_as(vp), rig.depth_colors.ctypes.data_as(vp), rig.widths.ctypes.data_as(vp),
        rig.heights.ctypes.data_as(vp), rig.intr.ctypes.data_as(vp), rig.wt.ctypes.data_as(vp)]
bnd = [float(x) for x in rig.bounds]
mesh = native.Mesh()
import time  # noqa: E402
ts = []
for i in range(calls):
    if kind == "tick":
        np.copyto(rig.depth_maps, raw_d); np.copyto(rig.depth_colors, raw_c)
    t0 = time.perf_counter()
    if kind == "tick":
        L.lsnCorrectAndGenerateMesh(*argv, C.byref(mesh), *bnd, 1)
    else:
        L.generateMeshFromDepthMaps(*argv, C.byref(mesh), False, *bnd, False)
    nv, nt = mesh.nVertices, mesh.nTriangles
    L.deleteMesh(C.byref(mesh))
    ts.append(time.perf_counter() - t0)
print(f"{kind}: {nv} vertices, {nt} triangles; median call {1e3 * sorted(ts)[len(ts) // 2]:.3f} ms, best {1e3 * min(ts):.3f} ms", file=sys.stderr)
