#!/usr/bin/env python3
"""What a one-GPU box can say about merge calls sharded over D devices: with LSN_HOST_SHARD_SOLO=1 the D parts of a call run one after the other,
each ALONE on the box's one PCIe link -- upload of its sensor block, count, exchange, stores, triangles -- and the library keeps every part's
wall time.  On D devices with a link each the parts run side by side, so a call takes about as long as its SLOWEST part (+ the hand-over to the
worker threads, measured by the ordinary rehearsal).  Not a multi-device measurement: host memory bandwidth and root-complex contention of D
concurrent transfers are not in it.

    python3 tools/shard_parts.py            # D = 2, 4, 8 (a child process each: the device list is read once per process)
"""
import ctypes as C
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(D):
    import numpy as np
    from livescan3d_amd import native, synth
    native.require_gpu()
    L = native.lib()
    vp = C.c_void_p
    S, w, h = 8, 512, 424
    out = {"devices": D}
    for kind in ("noise", "scene"):
        rig = synth.make_rig(kind, S, w, h, seed=1, bounds=synth.CROP_BOUNDS) if kind == "noise" else synth.make_rig(kind, S, w, h, seed=4, perturb=True)
        bnd = [float(x) for x in rig.bounds]
        mesh = native.Mesh()
        wd, wc = rig.depth_maps.copy(), rig.depth_colors.copy()
        argv = [S, wd.ctypes.data_as(vp), wc.ctypes.data_as(vp), rig.widths.ctypes.data_as(vp), rig.heights.ctypes.data_as(vp), rig.intr.ctypes.data_as(vp),
                rig.wt.ctypes.data_as(vp)]

        def merge():
            L.generateMeshFromDepthMaps(*argv, C.byref(mesh), False, *bnd, False)
            n = (mesh.nVertices, mesh.nTriangles)
            L.deleteMesh(C.byref(mesh))
            return n

        def tick():
            np.copyto(wd, rig.depth_maps); np.copyto(wc, rig.depth_colors)
            L.lsnCorrectAndGenerateMesh(*argv, C.byref(mesh), *bnd, 1)
            n = (mesh.nVertices, mesh.nTriangles)
            L.deleteMesh(C.byref(mesh))
            return n

        for name, fn in ((f"merge_{kind}", merge),) + ((("tick_one_call_scene", tick),) if kind == "scene" else ()):
            for _ in range(5):
                nv, nt = fn()
            slow, total, parts = [], [], []
            for _ in range(int(os.environ.get("LSN_SHARD_PARTS_REPS", "60"))):
                fn()
                us = native.host_shard_part_micros()
                slow.append(max(us)); total.append(sum(us)); parts.append(us)
            out[name] = {"vertices": nv, "triangles": nt, "slowest_part_us_median": statistics.median(slow), "all_parts_us_median": statistics.median(total),
                         "parts_us_median": [statistics.median(p[d] for p in parts) for d in range(len(parts[0]))]}
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]))
    else:
        for D in (2, 4, 8):
            env = dict(os.environ, LSN_HOST_DEVICES=",".join(["0"] * D), LSN_HOST_SHARD_SOLO="1")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), str(D)], env=env, capture_output=True, text=True, timeout=300)
            if r.returncode != 0:
                print(f"D={D}: failed: {r.stderr[-500:]}")
                continue
            row = json.loads(r.stdout.strip().splitlines()[-1])
            for k in ("merge_noise", "merge_scene", "tick_one_call_scene"):
                v = row[k]
                print(f"D={D} {k:22s} slowest part {v['slowest_part_us_median']:7.1f} us   sum of parts {v['all_parts_us_median']:7.1f} us   parts {[round(x) for x in v['parts_us_median']]}")
            print(json.dumps(row))
