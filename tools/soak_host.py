#!/usr/bin/env python3
"""Soak of the host exports: three threads (merge calls, single-sensor calls, the tick as one call) hammer the library for N seconds;
every result must have the digest of that call's first result.  usage: soak_host.py [seconds]"""
import hashlib
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from livescan3d_amd import native, synth  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
native.require_gpu()
rig = synth.make_rig("scene", 8, 512, 424, seed=4, perturb=True)
rig_n = synth.make_rig("noise", 8, 512, 424, seed=1, bounds=synth.CROP_BOUNDS)
errors, counts = [], {}


def dig(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def loop(name, fn):
    try:
        want, n, t_end = None, 0, time.time() + secs
        while time.time() < t_end:
            got = fn(n)
            if want is None:
                want = {}
            key = got[0]
            if key not in want:
                want[key] = got[1]
            elif want[key] != got[1]:
                errors.append(f"{name}: call {n} ({key}) differs from the first one")
                return
            n += 1
        counts[name] = n
    except Exception as ex:  # noqa: BLE001
        errors.append(f"{name}: {ex!r}")


def merge(n):
    r = rig if n % 2 else rig_n
    v, t = native.generate_mesh_from_depth_maps(r.depth_maps, r.depth_colors, r.widths, r.heights, r.intr, r.wt, r.bounds)
    # (lsnLastMesh* is not exercised here: "the mesh of the lane's last call" is the tick thread's as often as this thread's)
    return ("m%d" % (n % 2), dig(v, t))


def single(n):
    i = n % 8
    v = native.generate_vertices_from_depth_map(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds, i)
    return ("s%d" % i, dig(v))


def tick(n):
    v, t, d, c = native.correct_and_generate_mesh(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds, write_back=bool(n % 2))
    return ("t%d" % (n % 2), dig(v, t, d, c))


threads = [threading.Thread(target=loop, args=a) for a in (("merge", merge), ("single", single), ("tick", tick))]
for t in threads:
    t.start()
for t in threads:
    t.join()
print("calls", counts, "errors", errors)
sys.exit(1 if errors else 0)
