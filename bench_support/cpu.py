"""bench_support.cpu -- the CPU side: the oracle (port of the reference path) and, where it can be built, the reference's own code, timed on this host.
The only part of the bench that loads oracle/ (as the baseline beside the product, never as the product)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from .common import host_description


def cpu_tick(synth, S, w, h):
    """One tick of the reference's real work on the host CPU, per stage, on scene frames (the same generator the GPU legs use):
    the port (oracle/lsn_oracle.c, single thread unless stated) and, where it can be built, the reference's own code."""
    from oracle import orc
    rig = synth.make_rig("scene", S, w, h, seed=4, tick=0)

    def best_of(fn, reps=3):
        best, val = float("inf"), None
        for _ in range(reps):
            t0 = time.perf_counter()
            val = fn()
            best = min(best, time.perf_counter() - t0)
        return 1e3 * best, val

    threads = min(S, os.cpu_count() or 1)
    radial_ms, corrected = best_of(lambda: orc.radial_correction(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, n_threads=threads))
    cd = np.ascontiguousarray(np.asarray(corrected[0])).view(np.uint8).ravel()
    cc = np.ascontiguousarray(np.asarray(corrected[1])).ravel()
    mesh_ms, mesh = best_of(lambda: orc.generate_mesh(cd, cc, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds))
    out = {"workload": f"one tick of {S} x {w}x{h} scene frames", "kind": "port",
           "radial_ms": radial_ms, "radial_threads": threads, "mesh_ms": mesh_ms, "mesh_threads": 1, "full_tick_ms": radial_ms + mesh_ms,
           "vertices": int(len(mesh[0])), "triangles": int(len(mesh[2])),
           "note": "radial: depthMapAndColorSetRadialCorrection's port, one thread per sensor like depthprocessing.cpp:1794-1815; mesh: "
                   "createVertices + generateTrianglesGradients + formMesh, sensors one after the other; best of 3"}
    if orc.have_ref_tri():
        # the REFERENCE's own triangulation (src/NativeUtils/meshGenerator.cpp compiled in place, 4 row-band threads inside like the
        # reference runs it) on the same corrected frames, sensor after sensor -- what cpu_reference_tri_ms reports
        P = w * h
        maps = []
        for i in range(S):
            d = cd.view(np.uint16)[i * P:(i + 1) * P].reshape(h, w)
            c = cc[3 * i * P:3 * (i + 1) * P].reshape(h, w, 3)
            _, _, p2v = orc.create_vertices(d, c, rig.intr[7 * i:7 * i + 7], rig.wt[12 * i:12 * i + 12], rig.bounds, want_maps=True)
            maps.append((np.ascontiguousarray(d), np.ascontiguousarray(p2v.reshape(h, w))))
        ref_ms, n_tri = best_of(lambda: sum(len(orc.ref_triangles(d, m)) for d, m in maps))
        out["reference_triangulation_ms"] = ref_ms
        out["reference_triangulation_kind"] = "reference (meshGenerator.cpp:147-181 compiled in place, its own 4 threads), all sensors of the tick one after the other"
        out["reference_triangles"] = int(n_tri)
    return out


def cpu_baseline(args, synth, S, w, h, bounds):
    """The CPU oracle (port of createVertices/formMesh, one thread per sensor like the reference's std::thread fan-out)
    on the same tick shape, for about --cpu-seconds of wall time."""
    from oracle import orc
    cores = os.cpu_count() or 1
    threads = min(S, cores)
    rig = synth.make_rig("noise", S, w, h, seed=1, bounds=bounds)
    orc.generate_mesh_vertices(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds, n_threads=threads)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < args.cpu_seconds:
        orc.generate_mesh_vertices(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds, n_threads=threads)
        n += 1
    dt = time.perf_counter() - t0
    # the same call on one thread (SURVEY 8d asks for both), a quarter of the time budget
    n1, t1 = 0, time.perf_counter()
    while time.perf_counter() - t1 < args.cpu_seconds / 4:
        orc.generate_mesh_vertices(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds, n_threads=1)
        n1 += 1
    dt1 = time.perf_counter() - t1
    # BASELINE configs[0]: one 512x424 frame, unproject + transform + crop on one CPU thread
    rig0 = synth.make_rig("noise", 1, 512, 424, seed=1, bounds=bounds)
    v0, _ = orc.generate_mesh_vertices(rig0.depth_maps, rig0.depth_colors, rig0.widths, rig0.heights, rig0.intr, rig0.wt, rig0.bounds, n_threads=1)
    n0, t2 = 0, time.perf_counter()
    while time.perf_counter() - t2 < 1.0:
        orc.generate_mesh_vertices(rig0.depth_maps, rig0.depth_colors, rig0.widths, rig0.heights, rig0.intr, rig0.wt, rig0.bounds, n_threads=1)
        n0 += 1
    ms0 = 1e3 * (time.perf_counter() - t2) / n0
    config0 = {"workload": "configs[0]: 1 x 512x424, CPU port, 1 thread", "ms_per_frame": ms0, "frames_per_s": 1e3 / ms0,
               "algorithmic_GBps": (2 * 512 * 424 + 19 * len(v0)) / (ms0 * 1e-3) / 1e9}
    return {"config0": config0, "value": n / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{n} merge calls of {S} x {w}x{h} (same generator, tick 0) in {dt:.1f} s, {threads} threads (one per sensor), host has {cores} cores; "
                      "the threaded figure is allocation- and concatenation-bound like the reference it mirrors (a 28 MB scratch malloc'ed and page-faulted "
                      "per call, the per-sensor clouds concatenated serially: oracle/lsn_oracle.c:98-111 = depthprocessing.cpp:128-136,1578-1608), not compute-bound",
            "single_thread_value": n1 / dt1, "host": host_description()}

