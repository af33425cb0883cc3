"""bench_support.legs_host -- the exports on HOST arrays (PCIe-inclusive; never `value`) and the outbound formats."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from .common import PCIE_GBS


def bench_host_path(native, synth, S, w, h, bounds, only=None, seconds=1.5, rehearse_sharded=True):
    """The reference's own exports on HOST arrays, exactly as KinectServer calls them (KinectServer.cs:354-389, 527-554): upload,
    kernels, download, deleteMesh.  PCIe-bound: every variant is set against bytes_up / 63 GB/s + bytes_down / 63 GB/s (the
    download cannot start before the upload has been consumed)."""
    import ctypes as C
    L = native.lib()
    vp = C.c_void_p
    out = {"pcie_peak_GBs_per_direction": PCIE_GBS,
           "note": "calls timed back to back from one host thread for ~1.5 s each; frac_of_pcie_bound = (bytes_up + bytes_down) / 63 GB/s / time per call, "
                   "frac_of_full_duplex_bound = max(bytes_up, bytes_down) / 63 GB/s / time per call",
           "host_path": os.environ.get("LSN_HOST_PATH", "direct"), "sensors_per_group": os.environ.get("LSN_HOST_GROUP", "by size (copies >= 1 MiB)"),
           "host_devices": os.environ.get("LSN_HOST_DEVICES", "one (no sharding)")}
    wanted = (lambda name: True) if only is None else (lambda name: name in only)

    def row(describe, dt, bytes_up, bytes_down, nv, nt):
        # two bounds: the link used one way at a time (what a call that uploads everything before the first byte leaves can reach),
        # and full duplex (both directions at the 63 GB/s of the spec at once: the longer of the two transfers)
        half = (bytes_up + bytes_down) / (PCIE_GBS * 1e9)
        full = max(bytes_up, bytes_down) / (PCIE_GBS * 1e9)
        return {"what": describe, "calls_per_s": 1.0 / dt, "ms_per_call": 1e3 * dt, "bytes_up": int(bytes_up), "bytes_down": int(bytes_down),
                "vertices": int(nv), "triangles": int(nt), "pcie_bound_ms": 1e3 * half, "frac_of_pcie_bound": half / dt,
                "pcie_full_duplex_bound_ms": 1e3 * full, "frac_of_full_duplex_bound": full / dt}

    def run(name, rig, call, bytes_up, describe):
        for _ in range(4):           # the caller's arrays get registered on their second sighting
            nv, nt = call()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            call()
            n += 1
        dt = (time.perf_counter() - t0) / n
        bytes_down = 16 * nv + 12 * nt
        out[name] = row(describe, dt, bytes_up, bytes_down, nv, nt)

    try:   # what a plain 15 MB copy reaches on this box (pinned host memory, either direction): the practical ceiling under the 63 GB/s of the spec
        import torch
        hbuf = torch.empty(15 << 20, dtype=torch.uint8).pin_memory()
        dbuf = torch.empty(15 << 20, dtype=torch.uint8, device="cuda")
        rates = {}
        for name, dst, src in (("h2d", dbuf, hbuf), ("d2h", hbuf, dbuf)):
            best = None
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                dst.copy_(src, non_blocking=True)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None or dt < best else best
            rates[name] = (15 << 20) / best / 1e9
        out["plain_copy_15MB_GBs"] = rates
        del hbuf, dbuf
    except Exception as e:  # noqa: BLE001
        out["plain_copy_15MB_GBs"] = f"not measured: {e}"

    for kind in ("noise", "scene"):
        rig = synth.make_rig(kind, S, w, h, seed=1, bounds=bounds) if kind == "noise" else synth.make_rig(kind, S, w, h, seed=4, perturb=True)
        argv = [S, rig.depth_maps.ctypes.data_as(vp), rig.depth_colors.ctypes.data_as(vp), rig.widths.ctypes.data_as(vp),
                rig.heights.ctypes.data_as(vp), rig.intr.ctypes.data_as(vp), rig.wt.ctypes.data_as(vp)]
        bnd = [float(x) for x in rig.bounds]
        mesh = native.Mesh()

        def merge():   # exactly what KinectServer.GenerateMesh does around the P/Invoke, minus the managed copies
            L.generateMeshFromDepthMaps(*argv, C.byref(mesh), False, *bnd, False)
            n = (mesh.nVertices, mesh.nTriangles)
            L.deleteMesh(C.byref(mesh))
            return n

        def singles():  # GetLatestFrameVerticesOnly: one generateVerticesFromDepthMap per sensor (the refine path's input)
            nv = 0
            for i in range(S):
                L.generateVerticesFromDepthMap(*argv[1:], C.byref(mesh), *bnd, i)
                nv += mesh.nVertices
                L.deleteMesh(C.byref(mesh))
            return nv, 0

        up = rig.depth_maps.nbytes + rig.depth_colors.nbytes
        if kind == "scene":
            # the reference's tick through the boundary: CorrectRadialDistortionsForDepthMaps, then GenerateMesh (KinectServer.cs:518-525, :354-374).
            # The correction works in place on the caller's arrays, so every call starts from a fresh copy of the raw frames (the copy is
            # outside the timed part of a call).
            raw_d, raw_c = rig.depth_maps.copy(), rig.depth_colors.copy()
            wd, wc = rig.depth_maps.copy(), rig.depth_colors.copy()
            argv_w = [S, wd.ctypes.data_as(vp), wc.ctypes.data_as(vp)] + argv[3:]

            def timed_tick(name, fn, bytes_up, bytes_down_extra, describe):
                nv = nt = 0
                for _ in range(3):
                    np.copyto(wd, raw_d); np.copyto(wc, raw_c)
                    nv, nt = fn()
                n, acc, t_end = 0, 0.0, time.perf_counter() + seconds
                while time.perf_counter() < t_end:
                    np.copyto(wd, raw_d); np.copyto(wc, raw_c)
                    t0 = time.perf_counter()
                    fn()
                    acc += time.perf_counter() - t0
                    n += 1
                dt = acc / n
                bytes_down = 16 * nv + 12 * nt + bytes_down_extra
                out[name] = row(describe, dt, bytes_up, bytes_down, nv, nt)

            def radial_only():
                L.depthMapAndColorSetRadialCorrection(*argv_w[:6])
                return 0, 0

            def tick_two_calls():
                L.depthMapAndColorSetRadialCorrection(*argv_w[:6])
                L.generateMeshFromDepthMaps(*argv_w, C.byref(mesh), False, *bnd, False)
                n = (mesh.nVertices, mesh.nTriangles)
                L.deleteMesh(C.byref(mesh))
                return n

            def tick_one_call():
                L.lsnCorrectAndGenerateMesh(*argv_w, C.byref(mesh), *bnd, 1)
                n = (mesh.nVertices, mesh.nTriangles)
                L.deleteMesh(C.byref(mesh))
                return n

            if wanted("radial_scene"):
                timed_tick("radial_scene", radial_only, up, up, f"depthMapAndColorSetRadialCorrection, {S} x {w}x{h} scene frames, corrected in place in the caller's arrays")
            if wanted("tick_two_calls_scene"):
                timed_tick("tick_two_calls_scene", tick_two_calls, 2 * up, up,
                           "the reference's tick: depthMapAndColorSetRadialCorrection then generateMeshFromDepthMaps + deleteMesh (the frames cross PCIe twice on the way up)")
            if wanted("tick_one_call_scene"):
                timed_tick("tick_one_call_scene", tick_one_call, up, up,
                           "lsnCorrectAndGenerateMesh + deleteMesh: the same tick with one upload (corrected maps written back, vertices + triangles back)")
        if wanted(f"merge_{kind}"):
            run(f"merge_{kind}", rig, merge, up, f"generateMeshFromDepthMaps + deleteMesh, {S} x {w}x{h} {kind} frames, vertices + triangles back")
        if kind == "scene" and wanted("vertices_only_scene"):
            run("vertices_only_scene", rig, singles, up, f"{S} x (generateVerticesFromDepthMap + deleteMesh), the {S} sensors of one scene tick, vertices only")
    if rehearse_sharded and not os.environ.get("LSN_HOST_DEVICES"):
        out["sharded_rehearsal"] = sharded_rehearsal(S, w, h)
    return out


def sharded_rehearsal(S, w, h):
    """The merge calls sharded over "several devices" on a box that has one GPU ($LSN_HOST_DEVICES lists it several times; the list is read
    once per process, so child processes run tools/host_path.py / tools/shard_parts.py).  Two things a one-GPU box CAN measure:
      (1) the flow as it runs, two parts sharing the one link: its control flow and its fixed costs, nothing about several links;
      (2) every part of a call ALONE on the link ($LSN_HOST_SHARD_SOLO=1: the parts run one after the other), for 2 / 4 / 8 parts -- what a
          part would cost on a device and link of its own.  Side by side on real devices a call takes about as long as its slowest part
          plus the hand-over to the worker threads (~20 us); host memory bandwidth and root-complex contention of concurrent transfers are
          NOT in that estimate.
    Neither is a multi-device measurement."""
    import subprocess
    res = {"status": "UNMEASURED ON MULTI-DEVICE HARDWARE: a one-GPU box; (1) both parts share its GPU and its one PCIe link, (2) the parts of a call "
                     "one after the other, each alone on that link", "devices": "0,0"}
    try:
        env = dict(os.environ, LSN_HOST_DEVICES="0,0", LSN_HOST_PATH_ROWS="merge_noise,merge_scene,tick_one_call_scene", LSN_HOST_PATH_SECONDS="0.6")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_path.py"), str(S), str(w), str(h)], env=env, capture_output=True, text=True, timeout=180)
        rows = json.loads(r.stdout.strip().splitlines()[-1])
        for k in ("merge_noise", "merge_scene", "tick_one_call_scene"):
            res[k] = {"ms_per_call_two_parts_one_link": rows[k]["ms_per_call"], "vertices": rows[k]["vertices"], "triangles": rows[k]["triangles"],
                      "slowest_part_alone_ms": {}, "estimated_ms_per_call_on_own_links": {}}
        if (S, w, h) == (8, 512, 424):
            for D in (2, 4, 8):
                env = dict(os.environ, LSN_HOST_DEVICES=",".join(["0"] * D), LSN_HOST_SHARD_SOLO="1", LSN_SHARD_PARTS_REPS="25")
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shard_parts.py"), str(D)], env=env, capture_output=True, text=True, timeout=180)
                row = json.loads(r.stdout.strip().splitlines()[-1])
                for k in ("merge_noise", "merge_scene", "tick_one_call_scene"):
                    ms = row[k]["slowest_part_us_median"] / 1e3
                    res[k]["slowest_part_alone_ms"][f"{D}_parts"] = ms
                    res[k]["estimated_ms_per_call_on_own_links"][f"{D}_devices"] = ms + 0.02
    except Exception as ex:  # noqa: BLE001
        res["error"] = f"{type(ex).__name__}: {ex}"
    return res


def bench_wire(args, torch, native, synth, dev, stream, S, w, h, bounds, with_cpu):
    """TransferSocket.SendFrame stream (with TransferServer's chunking) and binary PLY image of one tick's
    merged mesh (scene frames: a real triangulated surface), device resident in and out."""
    from livescan3d_amd.fusion import DeviceFusion
    rig = synth.make_rig("scene", S, w, h, seed=3, bounds=bounds)
    fus = DeviceFusion(1, rig.widths, rig.heights, device=dev.index)
    fus.set_params(rig.intr, rig.wt, rig.bounds)
    P = w * h
    depth = torch.from_numpy(rig.depth_maps.view(np.int16).copy()).to(dev).view(1, S * P)
    rgb = torch.from_numpy(rig.depth_colors.copy()).to(dev).view(1, S * P * 3)
    cap = fus.capacity
    tri = torch.empty((1, 2 * cap, 3), dtype=torch.int32, device=dev)
    toff = torch.zeros((1, S + 1), dtype=torch.int32, device=dev)
    fus.plan.run_mesh(depth.data_ptr(), rgb.data_ptr(), fus.vertices.data_ptr(), fus.offsets.data_ptr(), tri.data_ptr(), toff.data_ptr(), stream)
    torch.cuda.synchronize()
    nv, nt = int(fus.offsets[0, -1].item()), int(toff[0, -1].item())
    bound = native.transfer_frame_bound(nv, nt)
    out = torch.empty(bound + 64, dtype=torch.uint8, device=dev)
    packer = native.TransferPacker(dev.index, nv, nt)
    reps = 10
    n = packer.pack(fus.vertices.data_ptr(), nv, tri.data_ptr(), nt, out.data_ptr(), bound, stream)
    t0 = time.perf_counter()
    for _ in range(reps):
        n = packer.pack(fus.vertices.data_ptr(), nv, tri.data_ptr(), nt, out.data_ptr(), bound, stream)
    t_stream = (time.perf_counter() - t0) / reps
    n_chunks = int(out[8:12].view(torch.int32).item())
    pb = native.ply_binary_bytes(nv, nt)
    native.ply_pack(dev.index, fus.vertices.data_ptr(), nv, tri.data_ptr(), nt, out.data_ptr(), pb, stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        native.ply_pack(dev.index, fus.vertices.data_ptr(), nv, tri.data_ptr(), nt, out.data_ptr(), pb, stream)
    torch.cuda.synchronize()
    t_ply = (time.perf_counter() - t0) / reps
    res = {"workload": f"one tick of {S} x {w}x{h} scene frames: {nv} vertices, {nt} triangles",
           "transfer_stream": {"ms": 1e3 * t_stream, "bytes": n, "chunks": n_chunks,
                               "note": "lsnTransferPack: formMeshChunks re-indexing + SendFrame layout on the device, incl. its host synchronisations"},
           "ply": {"ms": 1e3 * t_ply, "bytes": pb, "GBps": (16 * nv + 12 * nt + pb) / t_ply / 1e9,
                   "note": "lsnPlyPack: reads 16 B/vertex + 12 B/triangle, writes the 15 B / 13 B records"}}
    if with_cpu:
        from oracle import orc
        v = fus.vertices[0, :nv].cpu().numpy().view(native.VERTEX_DTYPE).reshape(-1)
        t = tri[0, :nt].cpu().numpy()
        t0 = time.perf_counter()
        ref = orc.transfer_frame(v, t)
        res["transfer_stream"]["cpu_port_ms"] = 1e3 * (time.perf_counter() - t0)
        packer.pack(fus.vertices.data_ptr(), nv, tri.data_ptr(), nt, out.data_ptr(), bound, stream)
        res["transfer_stream"]["identical_to_cpu_port"] = out[:n].cpu().numpy().tobytes() == ref
        t0 = time.perf_counter()
        orc.ply_binary(v, t)
        res["ply"]["cpu_port_ms"] = 1e3 * (time.perf_counter() - t0)
    return res

