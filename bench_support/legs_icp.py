"""bench_support.legs_icp -- ICP (configs[1], configs[2]) and the whole refine pass, with a roofline per kernel group."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from .common import HBM_PEAK_GBS, VALU_F32_PEAK_TF


def _scene_clouds_on_device(torch, synth, dev, n_sensors, w=512, h=424):
    """The per-sensor clouds of one scene tick, produced by the fusion kernels themselves (device tensors [n_i, 3] f32)."""
    from livescan3d_amd.fusion import DeviceFusion, upload_rig
    rig = synth.make_rig("scene", n_sensors, w, h, seed=4, perturb=True)
    fus = DeviceFusion(1, rig.widths, rig.heights, device=dev.index)
    fus.set_params(rig.intr, rig.wt, rig.bounds)
    d, c = upload_rig(rig, 1, dev.index)
    v, off = fus.run(d, c)
    torch.cuda.synchronize()
    off = off[0].cpu().numpy()
    xyz = v[0, :int(off[-1]), 4:16].contiguous().view(torch.float32).view(-1, 3)
    return [xyz[int(off[i]):int(off[i + 1])].contiguous() for i in range(n_sensors)]


def _time_icp(torch, native, ws, tgt, src0, iters, mode, reps, stream, dev, profile=False):
    """Best-of-reps wall time of one lsnIcpRun (HIP events on the launch stream); with profile=True also the library's own
    phase timing of the best run and the number of one-to-one matches of the last iteration."""
    n1, n2 = tgt.shape[0], src0.shape[0]
    best, best_prof = None, None
    ws.set_profiling(profile)
    for r in range(reps + 1):
        src = src0.clone()
        Rt = torch.tensor([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=torch.float32, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        ws.run(tgt.data_ptr(), n1, src.data_ptr(), n2, Rt.data_ptr(), Rt.data_ptr() + 36, iters, mode, stream)
        e1.record()
        torch.cuda.synchronize()
        if r > 0 or reps == 1:
            t = e0.elapsed_time(e1)
            if best is None or t < best:
                best = t
                best_prof = ws.profile(stream) if profile else None
    ws.set_profiling(False)
    return best, best_prof


def _icp_roofline(n1, n2, m, iters, prof, brute_ms=None):
    """Per kernel group: SURVEY 8(d)'s algorithmic bytes per iteration / the library's HIP-event time per iteration / 8 TB/s.
    The apply pass of iteration k rides in the first NN kernel of iteration k+1, so the NN group carries its 24 n2 bytes."""
    out = {}
    nn_bytes = 12 * n1 + 12 * n2 + 8 * n2 + 24 * n2 * (iters - 1) / iters
    mr_bytes = 16 * n2 + 8 * n1 + 24 * m
    for name, nbytes, ms in (("nn_and_apply", nn_bytes, prof["nn"] / iters), ("match_reject_reduce_solve", mr_bytes, prof["match_reduce_solve"] / iters)):
        gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        out[name] = {"bound": "hbm", "algorithmic_bytes": int(nbytes), "ms": ms, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
    out["build_ms_per_call"] = prof["build"]
    out["final_apply_ms"] = prof["final_apply"]
    if brute_ms is not None:
        tf = 8.0 * n1 * n2 / (brute_ms * 1e-3) / 1e12
        out["nn_brute"] = {"bound": "valu", "flop": 8.0 * n1 * n2, "ms": brute_ms, "achieved": tf, "peak": VALU_F32_PEAK_TF, "unit": "TFLOP/s", "frac": tf / VALU_F32_PEAK_TF}
    return out


def _time_nn(torch, native, ws, tgt, src, mode, stream, dev, reps=3):
    n1, n2 = tgt.shape[0], src.shape[0]
    idx = torch.empty(n2, dtype=torch.int32, device=dev)
    d2 = torch.empty(n2, dtype=torch.float32, device=dev)
    best = None
    for r in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        ws.nearest(tgt.data_ptr(), n1, src.data_ptr(), n2, idx.data_ptr(), d2.data_ptr(), mode, stream)
        e1.record()
        torch.cuda.synchronize()
        if r > 0:
            t = e0.elapsed_time(e1)
            best = t if best is None or t < best else best
    return best, idx, d2


def bench_icp(args, torch, native, synth, dev, stream, with_cpu):
    """configs[1]: 2 sensors x 512x424 'scene' frames, sensor 1 mis-calibrated; ICP(maxIter=10), device resident.
    configs[2] (under "config2"): 8 sensors x 512x424, target = 7 sensors, source = 1: voxel-grid NN vs brute-force NN."""
    iters = 10
    clouds = _scene_clouds_on_device(torch, synth, dev, 2)
    tgt, src0 = clouds[0], clouds[1]
    n1, n2 = tgt.shape[0], src0.shape[0]
    ws = native.IcpWorkspace(dev.index, n1, n2)
    out = {"workload": "configs[1]: 2 sensors x 512x424 scene frames, ICP(maxIter=10), device resident", "n1": n1, "n2": n2,
           "parity": "NN pinned to the reference's nanoflann fixtures; match / rejection / Kabsch steps PARITY UNPINNED (OpenCV 3.2 binaries absent), checked against the CPU restatement at 1e-4"}
    t_grid, _ = _time_icp(torch, native, ws, tgt, src0, iters, native.NN_GRID, args.icp_reps, stream, dev)
    t_brute, _ = _time_icp(torch, native, ws, tgt, src0, iters, native.NN_BRUTE, 1, stream, dev)
    out["iter_ms_grid"] = t_grid / iters
    out["iter_ms_brute"] = t_brute / iters
    out["iter_ms"] = out["iter_ms_grid"]
    # per-group roofline from the library's own events (a separate profiled run: the events cost a few microseconds per iteration)
    t_prof, prof = _time_icp(torch, native, ws, tgt, src0, iters, native.NN_GRID, 2, stream, dev, profile=True)
    m_last = int(ws.trace(iters, stream)[-1][0])
    brute_nn_ms, _, _ = _time_nn(torch, native, ws, tgt, src0, native.NN_BRUTE, stream, dev, reps=1)
    out["roofline"] = _icp_roofline(n1, n2, m_last, iters, prof, brute_ms=brute_nn_ms)
    out["profiled_iter_ms"] = t_prof / iters
    if with_cpu:
        from oracle import orc
        cores = os.cpu_count() or 1
        t0 = time.perf_counter()
        orc.icp(tgt.cpu().numpy(), src0.cpu().numpy(), max_iter=2, nn_mode="kdtree", n_threads=cores)
        out["cpu_iter_ms"] = 1e3 * (time.perf_counter() - t0) / 2
        out["cpu_cores"] = cores
        out["cpu_kind"] = "port (oracle kd-tree NN with OpenMP queries like icp.cpp:25-31, 2 iterations timed)"
        if orc.have_ref_nn():
            # the reference's OWN nearest-neighbour step (its vendored nanoflann 1.1.9 + PointCloud adaptor, compiled from the
            # reference's headers into oracle/_ref): tree build + all queries = the dominant cost of a reference ICP iteration
            t_np, s_np = tgt.cpu().numpy(), src0.cpu().numpy()
            orc.ref_nn(t_np[:1000], s_np[:1000])
            t0 = time.perf_counter()
            orc.ref_nn(t_np, s_np)
            out["cpu_reference_nn_ms"] = 1e3 * (time.perf_counter() - t0)
            out["cpu_reference_nn_kind"] = "reference (kd-tree build + OpenMP queries of icp.cpp:18-32 on the same clouds, one iteration's worth)"
    ws.close()

    # configs[2]: the refine loop's shape for one of 8 sensors (MainWindowForm.cs:349-376): target = all other sensors' clouds
    clouds = _scene_clouds_on_device(torch, synth, dev, 8)
    src8 = clouds[0]
    tgt8 = torch.cat(clouds[1:]).contiguous()
    n1, n2 = tgt8.shape[0], src8.shape[0]
    ws = native.IcpWorkspace(dev.index, n1, n2)
    c2 = {"workload": "configs[2]: 8 sensors x 512x424 scene frames, target = 7 sensors' clouds, source = sensor 0, ICP(maxIter=10), device resident; voxel-grid NN vs brute-force NN",
          "n1": n1, "n2": n2}
    t_grid, _ = _time_icp(torch, native, ws, tgt8, src8, iters, native.NN_GRID, args.icp_reps, stream, dev)
    t_brute, _ = _time_icp(torch, native, ws, tgt8, src8, iters, native.NN_BRUTE, 1, stream, dev)
    c2["iter_ms_grid"] = t_grid / iters
    c2["iter_ms_brute"] = t_brute / iters
    t_prof, prof = _time_icp(torch, native, ws, tgt8, src8, iters, native.NN_GRID, 2, stream, dev, profile=True)
    m_last = int(ws.trace(iters, stream)[-1][0])
    nn_grid_ms, gi, gd = _time_nn(torch, native, ws, tgt8, src8, native.NN_GRID, stream, dev)
    nn_brute_ms, bi, bd = _time_nn(torch, native, ws, tgt8, src8, native.NN_BRUTE, stream, dev, reps=1)
    c2["nn_step_ms_grid_unseeded"] = nn_grid_ms
    c2["nn_step_ms_brute"] = nn_brute_ms
    c2["nn_modes_identical"] = bool(torch.equal(gi, bi) and torch.equal(gd.view(torch.int32), bd.view(torch.int32)))
    c2["roofline"] = _icp_roofline(n1, n2, m_last, iters, prof, brute_ms=nn_brute_ms)
    out["config2"] = c2
    ws.close()
    return out


def bench_refine(args, native, synth, S, w, h, with_cpu):
    """refineWorker_DoWork (LiveScanServer/MainWindowForm.cs:330-410) as one native call: host clouds in, host clouds out,
    everything in between resident in HBM.  Clouds = the S sensors' cropped clouds of one scene tick (CPU-made here)."""
    from oracle import orc
    rig = synth.make_rig("scene", S, w, h, seed=4, perturb=True)
    v, counts = orc.generate_mesh_vertices(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds,
                                           n_threads=min(S, os.cpu_count() or 1))
    e = np.concatenate([[0], np.cumsum(counts)])
    xyz = np.stack([v["X"], v["Y"], v["Z"]], axis=1).astype(np.float32)
    clouds = [np.ascontiguousarray(xyz[e[i]:e[i + 1]]) for i in range(S)]
    wR = np.stack([rig.wt[12 * i + 3:12 * i + 12].reshape(3, 3) for i in range(S)])
    wt = np.stack([rig.wt[12 * i:12 * i + 3] for i in range(S)])
    refine_iters, icp_iters = 2, 10                                  # KinectSettings.cs:45-46
    native.refine(clouds, wR, wt, 1, 1)                              # warm-up (workspace allocation)
    t0 = time.perf_counter()
    native.refine(clouds, wR, wt, refine_iters, icp_iters)
    dt = time.perf_counter() - t0
    n_icp = S * refine_iters * icp_iters
    out = {"workload": f"{S} sensors x {w}x{h} scene clouds ({int(e[-1])} points), {refine_iters} refine passes x {icp_iters} ICP iterations, host clouds in/out",
           "total_ms": 1e3 * dt, "ms_per_icp_iteration": 1e3 * dt / n_icp}
    if with_cpu:
        t0 = time.perf_counter()
        orc.refine(clouds, wR, wt, n_refine_iters=1, n_icp_iters=1, nn_mode="kdtree",
                   n_threads=os.cpu_count() or 1)
        dtc = time.perf_counter() - t0
        out["cpu_port_ms_per_icp_iteration"] = 1e3 * dtc / S
        out["cpu_sample"] = f"one refine pass with one ICP iteration per sensor ({S} ICP iterations) on {os.cpu_count()} threads"
    return out

