"""bench_support.legs_device -- device-resident legs beside the headline: the other BASELINE shapes, the chained tick, scene frames, ablations."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from .common import HBM_PEAK_GBS, settle


def bench_shapes(args, torch, synth, DeviceFusion, dev, dev_index):
    """BASELINE.json's other shapes on one GPU, device resident like the headline: configs[4]'s 1-GPU forms (16 x 1024x1024 = the whole
    rig on one GPU, 2 x 1024x1024 = its per-GPU share at 8 GPUs) and the latency case (8 x 512x424, ONE tick per call).  Per shape:
    ms per step, the write kernel's HBM fraction (HIP events inside the library) and the whole step's."""
    out = {"note": "hash-noise frames, count -> scan -> write; frac = (2 P + 19 V) bytes / time of the kernel named / 8 TB/s, step_frac = the same "
                   "bytes / step time.  A one-tick plan of up to 2048 tiles takes the single pass by itself (fuse_kernel<4>: one launch); `_three_launches` is "
                   "the same plan made with LSN_ONE_TICK_SINGLE_PASS=0 (count -> scan -> write)"}
    stream = torch.cuda.current_stream().cuda_stream
    for name, S, w, h, T, single_pass in (("16x1024x1024_x8ticks", 16, 1024, 1024, 8, None), ("2x1024x1024_x32ticks", 2, 1024, 1024, 32, None),
                                          ("8x512x424_x1tick", 8, 512, 424, 1, None), ("8x512x424_x1tick_three_launches", 8, 512, 424, 1, "0")):
        P = w * h
        rig = synth.make_rig("noise", S, w, h, seed=1, bounds=synth.CROP_BOUNDS)
        if single_pass is not None:
            os.environ["LSN_ONE_TICK_SINGLE_PASS"] = single_pass
        try:
            fus = DeviceFusion(T, [w] * S, [h] * S, device=dev_index, mode=0)
        finally:
            os.environ.pop("LSN_ONE_TICK_SINGLE_PASS", None)
        fus.set_params(rig.intr, rig.wt, rig.bounds)
        d, c = synth.noise_frames_torch(dev, 1, T, S, w, h)
        d, c = d.view(T, S * P), c.view(T, S * P * 3)
        for _ in range(4):
            fus.run(d, c)
        torch.cuda.synchronize()
        n = max(20, min(400, int(0.25 / max(1e-5, 2e-9 * T * S * P))))      # ~0.25 s of steps
        # the step time WITHOUT the library's kernel timing (two event records per call: 20 instead of 13 us on the one-tick shape), then
        # the same steps again with it for the kernel's own time
        t0 = time.perf_counter()
        for _ in range(n):
            fus.run(d, c)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        fus.plan.profile(True)
        fus.plan.kernel_stats(reset=True)
        for _ in range(n):
            fus.run(d, c)
        torch.cuda.synchronize()
        ks = fus.plan.kernel_stats(reset=True)
        fus.plan.profile(False)
        V = int(fus.offsets[:, -1].sum().item())
        alg = 2 * P * S * T + 19 * V
        verified = None
        if not args.no_cpu:   # the first tick of the last timed step against the oracle (bench_support/verify.py)
            from .verify import checked, verify_clouds
            verified = checked(verify_clouds)(torch, d, c, fus.vertices, fus.offsets, [0], [w] * S, [h] * S, rig.intr, rig.wt, rig.bounds)
        out[name] = {**({"value_verified": verified} if verified is not None else {}), "sensors": S, "width": w, "height": h, "ticks_per_step": T, "ms_per_step": 1e3 * dt, "frames_per_s": T / dt,
                     "kernel": ks["kernel"], "kernel_avg_ms": ks["avg_ms"], "algorithmic_bytes_per_step": alg,
                     "frac": alg / (ks["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if ks["avg_ms"] > 0 else None,
                     "step_frac": alg / dt / 1e9 / HBM_PEAK_GBS}
        del fus, d, c
        torch.cuda.empty_cache()
    return out


def bench_full_tick(args, torch, synth, fus, depth, rgb, intr_loc, S, B, w, h, dev, stream, wt_loc=None, bounds=None):
    """radial correction (out of place) -> unproject / transform / crop / compaction -> triangulation, launched back to back on the same
    stream for B ticks of S sensors resident in HBM; ticks per second and the split by stage (each stage alone, same inputs)."""
    cap = fus.capacity
    tri = torch.empty((B, 2 * cap, 3), dtype=torch.int32, device=dev)
    toff = torch.zeros((B, S + 1), dtype=torch.int32, device=dev)
    P = w * h
    from livescan3d_amd import native
    out = {"unit": "ticks/s", "chain": "lsnTickRun = lsnFusionRadialCorrectTo -> lsnFusionRunMesh (count, scan, write, triangle count, scan, triangle write)",
           "note": "one step = B ticks through the whole chain, HBM resident in and out; value: the chain as one call (two halves of the batch side by side "
                   "on two streams); one_plan: the two calls on one plan and one stream (what rounds 3-5 reported); stages_ms: every stage alone on the same frames"}

    def timed(fn, reps):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    n_rep = max(24, args.steps // 4)   # (a step is ~1.5 ms: five repetitions, as the driver's --steps 20 used to give, read 3-4 % below settled clocks)
    for kind in ("noise", "scene"):
        if kind == "noise":
            d_in, c_in = depth, rgb
        else:
            rigs = [synth.make_rig("scene", S, w, h, seed=4, tick=k) for k in range(8)]
            d_in = torch.from_numpy(np.stack([rigs[k % 8].depth_maps.view(np.int16) for k in range(B)])).to(dev)
            c_in = torch.from_numpy(np.stack([rigs[k % 8].depth_colors for k in range(B)])).to(dev)
        d_corr, c_corr = torch.empty_like(d_in), torch.empty_like(c_in)
        plan = fus.plan

        def radial():
            plan.radial_correct_to(intr_loc, d_in.data_ptr(), c_in.data_ptr(), d_corr.data_ptr(), c_corr.data_ptr(), stream)

        def vertices():
            plan.run(d_corr.data_ptr(), c_corr.data_ptr(), fus.vertices.data_ptr(), fus.offsets.data_ptr(), stream)

        def mesh():
            plan.run_mesh(d_corr.data_ptr(), c_corr.data_ptr(), fus.vertices.data_ptr(), fus.offsets.data_ptr(), tri.data_ptr(), toff.data_ptr(), stream)

        def tick():
            radial()
            mesh()

        # the same chain as ONE call (lsnTickRun: from 8 ticks up the batch runs as two halves side by side on two streams) -- this is `value`;
        # the two calls on one plan and one stream are reported beside it (one_plan)
        tp = native.TickPipeline(dev.index, B, [w] * S, [h] * S)
        tp.set_params(intr_loc, wt_loc, bounds)

        def tick_one_call():
            tp.run(d_in.data_ptr(), c_in.data_ptr(), d_corr.data_ptr(), c_corr.data_ptr(), fus.vertices.data_ptr(), fus.offsets.data_ptr(), tri.data_ptr(),
                   toff.data_ptr(), stream)

        dt_one_plan = timed(tick, n_rep)
        dt = timed(tick_one_call, n_rep)
        nv = float(fus.offsets[:, -1].float().mean().item())
        nt = float(toff[:, -1].float().mean().item())
        t_r, t_v, t_m = timed(radial, n_rep), timed(vertices, n_rep), timed(mesh, n_rep)
        # algorithmic bytes of the chain per sensor-frame: radial 5 B in + 5 B out per pixel; fusion 2 P + 19 V; triangulation reads the
        # corrected depth again (2 P) and writes 12 B per triangle
        alg = B * (S * P * (10 + 2 + 2) + 19 * nv + 12 * nt)
        verified = None
        if kind == "scene" and not args.no_cpu:
            # the chain once more, then tick 0 (and the last tick) of what it left against the oracle: corrected maps, cloud, offsets, triangles
            from .verify import checked, verify_mesh_tick
            verify_mesh_tick = checked(verify_mesh_tick)
            tick_one_call()
            verified = verify_mesh_tick(torch, 0, d_in, c_in, d_corr, c_corr, fus.vertices, fus.offsets, tri, toff, [w] * S, [h] * S,
                                        intr_loc, wt_loc, bounds)
            if verified["bitexact"]:
                last = verify_mesh_tick(torch, B - 1, d_in, c_in, d_corr, c_corr, fus.vertices, fus.offsets, tri, toff, [w] * S, [h] * S,
                                        intr_loc, wt_loc, bounds)
                verified = {**last, "ticks": [0, B - 1]} if last["bitexact"] else last
        out[kind] = {**({"value_verified": verified} if verified is not None else {}),
                     "value": B / dt, "ms_per_step": 1e3 * dt, "parts": tp.parts,
                     "one_plan": {"value": B / dt_one_plan, "ms_per_step": 1e3 * dt_one_plan}, "vertices_per_tick": nv, "triangles_per_tick": nt,
                     "stages_ms": {"radial_correction": 1e3 * t_r, "vertices": 1e3 * t_v, "vertices_and_triangles": 1e3 * t_m},
                     "algorithmic_GB_per_step": alg / 1e9, "achieved_GBps": alg / dt / 1e9, "frac_of_hbm_peak": alg / dt / 1e9 / HBM_PEAK_GBS}
        tp.close()
        del d_corr, c_corr
    out["kernels"] = valu_bound_kernels()
    return out


def valu_bound_kernels():
    """The two kernels of the chained tick that are bound by VALU issue, not by HBM: evidence from the committed PMC passes
    (profiles/pmc_tick_scene.json = tools/pmc_tick.sh on scene frames: separate rocprofv3 --pmc runs).  VALU-busy = SQ_ACTIVE_INST_VALU x 4
    cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); instructions per pixel = SQ_INSTS_VALU / SQ_WAVES / 8 pixels per lane (tri) or
    x 64 lanes / pixels (radial).  None while the kernel sources differ from the ones the counters were read from."""
    import hashlib
    path = os.path.join(ROOT, "profiles", "pmc_tick_scene.json")
    if not os.path.exists(path):
        return None
    rec = json.load(open(path))
    hsh = hashlib.sha256()
    for f in ("mesh.hip", "radial.hip", "fusion.hip", "fusion_shared.hpp"):
        hsh.update(open(os.path.join(ROOT, "livescan3d_amd", "csrc", f), "rb").read())
    if rec.get("_sources_sha256") != hsh.hexdigest():
        return None
    out = {"source": "profiles/pmc_tick_scene.json (tools/pmc_tick.sh scene: 16 ticks x 8 x 512x424 scene frames per launch)"}
    pixels = 16 * 8 * 512 * 424
    for name, key in (("tri_kernel<0>", "tri_kernel<0, true, false>"), ("radial_band_kernel", "radial_band_kernel<true, true>")):
        c = rec.get(key) or rec.get(key.replace(", false>", ">"))
        if not c:
            continue
        busy = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)
        out[name] = {"bound": "valu", "achieved": busy, "peak": 1.0, "unit": "fraction of VALU issue cycles busy", "frac": busy,
                     "valu_instructions_per_pixel": c["SQ_INSTS_VALU"] * 64.0 / pixels,
                     "lds_bank_conflict_share_of_lds_cycles": (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else None,
                     "hbm_bytes_per_launch": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 if "FETCH_SIZE" in c else None}   # KB counters; FETCH doubled (gfx950)
    return out


def bench_scene_input(args, torch, synth, DeviceFusion, dev_index, S, B, w, h):
    """The same step on ray-cast scene frames (8 distinct ticks, repeated): survivors are spatially coherent, as in real
    recordings -- whole regions of a frame lie outside the crop box.  Default write pass (colours fetched only by lanes that kept
    a pixel) and the eager one ($LSN_LAZY_RGB=0: colours fly together with the depth, rejected areas included)."""
    rigs = [synth.make_rig("scene", S, w, h, seed=4, tick=k, perturb=True) for k in range(8)]
    depth = torch.from_numpy(np.stack([rigs[k % 8].depth_maps.view(np.int16) for k in range(B)])).cuda()
    rgb = torch.from_numpy(np.stack([rigs[k % 8].depth_colors for k in range(B)])).cuda()
    depth_b, rgb_b = depth.clone(), rgb.clone()
    P = w * h
    out = {"workload": f"{S} x {w}x{h} ray-cast scene frames per tick, {B} ticks per step"}
    ref_off = None
    plans = {}
    for name, lazy in (("default", True), ("eager_rgb", False)):
        if not lazy:
            os.environ["LSN_LAZY_RGB"] = "0"      # read when a plan is created
        try:
            fus = DeviceFusion(B, [w] * S, [h] * S, device=dev_index, mode=0)
        finally:
            os.environ.pop("LSN_LAZY_RGB", None)
        fus.set_params(rigs[0].intr, rigs[0].wt, rigs[0].bounds)
        fus.plan.thresholds(copy=False)
        plans[name] = fus
    # Both variants are timed twice, in the order default, eager, eager, default, each time after its own settling phase, and the better
    # run of each counts: with a 20-step timed region the first variant measured would otherwise read colder clocks than the second
    # (round 2's driver line: 215.3 k vs 214.7 k, where the 1000-step run of the same code read 247 k vs 218 k).
    runs = {"default": [], "eager_rgb": []}
    i_run = [0]
    for name in ("default", "eager_rgb", "eager_rgb", "default"):
        fus = plans[name]

        def one():
            i_run[0] += 1
            fus.run(depth if i_run[0] & 1 else depth_b, rgb if i_run[0] & 1 else rgb_b)
        settle(one, torch.cuda.synchronize, min(0.25, args.settle_seconds))
        for _ in range(args.warmup + 2):
            one()
        torch.cuda.synchronize()
        fus.plan.profile(True, every=4 if args.steps >= 16 else 1)   # (as in the headline region: event records on every 4th step)
        fus.plan.kernel_stats(reset=True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        ks = fus.plan.kernel_stats(reset=True)
        fus.plan.profile(False)
        runs[name].append((dt, ks["avg_ms"]))
    for name in ("default", "eager_rgb"):
        fus = plans[name]
        dt, k_ms = min(runs[name])
        off = fus.offsets.cpu().numpy().astype(np.int64)
        V = int(off[:, -1].sum())
        alg = 2 * P * S * B + 19 * V
        if ref_off is None:
            ref_off, ref_v = off, fus.vertices[0, :int(off[0, -1])].clone()
            same = True
        else:
            same = bool(np.array_equal(off, ref_off)) and bool(torch.equal(fus.vertices[0, :int(off[0, -1])], ref_v))
        out[name] = {"value": B / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt, "survivor_fraction": V / float(B * S * P),
                     "kernel_avg_ms": k_ms, "algorithmic_bytes_per_launch": alg,
                     "kernel_achieved_GBps": alg / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0,
                     "step_achieved_GBps": alg / dt / 1e9, "identical_to_default": same,
                     "both_runs_ms_per_step": [1e3 * r[0] for r in runs[name]]}
    plans.clear()
    return out



# ---- legs that reuse the headline's plan and resident inputs; cx = the namespace bench.py's main() fills -----------------------------

def leg_arithmetic_count_pass(cx):
    """The same steps with the arithmetic count pass (LSN_NO_THRESHOLDS=1 when the plan is created: no per-pixel depth thresholds)."""
    args, torch, B, w, h = cx.args, cx.torch, cx.B, cx.w, cx.h
    os.environ["LSN_NO_THRESHOLDS"] = "1"          # read when a plan is created
    try:
        fus_a = cx.DeviceFusion(B, [w] * cx.S_loc, [h] * cx.S_loc, device=cx.dev_index, mode=0)
    finally:
        del os.environ["LSN_NO_THRESHOLDS"]
    fus_a.set_params(cx.intr_loc, cx.wt_loc, cx.bounds)
    for _ in range(args.warmup + 1):
        fus_a.run(cx.depth, cx.rgb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fus_a.run(cx.depth, cx.rgb)
    torch.cuda.synchronize()
    dta = time.perf_counter() - t0
    same = bool(torch.equal(fus_a.offsets, cx.fus.offsets))
    return {"value": B * args.steps / dta, "unit": "frames/s", "ms_per_step": 1e3 * dta / args.steps, "offsets_identical": same,
            "note": "LSN_NO_THRESHOLDS=1: the count pass re-evaluates unproject + transform + crop per pixel "
                    "(fuse_kernel<0>) instead of comparing the depth with the per-pixel interval"}


def leg_pipelined(cx):
    """count(k+1) beside write(k) on an internal side stream (lsnFusionSetPipelined)."""
    args, torch, B, fus = cx.args, cx.torch, cx.B, cx.fus
    fus.plan.set_pipelined(True)
    try:
        for _ in range(args.warmup + 1):
            fus.run(cx.depth, cx.rgb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fus.run(cx.depth, cx.rgb)
        torch.cuda.synchronize()
        dtp = time.perf_counter() - t0
        ok = bool(torch.equal(fus.offsets.cpu(), torch.from_numpy(cx.off.astype(np.int32))))
    finally:
        fus.plan.set_pipelined(False)
    return {"value": B * args.steps / dtp, "unit": "frames/s", "ms_per_step": 1e3 * dtp / args.steps, "offsets_identical": ok,
            "note": "same steps with lsnFusionSetPipelined: the VALU-bound count pass of call k+1 overlaps the "
                    "HBM-bound write kernel of call k (inputs resident, double-buffered scratch)"}


def leg_streamed(cx):
    """write(k) and count(k+1) inside one kernel (lsnFusionRunStreamed)."""
    args, torch, B, fus, depth, rgb, stream = cx.args, cx.torch, cx.B, cx.fus, cx.depth, cx.rgb, cx.stream
    d2 = depth.clone()                       # a second resident batch, so that "next" is a different buffer
    bufs = [depth, d2]
    fus.plan.profile(True, every=4 if args.steps >= 16 else 1)
    fus.plan.kernel_stats(reset=True)

    def sstep(i):
        fus.plan.run_streamed(bufs[i & 1].data_ptr(), rgb.data_ptr(), fus.vertices.data_ptr(), fus.offsets.data_ptr(),
                              bufs[(i + 1) & 1].data_ptr(), stream)
    for i in range(args.warmup + 1):
        sstep(i)
    torch.cuda.synchronize()
    fus.plan.kernel_stats(reset=True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        sstep(i + args.warmup + 1)
    torch.cuda.synchronize()
    dts = time.perf_counter() - t0
    ks = fus.plan.kernel_stats(reset=True)
    fus.plan.profile(False)
    ok = bool(torch.equal(fus.offsets.cpu(), torch.from_numpy(cx.off.astype(np.int32))))
    return {"value": B * args.steps / dts, "unit": "frames/s", "ms_per_step": 1e3 * dts / args.steps, "offsets_identical": ok,
            "kernel_avg_ms": ks["avg_ms"], "achieved_GBps": cx.alg_bytes / (ks["avg_ms"] * 1e-3) / 1e9 if ks["avg_ms"] > 0 else 0.0,
            "note": "lsnFusionRunStreamed: one kernel writes batch k (HBM-bound) and counts the resident batch k+1 "
                    "(VALU-bound); same work per step as the default path, no separate count launch"}


def _scene_batch(cx, n_sensors):
    """B ticks of ray-cast scene frames (8 distinct ticks, repeated) on the device + the rig of tick 0."""
    torch, synth, B, w, h = cx.torch, cx.synth, cx.B, cx.w, cx.h
    rigs = [synth.make_rig("scene", n_sensors, w, h, seed=4, tick=k) for k in range(8)]
    d = torch.from_numpy(np.stack([rigs[k % 8].depth_maps.view(np.int16) for k in range(B)])).to(cx.dev)
    c = torch.from_numpy(np.stack([rigs[k % 8].depth_colors for k in range(B)])).to(cx.dev)
    return rigs[0], d, c


def leg_mesh(cx):
    """The complete merge call incl. the reference's always-on triangulation, device resident (never `value`)."""
    args, torch, B, fus, stream = cx.args, cx.torch, cx.B, cx.fus, cx.stream
    cap = fus.capacity
    tri = torch.empty((B, 2 * cap, 3), dtype=torch.int32, device=cx.dev)
    toff = torch.zeros((B, cx.S_loc + 1), dtype=torch.int32, device=cx.dev)

    def mesh_rate(d_in, c_in, plan_obj):
        def mesh_step():
            plan_obj.plan.run_mesh(d_in.data_ptr(), c_in.data_ptr(), plan_obj.vertices.data_ptr(), plan_obj.offsets.data_ptr(), tri.data_ptr(),
                                   toff.data_ptr(), stream)
        for _ in range(2):
            mesh_step()
        torch.cuda.synchronize()
        n_rep = max(3, args.steps // 4)
        t0 = time.perf_counter()
        for _ in range(n_rep):
            mesh_step()
        torch.cuda.synchronize()
        return B * n_rep / (time.perf_counter() - t0), float(toff[:, -1].float().mean().item())

    rate_n, tri_n = mesh_rate(cx.depth, cx.rgb, fus)
    # the same on ray-cast scene frames (8 distinct ticks, repeated): coherent surfaces, ~1.6 M triangles per tick
    rig0, d_m, c_m = _scene_batch(cx, cx.S_loc)
    fus_m = cx.DeviceFusion(B, [cx.w] * cx.S_loc, [cx.h] * cx.S_loc, device=cx.dev_index, mode=0)
    fus_m.set_params(rig0.intr, rig0.wt, rig0.bounds)
    rate_s, tri_s = mesh_rate(d_m, c_m, fus_m)
    return {"frames_per_s": rate_n, "triangles_per_tick": tri_n,
            "scene_frames": {"frames_per_s": rate_s, "triangles_per_tick": tri_s},
            "note": "vertices + triangulation (meshGenerator.cpp) per tick on the same noise inputs; hash-noise depth "
                    "exercises every rejection branch but yields few triangles; scene_frames: ray-cast scene frames"}


def leg_radial(cx):
    """depthMapAndColorSetRadialCorrection, the step before the merge call on every tick, device resident."""
    torch, B, fus, stream = cx.torch, cx.B, cx.fus, cx.stream

    def radial_ms(d_src, c_src):
        d2, c2 = d_src.clone(), c_src.clone()
        best = float("inf")
        for _ in range(4):                      # the first call builds the warp table of the calibration
            d2.copy_(d_src); c2.copy_(c_src)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fus.plan.radial_correct(cx.intr_loc, d2.data_ptr(), c2.data_ptr(), stream)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return 1e3 * best

    ms_noise = radial_ms(cx.depth, cx.rgb)
    # ray-cast scene frames (8 distinct ticks, repeated): coherent surfaces and invalid regions, where the hole closing
    # actually fills pixels (on hash noise it never does: no five neighbours within 30 mm of each other)
    _, d_s, c_s = _scene_batch(cx, cx.S_loc)
    ms_scene = radial_ms(d_s, c_s)
    return {"frames_per_s": B / (1e-3 * ms_noise), "ms_per_step": ms_noise,
            "scene_frames": {"frames_per_s": B / (1e-3 * ms_scene), "ms_per_step": ms_scene},
            "note": "depthMapAndColorSetRadialCorrection on the same ticks, HBM resident, best of 4; "
                    "scene_frames: the same on ray-cast scene frames"}
