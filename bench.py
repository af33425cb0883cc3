#!/usr/bin/env python3
"""bench.py -- fused frames/s of the LiveScan3D fusion hot path on MI355X (+ ICP iteration ms).

  python bench.py --gpus N --steps K --warmup W          (N > 1: one rank per GPU -- under torch.distributed.run, or started by
                                                          bench.py itself as a child process when no launcher is around it)

A "step" fuses `--ticks` ticks of `--sensors` synthetic 512x424 Kinect streams (depth u16 + RGB8, resident in HBM
before the timed region) into `--ticks` merged coloured clouds: unproject + R(p+t) + AABB crop + raster-order
compaction (+ for N > 1 the RCCL all-gather of the per-GPU sensor shards and the merged-cloud assembly).
value = merged clouds per second = ticks * K / wall time (max over ranks).  Sensors are sharded in contiguous blocks
over the ranks (fixed total work: "strong" scaling).  Prints ONE JSON line on rank 0.

Extra objects in the line: "roofline" (dominant kernel: algorithmic bytes 2P + 19V per sensor-frame / HIP-event
kernel time vs 8 TB/s), "cpu_baseline" (the CPU oracle = port of the reference path, timed on this host's cores on
a bounded sample, rank 0 at N = 1 only), "icp" (configs[1]: 2 sensors x 512x424, ICP(maxIter=10) ms per iteration, with
a roofline per kernel group from the library's own HIP events), "icp_config2" (configs[2]: 8 sensors, target = 7 sensors'
clouds: voxel-grid NN vs brute-force NN).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--sensors", type=int, default=8, help="streams per tick (north-star target: 8 x 512x424)")
    ap.add_argument("--ticks", type=int, default=64, help="ticks fused per step (one launch sequence)")
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--height", type=int, default=424)
    ap.add_argument("--mode", type=int, default=int(os.environ.get("LSN_FUSE_MODE", "0")), help="0 two-pass (default, fastest), 1 look-back per run of tiles, 2 single pass with look-back per tile")
    ap.add_argument("--no-icp", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-host-path", action="store_true")
    ap.add_argument("--exchange", choices=["survivors", "survivors-python", "vertices"], default="survivors",
                    help="N > 1: what the all-gathers carry (survivors: 5 B + 1 bit per pixel, rebuilt on every GPU, through the library's "
                         "lsnShard* exports; survivors-python: the same protocol driven over torch.distributed; vertices: 16 B)")
    ap.add_argument("--compare-exchanges", action="store_true",
                    help="N > 1: also time the vertex exchange and the Python-driven survivor exchange (more collectives through "
                         "torch.distributed after the headline; always on under LSN_BENCH_FORCE_DIST=1)")
    ap.add_argument("--no-mesh", action="store_true")
    ap.add_argument("--core-only", action="store_true", help="only the timed region behind `value` (for rocprofv3 summaries): no extra legs")
    ap.add_argument("--padded-exchange", action="store_true", help="N > 1: all-gather full-capacity slabs (no host sync)")
    ap.add_argument("--no-tick-parallel", action="store_true", help="N > 1: skip the extra tick-parallel (no-exchange) leg")
    ap.add_argument("--icp-reps", type=int, default=5)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--settle-seconds", type=float, default=0.5,
                    help="untimed steps run for this long before the --warmup steps so that the timed region reads settled clocks (reported as settle_ms)")
    args = ap.parse_args()
    if args.core_only:
        args.no_icp = args.no_cpu = args.no_host_path = args.no_mesh = args.no_tick_parallel = True
    return args


from bench_support.common import HBM_PEAK_GBS, leg, pmc_traffic, settle   # noqa: E402  (no torch, no HIP: safe before the ranks exist)
from bench_support.launch import launch_probe, self_launch               # noqa: E402


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    if os.environ.get("LSN_BENCH_LAUNCH_PROBE") == "1":
        return launch_probe(args)
    # stdout carries the ONE JSON line and nothing else: RCCL (version banner, "NCCL WARN ..." lines) and other native code
    # print on fd 1, so fd 1 is pointed at stderr for the whole run and the line goes out through a private copy of the real stdout
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import types
    import torch
    import torch.distributed as dist
    from livescan3d_amd import native, synth
    from livescan3d_amd.fusion import DeviceFusion
    from livescan3d_amd.sharding import sensor_block

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: libNativeUtils has no CPU path")
    # one rank per GPU; LSN_BENCH_SHARE_GPU=1 lets several ranks share GPU 0 over gloo (a control-flow rehearsal on a
    # 1-GPU box only -- RCCL refuses duplicate devices; numbers from such a run mean nothing)
    share = os.environ.get("LSN_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # LSN_BENCH_FORCE_DIST=1: take the N > 1 code path (RCCL init, exchange step, collectives) with whatever world size
    # was launched, including 1 -- the only way to drive the RCCL calls on a one-GPU box
    multi = world > 1 or os.environ.get("LSN_BENCH_FORCE_DIST") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # A rank that raises INSIDE a collective (of torch's group here, or of the library's own RCCL communicator in lsnShardStep) leaves
        # its peers waiting in it: that cannot be repaired from inside the run.  The group's timeout turns it into a non-zero exit of the
        # job instead of a hang (torch's watchdog for its own collectives; the driver's clock for the library's).
        import datetime
        pg_timeout = datetime.timedelta(seconds=int(os.environ.get("LSN_BENCH_PG_TIMEOUT_S", "300")))
        if share:
            dist.init_process_group("gloo", timeout=pg_timeout)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=pg_timeout)
    native.require_gpu()

    S, B, w, h = args.sensors, args.ticks, args.width, args.height
    try:
        s0, s1 = sensor_block(S, world, rank)   # contiguous sensor block: rank order = formMesh sensor order
    except ValueError as e:
        raise SystemExit(str(e))
    S_loc = s1 - s0
    P = w * h
    bounds = synth.CROP_BOUNDS
    intr_all = np.concatenate([synth.kinect_intrinsics(w, h)] * S)
    wt_all = np.concatenate([synth.pack_pose(*synth.ring_pose(s, S)) for s in range(S)])

    # ---- synthetic inputs, resident in HBM -------------------------------------------------------------------
    depth, rgb = synth.noise_frames_torch(dev, 1, B, S_loc, w, h, sensor0=s0)
    depth = depth.view(B, S_loc * P)
    rgb = rgb.view(B, S_loc * P * 3)

    fus = DeviceFusion(B, [w] * S_loc, [h] * S_loc, device=dev_index, mode=args.mode)
    fus.set_params(intr_all[7 * s0:7 * (s0 + S_loc)], wt_all[12 * s0:12 * (s0 + S_loc)], bounds)
    stream = int(torch.cuda.current_stream().cuda_stream)
    # calibration-time work, once per (poses, intrinsics, crop box): the per-pixel depth thresholds of the count pass.  The
    # library would build them on the second run by itself; doing it here keeps a short --warmup from putting the one-off
    # build (reported under config.threshold_build_ms_once_per_calibration) into the timed steps.
    fus.plan.thresholds(copy=False)

    def sync():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # what the legs in bench_support/ work on
    cx = types.SimpleNamespace(args=args, torch=torch, dist=dist, native=native, synth=synth, DeviceFusion=DeviceFusion, dev=dev, dev_index=dev_index,
                               stream=stream, S=S, B=B, w=w, h=h, S_loc=S_loc, s0=s0, world=world, rank=rank, share=share, multi=multi,
                               intr_all=intr_all, wt_all=wt_all, intr_loc=intr_all[7 * s0:7 * (s0 + S_loc)], wt_loc=wt_all[12 * s0:12 * (s0 + S_loc)],
                               bounds=bounds, depth=depth, rgb=rgb, fus=fus, sync=sync)

    # N > 1: one exchange step per step forms the merged cloud on every GPU (sensor order = rank order); bench_support/multi.py
    ex = None
    if multi:
        from bench_support.multi import Exchange, comparison_legs, tick_parallel_leg
        ex = Exchange(cx)
    use_shard, use_sx = bool(ex and ex.use_shard), bool(ex and ex.use_sx)
    prof_plan = ex.profiled_plan if ex else fus.plan

    # Two resident input sets, used alternately: a real stream brings new frames every step, so nothing a step leaves in
    # L2 / Infinity Cache may serve the next one (the same buffer every step would let the count pass hit the cache).
    depth_b, rgb_b = depth.clone(), rgb.clone()
    step_no = [0]

    def step():
        d_in, c_in = (depth, rgb) if step_no[0] & 1 == 0 else (depth_b, rgb_b)
        step_no[0] += 1
        if ex:
            ex.step(d_in, c_in)
        else:
            fus.run(d_in, c_in)

    def agree(go):
        flag = torch.tensor([1 if go else 0], dtype=torch.int32, device="cpu" if share else dev)
        dist.broadcast(flag, src=0)
        return bool(int(flag.item()))

    settle_ms = settle(step, sync, args.settle_seconds, agree if multi else None) if args.settle_seconds > 0 else 0.0
    step_no[0] = 0
    for _ in range(args.warmup):
        step()
    sync()
    # the dominant kernel is timed live, inside the region `value` comes from -- on every 4th step: the two event records around a launch
    # take ~2 us of stream time each (value with them around EVERY launch: 1.5 % lower than its own repeats without any)
    time_every = 4 if (args.steps >= 16 and not multi) else 1   # (N > 1: a step may be several launches of the timed kernel; every one is timed)
    prof_plan.profile(True, every=time_every)
    prof_plan.kernel_stats(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if multi:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    kstats = prof_plan.kernel_stats(reset=True)
    prof_plan.profile(False)
    thr_table, thr_build_ms = fus.plan.thresholds(copy=False)   # already built by the second warm-up run; reports its build time

    # run-to-run spread of the number above: the same K steps four more times, outside the region `value` comes from
    repeats = [B * args.steps / elapsed]
    if not multi and not args.core_only:
        for _ in range(4):
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            sync()
            repeats.append(B * args.steps / (time.perf_counter() - t0))

    # the outputs behind `value`, checked after the timed region: ticks 0, B/2 - 1 and B - 1 of the last step's merged clouds + offsets against
    # the oracle's merge of the same frames (both resident input sets hold the same frames); N > 1: the merged cloud of the WHOLE rig on rank 0
    verified = None
    if rank == 0 and not args.no_cpu:
        from bench_support.verify import checked, verify_clouds
        check_ticks = sorted({0, max(0, B // 2 - 1), B - 1})

        @checked
        def verify_value():
            if multi:
                m_v, m_o = ex.merged_cloud()
                frames = [synth.noise_frames_torch(dev, 1, 1, S, w, h, tick0=t) for t in range(B) if t in check_ticks]
                d_chk = torch.zeros((B, S * P), dtype=torch.int16, device=dev)
                c_chk = torch.zeros((B, S * P * 3), dtype=torch.uint8, device=dev)
                for t, (dd, cc) in zip(check_ticks, frames):
                    d_chk[t], c_chk[t] = dd.view(-1), cc.view(-1)
                return verify_clouds(torch, d_chk, c_chk, m_v, m_o, check_ticks, [w] * S, [h] * S, intr_all, wt_all, bounds)
            return verify_clouds(torch, depth, rgb, fus.vertices, fus.offsets, check_ticks, [w] * S_loc, [h] * S_loc, cx.intr_loc, cx.wt_loc, bounds)
        verified = verify_value()
    # algorithmic bytes of one launch of the dominant kernel on this rank
    if use_shard:
        moff = ex.merged[1].cpu().numpy().astype(np.int64)
        off = moff[:, s0:s1 + 1] - moff[:, s0:s0 + 1]          # this rank's block inside the merged offsets
    else:
        off = (ex.sx.offsets if use_sx else fus.offsets).cpu().numpy().astype(np.int64)
    V_local = int(off[:, -1].sum())
    if use_shard or use_sx:
        # recon_kernel rebuilds the WHOLE merged cloud on every GPU: 5 B read + 16 B written per vertex, 1 bit per pixel of mask
        V_total = int(ex.merged[1][:, -1].sum().item()) if use_shard else int(ex.sx.merged_off[:, -1].sum().item())
        alg_bytes = 21 * V_total + (B * S * P) // 8
    else:
        alg_bytes = 2 * P * S_loc * B + 19 * V_local            # fuse_kernel<1>: sum over its sensor-frames of 2P + 19V
        V_total = int(ex.xch.merged_off[:, -1].sum().item()) if multi else V_local
    cx.off, cx.alg_bytes = off, alg_bytes
    if args.mode in (1, 2) and fus.plan.lookback_failed(stream):
        raise SystemExit("look-back compaction gave up on a bounded spin: results invalid")

    # N > 1: how many ranks actually took part, as the communicators themselves report it (not what the command line asked for): every rank
    # adds a one through torch's group; the library's own RCCL communicator (lsnShard*) is asked for its ncclCommCount
    ranks_seen = None
    if multi:
        ones = torch.ones(1, dtype=torch.int32, device="cpu" if share else dev)
        dist.all_reduce(ones)
        ranks_seen = {"process_group": int(ones.item()), "library_communicator": ex.ranks_seen_by_library()}

    result = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        # the dominant kernel may be launched several times per step (lsnShardStep reconstructs tick group by tick group)
        launches_per_step = max(1.0, kstats["launches"] / float(args.steps))
        achieved = alg_bytes / (kstats["avg_ms"] * launches_per_step * 1e-3) / 1e9 if kstats["avg_ms"] > 0 else 0.0
        result = {
            "metric": "fused frames/s (N x 512x424 depth -> merged cloud)",
            "value": B * args.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            **({"n_ranks_seen": ranks_seen["process_group"], "n_ranks_seen_by": ranks_seen} if multi else {}),
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_ms": settle_ms,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            **({"value_verified": verified} if verified is not None else {}),
            **({"value_repeats": repeats, "value_spread_pct": 100.0 * (max(repeats) - min(repeats)) / float(np.median(repeats)),
                "value_spread_note": "the same K steps timed four more times right after the region `value` comes from (value_repeats[0] = value); "
                                     "across fresh boxes the driver command has read 210-220 k (+-2 %)"} if len(repeats) > 1 else {}),
            "config": {
                "workload": f"{S} synthetic {w}x{h} Kinect streams per tick (BASELINE configs[2]/[3], the north-star target shape), "
                            f"{B} ticks fused per step; ICP: configs[1] (2 sensors x 512x424, 10 iterations) reported under 'icp'",
                "sensors": S, "width": w, "height": h, "ticks_per_step": B,
                "sensors_per_gpu": S_loc,
                "survivor_fraction": V_total / float(B * S * P),
                "compaction": {0: "two-pass", 1: "look-back per run of tiles", 2: "single pass, look-back per tile"}[args.mode],
                "count_pass": ("arithmetic (LSN_NO_THRESHOLDS=1)" if os.environ.get("LSN_NO_THRESHOLDS", "0") not in ("", "0")
                               else "per-pixel depth thresholds"),
                "threshold_build_ms_once_per_calibration": thr_build_ms,
                "parallelism": f"sensor-shard{world}" + (ex.parallelism() if multi else ""),
                "bounds": [float(x) for x in bounds],
                **({"shard_preflight": ex.preflight, "rccl_library": native.shard_rccl_path() if (use_shard or ex.preflight) else None,
                    "collective_failure": "not recoverable mid-collective: the process group's timeout (LSN_BENCH_PG_TIMEOUT_S, 300 s) ends the job non-zero",
                    "multi_gpu_note": "NO scaling curve has been measured on real RCCL ranks (no multi-GPU node was in reach in any round); this exchange "
                                      "form rebuilds the whole merged cloud on every GPU and is expected to scale below one GPU (DESIGN.md section 7); the form "
                                      "that can win -- the host exports sharded over devices, one PCIe link per sensor block -- is LSN_HOST_DEVICES (host_path leg)"}
                   if multi else {}),
                "parity": "outputs bit-identical to the CPU restatement of the reference (tests/, -m gpu); that restatement is PARITY UNPINNED for "
                          "the depth -> cloud path, the radial correction and the non-NN part of ICP (the reference ships no fixtures and "
                          "depthprocessing.cpp / icp.cpp cannot be built here without stand-ins); nearest neighbour and triangulation are "
                          "pinned to the reference's own code (oracle/_ref, tests/golden)",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kstats["kernel"],
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                # the whole step (count + scan + write, launch boundaries included) against the same algorithmic bytes
                "step_achieved": alg_bytes / (ms_per_step * 1e-3) / 1e9,
                "step_frac": alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": None if (use_sx or use_shard) else pmc_traffic(args, S_loc, B, w, h),
                "traffic_source": "profiles/pmc_traffic.json: separate rocprofv3 --pmc passes of tools/pmc.sh over the same kernel and workload "
                                  "(2 x FETCH_SIZE + WRITE_SIZE); null when the kernel sources have changed since that pass",
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_avg_ms": kstats["avg_ms"],
                "kernel_launches": kstats["launches"],
                "kernel_launches_per_step": launches_per_step,
                "kernel_timing": f"HIP events on the launch stream around the kernel of every {time_every}. step of the timed region" if time_every > 1
                                 else "HIP events on the launch stream around the kernel of every step of the timed region",
            },
        }

    # ---- extra legs (never `value`); each is recorded under its name, a failure as {"error": ...} ---------------------------------
    solo = rank == 0 and not multi
    if solo and args.mode == 0 and not args.core_only:
        from bench_support import legs_device as dl
        if os.environ.get("LSN_NO_THRESHOLDS", "0") in ("", "0"):
            with leg(result, "arithmetic_count_pass"):
                result["arithmetic_count_pass"] = dl.leg_arithmetic_count_pass(cx)
        with leg(result, "scene_input"):    # ray-cast scene frames instead of hash noise, and the lazy colour load
            result["scene_input"] = dl.bench_scene_input(args, torch, synth, DeviceFusion, dev_index, S, B, w, h)
        with leg(result, "pipelined"):
            result["pipelined"] = dl.leg_pipelined(cx)
        with leg(result, "streamed"):
            result["streamed"] = dl.leg_streamed(cx)
    if multi and (args.compare_exchanges or os.environ.get("LSN_BENCH_FORCE_DIST") == "1" or share) and (use_sx or use_shard) and not args.no_tick_parallel:
        comparison_legs(cx, ex, result)
    if multi and not args.no_tick_parallel:
        tick_parallel_leg(cx, ex, result)
    if solo and not args.no_mesh:
        from bench_support import legs_device as dl
        from bench_support import legs_host as hl
        with leg(result, "mesh"):
            result["mesh"] = dl.leg_mesh(cx)
        with leg(result, "radial_correction"):
            result["radial_correction"] = dl.leg_radial(cx)
        # the reference's real tick, chained: CorrectRadialDistortionsForDepthMaps then GenerateMesh on every tick (KinectServer.cs:518-525,
        # :354-374), and the merge call always triangulates (depthprocessing.cpp:1786): `value` is the vertices-only fusion of the named path
        with leg(result, "full_tick"):
            result["full_tick"] = dl.bench_full_tick(args, torch, synth, fus, depth, rgb, cx.intr_loc, S_loc, B, w, h, dev, stream, wt_loc=cx.wt_loc, bounds=bounds)
        with leg(result, "wire"):           # outbound formats of one tick's mesh, built in HBM
            result["wire"] = hl.bench_wire(args, torch, native, synth, dev, stream, S, w, h, bounds, with_cpu=not args.no_cpu)
    if solo and args.mode == 0 and not args.core_only:
        from bench_support import legs_device as dl
        with leg(result, "shapes"):         # the other BASELINE shapes: configs[4]'s 1-GPU share and one tick per call
            result["shapes"] = dl.bench_shapes(args, torch, synth, DeviceFusion, dev, dev_index)
    if rank == 0 and world == 1 and not args.no_cpu and not args.no_mesh:
        from bench_support.cpu import cpu_tick
        with leg(result, "cpu_tick"):       # the CPU side of the reference's tick: port timings, the reference's own triangulation
            result["cpu_tick"] = cpu_tick(synth, S, w, h)
            for k_leg, k_cpu in (("mesh", "mesh_ms"), ("radial_correction", "radial_ms"), ("full_tick", "full_tick_ms")):
                if isinstance(result.get(k_leg), dict):
                    result[k_leg]["cpu_port_ms_per_tick"] = result["cpu_tick"][k_cpu]
            if isinstance(result.get("mesh"), dict) and "reference_triangulation_ms" in result["cpu_tick"]:
                result["mesh"]["cpu_reference_tri_ms_per_tick"] = result["cpu_tick"]["reference_triangulation_ms"]
    if rank == 0 and not args.no_host_path:
        from bench_support import legs_host as hl
        with leg(result, "host_path"):      # drop-in exports on host buffers (PCIe-inclusive; never `value`)
            result["host_path"] = hl.bench_host_path(native, synth, S, w, h, bounds)
            result["host_path_frames_per_s"] = result["host_path"]["merge_noise"]["calls_per_s"]
    if rank == 0 and not args.no_icp:
        from bench_support import legs_icp as il
        with leg(result, "icp"):            # configs[1] and configs[2]
            result["icp"] = il.bench_icp(args, torch, native, synth, dev, stream, with_cpu=(world == 1 and not args.no_cpu))
            result["icp_config2"] = result["icp"].pop("config2")
            # the second half of BASELINE.json's metric ("... + ICP iter ms") as top-level scalars
            result["icp_iter_ms"] = result["icp"]["iter_ms"]
            result["icp_iter_ms_config2"] = result["icp_config2"]["iter_ms_grid"]
        if not multi:
            with leg(result, "refine"):     # H2 / f-3: N sensors x 2 refine passes x 10 ICP iterations in one call
                result["refine"] = il.bench_refine(args, native, synth, S, w, h, with_cpu=not args.no_cpu)
    if rank == 0 and world == 1 and not args.no_cpu:
        from bench_support.cpu import cpu_baseline
        with leg(result, "cpu_baseline"):
            result["cpu_baseline"] = cpu_baseline(args, synth, S, w, h, bounds)

    abandoned = bool(getattr(cx, "abandoned_thread", False))   # a thread is still inside a library call that never returned (multi.py)
    if multi:
        dist.barrier()
        if not abandoned:
            dist.destroy_process_group()
    status = 0
    if rank == 0 and abandoned:
        result["degraded"] = ("the library's communicator set-up never returned on this node: the timed steps ran the Python-driven survivor exchange "
                              "(config.shard_preflight) and the process leaves through os._exit with a thread still inside lsnShardConnect")
    if rank == 0:
        from bench_support.verify import failures
        bad = failures(result)
        if bad:   # a number whose outputs do not match the oracle is not a measurement: the line says so and the run fails
            result["error"] = "outputs differ from the oracle: " + "; ".join(f"{path}: {rec.get('first_mismatch', rec)}" for path, rec in bad)
            status = 1
        real_stdout.write(json.dumps(result) + "\n")
        real_stdout.flush()
    if abandoned:
        sys.stderr.flush()
        os._exit(status)   # the line is out; an orderly shutdown would wait for that thread
    if status:
        sys.exit(status)


if __name__ == "__main__":
    main()
