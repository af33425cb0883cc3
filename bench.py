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
import contextlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


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


def settle(fn, sync, seconds, agree=None):
    """Clock settling, separate from --warmup: runs fn back to back for `seconds` of wall time (a fresh box starts a run at idle clocks, and
    a 20-step timed region is over in 6 ms -- before the clocks have moved).  agree (N > 1): turns this rank's "go on" into rank 0's, so
    every rank runs the same number of (collective) steps.  Returns the milliseconds actually spent."""
    t0 = time.perf_counter()
    while True:
        go = time.perf_counter() - t0 < seconds
        if agree is not None:
            go = agree(go)
        if not go:
            break
        for _ in range(8):
            fn()
        sync()
    return 1e3 * (time.perf_counter() - t0)


@contextlib.contextmanager
def leg(result, name):
    """An extra leg of the bench line must never cost the headline: a failure is recorded under its name instead."""
    try:
        yield
    except Exception as ex:  # noqa: BLE001
        result[name] = {"error": f"{type(ex).__name__}: {ex}"}


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves -- as a CHILD process (never an exec:
    nothing in this process has touched the GPU yet, and nothing will), `python -m torch.distributed.run --nproc-per-node N bench.py <same
    arguments>` on 127.0.0.1 and a free port -- relay rank 0's JSON line and leave with the child's status.  Under an existing launcher
    (WORLD_SIZE set) this is never reached."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL between processes needs on these hosts
    print(f"[bench] --gpus {args.gpus} without a launcher: starting {' '.join(cmd)}", file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in child.stdout:                                # rank 0's line is the only thing the ranks put on stdout; anything else goes by
        if out.lstrip().startswith("{"):
            line = out
        else:
            sys.stderr.write(out)
    rc = child.wait()
    if line is not None:
        sys.stdout.write(line)
        sys.stdout.flush()
    raise SystemExit(rc if rc != 0 or line is not None else 1)


def launch_probe(args):
    """LSN_BENCH_LAUNCH_PROBE=1: the ranks meet over gloo, count each other and rank 0 prints one line -- the launch path of an N > 1 run
    (self_launch or an outer launcher, rendezvous, the relay of the line) without a GPU.  tests/test_sharding_gloo.py."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    seen = torch.ones(1, dtype=torch.int32)
    dist.all_reduce(seen)
    if dist.get_rank() == 0:
        print(json.dumps({"probe": True, "n_gpus": args.gpus, "n_ranks_seen": int(seen.item()), "world_size": dist.get_world_size()}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


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
    import torch
    import torch.distributed as dist
    from livescan3d_amd import native, synth
    from livescan3d_amd.fusion import DeviceFusion

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
    from livescan3d_amd.sharding import sensor_block
    from tests.exchange_rehearsal import MergedCloudExchange   # comparison legs only (the product path is ShardedFusion / lsnShard*)
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

    # N > 1: one exchange step per step forms the merged cloud on every GPU (sensor order = rank order).  Default: the
    # all-gathers carry the survivors' inputs (5 B + a 1-bit/pixel mask) and every GPU rebuilds all vertices with the same
    # arithmetic; --exchange vertices (and rigs whose widths are not multiples of 8) all-gather the 16-byte vertices.
    # The headline N > 1 step goes through the library's own lsnShard* exports (C++ host glue + RCCL inside libNativeUtils.so);
    # `--exchange survivors-python` drives the same protocol from Python over torch.distributed instead (also what a gloo
    # rehearsal with LSN_BENCH_SHARE_GPU=1 uses: RCCL refuses two ranks on one device).
    xch = sx = whole = shard = None
    survivors_ok = multi and w % 8 == 0 and args.mode == 0
    fake_rccl = bool(os.environ.get("LSN_RCCL_LIBRARY"))   # tests/fake_rccl: the C++ step with several ranks on one GPU (rehearsal only)
    use_shard = survivors_ok and args.exchange == "survivors" and (not share or fake_rccl)
    use_sx = survivors_ok and not use_shard and args.exchange in ("survivors", "survivors-python")
    if multi:
        from livescan3d_amd.sharding import ShardedFusion
        from tests.exchange_rehearsal import SurvivorExchange
        shard_preflight = None
        if use_shard:
            # The library's own RCCL step.  ShardedFusion prepares every rank locally, lets the ranks agree that all are ready and only
            # then enters the blocking communicator set-up, so a rank that cannot prepare (e.g. librccl missing) makes EVERY rank raise
            # here; the flag below turns "any rank failed" into a collective decision to fall back to the Python-driven protocol.
            err = None
            try:
                shard = ShardedFusion(rank, world, B, [w] * S, [h] * S, dev)
                shard.set_params(intr_all, wt_all, bounds)
            except Exception as ex:  # noqa: BLE001
                err, shard = f"{type(ex).__name__}: {ex}", None
            flag = torch.tensor([1 if err else 0], dtype=torch.int32, device="cpu" if share else dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()):
                print(f"[bench rank {rank}] lsnShard* unavailable ({err}); falling back to the Python-driven survivor exchange", file=sys.stderr)
                if shard is not None:
                    shard.close()
                shard, use_shard, use_sx = None, False, True
                shard_preflight = f"unavailable: {err}"
        if use_shard and world > 1:
            # Preflight of the first real N > 1 run: one step through lsnShardStep and one through the Python-driven survivor exchange
            # (the protocol the gloo tests cover) on the same frames; offsets and one tick's cloud must agree on every rank, else all
            # ranks take the Python-driven path for the timed steps and the line says so.
            bad = 0
            try:
                whole = DeviceFusion(B, [w] * S, [h] * S, device=dev_index, mode=0)
                whole.set_params(intr_all, wt_all, bounds)
                sx = SurvivorExchange(world, fus, whole, via_host=share)
                m_v, m_o = shard.step(depth, rgb, stream)
                p_v, p_o = sx.exchange(depth, rgb, stream)
                torch.cuda.synchronize()
                n0 = int(p_o[0, -1].item())
                bad = 0 if (bool(torch.equal(m_o, p_o)) and n0 > 0 and bool(torch.equal(m_v[0, :n0], p_v[0, :n0]))) else 1
            except Exception as ex:  # noqa: BLE001
                print(f"[bench rank {rank}] shard preflight raised {type(ex).__name__}: {ex}", file=sys.stderr)
                bad = 1
            flag = torch.tensor([bad], dtype=torch.int32, device="cpu" if share else dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()):
                shard_preflight = "mismatch"
                shard.close()
                shard, use_shard, use_sx = None, False, True     # sx / whole are kept for the timed steps
            else:
                shard_preflight = "ok"
                sx = whole = None
        if use_shard:
            pass
        elif use_sx:
            if sx is None:
                whole = DeviceFusion(B, [w] * S, [h] * S, device=dev_index, mode=0)
                whole.set_params(intr_all, wt_all, bounds)
                sx = SurvivorExchange(world, fus, whole, via_host=share)
        else:
            xch = MergedCloudExchange(world, B, S_loc, fus.capacity, dev, compact=not args.padded_exchange, via_host=share)
    prof_plan = shard.shard.plan(True) if use_shard else (whole.plan if use_sx else fus.plan)
    merged_out = [None, None]

    # Two resident input sets, used alternately: a real stream brings new frames every step, so nothing a step leaves in
    # L2 / Infinity Cache may serve the next one (the same buffer every step would let the count pass hit the cache).
    depth_b, rgb_b = depth.clone(), rgb.clone()
    step_no = [0]

    def step():
        d_in, c_in = (depth, rgb) if step_no[0] & 1 == 0 else (depth_b, rgb_b)
        step_no[0] += 1
        if use_shard:
            merged_out[0], merged_out[1] = shard.step(d_in, c_in, stream)
        elif use_sx:
            sx.exchange(d_in, c_in, stream)
        else:
            fus.run(d_in, c_in)
            if xch is not None:
                xch.exchange(fus.vertices, fus.offsets)

    def sync():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def agree(go):
        flag = torch.tensor([1 if go else 0], dtype=torch.int32, device="cpu" if share else dev)
        dist.broadcast(flag, src=0)
        return bool(int(flag.item()))

    settle_ms = settle(step, sync, args.settle_seconds, agree if multi else None) if args.settle_seconds > 0 else 0.0
    step_no[0] = 0
    for _ in range(args.warmup):
        step()
    sync()
    prof_plan.profile(True)
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

    # algorithmic bytes of one launch of the dominant kernel on this rank
    if use_shard:
        moff = merged_out[1].cpu().numpy().astype(np.int64)
        off = moff[:, s0:s1 + 1] - moff[:, s0:s0 + 1]          # this rank's block inside the merged offsets
    else:
        off = (sx.offsets if use_sx else fus.offsets).cpu().numpy().astype(np.int64)
    V_local = int(off[:, -1].sum())
    if use_shard or use_sx:
        # recon_kernel rebuilds the WHOLE merged cloud on every GPU: 5 B read + 16 B written per vertex, 1 bit per pixel of mask
        V_total = int(merged_out[1][:, -1].sum().item()) if use_shard else int(sx.merged_off[:, -1].sum().item())
        alg_bytes = 21 * V_total + (B * S * P) // 8
    else:
        alg_bytes = 2 * P * S_loc * B + 19 * V_local            # fuse_kernel<1>: sum over its sensor-frames of 2P + 19V
        V_total = int(xch.merged_off[:, -1].sum().item()) if multi else V_local
    if args.mode in (1, 2) and fus.plan.lookback_failed(stream):
        raise SystemExit("look-back compaction gave up on a bounded spin: results invalid")

    # N > 1: how many ranks actually took part, as the communicators themselves report it (not what the command line asked for): every rank
    # adds a one through torch's group; the library's own RCCL communicator (lsnShard*) is asked for its ncclCommCount
    ranks_seen = None
    if multi:
        ones = torch.ones(1, dtype=torch.int32, device="cpu" if share else dev)
        dist.all_reduce(ones)
        ranks_seen = {"process_group": int(ones.item()), "library_communicator": shard.shard.ranks_seen() if use_shard else None}

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
                "parallelism": f"sensor-shard{world}" + (("+allgather(survivors; lsnShard* = C++ host glue + RCCL inside the library)" if use_shard else
                                                          "+allgather(survivors; Python over torch.distributed)" if use_sx else "+allgather(vertices)") if multi else ""),
                "bounds": [float(x) for x in bounds],
                **({"shard_preflight": shard_preflight, "rccl_library": native.shard_rccl_path() if (use_shard or shard_preflight) else None,
                    "collective_failure": "not recoverable mid-collective: the process group's timeout (LSN_BENCH_PG_TIMEOUT_S, 300 s) ends the job non-zero"}
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
            },
        }

    # ---- ablation (extra field): the same steps with the arithmetic count pass (no per-pixel depth thresholds) ------------
    if rank == 0 and not multi and args.mode == 0 and not args.core_only and os.environ.get("LSN_NO_THRESHOLDS", "0") in ("", "0"):
        with leg(result, "arithmetic_count_pass"):
            os.environ["LSN_NO_THRESHOLDS"] = "1"          # read when a plan is created
            try:
                fus_a = DeviceFusion(B, [w] * S_loc, [h] * S_loc, device=dev_index, mode=0)
            finally:
                del os.environ["LSN_NO_THRESHOLDS"]
            fus_a.set_params(intr_all[7 * s0:7 * (s0 + S_loc)], wt_all[12 * s0:12 * (s0 + S_loc)], bounds)
            for _ in range(args.warmup + 1):
                fus_a.run(depth, rgb)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fus_a.run(depth, rgb)
            torch.cuda.synchronize()
            dta = time.perf_counter() - t0
            same = bool(torch.equal(fus_a.offsets, fus.offsets))
            result["arithmetic_count_pass"] = {"value": B * args.steps / dta, "unit": "frames/s", "ms_per_step": 1e3 * dta / args.steps,
                                               "offsets_identical": same,
                                               "note": "LSN_NO_THRESHOLDS=1: the count pass re-evaluates unproject + transform + crop per pixel "
                                                       "(fuse_kernel<0>) instead of comparing the depth with the per-pixel interval"}
            del fus_a

    # ---- spatially coherent input (extra field): ray-cast scene frames instead of hash noise, and the lazy colour load ----------
    if rank == 0 and not multi and args.mode == 0 and not args.core_only:
        with leg(result, "scene_input"):
            result["scene_input"] = bench_scene_input(args, torch, synth, DeviceFusion, dev_index, S, B, w, h)

    # ---- pipelined calls (extra field): count(k+1) beside write(k) on an internal side stream -------------------------
    if rank == 0 and not multi and args.mode == 0 and not args.core_only:
        with leg(result, "pipelined"):
            fus.plan.set_pipelined(True)
            for _ in range(args.warmup + 1):
                fus.run(depth, rgb)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fus.run(depth, rgb)
            torch.cuda.synchronize()
            dtp = time.perf_counter() - t0
            ok = bool(torch.equal(fus.offsets.cpu(), torch.from_numpy(off.astype(np.int32))))
            fus.plan.set_pipelined(False)
            result["pipelined"] = {"value": B * args.steps / dtp, "unit": "frames/s", "ms_per_step": 1e3 * dtp / args.steps, "offsets_identical": ok,
                                   "note": "same steps with lsnFusionSetPipelined: the VALU-bound count pass of call k+1 overlaps the "
                                           "HBM-bound write kernel of call k (inputs resident, double-buffered scratch)"}

    # ---- streamed calls (extra field): write(k) and count(k+1) inside one kernel -----------------------------------------
    if rank == 0 and not multi and args.mode == 0 and not args.core_only:
        with leg(result, "streamed"):
            d2 = depth.clone()                       # a second resident batch, so that "next" is a different buffer
            bufs = [depth, d2]
            fus.plan.profile(True)
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
            ok = bool(torch.equal(fus.offsets.cpu(), torch.from_numpy(off.astype(np.int32))))
            result["streamed"] = {"value": B * args.steps / dts, "unit": "frames/s", "ms_per_step": 1e3 * dts / args.steps, "offsets_identical": ok,
                                  "kernel_avg_ms": ks["avg_ms"], "achieved_GBps": alg_bytes / (ks["avg_ms"] * 1e-3) / 1e9 if ks["avg_ms"] > 0 else 0.0,
                                  "note": "lsnFusionRunStreamed: one kernel writes batch k (HBM-bound) and counts the resident batch k+1 "
                                          "(VALU-bound); same work per step as the default path, no separate count launch"}
            del d2

    # ---- N > 1, extra leg: the exchange step carrying 16-byte vertices (what the survivor exchange is compared with) ------
    compare = args.compare_exchanges or os.environ.get("LSN_BENCH_FORCE_DIST") == "1" or share
    if (use_sx or use_shard) and compare and not args.no_tick_parallel:
        vx = MergedCloudExchange(world, B, S_loc, fus.capacity, dev, compact=not args.padded_exchange, via_host=share)
        # the headline step once more on `depth` / `rgb`, the inputs the comparison legs use
        if use_shard:
            m_v, m_o = shard.step(depth, rgb, stream)
        else:
            m_v, m_o = sx.exchange(depth, rgb, stream)

        def vstep():
            fus.run(depth, rgb)
            vx.exchange(fus.vertices, fus.offsets)
        for _ in range(max(1, args.warmup)):
            vstep()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            vstep()
        sync()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        same = bool(torch.equal(m_o, vx.merged_off))
        for k in (0, B - 1):
            n_chk = int(m_o[k, -1].item())
            same = same and bool(torch.equal(m_v[k, :n_chk], vx.merged[k, :n_chk]))
        if rank == 0:
            result["vertex_exchange"] = {
                "value": B * args.steps / float(el.item()), "unit": "frames/s", "scaling": "strong", "ms_per_step": 1e3 * float(el.item()) / args.steps,
                "merged_cloud_identical_to_survivor_exchange": same, "slab_vertices": vx.last_slab,
                "note": "the same step with all-gathers of the 16-byte vertices + lsnMergeShards (bench.py --exchange vertices makes it `value`)"}
            if use_shard:
                result["config"]["exchange_bytes_sent_per_rank_per_step"] = shard.shard.last_bytes_sent()
            else:
                result["config"]["exchange_slab_survivors"] = sx.last_slab
        del vx
        if use_shard and survivors_ok:
            # the same protocol driven from Python over torch.distributed (round 1's path): what moving the host glue into the library bought
            whole_p = DeviceFusion(B, [w] * S, [h] * S, device=dev_index, mode=0)
            whole_p.set_params(intr_all, wt_all, bounds)
            sxp = SurvivorExchange(world, fus, whole_p, via_host=share)
            for _ in range(max(1, args.warmup)):
                sxp.exchange(depth, rgb, stream)
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                sxp.exchange(depth, rgb, stream)
            sync()
            el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            same = bool(torch.equal(m_o, sxp.merged_off))
            if rank == 0:
                result["python_survivor_exchange"] = {
                    "value": B * args.steps / float(el.item()), "unit": "frames/s", "ms_per_step": 1e3 * float(el.item()) / args.steps,
                    "merged_offsets_identical": same,
                    "note": "the same survivor exchange driven from Python: five torch.distributed all-gathers, two staging copies and a .item() per step"}
            del sxp, whole_p

    # ---- N > 1, extra leg: the same ticks spread over the GPUs instead of the sensors (no exchange step at all) ------
    if multi and not args.no_tick_parallel:
        fus_all = fus
        d_all, c_all = depth, rgb
        if S_loc != S:
            fus_all = DeviceFusion(B, [w] * S, [h] * S, device=dev_index, mode=args.mode)
            fus_all.set_params(intr_all, wt_all, bounds)
            d_all, c_all = synth.noise_frames_torch(dev, 1, B, S, w, h, tick0=rank * B)
            d_all, c_all = d_all.view(B, S * P), c_all.view(B, S * P * 3)
        for _ in range(args.warmup):
            fus_all.run(d_all, c_all)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fus_all.run(d_all, c_all)
        sync()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        if rank == 0:
            result["tick_parallel"] = {
                "value": world * B * args.steps / float(el.item()), "unit": "frames/s", "scaling": "weak",
                "note": "every GPU fuses whole ticks (all sensors) of its own tick range: no exchange step, no collective in "
                        "the timed region; reported beside the north-star's sensor-sharded + all-gather scheme"}
            if xch is not None:
                result["config"]["exchange_slab_vertices"] = xch.last_slab
        if fus_all is not fus:
            del fus_all, d_all, c_all

    # ---- the complete merge call incl. the reference's always-on triangulation (extra field, never `value`) --------
    if rank == 0 and not multi and not args.no_mesh:
        with leg(result, "mesh"):
            cap = fus.capacity
            tri = torch.empty((B, 2 * cap, 3), dtype=torch.int32, device=dev)
            toff = torch.zeros((B, S_loc + 1), dtype=torch.int32, device=dev)

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

            rate_n, tri_n = mesh_rate(depth, rgb, fus)
            # the same on ray-cast scene frames (8 distinct ticks, repeated): coherent surfaces, ~1.6 M triangles per tick
            rigs_m = [synth.make_rig("scene", S_loc, w, h, seed=4, tick=k) for k in range(8)]
            d_m = torch.from_numpy(np.stack([rigs_m[k % 8].depth_maps.view(np.int16) for k in range(B)])).to(dev)
            c_m = torch.from_numpy(np.stack([rigs_m[k % 8].depth_colors for k in range(B)])).to(dev)
            fus_m = DeviceFusion(B, [w] * S_loc, [h] * S_loc, device=dev_index, mode=0)
            fus_m.set_params(rigs_m[0].intr, rigs_m[0].wt, rigs_m[0].bounds)
            rate_s, tri_s = mesh_rate(d_m, c_m, fus_m)
            result["mesh"] = {"frames_per_s": rate_n, "triangles_per_tick": tri_n,
                              "scene_frames": {"frames_per_s": rate_s, "triangles_per_tick": tri_s},
                              "note": "vertices + triangulation (meshGenerator.cpp) per tick on the same noise inputs; hash-noise depth "
                                      "exercises every rejection branch but yields few triangles; scene_frames: ray-cast scene frames"}
            del tri, toff, d_m, c_m, fus_m

    # ---- radial correction, the step before the merge call on every tick (extra field) ---------------------------
    if rank == 0 and not multi and not args.no_mesh:
        with leg(result, "radial_correction"):
            intr_loc = intr_all[7 * s0:7 * (s0 + S_loc)]

            def radial_ms(d_src, c_src):
                d2, c2 = d_src.clone(), c_src.clone()
                best = float("inf")
                for _ in range(4):                      # the first call builds the warp table of the calibration
                    d2.copy_(d_src); c2.copy_(c_src)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    fus.plan.radial_correct(intr_loc, d2.data_ptr(), c2.data_ptr(), stream)
                    torch.cuda.synchronize()
                    best = min(best, time.perf_counter() - t0)
                return 1e3 * best

            ms_noise = radial_ms(depth, rgb)
            # ray-cast scene frames (8 distinct ticks, repeated): coherent surfaces and invalid regions, where the hole closing
            # actually fills pixels (on hash noise it never does: no five neighbours within 30 mm of each other)
            rigs_s = [synth.make_rig("scene", S_loc, w, h, seed=4, tick=k) for k in range(8)]
            d_s = torch.from_numpy(np.stack([rigs_s[k % 8].depth_maps.view(np.int16) for k in range(B)])).to(dev)
            c_s = torch.from_numpy(np.stack([rigs_s[k % 8].depth_colors for k in range(B)])).to(dev)
            ms_scene = radial_ms(d_s, c_s)
            result["radial_correction"] = {"frames_per_s": B / (1e-3 * ms_noise), "ms_per_step": ms_noise,
                                           "scene_frames": {"frames_per_s": B / (1e-3 * ms_scene), "ms_per_step": ms_scene},
                                           "note": "depthMapAndColorSetRadialCorrection on the same ticks, HBM resident, best of 4; "
                                                   "scene_frames: the same on ray-cast scene frames"}
            del d_s, c_s

    # ---- the reference's real tick, chained: radial correction -> fusion -> triangulation (extra field) ----------------
    # LiveScanServer runs CorrectRadialDistortionsForDepthMaps and then GenerateMesh on every tick (KinectServer.cs:518-525, :354-374),
    # and the merge call always triangulates (depthprocessing.cpp:1786): `value` above is the vertices-only fusion of the named hot
    # path, this is the whole tick as one unit on HBM-resident frames.
    if rank == 0 and not multi and not args.no_mesh:
        with leg(result, "full_tick"):
            result["full_tick"] = bench_full_tick(args, torch, synth, fus, depth, rgb, intr_all[7 * s0:7 * (s0 + S_loc)], S_loc, B, w, h, dev, stream)

    # ---- outbound formats of one tick's mesh, built in HBM (extra field) -----------------------------------------
    if rank == 0 and not multi and not args.no_mesh:
        with leg(result, "wire"):
            result["wire"] = bench_wire(args, torch, native, synth, dev, stream, S, w, h, bounds, with_cpu=not args.no_cpu)

    # ---- the other BASELINE shapes, device resident (extra field): configs[4]'s 1-GPU share and one tick per call ----------------
    if rank == 0 and not multi and args.mode == 0 and not args.core_only:
        with leg(result, "shapes"):
            result["shapes"] = bench_shapes(args, torch, synth, DeviceFusion, dev, dev_index)

    # ---- the CPU side of the reference's tick (extra field): port timings for radial / mesh / whole tick, the reference's own triangulation ---
    if rank == 0 and world == 1 and not args.no_cpu and not args.no_mesh:
        with leg(result, "cpu_tick"):
            result["cpu_tick"] = cpu_tick(synth, S, w, h)
            for k_leg, k_cpu in (("mesh", "mesh_ms"), ("radial_correction", "radial_ms"), ("full_tick", "full_tick_ms")):
                if isinstance(result.get(k_leg), dict):
                    result[k_leg]["cpu_port_ms_per_tick"] = result["cpu_tick"][k_cpu]
            if isinstance(result.get("mesh"), dict) and "reference_triangulation_ms" in result["cpu_tick"]:
                result["mesh"]["cpu_reference_tri_ms_per_tick"] = result["cpu_tick"]["reference_triangulation_ms"]

    # ---- drop-in export on host buffers (PCIe-inclusive; never `value`) -----------------------------------------
    if rank == 0 and not args.no_host_path:
        with leg(result, "host_path"):
            result["host_path"] = bench_host_path(native, synth, S, w, h, bounds)
            result["host_path_frames_per_s"] = result["host_path"]["merge_noise"]["calls_per_s"]

    # ---- ICP, configs[1] ------------------------------------------------------------------------------------------
    if rank == 0 and not args.no_icp:
        with leg(result, "icp"):
            result["icp"] = bench_icp(args, torch, native, synth, dev, stream, with_cpu=(world == 1 and not args.no_cpu))
            result["icp_config2"] = result["icp"].pop("config2")
            # the second half of BASELINE.json's metric ("... + ICP iter ms") as top-level scalars
            result["icp_iter_ms"] = result["icp"]["iter_ms"]
            result["icp_iter_ms_config2"] = result["icp_config2"]["iter_ms_grid"]

    # ---- the whole pose-refinement pass (H2 / f-3): N sensors x 2 refine passes x 10 ICP iterations in one call ---------
    if rank == 0 and not multi and not args.no_icp:
        with leg(result, "refine"):
            result["refine"] = bench_refine(args, native, synth, S, w, h, with_cpu=not args.no_cpu)

    # ---- CPU baseline (rank 0, N = 1 only) ------------------------------------------------------------------------
    if rank == 0 and world == 1 and not args.no_cpu:
        with leg(result, "cpu_baseline"):
            result["cpu_baseline"] = cpu_baseline(args, synth, S, w, h, bounds)

    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        real_stdout.write(json.dumps(result) + "\n")
        real_stdout.flush()


PCIE_GBS = 63.0   # MI355X_MICROARCH.md: PCIe 5.0 x16, per direction


def bench_shapes(args, torch, synth, DeviceFusion, dev, dev_index):
    """BASELINE.json's other shapes on one GPU, device resident like the headline: configs[4]'s 1-GPU forms (16 x 1024x1024 = the whole
    rig on one GPU, 2 x 1024x1024 = its per-GPU share at 8 GPUs) and the latency case (8 x 512x424, ONE tick per call).  Per shape:
    ms per step, the write kernel's HBM fraction (HIP events inside the library) and the whole step's."""
    out = {"note": "hash-noise frames; several ticks per step: count -> scan -> write, one tick per step: the single pass (fuse_kernel<4>); "
                   "frac = (2 P + 19 V) bytes / time of the kernel named / 8 TB/s, step_frac = the same bytes / step time"}
    stream = torch.cuda.current_stream().cuda_stream
    # one-tick plans take the single pass by themselves (one launch instead of count -> scan -> write); `_two_pass`: the same plan made with
    # LSN_ONE_TICK_TWO_PASS=1 (read when a plan is created), i.e. round 4's three launches
    for name, S, w, h, T, two_pass in (("16x1024x1024_x8ticks", 16, 1024, 1024, 8, False), ("2x1024x1024_x32ticks", 2, 1024, 1024, 32, False),
                                       ("8x512x424_x1tick", 8, 512, 424, 1, False), ("8x512x424_x1tick_two_pass", 8, 512, 424, 1, True)):
        P = w * h
        rig = synth.make_rig("noise", S, w, h, seed=1, bounds=synth.CROP_BOUNDS)
        if two_pass:
            os.environ["LSN_ONE_TICK_TWO_PASS"] = "1"
        try:
            fus = DeviceFusion(T, [w] * S, [h] * S, device=dev_index, mode=0)
        finally:
            os.environ.pop("LSN_ONE_TICK_TWO_PASS", None)
        fus.set_params(rig.intr, rig.wt, rig.bounds)
        d, c = synth.noise_frames_torch(dev, 1, T, S, w, h)
        d, c = d.view(T, S * P), c.view(T, S * P * 3)
        for _ in range(4):
            fus.run(d, c)
        torch.cuda.synchronize()
        n = max(20, min(400, int(0.25 / max(1e-5, 2e-9 * T * S * P))))      # ~0.25 s of steps
        fus.plan.profile(True)
        fus.plan.kernel_stats(reset=True)
        t0 = time.perf_counter()
        for _ in range(n):
            fus.run(d, c)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        ks = fus.plan.kernel_stats(reset=True)
        fus.plan.profile(False)
        V = int(fus.offsets[:, -1].sum().item())
        alg = 2 * P * S * T + 19 * V
        out[name] = {"sensors": S, "width": w, "height": h, "ticks_per_step": T, "ms_per_step": 1e3 * dt, "frames_per_s": T / dt,
                     "kernel": ks["kernel"], "kernel_avg_ms": ks["avg_ms"], "algorithmic_bytes_per_step": alg,
                     "frac": alg / (ks["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if ks["avg_ms"] > 0 else None,
                     "step_frac": alg / dt / 1e9 / HBM_PEAK_GBS}
        del fus, d, c
        torch.cuda.empty_cache()
    return out


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


def bench_host_path(native, synth, S, w, h, bounds):
    """The reference's own exports on HOST arrays, exactly as KinectServer calls them (KinectServer.cs:354-389, 527-554): upload,
    kernels, download, deleteMesh.  PCIe-bound: every variant is set against bytes_up / 63 GB/s + bytes_down / 63 GB/s (the
    download cannot start before the upload has been consumed)."""
    import ctypes as C
    L = native.lib()
    vp = C.c_void_p
    out = {"pcie_peak_GBs_per_direction": PCIE_GBS,
           "note": "calls timed back to back from one host thread for ~1.5 s each; frac_of_pcie_bound = (bytes_up + bytes_down) / 63 GB/s / time per call, "
                   "frac_of_full_duplex_bound = max(bytes_up, bytes_down) / 63 GB/s / time per call",
           "host_path": os.environ.get("LSN_HOST_PATH", "direct"), "sensors_per_group": os.environ.get("LSN_HOST_GROUP", "by size (copies >= 1 MiB)")}

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
        while time.perf_counter() - t0 < 1.5:
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
                n, acc, t_end = 0, 0.0, time.perf_counter() + 1.5
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

            timed_tick("radial_scene", radial_only, up, up, f"depthMapAndColorSetRadialCorrection, {S} x {w}x{h} scene frames, corrected in place in the caller's arrays")
            timed_tick("tick_two_calls_scene", tick_two_calls, 2 * up, up,
                       "the reference's tick: depthMapAndColorSetRadialCorrection then generateMeshFromDepthMaps + deleteMesh (the frames cross PCIe twice on the way up)")
            timed_tick("tick_one_call_scene", tick_one_call, up, up,
                       "lsnCorrectAndGenerateMesh + deleteMesh: the same tick with one upload (corrected maps written back, vertices + triangles back)")
        run(f"merge_{kind}", rig, merge, up, f"generateMeshFromDepthMaps + deleteMesh, {S} x {w}x{h} {kind} frames, vertices + triangles back")
        if kind == "scene":
            run("vertices_only_scene", rig, singles, up, f"{S} x (generateVerticesFromDepthMap + deleteMesh), the {S} sensors of one scene tick, vertices only")
    return out


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


def bench_full_tick(args, torch, synth, fus, depth, rgb, intr_loc, S, B, w, h, dev, stream):
    """radial correction (out of place) -> unproject / transform / crop / compaction -> triangulation, launched back to back on the same
    stream for B ticks of S sensors resident in HBM; ticks per second and the split by stage (each stage alone, same inputs)."""
    cap = fus.capacity
    tri = torch.empty((B, 2 * cap, 3), dtype=torch.int32, device=dev)
    toff = torch.zeros((B, S + 1), dtype=torch.int32, device=dev)
    P = w * h
    out = {"unit": "ticks/s", "chain": "lsnFusionRadialCorrectTo -> lsnFusionRunMesh (count, scan, write, triangle count, scan, triangle write)",
           "note": "one step = B ticks through the whole chain, HBM resident in and out; stages_ms: every stage alone on the same frames"}

    def timed(fn, reps):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    n_rep = max(3, args.steps // 4)
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

        dt = timed(tick, n_rep)
        nv = float(fus.offsets[:, -1].float().mean().item())
        nt = float(toff[:, -1].float().mean().item())
        t_r, t_v, t_m = timed(radial, n_rep), timed(vertices, n_rep), timed(mesh, n_rep)
        # algorithmic bytes of the chain per sensor-frame: radial 5 B in + 5 B out per pixel; fusion 2 P + 19 V; triangulation reads the
        # corrected depth again (2 P) and writes 12 B per triangle
        alg = B * (S * P * (10 + 2 + 2) + 19 * nv + 12 * nt)
        out[kind] = {"value": B / dt, "ms_per_step": 1e3 * dt, "vertices_per_tick": nv, "triangles_per_tick": nt,
                     "stages_ms": {"radial_correction": 1e3 * t_r, "vertices": 1e3 * t_v, "vertices_and_triangles": 1e3 * t_m},
                     "algorithmic_GB_per_step": alg / 1e9, "achieved_GBps": alg / dt / 1e9, "frac_of_hbm_peak": alg / dt / 1e9 / HBM_PEAK_GBS}
        del d_corr, c_corr
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
        fus.plan.profile(True)
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


def _kernel_sources_sha256():
    import hashlib
    hsh = hashlib.sha256()
    for f in ("fusion.hip", "fusion_shared.hpp"):
        hsh.update(open(os.path.join(ROOT, "livescan3d_amd", "csrc", f), "rb").read())
    return hsh.hexdigest()


def pmc_traffic(args, S_loc, B, w, h):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/pmc_traffic.json, written by
    tools/pmc.sh on the GPU box: separate --pmc runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
    None when no pass was recorded for this exact workload."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    key = f"mode{args.mode}-{S_loc}x{w}x{h}-ticks{B}"
    rec = json.load(open(path)).get(key)
    if rec is None or rec.get("kernel_sources_sha256") != _kernel_sources_sha256():
        return None          # no pass for this workload, or the kernel has changed since: a stale counter is not a measurement
    return rec["hbm_bytes_per_launch"]


VALU_F32_PEAK_TF = 157.3    # MI355X_MICROARCH.md: dense fp32 vector peak (FMA, packed)


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


def host_description():
    """CPU model, core count and how the CPU port was built (BASELINE.md asks for these beside every CPU figure)."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    flags = "unknown"
    try:
        mk = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle", "Makefile")).read()
        flags = [l.split("=", 1)[1].strip() for l in mk.splitlines() if l.startswith("CFLAGS")][0]
    except (OSError, IndexError):
        pass
    return {"cpu_model": model, "nproc": os.cpu_count(), "compiler": "gcc " + flags}


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


if __name__ == "__main__":
    main()
