"""bench_support.multi -- what an N > 1 step of bench.py runs, and the comparison legs beside it.

The headline N > 1 step goes through the library's own lsnShard* exports (C++ host glue + RCCL inside libNativeUtils.so: ShardedFusion).
`--exchange survivors-python` drives the same protocol from Python over torch.distributed instead (also what a gloo rehearsal with
LSN_BENCH_SHARE_GPU=1 uses when no test double of RCCL is given: RCCL refuses two ranks on one device); `--exchange vertices` (and rigs
whose widths are not multiples of 8) all-gather the 16-byte vertices.  cx = the namespace bench.py's main() fills."""
import os
import sys
import threading
import time

from .exchange import MergedCloudExchange, SurvivorExchange


def call_with_timeout(fn, seconds):
    """fn() on a daemon thread: ("ok", value) | ("error", exception) | ("timeout", None) when it has not returned after `seconds`.  What a
    blocking call into a library that never comes back (a communicator rendezvous, say) needs: the caller goes on, the thread is left behind
    and the process must end with os._exit."""
    box = {}

    def run():
        try:
            box["value"] = fn()
        except BaseException as ex:  # noqa: BLE001
            box["error"] = ex

    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        return "timeout", None
    return ("error", box["error"]) if "error" in box else ("ok", box.get("value"))


class Exchange:
    def __init__(self, cx):
        args, torch, dist, B, w, h, S = cx.args, cx.torch, cx.dist, cx.B, cx.w, cx.h, cx.S
        from livescan3d_amd.sharding import ShardedFusion
        self.cx = cx
        self.shard = self.sx = self.whole = self.xch = None
        self.preflight = None
        survivors_ok = w % 8 == 0 and args.mode == 0
        fake_rccl = bool(os.environ.get("LSN_RCCL_LIBRARY"))   # tests/fake_rccl: the C++ step with several ranks on one GPU (rehearsal only)
        self.survivors_ok = survivors_ok
        self.use_shard = survivors_ok and args.exchange == "survivors" and (not cx.share or fake_rccl)
        self.use_sx = survivors_ok and not self.use_shard and args.exchange in ("survivors", "survivors-python")
        flag_dev = "cpu" if cx.share else cx.dev
        if self.use_shard:
            # The library's own RCCL step.  ShardedFusion prepares every rank locally, lets the ranks agree that all are ready and only
            # then enters the blocking communicator set-up, so a rank that cannot prepare (e.g. librccl missing) makes EVERY rank raise
            # here; the flag below turns "any rank failed" into a collective decision to fall back to the Python-driven protocol.
            # The communicator set-up itself (ncclCommInitRank inside lsnShardConnect) blocks until every rank is in it, and on a node it
            # has never run on it may not come back at all (bootstrap interface, ...): it runs on a thread of its own, and a rank that has
            # waited $LSN_BENCH_CONNECT_TIMEOUT_S for it reports that instead (the thread stays behind: the process then ends with os._exit).
            # Only that one call is on the watchdog's thread: the local preparation and the ranks' "all ready" agreement -- a collective of
            # torch's own group -- stay on this thread, so no torch.distributed call of a slow rank can still be in flight on another thread
            # when this one goes on to the all_reduce below.
            err = None
            limit = float(os.environ.get("LSN_BENCH_CONNECT_TIMEOUT_S", "120"))

            class _ConnectTimeout(Exception):
                pass

            def watched(fn):
                def on_thread():
                    torch.cuda.set_device(cx.dev)
                    fn()
                status, val = call_with_timeout(on_thread, limit)
                if status == "timeout":
                    raise _ConnectTimeout()
                if status == "error":
                    raise val
            try:
                self.shard = ShardedFusion(cx.rank, cx.world, B, [w] * S, [h] * S, cx.dev, run_connect=watched)
                self.shard.set_params(cx.intr_all, cx.wt_all, cx.bounds)
            except _ConnectTimeout:
                err = f"the library's communicator set-up did not return within {limit:.0f} s"
                cx.abandoned_thread = True
                self.shard = None      # (its handle stays with the thread that is still inside lsnShardConnect)
            except Exception as ex:  # noqa: BLE001
                err = f"{type(ex).__name__}: {ex}"
                self.shard = None
            flag = torch.tensor([1 if err else 0], dtype=torch.int32, device=flag_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()):
                print(f"[bench rank {cx.rank}] lsnShard* unavailable ({err}); falling back to the Python-driven survivor exchange", file=sys.stderr)
                if self.shard is not None:
                    self.shard.close()
                self.shard, self.use_shard, self.use_sx = None, False, True
                self.preflight = f"unavailable: {err}"
        if self.use_shard and cx.world > 1:
            # Preflight of the first real N > 1 run: one step through lsnShardStep and one through the Python-driven survivor exchange
            # (the protocol the gloo tests cover) on the same frames; offsets and one tick's cloud must agree on every rank, else all
            # ranks take the Python-driven path for the timed steps and the line says so.
            bad = 0
            try:
                self._python_survivors()
                m_v, m_o = self.shard.step(cx.depth, cx.rgb, cx.stream)
                p_v, p_o = self.sx.exchange(cx.depth, cx.rgb, cx.stream)
                torch.cuda.synchronize()
                n0 = int(p_o[0, -1].item())
                bad = 0 if (bool(torch.equal(m_o, p_o)) and n0 > 0 and bool(torch.equal(m_v[0, :n0], p_v[0, :n0]))) else 1
            except Exception as ex:  # noqa: BLE001
                print(f"[bench rank {cx.rank}] shard preflight raised {type(ex).__name__}: {ex}", file=sys.stderr)
                bad = 1
            flag = torch.tensor([bad], dtype=torch.int32, device=flag_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()):
                self.preflight = "mismatch"
                self.shard.close()
                self.shard, self.use_shard, self.use_sx = None, False, True     # sx / whole are kept for the timed steps
            else:
                self.preflight = "ok"
                self.sx = self.whole = None
        if self.use_shard:
            pass
        elif self.use_sx:
            if self.sx is None:
                self._python_survivors()
        else:
            self.xch = MergedCloudExchange(cx.world, B, cx.S_loc, cx.fus.capacity, cx.dev, compact=not args.padded_exchange, via_host=cx.share)
        self.merged = [None, None]

    def _python_survivors(self):
        cx = self.cx
        self.whole = cx.DeviceFusion(cx.B, [cx.w] * cx.S, [cx.h] * cx.S, device=cx.dev_index, mode=0)
        self.whole.set_params(cx.intr_all, cx.wt_all, cx.bounds)
        self.sx = SurvivorExchange(cx.world, cx.fus, self.whole, via_host=cx.share)

    @property
    def profiled_plan(self):
        cx = self.cx
        return self.shard.shard.plan(True) if self.use_shard else (self.whole.plan if self.use_sx else cx.fus.plan)

    def step(self, d_in, c_in):
        cx = self.cx
        if self.use_shard:
            self.merged[0], self.merged[1] = self.shard.step(d_in, c_in, cx.stream)
        elif self.use_sx:
            self.sx.exchange(d_in, c_in, cx.stream)
        else:
            cx.fus.run(d_in, c_in)
            self.xch.exchange(cx.fus.vertices, cx.fus.offsets)

    def merged_cloud(self):
        """(vertices [B, capacity, 16] u8, offsets [B, S + 1] i32) of the last step on this rank: the merged cloud of the whole rig."""
        if self.use_shard:
            return self.merged[0], self.merged[1]
        x = self.sx if self.use_sx else self.xch
        return x.merged, x.merged_off

    def ranks_seen_by_library(self):
        return self.shard.shard.ranks_seen() if self.use_shard else None

    def parallelism(self):
        return ("+allgather(survivors; lsnShard* = C++ host glue + RCCL inside the library)" if self.use_shard else
                "+allgather(survivors; Python over torch.distributed)" if self.use_sx else "+allgather(vertices)")


def _timed_collective(cx, fn):
    args, torch, dist = cx.args, cx.torch, cx.dist
    for _ in range(max(1, args.warmup)):
        fn()
    cx.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fn()
    cx.sync()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=cx.dev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    return float(el.item())


def comparison_legs(cx, ex, result):
    """The exchange step carrying 16-byte vertices, and the survivor exchange driven from Python: what the headline step is compared with.
    Collective: every rank calls this; only rank 0 has a `result` to fill."""
    args, torch, B = cx.args, cx.torch, cx.B
    vx = MergedCloudExchange(cx.world, B, cx.S_loc, cx.fus.capacity, cx.dev, compact=not args.padded_exchange, via_host=cx.share)
    # the headline step once more on `depth` / `rgb`, the inputs the comparison legs use
    m_v, m_o = ex.shard.step(cx.depth, cx.rgb, cx.stream) if ex.use_shard else ex.sx.exchange(cx.depth, cx.rgb, cx.stream)

    def vstep():
        cx.fus.run(cx.depth, cx.rgb)
        vx.exchange(cx.fus.vertices, cx.fus.offsets)
    el = _timed_collective(cx, vstep)
    same = bool(torch.equal(m_o, vx.merged_off))
    for k in (0, B - 1):
        n_chk = int(m_o[k, -1].item())
        same = same and bool(torch.equal(m_v[k, :n_chk], vx.merged[k, :n_chk]))
    if cx.rank == 0:
        result["vertex_exchange"] = {
            "value": B * args.steps / el, "unit": "frames/s", "scaling": "strong", "ms_per_step": 1e3 * el / args.steps,
            "merged_cloud_identical_to_survivor_exchange": same, "slab_vertices": vx.last_slab,
            "note": "the same step with all-gathers of the 16-byte vertices + lsnMergeShards (bench.py --exchange vertices makes it `value`)"}
        if ex.use_shard:
            result["config"]["exchange_bytes_sent_per_rank_per_step"] = ex.shard.shard.last_bytes_sent()
        else:
            result["config"]["exchange_slab_survivors"] = ex.sx.last_slab
    del vx
    if ex.use_shard and ex.survivors_ok:
        # the same protocol driven from Python over torch.distributed (round 1's path): what moving the host glue into the library bought
        whole_p = cx.DeviceFusion(B, [cx.w] * cx.S, [cx.h] * cx.S, device=cx.dev_index, mode=0)
        whole_p.set_params(cx.intr_all, cx.wt_all, cx.bounds)
        sxp = SurvivorExchange(cx.world, cx.fus, whole_p, via_host=cx.share)
        el = _timed_collective(cx, lambda: sxp.exchange(cx.depth, cx.rgb, cx.stream))
        same = bool(torch.equal(m_o, sxp.merged_off))
        if cx.rank == 0:
            result["python_survivor_exchange"] = {
                "value": B * args.steps / el, "unit": "frames/s", "ms_per_step": 1e3 * el / args.steps, "merged_offsets_identical": same,
                "note": "the same survivor exchange driven from Python: five torch.distributed all-gathers, two staging copies and a .item() per step"}
        del sxp, whole_p


def tick_parallel_leg(cx, ex, result):
    """The same ticks spread over the GPUs instead of the sensors: every GPU fuses whole ticks of its own tick range, no exchange step at all.
    Collective."""
    args, torch, dist, synth, B, S, w, h = cx.args, cx.torch, cx.dist, cx.synth, cx.B, cx.S, cx.w, cx.h
    P = w * h
    fus_all, d_all, c_all = cx.fus, cx.depth, cx.rgb
    if cx.S_loc != S:
        fus_all = cx.DeviceFusion(B, [w] * S, [h] * S, device=cx.dev_index, mode=args.mode)
        fus_all.set_params(cx.intr_all, cx.wt_all, cx.bounds)
        d_all, c_all = synth.noise_frames_torch(cx.dev, 1, B, S, w, h, tick0=cx.rank * B)
        d_all, c_all = d_all.view(B, S * P), c_all.view(B, S * P * 3)
    for _ in range(args.warmup):
        fus_all.run(d_all, c_all)
    cx.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fus_all.run(d_all, c_all)
    cx.sync()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=cx.dev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    if cx.rank == 0:
        result["tick_parallel"] = {
            "value": cx.world * B * args.steps / float(el.item()), "unit": "frames/s", "scaling": "weak",
            "note": "every GPU fuses whole ticks (all sensors) of its own tick range: no exchange step, no collective in "
                    "the timed region; reported beside the north-star's sensor-sharded + all-gather scheme"}
        if ex.xch is not None:
            result["config"]["exchange_slab_vertices"] = ex.xch.last_slab
