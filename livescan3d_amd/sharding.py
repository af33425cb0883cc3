"""One block of sensors per GPU + all-gather of the merged cloud.

ShardedFusion is a thin caller of the library's lsnShard* exports (C++ host glue + RCCL inside libNativeUtils.so,
include/NativeUtils.h part 2b); torch.distributed is only used to hand rank 0's 128-byte RCCL id to the other ranks and to agree
that every rank is ready.  (The same protocol driven from Python over torch.distributed -- the CPU rehearsal of the N > 1 logic and
bench.py's comparison legs -- lives in bench_support/exchange.py; it is not part of the package.)

The reference fans createVertices out over one std::thread per sensor and concatenates the per-sensor clouds in
sensor order (src/NativeUtils/depthprocessing.cpp:708-733, formMesh :1594-1608).  Across GPUs the same structure is:
rank r fuses the contiguous sensor block [r*S/G, (r+1)*S/G) of every tick, then ONE exchange step forms the merged
cloud on every rank: an all-gather of the fixed-capacity per-rank slabs and of their offset tables, followed by a local
packing pass (lsnMergeShards) that makes every tick contiguous again, sensor order = rank order.
No other collective exists on this path; ICP calls are independent ("replicas only").
"""
import torch
import torch.distributed as dist

from . import native


def sensor_block(n_sensors, world, rank):
    """Contiguous block of sensors owned by `rank` (so that rank order is formMesh's sensor order)."""
    if n_sensors % world != 0:
        raise ValueError(f"{n_sensors} sensors cannot be split evenly over {world} GPUs")
    per = n_sensors // world
    return rank * per, (rank + 1) * per


class ShardedFusion:
    """lsnShard* (native.Shard): this rank's block of sensors in, the merged cloud of all sensors out.

    group: a torch.distributed process group used ONLY for the rendezvous (broadcast of rank 0's unique id); None with
    world == 1.  depth_local [T, mpr*P] u16-pattern, rgb_local [T, mpr*P*3] u8, resident on this rank's GPU."""

    def __init__(self, rank, world, n_ticks, widths, heights, device, group=None, run_connect=None):
        """run_connect: optional wrapper around the one blocking step (lsnShardConnect = ncclCommInitRank), called as run_connect(fn) -- a
        caller that wants a watchdog around the rendezvous runs fn on a thread of its own there; everything else (local preparation and the
        ranks' agreement over torch.distributed) runs on the calling thread whatever it is."""
        self.rank, self.world, self.n_ticks = rank, world, n_ticks
        self.device = torch.device(device)
        # 1. everything that can fail on this rank alone (argument checks, loading RCCL, device buffers); rank 0 also draws the id
        err, ident, self.shard = None, None, None
        try:
            self.shard = native.Shard(self.device.index, rank, world, None, n_ticks, widths, heights)
            if rank == 0:
                ident = native.shard_unique_id()
        except Exception as ex:  # noqa: BLE001 -- ANY failure must reach the gather below: a rank that raises here alone leaves its peers waiting
            err = f"rank {rank}: {type(ex).__name__}: {ex}"
        # 2. the ranks agree that ALL of them are ready before anybody enters the blocking ncclCommInitRank: a rank that failed in
        #    step 1 would otherwise leave its peers waiting in it forever
        reports = [(err, ident)]
        if world > 1:
            reports = [None] * world
            dist.all_gather_object(reports, (err, ident), group=group)
        errors = [e for e, _ in reports if e]
        if errors:
            if self.shard is not None:
                self.shard.close()
                self.shard = None
            raise native.NativeUtilsError("lsnShardPrepare failed on " + "; ".join(errors))
        # 3. the collective part
        if run_connect is None:
            self.shard.connect(reports[0][1])
        else:
            run_connect(lambda: self.shard.connect(reports[0][1]))
        self.n_maps = self.shard.n_maps
        self.capacity = self.shard.capacity

    def set_params(self, intr_all, wt_all, bounds):
        self.shard.set_params(intr_all, wt_all, bounds, int(torch.cuda.current_stream().cuda_stream))

    def step(self, depth_local, rgb_local, stream=None):
        """Returns (merged [T, capacity, 16] u8, merged_offsets [T, n_maps + 1] i32) as torch views of the handle's buffers
        (valid until the next step)."""
        st = int(torch.cuda.current_stream().cuda_stream) if stream is None else stream
        mv, mo = self.shard.step(depth_local.data_ptr(), rgb_local.data_ptr(), st)
        return _device_view(mv, (self.n_ticks, self.capacity, 16), torch.uint8, self.device), \
            _device_view(mo, (self.n_ticks, self.n_maps + 1), torch.int32, self.device)

    def close(self):
        if self.shard is not None:
            self.shard.close()


class _Span:
    """Just enough of the CUDA array interface for torch.as_tensor to wrap library-owned device memory without a copy."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def _device_view(ptr, shape, dtype, device):
    typestr = {torch.uint8: "|u1", torch.int32: "<i4"}[dtype]
    return torch.as_tensor(_Span(ptr, shape, typestr), device=device)
