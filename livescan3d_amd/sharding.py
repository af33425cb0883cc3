"""One block of sensors per GPU + all-gather of the merged cloud (torch.distributed; backend "nccl" = RCCL over xGMI).

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


class MergedCloudExchange:
    """Buffers + the exchange step for T ticks, `maps_per_rank` sensors per rank, `shard_cap` vertices per rank-tick.

    compact=True (default) first all-gathers the (tiny) offset tables, reads the largest per-tick shard count M back to
    the host (one synchronisation per step) and then moves slabs of M instead of shard_cap vertices per tick: the crop
    usually keeps ~half of the pixels, so about half of the xGMI traffic disappears.  compact=False moves the padded
    slabs and never touches the host.  via_host=True runs the collectives on host copies (gloo rehearsal on a box
    without one GPU per rank); it is not a product path."""

    def __init__(self, world, n_ticks, maps_per_rank, shard_cap, device, merge_fn=None, group=None, compact=True, via_host=False):
        self.world, self.n_ticks, self.mpr, self.shard_cap = world, n_ticks, maps_per_rank, int(shard_cap)
        self.device = torch.device(device)
        self.group = group
        self.merge_fn = merge_fn
        self.compact = compact
        self.via_host = via_host
        if self.device.type != "cuda" and merge_fn is None:
            raise native.NativeUtilsError("MergedCloudExchange on a non-GPU device needs an explicit merge_fn (tests only); "
                                          "the product path packs the shards with the HIP kernel lsnMergeShards")
        self.g_flat = torch.empty((world * n_ticks * self.shard_cap, 16), dtype=torch.uint8, device=self.device)
        self.stage = torch.empty((n_ticks * self.shard_cap, 16), dtype=torch.uint8, device=self.device) if compact else None
        self.g_off = torch.empty((world, n_ticks, maps_per_rank + 1), dtype=torch.int32, device=self.device)
        self.merged = torch.empty((n_ticks, self.shard_cap * world, 16), dtype=torch.uint8, device=self.device)
        self.merged_off = torch.zeros((n_ticks, world * maps_per_rank + 1), dtype=torch.int32, device=self.device)
        self.last_slab = self.shard_cap

    def _all_gather(self, out, inp):
        # output = the rank slabs concatenated along dim 0 (the layout both RCCL and gloo accept)
        if self.via_host:
            o = out.cpu()
            dist.all_gather_into_tensor(o, inp.cpu(), group=self.group)
            out.copy_(o)
        else:
            dist.all_gather_into_tensor(out, inp, group=self.group)

    def exchange(self, local_vertices, local_offsets):
        """local_vertices [T, shard_cap, 16] u8, local_offsets [T, maps_per_rank+1] i32 (lsnFusionRun outputs).
        Returns (merged [T, world*shard_cap, 16], merged_offsets [T, S+1]); asynchronous on the current stream
        (compact=True synchronises once to learn the slab size)."""
        T, W = self.n_ticks, self.world
        assert tuple(local_vertices.shape) == (T, self.shard_cap, 16)
        assert tuple(local_offsets.shape) == (T, self.mpr + 1)
        self._all_gather(self.g_off.view(W * T, self.mpr + 1), local_offsets)
        m = self.shard_cap
        src = local_vertices
        if self.compact:
            m = max(1, int(self.g_off[:, :, self.mpr].max().item()))          # largest shard of any rank / tick
            src = self.stage[: T * m].view(T, m, 16)
            src.copy_(local_vertices[:, :m])                                  # strided slabs -> one contiguous block
        self.last_slab = m
        g = self.g_flat[: W * T * m].view(W * T, m, 16)
        self._all_gather(g, src)
        if self.merge_fn is not None:
            self.merge_fn(g.view(W, T, m, 16), self.g_off, self.merged, self.merged_off)
        else:
            native.merge_shards(self.device.index, W, T, self.mpr, g.data_ptr(), m, self.g_off.data_ptr(), self.merged.data_ptr(),
                                self.shard_cap * W, self.merged_off.data_ptr(), int(torch.cuda.current_stream().cuda_stream))
        return self.merged, self.merged_off
