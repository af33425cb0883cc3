"""The multi-GPU exchange step driven from Python over torch.distributed (backend "nccl" = RCCL, or gloo on host copies).

TEST / COMPARISON INFRASTRUCTURE, not part of the package: the product path is livescan3d_amd.sharding.ShardedFusion (lsnShard*, RCCL
inside the library).  These two classes rehearse the same protocol -- contiguous sensor blocks, the all-gathers, the packing
contract of lsnMergeShards / lsnFusionReconstruct -- on CPU ranks (tests/test_sharding_gloo.py, world 2 and world 8) and serve as
comparison legs in bench.py (--exchange vertices / survivors-python, --compare-exchanges)."""
import torch
import torch.distributed as dist

from livescan3d_amd import native


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


class SurvivorExchange:
    """The same exchange step with three times fewer bytes on xGMI: a vertex is 16 bytes, what it is computed from is 5
    (u16 depth + RGB8) plus one bit per pixel.  Rank r packs its sensors' survivors as compact depth / colour streams in
    vertex order (lsnFusionPackSurvivors), five all-gathers move the offset tables, the tile prefixes, the survivor masks
    and the two streams (cut to the largest shard of the step), and every rank rebuilds ALL sensors' vertices with the
    same arithmetic straight into the merged cloud (lsnFusionReconstruct on a plan over the whole rig) -- bit-identical
    to fusing every sensor on one GPU.  Needs identically sized sensors whose width is a multiple of 8.

    local: the rank's DeviceFusion (its block of sensors); whole: a DeviceFusion over all sensors with all parameters set.
    pack_fn / recon_fn replace the two HIP entry points in CPU tests (gloo); via_host as in MergedCloudExchange."""

    def __init__(self, world, local, whole, group=None, via_host=False, pack_fn=None, recon_fn=None):
        self.world, self.local, self.whole, self.group, self.via_host = world, local, whole, group, via_host
        self.pack_fn, self.recon_fn = pack_fn, recon_fn
        self.T, self.mpr = local.n_ticks, local.n_maps
        self.cap_loc, self.tiles_loc = int(local.capacity), int(local.tiles_per_tick)
        dev = torch.device(local.device)
        T, W = self.T, world
        self.mask = torch.zeros((T, self.cap_loc // 8), dtype=torch.uint8, device=dev)
        self.depth_c = torch.zeros((T, self.cap_loc), dtype=torch.int16, device=dev)
        self.rgb_c = torch.zeros((T, self.cap_loc, 3), dtype=torch.uint8, device=dev)
        self.tile_prefix = torch.zeros((T, self.tiles_loc), dtype=torch.int32, device=dev)
        self.offsets = torch.zeros((T, self.mpr + 1), dtype=torch.int32, device=dev)
        self.g_off = torch.empty((W, T, self.mpr + 1), dtype=torch.int32, device=dev)
        self.g_tp = torch.empty((W, T, self.tiles_loc), dtype=torch.int32, device=dev)
        self.g_mask = torch.empty((W, T, self.cap_loc // 8), dtype=torch.uint8, device=dev)
        self.g_dc = torch.empty((W * T * self.cap_loc,), dtype=torch.int16, device=dev)
        self.g_cc = torch.empty((W * T * self.cap_loc * 3,), dtype=torch.uint8, device=dev)
        self.stage_d = torch.empty((T * self.cap_loc,), dtype=torch.int16, device=dev)
        self.stage_c = torch.empty((T * self.cap_loc * 3,), dtype=torch.uint8, device=dev)
        self.merged = torch.empty((T, int(whole.capacity), 16), dtype=torch.uint8, device=dev)
        self.merged_off = torch.zeros((T, W * self.mpr + 1), dtype=torch.int32, device=dev)
        self.last_slab = self.cap_loc

    def _all_gather(self, out, inp):
        if self.via_host:
            o = out.cpu()
            dist.all_gather_into_tensor(o, inp.cpu(), group=self.group)
            out.copy_(o)
        else:
            dist.all_gather_into_tensor(out, inp, group=self.group)

    def exchange(self, depth, rgb, stream=None):
        """depth [T, mpr*P] u16, rgb [T, mpr*P*3] u8 (the rank's resident inputs).  Returns (merged, merged_offsets).
        stream: HIP stream handle of the pack / reconstruct launches; default = torch's current stream, the one the staging
        copies and the collectives are ordered on."""
        T, W = self.T, self.world
        if stream is None:
            stream = int(torch.cuda.current_stream().cuda_stream) if self.pack_fn is None or self.recon_fn is None else 0
        if self.pack_fn is not None:
            self.pack_fn(depth, rgb, self.mask, self.depth_c, self.rgb_c, self.tile_prefix, self.offsets)
        else:
            self.local.plan.pack_survivors(depth.data_ptr(), rgb.data_ptr(), self.mask.data_ptr(), self.depth_c.data_ptr(), self.rgb_c.data_ptr(),
                                           self.tile_prefix.data_ptr(), self.offsets.data_ptr(), stream)
        self._all_gather(self.g_off.view(W * T, self.mpr + 1), self.offsets)
        m = max(1, int(self.g_off[:, :, self.mpr].max().item()))          # largest shard of any rank / tick
        self.last_slab = m
        self._all_gather(self.g_tp.view(W * T, self.tiles_loc), self.tile_prefix)
        self._all_gather(self.g_mask.view(W * T, self.cap_loc // 8), self.mask)
        sd = self.stage_d[: T * m].view(T, m)
        sd.copy_(self.depth_c[:, :m])
        sc = self.stage_c[: T * m * 3].view(T, m, 3)
        sc.copy_(self.rgb_c[:, :m])
        gd = self.g_dc[: W * T * m].view(W * T, m)
        gc = self.g_cc[: W * T * m * 3].view(W * T, m, 3)
        # collectives move bytes: neither RCCL nor gloo all-gathers 16-bit integers
        self._all_gather(gd.view(torch.uint8), sd.view(torch.uint8))
        self._all_gather(gc, sc)
        if self.recon_fn is not None:
            self.recon_fn(self.g_mask, gd.view(W, T, m), gc.view(W, T, m, 3), self.g_tp, self.g_off, self.merged, self.merged_off)
        else:
            self.whole.plan.reconstruct(W, self.mpr, self.g_mask.data_ptr(), gd.data_ptr(), gc.data_ptr(), m, self.g_tp.data_ptr(),
                                        self.g_off.data_ptr(), self.merged.data_ptr(), self.merged_off.data_ptr(), stream)
        return self.merged, self.merged_off
