"""The exchange step on the real stack: RCCL (torch.distributed backend "nccl") + lsnMergeShards on the MI355X.

A one-GPU box can only host one RCCL rank (RCCL refuses duplicate devices), so this drives the product's
MergedCloudExchange with world_size 1 in a child process: process-group creation on the device, both
all_gather_into_tensor calls on device tensors of the shapes/dtypes the N > 1 bench uses, the compact-slab staging and the
HIP packing kernel.  With one rank the merged cloud must equal the rank's own cloud, offsets included.  The two-rank
logic (block ownership, rank order = sensor order) is covered on CPU by tests/test_sharding_gloo.py."""
import os
import socket
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, port, compact, out):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from livescan3d_amd import synth
    from livescan3d_amd.fusion import DeviceFusion
    from bench_support.exchange import MergedCloudExchange
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    T, S, w, h = 4, 2, 512, 424
    P = w * h
    depth, rgb = synth.noise_frames_torch(dev, 1, T, S, w, h)
    fus = DeviceFusion(T, [w] * S, [h] * S, device=0)
    import numpy as np
    fus.set_params(np.concatenate([synth.kinect_intrinsics(w, h)] * S),
                   np.concatenate([synth.pack_pose(*synth.ring_pose(s, S)) for s in range(S)]), synth.CROP_BOUNDS)
    fus.run(depth.view(T, S * P), rgb.view(T, S * P * 3))
    xch = MergedCloudExchange(1, T, S, fus.capacity, dev, compact=compact)
    merged, merged_off = xch.exchange(fus.vertices, fus.offsets)
    torch.cuda.synchronize()
    ok = bool(torch.equal(merged_off, fus.offsets))
    off = fus.offsets.cpu()
    for k in range(T):
        n = int(off[k, -1])
        ok = ok and n > 0 and bool(torch.equal(merged[k, :n], fus.vertices[k, :n]))
    ok = ok and (xch.last_slab < fus.capacity if compact else xch.last_slab == fus.capacity)
    dist.barrier()
    dist.destroy_process_group()
    with open(out, "w") as f:
        f.write("ok" if ok else "mismatch")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("compact", [True, False])
def test_rccl_exchange_world1(gpu, tmp_path, compact):
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(_free_port(), compact, out), nprocs=1, join=True)
    assert open(out).read() == "ok"


@pytest.mark.parametrize("S,G,w,h", [(4, 2, 512, 424), (8, 8, 64, 48), (6, 3, 128, 96),
                                     (3, 3, 16, 7), (8, 2, 24, 33), (5, 5, 72, 5), (4, 1, 104, 61), (6, 2, 8, 200), (7, 7, 40, 9), (2, 2, 1000, 3)])
def test_survivor_exchange_kernels_on_one_gpu(gpu, S, G, w, h):
    """The survivor exchange with the all-gather played by hand: every shard packs its sensors (compact depth / colour
    streams + survivor mask), the arrays are laid out like an all-gather result, and the whole-rig plan reconstructs
    the merged cloud -- which must equal fusing all sensors in one plan (and therefore the oracle) bit for bit."""
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from livescan3d_amd import synth
    from livescan3d_amd.fusion import DeviceFusion
    from oracle import orc
    T, mpr, P = 3, S // G, w * h
    rigs = [synth.make_rig("noise" if k % 2 else "scene", S, w, h, seed=23, tick=k, bounds=synth.CROP_BOUNDS) for k in range(T)]
    intr, wt, bounds = rigs[0].intr, rigs[0].wt, rigs[0].bounds
    depth_all = torch.from_numpy(np.stack([r.depth_maps.view(np.int16) for r in rigs])).cuda()          # [T, S*P]
    rgb_all = torch.from_numpy(np.stack([r.depth_colors for r in rigs])).cuda()                          # [T, S*P*3]
    st = int(torch.cuda.current_stream().cuda_stream)
    whole = DeviceFusion(T, [w] * S, [h] * S)
    whole.set_params(intr, wt, bounds)
    cap_loc = mpr * P
    tiles_loc = None
    masks, dcs, ccs, tps, offs, runs = [], [], [], [], [], []
    for r in range(G):
        s0, s1 = r * mpr, (r + 1) * mpr
        local = DeviceFusion(T, [w] * mpr, [h] * mpr)
        local.set_params(intr[7 * s0:7 * s1], wt[12 * s0:12 * s1], bounds)
        d = depth_all[:, s0 * P:s1 * P].contiguous()
        c = rgb_all[:, 3 * s0 * P:3 * s1 * P].contiguous()
        tiles_loc = local.plan.tiles_per_tick
        mask = torch.zeros((T, cap_loc // 8), dtype=torch.uint8, device="cuda")
        dc = torch.zeros((T, cap_loc), dtype=torch.int16, device="cuda")
        cc = torch.zeros((T, cap_loc, 3), dtype=torch.uint8, device="cuda")
        tp = torch.zeros((T, tiles_loc), dtype=torch.int32, device="cuda")
        off = torch.zeros((T, mpr + 1), dtype=torch.int32, device="cuda")
        for _ in range(2 if r == 0 else 1):                      # shard 0 twice: the second pack counts from the depth thresholds
            local.plan.pack_survivors(d.data_ptr(), c.data_ptr(), mask.data_ptr(), dc.data_ptr(), cc.data_ptr(), tp.data_ptr(), off.data_ptr(), st)
        masks.append(mask); dcs.append(dc); ccs.append(cc); tps.append(tp); offs.append(off)
        # the layout lsnShardStep sends: all ticks of the shard back to back, one contiguous run
        rdc = torch.zeros((T * cap_loc,), dtype=torch.int16, device="cuda")
        rcc = torch.zeros((T * cap_loc, 3), dtype=torch.uint8, device="cuda")
        rmask, rtp, roff = torch.zeros_like(mask), torch.zeros_like(tp), torch.zeros_like(off)
        tb = torch.zeros((T,), dtype=torch.int32, device="cuda")
        local.plan.pack_survivors_run(d.data_ptr(), c.data_ptr(), rmask.data_ptr(), rdc.data_ptr(), rcc.data_ptr(), rtp.data_ptr(), roff.data_ptr(),
                                      tb.data_ptr(), st)
        runs.append((rdc, rcc, rmask, rtp, roff, tb))
    torch.cuda.synchronize()
    g_off = torch.stack(offs)                                     # [G, T, mpr+1]
    m = int(g_off[:, :, mpr].max().item())
    assert 0 < m < cap_loc
    g_mask, g_tp = torch.stack(masks), torch.stack(tps)
    g_dc = torch.stack([x[:, :m] for x in dcs]).contiguous()     # [G, T, m]
    g_cc = torch.stack([x[:, :m] for x in ccs]).contiguous()     # [G, T, m, 3]
    merged = torch.zeros((T, whole.capacity, 16), dtype=torch.uint8, device="cuda")
    moff = torch.zeros((T, S + 1), dtype=torch.int32, device="cuda")
    whole.plan.reconstruct(G, mpr, g_mask.data_ptr(), g_dc.data_ptr(), g_cc.data_ptr(), m, g_tp.data_ptr(), g_off.data_ptr(),
                           merged.data_ptr(), moff.data_ptr(), st)
    want_v, want_o = whole.run(depth_all, rgb_all)
    torch.cuda.synchronize()
    assert torch.equal(moff, want_o)
    for k in range(T):
        n = int(want_o[k, -1])
        assert n > 0 and torch.equal(merged[k, :n], want_v[k, :n]), f"tick {k}"
        ov, _ = orc.generate_mesh_vertices(rigs[k].depth_maps, rigs[k].depth_colors, rigs[k].widths, rigs[k].heights, intr, wt, bounds)
        assert merged[k, :n].cpu().numpy().tobytes() == ov.tobytes(), f"tick {k} vs oracle"
    # the same with the run layout: identical masks / prefixes / offsets, tick bases = prefix sums of the tick totals, and the
    # whole-rig plan rebuilds the same merged cloud from [G][run_len] streams
    for r in range(G):
        rdc, rcc, rmask, rtp, roff, tb = runs[r]
        assert torch.equal(rmask, masks[r]) and torch.equal(rtp, tps[r]) and torch.equal(roff, offs[r])
        tot = offs[r][:, mpr].cpu().numpy()
        assert list(tb.cpu().numpy()) == list(np.concatenate([[0], np.cumsum(tot)[:-1]]))
    run_len = (max(int(o[:, mpr].sum()) for o in offs) + 7) & ~7
    assert run_len < T * cap_loc
    g_rdc = torch.stack([x[0][:run_len] for x in runs]).contiguous()
    g_rcc = torch.stack([x[1][:run_len] for x in runs]).contiguous()
    merged2 = torch.zeros_like(merged)
    moff2 = torch.zeros_like(moff)
    scratch = torch.zeros((G, T), dtype=torch.int32, device="cuda")
    whole.plan.reconstruct_run(G, mpr, g_mask.data_ptr(), g_rdc.data_ptr(), g_rcc.data_ptr(), run_len, g_tp.data_ptr(), g_off.data_ptr(),
                               merged2.data_ptr(), moff2.data_ptr(), scratch.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(moff2, want_o)
    for k in range(T):
        n = int(want_o[k, -1])
        assert torch.equal(merged2[k, :n], want_v[k, :n]), f"run layout, tick {k}"


def _survivor_worker(rank, port, out):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import torch.distributed as dist
    from livescan3d_amd import synth
    from livescan3d_amd.fusion import DeviceFusion
    from bench_support.exchange import SurvivorExchange
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    T, S, w, h = 4, 2, 512, 424
    P = w * h
    depth, rgb = synth.noise_frames_torch(dev, 1, T, S, w, h)
    depth, rgb = depth.view(T, S * P), rgb.view(T, S * P * 3)
    intr = np.concatenate([synth.kinect_intrinsics(w, h)] * S)
    wt = np.concatenate([synth.pack_pose(*synth.ring_pose(s, S)) for s in range(S)])
    local = DeviceFusion(T, [w] * S, [h] * S, device=0)
    local.set_params(intr, wt, synth.CROP_BOUNDS)
    whole = DeviceFusion(T, [w] * S, [h] * S, device=0)
    whole.set_params(intr, wt, synth.CROP_BOUNDS)
    xch = SurvivorExchange(1, local, whole)
    st = int(torch.cuda.current_stream().cuda_stream)
    merged, merged_off = xch.exchange(depth, rgb, st)
    want_v, want_o = whole.run(depth, rgb)
    torch.cuda.synchronize()
    ok = bool(torch.equal(merged_off, want_o)) and xch.last_slab < local.capacity
    for k in range(T):
        n = int(want_o[k, -1])
        ok = ok and n > 0 and bool(torch.equal(merged[k, :n], want_v[k, :n]))
    dist.barrier()
    dist.destroy_process_group()
    with open(out, "w") as f:
        f.write("ok" if ok else "mismatch")


def test_rccl_survivor_exchange_world1(gpu, tmp_path):
    """The survivor exchange through RCCL (one rank): the five all-gathers on device tensors (uint8 / int32 only --
    collectives do not move 16-bit integers) + pack and reconstruct kernels; the result must equal a plain fusion."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    mp.spawn(_survivor_worker, args=(_free_port(), out), nprocs=1, join=True)
    assert open(out).read() == "ok"


def _shard_worker(rank, padded, chunks, out):
    sys.path.insert(0, ROOT)
    if padded:
        os.environ["LSN_SHARD_PADDED"] = "1"
    os.environ["LSN_SHARD_CHUNKS"] = str(chunks)
    import numpy as np
    import torch
    from livescan3d_amd import synth
    from livescan3d_amd.fusion import DeviceFusion
    from livescan3d_amd.sharding import ShardedFusion
    from oracle import orc
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    T, S, w, h = 3, 4, 256, 212
    rigs = [synth.make_rig("noise" if k % 2 else "scene", S, w, h, seed=31, tick=k, bounds=synth.CROP_BOUNDS) for k in range(T)]
    intr, wt, bounds = rigs[0].intr, rigs[0].wt, rigs[0].bounds
    depth = torch.from_numpy(np.stack([r.depth_maps.view(np.int16) for r in rigs])).to(dev)
    rgb = torch.from_numpy(np.stack([r.depth_colors for r in rigs])).to(dev)
    sf = ShardedFusion(0, 1, T, [w] * S, [h] * S, dev)             # lsnShardUniqueId / lsnShardCreate: RCCL communicator of one rank
    sf.set_params(intr, wt, bounds)
    ok = True
    for rep in range(3):                                            # the second step on counts from the depth thresholds
        merged, moff = sf.step(depth, rgb)
        torch.cuda.synchronize()
        oh = moff.cpu().numpy()
        for k in range(T):
            want, counts = orc.generate_mesh_vertices(rigs[k].depth_maps, rigs[k].depth_colors, rigs[k].widths, rigs[k].heights, intr, wt, bounds)
            n = int(oh[k, -1])
            ok = ok and n == len(want) and list(np.diff(oh[k])) == list(counts) and merged[k, :n].cpu().numpy().tobytes() == want.tobytes()
    whole = DeviceFusion(T, [w] * S, [h] * S, device=0)
    whole.set_params(intr, wt, bounds)
    want_v, want_o = whole.run(depth, rgb)
    torch.cuda.synchronize()
    ok = ok and bool(torch.equal(moff, want_o))
    sent = sf.shard.last_bytes_sent()
    full = T * S * w * h * 5
    ok = ok and (sent > full if padded else sent < 0.8 * full)      # compact: survivors only; padded: whole capacity, no host read
    sf.close()
    with open(out, "w") as f:
        f.write("ok" if ok else "mismatch")


@pytest.mark.parametrize("padded,chunks", [(False, 1), (False, 3), (False, 2), (True, 1)])
def test_shard_exports_rccl_world1(gpu, tmp_path, padded, chunks):
    """lsnShardUniqueId / lsnShardCreate / lsnShardSetParams / lsnShardStep: the multi-GPU step behind the C-ABI (C++ host glue +
    RCCL inside libNativeUtils.so, no torch.distributed) with a communicator of one rank: pack, the grouped all-gathers, the
    pinned read-back of the offset tables (or none with $LSN_SHARD_PADDED=1) and the reconstruction must give exactly what a
    single-plan fusion and the oracle give.  chunks > 1: the streams travel in groups of ticks on a second stream while the
    previous group is reconstructed (the default with more than one rank)."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    mp.spawn(_shard_worker, args=(padded, chunks, out), nprocs=1, join=True)
    assert open(out).read() == "ok"


def _shard_worker_ragged(rank, padded, out):
    sys.path.insert(0, ROOT)
    if padded:
        os.environ["LSN_SHARD_PADDED"] = "1"
    import numpy as np
    import torch
    from livescan3d_amd import synth
    from livescan3d_amd.sharding import ShardedFusion
    from oracle import orc
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    T = 2
    sizes = [(61, 37), (512, 424), (100, 3), (250, 120)]          # sensors of different sizes, widths that are not multiples of 8
    rigs = []
    for k in range(T):
        depths, rgbs, intr, wt = [], [], [], []
        for s, (w, h) in enumerate(sizes):
            d, c = synth.noise_frame(17, k, s, w, h)
            depths.append(d); rgbs.append(c)
            intr.append(synth.kinect_intrinsics(w, h))
            wt.append(synth.pack_pose(*synth.ring_pose(s, len(sizes))))
        rigs.append(synth.Rig(depths, rgbs, np.concatenate(intr), np.concatenate(wt), [-1.0, -1.2, -1.5, 1.3, 1.1, 1.6]))
    depth = torch.from_numpy(np.stack([r.depth_maps.view(np.int16) for r in rigs])).to(dev)
    rgb = torch.from_numpy(np.stack([r.depth_colors for r in rigs])).to(dev)
    sf = ShardedFusion(0, 1, T, [w for w, _ in sizes], [h for _, h in sizes], dev)
    sf.set_params(rigs[0].intr, rigs[0].wt, rigs[0].bounds)
    ok = True
    for rep in range(2):
        merged, moff = sf.step(depth, rgb)
        torch.cuda.synchronize()
        oh = moff.cpu().numpy()
        for k in range(T):
            r = rigs[k]
            want, counts = orc.generate_mesh_vertices(r.depth_maps, r.depth_colors, r.widths, r.heights, r.intr, r.wt, r.bounds)
            n = int(oh[k, -1])
            ok = ok and n == len(want) and list(np.diff(oh[k])) == list(counts) and merged[k, :n].cpu().numpy().tobytes() == want.tobytes()
    cap = sum(w * h for w, h in sizes)
    sent = sf.shard.last_bytes_sent()
    ok = ok and (sent >= T * cap * 16 if padded else 0 < sent < T * cap * 16)
    sf.close()
    with open(out, "w") as f:
        f.write("ok" if ok else "mismatch")


@pytest.mark.parametrize("padded", [False, True])
def test_shard_exports_ragged_rig_world1(gpu, tmp_path, padded):
    """A rig the survivor exchange cannot serve (sensors of different sizes, widths not multiples of 8): lsnShardCreate falls back
    to exchanging the 16-byte vertices (lsnFusionRun on the block, one all-gather per tick cut to the step's largest shard, the
    packing pass of lsnMergeShards) and the merged cloud is still the oracle's, byte for byte."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    mp.spawn(_shard_worker_ragged, args=(padded, out), nprocs=1, join=True)
    assert open(out).read() == "ok"


FAKE_RCCL = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")


def _shard_worker_multi(rank, world, port, sizes, chunks, padded, out):
    sys.path.insert(0, ROOT)
    os.environ["LSN_RCCL_LIBRARY"] = FAKE_RCCL                      # the shared-memory test double: several ranks on this one GPU
    os.environ["LSN_SHARD_CHUNKS"] = str(chunks)
    if padded:
        os.environ["LSN_SHARD_PADDED"] = "1"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import numpy as np
    import torch
    import torch.distributed as dist
    from livescan3d_amd import synth
    from livescan3d_amd.fusion import DeviceFusion
    from livescan3d_amd.sharding import ShardedFusion
    dist.init_process_group("gloo", rank=rank, world_size=world)    # carries the 128-byte id only
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    T, S = 5, len(sizes)
    mpr = S // world
    rigs = []
    for k in range(T):
        depths, rgbs, intr, wt = [], [], [], []
        for s, (w, h) in enumerate(sizes):
            d, c = synth.noise_frame(29, k, s, w, h) if (k + s) % 3 else synth.scene_frame(29, k, s, S, w, h)
            depths.append(d); rgbs.append(c)
            intr.append(synth.kinect_intrinsics(w, h))
            wt.append(synth.pack_pose(*synth.ring_pose(s, S)))
        rigs.append(synth.Rig(depths, rgbs, np.concatenate(intr), np.concatenate(wt), synth.CROP_BOUNDS))
    pix = [w * h for w, h in sizes]
    p0, p1 = sum(pix[:rank * mpr]), sum(pix[:(rank + 1) * mpr])
    depth = torch.from_numpy(np.stack([r.depth_maps.view(np.int16) for r in rigs])).to(dev)
    rgb = torch.from_numpy(np.stack([r.depth_colors for r in rigs])).to(dev)
    mine_d = depth[:, p0:p1].contiguous()
    mine_c = rgb[:, 3 * p0:3 * p1].contiguous()
    widths, heights = [w for w, _ in sizes], [h for _, h in sizes]
    sf = ShardedFusion(rank, world, T, widths, heights, dev)
    sf.set_params(rigs[0].intr, rigs[0].wt, rigs[0].bounds)
    whole = DeviceFusion(T, widths, heights, device=0)
    whole.set_params(rigs[0].intr, rigs[0].wt, rigs[0].bounds)
    want_v, want_o = whole.run(depth, rgb)
    torch.cuda.synchronize()
    ok = True
    for rep in range(3):
        merged, moff = sf.step(mine_d, mine_c)
        torch.cuda.synchronize()
        ok = ok and bool(torch.equal(moff, want_o))
        for k in range(T):
            n = int(want_o[k, -1])
            ok = ok and n > 0 and bool(torch.equal(merged[k, :n], want_v[k, :n]))
    sent = sf.shard.last_bytes_sent()
    ok = ok and sent > 0
    if rank == 0:
        # ... and straight against the CPU oracle, not only against the single-plan HIP fusion: the merged cloud a rank ends up with
        # is the oracle's merge of ALL sensors, byte for byte
        from oracle import orc
        mv, mo = merged.cpu().numpy(), moff.cpu().numpy()
        for k in range(T):
            want, counts = orc.generate_mesh_vertices(rigs[k].depth_maps, rigs[k].depth_colors, widths, heights, rigs[0].intr, rigs[0].wt, rigs[0].bounds)
            n = int(mo[k, -1])
            ok = ok and n == len(want) and list(np.diff(mo[k])) == list(counts) and mv[k, :n].tobytes() == want.tobytes()
    sf.close()
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    dist.destroy_process_group()
    if rank == 0:
        with open(out, "w") as f:
            f.write("ok" if all(flags) else f"mismatch {flags}")


UNIFORM4 = [(256, 212)] * 4
UNIFORM8 = [(128, 96)] * 8
RAGGED4 = [(61, 37), (250, 120), (250, 120), (61, 37)]               # equal pixel totals per rank for world 2; not for world 4
# (the ranks' blocks must hold the same number of pixels -- an all-gather moves equal blocks; lsnShardPrepare refuses anything else)
MIXED8_4 = [(64, 48), (128, 96), (128, 96), (64, 48)]
MIXED8_6 = [(64, 48), (128, 24), (96, 32), (96, 32), (128, 24), (64, 48)]
MIXED8_8 = [(64, 48), (128, 24), (96, 32), (96, 32), (128, 24), (64, 48), (192, 16), (24, 128)]
ODD3 = [(7, 5), (5, 7), (35, 1)]
ODD8 = [(61, 37), (11, 4), (11, 4), (61, 37), (37, 61), (4, 11), (2257, 1), (2, 22)]


@pytest.mark.parametrize("world,sizes,chunks,padded", [
    (2, UNIFORM4, 1, False), (2, UNIFORM4, 3, False), (4, UNIFORM8, 4, False), (4, UNIFORM4, 1, True), (2, UNIFORM8, 5, False),
    (4, UNIFORM4, 4, False), (4, UNIFORM4, 1, False),                 # one sensor per rank (the shape of BASELINE configs[3]), compact: chunked and one shot
    (2, RAGGED4, 1, False), (2, RAGGED4, 1, True),
    # survivor exchange (every width a multiple of 8) on sensors of different sizes (every rank's block the same pixel total, as lsnShardPrepare
    # asks: 15360 / 6144 / 6144 pixels per rank here), an odd world
    (2, MIXED8_4, 3, False), (4, MIXED8_8, 2, False), (3, MIXED8_6, 1, False), (3, MIXED8_6, 4, True),
    # vertex exchange (widths that are not multiples of 8) over three and four ranks
    (3, ODD3, 1, False), (4, ODD8, 2, False)])
def test_shard_step_several_ranks_on_one_gpu(gpu, tmp_path, world, sizes, chunks, padded):
    """lsnShardStep with world > 1.  RCCL refuses two ranks on one device, so the library is pointed ($LSN_RCCL_LIBRARY) at
    tests/fake_rccl -- the seven nccl* entry points over a shared-memory segment -- and `world` processes share this GPU:
    rank offsets, the grouped all-gathers, the chunked second-stream pipeline, the reconstruction of every rank's sensors
    (or the vertex exchange on the ragged rig) all run as they would on `world` GPUs, and every rank must end up with the
    single-plan merged cloud; rank 0 also holds its result against the CPU oracle's merge of all sensors.  (World 8 = configs[3] itself:
    test_configs3_true_split_world8_on_one_gpu, rank threads inside 4 processes.)"""
    if not os.path.exists(FAKE_RCCL):
        pytest.fail("tests/fake_rccl/libfake_rccl.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    mp.spawn(_shard_worker_multi, args=(world, _free_port(), sizes, chunks, padded, out), nprocs=world, join=True)
    assert open(out).read() == "ok"


def _shard_worker_world8(proc, n_procs, port, out, S=8, w=512, h=424, T=2):
    """BASELINE configs[3] at its true split: 8 shards x 1 sensor x 512x424 (or configs[4]'s: 8 shards x 2 sensors x 1024x1024).  A GPU box
    admits 6 processes on its card, so the 8 ranks are 8 LsnShard handles driven by 2 threads in each of 4 processes (ctypes releases the
    GIL: the blocking rendezvous and collectives of the ranks of one process run side by side); every rank holds its merged cloud against
    the CPU oracle's merge of all S sensors."""
    sys.path.insert(0, ROOT)
    os.environ["LSN_RCCL_LIBRARY"] = FAKE_RCCL
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import threading
    import numpy as np
    import torch
    import torch.distributed as dist
    from livescan3d_amd import native, synth
    from oracle import orc
    dist.init_process_group("gloo", rank=proc, world_size=n_procs)   # carries the 128-byte id and the verdicts only
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    world = 8
    per_proc = world // n_procs
    P, mpr = w * h, S // world
    rigs = [synth.make_rig("scene" if k else "noise", S, w, h, seed=31, tick=k, bounds=synth.CROP_BOUNDS) for k in range(T)]
    depth = torch.from_numpy(np.stack([r.depth_maps.view(np.int16) for r in rigs])).to(dev)     # [T, S * P]
    rgb = torch.from_numpy(np.stack([r.depth_colors for r in rigs])).to(dev)
    want = [orc.generate_mesh_vertices(r.depth_maps, r.depth_colors, r.widths, r.heights, rigs[0].intr, rigs[0].wt, rigs[0].bounds) for r in rigs]
    # 1. every rank prepares on its own; the processes agree that all 8 are ready before anybody enters the blocking rendezvous
    shards, errs = {}, []
    for t in range(per_proc):
        rank = proc * per_proc + t
        try:
            shards[rank] = native.Shard(0, rank, world, None, T, [w] * S, [h] * S)
        except Exception as ex:  # noqa: BLE001
            errs.append(f"rank {rank}: {ex}")
    ident = native.shard_unique_id() if proc == 0 and not errs else None
    reports = [None] * n_procs
    dist.all_gather_object(reports, (errs, ident))
    verdicts = {}
    if not any(e for e, _ in reports):
        ident = reports[0][1]

        def drive(rank):
            try:
                sh = shards[rank]
                st = torch.cuda.Stream(device=dev)
                sh.connect(ident)                                       # ncclCommInitRank: returns once all 8 ranks have called it
                seen = sh.ranks_seen()
                sh.set_params(rigs[0].intr, rigs[0].wt, rigs[0].bounds, st.cuda_stream)
                mine_d = depth[:, rank * mpr * P:(rank + 1) * mpr * P].contiguous()
                mine_c = rgb[:, 3 * rank * mpr * P:3 * (rank + 1) * mpr * P].contiguous()
                torch.cuda.synchronize()
                ok = seen == world
                for rep in range(2):
                    mv, mo = sh.step(mine_d.data_ptr(), mine_c.data_ptr(), st.cuda_stream)
                    st.synchronize()
                    from livescan3d_amd.sharding import _device_view
                    merged = _device_view(mv, (T, sh.capacity, 16), torch.uint8, dev).cpu().numpy()
                    moff = _device_view(mo, (T, S + 1), torch.int32, dev).cpu().numpy()
                    for k in range(T):
                        verts, counts = want[k]
                        n = int(moff[k, -1])
                        ok = ok and n == len(verts) and n > 0 and list(np.diff(moff[k])) == list(counts) and merged[k, :n].tobytes() == verts.tobytes()
                verdicts[rank] = "ok" if ok else f"mismatch (ranks seen {seen})"
            except Exception as ex:  # noqa: BLE001
                verdicts[rank] = f"{type(ex).__name__}: {ex}"

        threads = [threading.Thread(target=drive, args=(r,)) for r in shards]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
    else:
        verdicts = {r: "not started: " + "; ".join(sum((e for e, _ in reports), [])) for r in range(proc * per_proc, (proc + 1) * per_proc)}
    for sh in shards.values():
        sh.close()
    everything = [None] * n_procs
    dist.all_gather_object(everything, verdicts)
    dist.destroy_process_group()
    if proc == 0:
        flat = {r: v for d in everything for r, v in d.items()}
        with open(out, "w") as f:
            f.write("ok" if len(flat) == world and all(v == "ok" for v in flat.values()) else f"failed {flat}")


def test_configs3_true_split_world8_on_one_gpu(gpu, tmp_path):
    """BASELINE configs[3]: 8 sensors sharded one per rank, 512x424, through lsnShardPrepare / lsnShardConnect / lsnShardStep with a world of 8
    (tests/fake_rccl carries the collectives: RCCL refuses several ranks on one device; 4 processes x 2 rank threads keep inside the box's
    6-process guard).  Every one of the 8 ranks must end with the oracle's merged cloud of all 8 sensors, byte for byte, and its communicator
    must report 8 ranks.  What this cannot show: the real library's rendezvous and stream ordering at world > 1 (no multi-GPU box in reach)."""
    if not os.path.exists(FAKE_RCCL):
        pytest.fail("tests/fake_rccl/libfake_rccl.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    mp.spawn(_shard_worker_world8, args=(4, _free_port(), out), nprocs=4, join=True)
    assert open(out).read() == "ok"


def test_configs4_true_split_world8_on_one_gpu(gpu, tmp_path):
    """BASELINE configs[4] at its own split: 16 sensors x 1024x1024 over 8 ranks x 2 sensors, through lsnShardPrepare / lsnShardConnect /
    lsnShardStep (4 processes x 2 rank threads over tests/fake_rccl, like configs[3]'s test above).  Every one of the 8 ranks must end with
    the oracle's merged cloud of all 16 sensors (4.7 M vertices per tick), byte for byte, offsets included."""
    if not os.path.exists(FAKE_RCCL):
        pytest.fail("tests/fake_rccl/libfake_rccl.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    mp.spawn(_shard_worker_world8, args=(4, _free_port(), out, 16, 1024, 1024, 1), nprocs=4, join=True)
    assert open(out).read() == "ok"


def _shard_worker_big(rank, out):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from livescan3d_amd import synth
    from livescan3d_amd.fusion import DeviceFusion
    from livescan3d_amd.sharding import ShardedFusion
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    T, S, w, h = 2, 16, 1024, 1024                                   # BASELINE configs[4]: 16 sensors x 1024x1024
    P = w * h
    depth, rgb = synth.noise_frames_torch(dev, 3, T, S, w, h)
    depth, rgb = depth.view(T, S * P), rgb.view(T, S * P * 3)
    intr = np.concatenate([synth.kinect_intrinsics(w, h)] * S)
    wt = np.concatenate([synth.pack_pose(*synth.ring_pose(s, S)) for s in range(S)])
    os.environ["LSN_SHARD_CHUNKS"] = "2"
    sf = ShardedFusion(0, 1, T, [w] * S, [h] * S, dev)
    sf.set_params(intr, wt, synth.CROP_BOUNDS)
    merged, moff = sf.step(depth, rgb)
    whole = DeviceFusion(T, [w] * S, [h] * S, device=0)
    whole.set_params(intr, wt, synth.CROP_BOUNDS)
    want_v, want_o = whole.run(depth, rgb)
    torch.cuda.synchronize()
    ok = bool(torch.equal(moff, want_o))
    for k in range(T):
        n = int(want_o[k, -1])
        ok = ok and n > S * P // 4 and bool(torch.equal(merged[k, :n], want_v[k, :n]))
    sf.close()
    with open(out, "w") as f:
        f.write("ok" if ok else "mismatch")


def test_shard_exports_configs4_shape_world1(gpu, tmp_path):
    """BASELINE configs[4]'s rig (16 sensors x 1024x1024, 16.8 M pixels per tick) through lsnShard* with one rank: the merged
    cloud (> 4 M vertices per tick) is bit-identical to a single-plan fusion of the same frames."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    mp.spawn(_shard_worker_big, args=(out,), nprocs=1, join=True)
    assert open(out).read() == "ok"


def _shard_worker_prepare_failure(rank, world, port, out):
    """Rank 1 asks for an impossible rig (its sensor blocks hold different pixel counts -> lsnShardPrepare fails there only): every
    rank must get the error instead of rank 0 hanging in ncclCommInitRank."""
    sys.path.insert(0, ROOT)
    os.environ["LSN_RCCL_LIBRARY"] = FAKE_RCCL
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from livescan3d_amd import native
    from livescan3d_amd.sharding import ShardedFusion
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    sizes = [(64, 48), (64, 48)] if rank == 0 else [(64, 48), (32, 48)]
    msg = "no error"
    try:
        ShardedFusion(rank, world, 2, [w for w, _ in sizes], [h for _, h in sizes], torch.device("cuda", 0))
    except native.NativeUtilsError as ex:
        msg = str(ex)
    msgs = [None] * world
    dist.all_gather_object(msgs, msg)
    dist.destroy_process_group()
    if rank == 0:
        with open(out, "w") as f:
            f.write("ok" if all("rank 1" in m and "lsnShardPrepare" in m for m in msgs) else f"unexpected {msgs}")


def test_a_rank_that_cannot_prepare_does_not_leave_its_peers_waiting(gpu, tmp_path):
    if not os.path.exists(FAKE_RCCL):
        pytest.fail("tests/fake_rccl/libfake_rccl.so is not built")
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    mp.spawn(_shard_worker_prepare_failure, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_bench_goes_on_when_the_communicator_setup_never_returns(gpu):
    """First contact of an N > 1 bench run with a node it has never seen: the library's own communicator rendezvous (ncclCommInitRank inside
    lsnShardConnect) may simply not come back.  The RCCL test double is told to hang there; `python bench.py --gpus 2` (two ranks sharing
    this GPU) must notice after $LSN_BENCH_CONNECT_TIMEOUT_S, agree on the Python-driven survivor exchange over torch.distributed instead,
    print its line -- saying what happened -- and leave with status 0 although a thread is still stuck in the library."""
    import json
    import subprocess
    if not os.path.exists(FAKE_RCCL):
        pytest.fail("tests/fake_rccl/libfake_rccl.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(LSN_RCCL_LIBRARY=FAKE_RCCL, LSN_BENCH_SHARE_GPU="1", FAKE_RCCL_HANG_INIT="1", LSN_BENCH_CONNECT_TIMEOUT_S="5")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-icp", "--no-cpu",
                        "--no-host-path", "--no-mesh", "--no-tick-parallel"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert "did not return within" in str(line["config"].get("shard_preflight")), line["config"]
    assert "never returned" in line.get("degraded", ""), line.get("degraded")
    assert "Python over torch.distributed" in line["config"]["parallelism"], line["config"]["parallelism"]
