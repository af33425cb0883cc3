"""The N > 1 path on CPU: gloo ranks (world 2, and world 8 x 1 sensor = BASELINE configs[3]'s split), one block of sensors each,
all-gathers + merged-cloud assembly, both exchange protocols.

The per-rank fusion is done by the CPU oracle here (tests may use it); what is under test is the sharding logic: contiguous sensor
blocks (livescan3d_amd/sharding.py), ownership and offsets of every rank's shard, the all-gathers (bench_support/exchange.py, the
same protocol lsnShard* runs over RCCL), and the packing contracts of lsnMergeShards / lsnFusionPackSurvivors / lsnFusionReconstruct
(restated in numpy below, because the HIP kernels need a GPU).  The merged cloud on every rank must
equal the single-process merged cloud byte for byte, in formMesh's sensor order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def numpy_merge(g_verts, g_off, merged, merged_off):
    """Contract of lsnMergeShards (include/NativeUtils.h): shards concatenated per tick in rank order."""
    world, T, cap, _ = g_verts.shape
    mpr = g_off.shape[2] - 1
    for k in range(T):
        base = 0
        for r in range(world):
            cnt = int(g_off[r, k, mpr])
            merged[k, base:base + cnt] = g_verts[r, k, :cnt]
            for j in range(mpr):
                merged_off[k, r * mpr + j] = base + int(g_off[r, k, j])
            base += cnt
        merged_off[k, world * mpr] = base


def _worker(rank, world, port, S, T, w, h, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from livescan3d_amd import synth
    from livescan3d_amd.sharding import sensor_block
    from bench_support.exchange import MergedCloudExchange
    from oracle import orc
    s0, s1 = sensor_block(S, world, rank)
    mpr = s1 - s0
    cap = mpr * w * h
    verts = torch.zeros((T, cap, 16), dtype=torch.uint8)
    offs = torch.zeros((T, mpr + 1), dtype=torch.int32)
    for k in range(T):
        rig = synth.make_rig("scene" if k % 2 == 0 else "noise", S, w, h, seed=13, tick=k, bounds=synth.CROP_BOUNDS)
        P = w * h
        v, counts = orc.generate_mesh_vertices(rig.depth_maps[2 * P * s0:2 * P * s1], rig.depth_colors[3 * P * s0:3 * P * s1],
                                               rig.widths[s0:s1], rig.heights[s0:s1], rig.intr[7 * s0:7 * s1], rig.wt[12 * s0:12 * s1], rig.bounds)
        verts[k, :len(v)] = torch.from_numpy(v.view(np.uint8).reshape(-1, 16).copy())
        offs[k] = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32))
    xch = MergedCloudExchange(world, T, mpr, cap, "cpu",
                              merge_fn=lambda gv, go, m, mo: numpy_merge(gv.numpy(), go.numpy(), m.numpy(), mo.numpy()))
    merged, merged_off = xch.exchange(verts, offs)
    np.save(os.path.join(out_dir, f"merged_{rank}.npy"), merged.numpy())
    np.save(os.path.join(out_dir, f"off_{rank}.npy"), merged_off.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _skip_world8_beside_a_gpu(world):
    """A GPU box admits at most 6 processes with its card open, and importing torch there opens it: the world-8 rehearsal is a CPU-container
    test (where the driver runs the `not gpu` suite)."""
    if world > 4 and torch.cuda.device_count() > 0:
        pytest.skip("world 8 on gloo runs in the CPU container (a GPU box's process guard admits 6 processes on the card)")


def _check_every_rank(tmp_path, orc, world, S, T, w, h):
    """Every rank's merged cloud and offset table equal the single-process merge of all S sensors, byte for byte."""
    from livescan3d_amd import synth
    merged = [np.load(tmp_path / f"merged_{r}.npy") for r in range(world)]
    offs = [np.load(tmp_path / f"off_{r}.npy") for r in range(world)]
    for o in offs[1:]:
        assert np.array_equal(offs[0], o)
    for k in range(T):
        rig = synth.make_rig("scene" if k % 2 == 0 else "noise", S, w, h, seed=13, tick=k, bounds=synth.CROP_BOUNDS)
        want, counts = orc.generate_mesh_vertices(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
        n = int(offs[0][k, -1])
        assert n == len(want) and list(np.diff(offs[0][k])) == list(counts)
        for m in merged:                                    # every rank holds the whole merged cloud
            assert m[k, :n].tobytes() == want.tobytes()


# (8, 8): BASELINE configs[3] -- 8 sensors sharded one per rank -- as a CPU rehearsal of ownership and offsets
@pytest.mark.parametrize("world,S", [(2, 2), (2, 4), (8, 8)])
def test_allgather_of_vertex_shards_equals_single_process_merge(tmp_path, orc, world, S):
    _skip_world8_beside_a_gpu(world)
    T, w, h = (3, 64, 48) if world == 2 else (2, 64, 48)
    mp.spawn(_worker, args=(world, _free_port(), S, T, w, h, str(tmp_path)), nprocs=world, join=True)
    _check_every_rank(tmp_path, orc, world, S, T, w, h)


def test_sensor_blocks():
    from livescan3d_amd.sharding import sensor_block
    assert [sensor_block(8, 4, r) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 8)]
    assert sensor_block(8, 1, 0) == (0, 8) and sensor_block(8, 8, 7) == (7, 8)
    with pytest.raises(ValueError):
        sensor_block(8, 3, 0)


# ---- the survivor exchange (5 bytes per survivor + a mask instead of 16-byte vertices) on two gloo ranks -----------------

KTILE = 2048      # pixels per tile in the library (only the packed tile prefixes depend on it)


def _survivor_worker(rank, world, port, S, T, w, h, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from types import SimpleNamespace
    from livescan3d_amd import synth
    from livescan3d_amd.sharding import sensor_block
    from bench_support.exchange import SurvivorExchange
    from oracle import orc
    s0, s1 = sensor_block(S, world, rank)
    mpr, P = s1 - s0, w * h
    tiles_per_frame = (P + KTILE - 1) // KTILE
    rigs = [synth.make_rig("scene" if k % 2 == 0 else "noise", S, w, h, seed=13, tick=k, bounds=synth.CROP_BOUNDS) for k in range(T)]
    intr, wt, bounds = rigs[0].intr, rigs[0].wt, rigs[0].bounds

    def pack_fn(depth, rgb, mask, depth_c, rgb_c, tile_prefix, offsets):
        """Contract of lsnFusionPackSurvivors, restated with the oracle."""
        for k in range(T):
            bits = np.zeros(mpr * P, dtype=bool)
            base, tile_counts, offs = 0, [], [0]
            for j in range(mpr):
                d = depth[k, j * P:(j + 1) * P].numpy().view(np.uint16).reshape(h, w)
                c = rgb[k, 3 * j * P:3 * (j + 1) * P].numpy().reshape(h, w, 3)
                s = s0 + j
                v, v2p, p2v = orc.create_vertices(d, c, intr[7 * s:7 * s + 7], wt[12 * s:12 * s + 12], bounds, want_maps=True)
                n = len(v)
                depth_c[k, base:base + n] = torch.from_numpy(d.ravel()[v2p].view(np.int16).copy())
                rgb_c[k, base:base + n] = torch.from_numpy(c.reshape(-1, 3)[v2p].copy())
                bits[j * P:(j + 1) * P] = p2v >= 0
                keep = (p2v >= 0).astype(np.int64)
                tile_counts += [int(keep[t * KTILE:(t + 1) * KTILE].sum()) for t in range(tiles_per_frame)]
                base += n
                offs.append(base)
            mask[k] = torch.from_numpy(np.packbits(bits, bitorder="little"))
            tile_prefix[k] = torch.from_numpy(np.concatenate([[0], np.cumsum(tile_counts)[:-1]]).astype(np.int32))
            offsets[k] = torch.tensor(offs, dtype=torch.int32)

    def recon_fn(g_mask, gd, gc, g_tp, g_off, merged, merged_off):
        """Contract of lsnFusionReconstruct: expand every shard's streams back into frames and apply the reference arithmetic."""
        W = g_mask.shape[0]
        for k in range(T):
            pos, moff = 0, [0]
            for r in range(W):
                bits = np.unpackbits(g_mask[r, k].numpy(), bitorder="little")[:mpr * P].astype(bool)
                go = g_off[r, k].numpy()
                for j in range(mpr):
                    sel = bits[j * P:(j + 1) * P]
                    n = int(go[j + 1] - go[j])
                    assert int(sel.sum()) == n
                    d = np.zeros(P, dtype=np.uint16)
                    c = np.zeros((P, 3), dtype=np.uint8)
                    d[sel] = gd[r, k, go[j]:go[j + 1]].numpy().view(np.uint16)
                    c[sel] = gc[r, k, go[j]:go[j + 1]].numpy()
                    s = r * mpr + j
                    v, _, _ = orc.create_vertices(d.reshape(h, w), c.reshape(h, w, 3), intr[7 * s:7 * s + 7], wt[12 * s:12 * s + 12], bounds, want_maps=True)
                    assert len(v) == n
                    merged[k, pos:pos + n] = torch.from_numpy(v.view(np.uint8).reshape(-1, 16).copy())
                    pos += n
                    moff.append(pos)
            merged_off[k] = torch.tensor(moff, dtype=torch.int32)

    local = SimpleNamespace(n_ticks=T, n_maps=mpr, capacity=mpr * P, tiles_per_tick=mpr * tiles_per_frame, device="cpu", plan=None)
    whole = SimpleNamespace(capacity=S * P, plan=None)
    depth = torch.from_numpy(np.stack([r.depth_maps.view(np.int16)[s0 * P:s1 * P] for r in rigs]))
    rgb = torch.from_numpy(np.stack([r.depth_colors[3 * s0 * P:3 * s1 * P] for r in rigs]))
    xch = SurvivorExchange(world, local, whole, pack_fn=pack_fn, recon_fn=recon_fn)
    merged, merged_off = xch.exchange(depth, rgb)
    assert xch.last_slab < mpr * P
    np.save(os.path.join(out_dir, f"merged_{rank}.npy"), merged.numpy())
    np.save(os.path.join(out_dir, f"off_{rank}.npy"), merged_off.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,S", [(2, 4), (8, 8)])
def test_survivor_exchange_equals_single_process_merge(tmp_path, orc, world, S):
    _skip_world8_beside_a_gpu(world)
    T, w, h = 2, 64, 48
    mp.spawn(_survivor_worker, args=(world, _free_port(), S, T, w, h, str(tmp_path)), nprocs=world, join=True)
    _check_every_rank(tmp_path, orc, world, S, T, w, h)


def test_bench_starts_its_own_ranks_when_no_launcher_is_around_it():
    """`python bench.py --gpus 2` with no torch.distributed.run around it (what the driver's N > 1 tier would run if it launched the bench the
    way it launches N = 1): bench.py starts `python -m torch.distributed.run --nproc-per-node 2 bench.py ...` as a child process before it
    imports torch, relays rank 0's one JSON line and leaves with the child's status.  LSN_BENCH_LAUNCH_PROBE=1 swaps the GPU work for a gloo
    head count, so the launch path itself runs here."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["LSN_BENCH_LAUNCH_PROBE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                       # ONE line on stdout, whatever the ranks and the launcher print
    line = json.loads(lines[0])
    assert line["n_ranks_seen"] == 2 and line["world_size"] == 2 and line["n_gpus"] == 2
    assert "without a launcher" in r.stderr


def test_bench_without_a_gpu_fails_loudly_through_the_self_launch():
    """The same command for real on this GPU-less container: the ranks refuse to run (no CPU path) and the parent reports their status."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "LSN_BENCH_LAUNCH_PROBE")}
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for the GPU-less container")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--core-only"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "needs a HIP device" in r.stderr
