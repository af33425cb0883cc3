"""The N > 1 path on CPU: two gloo ranks, one block of sensors each, all-gather + merged-cloud assembly.

The per-rank fusion is done by the CPU oracle here (tests may use it); what is under test is the product's sharding
logic (livescan3d_amd/sharding.py): contiguous sensor blocks, the two all-gathers, and the packing contract of
lsnMergeShards (restated in numpy below, because the HIP kernel needs a GPU).  The merged cloud on every rank must
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
    from livescan3d_amd.sharding import MergedCloudExchange, sensor_block
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


@pytest.mark.parametrize("S", [2, 4])
def test_two_rank_allgather_equals_single_process_merge(tmp_path, orc, S):
    from livescan3d_amd import synth
    world, T, w, h = 2, 3, 64, 48
    mp.spawn(_worker, args=(world, _free_port(), S, T, w, h, str(tmp_path)), nprocs=world, join=True)
    m0, m1 = np.load(tmp_path / "merged_0.npy"), np.load(tmp_path / "merged_1.npy")
    o0, o1 = np.load(tmp_path / "off_0.npy"), np.load(tmp_path / "off_1.npy")
    assert np.array_equal(o0, o1)
    for k in range(T):
        rig = synth.make_rig("scene" if k % 2 == 0 else "noise", S, w, h, seed=13, tick=k, bounds=synth.CROP_BOUNDS)
        want, counts = orc.generate_mesh_vertices(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
        n = int(o0[k, -1])
        assert n == len(want) and list(np.diff(o0[k])) == list(counts)
        for m in (m0, m1):                                  # every rank holds the whole merged cloud
            assert m[k, :n].tobytes() == want.tobytes()


def test_sensor_blocks():
    from livescan3d_amd.sharding import sensor_block
    assert [sensor_block(8, 4, r) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 8)]
    assert sensor_block(8, 1, 0) == (0, 8) and sensor_block(8, 8, 7) == (7, 8)
    with pytest.raises(ValueError):
        sensor_block(8, 3, 0)
