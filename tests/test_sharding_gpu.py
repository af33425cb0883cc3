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
    from livescan3d_amd.sharding import MergedCloudExchange
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
