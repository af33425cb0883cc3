#!/usr/bin/env python3
"""Dev helper: the two kernels of the multi-GPU step (pack_kernel, recon_kernel) on one GPU WITHOUT RCCL -- the shard is the whole
rig (one "rank"), the all-gather is the identity -- for rocprofv3 passes (`--pmc` and torch.distributed do not get along here).
64 ticks x 8 x 512x424 noise frames, a few dispatches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from livescan3d_amd import synth
from livescan3d_amd.fusion import DeviceFusion

T, S, w, h = 64, 8, 512, 424
P = w * h
dev = torch.device("cuda", 0)
# numpy-generated frames, few dispatches (the torch generator's thousands of small launches do not survive a --pmc pass)
rigs = [synth.make_rig("noise", S, w, h, seed=1, tick=k, bounds=synth.CROP_BOUNDS) for k in range(8)]
depth = torch.from_numpy(np.stack([rigs[k % 8].depth_maps.view(np.int16) for k in range(T)])).to(dev)
rgb = torch.from_numpy(np.stack([rigs[k % 8].depth_colors for k in range(T)])).to(dev)
plan = DeviceFusion(T, [w] * S, [h] * S)
plan.set_params(rigs[0].intr, rigs[0].wt, rigs[0].bounds)
cap = S * P
tiles = plan.plan.tiles_per_tick
st = int(torch.cuda.current_stream().cuda_stream)
mask = torch.empty((T, cap // 8), dtype=torch.uint8, device=dev)
dc = torch.empty((T * cap + 64,), dtype=torch.int16, device=dev)
cc = torch.empty((T * cap + 64, 3), dtype=torch.uint8, device=dev)
tp = torch.zeros((T, tiles), dtype=torch.int32, device=dev)
off = torch.zeros((T, S + 1), dtype=torch.int32, device=dev)
tb = torch.zeros((T,), dtype=torch.int32, device=dev)
merged = torch.empty((T, cap, 16), dtype=torch.uint8, device=dev)
moff = torch.zeros((T, S + 1), dtype=torch.int32, device=dev)
scratch = torch.zeros((1, T), dtype=torch.int32, device=dev)
for rep in range(int(os.environ.get("SHARD_REPS", "4"))):
    plan.plan.pack_survivors_run(depth.data_ptr(), rgb.data_ptr(), mask.data_ptr(), dc.data_ptr(), cc.data_ptr(), tp.data_ptr(), off.data_ptr(),
                                 tb.data_ptr(), st)
    torch.cuda.synchronize()
    run_len = (int(off[:, S].sum().item()) + 7) & ~7
    plan.plan.reconstruct_run(1, S, mask.data_ptr(), dc.data_ptr(), cc.data_ptr(), run_len, tp.data_ptr(), off.data_ptr(), merged.data_ptr(),
                              moff.data_ptr(), scratch.data_ptr(), st)
    torch.cuda.synchronize()
want_v, want_o = plan.run(depth, rgb)
torch.cuda.synchronize()
print("survivors", run_len, "offsets identical", bool(torch.equal(moff, want_o)),
      "algorithmic bytes pack", 5 * T * cap + T * cap // 8 + 5 * run_len, "recon", 21 * run_len + T * cap // 8)
