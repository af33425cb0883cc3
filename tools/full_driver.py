#!/usr/bin/env python3
"""Dev helper: one pass of radial correction + full mesh on 16 ticks x 8 sensors (scene data), for kernel traces."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from livescan3d_amd import native, synth

T, S, w, h = 16, 8, 512, 424
kind = os.environ.get("DRV_KIND", "scene")
rigs = [synth.make_rig(kind, S, w, h, seed=1, tick=k, bounds=synth.CROP_BOUNDS) for k in range(2)]
depth = torch.from_numpy(np.stack([rigs[k % 2].depth_maps.view(np.int16) for k in range(T)])).cuda()
rgb = torch.from_numpy(np.stack([rigs[k % 2].depth_colors for k in range(T)])).cuda()
plan = native.FusionPlan(0, T, rigs[0].widths, rigs[0].heights)
plan.set_params(rigs[0].intr, rigs[0].wt, rigs[0].bounds)
cap = plan.capacity
verts = torch.zeros((T, cap, 16), dtype=torch.uint8, device="cuda")
off = torch.zeros((T, S + 1), dtype=torch.int32, device="cuda")
tri = torch.zeros((T, 2 * cap, 3), dtype=torch.int32, device="cuda")
toff = torch.zeros((T, S + 1), dtype=torch.int32, device="cuda")
st = int(torch.cuda.current_stream().cuda_stream)
for rep in range(2):
    d2, c2 = depth.clone(), rgb.clone()
    plan.radial_correct(rigs[0].intr, d2.data_ptr(), c2.data_ptr(), st)
    plan.run_mesh(d2.data_ptr(), c2.data_ptr(), verts.data_ptr(), off.data_ptr(), tri.data_ptr(), toff.data_ptr(), st)
    torch.cuda.synchronize()
print("vertices/tick", float(off[:, -1].float().mean()), "triangles/tick", float(toff[:, -1].float().mean()))
