#!/usr/bin/env python3
"""Dev helper: minimal driver for PMC passes (few dispatches: numpy-generated frames, a handful of launches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from livescan3d_amd import synth
from livescan3d_amd.fusion import DeviceFusion

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
T, S, w, h = int(os.environ.get("PMC_TICKS", "64")), 8, 512, 424
rigs = [synth.make_rig("noise", S, w, h, seed=1, tick=k, bounds=synth.CROP_BOUNDS) for k in range(8)]
d = np.stack([rigs[k % 8].depth_maps.view(np.int16) for k in range(T)])
c = np.stack([rigs[k % 8].depth_colors for k in range(T)])
depth = torch.from_numpy(d).cuda(); rgb = torch.from_numpy(c).cuda()
fus = DeviceFusion(T, [w] * S, [h] * S, mode=mode)
fus.set_params(rigs[0].intr, rigs[0].wt, rigs[0].bounds)
for _ in range(4):
    fus.run(depth, rgb)
torch.cuda.synchronize()
off = fus.offsets.cpu().numpy()
V = int(off[:, -1].sum()); P = w * h * S * T
print("pixels", P, "vertices", V, "algorithmic_bytes", 2 * P + 19 * V, "actual_min_bytes", 5 * P + 16 * V)
