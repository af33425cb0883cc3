#!/usr/bin/env python3
"""Dev helper: the chained tick (radial correction out of place -> vertices -> triangulation) on T ticks x 8 sensors, for kernel traces.
usage: python3 tools/tick_driver.py [scene|noise] [ticks] [reps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from livescan3d_amd import native, synth

kind = sys.argv[1] if len(sys.argv) > 1 else "scene"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
S, w, h = 8, 512, 424
dev = torch.device("cuda", 0)
rig0 = synth.make_rig("scene", S, w, h, seed=4, tick=0, bounds=synth.CROP_BOUNDS)
if kind == "noise":
    depth, rgb = synth.noise_frames_torch(dev, 1, T, S, w, h)
    depth, rgb = depth.view(T, -1), rgb.view(T, -1)
else:
    rigs = [synth.make_rig("scene", S, w, h, seed=4, tick=k) for k in range(min(T, 8))]
    depth = torch.from_numpy(np.stack([rigs[k % len(rigs)].depth_maps.view(np.int16) for k in range(T)])).to(dev)
    rgb = torch.from_numpy(np.stack([rigs[k % len(rigs)].depth_colors for k in range(T)])).to(dev)
plan = native.FusionPlan(0, T, rig0.widths, rig0.heights)
plan.set_params(rig0.intr, rig0.wt, rig0.bounds)
cap = plan.capacity
verts = torch.zeros((T, cap, 16), dtype=torch.uint8, device=dev)
off = torch.zeros((T, S + 1), dtype=torch.int32, device=dev)
tri = torch.zeros((T, 2 * cap, 3), dtype=torch.int32, device=dev)
toff = torch.zeros((T, S + 1), dtype=torch.int32, device=dev)
d2, c2 = torch.empty_like(depth), torch.empty_like(rgb)
st = int(torch.cuda.current_stream().cuda_stream)
for rep in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.radial_correct_to(rig0.intr, depth.data_ptr(), rgb.data_ptr(), d2.data_ptr(), c2.data_ptr(), st)
    plan.run_mesh(d2.data_ptr(), c2.data_ptr(), verts.data_ptr(), off.data_ptr(), tri.data_ptr(), toff.data_ptr(), st)
    torch.cuda.synchronize()
    print(kind, T, "ticks:", round(1e3 * (time.perf_counter() - t0), 3), "ms", flush=True)
print("vertices/tick", float(off[:, -1].float().mean()), "triangles/tick", float(toff[:, -1].float().mean()))
