"""Minimal radial-correction run for rocprofv3: python3 tools/radial_driver.py [noise|scene] [ticks]."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from livescan3d_amd import synth
from livescan3d_amd.fusion import DeviceFusion
kind = sys.argv[1] if len(sys.argv) > 1 else "noise"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
S, w, h = 8, 512, 424
dev = torch.device("cuda", 0)
if kind == "noise":
    depth, rgb = synth.noise_frames_torch(dev, 1, T, S, w, h)
    depth, rgb = depth.view(T, -1), rgb.view(T, -1)
else:
    rigs = [synth.make_rig("scene", S, w, h, seed=3, tick=k) for k in range(T)]
    depth = torch.from_numpy(np.stack([r.depth_maps.view(np.int16) for r in rigs])).to(dev)
    rgb = torch.from_numpy(np.stack([r.depth_colors for r in rigs])).to(dev)
fus = DeviceFusion(T, [w] * S, [h] * S, device=0)
intr = np.concatenate([synth.kinect_intrinsics(w, h)] * S)
st = int(torch.cuda.current_stream().cuda_stream)
for rep in range(3):
    d2, c2 = depth.clone(), rgb.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fus.plan.radial_correct(intr, d2.data_ptr(), c2.data_ptr(), st)
    torch.cuda.synchronize()
    print(kind, T, "ticks:", round(1e3 * (time.perf_counter() - t0), 3), "ms")
