"""Minimal radial-correction run for rocprofv3: python3 tools/radial_driver.py [noise|scene] [ticks] [reps] [sensors width height]: in place, then out of place."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from livescan3d_amd import synth
from livescan3d_amd.fusion import DeviceFusion
kind = sys.argv[1] if len(sys.argv) > 1 else "noise"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
S, w, h = [int(x) for x in sys.argv[4:7]] if len(sys.argv) >= 7 else (8, 512, 424)
dev = torch.device("cuda", 0)
if kind == "noise":
    depth, rgb = synth.noise_frames_torch(dev, 1, T, S, w, h)
    depth, rgb = depth.view(T, -1), rgb.view(T, -1)
else:
    rigs = [synth.make_rig("scene", S, w, h, seed=3, tick=k) for k in range(min(T, 8))]
    depth = torch.from_numpy(np.stack([rigs[k % len(rigs)].depth_maps.view(np.int16) for k in range(T)])).to(dev)
    rgb = torch.from_numpy(np.stack([rigs[k % len(rigs)].depth_colors for k in range(T)])).to(dev)
fus = DeviceFusion(T, [w] * S, [h] * S, device=0)
intr = np.concatenate([synth.kinect_intrinsics(w, h)] * S)
st = int(torch.cuda.current_stream().cuda_stream)
for rep in range(reps):
    d2, c2 = depth.clone(), rgb.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fus.plan.radial_correct(intr, d2.data_ptr(), c2.data_ptr(), st)
    torch.cuda.synchronize()
    print(kind, T, "ticks, in place:", round(1e3 * (time.perf_counter() - t0), 3), "ms", flush=True)
d3, c3 = torch.empty_like(depth), torch.empty_like(rgb)
for rep in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fus.plan.radial_correct_to(intr, depth.data_ptr(), rgb.data_ptr(), d3.data_ptr(), c3.data_ptr(), st)
    torch.cuda.synchronize()
    print(kind, T, "ticks, out of place:", round(1e3 * (time.perf_counter() - t0), 3), "ms", flush=True)
print("same result:", bool(torch.equal(d2, d3) and torch.equal(c2, c3)))
