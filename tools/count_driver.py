"""Times the count pass variants: python3 tools/count_driver.py [ticks sensors width height]  (reads $LSN_TICK_GROUP / $LSN_NO_THRESHOLDS)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from livescan3d_amd import synth
from livescan3d_amd.fusion import DeviceFusion
T, S, w, h = [int(x) for x in sys.argv[1:5]] if len(sys.argv) >= 5 else (64, 8, 512, 424)
dev = torch.device("cuda", 0)
depth, rgb = synth.noise_frames_torch(dev, 1, T, S, w, h)
depth, rgb = depth.view(T, -1), rgb.view(T, -1)
fus = DeviceFusion(T, [w] * S, [h] * S, device=0)
fus.set_params(np.concatenate([synth.kinect_intrinsics(w, h)] * S), np.concatenate([synth.pack_pose(*synth.ring_pose(s, S)) for s in range(S)]), synth.CROP_BOUNDS)
for _ in range(4):
    fus.run(depth, rgb)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    fus.run(depth, rgb)
torch.cuda.synchronize()
print("shape", T, S, w, h, "G", os.environ.get("LSN_TICK_GROUP"), "nothr", os.environ.get("LSN_NO_THRESHOLDS"), "ms/step", round((time.perf_counter() - t0) / 50 * 1e3, 4))
