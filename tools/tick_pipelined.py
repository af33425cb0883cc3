#!/usr/bin/env python3
"""Dev experiment: the chained tick on two plans / two streams (half the ticks each) against one plan on one stream."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from livescan3d_amd import native, synth

kind = sys.argv[1] if len(sys.argv) > 1 else "scene"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 2
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

def make(n_ticks):
    plan = native.FusionPlan(0, n_ticks, rig0.widths, rig0.heights)
    plan.set_params(rig0.intr, rig0.wt, rig0.bounds)
    cap = plan.capacity
    return dict(plan=plan, verts=torch.zeros((n_ticks, cap, 16), dtype=torch.uint8, device=dev), off=torch.zeros((n_ticks, S + 1), dtype=torch.int32, device=dev),
                tri=torch.zeros((n_ticks, 2 * cap, 3), dtype=torch.int32, device=dev), toff=torch.zeros((n_ticks, S + 1), dtype=torch.int32, device=dev),
                d2=torch.empty((n_ticks, depth.shape[1]), dtype=depth.dtype, device=dev), c2=torch.empty((n_ticks, rgb.shape[1]), dtype=rgb.dtype, device=dev))

def tick(p, d, c, st):
    p["plan"].radial_correct_to(rig0.intr, d.data_ptr(), c.data_ptr(), p["d2"].data_ptr(), p["c2"].data_ptr(), st)
    p["plan"].run_mesh(p["d2"].data_ptr(), p["c2"].data_ptr(), p["verts"].data_ptr(), p["off"].data_ptr(), p["tri"].data_ptr(), p["toff"].data_ptr(), st)

def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps

one = make(T)
st0 = int(torch.cuda.current_stream().cuda_stream)
ms1 = timed(lambda: tick(one, depth, rgb, st0))
print(kind, T, "ticks, one plan / one stream:", round(ms1, 3), "ms ->", round(T / ms1, 1), "k ticks/s", flush=True)
del one
torch.cuda.empty_cache()
per = T // parts
ps = [make(per) for _ in range(parts)]
streams = [torch.cuda.Stream() for _ in range(parts)]
def piped():
    for i in range(parts):
        tick(ps[i], depth[i * per:(i + 1) * per], rgb[i * per:(i + 1) * per], int(streams[i].cuda_stream))
ms2 = timed(piped)
print(kind, T, f"ticks, {parts} plans / {parts} streams:", round(ms2, 3), "ms ->", round(T / ms2, 1), "k ticks/s", flush=True)
