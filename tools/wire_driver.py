"""lsnTransferPack on one tick's merged mesh (8 x 512x424 scene frames), `reps` times: wall time per call, path taken.
LSN_TRANSFER_WINDOW_WALK=1 selects the chunk-after-chunk walk.  Usage: python3 tools/wire_driver.py [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from livescan3d_amd import native, synth
from livescan3d_amd.fusion import DeviceFusion

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
S, w, h = 8, 512, 424
dev = torch.device("cuda:0")
rig = synth.make_rig("scene", S, w, h, seed=3)
fus = DeviceFusion(1, rig.widths, rig.heights, device=0)
fus.set_params(rig.intr, rig.wt, rig.bounds)
P = w * h
depth = torch.from_numpy(rig.depth_maps.view(np.int16).copy()).to(dev).view(1, S * P)
rgb = torch.from_numpy(rig.depth_colors.copy()).to(dev).view(1, S * P * 3)
tri = torch.empty((1, 2 * fus.capacity, 3), dtype=torch.int32, device=dev)
toff = torch.zeros((1, S + 1), dtype=torch.int32, device=dev)
fus.plan.run_mesh(depth.data_ptr(), rgb.data_ptr(), fus.vertices.data_ptr(), fus.offsets.data_ptr(), tri.data_ptr(), toff.data_ptr(), 0)
torch.cuda.synchronize()
nv, nt = int(fus.offsets[0, -1].item()), int(toff[0, -1].item())
bound = native.transfer_frame_bound(nv, nt)
out = torch.empty(bound + 64, dtype=torch.uint8, device=dev)
packer = native.TransferPacker(0, nv, nt)
for _ in range(3):
    n = packer.pack(fus.vertices.data_ptr(), nv, tri.data_ptr(), nt, out.data_ptr(), bound, 0)
t0 = time.perf_counter()
for _ in range(reps):
    n = packer.pack(fus.vertices.data_ptr(), nv, tri.data_ptr(), nt, out.data_ptr(), bound, 0)
dt = (time.perf_counter() - t0) / reps
print({"vertices": nv, "triangles": nt, "bytes": n, "chunks": int(out[8:12].view(torch.int32).item()), "path": packer.last_path(),
       "ms_per_call": round(1e3 * dt, 4)})
