#!/usr/bin/env python3
"""Dev helper: minimal ICP driver for profiling (configs[1]-sized clouds, few dispatches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from livescan3d_amd import native, synth
from livescan3d_amd.fusion import DeviceFusion, upload_rig

n_sens = int(os.environ.get("ICP_SENSORS", "2"))
rig = synth.make_rig("scene", n_sens, 512, 424, seed=4, perturb=True)
fus = DeviceFusion(1, rig.widths, rig.heights)
fus.set_params(rig.intr, rig.wt, rig.bounds)
d, c = upload_rig(rig, 1)
v, off = fus.run(d, c)
torch.cuda.synchronize()
off = off[0].cpu().numpy()
xyz = v[0, :int(off[-1]), 4:16].contiguous().view(torch.float32).view(-1, 3)
src0 = xyz[int(off[0]):int(off[1])].contiguous()          # sensor 0 is refined against all the others
tgt = xyz[int(off[1]):].contiguous()
n1, n2 = tgt.shape[0], src0.shape[0]
ws = native.IcpWorkspace(0, n1, n2)
for rep in range(int(os.environ.get("ICP_REPS", "3"))):
    src = src0.clone()
    Rt = torch.tensor([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=torch.float32, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ws.run(tgt.data_ptr(), n1, src.data_ptr(), n2, Rt.data_ptr(), Rt.data_ptr() + 36, 10, native.NN_BRUTE if os.environ.get("ICP_BRUTE") == "1" else native.NN_GRID, int(torch.cuda.current_stream().cuda_stream))
    e1.record()
    torch.cuda.synchronize()
    print("n1", n1, "n2", n2, "ms/iter", e0.elapsed_time(e1) / 10)
if hasattr(native.lib(), "lsnIcpNearResolved"):
    print("near path settled", ws.near_resolved(int(torch.cuda.current_stream().cuda_stream)), "of", n2, "queries in the last step")
tr = ws.trace(10)
print("matched/kept per iter:", [(int(t[0]), int(t[1])) for t in tr])
