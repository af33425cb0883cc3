#!/usr/bin/env python3
"""Dev helper (a build with -DLSN_CULL_STAMPS): where a workgroup of nn_cull_kernel<true> spends its time (100 MHz clock stamps
of the LAST seeded step of an ICP run): phase A | barrier | near phases | barrier | group info | list append."""
import ctypes as C, os, sys
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
src = xyz[int(off[0]):int(off[1])].contiguous().clone()
tgt = xyz[int(off[1]):].contiguous()
n1, n2 = tgt.shape[0], src.shape[0]
ws = native.IcpWorkspace(0, n1, n2)
Rt = torch.tensor([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=torch.float32, device="cuda")
for rep in range(2):
    ws.run(tgt.data_ptr(), n1, src.data_ptr(), n2, Rt.data_ptr(), Rt.data_ptr() + 36, 5, native.NN_GRID, int(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
nb = (n2 + 63) // 64
st = np.zeros((min(nb, 8192), 8), np.int64)
L = native.lib()
assert L.lsnDevCullStamps(st.ctypes.data_as(C.c_void_p), int(nb)) == 0
t0 = st[:, 0].min()
names = ["start->A done", "A done->after barrier", "team phase (runs -> chunks)", "chunk walk", "walk done->group info", "group info->list written"]
print(f"n2={n2} groups={nb}; first workgroup starts at 0, last at {(st[:,0].max()-t0)/100:.2f} us; stamps in us (100 MHz clock)")
last = st[:, 0].copy()
for k, nm in enumerate(names):
    b = st[:, k + 1]
    ok = b >= last          # a stamp the build / the path did not take keeps an older value
    dd = (b - last)[ok] / 100.0
    if ok.sum():
        print(f"  {nm:28s} n={ok.sum():5d} mean {dd.mean():6.2f} p50 {np.median(dd):6.2f} p95 {np.percentile(dd,95):6.2f} max {dd.max():6.2f}")
    last = np.where(ok, b, last)
print(f"  workgroup total: mean {((last-st[:,0])/100).mean():.2f} max {((last-st[:,0])/100).max():.2f}; kernel span {(last.max()-t0)/100:.2f} us")
