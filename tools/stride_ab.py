#!/usr/bin/env python3
"""A/B: the device-resident fusion step with padded per-tick strides (lsnFusionSetTickStrides) and the single-pass kernel's block orders.

    python3 tools/stride_ab.py <mode 0|2> <pad_bytes> [steps]        ($LSN_FUSE_CHUNK: block order of mode 2; 1 = ticks fastest)

The 64 ticks of a step lie a multiple of 32-512 KB apart when packed (depth 53 x 64 KB, colours 159 x 32 KB, vertices 53 x 512 KB):
block orders that run the same tile of many ticks at once then hit the same HBM channels."""
import ctypes as C
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402,F401
import torch  # noqa: E402

from livescan3d_amd import native, synth  # noqa: E402

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
pad = int(sys.argv[2]) if len(sys.argv) > 2 else 0
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
T, S, w, h = 64, 8, 512, 424
P = w * h
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
rig = synth.make_rig("noise", S, w, h, seed=1, bounds=synth.CROP_BOUNDS)
plan = native.FusionPlan(0, T, rig.widths, rig.heights)
plan.set_mode(mode)
plan.set_params(rig.intr, rig.wt, rig.bounds, 0)
cap = plan.capacity
sd, sc, sv = S * P + pad // 2, S * P * 3 + pad, cap + pad // 16          # u16 elements, bytes, vertices
L = native.lib()
if pad:
    if not hasattr(L, "lsnFusionSetTickStrides"):
        sys.exit("padded strides need lsnFusionSetTickStrides, the stride override that was built for this measurement and removed after it "
                 "(EXPERIMENTS.md R4-7; `git log -S lsnFusionSetTickStrides`): run with pad 0, or restore it")
    L.lsnFusionSetTickStrides.restype = C.c_int
    L.lsnFusionSetTickStrides.argtypes = [C.c_void_p, C.c_longlong, C.c_longlong, C.c_longlong]
    assert L.lsnFusionSetTickStrides(plan._h, sd, sc, sv) == 0, native.last_error()
d0, c0 = synth.noise_frames_torch(dev, 1, T, S, w, h)
depth = torch.zeros(T * sd, dtype=torch.int16, device=dev)
rgb = torch.zeros(T * sc, dtype=torch.uint8, device=dev)
depth.view(T, sd)[:, :S * P] = d0.view(T, S * P)
rgb.view(T, sc)[:, :S * P * 3] = c0.view(T, S * P * 3)
verts = torch.zeros(T * sv * 16, dtype=torch.uint8, device=dev)
offs = torch.zeros((T, S + 1), dtype=torch.int32, device=dev)
st = int(torch.cuda.current_stream().cuda_stream)


def run():
    plan.run(depth.data_ptr(), rgb.data_ptr(), verts.data_ptr(), offs.data_ptr(), st)


for _ in range(30):
    run()
torch.cuda.synchronize()
t_end = time.perf_counter() + 0.5
while time.perf_counter() < t_end:
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
V = int(offs[:, -1].sum().item())
alg = 2 * P * S * T + 19 * V
n0, nl = int(offs[0, -1]), int(offs[T - 1, -1])       # digest of the first and last tick's cloud, to compare variants
hsh = hashlib.sha256(verts[:n0 * 16].cpu().numpy().tobytes() + verts[(T - 1) * sv * 16:(T - 1) * sv * 16 + nl * 16].cpu().numpy().tobytes()).hexdigest()[:12]
print(f"mode {mode} pad {pad:6d} B chunk {os.environ.get('LSN_FUSE_CHUNK', '-'):>3s}: {1e3 * dt:.4f} ms/step  {T / dt / 1e3:7.1f} k frames/s  step_frac {alg / dt / 1e9 / 8000:.3f}  "
      f"failed {plan.lookback_failed(st) if mode else 0}  digest {hsh}")
