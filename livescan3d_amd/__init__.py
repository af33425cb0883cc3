"""livescan3d_amd -- MI355X-native drop-in for LiveScan3D's NativeUtils fusion path.

The product is the C-ABI shared library livescan3d_amd/lib/libNativeUtils.so (HIP, gfx950), declared in
include/NativeUtils.h.  This package only holds what surrounds it:

  csrc/      hand-written HIP kernels + the C-ABI
  native.py  ctypes binding of the C-ABI (mirror of LiveScanServer's P/Invoke declarations)
  server.py  host-side mirror of the reference callers (KinectServer.GenerateMesh, refineWorker_DoWork)
  sharding.py one-sensor-per-GPU sharding + all-gather of the merged cloud (torch.distributed / RCCL)
  synth.py   seeded synthetic Kinect-like inputs

There is no CPU fallback: every compute entry point raises when the HIP library or a GPU is missing.
"""
from . import native  # noqa: F401

__all__ = ["native"]
