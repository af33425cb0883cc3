"""livescan3d_amd -- MI355X-native drop-in for LiveScan3D's NativeUtils fusion path.

The product is the C-ABI shared library livescan3d_amd/lib/libNativeUtils.so (HIP, gfx950), declared in
include/NativeUtils.h.  This package only holds what surrounds it:

  csrc/      hand-written HIP kernels + the C-ABI
  native.py   ctypes binding of the C-ABI (mirror of LiveScanServer's P/Invoke declarations; generate_mesh_from_depth_maps /
              icp / refine mirror KinectServer.GenerateMesh, MainWindowForm's ICP call and refineWorker_DoWork)
  fusion.py   torch-buffer convenience wrapper around native.FusionPlan (device-resident batches)
  sharding.py ShardedFusion = thin caller of the library's lsnShard* exports (sensor block per GPU, RCCL inside the
              library); the same protocol over torch.distributed for CPU rehearsals and comparison legs
  replay.py   the reference's capture / replay file format (depthprocessing.cpp:1316-1341)
  synth.py    seeded synthetic Kinect-like inputs

There is no CPU fallback: every compute entry point raises when the HIP library or a GPU is missing.
"""
from . import native  # noqa: F401

__all__ = ["native"]
