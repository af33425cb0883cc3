"""ctypes binding of libNativeUtils.so (include/NativeUtils.h).

Part 1 mirrors LiveScanServer's P/Invoke declarations (LiveScanServer/KinectServer.cs:35-60,
MainWindowForm.cs:42-43): same entry points, same array conventions.  Part 2 binds the device-resident
API; device pointers are plain integers (e.g. torch.Tensor.data_ptr()).

The library is loaded from livescan3d_amd/lib/ (built in-tree by __graft_entry__.build()).  If it is missing
or a call fails, NativeUtilsError is raised -- nothing here computes on the CPU.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# $LSN_NATIVE_LIB: another build of the same library (development A/Bs); the default is the in-tree build
LIB_PATH = os.environ.get("LSN_NATIVE_LIB") or os.path.join(_HERE, "lib", "libNativeUtils.so")

VERTEX_DTYPE = np.dtype([("R", "u1"), ("G", "u1"), ("B", "u1"), ("A", "u1"),
                         ("X", "<f4"), ("Y", "<f4"), ("Z", "<f4")])  # VertexC4ubV3f, 16 bytes

# every symbol include/NativeUtils.h declares
EXPORTS = [
    "generateVerticesFromDepthMap", "generateMeshFromDepthMaps", "depthMapAndColorSetRadialCorrection", "createMesh", "deleteMesh", "ICP",
    "lsnGetLastError", "lsnDeviceCount", "lsnCorrectAndGenerateMesh", "lsnHostScheduleDescribe", "lsnHostShardDescribe", "lsnHostShardPartMicros", "lsnTestFaultPoints", "lsnHostPoolStats",
    "lsnFusionCreate", "lsnFusionDestroy", "lsnFusionTickCapacity", "lsnFusionSetParams", "lsnPackSensorParams", "lsnFusionSetMode",
    "lsnFusionRun", "lsnFusionRunStreamed", "lsnFusionSetPipelined", "lsnFusionRadialCorrect", "lsnFusionRadialCorrectTo", "lsnFusionRadialCountersLeft", "lsnFusionRunMesh", "lsnFusionTickTriangleCapacity", "lsnFusionProfile", "lsnFusionKernelStats", "lsnFusionLookbackFailed", "lsnFusionCheck", "lsnFusionThresholds", "lsnMergeShards",
    "lsnFusionTilesPerTick", "lsnFusionPackSurvivors", "lsnFusionReconstruct",
    "lsnDeviceMalloc", "lsnDeviceFree", "lsnDeviceUpload", "lsnDeviceDownload", "lsnStreamCreate", "lsnStreamDestroy", "lsnStreamSynchronize",
    "lsnFusionPackSurvivorsRun", "lsnFusionReconstructRun", "lsnShardUniqueId", "lsnShardPlan", "lsnShardCreate", "lsnShardPrepare", "lsnShardConnect", "lsnShardRcclPath", "lsnShardDestroy", "lsnShardMergedCapacity", "lsnShardSetParams", "lsnShardStep", "lsnShardLastBytesSent", "lsnShardRanksSeen",
    "lsnIcpCreate", "lsnIcpDestroy", "lsnIcpRun", "lsnIcpNearest", "lsnIcpTrace", "lsnIcpSetProfiling", "lsnIcpProfile", "lsnIcpNearResolved", "lsnRefine",
    "lsnTickCreate", "lsnTickDestroy", "lsnTickSetParams", "lsnTickCapacity", "lsnTickTriangleCapacity", "lsnTickParts", "lsnTickRun",
    "lsnTransferCreate", "lsnTransferDestroy", "lsnTransferFrameBound", "lsnTransferPack", "lsnTransferLastPath", "lsnPlyBinaryBytes", "lsnPlyPack",
    "lsnLastMeshTransferFrame", "lsnLastMeshPly",
    "lsnZstdAvailable", "lsnFrameParseHeader", "lsnFrameDecode", "lsnFrameEncode", "lsnRecordingNext", "lsnRecordingAppend",
]


class NativeUtilsError(RuntimeError):
    pass


class Mesh(C.Structure):
    """struct Mesh (include/NativeUtils/depthprocessing.h:42-48; C# mirror Utils.cs:335-342)."""
    _fields_ = [("nVertices", C.c_int), ("vertices", C.c_void_p), ("nTriangles", C.c_int), ("triangles", C.c_void_p)]


assert C.sizeof(Mesh) == 32


class FrameInfo(C.Structure):
    """LsnFrameInfo: the 16-byte header of a frame message (KinectSocket.cs:229-239)."""
    _fields_ = [("payload_bytes", C.c_int), ("compressed", C.c_int), ("width", C.c_int), ("height", C.c_int)]

_lib = None


def lib():
    """Loads libNativeUtils.so and declares the prototypes.  Raises NativeUtilsError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeUtilsError(f"{LIB_PATH} is missing -- run __graft_entry__.build() (there is no CPU fallback)")
    # PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 under the same SONAME as /opt/rocm's.  A
    # process must run ONE HIP runtime (device pointers handed over by torch have to belong to the runtime that launches
    # our kernels), and the dynamic loader keeps whichever copy is loaded first -- so when torch is installed it is
    # imported before the library.  Hosts without torch (LiveScanServer, C/C++ callers) simply get /opt/rocm's runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise NativeUtilsError(f"cannot load {LIB_PATH}: {e}") from e
    vp, fp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)
    f = C.c_float
    L.generateVerticesFromDepthMap.restype = None
    L.generateVerticesFromDepthMap.argtypes = [vp, vp, vp, vp, vp, vp, C.POINTER(Mesh), f, f, f, f, f, f, C.c_int]
    L.generateMeshFromDepthMaps.restype = None
    L.generateMeshFromDepthMaps.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp, C.POINTER(Mesh), C.c_bool,
                                            f, f, f, f, f, f, C.c_bool]
    L.lsnCorrectAndGenerateMesh.restype = None
    L.lsnCorrectAndGenerateMesh.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp, C.POINTER(Mesh), f, f, f, f, f, f, C.c_int]
    L.depthMapAndColorSetRadialCorrection.restype = None
    L.depthMapAndColorSetRadialCorrection.argtypes = [C.c_int, vp, vp, vp, vp, vp]
    L.lsnFusionRadialCorrect.restype = C.c_int
    L.lsnFusionRadialCorrect.argtypes = [vp, vp, vp, vp, vp]
    L.lsnFusionRadialCountersLeft.restype = C.c_int
    L.lsnFusionRadialCountersLeft.argtypes = [vp, vp]
    L.lsnFusionRadialCorrectTo.restype = C.c_int
    L.lsnFusionRadialCorrectTo.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.createMesh.restype = C.POINTER(Mesh)
    L.createMesh.argtypes = []
    L.deleteMesh.restype = None
    L.deleteMesh.argtypes = [C.POINTER(Mesh)]
    L.ICP.restype = C.c_float
    L.ICP.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, C.c_int]
    L.lsnGetLastError.restype = C.c_int
    L.lsnGetLastError.argtypes = [C.c_char_p, C.c_int]
    L.lsnDeviceCount.restype = C.c_int
    L.lsnDeviceCount.argtypes = []
    L.lsnFusionCreate.restype = vp
    L.lsnFusionCreate.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp]
    L.lsnFusionDestroy.restype = None
    L.lsnFusionDestroy.argtypes = [vp]
    L.lsnFusionTickCapacity.restype = C.c_longlong
    L.lsnFusionTickCapacity.argtypes = [vp]
    L.lsnFusionSetParams.restype = C.c_int
    L.lsnFusionSetParams.argtypes = [vp, vp, vp, vp, vp]
    L.lsnPackSensorParams.restype = C.c_int
    L.lsnPackSensorParams.argtypes = [vp, vp, vp]
    L.lsnHostShardDescribe.restype = C.c_int
    L.lsnHostShardDescribe.argtypes = [C.c_int, C.c_int, vp, C.c_char_p, C.c_int]
    L.lsnHostShardPartMicros.restype = C.c_int
    L.lsnHostShardPartMicros.argtypes = [vp, C.c_int]
    L.lsnTestFaultPoints.restype = C.c_longlong
    L.lsnTestFaultPoints.argtypes = [C.c_int]
    L.lsnHostPoolStats.restype = C.c_int
    L.lsnHostPoolStats.argtypes = [vp, vp, vp]
    L.lsnHostScheduleDescribe.restype = C.c_int
    L.lsnHostScheduleDescribe.argtypes = [C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
    L.lsnFusionSetMode.restype = C.c_int
    L.lsnFusionSetMode.argtypes = [vp, C.c_int]
    L.lsnFusionRunStreamed.restype = C.c_int
    L.lsnFusionRunStreamed.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.lsnFusionSetPipelined.restype = C.c_int
    L.lsnFusionSetPipelined.argtypes = [vp, C.c_int]
    L.lsnFusionRun.restype = C.c_int
    L.lsnFusionRun.argtypes = [vp, vp, vp, vp, vp, vp]
    L.lsnFusionRunMesh.restype = C.c_int
    L.lsnFusionRunMesh.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.lsnFusionTickTriangleCapacity.restype = C.c_longlong
    L.lsnFusionTickTriangleCapacity.argtypes = [vp]
    L.lsnFusionProfile.restype = C.c_int
    L.lsnFusionProfile.argtypes = [vp, C.c_int]
    L.lsnFusionKernelStats.restype = C.c_int
    L.lsnFusionKernelStats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.c_char_p, C.c_int, C.c_int]
    L.lsnFusionThresholds.restype = C.c_int
    L.lsnFusionThresholds.argtypes = [vp, vp, C.POINTER(C.c_float), vp]
    L.lsnFusionTilesPerTick.restype = C.c_int
    L.lsnFusionTilesPerTick.argtypes = [vp]
    L.lsnFusionPackSurvivors.restype = C.c_int
    L.lsnFusionPackSurvivors.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.lsnFusionReconstruct.restype = C.c_int
    L.lsnFusionReconstruct.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, C.c_longlong, vp, vp, vp, vp, vp]
    L.lsnFusionLookbackFailed.restype = C.c_int
    L.lsnFusionLookbackFailed.argtypes = [vp, vp]
    L.lsnFusionCheck.restype = C.c_int
    L.lsnFusionCheck.argtypes = [vp, vp]
    L.lsnMergeShards.restype = C.c_int
    L.lsnMergeShards.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_longlong, vp, vp, C.c_longlong, vp, vp]
    L.lsnIcpCreate.restype = vp
    L.lsnIcpCreate.argtypes = [C.c_int, C.c_int, C.c_int]
    L.lsnIcpDestroy.restype = None
    L.lsnIcpDestroy.argtypes = [vp]
    L.lsnIcpRun.restype = C.c_int
    L.lsnIcpRun.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp]
    L.lsnIcpNearest.restype = C.c_int
    L.lsnIcpNearest.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, vp]
    L.lsnRefine.restype = C.c_int
    L.lsnRefine.argtypes = [C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]
    L.lsnDeviceMalloc.restype = vp
    L.lsnDeviceMalloc.argtypes = [C.c_int, C.c_longlong]
    L.lsnDeviceFree.restype = C.c_int
    L.lsnDeviceFree.argtypes = [C.c_int, vp]
    L.lsnDeviceUpload.restype = C.c_int
    L.lsnDeviceUpload.argtypes = [C.c_int, vp, vp, C.c_longlong, vp]
    L.lsnDeviceDownload.restype = C.c_int
    L.lsnDeviceDownload.argtypes = [C.c_int, vp, vp, C.c_longlong, vp]
    L.lsnStreamCreate.restype = vp
    L.lsnStreamCreate.argtypes = [C.c_int]
    L.lsnStreamDestroy.restype = C.c_int
    L.lsnStreamDestroy.argtypes = [C.c_int, vp]
    L.lsnStreamSynchronize.restype = C.c_int
    L.lsnStreamSynchronize.argtypes = [C.c_int, vp]
    L.lsnFusionPackSurvivorsRun.restype = C.c_int
    L.lsnFusionPackSurvivorsRun.argtypes = [vp] * 10
    L.lsnFusionReconstructRun.restype = C.c_int
    L.lsnFusionReconstructRun.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, C.c_longlong, vp, vp, vp, vp, vp, vp]
    L.lsnShardUniqueId.restype = C.c_int
    L.lsnShardUniqueId.argtypes = [vp]
    L.lsnShardCreate.restype = vp
    L.lsnShardCreate.argtypes = [C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp]
    L.lsnShardPrepare.restype = vp
    L.lsnShardPrepare.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    L.lsnShardConnect.restype = C.c_int
    L.lsnShardConnect.argtypes = [vp, vp]
    L.lsnShardRcclPath.restype = C.c_int
    L.lsnShardRcclPath.argtypes = [C.c_char_p, C.c_int]
    L.lsnShardPlan.restype = vp
    L.lsnShardPlan.argtypes = [vp, C.c_int]
    L.lsnShardDestroy.restype = None
    L.lsnShardDestroy.argtypes = [vp]
    L.lsnShardMergedCapacity.restype = C.c_longlong
    L.lsnShardMergedCapacity.argtypes = [vp]
    L.lsnShardLastBytesSent.restype = C.c_longlong
    L.lsnShardLastBytesSent.argtypes = [vp]
    L.lsnShardRanksSeen.restype = C.c_int
    L.lsnShardRanksSeen.argtypes = [vp]
    L.lsnShardSetParams.restype = C.c_int
    L.lsnShardSetParams.argtypes = [vp, vp, vp, vp, vp]
    L.lsnShardStep.restype = C.c_int
    L.lsnShardStep.argtypes = [vp, vp, vp, vp, vp, vp]
    L.lsnIcpTrace.restype = C.c_int
    L.lsnIcpTrace.argtypes = [vp, vp, C.c_int, vp]
    L.lsnIcpSetProfiling.restype = C.c_int
    L.lsnIcpSetProfiling.argtypes = [vp, C.c_int]
    L.lsnIcpProfile.restype = C.c_int
    L.lsnIcpProfile.argtypes = [vp, vp, vp]
    if hasattr(L, "lsnTickCreate"):
        L.lsnTickCreate.restype = vp
        L.lsnTickCreate.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp]
        L.lsnTickDestroy.restype = None
        L.lsnTickDestroy.argtypes = [vp]
        L.lsnTickSetParams.restype = C.c_int
        L.lsnTickSetParams.argtypes = [vp, vp, vp, vp, vp]
        L.lsnTickCapacity.restype = C.c_longlong
        L.lsnTickCapacity.argtypes = [vp]
        L.lsnTickTriangleCapacity.restype = C.c_longlong
        L.lsnTickTriangleCapacity.argtypes = [vp]
        L.lsnTickParts.restype = C.c_int
        L.lsnTickParts.argtypes = [vp]
        L.lsnTickRun.restype = C.c_int
        L.lsnTickRun.argtypes = [vp] + [vp] * 9
    if hasattr(L, "lsnIcpNearResolved"):   # (absent from an older build loaded through $LSN_NATIVE_LIB)
        L.lsnIcpNearResolved.restype = C.c_int
        L.lsnIcpNearResolved.argtypes = [vp, vp]
    ll = C.c_longlong
    L.lsnTransferCreate.restype = vp
    L.lsnTransferCreate.argtypes = [C.c_int, C.c_int, C.c_int]
    L.lsnTransferDestroy.restype = None
    L.lsnTransferDestroy.argtypes = [vp]
    L.lsnTransferLastPath.restype = C.c_int
    L.lsnTransferLastPath.argtypes = [C.c_void_p]
    L.lsnTransferFrameBound.restype = ll
    L.lsnTransferFrameBound.argtypes = [C.c_int, C.c_int]
    L.lsnTransferPack.restype = ll
    L.lsnTransferPack.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, ll, vp]
    L.lsnPlyBinaryBytes.restype = ll
    L.lsnPlyBinaryBytes.argtypes = [C.c_int, C.c_int]
    L.lsnPlyPack.restype = ll
    L.lsnPlyPack.argtypes = [C.c_int, vp, C.c_int, vp, C.c_int, vp, ll, vp]
    L.lsnLastMeshTransferFrame.restype = ll
    L.lsnLastMeshTransferFrame.argtypes = [vp, ll]
    L.lsnLastMeshPly.restype = ll
    L.lsnLastMeshPly.argtypes = [vp, ll]
    L.lsnZstdAvailable.restype = C.c_int
    L.lsnZstdAvailable.argtypes = []
    L.lsnFrameParseHeader.restype = C.c_int
    L.lsnFrameParseHeader.argtypes = [vp, C.POINTER(FrameInfo)]
    L.lsnFrameDecode.restype = ll
    L.lsnFrameDecode.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, C.POINTER(C.c_int)]
    L.lsnFrameEncode.restype = ll
    L.lsnFrameEncode.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, ll]
    L.lsnRecordingNext.restype = ll
    L.lsnRecordingNext.argtypes = [vp, ll, ll, C.POINTER(ll), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.lsnRecordingAppend.restype = ll
    L.lsnRecordingAppend.argtypes = [vp, ll, vp, C.c_int, C.c_int]
    _lib = L
    return L


def last_error():
    buf = C.create_string_buffer(1024)
    lib().lsnGetLastError(buf, len(buf))
    return buf.value.decode("utf-8", "replace")


def _check(rc, what):
    if rc != 0:
        raise NativeUtilsError(f"{what} failed: {last_error()}")


def device_count():
    return int(lib().lsnDeviceCount())


def require_gpu():
    if device_count() <= 0:
        raise NativeUtilsError("no HIP device visible: libNativeUtils has no CPU path")


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _as(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# ----------------------------------------------------------------------------------------------------------
# Part 1: the reference's exports (host buffers in, host buffers out)
# ----------------------------------------------------------------------------------------------------------

def _copy_mesh(mesh):
    """KinectServer.CopyMeshToVerticesWithColoursArray (KinectServer.cs:376-389) + deleteMesh."""
    n = mesh.nVertices
    if n > 0:
        if not mesh.vertices:
            raise NativeUtilsError("mesh has nVertices > 0 but a null vertices pointer")
        raw = C.string_at(mesh.vertices, n * 16)
        verts = np.frombuffer(raw, dtype=VERTEX_DTYPE).copy()
    else:
        verts = np.zeros(0, dtype=VERTEX_DTYPE)
    ntri = mesh.nTriangles
    tris = np.zeros((0, 3), dtype=np.int32)
    if ntri > 0:
        tris = np.frombuffer(C.string_at(mesh.triangles, ntri * 12), dtype=np.int32).reshape(-1, 3).copy()
    lib().deleteMesh(C.byref(mesh))
    return verts, tris


def generate_mesh_from_depth_maps(depth_maps, depth_colors, widths, heights, intr, wt, bounds,
                                  color_transfer=False, generate_triangles=False):
    """KinectServer.GenerateMesh (KinectServer.cs:354-374).  Returns (vertices[VERTEX_DTYPE], triangles int32)."""
    require_gpu()
    widths, heights = _as(widths, np.int32), _as(heights, np.int32)
    n = len(widths)
    dm = np.ascontiguousarray(depth_maps).view(np.uint8).ravel()
    dc = _as(depth_colors, np.uint8).ravel()
    intr, wt, b = _as(intr, np.float32).ravel(), _as(wt, np.float32).ravel(), _as(bounds, np.float32).ravel()
    assert intr.size == 7 * n and wt.size == 12 * n and b.size == 6
    mesh = Mesh()
    lib().generateMeshFromDepthMaps(n, _ptr(dm), _ptr(dc), _ptr(widths), _ptr(heights), _ptr(intr), _ptr(wt),
                                    C.byref(mesh), bool(color_transfer), *[float(x) for x in b], bool(generate_triangles))
    err = last_error()
    if err and mesh.nVertices == 0 and not (color_transfer or generate_triangles):
        lib().deleteMesh(C.byref(mesh))
        raise NativeUtilsError(err)
    return _copy_mesh(mesh)


def host_shards(n_maps, n_devices=0):
    """How a merge call over n_maps sensors is cut over n_devices devices of $LSN_HOST_DEVICES (lsnHostShardDescribe; needs no GPU).
    Returns (block bounds [first_0, ..., first_D], text such as "0:[0-3] 1:[4-7]")."""
    buf = C.create_string_buffer(1024)
    first = (C.c_int * 18)()
    d = lib().lsnHostShardDescribe(int(n_maps), int(n_devices), first, buf, len(buf))
    if d < 0:
        raise NativeUtilsError(last_error())
    return [int(first[i]) for i in range(d + 1)], buf.value.decode()


def host_shard_part_micros():
    """Wall time (us) of every part of the last sharded call when the parts ran one after the other ($LSN_HOST_SHARD_SOLO=1); [] if none."""
    out = (C.c_longlong * 16)()
    d = lib().lsnHostShardPartMicros(out, 16)
    return [int(out[i]) for i in range(max(0, d))]


def host_pool_stats():
    """(blocks out with callers, blocks waiting for reuse, bytes out) of the pool of pinned mesh blocks (lsnHostPoolStats)."""
    live, pooled, nbytes = C.c_int(0), C.c_int(0), C.c_longlong(0)
    _check(lib().lsnHostPoolStats(C.byref(live), C.byref(pooled), C.byref(nbytes)), "lsnHostPoolStats")
    return live.value, pooled.value, nbytes.value


def host_schedule(widths, heights, first=0, count=None, radial=False, sensors_per_group=0):
    """The upload schedule the host exports follow for these frames (lsnHostScheduleDescribe; needs no GPU).
    Returns (number of groups, text such as "D[0-2] C[0-2] | D[3-7] C[3-5] | C[6-7]")."""
    widths, heights = _as(widths, np.int32), _as(heights, np.int32)
    n = len(widths)
    buf = C.create_string_buffer(4096)
    g = lib().lsnHostScheduleDescribe(n, _ptr(widths), _ptr(heights), int(first), int(n - first if count is None else count), int(radial),
                                      int(sensors_per_group), buf, len(buf))
    if g < 0:
        raise NativeUtilsError(last_error())
    return g, buf.value.decode()


def radial_correction(depth_maps, depth_colors, widths, heights, intr):
    """KinectServer.CorrectRadialDistortionsForDepthMaps (KinectServer.cs:518-525): returns corrected copies
    (depth as a uint8 view of the u16 maps, colours); the export itself works in place on the arrays it is given."""
    require_gpu()
    widths, heights = _as(widths, np.int32), _as(heights, np.int32)
    n = len(widths)
    dm = np.ascontiguousarray(depth_maps).view(np.uint8).ravel().copy()
    dc = _as(depth_colors, np.uint8).ravel().copy()
    intr = _as(intr, np.float32).ravel()
    assert intr.size == 7 * n
    lib().depthMapAndColorSetRadialCorrection(n, _ptr(dm), _ptr(dc), _ptr(widths), _ptr(heights), _ptr(intr))
    err = last_error()
    if err:
        raise NativeUtilsError(err)
    return dm, dc


def correct_and_generate_mesh(depth_maps, depth_colors, widths, heights, intr, wt, bounds, write_back=True):
    """One call per tick (extension): radial correction + merge call with a single upload.  Returns (vertices, triangles,
    corrected depth as uint8, corrected colours); with write_back=False the last two are the untouched inputs."""
    require_gpu()
    widths, heights = _as(widths, np.int32), _as(heights, np.int32)
    n = len(widths)
    dm = np.ascontiguousarray(depth_maps).view(np.uint8).ravel().copy()
    dc = _as(depth_colors, np.uint8).ravel().copy()
    intr, wt, b = _as(intr, np.float32).ravel(), _as(wt, np.float32).ravel(), _as(bounds, np.float32).ravel()
    assert intr.size == 7 * n and wt.size == 12 * n and b.size == 6
    mesh = Mesh()
    lib().lsnCorrectAndGenerateMesh(n, _ptr(dm), _ptr(dc), _ptr(widths), _ptr(heights), _ptr(intr), _ptr(wt), C.byref(mesh),
                                    *[float(x) for x in b], 1 if write_back else 0)
    err = last_error()
    if err and mesh.nVertices == 0:
        lib().deleteMesh(C.byref(mesh))
        raise NativeUtilsError(err)
    v, t = _copy_mesh(mesh)
    return v, t, dm, dc


def generate_vertices_from_depth_map(depth_maps, depth_colors, widths, heights, intr, wt, bounds, index):
    """One sensor's cropped cloud, as KinectServer.GetLatestFrameVerticesOnly calls it (KinectServer.cs:527-554)."""
    require_gpu()
    widths, heights = _as(widths, np.int32), _as(heights, np.int32)
    n = len(widths)
    dm = np.ascontiguousarray(depth_maps).view(np.uint8).ravel()
    dc = _as(depth_colors, np.uint8).ravel()
    intr, wt, b = _as(intr, np.float32).ravel(), _as(wt, np.float32).ravel(), _as(bounds, np.float32).ravel()
    assert intr.size == 7 * n and wt.size == 12 * n and b.size == 6
    mesh = Mesh()
    lib().generateVerticesFromDepthMap(_ptr(dm), _ptr(dc), _ptr(widths), _ptr(heights), _ptr(intr), _ptr(wt),
                                       C.byref(mesh), *[float(x) for x in b], int(index))
    err = last_error()
    if err and mesh.nVertices == 0:
        lib().deleteMesh(C.byref(mesh))
        raise NativeUtilsError(err)
    return _copy_mesh(mesh)[0]


def icp(verts1, verts2, R=None, t=None, max_iter=10):
    """ICP export (MainWindowForm.cs:42-43,370).  Returns (verts2_out, R_out[3,3], t_out[3]); inputs are not modified."""
    require_gpu()
    v1 = _as(verts1, np.float32).reshape(-1, 3)
    v2 = _as(verts2, np.float32).reshape(-1, 3).copy()
    R = np.eye(3, dtype=np.float32).ravel() if R is None else _as(R, np.float32).ravel().copy()
    t = np.zeros(3, dtype=np.float32) if t is None else _as(t, np.float32).ravel().copy()
    lib().ICP(_ptr(v1), _ptr(v2), len(v1), len(v2), _ptr(R), _ptr(t), int(max_iter))
    err = last_error()
    if err:
        raise NativeUtilsError(err)
    return v2, R.reshape(3, 3), t


def refine(clouds, world_R, world_t, n_refine_iters=2, n_icp_iters=10, device=0):
    """refineWorker_DoWork (MainWindowForm.cs:330-410) through lsnRefine.  clouds: list of [n_i, 3] float arrays.
    Returns (clouds_out, world_R [n,3,3], world_t [n,3], Rs [n,3,3], Ts [n,3]); inputs are not modified."""
    require_gpu()
    cl = [_as(c, np.float32).reshape(-1, 3).copy() for c in clouds]
    n = np.array([len(c) for c in cl], dtype=np.int32)
    ptrs = (C.c_void_p * len(cl))(*[c.ctypes.data for c in cl])
    wR = _as(world_R, np.float32).reshape(-1).copy()
    wt = _as(world_t, np.float32).reshape(-1).copy()
    assert wR.size == 9 * len(cl) and wt.size == 3 * len(cl)
    Rs = np.zeros(9 * len(cl), dtype=np.float32)
    Ts = np.zeros(3 * len(cl), dtype=np.float32)
    _check(lib().lsnRefine(int(device), len(cl), C.cast(ptrs, C.c_void_p), _ptr(n), int(n_refine_iters), int(n_icp_iters),
                           _ptr(wR), _ptr(wt), _ptr(Rs), _ptr(Ts)), "lsnRefine")
    return cl, wR.reshape(-1, 3, 3), wt.reshape(-1, 3), Rs.reshape(-1, 3, 3), Ts.reshape(-1, 3)


# ----------------------------------------------------------------------------------------------------------
# Part 2: device-resident API
# ----------------------------------------------------------------------------------------------------------

class FusionPlan:
    """lsnFusion*: T ticks x N sensors fused per call on HBM-resident inputs."""

    def __init__(self, device, n_ticks, widths, heights):
        require_gpu()
        self.widths, self.heights = _as(widths, np.int32), _as(heights, np.int32)
        self.n_maps, self.n_ticks, self.device = len(self.widths), int(n_ticks), int(device)
        self._h = lib().lsnFusionCreate(self.device, self.n_ticks, self.n_maps, _ptr(self.widths), _ptr(self.heights))
        if not self._h:
            raise NativeUtilsError(f"lsnFusionCreate failed: {last_error()}")
        self.capacity = int(lib().lsnFusionTickCapacity(self._h))
        self.pixels_per_tick = int(np.sum(self.widths.astype(np.int64) * self.heights))

    def set_params(self, intr, wt, bounds, stream=0):
        intr, wt, b = _as(intr, np.float32).ravel(), _as(wt, np.float32).ravel(), _as(bounds, np.float32).ravel()
        assert intr.size == 7 * self.n_maps and wt.size == 12 * self.n_maps and b.size == 6
        _check(lib().lsnFusionSetParams(self._h, _ptr(intr), _ptr(wt), _ptr(b), stream), "lsnFusionSetParams")

    def set_pipelined(self, enable=True):
        """Overlap the count pass of the next call with the write kernel of the current one (inputs must be resident)."""
        _check(lib().lsnFusionSetPipelined(self._h, 1 if enable else 0), "lsnFusionSetPipelined")

    def set_mode(self, mode):
        _check(lib().lsnFusionSetMode(self._h, int(mode)), "lsnFusionSetMode")

    def run(self, d_depth, d_colors, d_vertices, d_offsets, stream=0):
        """All four are device pointers (ints); asynchronous on `stream` (a hipStream_t as int, 0 = null stream)."""
        _check(lib().lsnFusionRun(self._h, d_depth, d_colors, d_vertices, d_offsets, stream), "lsnFusionRun")

    def radial_correct(self, intr, d_depth, d_colors, stream=0):
        """In-place radial correction of the plan's n_ticks x n_maps frames resident in HBM."""
        intr = _as(intr, np.float32).ravel()
        assert intr.size == 7 * self.n_maps
        _check(lib().lsnFusionRadialCorrect(self._h, _ptr(intr), d_depth, d_colors, stream), "lsnFusionRadialCorrect")

    def radial_correct_to(self, intr, d_depth, d_colors, d_depth_out, d_colors_out, stream=0):
        """Out-of-place radial correction (the cheaper form: the un-closed maps never leave the GPU's LDS)."""
        intr = _as(intr, np.float32).ravel()
        assert intr.size == 7 * self.n_maps
        _check(lib().lsnFusionRadialCorrectTo(self._h, _ptr(intr), d_depth, d_colors, d_depth_out, d_colors_out, stream),
               "lsnFusionRadialCorrectTo")

    def radial_counters_left(self, stream=0):
        """Work counters of the hole-closing chain that are not zero once `stream` has drained (test hook; 0 after a complete chain)."""
        n = lib().lsnFusionRadialCountersLeft(self._h, stream)
        if n < 0:
            raise NativeUtilsError(f"lsnFusionRadialCountersLeft failed: {last_error()}")
        return n

    def run_mesh(self, d_depth, d_colors, d_vertices, d_offsets, d_triangles, d_tri_offsets, stream=0):
        """Vertices + triangles (the reference's complete merge call); d_triangles: n_ticks x 2*capacity x 3 int32."""
        _check(lib().lsnFusionRunMesh(self._h, d_depth, d_colors, d_vertices, d_offsets, d_triangles, d_tri_offsets, stream),
               "lsnFusionRunMesh")

    def thresholds(self, capacity=None, stream=0, copy=True):
        """Builds the per-pixel depth thresholds now.  Returns (table uint32[capacity] or None, build_ms); table is None when
        the plan does not use thresholds ($LSN_NO_THRESHOLDS=1)."""
        out = np.zeros(int(capacity or self.capacity), dtype=np.uint32) if copy else None
        ms = C.c_float(0)
        rc = lib().lsnFusionThresholds(self._h, _ptr(out) if copy else None, C.byref(ms), stream or None)
        if rc < 0:
            raise NativeUtilsError(f"lsnFusionThresholds failed: {last_error()}")
        return (None if rc == 1 else out), ms.value

    @property
    def tiles_per_tick(self):
        return int(lib().lsnFusionTilesPerTick(self._h))

    def pack_survivors(self, d_depth, d_colors, d_mask, d_depth_c, d_rgb_c, d_tile_prefix, d_offsets, stream=0):
        _check(lib().lsnFusionPackSurvivors(self._h, d_depth, d_colors, d_mask, d_depth_c, d_rgb_c, d_tile_prefix, d_offsets, stream or None),
               "lsnFusionPackSurvivors")

    def reconstruct(self, n_shards, maps_per_shard, d_masks, d_depth_c, d_rgb_c, slab, d_tile_prefix, d_shard_offsets, d_merged, d_merged_offsets,
                    stream=0):
        """Called on the whole-rig plan: rebuilds all shards' vertices from the gathered survivor streams."""
        _check(lib().lsnFusionReconstruct(self._h, int(n_shards), int(maps_per_shard), d_masks, d_depth_c, d_rgb_c, int(slab), d_tile_prefix,
                                          d_shard_offsets, d_merged, d_merged_offsets, stream or None), "lsnFusionReconstruct")

    def pack_survivors_run(self, d_depth, d_colors, d_mask, d_depth_c, d_rgb_c, d_tile_prefix, d_offsets, d_tick_base, stream=0):
        """pack_survivors with all ticks back to back (one contiguous run per shard); fills d_tick_base [n_ticks]."""
        _check(lib().lsnFusionPackSurvivorsRun(self._h, d_depth, d_colors, d_mask, d_depth_c, d_rgb_c, d_tile_prefix, d_offsets, d_tick_base,
                                               stream or None), "lsnFusionPackSurvivorsRun")

    def reconstruct_run(self, n_shards, maps_per_shard, d_masks, d_depth_c, d_rgb_c, run_len, d_tile_prefix, d_shard_offsets, d_merged,
                        d_merged_offsets, d_tick_base_scratch, stream=0):
        _check(lib().lsnFusionReconstructRun(self._h, int(n_shards), int(maps_per_shard), d_masks, d_depth_c, d_rgb_c, int(run_len), d_tile_prefix,
                                             d_shard_offsets, d_merged, d_merged_offsets, d_tick_base_scratch, stream or None),
               "lsnFusionReconstructRun")

    def lookback_failed(self, stream=0):
        return int(lib().lsnFusionLookbackFailed(self._h, stream))

    def check(self, stream=0):
        """The plan's sticky device-side error flag (0 fine, 1 look-back gave up, 2 inputs changed between count and write); clears it."""
        return int(lib().lsnFusionCheck(self._h, stream))

    def run_streamed(self, d_depth, d_colors, d_vertices, d_offsets, d_next_depth=None, stream=0):
        """This batch is written while the next batch's depth (already resident) is counted in the same kernel."""
        _check(lib().lsnFusionRunStreamed(self._h, d_depth, d_colors, d_vertices, d_offsets, d_next_depth, stream), "lsnFusionRunStreamed")

    def profile(self, enable=True, every=1):
        """HIP events around the dominant kernel of every launch sequence (every = 1) or of every n-th one."""
        _check(lib().lsnFusionProfile(self._h, max(1, int(every)) if enable else 0), "lsnFusionProfile")

    def kernel_stats(self, reset=True):
        avg, n = C.c_double(0), C.c_longlong(0)
        name = C.create_string_buffer(128)
        _check(lib().lsnFusionKernelStats(self._h, C.byref(avg), C.byref(n), name, len(name), 1 if reset else 0),
               "lsnFusionKernelStats")
        return {"kernel": name.value.decode(), "avg_ms": avg.value, "launches": n.value}

    def close(self):
        if self._h:
            lib().lsnFusionDestroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merge_shards(device, n_shards, n_ticks, maps_per_shard, d_shards, shard_cap, d_shard_offsets, d_merged, merged_cap,
                 d_merged_offsets, stream=0):
    _check(lib().lsnMergeShards(int(device), int(n_shards), int(n_ticks), int(maps_per_shard), d_shards, int(shard_cap),
                                d_shard_offsets, d_merged, int(merged_cap), d_merged_offsets, stream), "lsnMergeShards")


def shard_rccl_path():
    """The file the library's nccl* entry points came from (an RCCL already mapped in the process is preferred)."""
    buf = C.create_string_buffer(1024)
    n = lib().lsnShardRcclPath(buf, len(buf))
    return buf.value.decode() if n >= 0 else None


def shard_unique_id():
    """128 bytes from rank 0's RCCL (lsnShardUniqueId); every rank passes the same ones to Shard()."""
    require_gpu()
    buf = (C.c_ubyte * 128)()
    _check(lib().lsnShardUniqueId(buf), "lsnShardUniqueId")
    return bytes(buf)


class Shard:
    """lsnShard*: this rank's block of sensors in, the merged cloud of all sensors out (RCCL all-gathers inside the library)."""

    def __init__(self, device, rank, world, unique_id, n_ticks, widths, heights):
        """unique_id: rank 0's 128 bytes -> prepare + connect at once (lsnShardCreate); None -> lsnShardPrepare only, the caller
        connects (connect()) once every rank has reported that its own preparation worked."""
        require_gpu()
        w, h = _as(widths, np.int32), _as(heights, np.int32)
        self.n_ticks, self.n_maps, self.world, self.rank = int(n_ticks), len(w), int(world), int(rank)
        if unique_id is None:
            self._h = lib().lsnShardPrepare(int(device), self.rank, self.world, self.n_ticks, self.n_maps, _ptr(w), _ptr(h))
            what = "lsnShardPrepare"
        else:
            idb = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
            self._h = lib().lsnShardCreate(int(device), self.rank, self.world, idb, self.n_ticks, self.n_maps, _ptr(w), _ptr(h))
            what = "lsnShardCreate"
        if not self._h:
            raise NativeUtilsError(f"{what} failed: {last_error()}")
        self.capacity = int(lib().lsnShardMergedCapacity(self._h))

    def connect(self, unique_id):
        idb = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        _check(lib().lsnShardConnect(self._h, idb), "lsnShardConnect")

    def set_params(self, intr_all, wt_all, bounds, stream=0):
        intr, wt, b = _as(intr_all, np.float32).ravel(), _as(wt_all, np.float32).ravel(), _as(bounds, np.float32).ravel()
        assert intr.size == 7 * self.n_maps and wt.size == 12 * self.n_maps and b.size == 6
        _check(lib().lsnShardSetParams(self._h, _ptr(intr), _ptr(wt), _ptr(b), stream), "lsnShardSetParams")

    def step(self, d_depth_local, d_colors_local, stream=0):
        """Returns (device pointer of the merged cloud [n_ticks][capacity] vertices, device pointer of its offsets [n_ticks][n_maps + 1])."""
        mv, mo = C.c_void_p(), C.c_void_p()
        _check(lib().lsnShardStep(self._h, d_depth_local, d_colors_local, C.byref(mv), C.byref(mo), stream), "lsnShardStep")
        return mv.value, mo.value

    def last_bytes_sent(self):
        return int(lib().lsnShardLastBytesSent(self._h))

    def ranks_seen(self):
        """The rank count the connected communicator itself reports (ncclCommCount); -1 if it cannot say."""
        return int(lib().lsnShardRanksSeen(self._h))

    def plan(self, whole=True):
        """A non-owning FusionPlan view of one of the handle's plans (profile / kernel_stats / check only)."""
        v = object.__new__(FusionPlan)
        v._h = lib().lsnShardPlan(self._h, 1 if whole else 0)
        v.close = lambda: None
        return v

    def close(self):
        if self._h:
            lib().lsnShardDestroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


NN_BRUTE, NN_GRID = 0, 1


class TickPipeline:
    """lsnTick*: the chained tick (radial correction out of place -> vertices -> triangulation) of n_ticks x n_maps frames in HBM as one call."""

    def __init__(self, device, n_ticks, widths, heights):
        require_gpu()
        self.widths, self.heights = _as(widths, np.int32), _as(heights, np.int32)
        self.n_ticks, self.n_maps = int(n_ticks), len(self.widths)
        self._h = lib().lsnTickCreate(int(device), self.n_ticks, self.n_maps, _ptr(self.widths), _ptr(self.heights))
        if not self._h:
            raise NativeUtilsError(f"lsnTickCreate failed: {last_error()}")
        self.capacity = int(lib().lsnTickCapacity(self._h))
        self.tri_capacity = int(lib().lsnTickTriangleCapacity(self._h))
        self.parts = int(lib().lsnTickParts(self._h))

    def set_params(self, intr, wt, bounds, stream=0):
        intr, wt, bounds = _as(intr, np.float32).ravel(), _as(wt, np.float32).ravel(), _as(bounds, np.float32).ravel()
        assert intr.size == 7 * self.n_maps and wt.size == 12 * self.n_maps and bounds.size == 6
        _check(lib().lsnTickSetParams(self._h, _ptr(intr), _ptr(wt), _ptr(bounds), stream), "lsnTickSetParams")

    def run(self, d_depth_in, d_colors_in, d_depth_corr, d_colors_corr, d_vertices, d_offsets, d_triangles, d_tri_offsets, stream=0):
        _check(lib().lsnTickRun(self._h, d_depth_in, d_colors_in, d_depth_corr, d_colors_corr, d_vertices, d_offsets, d_triangles, d_tri_offsets, stream),
               "lsnTickRun")

    def close(self):
        if self._h:
            lib().lsnTickDestroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class IcpWorkspace:
    """lsnIcp*: device-resident ICP for clouds up to (max_n1, max_n2)."""

    def __init__(self, device, max_n1, max_n2):
        require_gpu()
        self.device = int(device)
        self._h = lib().lsnIcpCreate(self.device, int(max_n1), int(max_n2))
        if not self._h:
            raise NativeUtilsError(f"lsnIcpCreate failed: {last_error()}")

    def run(self, d_verts1, n1, d_verts2, n2, d_R, d_t, max_iter=10, nn_mode=NN_GRID, stream=0):
        _check(lib().lsnIcpRun(self._h, d_verts1, int(n1), d_verts2, int(n2), d_R, d_t, int(max_iter), int(nn_mode), stream),
               "lsnIcpRun")

    def nearest(self, d_verts1, n1, d_verts2, n2, d_idx, d_dist2, nn_mode=NN_GRID, stream=0):
        _check(lib().lsnIcpNearest(self._h, d_verts1, int(n1), d_verts2, int(n2), d_idx, d_dist2, int(nn_mode), stream),
               "lsnIcpNearest")

    def set_profiling(self, on=True):
        _check(lib().lsnIcpSetProfiling(self._h, 1 if on else 0), "lsnIcpSetProfiling")

    def profile(self, stream=0):
        """Milliseconds of the last profiled run(): {build, nn, match_reduce_solve, final_apply} (synchronises the stream)."""
        ms = np.zeros(4, dtype=np.float32)
        if lib().lsnIcpProfile(self._h, _ptr(ms), stream) < 0:
            raise NativeUtilsError(f"lsnIcpProfile failed: {last_error()}")
        return {"build": float(ms[0]), "nn": float(ms[1]), "match_reduce_solve": float(ms[2]), "final_apply": float(ms[3])}

    def near_resolved(self, stream=0):
        """Queries of the last voxel-grid NN step that the near path settled (diagnostic; synchronises the stream)."""
        n = lib().lsnIcpNearResolved(self._h, stream)
        if n < 0:
            raise NativeUtilsError(f"lsnIcpNearResolved failed: {last_error()}")
        return int(n)

    def trace(self, max_iters, stream=0):
        out = np.zeros((max(max_iters, 1), 16), dtype=np.float32)
        n = lib().lsnIcpTrace(self._h, _ptr(out), int(max_iters), stream)
        if n < 0:
            raise NativeUtilsError(f"lsnIcpTrace failed: {last_error()}")
        return out[:n]

    def close(self):
        if self._h:
            lib().lsnIcpDestroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ----------------------------------------------------------------------------------------------------------
# Part 3: wire / disk formats either side of the path
# ----------------------------------------------------------------------------------------------------------

class TransferPacker:
    """LsnTransfer: builds the TransferSocket.SendFrame byte stream (TransferSocket.cs:50-104; chunks as
    TransferServer.cs:177-270) on the device."""

    def __init__(self, device, max_vertices, max_triangles):
        require_gpu()
        self.h = lib().lsnTransferCreate(int(device), int(max_vertices), int(max_triangles))
        if not self.h:
            raise NativeUtilsError(f"lsnTransferCreate failed: {last_error()}")

    def pack(self, d_vertices, n_vertices, d_triangles, n_triangles, d_out, out_cap, stream=0):
        n = lib().lsnTransferPack(self.h, d_vertices, int(n_vertices), d_triangles or None, int(n_triangles), d_out, int(out_cap), stream or None)
        if n < 0:
            raise NativeUtilsError(f"lsnTransferPack failed: {last_error()}")
        return int(n)

    def last_path(self):
        """0 vertices only, 1 all chunks from one prefix sum, 2 chunk after chunk (lsnTransferLastPath)."""
        return int(lib().lsnTransferLastPath(self.h))

    def close(self):
        if getattr(self, "h", None):
            lib().lsnTransferDestroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def transfer_frame_bound(n_vertices, n_triangles):
    return int(lib().lsnTransferFrameBound(int(n_vertices), int(n_triangles)))


def ply_binary_bytes(n_vertices, n_triangles):
    return int(lib().lsnPlyBinaryBytes(int(n_vertices), int(n_triangles)))


def ply_pack(device, d_vertices, n_vertices, d_triangles, n_triangles, d_out, out_cap, stream=0):
    require_gpu()
    n = lib().lsnPlyPack(int(device), d_vertices, int(n_vertices), d_triangles or None, int(n_triangles), d_out, int(out_cap), stream or None)
    if n < 0:
        raise NativeUtilsError(f"lsnPlyPack failed: {last_error()}")
    return int(n)


def _last_mesh(fn, what):
    require_gpu()
    cap = fn(None, 0)
    if cap < 0:
        raise NativeUtilsError(f"{what} failed: {last_error()}")
    out = np.zeros(cap, dtype=np.uint8)
    n = fn(_ptr(out), cap)
    if n < 0:
        raise NativeUtilsError(f"{what} failed: {last_error()}")
    return out[:n].tobytes()


def last_mesh_transfer_frame():
    """SendFrame stream (TransferSocket.cs:50-104) of the mesh the last merge call returned, built in HBM."""
    return _last_mesh(lib().lsnLastMeshTransferFrame, "lsnLastMeshTransferFrame")


def last_mesh_ply():
    """Binary PLY file image (Utils.cs:222-262) of the mesh the last merge call returned, built in HBM."""
    return _last_mesh(lib().lsnLastMeshPly, "lsnLastMeshPly")


def zstd_available():
    return bool(lib().lsnZstdAvailable())


def frame_parse_header(header16):
    """Returns FrameInfo, or None for the "no more frames" header (payload_bytes <= 0)."""
    buf = np.frombuffer(bytes(header16[:16]), dtype=np.uint8)
    if buf.size != 16:
        raise NativeUtilsError("a frame header is 16 bytes")
    info = FrameInfo()
    rc = lib().lsnFrameParseHeader(_ptr(buf), C.byref(info))
    if rc < 0:
        raise NativeUtilsError(f"lsnFrameParseHeader failed: {last_error()}")
    return None if rc == 1 else info


def frame_decode(message):
    """One whole frame message (header + payload) -> (depth u16 [h,w], rgb u8 [h,w,3], bodies bytes, n_bodies)."""
    msg = np.frombuffer(bytes(message), dtype=np.uint8)
    info = frame_parse_header(msg[:16])
    if info is None:
        return None
    if 16 + info.payload_bytes > msg.size:
        raise NativeUtilsError("frame message shorter than its header says")
    w, h = info.width, info.height
    depth = np.zeros((h, w), dtype=np.uint16)
    rgb = np.zeros((h, w, 3), dtype=np.uint8)
    bodies = np.zeros(1 << 16, dtype=np.uint8)
    nb = C.c_int(0)
    payload = np.ascontiguousarray(msg[16:16 + info.payload_bytes])
    bl = lib().lsnFrameDecode(_ptr(payload), info.payload_bytes, info.compressed, w, h, _ptr(depth), _ptr(rgb), _ptr(bodies), bodies.size, C.byref(nb))
    if bl < 0:
        raise NativeUtilsError(f"lsnFrameDecode failed: {last_error()}")
    return depth, rgb, bodies[:bl].tobytes(), nb.value


def frame_encode(depth, rgb, bodies=None, compression_level=0):
    depth = _as(depth, np.uint16)
    h, w = depth.shape
    rgb = _as(rgb, np.uint8).reshape(h, w, 3)
    b = np.frombuffer(bodies, dtype=np.uint8) if bodies else None
    out = np.zeros(16 + 5 * w * h + (b.size if b is not None else 4) + 1024 + (w * h * 5) // 64, dtype=np.uint8)
    n = lib().lsnFrameEncode(_ptr(depth), _ptr(rgb), w, h, _ptr(b) if b is not None else None, b.size if b is not None else 0,
                             int(compression_level), _ptr(out), out.size)
    if n < 0:
        raise NativeUtilsError(f"lsnFrameEncode failed: {last_error()}")
    return out[:n].tobytes()


def recording_frames(file_bytes):
    """Iterates (timestamp_ms, frame message bytes) over the memory image of a client recording file."""
    buf = np.frombuffer(file_bytes, dtype=np.uint8)
    pos = 0
    off, ln, ts = C.c_longlong(0), C.c_int(0), C.c_int(0)
    while True:
        nxt = lib().lsnRecordingNext(_ptr(buf), buf.size, pos, C.byref(off), C.byref(ln), C.byref(ts))
        if nxt < 0:
            err = last_error()
            if err:
                raise NativeUtilsError(err)
            return
        yield ts.value, buf[off.value:off.value + ln.value].tobytes()
        pos = nxt


def recording_append(frame, timestamp_ms):
    f = np.frombuffer(bytes(frame), dtype=np.uint8)
    out = np.zeros(f.size + 96, dtype=np.uint8)
    n = lib().lsnRecordingAppend(_ptr(out), out.size, _ptr(f) if f.size else None, f.size, int(timestamp_ms))
    if n < 0:
        raise NativeUtilsError(f"lsnRecordingAppend failed: {last_error()}")
    return out[:n].tobytes()
