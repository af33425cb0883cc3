"""ctypes binding of the CPU oracle (oracle/liblsn_oracle.so) and of the compiled reference NN (oracle/_ref).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never from livescan3d_amd/.  Array conventions follow the reference C-ABI
(include/NativeUtils/depthprocessing.h:103-112, include/NativeUtils/icp.h:65).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liblsn_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libref_nn.so")
_REF_BIN = os.path.join(_HERE, "_ref", "ref_nn")

VERTEX_DTYPE = np.dtype([("R", "u1"), ("G", "u1"), ("B", "u1"), ("A", "u1"),
                         ("X", "<f4"), ("Y", "<f4"), ("Z", "<f4")])
assert VERTEX_DTYPE.itemsize == 16

ICP_ITER_DTYPE = np.dtype([("n_matched", "<i4"), ("n_kept", "<i4"), ("mean", "<f4"), ("stddev", "<f4"),
                           ("T", "<f4", (3,)), ("Rn", "<f4", (9,))])


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(_LIB) or \
            os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "lsn_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)
    elif os.path.exists("/root/reference/include/nanoflann.h") and not os.path.exists(_REF_SO):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        p = C.c_void_p
        L.orc_create_vertices.restype = C.c_int
        L.orc_create_vertices.argtypes = [p, p, C.c_int, C.c_int, p, p, p, p, p, p]
        L.orc_generate_mesh_vertices.restype = C.c_long
        L.orc_generate_mesh_vertices.argtypes = [C.c_int, p, p, p, p, p, p, p, p, p, C.c_int]
        L.orc_generate_vertices_from_depth_map.restype = C.c_int
        L.orc_generate_vertices_from_depth_map.argtypes = [p, p, p, p, p, p, p, C.c_int, p]
        L.orc_radial_correction_all.restype = None
        L.orc_radial_correction_all.argtypes = [C.c_int, p, p, p, p, p, C.c_int]
        L.orc_generate_triangles.restype = C.c_long
        L.orc_generate_triangles.argtypes = [p, p, C.c_int, C.c_int, C.c_int, p]
        L.orc_generate_mesh.restype = C.c_long
        L.orc_generate_mesh.argtypes = [C.c_int, p, p, p, p, p, p, p, p, p, p, p]
        for f in (L.orc_nn_brute, L.orc_nn_kdtree):
            f.restype = None
            f.argtypes = [p, C.c_int, p, C.c_int, p, p, C.c_int]
        L.orc_icp.restype = C.c_float
        L.orc_icp.argtypes = [p, p, C.c_int, C.c_int, p, p, C.c_int, C.c_int, C.c_int, p]
        L.orc_refine.restype = None
        L.orc_refine.argtypes = [C.c_int, p, p, C.c_int, C.c_int, p, p, p, p, C.c_int, C.c_int]
        L.orc_kabsch_rotation.restype = None
        L.orc_kabsch_rotation.argtypes = [p, p]
        L.orc_form_chunks.restype = C.c_int
        L.orc_form_chunks.argtypes = [p, C.c_int, p, C.c_int, p, p, p, p, p]
        L.orc_transfer_frame.restype = C.c_long
        L.orc_transfer_frame.argtypes = [p, C.c_int, p, C.c_int, p, C.c_long]
        L.orc_ply_binary.restype = C.c_long
        L.orc_ply_binary.argtypes = [p, C.c_int, p, C.c_int, p, C.c_long]
        L.orc_frame_encode.restype = C.c_long
        L.orc_frame_encode.argtypes = [p, p, C.c_int, C.c_int, p, C.c_int, p, C.c_long]
        L.orc_frame_decode.restype = C.c_int
        L.orc_frame_decode.argtypes = [p, C.c_long, p, p, p, p, p, p, p]
        L.orc_recording_append.restype = C.c_long
        L.orc_recording_append.argtypes = [p, C.c_long, p, C.c_int, C.c_int]
        L.orc_recording_next.restype = C.c_long
        L.orc_recording_next.argtypes = [p, C.c_long, C.c_long, p, p, p]
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if n is not None:
        assert a.size == n, (a.size, n)
    return a


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def create_vertices(depth, rgb, intr7, wt12, bounds6, want_maps=False):
    """createVertices + formMesh repack for one sensor.  depth: (h,w) u16, rgb: (h,w,3) u8."""
    depth = np.ascontiguousarray(depth, dtype=np.uint16)
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = depth.shape
    out = np.zeros(h * w, dtype=VERTEX_DTYPE)
    v2p = np.zeros(h * w, dtype=np.int32) if want_maps else None
    p2v = np.zeros(h * w, dtype=np.int32) if want_maps else None
    intr7, wt12, bounds6 = _f32(intr7, 7), _f32(wt12, 12), _f32(bounds6, 6)
    n = lib().orc_create_vertices(_ptr(depth), _ptr(rgb), w, h, _ptr(intr7), _ptr(wt12), _ptr(bounds6),
                                  _ptr(out), _ptr(v2p), _ptr(p2v))
    if want_maps:
        return out[:n].copy(), v2p[:n].copy(), p2v
    return out[:n].copy()


def generate_mesh_vertices(depth_maps, depth_colors, widths, heights, intr, wt, bounds6, n_threads=1):
    """generateMeshFromDepthMaps (flags false,false), vertices only.  Returns (vertices, per_map_counts)."""
    widths, heights = _i32(widths), _i32(heights)
    n_maps = len(widths)
    depth_maps = np.ascontiguousarray(depth_maps).view(np.uint8).ravel()
    depth_colors = np.ascontiguousarray(depth_colors, dtype=np.uint8).ravel()
    total = int(np.sum(widths.astype(np.int64) * heights))
    assert depth_maps.size == total * 2 and depth_colors.size == total * 3
    out = np.zeros(max(total, 1), dtype=VERTEX_DTYPE)
    counts = np.zeros(max(n_maps, 1), dtype=np.int32)
    intr, wt, bounds6 = _f32(intr, 7 * n_maps), _f32(wt, 12 * n_maps), _f32(bounds6, 6)
    n = lib().orc_generate_mesh_vertices(n_maps, _ptr(depth_maps), _ptr(depth_colors), _ptr(widths), _ptr(heights),
                                         _ptr(intr), _ptr(wt), _ptr(bounds6), _ptr(out), _ptr(counts), n_threads)
    return out[:n].copy(), counts[:n_maps].copy()


def radial_correction(depth_maps, depth_colors, widths, heights, intr, n_threads=1):
    """depthMapAndColorSetRadialCorrection on copies of the concatenated buffers.  Returns (depth_maps u8 view, colours)."""
    widths, heights = _i32(widths), _i32(heights)
    n_maps = len(widths)
    dm = np.ascontiguousarray(depth_maps).view(np.uint8).ravel().copy()
    dc = np.ascontiguousarray(depth_colors, dtype=np.uint8).ravel().copy()
    intr = _f32(intr, 7 * n_maps)
    lib().orc_radial_correction_all(n_maps, _ptr(dm), _ptr(dc), _ptr(widths), _ptr(heights), _ptr(intr), n_threads)
    return dm, dc


def generate_triangles(depth, pix_to_vert, index_base=0):
    """generateTrianglesGradients for one sensor.  depth (h,w) u16, pix_to_vert (h*w) int32.  Returns int32 [n,3]."""
    depth = np.ascontiguousarray(depth, dtype=np.uint16)
    h, w = depth.shape
    p2v = np.ascontiguousarray(pix_to_vert, dtype=np.int32).ravel()
    assert p2v.size == h * w
    out = np.zeros((2 * h * w, 3), dtype=np.int32)
    n = lib().orc_generate_triangles(_ptr(depth), _ptr(p2v), w, h, int(index_base), _ptr(out))
    return out[:n].copy()


def generate_mesh(depth_maps, depth_colors, widths, heights, intr, wt, bounds6):
    """generateMeshFromDepthMaps (flags false,false) with the always-on triangulation.
    Returns (vertices, per_map_counts, triangles int32 [n,3])."""
    widths, heights = _i32(widths), _i32(heights)
    n_maps = len(widths)
    depth_maps = np.ascontiguousarray(depth_maps).view(np.uint8).ravel()
    depth_colors = np.ascontiguousarray(depth_colors, dtype=np.uint8).ravel()
    total = int(np.sum(widths.astype(np.int64) * heights))
    out = np.zeros(max(total, 1), dtype=VERTEX_DTYPE)
    counts = np.zeros(max(n_maps, 1), dtype=np.int32)
    tri = np.zeros((max(2 * total, 1), 3), dtype=np.int32)
    nt = C.c_long(0)
    intr, wt, bounds6 = _f32(intr, 7 * n_maps), _f32(wt, 12 * n_maps), _f32(bounds6, 6)
    n = lib().orc_generate_mesh(n_maps, _ptr(depth_maps), _ptr(depth_colors), _ptr(widths), _ptr(heights), _ptr(intr), _ptr(wt),
                                _ptr(bounds6), _ptr(out), _ptr(counts), _ptr(tri), C.byref(nt))
    return out[:n].copy(), counts[:n_maps].copy(), tri[:nt.value].copy()


def generate_vertices_from_depth_map(depth_maps, depth_colors, widths, heights, intr, wt, bounds6, index):
    widths, heights = _i32(widths), _i32(heights)
    n_maps = len(widths)
    depth_maps = np.ascontiguousarray(depth_maps).view(np.uint8).ravel()
    depth_colors = np.ascontiguousarray(depth_colors, dtype=np.uint8).ravel()
    out = np.zeros(int(widths[index]) * int(heights[index]), dtype=VERTEX_DTYPE)
    intr, wt, bounds6 = _f32(intr, 7 * n_maps), _f32(wt, 12 * n_maps), _f32(bounds6, 6)
    n = lib().orc_generate_vertices_from_depth_map(_ptr(depth_maps), _ptr(depth_colors), _ptr(widths), _ptr(heights),
                                                   _ptr(intr), _ptr(wt), _ptr(bounds6), index, _ptr(out))
    return out[:n].copy()


def nn(targets, queries, mode="kdtree", n_threads=1):
    """Exact 1-NN (lowest index on ties).  Returns (idx int64[n2], dist2 f32[n2])."""
    t, q = _f32(targets).reshape(-1, 3), _f32(queries).reshape(-1, 3)
    idx = np.zeros(len(q), dtype=np.int64)
    dist = np.zeros(len(q), dtype=np.float32)
    f = lib().orc_nn_brute if mode == "brute" else lib().orc_nn_kdtree
    f(_ptr(t), len(t), _ptr(q), len(q), _ptr(idx), _ptr(dist), n_threads)
    return idx, dist


def icp(verts1, verts2, R=None, t=None, max_iter=10, nn_mode="kdtree", n_threads=1, trace=False):
    """ICP (icp.cpp:75-177).  Returns (verts2_out, R_out, t_out[, trace])."""
    v1 = _f32(verts1).reshape(-1, 3)
    v2 = _f32(verts2).reshape(-1, 3).copy()
    R = np.eye(3, dtype=np.float32).ravel() if R is None else _f32(R, 9).copy().ravel()
    t = np.zeros(3, dtype=np.float32) if t is None else _f32(t, 3).copy().ravel()
    tr = np.zeros(max(max_iter, 1), dtype=ICP_ITER_DTYPE) if trace else None
    lib().orc_icp(_ptr(v1), _ptr(v2), len(v1), len(v2), _ptr(R), _ptr(t), max_iter,
                  0 if nn_mode == "brute" else 1, n_threads, _ptr(tr))
    if trace:
        return v2, R.reshape(3, 3), t, tr[:max_iter]
    return v2, R.reshape(3, 3), t


def refine(clouds, world_R, world_t, n_refine_iters=2, n_icp_iters=10, nn_mode="kdtree", n_threads=1):
    """refineWorker_DoWork (MainWindowForm.cs:330-410).  Returns (clouds_out, world_R, world_t, Rs, Ts)."""
    cl = [_f32(c).reshape(-1, 3).copy() for c in clouds]
    n = np.array([len(c) for c in cl], dtype=np.int32)
    ptrs = (C.c_void_p * len(cl))(*[c.ctypes.data for c in cl])
    wR = _f32(world_R, 9 * len(cl)).copy().reshape(-1)
    wt = _f32(world_t, 3 * len(cl)).copy().reshape(-1)
    Rs = np.zeros(9 * len(cl), dtype=np.float32)
    Ts = np.zeros(3 * len(cl), dtype=np.float32)
    lib().orc_refine(len(cl), C.cast(ptrs, C.c_void_p), _ptr(n), n_refine_iters, n_icp_iters,
                     _ptr(wR), _ptr(wt), _ptr(Rs), _ptr(Ts), 0 if nn_mode == "brute" else 1, n_threads)
    return cl, wR.reshape(-1, 3, 3), wt.reshape(-1, 3), Rs.reshape(-1, 3, 3), Ts.reshape(-1, 3)


def kabsch_rotation(M):
    M = _f32(M, 9).ravel()
    Rn = np.zeros(9, dtype=np.float32)
    lib().orc_kabsch_rotation(_ptr(M), _ptr(Rn))
    return Rn.reshape(3, 3)


# ---- compiled reference NN (oracle/_ref) -----------------------------------------------------------------

def have_ref_nn():
    return os.path.exists(_REF_SO)


_ref = None


def ref_nn(targets, queries):
    """The reference's own NN step: nanoflann 1.1.9 + PointCloud adaptor (icp.cpp:18-32) compiled from /root/reference."""
    global _ref
    if _ref is None:
        _ref = C.CDLL(_REF_SO)
        _ref.ref_nn_query.restype = None
        _ref.ref_nn_query.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    t, q = _f32(targets).reshape(-1, 3), _f32(queries).reshape(-1, 3)
    idx = np.zeros(len(q), dtype=np.int64)
    dist = np.zeros(len(q), dtype=np.float32)
    _ref.ref_nn_query(_ptr(t), len(t), _ptr(q), len(q), _ptr(idx), _ptr(dist))
    return idx, dist


# ---- compiled reference triangulation (oracle/_ref) --------------------------------------------------------

_REF_TRI_SO = os.path.join(_HERE, "_ref", "libref_tri.so")
_ref_tri = None


def have_ref_tri():
    return os.path.exists(_REF_TRI_SO)


def ref_triangles(depth, pix_to_vert):
    """The reference's own MeshGenerator::generateTrianglesGradients (src/NativeUtils/meshGenerator.cpp compiled from
    /root/reference).  depth (h,w) u16 = the sensor's depth map, pix_to_vert (h*w) int32.  Returns int32 [n,3]."""
    global _ref_tri
    if _ref_tri is None:
        _ref_tri = C.CDLL(_REF_TRI_SO)
        _ref_tri.ref_generate_triangles.restype = C.c_long
        _ref_tri.ref_generate_triangles.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    depth = np.ascontiguousarray(depth, dtype=np.uint16)
    h, w = depth.shape
    p2v = np.ascontiguousarray(pix_to_vert, dtype=np.int32).ravel()
    assert p2v.size == h * w
    out = np.zeros((2 * h * w, 3), dtype=np.int32)
    n = _ref_tri.ref_generate_triangles(_ptr(depth), _ptr(p2v), w, h, _ptr(out))
    return out[:n].copy()


# ---- wire / disk formats ---------------------------------------------------------------------

CHUNK_LIMIT = 65000 - 3


def _verts(v):
    v = np.ascontiguousarray(v)
    assert v.dtype == VERTEX_DTYPE
    return v


def form_chunks(vertices, triangles):
    """TransferServer.formMeshChunks / formVerticesChunks.  Returns (new_vertices, new_triangles [n,3], v_chunks, t_chunks)."""
    v = _verts(vertices)
    tri = np.ascontiguousarray(triangles, dtype=np.int32).reshape(-1, 3)
    nv, nt = len(v), len(tri)
    new_v = np.zeros(max(nv, 3 * nt, 1), dtype=VERTEX_DTYPE)
    new_tri = np.zeros((max(nt, 1), 3), dtype=np.int32)
    cap = 3 * nt // CHUNK_LIMIT + nv // CHUNK_LIMIT + 2
    vc, tc = np.zeros(cap, dtype=np.int32), np.zeros(cap, dtype=np.int32)
    n_new = C.c_int(0)
    nc = lib().orc_form_chunks(_ptr(v), nv, _ptr(tri), nt, _ptr(new_v), _ptr(new_tri), _ptr(vc), _ptr(tc), C.byref(n_new))
    return new_v[:n_new.value].copy(), new_tri[:nt].copy(), vc[:nc].copy(), tc[:nc].copy()


def transfer_frame(vertices, triangles):
    """The bytes of TransferSocket.SendFrame for this cloud / mesh."""
    v = _verts(vertices)
    tri = np.ascontiguousarray(triangles, dtype=np.int32).reshape(-1, 3)
    cap = 12 + 8 * (3 * len(tri) // CHUNK_LIMIT + len(v) // CHUNK_LIMIT + 2) + 15 * max(len(v), 3 * len(tri)) + 12 * len(tri)
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().orc_transfer_frame(_ptr(v), len(v), _ptr(tri), len(tri), _ptr(out), cap)
    assert n >= 0, n
    return out[:n].tobytes()


def ply_binary(vertices, triangles):
    v = _verts(vertices)
    tri = np.ascontiguousarray(triangles, dtype=np.int32).reshape(-1, 3)
    cap = 512 + 15 * len(v) + 13 * len(tri)
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().orc_ply_binary(_ptr(v), len(v), _ptr(tri), len(tri), _ptr(out), cap)
    assert n >= 0, n
    return out[:n].tobytes()


def frame_encode(depth, rgb, bodies=None):
    depth = np.ascontiguousarray(depth, dtype=np.uint16)
    h, w = depth.shape
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    b = np.frombuffer(bodies, dtype=np.uint8) if bodies else None
    out = np.zeros(16 + 5 * w * h + (b.size if b is not None else 4), dtype=np.uint8)
    n = lib().orc_frame_encode(_ptr(depth), _ptr(rgb), w, h, _ptr(b), 0 if b is None else b.size, _ptr(out), out.size)
    assert n == out.size, (n, out.size)
    return out.tobytes()


def frame_decode(message):
    msg = np.frombuffer(bytes(message), dtype=np.uint8)
    w, h, bb, nb = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    d, c, b = C.c_void_p(0), C.c_void_p(0), C.c_void_p(0)
    rc = lib().orc_frame_decode(_ptr(msg), msg.size, C.byref(w), C.byref(h), C.byref(d), C.byref(c), C.byref(b), C.byref(bb), C.byref(nb))
    if rc != 0:
        return None
    base = msg.ctypes.data
    P = w.value * h.value
    depth = msg[d.value - base:d.value - base + 2 * P].view(np.uint16).reshape(h.value, w.value).copy()
    rgb = msg[c.value - base:c.value - base + 3 * P].reshape(h.value, w.value, 3).copy()
    return depth, rgb, msg[b.value - base:b.value - base + bb.value].tobytes(), nb.value


def recording_append(frame, timestamp_ms):
    f = np.frombuffer(bytes(frame), dtype=np.uint8)
    out = np.zeros(f.size + 96, dtype=np.uint8)
    n = lib().orc_recording_append(_ptr(out), out.size, _ptr(f) if f.size else None, f.size, int(timestamp_ms))
    assert n > 0
    return out[:n].tobytes()


def recording_frames(file_bytes):
    buf = np.frombuffer(file_bytes, dtype=np.uint8)
    pos, out = 0, []
    off, ln, ts = C.c_long(0), C.c_int(0), C.c_int(0)
    while True:
        nxt = lib().orc_recording_next(_ptr(buf), buf.size, pos, C.byref(off), C.byref(ln), C.byref(ts))
        if nxt < 0:
            return out
        out.append((ts.value, buf[off.value:off.value + ln.value].tobytes()))
        pos = nxt
