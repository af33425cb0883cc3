"""The reference's capture/replay file of one merge call (storeAllFramesInformation / loadAllFramesInformation,
src/NativeUtils/depthprocessing.cpp:1316-1385) and the golden-mesh file of its regression main()
(src/NativeUtils/main.cpp:179-199), so fixtures written here can be replayed through the real NativeUtils unmodified.

frames file : int32 n; int32 w[n]; int32 h[n]; n x {u16 depth[w*h]; u8 rgb[w*h*3]}; f32 intr[7n]; f32 wt[12n]
mesh file   : int32 nTriangles; int32 triangles[3*nTriangles]; int32 nVertices; VertexC4ubV3f vertices[nVertices]
              (the reference writes only nTriangles ints of the index array -- main.cpp:182, a slip; all 3*n are kept here)
The crop bounds and the two flags are call arguments in the reference, not part of the frames file.
"""
import numpy as np

from . import synth
from .native import VERTEX_DTYPE


def save_frames(path, rig):
    with open(path, "wb") as f:
        np.array([rig.n], dtype="<i4").tofile(f)
        if rig.n > 0:
            rig.widths.astype("<i4").tofile(f)
            rig.heights.astype("<i4").tofile(f)
        pd = pc = 0
        for i in range(rig.n):
            npx = int(rig.widths[i]) * int(rig.heights[i])
            rig.depth_maps[pd:pd + 2 * npx].tofile(f)
            rig.depth_colors[pc:pc + 3 * npx].tofile(f)
            pd += 2 * npx
            pc += 3 * npx
        rig.intr.astype("<f4").tofile(f)
        rig.wt.astype("<f4").tofile(f)


def load_frames(path, bounds=synth.DEFAULT_BOUNDS):
    raw = np.fromfile(path, dtype=np.uint8)
    n = int(raw[:4].view("<i4")[0])
    pos = 4
    w = raw[pos:pos + 4 * n].view("<i4").copy(); pos += 4 * n
    h = raw[pos:pos + 4 * n].view("<i4").copy(); pos += 4 * n
    depths, rgbs = [], []
    for i in range(n):
        npx = int(w[i]) * int(h[i])
        depths.append(raw[pos:pos + 2 * npx].view("<u2").reshape(int(h[i]), int(w[i])).copy()); pos += 2 * npx
        rgbs.append(raw[pos:pos + 3 * npx].reshape(int(h[i]), int(w[i]), 3).copy()); pos += 3 * npx
    intr = raw[pos:pos + 28 * n].view("<f4").copy(); pos += 28 * n
    wt = raw[pos:pos + 48 * n].view("<f4").copy(); pos += 48 * n
    assert pos == raw.size, "trailing bytes in frames file"
    return synth.Rig(depths, rgbs, intr, wt, bounds)


def save_mesh(path, vertices, triangles):
    with open(path, "wb") as f:
        tri = np.ascontiguousarray(triangles, dtype="<i4").reshape(-1, 3)
        np.array([len(tri)], dtype="<i4").tofile(f)
        tri.tofile(f)
        np.array([len(vertices)], dtype="<i4").tofile(f)
        np.ascontiguousarray(vertices, dtype=VERTEX_DTYPE).tofile(f)


def load_mesh(path):
    raw = np.fromfile(path, dtype=np.uint8)
    nt = int(raw[:4].view("<i4")[0])
    tri = raw[4:4 + 12 * nt].view("<i4").reshape(-1, 3).copy()
    pos = 4 + 12 * nt
    nv = int(raw[pos:pos + 4].view("<i4")[0])
    verts = raw[pos + 4:pos + 4 + 16 * nv].view(VERTEX_DTYPE).copy()
    assert pos + 4 + 16 * nv == raw.size
    return verts, tri
