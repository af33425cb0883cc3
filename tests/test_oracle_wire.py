"""The oracle's wire / disk format restatements against independent pure-Python restatements of the
C# / C++ they follow: formMeshChunks + formVerticesChunks (LiveScanServer/TransferServer.cs:177-270), SendFrame
(TransferSocket.cs:50-104), saveToPly binary (Utils.cs:222-262), the frame message (liveScanClient.cpp:185-290 /
KinectSocket.cs:211-304) and the recording file (frameFileWriterReader.cpp:59-82,115-130).
PARITY UNPINNED (no C# toolchain, no sample files in the reference): two independent restatements must agree bit for bit.
The host-side parsers of the product (libNativeUtils part 3; no device needed) are checked against the same data."""
import struct

import numpy as np
import pytest

from livescan3d_amd import native


def py_form_chunks(verts, tri_flat, limit):
    """TransferServer.cs:203-270 transcribed to Python lists."""
    n_v, n_idx = len(verts), len(tri_flat)
    chunk_index = [-1] * n_v
    vmap = [0] * n_v
    new_v, new_t = [], [0] * n_idx
    vs, ts = [], []
    start = cur = in_chunk = 0
    for t in range(n_idx):
        val = int(tri_flat[t])
        if chunk_index[val] != cur:
            new_v.append(val)
            vmap[val] = in_chunk
            chunk_index[val] = cur
            new_t[t] = in_chunk
            in_chunk += 1
        else:
            new_t[t] = vmap[val]
        if in_chunk >= limit and (t + 1) % 3 == 0:
            cur += 1
            vs.append(in_chunk)
            ts.append((t - start) // 3)
            in_chunk = 0
            start = t
    if in_chunk != 0:
        vs.append(in_chunk)
        ts.append((n_idx - start) // 3)
    return new_v, new_t, vs, ts


def grid_mesh(rng, w, h, drop=0.03):
    """A depth-map-like mesh: vertex per pixel (some missing), two triangles per quad where all corners exist."""
    present = rng.random((h, w)) >= drop
    ids = np.where(present.ravel(), np.cumsum(present.ravel()) - 1, -1).reshape(h, w)
    tris = []
    for y in range(1, h):
        for x in range(w - 1):
            p, u, ur, r = ids[y, x], ids[y - 1, x], ids[y - 1, x + 1], ids[y, x + 1]
            if p >= 0 and u >= 0 and r >= 0:
                tris.append((r, u, p))
            if r >= 0 and ur >= 0 and u >= 0:
                tris.append((r, ur, u))
    n = int(present.sum())
    v = np.zeros(n, dtype=native.VERTEX_DTYPE)
    v["R"], v["G"], v["B"] = rng.integers(0, 256, n), rng.integers(0, 256, n), rng.integers(0, 256, n)
    v["A"] = 255
    v["X"], v["Y"], v["Z"] = rng.normal(size=n), rng.normal(size=n), rng.normal(size=n)
    return v, np.array(tris, dtype=np.int32).reshape(-1, 3)


@pytest.mark.parametrize("w,h", [(40, 30), (400, 300), (7, 3)])
def test_form_mesh_chunks_equals_python_transcription(orc, w, h):
    rng = np.random.default_rng(w + h)
    v, tri = grid_mesh(rng, w, h)
    new_v, new_t, vc, tc = orc.form_chunks(v, tri)
    pv, pt, pvs, pts = py_form_chunks(v, tri.ravel(), orc.CHUNK_LIMIT)
    assert list(vc) == pvs and list(tc) == pts
    assert np.array_equal(new_t.ravel(), np.array(pt, dtype=np.int32))
    assert new_v.tobytes() == v[np.array(pv, dtype=np.int64)].tobytes()
    if w == 400:
        assert len(vc) == 2 and vc[0] >= orc.CHUNK_LIMIT and vc[0] <= orc.CHUNK_LIMIT + 2
        assert tc.sum() == len(tri) - 1                        # the reference's off-by-one at the first chunk boundary
        # every chunk is self-contained: local indices stay below the chunk's vertex count
        edges = np.concatenate([[0], np.cumsum(tc)])
        edges[1:] += 1                                          # the triangle the first count misses belongs to chunk 0
        for c in range(len(vc)):
            assert new_t[edges[c]:edges[c + 1] if c + 1 < len(vc) else len(new_t)].max() < vc[c]
    else:
        assert len(vc) == 1 and tc[0] == len(tri) and vc[0] == len(new_v)
        # single chunk: the re-indexed mesh is the same surface
        assert new_v[new_t.ravel()].tobytes() == v[tri.ravel()].tobytes()


def test_form_vertices_chunks(orc):
    rng = np.random.default_rng(5)
    n = 2 * orc.CHUNK_LIMIT + 17
    v = np.zeros(n, dtype=native.VERTEX_DTYPE)
    v["X"] = rng.normal(size=n)
    new_v, new_t, vc, tc = orc.form_chunks(v, np.zeros((0, 3), np.int32))
    assert list(vc) == [orc.CHUNK_LIMIT, orc.CHUNK_LIMIT, 17] and list(tc) == [0, 0, 0]
    assert new_v.tobytes() == v.tobytes() and len(new_t) == 0


def test_transfer_frame_layout(orc):
    rng = np.random.default_rng(11)
    v, tri = grid_mesh(rng, 50, 40)
    for t in (tri, np.zeros((0, 3), np.int32)):
        blob = orc.transfer_frame(v, t)
        new_v, new_t, vc, tc = orc.form_chunks(v, t)
        nv, nt, nc = struct.unpack_from("<3i", blob, 0)
        assert (nv, nt, nc) == (len(new_v), len(t), len(vc))
        pos = 12
        assert np.array_equal(np.frombuffer(blob, "<i4", nc, pos), vc); pos += 4 * nc
        assert np.array_equal(np.frombuffer(blob, "<i4", nc, pos), tc); pos += 4 * nc
        xyz = np.frombuffer(blob, "<f4", 3 * nv, pos).reshape(-1, 3); pos += 12 * nv
        assert np.array_equal(xyz[:, 0].view(np.uint32), new_v["X"].view(np.uint32)) and np.array_equal(xyz[:, 2].view(np.uint32), new_v["Z"].view(np.uint32))
        rgb = np.frombuffer(blob, "u1", 3 * nv, pos).reshape(-1, 3); pos += 3 * nv
        assert np.array_equal(rgb[:, 0], new_v["R"]) and np.array_equal(rgb[:, 1], new_v["G"]) and np.array_equal(rgb[:, 2], new_v["B"])
        assert np.array_equal(np.frombuffer(blob, "<i4", 3 * nt, pos), new_t.ravel()[:3 * nt]); pos += 12 * nt
        assert pos == len(blob)


def test_ply_binary_layout(orc):
    rng = np.random.default_rng(12)
    v, tri = grid_mesh(rng, 20, 10)
    blob = orc.ply_binary(v, tri)
    head, _, body = blob.partition(b"end_header\n")
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\r\nelement vertex %d\n" % len(v))
    assert b"element face %d\nproperty list uchar int vertex_index\n" % len(tri) in head
    assert len(body) == 15 * len(v) + 13 * len(tri)
    rec = np.frombuffer(body, dtype=np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1")]), count=len(v))
    assert np.array_equal(rec["x"].view(np.uint32), v["X"].view(np.uint32)) and np.array_equal(rec["b"], v["B"])
    faces = np.frombuffer(body, dtype=np.dtype([("n", "u1"), ("i", "<i4", (3,))]), count=len(tri), offset=15 * len(v))
    assert (faces["n"] == 3).all() and np.array_equal(faces["i"], tri)
    assert native.ply_binary_bytes(len(v), len(tri)) == len(blob)


def bodies_block(rng, n_bodies, n_joints):
    out = struct.pack("<i", n_bodies)
    for _ in range(n_bodies):
        out += struct.pack("<?i", bool(rng.integers(0, 2)), n_joints)
        for j in range(n_joints):
            out += struct.pack("<ii3f2f", j, int(rng.integers(0, 3)), *rng.normal(size=5).astype(np.float32))
    return out


def test_frame_message_and_recording(orc):
    rng = np.random.default_rng(13)
    w, h = 64, 48
    depth = rng.integers(0, 6000, size=(h, w)).astype(np.uint16)
    rgb = rng.integers(0, 256, size=(h, w, 3)).astype(np.uint8)
    bodies = bodies_block(rng, 2, 25)
    msg = orc.frame_encode(depth, rgb, bodies)
    # byte-level layout (liveScanClient.cpp:281-288, :209-268)
    size, comp, ww, hh = struct.unpack_from("<4i", msg, 0)
    assert (size, comp, ww, hh) == (5 * w * h + len(bodies), 0, w, h) and len(msg) == 16 + size
    assert msg[16:16 + 2 * w * h] == depth.tobytes() and msg[16 + 2 * w * h:16 + 5 * w * h] == rgb.tobytes() and msg[16 + 5 * w * h:] == bodies
    d2, c2, b2, nb = orc.frame_decode(msg)
    assert np.array_equal(d2, depth) and np.array_equal(c2, rgb) and b2 == bodies and nb == 2
    # the product's host-side parser agrees with the oracle, raw and through zstd
    assert native.frame_encode(depth, rgb, bodies, 0) == msg
    for lvl in (0, 3):
        if lvl and not native.zstd_available():
            continue
        m = native.frame_encode(depth, rgb, bodies, lvl)
        d3, c3, b3, nb3 = native.frame_decode(m)
        assert np.array_equal(d3, depth) and np.array_equal(c3, rgb) and b3 == bodies and nb3 == 2
        if lvl:
            assert struct.unpack_from("<4i", m, 0)[1:] == (1, w, h) and m[16:20] == b"\x28\xb5\x2f\xfd"   # zstd frame magic
    # malformed messages
    assert orc.frame_decode(msg[:100]) is None
    broken = bytearray(msg); broken[16 + 5 * w * h] = 9           # nBodies = 9 but only 2 serialized
    assert orc.frame_decode(bytes(broken)) is None
    with pytest.raises(native.NativeUtilsError):
        native.frame_decode(bytes(broken))
    assert native.frame_parse_header(struct.pack("<4i", 0, 0, w, h)) is None      # "no more frames"
    # recording file: writer format, reader tolerance (fscanf) and product/oracle agreement
    rec = orc.recording_append(msg, 40) + orc.recording_append(b"", 73) + orc.recording_append(msg[:16] + msg[16:], 110)
    assert rec.startswith(b"bufferSize= %d\nframe_timestamp= 40\n" % len(msg))
    assert native.recording_append(msg, 40) + native.recording_append(b"", 73) + native.recording_append(msg, 110) == rec
    want = [(40, msg), (73, b""), (110, msg)]
    assert orc.recording_frames(rec) == want
    assert list(native.recording_frames(rec)) == want
    with pytest.raises(native.NativeUtilsError):
        list(native.recording_frames(rec[:-50]))


def test_inbound_parsers_survive_corrupted_input():
    """The frame message and the recording file come from the network / from disk: truncated, bit-flipped or random input
    must end in NativeUtilsError (or a clean end of iteration), never in a crash or an out-of-bounds read."""
    rng = np.random.default_rng(77)
    w, h = 32, 24
    depth = rng.integers(0, 6000, size=(h, w)).astype(np.uint16)
    rgb = rng.integers(0, 256, size=(h, w, 3)).astype(np.uint8)
    bodies = bodies_block(rng, 1, 25)
    msgs = [native.frame_encode(depth, rgb, bodies, 0)]
    if native.zstd_available():
        msgs.append(native.frame_encode(depth, rgb, bodies, 3))
    outcomes = {"ok": 0, "error": 0, "end": 0}
    for trial in range(1500):
        m = bytearray(msgs[trial % len(msgs)])
        kind = trial % 5
        if kind == 0:
            m = m[:int(rng.integers(0, len(m)))]                              # truncated
        elif kind == 1:
            for _ in range(int(rng.integers(1, 8))):
                m[int(rng.integers(0, len(m)))] ^= 1 << int(rng.integers(0, 8))   # bit flips anywhere
        elif kind == 2:
            m[:16] = rng.integers(0, 256, 16, dtype=np.uint8).tobytes()       # random header
        elif kind == 3:
            m[0:4] = struct.pack("<i", int(rng.integers(-2**31, 2**31)))      # absurd payload length
        else:
            m = bytearray(rng.integers(0, 256, int(rng.integers(0, 200)), dtype=np.uint8).tobytes())
        try:
            r = native.frame_decode(bytes(m))
            outcomes["end" if r is None else "ok"] += 1
        except native.NativeUtilsError:
            outcomes["error"] += 1
    assert outcomes["error"] > 300 and outcomes["ok"] > 0
    rec = native.recording_append(msgs[0], 1) + native.recording_append(msgs[0], 2)
    for trial in range(300):
        r = bytearray(rec)
        if trial % 2:
            r = r[:int(rng.integers(0, len(r)))]
        else:
            for _ in range(3):
                r[int(rng.integers(0, 60))] = int(rng.integers(0, 256))       # mangle the first record's text header
        try:
            for _, frame in native.recording_frames(bytes(r)):
                try:
                    native.frame_decode(frame)
                except native.NativeUtilsError:
                    pass
        except native.NativeUtilsError:
            pass
