"""GPU parity of the outbound formats against the CPU oracle, bit for bit: the TransferSocket.SendFrame
stream with TransferServer's chunking (LiveScanServer/TransferServer.cs:177-270, TransferSocket.cs:50-104) and the
binary PLY file image (LiveScanServer/Utils.cs:222-262), both built on the device through the C-ABI."""
import numpy as np
import pytest

from livescan3d_amd import native, synth

pytestmark = pytest.mark.gpu


def _cloud(rng, n):
    v = np.zeros(n, dtype=native.VERTEX_DTYPE)
    v["R"], v["G"], v["B"] = rng.integers(0, 256, n), rng.integers(0, 256, n), rng.integers(0, 256, n)
    v["A"] = 255
    v["X"], v["Y"], v["Z"] = rng.normal(size=n), rng.normal(size=n), rng.normal(size=n)
    return v


def _device(v, tri):
    import torch
    dv = torch.from_numpy(v.view(np.uint8).reshape(-1, 16).copy()).cuda() if len(v) else torch.zeros((1, 16), dtype=torch.uint8, device="cuda")
    dt = torch.from_numpy(np.ascontiguousarray(tri, dtype=np.int32).reshape(-1, 3)).cuda() if len(tri) else None
    return dv, dt


def _pack(v, tri, packer=None):
    import torch
    dv, dt = _device(v, tri)
    nt = 0 if dt is None else len(tri)
    cap = native.transfer_frame_bound(len(v), nt)
    out = torch.zeros(cap + 64, dtype=torch.uint8, device="cuda")
    p = packer or native.TransferPacker(0, max(len(v), 1), max(nt, 1))
    n = p.pack(dv.data_ptr(), len(v), 0 if dt is None else dt.data_ptr(), nt, out.data_ptr(), cap)
    assert (out[n:] == 0).all()                                 # nothing written past the stream
    _pack.last_path = p.last_path()
    return out[:n].cpu().numpy().tobytes()


def _ply(v, tri, misalign=0):
    import torch
    dv, dt = _device(v, tri)
    nt = 0 if dt is None else len(tri)
    need = native.ply_binary_bytes(len(v), nt)
    out = torch.zeros(need + 64 + misalign, dtype=torch.uint8, device="cuda")
    n = native.ply_pack(0, dv.data_ptr(), len(v), 0 if dt is None else dt.data_ptr(), nt, out.data_ptr() + misalign, need)
    torch.cuda.synchronize()
    assert n == need and (out[:misalign] == 0).all() and (out[misalign + n:] == 0).all()
    return out[misalign:misalign + n].cpu().numpy().tobytes()


@pytest.mark.parametrize("n", [1, 5, 1023, 1025, 64997, 64998, 2 * 64997 + 17])
def test_vertices_only_stream(gpu, orc, n):
    v = _cloud(np.random.default_rng(n), n)
    none = np.zeros((0, 3), np.int32)
    assert _pack(v, none) == orc.transfer_frame(v, none)
    assert _ply(v, none) == orc.ply_binary(v, none)


def _grid(rng, w, h, drop=0.03):
    present = rng.random((h, w)) >= drop
    ids = np.where(present.ravel(), np.cumsum(present.ravel()) - 1, -1).reshape(h, w)
    p, u, ur, r = ids[1:, :-1], ids[:-1, :-1], ids[:-1, 1:], ids[1:, 1:]
    a = np.stack([r, u, p], -1)
    b = np.stack([r, ur, u], -1)
    both = np.stack([a, b], 2).reshape(-1, 3)                   # per pixel: (R,U,P) then (R,UR,U), raster order
    tri = both[(both >= 0).all(1)].astype(np.int32)
    return _cloud(rng, int(present.sum())), tri


@pytest.mark.parametrize("w,h", [(7, 3), (40, 30), (400, 300), (1024, 700)])
def test_mesh_stream_matches_oracle(gpu, orc, w, h):
    v, tri = _grid(np.random.default_rng(w * h), w, h)
    got, want = _pack(v, tri), orc.transfer_frame(v, tri)
    assert len(got) == len(want)
    assert got == want
    for mis in (0, 1, 3):
        assert _ply(v, tri, mis) == orc.ply_binary(v, tri)


def test_chunk_that_spans_several_windows_and_reuse(gpu, orc):
    """Triangles that keep re-using a small vertex set never fill a chunk: the chunk must carry over many search windows
    (one window = 196608 triangles), then a grid part closes chunks normally; one packer serves several calls."""
    rng = np.random.default_rng(99)
    v, grid = _grid(rng, 600, 500)
    few = rng.integers(0, 1500, size=(450_000, 3)).astype(np.int32)
    packer = native.TransferPacker(0, len(v), len(few) + len(grid))
    for tri, path in ((few, 2), (np.concatenate([few, grid]), 2), (np.concatenate([grid[:150_000], few[:250_000], grid[150_000:]]), 2), (grid[:10], 1)):
        assert _pack(v, tri, packer) == orc.transfer_frame(v, tri)
        assert _pack.last_path == path


def test_which_path_forms_the_chunks(gpu, orc):
    """Grid meshes take the all-chunks-at-once path (1); a vertex used more than 16 times or two uses of one vertex a chunk's worth of
    index positions apart take the chunk-after-chunk walk (2); both give the reference's bytes, at the limits of either condition too."""
    rng = np.random.default_rng(5)
    v, grid = _grid(rng, 700, 420)
    none = np.zeros((0, 3), np.int32)
    assert _pack(v, none) == orc.transfer_frame(v, none) and _pack.last_path == 0
    assert _pack(v, grid) == orc.transfer_frame(v, grid) and _pack.last_path == 1
    hub = int(grid[1000, 0])
    uses = int((grid == hub).sum())
    assert uses <= 6
    far = grid[grid.min(1) > hub + 3000][:40]

    def fan(extra):                                             # `extra` more triangles on the hub vertex, right behind its own
        t = far[:extra].copy()
        t[:, 0] = hub
        return np.concatenate([grid[:1010], t, grid[1010:]])

    for extra, path in ((16 - uses, 1), (17 - uses, 2)):
        tri = fan(extra)
        assert int((tri == hub).sum()) == 16 + (path == 2)
        assert _pack(v, tri) == orc.transfer_frame(v, tri) and _pack.last_path == path
    # a link of exactly 64996 / 64997 index positions: vertex 0's last use in the grid, then once more that much later
    last = int(np.flatnonzero((grid == 0).ravel()).max())
    for gap, path in ((64996, 1), (64997, 2)):
        pos = last + gap
        tri = grid.copy()
        tri.reshape(-1)[pos] = 0
        assert _pack(v, tri) == orc.transfer_frame(v, tri) and _pack.last_path == path
    # degenerate triangles (one vertex three times), and a mesh that is one short of / exactly at / one past the first chunk end
    tri = grid.copy()
    tri[5000:5050] = tri[5000:5050, :1]
    assert _pack(v, tri) == orc.transfer_frame(v, tri) and _pack.last_path == 1
    want = orc.transfer_frame(v, grid)
    first = int(np.frombuffer(want, "<i4", 4)[3])
    assert first >= 64997
    n_chunks = int(np.frombuffer(want, "<i4", 3)[2])
    t0 = int(np.frombuffer(want, "<i4", 4 + 2 * n_chunks)[3 + n_chunks]) + 1     # triangles of the first chunk (its count is one short, :244)
    for nt in (t0 - 1, t0, t0 + 1, 21665, 21666):
        assert _pack(v, grid[:nt]) == orc.transfer_frame(v, grid[:nt]) and _pack.last_path == 1, nt


@pytest.mark.parametrize("seed", range(12))
def test_random_meshes_on_either_path(gpu, orc, seed):
    """Grid meshes disturbed at random -- indices replaced by nearby or by far-away vertices (valences up to and past 16, links up to
    and past a chunk's worth of positions), triangles dropped, duplicated and shuffled in blocks: whichever path the packer takes, the
    bytes are the reference's."""
    rng = np.random.default_rng(1000 + seed)
    w, h = int(rng.integers(200, 700)), int(rng.integers(150, 450))
    v, tri = _grid(rng, w, h, drop=float(rng.uniform(0.0, 0.2)))
    tri = tri.copy()
    n = len(tri)
    flat = tri.reshape(-1)
    k = int(rng.integers(0, n // 50 + 1))                        # nearby replacements: more uses of some vertices
    pos = rng.integers(0, flat.size, k)
    flat[pos] = np.clip(flat[pos] + rng.integers(-3, 4, k), 0, len(v) - 1)
    if seed % 3 == 1:                                            # a few far-away ones: long links (usually the walk)
        pos = rng.integers(0, flat.size, 5)
        flat[pos] = rng.integers(0, len(v), 5)
    if seed % 4 == 2:                                            # blocks of triangles swapped: vertices first used late
        a, b = sorted(rng.integers(0, n - 2000, 2))
        if b - a > 2000:
            blk = tri[a:a + 1000].copy()
            tri[a:a + 1000] = tri[b:b + 1000]
            tri[b:b + 1000] = blk
    keep = rng.random(n) >= rng.uniform(0.0, 0.3)
    tri = np.concatenate([tri[keep], tri[: int(rng.integers(0, 500))]])
    got, want = _pack(v, tri), orc.transfer_frame(v, tri)
    assert len(got) == len(want) and got == want, (seed, _pack.last_path)
    assert _pack.last_path in (1, 2)


def test_the_stream_does_not_depend_on_the_run(gpu):
    """The all-chunks-at-once path collects the uses of a vertex with atomics, in whatever order they land: the stream must not show it."""
    v, tri = _grid(np.random.default_rng(77), 900, 500)
    packer = native.TransferPacker(0, len(v), len(tri))
    first = _pack(v, tri, packer)
    assert _pack.last_path == 1
    for _ in range(7):
        assert _pack(v, tri, packer) == first


def test_fused_mesh_of_eight_sensors(gpu, orc):
    """The real thing: the merged mesh of 8 x 512x424 sensors (about 1 M vertices, 1.7 M triangles, ~25 chunks)."""
    rig = synth.make_rig("scene", 8, 512, 424, seed=3)
    v, tri = native.generate_mesh_from_depth_maps(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    assert len(tri) > 1_000_000
    got, want = _pack(v, tri), orc.transfer_frame(v, tri)
    assert len(got) == len(want) and got == want
    assert _pack.last_path == 1
    assert int(np.frombuffer(got, "<i4", 3)[2]) > 10
    assert _ply(v, tri, 2) == orc.ply_binary(v, tri)


def test_errors(gpu):
    import torch
    v = _cloud(np.random.default_rng(1), 100)
    dv, _ = _device(v, np.zeros((0, 3), np.int32))
    out = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    p = native.TransferPacker(0, 100, 10)
    bad = torch.tensor([[0, 1, 2], [3, 100, 4]], dtype=torch.int32, device="cuda")
    with pytest.raises(native.NativeUtilsError, match="outside"):
        p.pack(dv.data_ptr(), 100, bad.data_ptr(), 2, out.data_ptr(), out.numel())
    with pytest.raises(native.NativeUtilsError, match="capacity"):
        p.pack(dv.data_ptr(), 101, 0, 0, out.data_ptr(), out.numel())
    with pytest.raises(native.NativeUtilsError, match="needs"):
        p.pack(dv.data_ptr(), 100, 0, 0, out.data_ptr(), 100)
    with pytest.raises(native.NativeUtilsError):
        native.ply_pack(0, dv.data_ptr(), 100, 0, 0, out.data_ptr(), 10)


def test_last_mesh_exports_for_host_callers(gpu, orc):
    """LiveScanServer's order of calls: generateMeshFromDepthMaps, then the stream / the file of that same mesh without
    uploading it again (lsnLastMeshTransferFrame / lsnLastMeshPly read the copy that is still in HBM)."""
    rig = synth.make_rig("scene", 3, 512, 424, seed=4)
    v, tri = native.generate_mesh_from_depth_maps(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    assert native.last_mesh_transfer_frame() == orc.transfer_frame(v, tri)
    assert native.last_mesh_ply() == orc.ply_binary(v, tri)
    # a single-sensor call has no triangles: vertices-only chunks (TransferServer.cs:152-157)
    v1 = native.generate_vertices_from_depth_map(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds, 1)
    none = np.zeros((0, 3), np.int32)
    assert native.last_mesh_transfer_frame() == orc.transfer_frame(v1, none)
    assert native.last_mesh_ply() == orc.ply_binary(v1, none)


def test_handles_release_their_device_memory(gpu):
    """Creating and destroying plans / workspaces / packers repeatedly must not leak HBM (the server re-creates its plan
    whenever a client with another frame size connects)."""
    import torch
    from livescan3d_amd.fusion import DeviceFusion
    rig = synth.make_rig("noise", 2, 512, 424, seed=2)

    def cycle():
        fus = DeviceFusion(2, rig.widths, rig.heights)
        fus.set_params(rig.intr, rig.wt, rig.bounds)
        d = torch.from_numpy(np.stack([rig.depth_maps.view(np.int16)] * 2)).cuda()
        c = torch.from_numpy(np.stack([rig.depth_colors] * 2)).cuda()
        for _ in range(3):
            fus.run(d, c)                                           # third run: thresholds exist
        st = int(torch.cuda.current_stream().cuda_stream)
        fus.plan.radial_correct(rig.intr, d.data_ptr(), c.data_ptr(), st)
        ws = native.IcpWorkspace(0, 50_000, 50_000)
        pk = native.TransferPacker(0, 100_000, 200_000)
        torch.cuda.synchronize()
        ws.close(); pk.close(); fus.plan.close()
        del fus, d, c

    cycle()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(8):
        cycle()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, f"{(free0 - free1) >> 20} MiB of HBM lost over 8 create/destroy cycles"
