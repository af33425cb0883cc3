"""GPU parity of the fused unproject + transform + crop + compaction path against the CPU oracle.

Bar: bit-exact -- vertex counts, per-sensor offsets, order, RGBA bytes and XYZ bit patterns
(the reference's own correctness check is a bit-for-bit mesh compare, src/NativeUtils/main.cpp:211-245).
All calls go through the C-ABI (ctypes)."""
import os

import numpy as np
import pytest

from livescan3d_amd import native, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_cloud(orc, rig):
    return orc.generate_mesh_vertices(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)


def _assert_same(got, want, what):
    assert got.shape == want.shape, f"{what}: vertex count {got.shape} != {want.shape}"
    assert got.tobytes() == want.tobytes(), f"{what}: vertex bytes differ"


@pytest.mark.parametrize("kind,n,w,h", [
    ("noise", 1, 64, 48), ("noise", 3, 64, 48), ("scene", 2, 512, 424), ("noise", 2, 512, 424),
    ("scene", 8, 512, 424), ("noise", 2, 1024, 1024),
])
def test_export_generate_mesh_matches_oracle(gpu, orc, kind, n, w, h):
    rig = synth.make_rig(kind, n, w, h, seed=3)
    verts, tris = native.generate_mesh_from_depth_maps(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights,
                                                       rig.intr, rig.wt, rig.bounds)
    want, counts, want_tri = orc.generate_mesh(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    assert len(want) > 0 and len(want) < rig.widths.astype(np.int64) @ rig.heights   # crop and validity both bite
    _assert_same(verts, want, f"{kind} {n}x{w}x{h}")
    assert (verts["A"] == 255).all()
    # the always-on triangulation (meshGenerator.cpp): same triangles, same order, indices into the merged cloud
    assert tris.shape == want_tri.shape, (tris.shape, want_tri.shape)
    assert np.array_equal(tris, want_tri)
    if kind == "scene":
        assert len(want_tri) > len(want) // 2


def test_export_single_sensor_matches_oracle(gpu, orc):
    rig = synth.make_rig("scene", 3, 512, 424, seed=5)
    for i in range(rig.n):
        got = native.generate_vertices_from_depth_map(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights,
                                                      rig.intr, rig.wt, rig.bounds, i)
        want = orc.generate_vertices_from_depth_map(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights,
                                                    rig.intr, rig.wt, rig.bounds, i)
        _assert_same(got, want, f"sensor {i}")


def test_ragged_sizes_and_unaligned_rows(gpu, orc):
    """Sensors of different sizes, widths that are not multiples of 8, a partial last tile (scalar-load path)."""
    depths, rgbs, intr, wt = [], [], [], []
    for s, (w, h) in enumerate([(61, 37), (512, 424), (100, 3), (7, 5), (2049, 1)]):
        d, c = synth.noise_frame(11, 0, s, w, h)
        depths.append(d); rgbs.append(c)
        intr.append(synth.kinect_intrinsics(w, h))
        R, t = synth.ring_pose(s, 5)
        wt.append(synth.pack_pose(R, t))
    rig = synth.Rig(depths, rgbs, np.concatenate(intr), np.concatenate(wt), [-1.0, -1.2, -1.5, 1.3, 1.1, 1.6])
    verts, tris = native.generate_mesh_from_depth_maps(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights,
                                                       rig.intr, rig.wt, rig.bounds)
    want, _, want_tri = orc.generate_mesh(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    _assert_same(verts, want, "ragged")
    assert np.array_equal(tris, want_tri)


def test_edge_cases_empty_full_and_nan(gpu, orc):
    w, h = 64, 48
    intr = synth.kinect_intrinsics(w, h)
    R, t = synth.ring_pose(0, 1)
    wt = synth.pack_pose(R, t)
    rgb = synth.noise_frame(1, 0, 0, w, h)[1]
    # all-invalid depth -> empty mesh, non-null triangles
    rig = synth.Rig([np.zeros((h, w), np.uint16)], [rgb], intr, wt, synth.DEFAULT_BOUNDS)
    verts, tris = native.generate_mesh_from_depth_maps(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    assert len(verts) == 0 and tris.size == 0
    # every pixel valid and inside the default +-5 m bounds -> all P vertices, max depth 65535 mm cropped by Z
    d = np.full((h, w), 1500, np.uint16)
    d[0, :] = 65535
    rig = synth.Rig([d], [rgb], intr, wt, synth.DEFAULT_BOUNDS)
    verts, _ = native.generate_mesh_from_depth_maps(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    want, _ = _oracle_cloud(orc, rig)
    _assert_same(verts, want, "full")
    assert len(want) == w * (h - 1)
    # bounds collapsed to a plane / inverted bounds
    for b in ([0, -5, -5, 0, 5, 5], [1, 1, 1, -1, -1, -1]):
        rig = synth.Rig([d], [rgb], intr, wt, b)
        verts, _ = native.generate_mesh_from_depth_maps(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
        want, _ = _oracle_cloud(orc, rig)
        _assert_same(verts, want, f"bounds {b}")
    # NaN in the pose: the reference's comparison chain keeps NaN coordinates (depthprocessing.cpp:162)
    wt_nan = wt.copy()
    wt_nan[0] = np.nan
    rig = synth.Rig([d], [rgb], intr, wt_nan, synth.DEFAULT_BOUNDS)
    verts, _ = native.generate_mesh_from_depth_maps(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    want, _ = _oracle_cloud(orc, rig)
    assert len(want) > 0 and np.isnan(want["X"]).any()
    _assert_same(verts, want, "nan pose")
    # zero focal length -> inf/NaN coordinates
    intr0 = intr.copy(); intr0[2] = 0.0
    rig = synth.Rig([d], [rgb], intr0, wt, synth.DEFAULT_BOUNDS)
    verts, _ = native.generate_mesh_from_depth_maps(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    want, _ = _oracle_cloud(orc, rig)
    _assert_same(verts, want, "fx = 0")


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_device_resident_batch_matches_oracle(gpu, orc, mode):
    """T ticks x N sensors in one launch sequence on HBM-resident inputs; both compaction modes."""
    import torch
    from livescan3d_amd.fusion import DeviceFusion
    T, N, w, h = 5, 3, 512, 424
    rigs = [synth.make_rig("scene" if k % 2 else "noise", N, w, h, seed=7, tick=k, bounds=synth.CROP_BOUNDS) for k in range(T)]
    fus = DeviceFusion(T, rigs[0].widths, rigs[0].heights, mode=mode)
    fus.set_params(rigs[0].intr, rigs[0].wt, rigs[0].bounds)
    depth = torch.from_numpy(np.stack([r.depth_maps.view(np.int16) for r in rigs])).cuda()
    rgb = torch.from_numpy(np.stack([r.depth_colors for r in rigs])).cuda()
    fus.run(depth, rgb)
    torch.cuda.synchronize()
    for k in range(T):
        rk = rigs[k]
        want, counts = orc.generate_mesh_vertices(rk.depth_maps, rk.depth_colors, rk.widths, rk.heights, rigs[0].intr, rigs[0].wt, rigs[0].bounds)
        got, off = fus.tick_cloud(k)
        _assert_same(got, want, f"tick {k} mode {mode}")
        assert list(np.diff(off)) == list(counts)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_full_size_properties(gpu, mode):
    """BASELINE config sizes (8 x 512x424 and 16 x 1024x1024) through size-independent properties:
    offsets are monotone and end at the count, every vertex is inside the crop box, the count equals an independent
    torch count of the same predicate, and the cloud is invariant under re-running (idempotence)."""
    import torch
    from livescan3d_amd.fusion import DeviceFusion
    for (N, w, h, T) in [(8, 512, 424, 4), (16, 1024, 1024, 1)]:
        depth, rgb = synth.noise_frames_torch("cuda", 21, T, N, w, h)
        intr = np.concatenate([synth.kinect_intrinsics(w, h)] * N)
        wt = np.concatenate([synth.pack_pose(*synth.ring_pose(s, N)) for s in range(N)])
        b = synth.CROP_BOUNDS
        fus = DeviceFusion(T, [w] * N, [h] * N, mode=mode)
        fus.set_params(intr, wt, b)
        v, off = fus.run(depth.view(T, -1), rgb.view(T, -1))
        torch.cuda.synchronize()
        off_h = off.cpu().numpy()
        first = v.clone()
        assert (np.diff(off_h, axis=1) >= 0).all() and (off_h[:, 0] == 0).all()
        assert (off_h[:, -1] > 0).all() and (off_h[:, -1] < N * w * h).all()
        for k in range(T):
            n = int(off_h[k, -1])
            xyz = v[k, :n, 4:].contiguous().view(torch.float32).view(n, 3)
            lo = torch.tensor(b[:3], device="cuda"); hi = torch.tensor(b[3:], device="cuda")
            assert bool(((xyz >= lo) & (xyz <= hi)).all())
            assert bool((v[k, :n, 3] == 255).all())
        v2, off2 = fus.run(depth.view(T, -1), rgb.view(T, -1))
        torch.cuda.synchronize()
        assert torch.equal(off2.cpu(), torch.from_numpy(off_h))
        for k in range(T):
            n = int(off_h[k, -1])
            assert torch.equal(v2[k, :n], first[k, :n])


def test_merge_shards_kernel_matches_single_plan(gpu, orc):
    """The multi-GPU assembly step on one GPU: two sensor blocks fused separately (as two ranks would), their slabs laid
    out like an all-gather result, packed by lsnMergeShards -> identical to fusing all sensors in one plan (formMesh order)."""
    import torch
    from livescan3d_amd.fusion import DeviceFusion
    T, S, w, h, G = 3, 4, 512, 424, 2
    mpr = S // G
    rigs = [synth.make_rig("noise" if k % 2 else "scene", S, w, h, seed=17, tick=k, bounds=synth.CROP_BOUNDS) for k in range(T)]
    P = w * h
    shard_cap = mpr * P
    g_verts = torch.zeros((G, T, shard_cap, 16), dtype=torch.uint8, device="cuda")
    g_off = torch.zeros((G, T, mpr + 1), dtype=torch.int32, device="cuda")
    for r in range(G):
        s0, s1 = r * mpr, (r + 1) * mpr
        fus = DeviceFusion(T, [w] * mpr, [h] * mpr)
        fus.set_params(rigs[0].intr[7 * s0:7 * s1], rigs[0].wt[12 * s0:12 * s1], rigs[0].bounds)
        depth = torch.from_numpy(np.stack([rk.depth_maps.view(np.int16)[P * s0:P * s1] for rk in rigs])).cuda()
        rgb = torch.from_numpy(np.stack([rk.depth_colors[3 * P * s0:3 * P * s1] for rk in rigs])).cuda()
        v, o = fus.run(depth, rgb)
        torch.cuda.synchronize()
        g_verts[r].copy_(v)
        g_off[r].copy_(o)
    merged = torch.zeros((T, shard_cap * G, 16), dtype=torch.uint8, device="cuda")
    merged_off = torch.zeros((T, S + 1), dtype=torch.int32, device="cuda")
    native.merge_shards(0, G, T, mpr, g_verts.data_ptr(), shard_cap, g_off.data_ptr(), merged.data_ptr(), shard_cap * G,
                        merged_off.data_ptr(), int(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    mo = merged_off.cpu().numpy()
    for k in range(T):
        want, counts = orc.generate_mesh_vertices(rigs[k].depth_maps, rigs[k].depth_colors, rigs[k].widths, rigs[k].heights,
                                                  rigs[0].intr, rigs[0].wt, rigs[0].bounds)
        n = int(mo[k, -1])
        assert n == len(want) and list(np.diff(mo[k])) == list(counts)
        got = merged[k, :n].cpu().numpy().view(native.VERTEX_DTYPE).reshape(-1)
        _assert_same(got, want, f"merged tick {k}")


def test_triangulation_on_smooth_and_stepped_surfaces(gpu, orc):
    """Depth ramps, steps at the linearity thresholds, holes and image borders: every branch of checkTriangleConstraints
    (absolute, forward-linear, backward-linear, reject) and both triangle pairs, sizes with w % 8 == 0 and != 0."""
    rng = np.random.default_rng(12)
    for (w, h) in [(64, 48), (61, 37), (512, 424)]:
        yy, xx = np.mgrid[0:h, 0:w]
        base = 1200 + 3 * xx + 2 * yy                                   # smooth ramp: mostly triangles 0 and 1
        steps = base + 40 * ((xx // 7) % 2) + 25 * ((yy // 5) % 2)       # jumps around the 10..12 mm thresholds
        lin = 1000 + 14 * xx + 11 * yy                                  # steep but linear: the forward/backward rules pass
        holes = steps.copy(); holes[rng.random((h, w)) < 0.08] = 0
        noise = base + rng.integers(-9, 10, size=(h, w))
        for name, d in (("ramp", base), ("steps", steps), ("linear", lin), ("holes", holes), ("noise", noise)):
            depth = np.clip(d, 0, 65535).astype(np.uint16)
            rgb = synth.noise_frame(1, 0, 0, w, h)[1]
            rig = synth.Rig([depth], [rgb], synth.kinect_intrinsics(w, h), synth.pack_pose(*synth.ring_pose(0, 1)), synth.DEFAULT_BOUNDS)
            verts, tris = native.generate_mesh_from_depth_maps(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
            want, _, want_tri = orc.generate_mesh(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
            _assert_same(verts, want, f"{name} {w}x{h}")
            assert np.array_equal(tris, want_tri), f"{name} {w}x{h}: {tris.shape} vs {want_tri.shape}"
            if name in ("ramp", "linear") and w <= 64:
                assert len(want_tri) > 1.5 * (w - 4) * (h - 5)      # a smooth surface inside the bounds: ~2 triangles per pixel


def test_device_resident_mesh_batch(gpu, orc):
    """lsnFusionRunMesh on T ticks x N sensors resident in HBM: vertices, triangles and both offset tables."""
    import torch
    T, N, w, h = 3, 3, 512, 424
    rigs = [synth.make_rig("scene", N, w, h, seed=8, tick=k, bounds=synth.CROP_BOUNDS) for k in range(T)]
    plan = native.FusionPlan(0, T, rigs[0].widths, rigs[0].heights)
    plan.set_params(rigs[0].intr, rigs[0].wt, rigs[0].bounds)
    depth = torch.from_numpy(np.stack([r.depth_maps.view(np.int16) for r in rigs])).cuda()
    rgb = torch.from_numpy(np.stack([r.depth_colors for r in rigs])).cuda()
    cap = plan.capacity
    verts = torch.zeros((T, cap, 16), dtype=torch.uint8, device="cuda")
    off = torch.zeros((T, N + 1), dtype=torch.int32, device="cuda")
    tri = torch.zeros((T, 2 * cap, 3), dtype=torch.int32, device="cuda")
    toff = torch.zeros((T, N + 1), dtype=torch.int32, device="cuda")
    plan.run_mesh(depth.data_ptr(), rgb.data_ptr(), verts.data_ptr(), off.data_ptr(), tri.data_ptr(), toff.data_ptr(),
                  int(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    off_h, toff_h = off.cpu().numpy(), toff.cpu().numpy()
    for k in range(T):
        want, counts, want_tri = orc.generate_mesh(rigs[k].depth_maps, rigs[k].depth_colors, rigs[k].widths, rigs[k].heights,
                                                   rigs[0].intr, rigs[0].wt, rigs[0].bounds)
        nv, nt = int(off_h[k, -1]), int(toff_h[k, -1])
        assert nv == len(want) and nt == len(want_tri)
        assert verts[k, :nv].cpu().numpy().tobytes() == want.tobytes()
        assert np.array_equal(tri[k, :nt].cpu().numpy(), want_tri)
        assert (np.diff(toff_h[k]) >= 0).all() and toff_h[k, 0] == 0


def test_pipelined_calls_match(gpu, orc):
    """lsnFusionSetPipelined: count/scan of call k+1 on a side stream beside write(k); results identical, call after call."""
    import torch
    from livescan3d_amd.fusion import DeviceFusion
    T, N, w, h = 4, 2, 512, 424
    batches = []
    for b in range(3):
        rigs = [synth.make_rig("noise", N, w, h, seed=30 + b, tick=k, bounds=synth.CROP_BOUNDS) for k in range(T)]
        batches.append((rigs, torch.from_numpy(np.stack([r.depth_maps.view(np.int16) for r in rigs])).cuda(),
                        torch.from_numpy(np.stack([r.depth_colors for r in rigs])).cuda()))
    fus = DeviceFusion(T, [w] * N, [h] * N)
    fus.set_params(batches[0][0][0].intr, batches[0][0][0].wt, synth.CROP_BOUNDS)
    fus.plan.set_pipelined(True)
    outs = []
    for rep in range(2):
        for rigs, d, c in batches:               # back-to-back calls with different (resident) inputs
            v, o = fus.run(d, c)
            outs.append((rigs, v.clone(), o.clone()))   # clone is ordered after the call on the same stream
    torch.cuda.synchronize()
    fus.plan.set_pipelined(False)
    for rigs, v, o in outs:
        oh = o.cpu().numpy()
        for k in range(T):
            want, counts = orc.generate_mesh_vertices(rigs[k].depth_maps, rigs[k].depth_colors, rigs[k].widths, rigs[k].heights,
                                                      batches[0][0][0].intr, batches[0][0][0].wt, synth.CROP_BOUNDS)
            n = int(oh[k, -1])
            assert n == len(want) and list(np.diff(oh[k])) == list(counts)
            assert v[k, :n].cpu().numpy().tobytes() == want.tobytes()


def test_streamed_calls_match(gpu, orc):
    """lsnFusionRunStreamed: batch k is written while batch k+1 is counted inside the same kernel; a changed parameter set
    or an unexpected next batch falls back to counting on the spot.  Results identical to the oracle, call after call."""
    import torch
    T, N, w, h = 3, 2, 512, 424
    batches = []
    for b in range(4):
        rigs = [synth.make_rig("noise" if b % 2 else "scene", N, w, h, seed=40 + b, tick=k, bounds=synth.CROP_BOUNDS) for k in range(T)]
        batches.append((rigs, torch.from_numpy(np.stack([r.depth_maps.view(np.int16) for r in rigs])).cuda(),
                        torch.from_numpy(np.stack([r.depth_colors for r in rigs])).cuda()))
    plan = native.FusionPlan(0, T, [w] * N, [h] * N)
    intr, wt = batches[0][0][0].intr, batches[0][0][0].wt
    plan.set_params(intr, wt, synth.CROP_BOUNDS)
    st = int(torch.cuda.current_stream().cuda_stream)
    outs = []
    order = [0, 1, 2, 3, 3, 1]                      # includes a repeated batch and a batch that was not announced
    announce = [1, 2, 3, None, 0, None]             # what each call names as "next" (the 5th call lies: next is 0, 1 comes)
    bounds = synth.CROP_BOUNDS
    for i, b in enumerate(order):
        if i == 2:                                   # parameters change: counts made ahead are void
            bounds = np.array([-1.0, -0.8, -1.2, 1.1, 1.2, 1.3], np.float32)
            plan.set_params(intr, wt, bounds)
        rigs, d, c = batches[b]
        v = torch.zeros((T, plan.capacity, 16), dtype=torch.uint8, device="cuda")
        o = torch.zeros((T, N + 1), dtype=torch.int32, device="cuda")
        nxt = batches[announce[i]][1].data_ptr() if announce[i] is not None else None
        plan.run_streamed(d.data_ptr(), c.data_ptr(), v.data_ptr(), o.data_ptr(), nxt, st)
        outs.append((rigs, v, o, bounds.copy()))
    torch.cuda.synchronize()
    for rigs, v, o, bnd in outs:
        oh = o.cpu().numpy()
        for k in range(T):
            want, counts = orc.generate_mesh_vertices(rigs[k].depth_maps, rigs[k].depth_colors, rigs[k].widths, rigs[k].heights, intr, wt, bnd)
            n = int(oh[k, -1])
            assert n == len(want) and list(np.diff(oh[k])) == list(counts)
            assert v[k, :n].cpu().numpy().tobytes() == want.tobytes()


def test_streamed_refilled_next_buffer_is_a_clean_error(gpu, orc):
    """A caller that refills the buffer it announced as "next" after lsnFusionRunStreamed has counted it (the natural double-buffer
    mistake) must get an error flag, not overlapping tiles and writes past the tick's slab: the write pass compares every tile's
    survivors with what the count pass had seen and writes nothing where they differ."""
    import torch
    T, N, w, h = 2, 2, 512, 424
    mk = lambda seed, kind: [synth.make_rig(kind, N, w, h, seed=seed, tick=k, bounds=synth.CROP_BOUNDS) for k in range(T)]
    up = lambda rigs: (torch.from_numpy(np.stack([r.depth_maps.view(np.int16) for r in rigs])).cuda(),
                       torch.from_numpy(np.stack([r.depth_colors for r in rigs])).cuda())
    ra, rb, rc = mk(70, "scene"), mk(71, "scene"), mk(72, "noise")        # rc: many more survivors than rb
    da, ca = up(ra)
    db, cb = up(rb)
    dc, cc = up(rc)
    plan = native.FusionPlan(0, T, [w] * N, [h] * N)
    plan.set_params(ra[0].intr, ra[0].wt, synth.CROP_BOUNDS)
    st = int(torch.cuda.current_stream().cuda_stream)
    guard = 4096
    v = torch.zeros((T * plan.capacity + guard, 16), dtype=torch.uint8, device="cuda")
    o = torch.zeros((T, N + 1), dtype=torch.int32, device="cuda")
    plan.run_streamed(da.data_ptr(), ca.data_ptr(), v.data_ptr(), o.data_ptr(), db.data_ptr(), st)   # writes A, counts B
    assert plan.check(st) == 0
    db.copy_(dc); cb.copy_(cc)                                                # the caller refills "next" with other frames
    v.zero_()
    plan.run_streamed(db.data_ptr(), cb.data_ptr(), v.data_ptr(), o.data_ptr(), None, st)             # stale counts meet new pixels
    assert plan.check(st) == 2
    assert plan.check(st) == 0                                                # the flag is cleared by the check
    assert int(v[T * plan.capacity:].count_nonzero()) == 0                    # nothing ran past the last tick's slab
    # the same call again counts on the spot (nothing valid was counted ahead any more) and is right
    plan.run_streamed(db.data_ptr(), cb.data_ptr(), v.data_ptr(), o.data_ptr(), None, st)
    assert plan.check(st) == 0
    oh = o.cpu().numpy()
    for k in range(T):
        want, counts = orc.generate_mesh_vertices(rc[k].depth_maps, rc[k].depth_colors, rc[k].widths, rc[k].heights, ra[0].intr, ra[0].wt, synth.CROP_BOUNDS)
        n = int(oh[k, -1])
        assert n == len(want) and v[k * plan.capacity:k * plan.capacity + n].cpu().numpy().tobytes() == want.tobytes()


def test_triangles_match_reference_fixture(gpu):
    """tests/golden/tri_reference.npz::scene{0,1}_96x80 are the triangle lists the REFERENCE's own meshGenerator.cpp
    produced (tests/golden/make_tri_golden.py) for the two sensors of this rig; the export must return the same
    triangles, rebased per sensor (formMesh, depthprocessing.cpp:1614-1626)."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tri_reference.npz"))
    rig = synth.make_rig("scene", 2, 96, 80, seed=9, perturb=False)
    verts, tris = native.generate_mesh_from_depth_maps(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights,
                                                       rig.intr, rig.wt, rig.bounds)
    n0 = int((z["scene0_96x80_p2v"] >= 0).sum())
    n1 = int((z["scene1_96x80_p2v"] >= 0).sum())
    assert len(verts) == n0 + n1
    want = np.concatenate([z["scene0_96x80_tri"], z["scene1_96x80_tri"] + n0])
    assert tris.shape == want.shape and np.array_equal(tris, want)


def test_exports_called_concurrently_like_the_two_background_workers(gpu, orc):
    """LiveScanServer calls the merge export from updateWorker and single-sensor + ICP from refineWorker at the same time
    (MainWindowForm.cs:266-300, 330-410).  ctypes drops the GIL during a call, so these threads really overlap."""
    import threading
    rig = synth.make_rig("scene", 3, 512, 424, seed=8, perturb=True)
    want_v, counts, want_t = orc.generate_mesh(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    e = np.concatenate([[0], np.cumsum(counts)])
    xyz = np.stack([want_v["X"], want_v["Y"], want_v["Z"]], axis=1).astype(np.float32)
    target, source = np.ascontiguousarray(xyz[e[0]:e[1]]), np.ascontiguousarray(xyz[e[1]:e[2]])
    ref_src, ref_R, ref_t = native.icp(target, source.copy(), max_iter=3)
    errors = []

    def merge_worker():
        try:
            for _ in range(6):
                v, t = native.generate_mesh_from_depth_maps(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
                assert v.tobytes() == want_v.tobytes() and np.array_equal(t, want_t)
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)

    def refine_worker():
        try:
            for _ in range(6):
                one = native.generate_vertices_from_depth_map(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds, 1)
                assert one.tobytes() == want_v[e[1]:e[2]].tobytes()
                s2, R, t = native.icp(target, source.copy(), max_iter=3)
                assert s2.tobytes() == ref_src.tobytes() and R.tobytes() == ref_R.tobytes() and t.tobytes() == ref_t.tobytes()
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)

    threads = [threading.Thread(target=merge_worker), threading.Thread(target=refine_worker)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_soak_of_mixed_export_calls(gpu, orc):
    """A few hundred export calls in random order with changing sensor counts, frame sizes, calibrations and crop boxes --
    what a long LiveScanServer session does to the library's plan cache, pinned-buffer pool and per-calibration tables
    (thresholds are built on the second call with an unchanged calibration, so repeats matter).  Every result against the oracle."""
    rng = np.random.default_rng(99)
    shapes = [(64, 48), (40, 56), (57, 31), (128, 96)]
    rigs = []
    for i in range(6):
        n = int(rng.integers(1, 4))
        w, h = shapes[int(rng.integers(len(shapes)))]
        rigs.append(synth.make_rig("scene" if i % 2 else "noise", n, w, h, seed=50 + i, perturb=bool(i % 3 == 0),
                                   bounds=[synth.CROP_BOUNDS, synth.DEFAULT_BOUNDS, [-0.6, -0.4, -0.9, 0.7, 0.5, 0.4]][i % 3]))
    expect = {}
    for step in range(240):
        i = int(rng.integers(len(rigs)))
        rig = rigs[i]
        op = int(rng.integers(4))
        if op == 0 or op == 3:
            key = (i, "mesh")
            if key not in expect:
                v, _, t = orc.generate_mesh(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
                expect[key] = (v.tobytes(), t)
            verts, tris = native.generate_mesh_from_depth_maps(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
            assert verts.tobytes() == expect[key][0] and np.array_equal(tris, expect[key][1]), f"step {step}: merge call, rig {i}"
            if op == 3:
                assert native.last_mesh_ply() == orc.ply_binary(verts, tris), f"step {step}: PLY of rig {i}"
        elif op == 1:
            s = int(rng.integers(rig.n))
            key = (i, "one", s)
            if key not in expect:
                v, counts = orc.generate_mesh_vertices(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
                e = np.concatenate([[0], np.cumsum(counts)])
                expect[key] = v[e[s]:e[s + 1]].tobytes()
            one = native.generate_vertices_from_depth_map(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds, s)
            assert one.tobytes() == expect[key], f"step {step}: single sensor {s} of rig {i}"
        else:
            key = (i, "radial")
            if key not in expect:
                expect[key] = orc.radial_correction(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr)
            d, c = native.radial_correction(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr)
            assert np.array_equal(np.asarray(d).view(np.uint8).ravel(), np.asarray(expect[key][0]).view(np.uint8).ravel()), f"step {step}: radial depth, rig {i}"
            assert np.array_equal(np.asarray(c).ravel(), np.asarray(expect[key][1]).ravel()), f"step {step}: radial colour, rig {i}"
        if step % 40 == 39:                                         # a recalibration of one rig in the middle of the session
            j = int(rng.integers(len(rigs)))
            rigs[j].wt = (rigs[j].wt + rng.normal(scale=0.01, size=rigs[j].wt.shape)).astype(np.float32)
            for k in [k for k in expect if k[0] == j and k[1] != "radial"]:
                del expect[k]


# ---- the host flows of the exports (host_flows.hip): upload schedule, groups, kernel stores / copy engine ------------------------------------

_FLOW_SCRIPT = r"""
import hashlib, json, sys
import numpy as np
sys.path.insert(0, {root!r})
from livescan3d_amd import native, synth
out = {{}}
for name, kind, n, w, h in {cases!r}:
    rig = synth.make_rig(kind, n, w, h, seed=17, bounds=synth.CROP_BOUNDS)
    v, t = native.generate_mesh_from_depth_maps(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    v2, t2 = native.generate_mesh_from_depth_maps(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    assert v.tobytes() == v2.tobytes() and t.tobytes() == t2.tobytes(), name + ": the second call differs from the first"
    v1, t1, d1, c1 = native.correct_and_generate_mesh(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds, write_back=True)
    frame = native.last_mesh_transfer_frame()      # the mesh of the last call, rebuilt in / read from HBM
    rd, rc = native.radial_correction(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr)   # the radial export on its own
    out[name] = [len(v), len(t), hashlib.sha256(v.tobytes() + t.tobytes()).hexdigest(),
                 len(v1), len(t1), hashlib.sha256(v1.tobytes() + t1.tobytes() + np.asarray(d1).tobytes() + np.asarray(c1).tobytes()).hexdigest(),
                 hashlib.sha256(frame).hexdigest(),
                 hashlib.sha256(np.ascontiguousarray(np.asarray(rd)).view(np.uint8).tobytes() + np.ascontiguousarray(np.asarray(rc)).tobytes()).hexdigest()]
print(json.dumps(out))
"""

_FLOW_CASES = [("two_groups", "scene", 3, 512, 424),      # D[0-2] C[0-1] | C[2]: a short tail joins its predecessor or stands alone
               ("four_groups", "scene", 8, 256, 424),     # colour runs of 5 sensors: 8 sensors -> 2 groups, depth in one run
               ("big_frames", "noise", 3, 1024, 768),     # every sensor its own group (2.4 MB of colours each)
               ("many_sensors", "noise", 20, 640, 560),   # more sensors with >= 1 MiB of colours than group events: regrouped
               ("odd_sizes", "scene", 5, 250, 121)]       # slices that break the wide-load alignment: one group


@pytest.mark.parametrize("env", [{}, {"LSN_HOST_PATH": "direct"}, {"LSN_HOST_PATH": "grouped"}, {"LSN_HOST_GROUP": "1"},
                                 {"LSN_HOST_GROUP": "3", "LSN_HOST_PATH": "grouped"},
                                 {"LSN_HOST_DEVICES": "0,0"}, {"LSN_HOST_DEVICES": "0,0,0"}, {"LSN_HOST_DEVICES": "0,0,0,0,0,0,0,0"}])
def test_every_host_flow_returns_the_oracles_mesh(gpu, orc, env):
    """generateMeshFromDepthMaps and lsnCorrectAndGenerateMesh through every flow of the library (kernel stores into the pinned mesh
    blocks / mesh in HBM + copy engine; sensors per upload group by size, forced to 1 and to 3; the call sharded over 2, 3 and 8 "devices"
    -- the one GPU of the box listed several times, the only rehearsal a one-GPU box allows: every shard has its own thread, streams,
    buffers and plans, only the links are not distinct) on rigs that exercise the schedule's corners: the meshes, the written-back
    corrected maps and the TransferServer stream of the call's mesh must be the oracle's."""
    import hashlib
    import subprocess
    import sys as _sys
    import json
    from livescan3d_amd import synth
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([_sys.executable, "-c", _FLOW_SCRIPT.format(root=ROOT, cases=_FLOW_CASES)], capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    for name, kind, n, w, h in _FLOW_CASES:
        rig = synth.make_rig(kind, n, w, h, seed=17, bounds=synth.CROP_BOUNDS)
        want_v, _, want_t = orc.generate_mesh(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
        cd, cc = orc.radial_correction(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr)
        cd = np.ascontiguousarray(np.asarray(cd)).view(np.uint8).ravel()
        cc = np.ascontiguousarray(np.asarray(cc)).ravel()
        v1, _, t1 = orc.generate_mesh(cd, cc, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
        g = got[name]
        assert g[:2] == [len(want_v), len(want_t)], (name, env, g[:2])
        assert g[2] == hashlib.sha256(want_v.tobytes() + want_t.tobytes()).hexdigest(), (name, env, "merge call")
        assert g[3:5] == [len(v1), len(t1)], (name, env, g[3:5])
        assert g[5] == hashlib.sha256(v1.tobytes() + t1.tobytes() + cd.tobytes() + cc.tobytes()).hexdigest(), (name, env, "tick as one call")
        assert g[6] == hashlib.sha256(orc.transfer_frame(v1, t1)).hexdigest(), (name, env, "stream of the last mesh")
        assert g[7] == hashlib.sha256(cd.tobytes() + cc.tobytes()).hexdigest(), (name, env, "radial export")


def test_last_mesh_is_the_calling_threads_own(gpu, orc):
    """lsnLastMesh* return the mesh of the CALLING thread's last mesh call: a single-sensor call made on another thread (LiveScanServer's refine
    worker) between this thread's merge call and its lsnLastMeshTransferFrame does not change what this thread gets."""
    import threading
    rig = synth.make_rig("scene", 3, 256, 212, seed=23, bounds=synth.CROP_BOUNDS)
    v, t = native.generate_mesh_from_depth_maps(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    other = {}

    def refine_worker():
        other["v"] = native.generate_vertices_from_depth_map(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds, 1)
        other["frame"] = native.last_mesh_transfer_frame()
    th = threading.Thread(target=refine_worker)
    th.start()
    th.join()
    assert native.last_mesh_transfer_frame() == orc.transfer_frame(v, t)                                   # this thread: its merge call's mesh
    assert other["frame"] == orc.transfer_frame(other["v"], np.zeros((0, 3), dtype=np.int32))              # that thread: its single-sensor cloud


_EMPTY_PART_SCRIPT = r"""
import hashlib, json, sys
import numpy as np
sys.path.insert(0, {root!r})
from livescan3d_amd import native, synth
rig = synth.make_rig("scene", 5, 256, 212, seed=19, bounds=synth.CROP_BOUNDS)
d = rig.depth_maps.view(np.uint16).reshape(5, -1).copy()
d[[0, 1, 4]] = 0                                      # sensors without a single valid pixel
dm = d.view(np.uint8).ravel()
out = []
for _ in range(2):
    v, t = native.generate_mesh_from_depth_maps(dm, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    out.append([len(v), len(t), hashlib.sha256(v.tobytes() + t.tobytes()).hexdigest()])
d[:] = 0                                              # ... and a tick without any
v, t = native.generate_mesh_from_depth_maps(d.view(np.uint8).ravel(), rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
out.append([len(v), len(t), native.host_shards(5, 0)[1]])
print(json.dumps(out))
"""


def test_sharded_call_with_empty_parts(gpu, orc):
    """A call sharded over three "devices" whose first part has no vertex at all and whose last part has some in one of its two sensors: the
    bases the parts derive from each other's counts (0 for an empty part) and the triangle index rebase must still give the oracle's mesh;
    and a tick without a single valid pixel gives an empty mesh, not an error."""
    import hashlib
    import json
    import subprocess
    import sys as _sys
    from livescan3d_amd import synth
    e = dict(os.environ, LSN_HOST_DEVICES="0,0,0")
    r = subprocess.run([_sys.executable, "-c", _EMPTY_PART_SCRIPT.format(root=ROOT)], capture_output=True, text=True, env=e, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    rig = synth.make_rig("scene", 5, 256, 212, seed=19, bounds=synth.CROP_BOUNDS)
    d = rig.depth_maps.view(np.uint16).reshape(5, -1).copy()
    d[[0, 1, 4]] = 0
    want_v, counts, want_t = orc.generate_mesh(d.view(np.uint8).ravel(), rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    assert list(counts)[0] == 0 and list(counts)[1] == 0 and list(counts)[4] == 0 and len(want_v) > 1000 and len(want_t) > 1000
    want = [len(want_v), len(want_t), hashlib.sha256(want_v.tobytes() + want_t.tobytes()).hexdigest()]
    assert got[0] == want and got[1] == want, (got, want)
    assert got[2] == [0, 0, "0:[0] 1:[1-2] 2:[3-4]"], got[2]


_RAGGED_SHARD_SCRIPT = r"""
import hashlib, json, sys
import numpy as np
sys.path.insert(0, {root!r})
from livescan3d_amd import native, synth
sizes = {sizes!r}
out = []
for seed in (11, 12):
    depths, rgbs, intr, wt = [], [], [], []
    for s, (w, h) in enumerate(sizes):
        d, c = synth.scene_frame(seed, 0, s, len(sizes), w, h) if w >= 64 and h >= 48 else synth.noise_frame(seed, 0, s, w, h)
        depths.append(d); rgbs.append(c)
        intr.append(synth.kinect_intrinsics(w, h))
        wt.append(synth.pack_pose(*synth.ring_pose(s, len(sizes))))
    rig = synth.Rig(depths, rgbs, np.concatenate(intr), np.concatenate(wt), synth.CROP_BOUNDS)
    v, t = native.generate_mesh_from_depth_maps(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    v1, t1, d1, c1 = native.correct_and_generate_mesh(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds, write_back=True)
    out.append([len(v), len(t), hashlib.sha256(v.tobytes() + t.tobytes()).hexdigest(),
                hashlib.sha256(v1.tobytes() + t1.tobytes() + np.asarray(d1).tobytes() + np.asarray(c1).tobytes()).hexdigest()])
print(json.dumps(out))
"""

_RAGGED_SIZES = [(61, 37), (512, 424), (100, 30), (256, 212), (7, 5), (640, 48), (250, 121)]


@pytest.mark.parametrize("devices", ["0,0", "0,0,0", "0,0,0,0,0"])
def test_sharded_call_on_a_ragged_rig(gpu, orc, devices):
    """Seven sensors of seven sizes (widths that are not multiples of 8 among them: the pixel-by-pixel vertex map and its index rebase, parts
    whose blocks differ in every dimension) cut over 2, 3 and 5 "devices": merge call and tick-as-one-call must return the oracle's meshes
    and corrected maps."""
    import hashlib
    import json
    import subprocess
    import sys as _sys
    from livescan3d_amd import synth
    e = dict(os.environ, LSN_HOST_DEVICES=devices)
    r = subprocess.run([_sys.executable, "-c", _RAGGED_SHARD_SCRIPT.format(root=ROOT, sizes=_RAGGED_SIZES)], capture_output=True, text=True, env=e, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    n = len(_RAGGED_SIZES)
    for k, seed in enumerate((11, 12)):
        depths, rgbs, intr, wt = [], [], [], []
        for s, (w, h) in enumerate(_RAGGED_SIZES):
            d, c = synth.scene_frame(seed, 0, s, n, w, h) if w >= 64 and h >= 48 else synth.noise_frame(seed, 0, s, w, h)
            depths.append(d); rgbs.append(c)
            intr.append(synth.kinect_intrinsics(w, h))
            wt.append(synth.pack_pose(*synth.ring_pose(s, n)))
        rig = synth.Rig(depths, rgbs, np.concatenate(intr), np.concatenate(wt), synth.CROP_BOUNDS)
        want_v, _, want_t = orc.generate_mesh(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
        cd, cc = orc.radial_correction(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr)
        cd = np.ascontiguousarray(np.asarray(cd)).view(np.uint8).ravel()
        cc = np.ascontiguousarray(np.asarray(cc)).ravel()
        v1, _, t1 = orc.generate_mesh(cd, cc, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
        assert len(want_v) > 1000
        assert got[k][:3] == [len(want_v), len(want_t), hashlib.sha256(want_v.tobytes() + want_t.tobytes()).hexdigest()], (devices, seed, "merge call")
        assert got[k][3] == hashlib.sha256(v1.tobytes() + t1.tobytes() + cd.tobytes() + cc.tobytes()).hexdigest(), (devices, seed, "tick as one call")


@pytest.mark.parametrize("n,w,h", [(1, 512, 424), (8, 512, 424), (3, 250, 120), (2, 1024, 1024)])
def test_one_tick_plans_single_pass_and_three_launches_give_the_same_bytes(gpu, orc, monkeypatch, n, w, h):
    """A one-tick mode-0 plan of up to 2048 tiles takes the single pass by itself (fuse_kernel<4>); $LSN_ONE_TICK_SINGLE_PASS=0 / 1 (read
    when the plan is created) forces the three launches / the single pass.  lsnFusionRun AND lsnFusionRunMesh (whose triangle passes read
    the pixel -> vertex map the vertex pass leaves) must give the same bytes either way -- the oracle's -- and no device-side flag."""
    import torch
    rig = synth.make_rig("scene", n, w, h, seed=12, bounds=synth.CROP_BOUNDS)
    want_v, want_counts, want_t = orc.generate_mesh(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    want_t = np.asarray(want_t, np.int32).reshape(-1, 3)
    depth = torch.from_numpy(rig.depth_maps.view(np.int16).copy()).cuda().unsqueeze(0).contiguous()
    rgb = torch.from_numpy(rig.depth_colors.copy()).cuda().unsqueeze(0).contiguous()
    outs = {}
    for force in ("0", "1"):
        monkeypatch.setenv("LSN_ONE_TICK_SINGLE_PASS", force)
        plan = native.FusionPlan(0, 1, rig.widths, rig.heights)
        monkeypatch.delenv("LSN_ONE_TICK_SINGLE_PASS")
        plan.set_params(rig.intr, rig.wt, rig.bounds)
        cap = plan.capacity
        res = []
        for rep in range(2):   # the second run counts from the per-pixel thresholds the first one left
            v = torch.zeros((1, cap, 16), dtype=torch.uint8, device="cuda"); o = torch.full((1, n + 1), -7, dtype=torch.int32, device="cuda")
            plan.run(depth.data_ptr(), rgb.data_ptr(), v.data_ptr(), o.data_ptr())
            v2 = torch.zeros_like(v); o2 = torch.full_like(o, -7)
            t2 = torch.zeros((1, 2 * cap, 3), dtype=torch.int32, device="cuda"); to2 = torch.full((1, n + 1), -7, dtype=torch.int32, device="cuda")
            plan.run_mesh(depth.data_ptr(), rgb.data_ptr(), v2.data_ptr(), o2.data_ptr(), t2.data_ptr(), to2.data_ptr())
            torch.cuda.synchronize()
            assert plan.check() == 0
            nv, nt = int(o[0, -1]), int(to2[0, -1])
            assert nv == len(want_v) == int(o2[0, -1]) and nt == len(want_t)
            res.append((o.cpu().numpy().tobytes(), v[0, :nv].cpu().numpy().tobytes(), o2.cpu().numpy().tobytes(), v2[0, :nv].cpu().numpy().tobytes(),
                        to2.cpu().numpy().tobytes(), t2[0, :nt].cpu().numpy().tobytes()))
        assert res[0] == res[1]
        assert res[0][1] == want_v.tobytes() and res[0][3] == want_v.tobytes() and res[0][5] == want_t.tobytes()
        outs[force] = res[0]
        plan.close()
    assert outs["0"] == outs["1"]


@pytest.mark.parametrize("T,sizes,parts_env", [(9, [(512, 424)] * 3, None), (16, [(250, 120), (61, 37), (128, 96)], None), (3, [(128, 96)] * 2, None),
                                               (5, [(128, 96)] * 2, "2"), (12, [(256, 212)] * 2, "1")])
def test_tick_pipeline_gives_the_two_calls_bytes(gpu, orc, monkeypatch, T, sizes, parts_env):
    """lsnTickRun = lsnFusionRadialCorrectTo + lsnFusionRunMesh as one call; from 8 ticks up (or $LSN_TICK_PARTS=2) the batch runs as two
    halves side by side on two streams.  Corrected maps, clouds, offsets, triangles: the oracle's, byte for byte, for every tick -- odd
    tick counts, ragged rigs (the second half's slices then start unaligned), both settings of the split; twice, on a side stream."""
    import torch
    if parts_env:
        monkeypatch.setenv("LSN_TICK_PARTS", parts_env)
    N = len(sizes)
    rigs = []
    for k in range(T):
        depths, rgbs, intr, wt = [], [], [], []
        for s_, (w, h) in enumerate(sizes):
            d, c = synth.scene_frame(21, k, s_, N, w, h) if (w, h) == (512, 424) or k % 2 else synth.noise_frame(21, k, s_, w, h)
            depths.append(d); rgbs.append(c)
            ki = synth.kinect_intrinsics(w, h).copy(); ki[4:7] = [0.09, -0.05, 0.01]
            intr.append(ki.astype(np.float32))
            wt.append(synth.pack_pose(*synth.ring_pose(s_, N)))
        rigs.append(synth.Rig(depths, rgbs, np.concatenate(intr), np.concatenate(wt), synth.CROP_BOUNDS))
    r0 = rigs[0]
    tp = native.TickPipeline(0, T, r0.widths, r0.heights)
    assert tp.parts == (int(parts_env) if parts_env else (2 if T >= 8 else 1))
    tp.set_params(r0.intr, r0.wt, r0.bounds)
    cap, tcap = tp.capacity, tp.tri_capacity
    depth = torch.from_numpy(np.stack([r.depth_maps.view(np.int16) for r in rigs])).cuda().contiguous()
    rgb = torch.from_numpy(np.stack([r.depth_colors for r in rigs])).cuda().contiguous()
    st = torch.cuda.Stream()
    for rep in range(2):
        cd, cc = torch.zeros_like(depth), torch.zeros_like(rgb)
        v = torch.zeros((T, cap, 16), dtype=torch.uint8, device="cuda"); o = torch.full((T, N + 1), -7, dtype=torch.int32, device="cuda")
        tr = torch.zeros((T, tcap, 3), dtype=torch.int32, device="cuda"); to = torch.full((T, N + 1), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        tp.run(depth.data_ptr(), rgb.data_ptr(), cd.data_ptr(), cc.data_ptr(), v.data_ptr(), o.data_ptr(), tr.data_ptr(), to.data_ptr(), int(st.cuda_stream))
        st.synchronize()                                        # the call's work is complete on the CALLER's stream (the join)
        for k, r in enumerate(rigs):
            wd, wc = orc.radial_correction(r.depth_maps, r.depth_colors, r.widths, r.heights, r.intr)
            wd = np.ascontiguousarray(np.asarray(wd)).view(np.uint8).ravel(); wc = np.ascontiguousarray(np.asarray(wc)).ravel()
            assert cd[k].cpu().numpy().view(np.uint8).tobytes() == wd.tobytes() and cc[k].cpu().numpy().tobytes() == wc.tobytes(), (rep, k)
            want_v, want_counts, want_t = orc.generate_mesh(wd, wc, r.widths, r.heights, r.intr, r.wt, r.bounds)
            oh = o[k].cpu().numpy()
            assert oh[-1] == len(want_v) and np.array_equal(np.diff(oh), np.asarray(want_counts).ravel()[:N]), (rep, k)
            assert v[k, :len(want_v)].cpu().numpy().tobytes() == want_v.tobytes(), (rep, k)
            want_t = np.asarray(want_t, np.int32).reshape(-1, 3)
            assert int(to[k, -1]) == len(want_t) and np.array_equal(tr[k, :len(want_t)].cpu().numpy(), want_t), (rep, k)
    tp.close()
