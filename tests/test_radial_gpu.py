"""GPU parity of depthMapAndColorSetRadialCorrection (forward warp + raster-order in-place hole closing) against the
CPU oracle: bit-exact depth and colour, through the C-ABI export and the device-resident entry point."""
import numpy as np
import pytest

from livescan3d_amd import native, synth

pytestmark = pytest.mark.gpu


def _rig(depths, rgbs, intr_list):
    n = len(depths)
    wt = np.concatenate([synth.pack_pose(*synth.ring_pose(s, n)) for s in range(n)])
    return synth.Rig(depths, rgbs, np.concatenate(intr_list), wt, synth.DEFAULT_BOUNDS)


def _check(orc, rig, what):
    got_d, got_c = native.radial_correction(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr)
    want_d, want_c = orc.radial_correction(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr)
    assert np.array_equal(got_d, want_d), f"{what}: depth differs at {np.flatnonzero(got_d != want_d)[:8]}"
    assert np.array_equal(got_c, want_c), f"{what}: colour differs"
    return want_d


def test_scene_and_noise_frames(gpu, orc):
    for kind, n, w, h in [("scene", 2, 512, 424), ("noise", 3, 64, 48), ("scene", 1, 128, 96), ("noise", 1, 512, 424)]:
        rig = synth.make_rig(kind, n, w, h, seed=6)
        out = _check(orc, rig, f"{kind} {n}x{w}x{h}")
        assert not np.array_equal(out, rig.depth_maps)        # the correction really moves pixels (r2,r4,r6 = .09,-.27,.09)


def test_hole_closing_chains(gpu, orc):
    """Patterns whose result depends on the in-place raster order: 1-pixel gap rows / columns / diagonals (every filled
    pixel feeds its right and lower neighbours), random holes, depth steps around the 30 mm acceptance window."""
    rng = np.random.default_rng(4)
    zero_dist = [0.0, 0.0, 0.0]                                # r2=r4=r6=0: the warp is the identity, only the closing acts
    for (w, h) in [(64, 48), (61, 37), (512, 424), (24, 1100)]:
        yy, xx = np.mgrid[0:h, 0:w]
        base = (1500 + 2 * xx + 3 * yy).astype(np.int64)
        cases = {}
        d = base.copy(); d[::7, :] = 0; cases["gap rows"] = d
        d = base.copy(); d[:, ::5] = 0; cases["gap columns"] = d
        d = base.copy(); d[(xx + yy) % 6 == 0] = 0; cases["gap diagonals"] = d
        d = base.copy(); d[rng.random((h, w)) < 0.25] = 0; cases["random holes"] = d
        d = base + 26 * ((xx // 3) % 2) + 33 * ((yy // 4) % 2); d[rng.random((h, w)) < 0.15] = 0; cases["steps"] = d
        d = base.copy(); d[h // 3:h // 2, w // 4:3 * w // 4] = 0; d[::4, :] = np.where(rng.random((h // 4 + (h % 4 > 0), w)) < 0.5, 0, d[::4, :]); cases["blob"] = d
        for name, dd in cases.items():
            depth = np.clip(dd, 0, 65535).astype(np.uint16)
            rgb = synth.noise_frame(3, 0, 0, w, h)[1]
            intr = synth.kinect_intrinsics(w, h).copy()
            intr[4:7] = zero_dist
            rig = _rig([depth], [rgb], [intr])
            out = _check(orc, rig, f"{name} {w}x{h}").view(np.uint16)
            if name in ("gap rows", "gap columns", "gap diagonals"):
                assert (out.reshape(h, w)[2:-2, 2:-2] != 0).mean() > 0.9       # most gaps got closed (the (int) cast itself opens a few)


def test_warp_edge_cases(gpu, orc):
    w, h = 64, 48
    rgb = synth.noise_frame(5, 0, 0, w, h)[1]
    depth = synth.noise_frame(5, 0, 1, w, h)[0]
    # [0.5..] / [0.7..] / [3..]: the radial map folds inside the frame (1 - 3 r2 r = 0), so far more than four sources pile up
    # on the destinations along the fold -> the per-calibration candidate table overflows and the atomicMax path runs;
    # [0, 0, 0] is the identity (exactly one candidate everywhere), [-0.8..] expands (gaps, at most one candidate)
    for r in ([0.5, 0.0, 0.0], [0.7, 0.0, 0.0], [0.0, 0.0, 0.0], [-0.8, 0.3, 0.0], [3.0, -5.0, 9.0], [np.nan, 0, 0], [1e30, 0, 0]):
        intr = synth.kinect_intrinsics(w, h).copy()
        intr[4:7] = r
        _check(orc, _rig([depth], [rgb], [intr]), f"r={r}")
    intr = synth.kinect_intrinsics(w, h).copy()
    intr[2] = 0.0                                              # fx = 0: u = +-inf / NaN -> every destination rejected in x
    _check(orc, _rig([depth], [rgb], [intr]), "fx=0")
    _check(orc, _rig([np.zeros((h, w), np.uint16)], [rgb], [synth.kinect_intrinsics(w, h)]), "all invalid")
    # frames of one pixel, one row, one column -- alone and between ordinary frames (found by tests/test_fuzz_gpu.py: a 1 x 1 frame's colour
    # was read "one byte before the last pixel", i.e. in front of the frame, and came back 0)
    tiny = [(1, 1), (1, 7), (9, 1), (2, 2)]
    frames = [(np.full((hh, ww), 900 + 10 * k, np.uint16), np.full((hh, ww, 3), 40 + k, np.uint8)) for k, (ww, hh) in enumerate(tiny)]
    for k, (ww, hh) in enumerate(tiny):
        _check(orc, _rig([frames[k][0]], [frames[k][1]], [synth.kinect_intrinsics(ww, hh)]), f"{ww} x {hh} alone")
    _check(orc, _rig([depth] + [f[0] for f in frames] + [depth], [rgb] + [f[1] for f in frames] + [rgb],
                     [synth.kinect_intrinsics(w, h)] + [synth.kinect_intrinsics(ww, hh) for ww, hh in tiny] + [synth.kinect_intrinsics(w, h)]), "tiny frames between others")


def test_device_resident_batch_then_fusion(gpu, orc):
    """The per-tick order of LiveScanServer (KinectServer.cs:518-525 then :354-374): radial correction, then the merge
    call, all on HBM-resident ticks."""
    import torch
    T, N, w, h = 3, 2, 512, 424
    rigs = [synth.make_rig("scene", N, w, h, seed=9, tick=k, bounds=synth.CROP_BOUNDS) for k in range(T)]
    plan = native.FusionPlan(0, T, rigs[0].widths, rigs[0].heights)
    plan.set_params(rigs[0].intr, rigs[0].wt, rigs[0].bounds)
    depth = torch.from_numpy(np.stack([r.depth_maps.view(np.int16) for r in rigs])).cuda()
    rgb = torch.from_numpy(np.stack([r.depth_colors for r in rigs])).cuda()
    st = int(torch.cuda.current_stream().cuda_stream)
    plan.radial_correct(rigs[0].intr, depth.data_ptr(), rgb.data_ptr(), st)
    verts = torch.zeros((T, plan.capacity, 16), dtype=torch.uint8, device="cuda")
    off = torch.zeros((T, N + 1), dtype=torch.int32, device="cuda")
    plan.run(depth.data_ptr(), rgb.data_ptr(), verts.data_ptr(), off.data_ptr(), st)
    torch.cuda.synchronize()
    for k in range(T):
        want_d, want_c = orc.radial_correction(rigs[k].depth_maps, rigs[k].depth_colors, rigs[k].widths, rigs[k].heights, rigs[0].intr)
        assert np.array_equal(depth[k].cpu().numpy().view(np.uint8), want_d)
        assert np.array_equal(rgb[k].cpu().numpy(), want_c)
        want_v, _ = orc.generate_mesh_vertices(want_d, want_c, rigs[k].widths, rigs[k].heights, rigs[0].intr, rigs[0].wt, rigs[0].bounds)
        n = int(off[k, -1])
        assert verts[k, :n].cpu().numpy().tobytes() == want_v.tobytes()


def test_both_warp_paths_agree_with_the_oracle(gpu, orc, monkeypatch):
    """The per-calibration candidate table and the atomicMax path it falls back to (forced here) on the same frames."""
    rig = synth.make_rig("scene", 2, 512, 424, seed=12)
    want_d, want_c = orc.radial_correction(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr)
    for forced in ("0", "1"):
        monkeypatch.setenv("LSN_RADIAL_FORCE_ATOMIC", forced)
        got_d, got_c = native.radial_correction(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr)
        assert np.array_equal(np.asarray(got_d).view(np.uint8).ravel(), np.asarray(want_d).view(np.uint8).ravel()), forced
        assert np.array_equal(np.asarray(got_c).ravel(), np.asarray(want_c).ravel()), forced


@pytest.mark.parametrize("env", [{}, {"LSN_RADIAL_TINY_LISTS": "1"}, {"LSN_RADIAL_CLOSE": "wavefront"}, {"LSN_RADIAL_BAND_ROWS": "5"},
                                 {"LSN_RADIAL_FORCE_ATOMIC": "1", "LSN_RADIAL_TINY_LISTS": "1"}],
                         ids=["two-pass", "two-pass-sweeps", "wavefront", "five-row-bands", "atomic-warp-sweeps"])
def test_hole_closing_paths_agree_with_the_oracle(gpu, orc, monkeypatch, env):
    """The two-pass hole closing (bands of rows warped and closed in LDS + per-frame re-evaluation rounds), the full sweeps a frame
    falls back to when a round list outgrows LDS (forced with tiny lists), odd band heights, the atomicMax warp in front of it, and
    the wavefront kernel alone: scene frames (thousands of fills, chains of fills feeding fills), a ragged rig (the pixel-by-pixel
    path) and a batch through the device-resident entry points, in place and out of place."""
    import torch
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for rig in (synth.make_rig("scene", 3, 512, 424, seed=21), synth.make_rig("scene", 2, 250, 120, seed=22), synth.make_rig("noise", 2, 128, 96, seed=23)):
        want_d, want_c = orc.radial_correction(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr)
        got_d, got_c = native.radial_correction(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr)
        assert np.array_equal(np.asarray(got_d).view(np.uint8).ravel(), np.asarray(want_d).view(np.uint8).ravel())
        assert np.array_equal(np.asarray(got_c).ravel(), np.asarray(want_c).ravel())
    T, N, w, h = 5, 4, 256, 212
    rigs = [synth.make_rig("scene", N, w, h, seed=24, tick=k) for k in range(T)]
    plan = native.FusionPlan(0, T, rigs[0].widths, rigs[0].heights)
    depth = torch.from_numpy(np.stack([r.depth_maps.view(np.int16) for r in rigs])).cuda()
    rgb = torch.from_numpy(np.stack([r.depth_colors for r in rigs])).cuda()
    plan.radial_correct(rigs[0].intr, depth.data_ptr(), rgb.data_ptr(), int(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    got_d, got_c = depth.cpu().numpy().view(np.uint8), rgb.cpu().numpy()
    depth2 = torch.from_numpy(np.stack([r.depth_maps.view(np.int16) for r in rigs])).cuda()
    rgb2 = torch.from_numpy(np.stack([r.depth_colors for r in rigs])).cuda()
    out_d, out_c = torch.zeros_like(depth2), torch.zeros_like(rgb2)
    plan.radial_correct_to(rigs[0].intr, depth2.data_ptr(), rgb2.data_ptr(), out_d.data_ptr(), out_c.data_ptr(), int(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert np.array_equal(depth2.cpu().numpy(), np.stack([r.depth_maps.view(np.int16) for r in rigs])), "out of place: the input was touched"
    got2_d, got2_c = out_d.cpu().numpy().view(np.uint8), out_c.cpu().numpy()
    for k in range(T):
        want_d, want_c = orc.radial_correction(rigs[k].depth_maps, rigs[k].depth_colors, rigs[k].widths, rigs[k].heights, rigs[0].intr)
        assert np.array_equal(got_d[k], np.asarray(want_d).view(np.uint8).ravel()), f"tick {k}: depth"
        assert np.array_equal(got_c[k], np.asarray(want_c).ravel()), f"tick {k}: colours"
        assert np.array_equal(got2_d[k], np.asarray(want_d).view(np.uint8).ravel()), f"tick {k}: depth (out of place)"
        assert np.array_equal(got2_c[k], np.asarray(want_c).ravel()), f"tick {k}: colours (out of place)"


def test_more_frames_than_compute_units(gpu, orc):
    """A batch with more sensor-frames than the GPU has compute units (the second pass runs one workgroup per frame), and a
    1024-wide frame (eight-row bands).  Every frame against the oracle."""
    import torch
    T, N, w, h = 33, 8, 512, 424                                 # 264 frames
    rigs = [synth.make_rig("noise" if k % 3 else "scene", N, w, h, seed=14, tick=k) for k in range(T)]
    plan = native.FusionPlan(0, T, rigs[0].widths, rigs[0].heights)
    depth = torch.from_numpy(np.stack([r.depth_maps.view(np.int16) for r in rigs])).cuda()
    rgb = torch.from_numpy(np.stack([r.depth_colors for r in rigs])).cuda()
    plan.radial_correct(rigs[0].intr, depth.data_ptr(), rgb.data_ptr(), int(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    got_d, got_c = depth.cpu().numpy().view(np.uint8), rgb.cpu().numpy()
    for k in range(T):
        want_d, want_c = orc.radial_correction(rigs[k].depth_maps, rigs[k].depth_colors, rigs[k].widths, rigs[k].heights, rigs[0].intr)
        assert np.array_equal(got_d[k], want_d), f"tick {k}: depth"
        assert np.array_equal(got_c[k], want_c), f"tick {k}: colour"
    _check(orc, synth.make_rig("noise", 1, 1024, 1024, seed=15), "1024x1024")


def test_one_call_per_tick_equals_the_two_exports(gpu, orc):
    """lsnCorrectAndGenerateMesh (one upload) against depthMapAndColorSetRadialCorrection followed by generateMeshFromDepthMaps
    (KinectServer.cs:518-525 then :354-374), and both against the oracle's chain: corrected maps, vertices and triangles bit for bit."""
    for rig in (synth.make_rig("scene", 8, 512, 424, seed=31, bounds=synth.CROP_BOUNDS), synth.make_rig("scene", 2, 250, 120, seed=32),
                synth.make_rig("noise", 3, 64, 48, seed=33)):
        d2, c2 = native.radial_correction(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr)
        v2, t2 = native.generate_mesh_from_depth_maps(d2, c2, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
        for write_back in (True, False):
            v1, t1, d1, c1 = native.correct_and_generate_mesh(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds,
                                                              write_back=write_back)
            assert v1.tobytes() == v2.tobytes() and np.array_equal(t1, t2)
            if write_back:
                assert np.array_equal(d1, d2) and np.array_equal(c1, c2)
            else:
                assert np.array_equal(d1, np.ascontiguousarray(rig.depth_maps).view(np.uint8).ravel()) and np.array_equal(c1, np.asarray(rig.depth_colors).ravel())
        want_d, want_c = orc.radial_correction(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr)
        want_v, _, want_t = orc.generate_mesh(want_d, want_c, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
        assert np.array_equal(d2, np.asarray(want_d).view(np.uint8).ravel()) and v2.tobytes() == want_v.tobytes() and np.array_equal(t2, want_t)


def test_partly_overlapping_buffers_are_refused_and_route_switches_leave_no_stale_counts(gpu, orc, monkeypatch):
    """Out of place the bands warp straight from the input: an output that overlaps the input IN PART would be overwritten by one band
    while another still reads it, so the entry point refuses it (identical pointers = in place, disjoint = out of place).  And a call that
    takes the wavefront closing after the band kernel must not leave work-list counts behind for the next two-pass call."""
    import torch
    T, N, w, h = 2, 2, 256, 212
    rigs = [synth.make_rig("scene", N, w, h, seed=31, tick=k) for k in range(T)]
    plan = native.FusionPlan(0, T, rigs[0].widths, rigs[0].heights)
    st = int(torch.cuda.current_stream().cuda_stream)
    npix = T * N * w * h
    big_d = torch.zeros(2 * npix, dtype=torch.int16, device="cuda")
    big_c = torch.zeros(6 * npix, dtype=torch.uint8, device="cuda")
    big_d[:npix] = torch.from_numpy(np.stack([r.depth_maps.view(np.int16) for r in rigs]).ravel()).cuda()
    big_c[:3 * npix] = torch.from_numpy(np.stack([r.depth_colors for r in rigs]).ravel()).cuda()
    with pytest.raises(native.NativeUtilsError, match="overlap"):
        plan.radial_correct_to(rigs[0].intr, big_d.data_ptr(), big_c.data_ptr(), big_d.data_ptr() + 2 * (npix // 2), big_c.data_ptr() + 3 * npix, st)
    with pytest.raises(native.NativeUtilsError, match="overlap"):
        plan.radial_correct_to(rigs[0].intr, big_d.data_ptr(), big_c.data_ptr(), big_d.data_ptr() + 2 * npix, big_c.data_ptr() + 16, st)
    want = [orc.radial_correction(r.depth_maps, r.depth_colors, r.widths, r.heights, rigs[0].intr) for r in rigs]

    def run_and_check(tag):
        plan.radial_correct_to(rigs[0].intr, big_d.data_ptr(), big_c.data_ptr(), big_d.data_ptr() + 2 * npix, big_c.data_ptr() + 3 * npix, st)
        torch.cuda.synchronize()
        got_d = big_d[npix:].cpu().numpy().view(np.uint8).reshape(T, -1)
        got_c = big_c[3 * npix:].cpu().numpy().reshape(T, -1)
        for k in range(T):
            assert np.array_equal(got_d[k], np.asarray(want[k][0]).view(np.uint8).ravel()), f"{tag}: tick {k} depth"
            assert np.array_equal(got_c[k], np.asarray(want[k][1]).ravel()), f"{tag}: tick {k} colours"

    run_and_check("two-pass")
    monkeypatch.setenv("LSN_RADIAL_CLOSE", "wavefront")
    run_and_check("wavefront")
    monkeypatch.delenv("LSN_RADIAL_CLOSE")
    run_and_check("two-pass after the wavefront route")


@pytest.mark.parametrize("ticks,env", [(2, {}), (2, {"LSN_RADIAL_TINY_LISTS": "1"}), (80, {}), (80, {"LSN_RADIAL_TINY_LISTS": "1"})])
def test_closing_chain_leaves_its_counters_cleared(gpu, orc, monkeypatch, ticks, env):
    """The next call skips the memset of the closing chain's work counters when the previous chain has run to its end on the same stream -- valid
    only if the chain really leaves all three counter blocks at zero.  Both routes (<= 128 frames: two grid-wide rounds + per-frame kernel;
    more: the per-frame kernel alone) with ordinary lists and with lists that overflow at once (full sweeps): no counter is left after a
    call, the second call -- which skipped the memset -- is as right as the first, and a call on ANOTHER stream (which clears the counters
    again: they might still be counting behind the first stream) is right as well."""
    import torch
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    N, w, h = 2, 128, 96
    rigs = [synth.make_rig("scene", N, w, h, seed=13, tick=k % 4) for k in range(ticks)]
    plan = native.FusionPlan(0, ticks, rigs[0].widths, rigs[0].heights)
    src_d = torch.from_numpy(np.stack([r.depth_maps.view(np.int16) for r in rigs])).cuda()
    src_c = torch.from_numpy(np.stack([r.depth_colors for r in rigs])).cuda()
    want = [orc.radial_correction(r.depth_maps, r.depth_colors, r.widths, r.heights, rigs[0].intr) for r in rigs[:4]]
    other = torch.cuda.Stream()
    st0 = int(torch.cuda.current_stream().cuda_stream)
    for rep, st in enumerate((st0, st0, int(other.cuda_stream), st0)):
        d, c = src_d.clone(), src_c.clone()
        torch.cuda.synchronize()
        plan.radial_correct(rigs[0].intr, d.data_ptr(), c.data_ptr(), st)
        assert plan.radial_counters_left(st) == 0, f"call {rep}: the chain left counters behind"
        for k in (0, 1, ticks - 1):
            assert np.array_equal(d[k].cpu().numpy().view(np.uint8), np.asarray(want[k % 4][0]).view(np.uint8).ravel()), f"call {rep}: tick {k} depth"
            assert np.array_equal(c[k].cpu().numpy(), np.asarray(want[k % 4][1]).ravel()), f"call {rep}: tick {k} colours"


def test_calls_on_two_streams_back_to_back_share_the_plans_scratch_safely(gpu, orc):
    """One plan, two streams, no synchronisation in between: the second call clears and refills the scratch (tables, lists, hole bitmap,
    counters) the first one's chain is still working on -- unless it waits for that chain's end, which radial_correct makes it do (an event
    behind every chain).  Both calls, and a third one back on the first stream, must be right; a batch big enough that the first chain is
    still running when the second call is issued."""
    import torch
    N, w, h, ticks = 4, 256, 212, 48
    rigs = [synth.make_rig("scene", N, w, h, seed=17, tick=k % 3) for k in range(ticks)]
    plan = native.FusionPlan(0, ticks, rigs[0].widths, rigs[0].heights)
    src_d = torch.from_numpy(np.stack([r.depth_maps.view(np.int16) for r in rigs])).cuda()
    src_c = torch.from_numpy(np.stack([r.depth_colors for r in rigs])).cuda()
    want = [orc.radial_correction(r.depth_maps, r.depth_colors, r.widths, r.heights, rigs[0].intr) for r in rigs[:3]]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(3):
        outs = [(src_d.clone(), src_c.clone()) for _ in range(3)]
        torch.cuda.synchronize()
        for (d, c), st in zip(outs, (s1, s2, s1)):
            plan.radial_correct(rigs[0].intr, d.data_ptr(), c.data_ptr(), int(st.cuda_stream))
        torch.cuda.synchronize()
        assert plan.radial_counters_left(int(s1.cuda_stream)) == 0
        for i, (d, c) in enumerate(outs):
            for k in (0, 1, 2, ticks - 1):
                assert np.array_equal(d[k].cpu().numpy().view(np.uint8), np.asarray(want[k % 3][0]).view(np.uint8).ravel()), f"round {rep} call {i}: tick {k} depth"
                assert np.array_equal(c[k].cpu().numpy(), np.asarray(want[k % 3][1]).ravel()), f"round {rep} call {i}: tick {k} colours"
    plan.close()
