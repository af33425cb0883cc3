"""The per-pixel depth thresholds behind the count pass (lsnFusionThresholds / thresh_kernel / count_thr_kernel).

1. The table itself against an exhaustive numpy-float32 evaluation of the reference's per-pixel arithmetic
   (src/NativeUtils/depthprocessing.cpp:144-163) over ALL 65 535 depth values, for sampled pixels of several rigs:
   a pixel's entry {lo, count} must describe its set of surviving depths exactly (or be flagged lo = 0).
2. Runs 1 (arithmetic count), 2 (thresholds just built), 3 and the streamed variant all equal the oracle bit for bit,
   across a parameter change, ragged sizes, degenerate boxes and non-finite calibrations."""
import numpy as np
import pytest

from livescan3d_amd import native, synth

pytestmark = pytest.mark.gpu


def survivors_all_depths(x, y, intr, wt, bounds):
    """keep[d] for d = 0..65535 at pixel (x, y): numpy float32, one rounding per operation, the reference's order."""
    f = np.float32
    cx, cy, fx, fy = [f(v) for v in intr[:4]]
    t = [f(v) for v in wt[:3]]
    R = np.asarray(wt[3:12], dtype=np.float32).reshape(3, 3)
    d = np.arange(65536, dtype=np.float32)
    with np.errstate(all="ignore"):
        Z = d / f(1000.0)
        X = (f(x) - cx) / fx
        Y = (cy - f(y)) / fy
        X = X * Z
        Y = Y * Z
        X = X + t[0]
        Y = Y + t[1]
        Z = Z + t[2]
        ox = (X * R[0, 0] + Y * R[0, 1]) + Z * R[0, 2]
        oy = (X * R[1, 0] + Y * R[1, 1]) + Z * R[1, 2]
        oz = (X * R[2, 0] + Y * R[2, 1]) + Z * R[2, 2]
        b = np.asarray(bounds, dtype=np.float32)
        rejected = (ox < b[0]) | (ox > b[3]) | (oy < b[1]) | (oy > b[4]) | (oz < b[2]) | (oz > b[5])
    keep = ~rejected
    keep[0] = False
    return keep


def _plan(rig, ticks=1):
    from livescan3d_amd.fusion import DeviceFusion
    fus = DeviceFusion(ticks, rig.widths, rig.heights)
    fus.set_params(rig.intr, rig.wt, rig.bounds)
    return fus


@pytest.mark.parametrize("kind,n,w,h,bounds", [
    ("scene", 8, 512, 424, synth.CROP_BOUNDS),
    ("scene", 3, 512, 424, synth.DEFAULT_BOUNDS),
    ("scene", 2, 100, 75, [-0.3, -0.2, -0.4, 0.25, 0.3, 0.1]),
    ("scene", 2, 64, 48, [float("-inf"), -1, -1, float("inf"), 1, 1]),
])
def test_table_is_exact_for_every_depth(gpu, kind, n, w, h, bounds):
    rig = synth.make_rig(kind, n, w, h, seed=31, bounds=bounds, perturb=True)
    fus = _plan(rig)
    table, ms = fus.plan.thresholds(fus.capacity)
    assert table is not None and ms > 0
    lo, cnt = (table & 0xFFFF).astype(np.int64), (table >> 16).astype(np.int64)
    assert (lo != 0).all(), "finite calibration: every pixel must have an interval"
    rng = np.random.default_rng(7)
    base = 0
    checked = partial = 0
    for i in range(n):
        ww, hh = int(rig.widths[i]), int(rig.heights[i])
        picks = [(0, 0), (ww - 1, 0), (0, hh - 1), (ww - 1, hh - 1), (ww // 2, hh // 2)] + \
                [(int(rng.integers(ww)), int(rng.integers(hh))) for _ in range(40)]
        for (x, y) in picks:
            keep = survivors_all_depths(x, y, rig.intr[7 * i:7 * i + 7], rig.wt[12 * i:12 * i + 12], rig.bounds)
            p = base + y * ww + x
            want = np.zeros(65536, dtype=bool)
            want[lo[p]:lo[p] + cnt[p]] = True
            assert np.array_equal(keep, want), (i, x, y, int(lo[p]), int(cnt[p]), np.flatnonzero(keep)[[0, -1]] if keep.any() else None)
            checked += 1
            partial += 0 < cnt[p] < 65535
        base += ww * hh
    assert checked >= 45 * n and partial > 0


def _oracle(orc, rig):
    return orc.generate_mesh_vertices(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)


def _run_and_compare(orc, fus, rigs, what, streamed=False):
    import torch
    T = len(rigs)
    depth = torch.from_numpy(np.stack([r.depth_maps.view(np.int16) for r in rigs])).cuda()
    rgb = torch.from_numpy(np.stack([r.depth_colors for r in rigs])).cuda()
    if streamed:
        st = int(torch.cuda.current_stream().cuda_stream)
        fus.plan.run_streamed(depth.data_ptr(), rgb.data_ptr(), fus.vertices.data_ptr(), fus.offsets.data_ptr(), depth.data_ptr(), st)
        fus.plan.run_streamed(depth.data_ptr(), rgb.data_ptr(), fus.vertices.data_ptr(), fus.offsets.data_ptr(), 0, st)   # offsets counted ahead by the call before
        v, o = fus.vertices, fus.offsets
    else:
        v, o = fus.run(depth, rgb)
    torch.cuda.synchronize()
    o = o.cpu().numpy()
    for k in range(T):
        want, counts = _oracle(orc, rigs[k])
        assert list(np.diff(o[k])) == list(counts), f"{what}: per-sensor counts of tick {k}"
        got = v[k, :len(want)].cpu().numpy().view(native.VERTEX_DTYPE).reshape(-1)
        assert got.tobytes() == want.tobytes(), f"{what}: vertices of tick {k}"


def test_runs_before_and_after_the_table_and_across_a_parameter_change(gpu, orc):
    from livescan3d_amd.fusion import DeviceFusion
    T, S, w, h = 11, 3, 512, 424                                  # 11 ticks: one full group of 8 and a partial one
    mk = lambda seed, **kw: [synth.make_rig("noise" if k % 2 else "scene", S, w, h, seed=seed, tick=k, bounds=synth.CROP_BOUNDS, **kw) for k in range(T)]
    rigs = mk(41)
    fus = DeviceFusion(T, [w] * S, [h] * S)
    fus.set_params(rigs[0].intr, rigs[0].wt, rigs[0].bounds)
    for r in rigs:
        r.intr, r.wt = rigs[0].intr, rigs[0].wt
    for i in range(3):                                             # 1: arithmetic count, 2: table built, 3: table reused
        _run_and_compare(orc, fus, rigs, f"run {i + 1}")
    _run_and_compare(orc, fus, rigs, "streamed", streamed=True)
    # recalibration: other poses and another box -> the table is stale, rebuilt on the second run
    rigs2 = mk(43, perturb=True)
    b2 = [-0.9, -0.7, -1.1, 1.2, 0.8, 0.6]
    for r in rigs2:
        r.intr, r.wt, r.bounds = rigs2[0].intr, rigs2[0].wt, np.asarray(b2, dtype=np.float32)
    fus.set_params(rigs2[0].intr, rigs2[0].wt, b2)
    for i in range(3):
        _run_and_compare(orc, fus, rigs2, f"after recalibration, run {i + 1}")
    _run_and_compare(orc, fus, rigs2, "after recalibration, streamed", streamed=True)


def test_ragged_sizes_and_hostile_calibrations(gpu, orc):
    from livescan3d_amd.fusion import DeviceFusion
    widths, heights = [61, 512, 33], [47, 424, 90]                 # rows that split lanes, frames that start unaligned
    rig = synth.make_rig("noise", 3, 64, 48, seed=5)               # template for poses
    rng = np.random.default_rng(3)
    frames_d = [rng.integers(0, 6000, size=(hh, ww)).astype(np.uint16) for ww, hh in zip(widths, heights)]
    for fd in frames_d:
        fd[rng.random(fd.shape) < 0.1] = 0
        fd.flat[:3] = (65535, 1, 0)
    frames_c = [rng.integers(0, 256, size=(hh, ww, 3)).astype(np.uint8) for ww, hh in zip(widths, heights)]
    intr = np.concatenate([synth.kinect_intrinsics(ww, hh) for ww, hh in zip(widths, heights)])
    wt = np.concatenate([synth.pack_pose(*synth.ring_pose(i, 3)) for i in range(3)])
    cases = {"plain": (intr, wt, synth.CROP_BOUNDS)}
    w_nan = wt.copy(); w_nan[12 + 4] = np.nan                       # NaN in sensor 1's rotation
    cases["nan pose"] = (intr, w_nan, synth.CROP_BOUNDS)
    w_inf = wt.copy(); w_inf[2] = np.inf                            # infinite translation on sensor 0
    cases["inf pose"] = (intr, w_inf, synth.CROP_BOUNDS)
    i0 = intr.copy(); i0[7 * 2 + 2] = 0.0                           # fx = 0 on sensor 2
    cases["fx = 0"] = (i0, wt, synth.CROP_BOUNDS)
    cases["inverted box"] = (intr, wt, [1, 1, 1, -1, -1, -1])
    cases["plane box"] = (intr, wt, [-5, -0.25, -5, 5, -0.25, 5])
    cases["nan bound"] = (intr, wt, [np.nan, -1, -1, 1, 1, 1])
    cases["huge pose"] = (intr, wt * np.float32(1e37), synth.CROP_BOUNDS)
    fus = DeviceFusion(1, widths, heights)
    for name, (ci, cw, cb) in cases.items():
        r = synth.Rig(frames_d, frames_c, ci, cw, cb)
        fus.set_params(r.intr, r.wt, r.bounds)
        for i in range(3):
            _run_and_compare(orc, fus, [r], f"{name}, run {i + 1}")


def test_fuzzed_calibrations_at_every_interval_boundary(gpu, orc):
    """Random (and some hostile) calibrations; the depth frames are built FROM the table so that every pixel sits exactly on
    and next to both ends of its interval (lo-1, lo, lo+n-1, lo+n): the threshold count (plan A, table built) and the
    arithmetic count (plan B, first run after SetParams) must agree on every tick, and both must equal the oracle."""
    import torch
    from livescan3d_amd.fusion import DeviceFusion
    rng = np.random.default_rng(2026)
    widths, heights = [64, 40, 57], [48, 56, 31]
    n = len(widths)
    P = [w * h for w, h in zip(widths, heights)]
    T = 4

    def rot(rng):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        a, b, c, d = q
        return np.array([[a*a+b*b-c*c-d*d, 2*(b*c-a*d), 2*(b*d+a*c)], [2*(b*c+a*d), a*a-b*b+c*c-d*d, 2*(c*d-a*b)],
                         [2*(b*d-a*c), 2*(c*d+a*b), a*a-b*b-c*c+d*d]], dtype=np.float32)

    plan_a = DeviceFusion(T, widths, heights)
    plan_b = DeviceFusion(T, widths, heights)
    flagged_seen = 0
    for trial in range(24):
        intr = np.concatenate([synth.kinect_intrinsics(w, h) * np.float32(rng.uniform(0.8, 1.2)) for w, h in zip(widths, heights)]).astype(np.float32)
        wt = np.concatenate([np.concatenate([rng.normal(scale=1.5, size=3).astype(np.float32), rot(rng).ravel()]) for _ in range(n)]).astype(np.float32)
        c = rng.normal(scale=1.0, size=3)
        half = rng.uniform(0.05, 3.0, size=3)
        bounds = np.concatenate([c - half, c + half]).astype(np.float32)
        if trial % 6 == 5:
            wt[12 * (trial % n) + 3 + trial % 9] = [np.nan, np.inf, 1e30, -np.inf][(trial // 6) % 4]      # a hostile entry in one sensor's pose
        if trial % 8 == 7:
            bounds[[0, 3]] = bounds[[3, 0]]                                                   # inverted X range
        plan_a.set_params(intr, wt, bounds)
        table, _ = plan_a.plan.thresholds()
        lo, cnt = (table & 0xFFFF).astype(np.int64), (table >> 16).astype(np.int64)
        flagged_seen += int((lo == 0).any())
        # ticks: lo-1, lo, lo+n-1, lo+n, clamped to u16; flagged / empty pixels get random depths
        rnd = rng.integers(0, 65536, size=(T, lo.size))
        d = np.stack([lo - 1, lo, lo + cnt - 1, lo + cnt])
        d = np.where((lo == 0)[None, :] | (cnt == 0)[None, :], rnd, d)
        d = np.clip(d, 0, 65535).astype(np.uint16)
        rgb = rng.integers(0, 256, size=(T, 3 * lo.size)).astype(np.uint8)
        depth_t, rgb_t = torch.from_numpy(d.view(np.int16)).cuda(), torch.from_numpy(rgb).cuda()
        va, oa = plan_a.run(depth_t, rgb_t)                        # table in use
        plan_b.set_params(intr, wt, bounds)
        vb, ob = plan_b.run(depth_t, rgb_t)                        # first run with these parameters: arithmetic count
        torch.cuda.synchronize()
        assert torch.equal(oa, ob), f"trial {trial}: per-sensor offsets differ between the two count passes"
        oa_h = oa.cpu().numpy()
        for k in range(T):
            nv = int(oa_h[k, -1])
            assert torch.equal(va[k, :nv], vb[k, :nv]), f"trial {trial} tick {k}"
        if trial % 4 == 0:
            pos = np.concatenate([[0], np.cumsum(P)])
            for k in range(T):
                r = synth.Rig([d[k, pos[i]:pos[i + 1]].reshape(heights[i], widths[i]) for i in range(n)],
                              [rgb[k, 3 * pos[i]:3 * pos[i + 1]].reshape(heights[i], widths[i], 3) for i in range(n)], intr, wt, bounds)
                want, counts = _oracle(orc, r)
                assert list(np.diff(oa_h[k])) == list(counts), f"trial {trial} tick {k}: counts vs oracle"
                got = va[k, :len(want)].cpu().numpy().view(native.VERTEX_DTYPE).reshape(-1)
                assert got.tobytes() == want.tobytes(), f"trial {trial} tick {k}: vertices vs oracle"
    assert flagged_seen >= 3
