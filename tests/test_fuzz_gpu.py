"""Seeded random rigs through the C-ABI against the oracle: sizes from 1 x 1 up, 1-6 sensors of different sizes, widths on both sides of
every multiple of 8, frames of noise, of constant depth, with stripes of holes; random poses, crop boxes (including empty and inverted
ones) and lens coefficients.  The merge call (vertices + triangles), the single-sensor call, the radial export and the tick as one call
must return what the oracle returns, bit for bit.  (The fixed cases of test_fusion_gpu.py / test_radial_gpu.py are the edge cases that
were thought of; these are the ones that were not.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

from livescan3d_amd import native, synth

pytestmark = pytest.mark.gpu

# $LSN_FUZZ_SCALE=k: k times as many cases of every kind (a campaign run; the suite's default is sized for seconds)
SCALE = max(1, int(os.environ.get("LSN_FUZZ_SCALE", "1")))
N_CASES = 160 * SCALE


def _random_rig(rng):
    n = int(rng.integers(1, 7))
    depths, rgbs, intr, wt = [], [], [], []
    for s in range(n):
        shape = int(rng.integers(0, 5))
        if shape == 0:
            w, h = int(rng.integers(1, 12)), int(rng.integers(1, 8))          # smaller than any tile, any stencil
        elif shape == 1:
            w, h = 8 * int(rng.integers(1, 20)) + int(rng.integers(-1, 2)), int(rng.integers(3, 40))   # either side of a multiple of 8
        elif shape == 2:
            w, h = 8 * int(rng.integers(2, 24)), int(rng.integers(4, 60))     # the wide-load paths
        elif shape == 3:
            w, h = int(rng.integers(200, 700)), int(rng.integers(1, 4))       # long and flat: tiles span rows
        else:
            w, h = int(rng.integers(16, 130)), int(rng.integers(16, 100))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            d, c = synth.noise_frame(int(rng.integers(1, 1000)), 0, s, w, h)
        elif kind == 1:
            d = np.full((h, w), int(rng.integers(400, 4000)), np.uint16)
            c = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        elif kind == 2:                                                       # a smooth surface with stripes and dots of holes
            yy, xx = np.mgrid[0:h, 0:w]
            d = (1200 + 3 * xx + 2 * yy + rng.integers(0, 6, size=(h, w))).astype(np.uint16)
            d[:, :: int(rng.integers(2, 9))] = 0
            d[rng.random((h, w)) < 0.05] = 0
            c = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        else:                                                                 # mostly holes
            d = np.zeros((h, w), np.uint16)
            m = rng.random((h, w)) < 0.3
            d[m] = rng.integers(500, 3000, size=int(m.sum())).astype(np.uint16)
            c = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        depths.append(np.ascontiguousarray(d)); rgbs.append(np.ascontiguousarray(c))
        k = synth.kinect_intrinsics(w, h).copy()
        k[:4] *= (1.0 + rng.uniform(-0.05, 0.05, size=4)).astype(np.float32)
        k[4:7] = rng.uniform(-0.3, 0.3, size=3).astype(np.float32)            # r2, r4, r6: the radial export's lens
        if rng.random() < 0.15:                                               # a lens that folds the frame onto itself: destinations with many sources
            k[4:7] = rng.uniform(-4.0, 4.0, size=3).astype(np.float32)
        intr.append(k.astype(np.float32))
        R, t = synth.ring_pose(s, n)
        t = (np.asarray(t, np.float64) + rng.uniform(-0.2, 0.2, size=3)).astype(np.float32)
        wt.append(synth.pack_pose(R, t))
    if rng.random() < 0.6:                                                    # a box most of the scene is in
        lo = rng.uniform(-6.0, -2.5, size=3)
        hi = rng.uniform(2.5, 6.0, size=3)
    else:                                                                     # a tight one, sometimes empty or inverted on an axis
        lo = rng.uniform(-2.0, 0.5, size=3)
        hi = lo + rng.uniform(-0.2, 4.0, size=3)
    bounds = np.concatenate([lo, hi]).astype(np.float32)
    return synth.Rig(depths, rgbs, np.concatenate(intr), np.concatenate(wt), bounds)


def _same_mesh(got_v, got_t, want_v, want_t, what):
    assert len(got_v) == len(want_v), f"{what}: {len(got_v)} vertices, oracle {len(want_v)}"
    assert got_v.tobytes() == want_v.tobytes(), f"{what}: vertices differ"
    assert np.array_equal(np.asarray(got_t).reshape(-1, 3), np.asarray(want_t).reshape(-1, 3)), f"{what}: triangles differ"


def _random_large_rig(rng):
    """2-8 sensors of 0.2-1.8 MB each (widths multiples of 8 or not): the sizes at which a call's frames go up in several groups."""
    n = int(rng.integers(2, 9))
    depths, rgbs, intr, wt = [], [], [], []
    for s in range(n):
        w, h = int(rng.integers(300, 700)), int(rng.integers(150, 520))
        if rng.random() < 0.6:
            w &= ~7
        d, c = synth.scene_frame(int(rng.integers(1, 100)), 0, s, n, w, h) if rng.random() < 0.7 else synth.noise_frame(int(rng.integers(1, 100)), 0, s, w, h)
        depths.append(d); rgbs.append(c)
        k = synth.kinect_intrinsics(w, h).copy()
        k[4:7] = rng.uniform(-0.2, 0.2, size=3).astype(np.float32)
        intr.append(k)
        wt.append(synth.pack_pose(*synth.ring_pose(s, n)))
    return synth.Rig(depths, rgbs, np.concatenate(intr), np.concatenate(wt), synth.CROP_BOUNDS)


N_LARGE = 10 * SCALE


@pytest.mark.parametrize("seed", range(N_CASES + N_LARGE))
def test_random_rig_matches_the_oracle(gpu, orc, seed):
    rng = np.random.default_rng(1000 + seed)
    rig = _random_rig(rng) if seed < N_CASES else _random_large_rig(rng)
    what = f"seed {seed}: sizes {list(zip(rig.widths.tolist(), rig.heights.tolist()))} bounds {rig.bounds.tolist()}"
    # the merge call
    want_v, _, want_t = orc.generate_mesh(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    got_v, got_t = native.generate_mesh_from_depth_maps(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    _same_mesh(got_v, got_t, want_v, want_t, what + " [merge]")
    # the stream / the file of that same mesh, rebuilt from what the call left on the device(s)
    assert native.last_mesh_transfer_frame() == orc.transfer_frame(want_v, want_t), what + " [SendFrame stream of the merge call's mesh]"
    assert native.last_mesh_ply() == orc.ply_binary(want_v, want_t), what + " [PLY of the merge call's mesh]"
    # one sensor alone
    i = int(rng.integers(0, len(rig.widths)))
    sv = native.generate_vertices_from_depth_map(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds, i)
    allv, counts = orc.generate_mesh_vertices(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    e = np.concatenate([[0], np.cumsum(counts)])
    assert sv.tobytes() == allv[e[i]:e[i + 1]].tobytes(), what + f" [sensor {i} alone]"
    # the radial export, then the merge call on what it left: the reference's two calls of a tick
    cd, cc = orc.radial_correction(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr)
    cd = np.ascontiguousarray(np.asarray(cd)).view(np.uint8).ravel()
    cc = np.ascontiguousarray(np.asarray(cc)).ravel()
    gd, gc = native.radial_correction(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr)
    assert np.asarray(gd).view(np.uint8).ravel().tobytes() == cd.tobytes(), what + " [radial depth]"
    assert np.asarray(gc).ravel().tobytes() == cc.tobytes(), what + " [radial colours]"
    want_v2, _, want_t2 = orc.generate_mesh(cd, cc, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    got_v2, got_t2 = native.generate_mesh_from_depth_maps(gd, gc, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    _same_mesh(got_v2, got_t2, want_v2, want_t2, what + " [merge after radial]")
    # ... and the same tick as ONE call (correction + merge with a single upload), with and without the corrected maps handed back
    for write_back in (True, False):
        v3, t3, d3, c3 = native.correct_and_generate_mesh(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds,
                                                           write_back=write_back)
        _same_mesh(v3, t3, want_v2, want_t2, what + f" [tick as one call, write_back={write_back}]")
        assert native.last_mesh_transfer_frame() == orc.transfer_frame(want_v2, want_t2), what + " [SendFrame stream of the one call's mesh]"
        if write_back:
            assert np.asarray(d3).view(np.uint8).ravel().tobytes() == cd.tobytes() and np.asarray(c3).ravel().tobytes() == cc.tobytes(), what + " [maps of the one call]"


@pytest.mark.parametrize("seed", range(24 * SCALE))
def test_random_ticks_device_resident(gpu, orc, seed):
    """lsnFusionRadialCorrectTo -> lsnFusionRunMesh on 1-5 DIFFERENT ticks of a random rig (the host exports above run one tick per
    call): every tick's corrected maps, cloud, offsets and triangles against the oracle."""
    import torch
    rng = np.random.default_rng(5000 + seed)
    T = int(rng.integers(1, 6))                                 # (one tick: the plan takes the single pass by itself)
    rigs = [_random_rig(np.random.default_rng(7000 + seed))]
    sizes = list(zip(rigs[0].widths.tolist(), rigs[0].heights.tolist()))
    for k in range(1, T):                                    # the same calibration and sizes, other frames
        depths, rgbs = [], []
        for s, (w, h) in enumerate(sizes):
            d, c = synth.noise_frame(int(rng.integers(1, 1000)), k, s, w, h)
            if rng.random() < 0.5:
                yy, xx = np.mgrid[0:h, 0:w]
                d = (1000 + 2 * xx + 3 * yy).astype(np.uint16)
                d[rng.random((h, w)) < 0.08] = 0
            depths.append(d); rgbs.append(c)
        rigs.append(synth.Rig(depths, rgbs, rigs[0].intr, rigs[0].wt, rigs[0].bounds))
    r0, S = rigs[0], len(sizes)
    plan = native.FusionPlan(0, T, r0.widths, r0.heights)
    cap, P = plan.capacity, plan.pixels_per_tick
    plan.set_params(r0.intr, r0.wt, r0.bounds)
    dev = torch.device("cuda", 0)
    depth = torch.from_numpy(np.stack([r.depth_maps.view(np.int16) for r in rigs])).to(dev).contiguous()
    rgb = torch.from_numpy(np.stack([r.depth_colors for r in rigs])).to(dev).contiguous()
    cd, cc = torch.empty_like(depth), torch.empty_like(rgb)
    verts = torch.zeros((T, cap, 16), dtype=torch.uint8, device=dev)
    offs = torch.zeros((T, S + 1), dtype=torch.int32, device=dev)
    tris = torch.zeros((T, 2 * cap, 3), dtype=torch.int32, device=dev)
    toffs = torch.zeros((T, S + 1), dtype=torch.int32, device=dev)
    plan.radial_correct_to(r0.intr, depth.data_ptr(), rgb.data_ptr(), cd.data_ptr(), cc.data_ptr())
    plan.run_mesh(cd.data_ptr(), cc.data_ptr(), verts.data_ptr(), offs.data_ptr(), tris.data_ptr(), toffs.data_ptr())
    torch.cuda.synchronize()
    for k, r in enumerate(rigs):
        what = f"seed {seed} tick {k} of {T}: sizes {sizes}"
        wd, wc = orc.radial_correction(r.depth_maps, r.depth_colors, r.widths, r.heights, r.intr)
        wd = np.ascontiguousarray(np.asarray(wd)).view(np.uint8).ravel()
        wc = np.ascontiguousarray(np.asarray(wc)).ravel()
        assert cd[k].cpu().numpy().view(np.uint8).tobytes() == wd.tobytes(), what + " [corrected depth]"
        assert cc[k].cpu().numpy().tobytes() == wc.tobytes(), what + " [corrected colours]"
        want_v, want_counts, want_t = orc.generate_mesh(wd, wc, r.widths, r.heights, r.intr, r.wt, r.bounds)
        o = offs[k].cpu().numpy()
        assert o[-1] == len(want_v), what
        assert np.array_equal(np.diff(o), np.asarray(want_counts).ravel()[:S]), what + " [offsets]"
        assert verts[k, :len(want_v)].cpu().numpy().tobytes() == want_v.tobytes(), what + " [vertices]"
        assert int(toffs[k, -1].item()) == len(want_t), what
        assert np.array_equal(tris[k, :len(want_t)].cpu().numpy(), np.asarray(want_t).reshape(-1, 3)), what + " [triangles]"
    # the cloud again through every way the compaction can get its offsets (lsnFusionSetMode), twice each: the second run of a mode counts
    # from the per-pixel depth thresholds the first one left
    for mode in (0, 1, 2, 0):
        plan.set_mode(mode)
        for rep in range(2):
            v2 = torch.zeros_like(verts)
            o2 = torch.zeros_like(offs)
            plan.run(cd.data_ptr(), cc.data_ptr(), v2.data_ptr(), o2.data_ptr())
            torch.cuda.synchronize()
            assert plan.check() == 0, f"seed {seed}: device-side flag after mode {mode}, run {rep}"
            assert torch.equal(o2, offs), f"seed {seed}: offsets of mode {mode}, run {rep}"
            for k in range(T):
                n = int(offs[k, -1].item())
                assert torch.equal(v2[k, :n], verts[k, :n]), f"seed {seed}: cloud of tick {k}, mode {mode}, run {rep}"
    plan.close()


@pytest.mark.parametrize("seed", range(24 * SCALE))
def test_random_clouds_icp(gpu, orc, seed):
    """The ICP export on random well-conditioned clouds (a bumpy surface patch seen twice, the second copy moved by a small rigid motion,
    resampled, with outliers and duplicated points): sizes from 40 to a few thousand, n2 above and below n1, 1-8 iterations.  1e-4 on
    the moved cloud and on R, t (the north-star's bar); brute-force oracle NN (lowest index on ties, as here)."""
    rng = np.random.default_rng(9000 + seed)
    n1, n2 = int(rng.integers(40, 4000)), int(rng.integers(40, 4000))

    def patch(n):
        xy = rng.uniform(-0.6, 0.6, size=(n, 2))
        z = 0.15 * np.sin(3.1 * xy[:, 0]) * np.cos(2.3 * xy[:, 1]) + 0.05 * xy[:, 0] * xy[:, 1] + 1.5
        return np.column_stack([xy, z])

    a = patch(n1)
    b = patch(n2)
    ang = np.radians(rng.uniform(-1.5, 1.5, size=3))
    Rm = synth.rot_y(ang[0]) @ synth.rot_x(ang[1])
    b = b @ Rm.T + rng.uniform(-0.01, 0.01, size=3)
    b += rng.normal(scale=0.001, size=b.shape)
    n_out = int(0.03 * n2)
    if n_out:
        b[rng.choice(n2, n_out, replace=False)] += rng.normal(scale=0.3, size=(n_out, 3))      # outliers: the 2.5 sigma rejection has work
    if seed % 3 == 0:
        a[rng.choice(n1, max(1, n1 // 20), replace=False)] = a[0]                                 # duplicated targets: distance ties
        b[rng.choice(n2, max(1, n2 // 20), replace=False)] = b[1]                                 # duplicated queries: contested targets
    a, b = a.astype(np.float32), b.astype(np.float32)
    iters = int(rng.integers(1, 9))
    got_v, got_R, got_t = native.icp(a, b, max_iter=iters)
    ref_v, ref_R, ref_t = orc.icp(a, b, max_iter=iters, nn_mode="brute")
    what = f"seed {seed}: n1 {n1} n2 {n2} iterations {iters}"
    assert np.isfinite(got_v).all(), what
    assert np.abs(got_v - ref_v).max() <= 1e-4, (what, float(np.abs(got_v - ref_v).max()))
    assert np.abs(got_R - ref_R).max() <= 1e-4 and np.abs(got_t - ref_t).max() <= 1e-4, what


def _random_nn_clouds(rng):
    """Targets and queries whose nearest-neighbour step exercises every branch of the near path and of the group search: volumes, surfaces,
    clusters, lattices with exact ties, any scale and offset from the origin, near and far queries mixed, duplicated and non-finite points."""
    n1, n2 = int(rng.integers(1, 20000)), int(rng.integers(1, 6000))
    kind = int(rng.integers(0, 5))
    if kind == 0:
        t = rng.uniform(-1, 1, size=(n1, 3))
    elif kind == 1:                                                           # a bumpy surface
        xy = rng.uniform(-1, 1, size=(n1, 2))
        t = np.column_stack([xy, 0.2 * np.sin(3 * xy[:, 0]) * np.cos(2 * xy[:, 1])])
    elif kind == 2:                                                           # clusters of very different density
        c = rng.uniform(-1, 1, size=(int(rng.integers(1, 12)), 3))
        t = c[rng.integers(0, len(c), size=n1)] + rng.normal(scale=10.0 ** rng.uniform(-4, -1), size=(n1, 3))
    elif kind == 3:                                                           # a lattice: exact f32 ties everywhere
        g = int(max(2, round(n1 ** (1 / 3))))
        t = np.stack(np.meshgrid(*[np.arange(g)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n1] * 0.0625
    else:                                                                     # a line and a plane: degenerate extents
        t = rng.uniform(-1, 1, size=(n1, 3)) * np.array([1.0, float(rng.integers(0, 2)), 0.0])
    t = np.asarray(t, np.float64)
    src = int(rng.integers(0, 4))
    if src == 0:                                                              # near: targets + small noise
        q = t[rng.integers(0, len(t), size=n2)] + rng.normal(scale=10.0 ** rng.uniform(-5, -1.5), size=(n2, 3))
    elif src == 1:                                                            # on the targets themselves / on lattice edge midpoints
        q = t[rng.integers(0, len(t), size=n2)] + (0.03125 if kind == 3 else 0.0)
    elif src == 2:                                                            # far and near mixed
        q = rng.uniform(-2, 2, size=(n2, 3))
        q[::3] = t[rng.integers(0, len(t), size=len(q[::3]))] + rng.normal(scale=1e-3, size=(len(q[::3]), 3))
    else:
        q = rng.uniform(-1.1, 1.1, size=(n2, 3))
    scale = 10.0 ** rng.uniform(-3, 3) if rng.random() < 0.5 else 1.0
    offset = (rng.uniform(-1, 1, size=3) * 10.0 ** rng.uniform(0, 3)) if rng.random() < 0.4 else np.zeros(3)
    t, q = (t * scale + offset).astype(np.float32), (q * scale + offset).astype(np.float32)
    if rng.random() < 0.3:
        t[rng.integers(0, len(t), size=max(1, len(t) // 10))] = t[0]          # duplicated targets
    if rng.random() < 0.15:
        t[int(rng.integers(0, len(t)))] = [np.nan, 0, 0]
        q[int(rng.integers(0, len(q)))] = [np.inf, 0, np.nan]
    return np.ascontiguousarray(t), np.ascontiguousarray(q)


@pytest.mark.parametrize("seed", range(40 * SCALE))
def test_random_clouds_nn_near_path_against_group_search_and_brute_force(gpu, seed, monkeypatch):
    """The NN step on random clouds three ways -- near path forced ($LSN_ICP_NEAR=2), off (every query through the box hierarchy), and the
    brute force -- must agree bit for bit on every finite query (index and f32 squared distance), and every index must be in range; then four
    ICP iterations on the same clouds with the near path forced and off: moved cloud, R, t and traces bit-identical.  (Coverage of the generator,
    measured once: the probe settles queries in 200 of 240 cases, 49 % of all queries.)"""
    import torch
    rng = np.random.default_rng(31000 + seed)
    t, q = _random_nn_clouds(rng)
    td, qd = torch.from_numpy(t).cuda(), torch.from_numpy(q).cuda()
    st = int(torch.cuda.current_stream().cuda_stream)
    outs = []
    for near, mode in (("2", native.NN_GRID), ("0", native.NN_GRID), ("1", native.NN_BRUTE)):
        monkeypatch.setenv("LSN_ICP_NEAR", near)
        ws = native.IcpWorkspace(0, len(t), len(q))
        idx = torch.full((len(q),), -7, dtype=torch.int32, device="cuda"); d2 = torch.zeros(len(q), dtype=torch.float32, device="cuda")
        ws.nearest(td.data_ptr(), len(t), qd.data_ptr(), len(q), idx.data_ptr(), d2.data_ptr(), mode, st)
        torch.cuda.synchronize()
        outs.append((idx.cpu().numpy(), d2.cpu().numpy()))
        ws.close()
    monkeypatch.delenv("LSN_ICP_NEAR")
    ok = np.isfinite(q).all(axis=1)
    what = f"seed {seed}: n1 {len(t)} n2 {len(q)}"
    for k in (1, 2):
        assert np.array_equal(outs[0][0][ok], outs[k][0][ok]), (what, k, np.flatnonzero(outs[0][0][ok] != outs[k][0][ok])[:5])
        assert np.array_equal(outs[0][1][ok].view(np.uint32), outs[k][1][ok].view(np.uint32)), (what, k)
    for i, _ in outs:
        assert ((i >= 0) & (i < len(t))).all(), what
    # ... and whole ICP runs (seeded steps: the bound is the previous neighbour's distance) with the near path forced and off: same bits
    if np.isfinite(t).all() and np.isfinite(q).all():
        runs = []
        for near in ("2", "0"):
            monkeypatch.setenv("LSN_ICP_NEAR", near)
            ws = native.IcpWorkspace(0, len(t), len(q))
            v2 = qd.clone()
            Rt = torch.tensor([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=torch.float32, device="cuda")
            ws.run(td.data_ptr(), len(t), v2.data_ptr(), len(q), Rt.data_ptr(), Rt.data_ptr() + 36, 4, native.NN_GRID, st)
            torch.cuda.synchronize()
            runs.append((v2.cpu().numpy().view(np.uint32), Rt.cpu().numpy().view(np.uint32), ws.trace(4, st).view(np.uint32)))
            ws.close()
        monkeypatch.delenv("LSN_ICP_NEAR")
        for k in range(3):
            assert np.array_equal(runs[0][k], runs[1][k]), (what, "ICP run", k)


@pytest.mark.parametrize("env", [{"LSN_HOST_DEVICES": "0,0,0"}, {"LSN_HOST_PATH": "grouped"}, {"LSN_HOST_PATH": "direct"}],
                         ids=["sharded-3-parts", "grouped", "direct"])
def test_random_rigs_in_the_other_host_flows(gpu, env):
    """The same rigs with the calls sharded over three "devices" (the one GPU listed three times) and with each of the two one-device flows
    forced: the flow is chosen once per process, so each runs as ONE child test run of this file."""
    if os.environ.get("LSN_FUZZ_CHILD"):
        pytest.skip("the child run")
    e = dict(os.environ, LSN_FUZZ_CHILD="1", **env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k", "random_rig_matches", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, env=e, timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1500:])
    assert f"{N_CASES + N_LARGE} passed" in r.stdout, r.stdout[-500:]
