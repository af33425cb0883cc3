"""GPU parity of the ICP path (exact NN, one-to-one matching, rejection, Kabsch) against the CPU oracle.

Bars: nearest-neighbour indices and f32 squared distances bit-exact (integer/index work); XYZ after ICP, R and t
within 1e-4 m / 1e-4 of the oracle (the north-star tolerance; GPU sums are in double, the reference's in f32)."""
import glob
import os

import numpy as np
import pytest

from livescan3d_amd import native, synth

pytestmark = pytest.mark.gpu

TOL = 1e-4  # metres, BASELINE.json north_star: "XYZ within 1e-4 m after ICP"


def _xyz(v):
    return np.stack([v["X"], v["Y"], v["Z"]], axis=1).astype(np.float32)


def _scene_clouds(orc, n, w, h, seed=4, perturb=True):
    rig = synth.make_rig("scene", n, w, h, seed=seed, perturb=perturb)
    v, counts = orc.generate_mesh_vertices(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    xyz = _xyz(v)
    edges = np.concatenate([[0], np.cumsum(counts)])
    return [xyz[edges[i]:edges[i + 1]].copy() for i in range(n)]


def _gpu_nn(targets, queries, mode):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(targets, np.float32)).cuda()
    q = torch.from_numpy(np.ascontiguousarray(queries, np.float32)).cuda()
    idx = torch.full((len(queries),), -7, dtype=torch.int32, device="cuda")
    d2 = torch.zeros(len(queries), dtype=torch.float32, device="cuda")
    ws = native.IcpWorkspace(0, len(targets), len(queries))
    ws.nearest(t.data_ptr(), len(targets), q.data_ptr(), len(queries), idx.data_ptr(), d2.data_ptr(), mode,
               int(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    ws.close()
    return idx.cpu().numpy().astype(np.int64), d2.cpu().numpy()


@pytest.mark.parametrize("mode", [native.NN_BRUTE, native.NN_GRID])
def test_nn_matches_oracle_bitexact(gpu, orc, mode):
    rng = np.random.default_rng(5)
    clouds = _scene_clouds(orc, 3, 256, 212)
    cases = [
        (clouds[1], clouds[0]),                                           # overlapping surfaces
        (np.concatenate([clouds[1], clouds[2]]), clouds[0]),              # merged target
        (rng.normal(size=(3000, 3)).astype(np.float32), rng.normal(size=(2000, 3)).astype(np.float32) * 3 + 1),  # volume + far outliers
        (rng.uniform(-1, 1, size=(1, 3)).astype(np.float32), rng.uniform(-1, 1, size=(500, 3)).astype(np.float32)),  # single target
        (np.repeat(rng.uniform(-1, 1, size=(50, 3)).astype(np.float32), 4, axis=0), rng.uniform(-1, 1, size=(300, 3)).astype(np.float32)),  # duplicated targets: ties -> lowest index
        ((rng.uniform(0, 1, size=(4000, 3)) * [1, 1, 0]).astype(np.float32), rng.uniform(-0.5, 1.5, size=(1500, 3)).astype(np.float32)),  # flat cloud (zero extent in z)
    ]
    for n2 in (63, 64, 65, 128, 129):                                    # query counts around the 64-query group size
        cases.append((rng.uniform(-1, 1, size=(777, 3)).astype(np.float32), rng.uniform(-1.1, 1.1, size=(n2, 3)).astype(np.float32)))
    cases.append((np.full((300, 3), 0.25, np.float32), rng.uniform(-1, 1, size=(200, 3)).astype(np.float32)))   # every target the same point
    for n1 in (2, 7, 8, 9, 15, 16, 17, 63, 65, 1031):                    # target counts around the brute-force kernel's 8-point steps and slices
        cases.append((rng.uniform(-1, 1, size=(n1, 3)).astype(np.float32), rng.uniform(-1.1, 1.1, size=(150, 3)).astype(np.float32)))
    for k, (t, q) in enumerate(cases):
        want_i, want_d = orc.nn(t, q, mode="brute", n_threads=8)
        got_i, got_d = _gpu_nn(t, q, mode)
        assert np.array_equal(got_d.view(np.uint32), want_d.view(np.uint32)), f"case {k}: squared distances differ"
        assert np.array_equal(got_i, want_i), f"case {k}: indices differ at {np.flatnonzero(got_i != want_i)[:5]}"


GOLDEN_NN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "nn_*.npz")))


@pytest.mark.parametrize("path", GOLDEN_NN, ids=[os.path.basename(p) for p in GOLDEN_NN])
@pytest.mark.parametrize("mode", [native.NN_BRUTE, native.NN_GRID])
def test_nn_against_reference_nanoflann_fixture_directly(gpu, path, mode):
    """The HIP NN straight against the indices / squared distances the reference's own nanoflann step produced
    (tests/golden/nn_*.npz, generated from oracle/_ref): no oracle in between."""
    g = np.load(path)
    got_i, got_d = _gpu_nn(g["targets"], g["queries"], mode)
    assert np.array_equal(got_d.view(np.uint32), g["dist2"].view(np.uint32))
    assert np.array_equal(got_i, g["idx"].astype(np.int64))


@pytest.mark.parametrize("name", ["configs1_2x512x424", "configs2_8x512x424", "lattice_2x192x160"])
def test_nn_full_size_scene_rigs_against_reference_fixture(gpu, orc, name):
    """configs[1] and configs[2] at their real sizes (2 / 8 sensors x 512x424; configs[2]: n1 = 738 k targets, the caller shape
    of MainWindowForm.cs:349-376) plus a lattice-snapped rig with 3.9 % exact f32 ties: voxel-grid NN and brute-force NN agree
    bit for bit with each other, with the oracle's kd-tree, and with what the reference's nanoflann step answered for every
    query (tests/golden/nn_scene_full.json) -- indices may differ from the reference's only at exact f32 ties, where ours
    is the lowest tied index (nanoflann's depends on its traversal, include/nanoflann.h:1200-1247)."""
    from tests import scene_cases
    case = scene_cases.load_cases()[name]
    tgt, src = scene_cases.case_clouds(orc, name, case)
    gi, gd = _gpu_nn(tgt, src, native.NN_GRID)
    bi, bd = _gpu_nn(tgt, src, native.NN_BRUTE)
    assert np.array_equal(gi, bi) and np.array_equal(gd.view(np.uint32), bd.view(np.uint32)), "grid and brute-force NN differ"
    scene_cases.check_against_reference(case, gi, gd, "HIP NN")
    oi, od = orc.nn(tgt, src, mode="kdtree", n_threads=8)
    assert np.array_equal(gi, oi) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))
    if orc.have_ref_nn() and case.get("lattice"):
        # live against the compiled reference where oracle/_ref travelled along: differences only at exact ties
        ri, rd = orc.ref_nn(tgt, src)
        assert np.array_equal(rd.view(np.uint32), gd.view(np.uint32))
        diff = np.flatnonzero(ri != gi)
        assert set(diff.tolist()) <= set(case["tie_queries"])
        d = tgt[ri[diff]] - src[diff]
        assert np.array_equal(((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).view(np.uint32), gd[diff].view(np.uint32))


def test_nn_list_overflow_falls_back_to_the_complete_walk(gpu, orc, monkeypatch):
    """The NN step's work lists have a fixed capacity; when one overflows every query group runs the complete hierarchy walk
    instead.  $LSN_ICP_TINY_LISTS (read when a workspace is created) forces that path: same bits."""
    monkeypatch.setenv("LSN_ICP_TINY_LISTS", "1")
    clouds = _scene_clouds(orc, 2, 256, 212)
    want_i, want_d = orc.nn(clouds[0], clouds[1], mode="kdtree", n_threads=8)
    got_i, got_d = _gpu_nn(clouds[0], clouds[1], native.NN_GRID)
    assert np.array_equal(got_d.view(np.uint32), want_d.view(np.uint32)) and np.array_equal(got_i, want_i)
    got_v, got_R, got_t = native.icp(clouds[0], clouds[1], max_iter=4)       # the export's workspace was created earlier: normal path
    monkeypatch.delenv("LSN_ICP_TINY_LISTS")
    ref_v, ref_R, ref_t = orc.icp(clouds[0], clouds[1], max_iter=4, n_threads=8)
    assert np.abs(got_v - ref_v).max() <= TOL


def _near_cases(orc):
    """Clouds that exercise the near path's corners: where the bound is tiny, zero, at a cell edge, far from the origin."""
    rng = np.random.default_rng(21)
    clouds = _scene_clouds(orc, 3, 256, 212)
    base = rng.uniform(-1, 1, size=(6000, 3)).astype(np.float32)
    lattice = (np.stack(np.meshgrid(np.arange(24), np.arange(24), np.arange(8), indexing="ij"), -1).reshape(-1, 3) * 0.0625).astype(np.float32)
    cases = {
        "overlapping surfaces": (clouds[1], clouds[0]),
        "merged target": (np.concatenate([clouds[1], clouds[2]]), clouds[0]),
        "source = target (bound 0)": (base, base.copy()),
        "source = target + 1e-4 noise": (base, base + rng.normal(scale=1e-4, size=base.shape).astype(np.float32)),
        "source = target + 2e-2 noise": (base, base + rng.normal(scale=2e-2, size=base.shape).astype(np.float32)),
        "duplicated targets, queries on them (ties -> lowest index)": (np.repeat(base[:1500], 4, axis=0), base[:1500].copy()),
        "lattice, queries at cell-edge midpoints (exact ties)": (lattice, lattice[::3] + np.float32(0.03125)),
        "lattice shuffled": (lattice[rng.permutation(len(lattice))], lattice[::2] + np.array([0.03125, 0, 0], np.float32)),
        "far from the origin (|q| >> r)": (base * 0.01 + 1000.0, base[:4000] * 0.01 + 1000.0 + rng.normal(scale=1e-4, size=(4000, 3)).astype(np.float32)),
        "huge coordinates": (base * 1e6, base[:3000] * 1e6 + rng.normal(scale=10.0, size=(3000, 3)).astype(np.float32)),
        "tiny coordinates": (base * 1e-12, base[:3000] * 1e-12),
        "flat target": ((base * [1, 1, 0]).astype(np.float32), (base[:3000] * [1, 1, 0]).astype(np.float32) + np.array([0, 0, 1e-3], np.float32)),
        "one cell holds everything + outlier": (np.concatenate([base[:3000] * 1e-3, [[50, 50, 50]]]).astype(np.float32), (base[:2000] * 1e-3).astype(np.float32)),
    }
    t = base[:5000].copy(); q = t[:3000] + rng.normal(scale=1e-3, size=(3000, 3)).astype(np.float32)
    t[7] = [np.nan, 0, 0]; t[4000] = [np.inf, 1, 1]; t[123, 2] = -np.inf
    cases["non-finite targets beside near queries"] = (t, q)
    return cases


def test_near_path_gives_the_group_searchs_bits(gpu, orc, monkeypatch):
    """The near path (a query with a small bound walks the grid cells around it, icp.hip near_search) against the same step with
    the path off ($LSN_ICP_NEAR=0: every query through the box hierarchy), against the brute force and against the oracle:
    indices and squared distances bit for bit -- and the path did settle queries in every case (lsnIcpNearResolved)."""
    import torch
    for name, (t, q) in _near_cases(orc).items():
        t = np.ascontiguousarray(t, np.float32); q = np.ascontiguousarray(q, np.float32)
        outs = {}
        for near in ("2", "0"):
            monkeypatch.setenv("LSN_ICP_NEAR", near)
            ws = native.IcpWorkspace(0, len(t), len(q))
            td, qd = torch.from_numpy(t).cuda(), torch.from_numpy(q).cuda()
            idx = torch.full((len(q),), -7, dtype=torch.int32, device="cuda"); d2 = torch.zeros(len(q), dtype=torch.float32, device="cuda")
            st = int(torch.cuda.current_stream().cuda_stream)
            ws.nearest(td.data_ptr(), len(t), qd.data_ptr(), len(q), idx.data_ptr(), d2.data_ptr(), native.NN_GRID, st)
            outs[near] = (idx.cpu().numpy().astype(np.int64), d2.cpu().numpy(), ws.near_resolved(st))
            ws.close()
        monkeypatch.delenv("LSN_ICP_NEAR")
        assert outs["0"][2] == 0, name
        if name.startswith("one cell"):
            assert outs["2"][2] == 0, name          # 3000 candidates in the query's cell: over the cap, left to the group search
        elif name.startswith(("tiny", "non-finite")):
            pass                                    # degenerate grids (an infinite box; a cell edge clamped at extent / 4): crowded cells, cap decides
        else:
            assert outs["2"][2] > 0, f"{name}: the near path settled nothing"
        assert np.array_equal(outs["2"][0], outs["0"][0]) and np.array_equal(outs["2"][1].view(np.uint32), outs["0"][1].view(np.uint32)), name
        bi, bd = _gpu_nn(t, q, native.NN_BRUTE)
        assert np.array_equal(outs["2"][0], bi) and np.array_equal(outs["2"][1].view(np.uint32), bd.view(np.uint32)), name
        want_i, want_d = orc.nn(t, q, mode="brute", n_threads=8)
        ok = np.isfinite(q).all(axis=1)
        assert np.array_equal(outs["2"][0][ok], want_i[ok]) and np.array_equal(outs["2"][1][ok].view(np.uint32), want_d[ok].view(np.uint32)), name


@pytest.mark.parametrize("cap", ["1", "7", "100000"])
def test_near_path_candidate_caps(gpu, orc, monkeypatch, cap):
    """$LSN_ICP_NEAR_PTS: whatever the cap hands to the group search, the bits stay."""
    monkeypatch.setenv("LSN_ICP_NEAR_PTS", cap)
    clouds = _scene_clouds(orc, 2, 256, 212)
    want_i, want_d = orc.nn(clouds[0], clouds[1], mode="kdtree", n_threads=8)
    got_i, got_d = _gpu_nn(clouds[0], clouds[1], native.NN_GRID)
    assert np.array_equal(got_d.view(np.uint32), want_d.view(np.uint32)) and np.array_equal(got_i, want_i)


def test_icp_runs_identically_with_and_without_the_near_path(gpu, orc, monkeypatch):
    """A whole lsnIcpRun (seeded iterations: the bound is the previous neighbour's distance) with the near path forced, chosen
    per step by the library (the default) and off: the moved cloud, R, t and every iteration's trace bit-identical; forced, the
    last step settled a good part of the queries."""
    import torch
    clouds = _scene_clouds(orc, 3, 256, 212)
    tgt, src = np.concatenate(clouds[1:]), clouds[0]
    outs = {}
    for near in ("2", "1", "0"):
        monkeypatch.setenv("LSN_ICP_NEAR", near)
        v1 = torch.from_numpy(tgt).cuda(); v2 = torch.from_numpy(src.copy()).cuda()
        Rt = torch.tensor([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=torch.float32, device="cuda")
        ws = native.IcpWorkspace(0, len(tgt), len(src))
        st = int(torch.cuda.current_stream().cuda_stream)
        ws.run(v1.data_ptr(), len(tgt), v2.data_ptr(), len(src), Rt.data_ptr(), Rt.data_ptr() + 36, 8, native.NN_GRID, st)
        outs[near] = (v2.cpu().numpy(), Rt.cpu().numpy(), ws.trace(8, st), ws.near_resolved(st))
        ws.close()
    monkeypatch.delenv("LSN_ICP_NEAR")
    assert outs["0"][3] == 0 and outs["2"][3] > len(src) // 4 and outs["1"][3] in (0, outs["2"][3])
    for near in ("2", "1"):
        for k in range(3):
            assert np.array_equal(outs[near][k].view(np.uint32), outs["0"][k].view(np.uint32)), (near, k)


def test_nn_non_finite_points_do_not_derail_the_search(gpu, orc):
    """NaN / inf coordinates in either cloud: the finite queries still get their exact neighbour among the comparable targets,
    the others an index in range (they take no part in the culling), and nothing hangs."""
    rng = np.random.default_rng(11)
    t = rng.uniform(-1, 1, size=(5000, 3)).astype(np.float32)
    q = rng.uniform(-1.2, 1.2, size=(3000, 3)).astype(np.float32)
    t[7] = [np.nan, 0, 0]; t[4000] = [np.inf, 1, 1]; t[123, 2] = -np.inf
    bad_q = [0, 63, 64, 1999, 2999]
    q[0] = [np.nan, np.nan, np.nan]; q[63] = [np.inf, 0, 0]; q[64, 1] = np.nan; q[1999] = [-np.inf, np.inf, 0]; q[2999, 2] = np.nan
    want_i, want_d = orc.nn(t, q, mode="brute", n_threads=8)
    ok = np.ones(len(q), bool); ok[bad_q] = False
    for mode in (native.NN_GRID, native.NN_BRUTE):
        got_i, got_d = _gpu_nn(t, q, mode)
        assert np.array_equal(got_i[ok], want_i[ok]) and np.array_equal(got_d[ok].view(np.uint32), want_d[ok].view(np.uint32))
        assert ((got_i >= 0) & (got_i < len(t))).all()


def test_nn_large_clouds(gpu, orc):
    """2 sensors x 1024x1024 (configs[4]'s frame size): half a million points per cloud -- grid NN == oracle kd-tree NN, bit-exact,
    and one ICP call stays inside the tolerance."""
    clouds = _scene_clouds(orc, 2, 1024, 1024)
    assert min(len(c) for c in clouds) > 300000
    want_i, want_d = orc.nn(clouds[0], clouds[1], mode="kdtree", n_threads=8)
    got_i, got_d = _gpu_nn(clouds[0], clouds[1], native.NN_GRID)
    assert np.array_equal(got_d.view(np.uint32), want_d.view(np.uint32))
    assert np.array_equal(got_i, want_i)
    got_v, got_R, got_t = native.icp(clouds[0], clouds[1], max_iter=3)
    ref_v, ref_R, ref_t = orc.icp(clouds[0], clouds[1], max_iter=3, n_threads=8)
    assert np.abs(got_v - ref_v).max() <= TOL and np.abs(got_R - ref_R).max() <= TOL and np.abs(got_t - ref_t).max() <= TOL


def test_nn_full_size(gpu, orc):
    """config-2 sized clouds (2 x 512x424): grid NN == oracle kd-tree NN, bit-exact."""
    clouds = _scene_clouds(orc, 2, 512, 424)
    want_i, want_d = orc.nn(clouds[0], clouds[1], mode="kdtree", n_threads=8)
    got_i, got_d = _gpu_nn(clouds[0], clouds[1], native.NN_GRID)
    assert np.array_equal(got_d.view(np.uint32), want_d.view(np.uint32))
    assert np.array_equal(got_i, want_i)


@pytest.mark.parametrize("w,h,iters", [(128, 96, 10), (512, 424, 10)])
def test_icp_export_matches_oracle(gpu, orc, w, h, iters):
    clouds = _scene_clouds(orc, 2, w, h)
    got_v, got_R, got_t = native.icp(clouds[0], clouds[1], max_iter=iters)
    ref_v, ref_R, ref_t = orc.icp(clouds[0], clouds[1], max_iter=iters, n_threads=8)
    assert np.abs(got_v - ref_v).max() <= TOL
    assert np.abs(got_R - ref_R).max() <= TOL and np.abs(got_t - ref_t).max() <= TOL
    # the call moved the cloud and is consistent with its own R, t: v_out = (v_in + t) R   
    assert np.abs(got_v - clouds[1]).max() > 1e-3
    recon = (clouds[1].astype(np.float64) + got_t.astype(np.float64)) @ got_R.astype(np.float64)
    assert np.abs(recon - got_v).max() <= 5e-5
    assert abs(np.linalg.det(got_R.astype(np.float64)) - 1.0) < 1e-5
    assert np.abs(got_R.astype(np.float64) @ got_R.astype(np.float64).T - np.eye(3)).max() < 1e-5


def test_icp_trace_and_modes_agree(gpu, orc):
    """Per-iteration match counts / T / Rn against the oracle's trace; brute-force and grid NN give identical runs."""
    import torch
    clouds = _scene_clouds(orc, 2, 192, 160)
    ref_v, ref_R, ref_t, tr = orc.icp(clouds[0], clouds[1], max_iter=6, trace=True, n_threads=8)
    outs = []
    for mode in (native.NN_BRUTE, native.NN_GRID):
        v1 = torch.from_numpy(clouds[0]).cuda()
        v2 = torch.from_numpy(clouds[1].copy()).cuda()
        Rt = torch.tensor([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=torch.float32, device="cuda")
        ws = native.IcpWorkspace(0, len(clouds[0]), len(clouds[1]))
        ws.run(v1.data_ptr(), len(clouds[0]), v2.data_ptr(), len(clouds[1]), Rt.data_ptr(), Rt.data_ptr() + 36, 6, mode,
               int(torch.cuda.current_stream().cuda_stream))
        g = ws.trace(6, int(torch.cuda.current_stream().cuda_stream))
        outs.append((v2.cpu().numpy(), Rt.cpu().numpy(), g))
        ws.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    g = outs[1][2]
    assert len(g) == 6
    for it in range(6):
        assert abs(int(g[it, 0]) - int(tr[it]["n_matched"])) <= 2
        assert abs(int(g[it, 1]) - int(tr[it]["n_kept"])) <= max(3, int(0.001 * tr[it]["n_kept"]))
        assert np.abs(g[it, 4:7] - tr[it]["T"]).max() <= 2e-5
        assert np.abs(g[it, 7:16] - tr[it]["Rn"]).max() <= 2e-5
    assert np.abs(outs[1][0] - ref_v).max() <= TOL


def test_icp_edge_cases(gpu, orc):
    rng = np.random.default_rng(9)
    a = rng.uniform(-1, 1, size=(400, 3)).astype(np.float32)
    # identical clouds: already aligned, stays put
    v, R, t = native.icp(a, a.copy(), max_iter=3)
    assert np.abs(v - a).max() <= 1e-6 and np.abs(R - np.eye(3)).max() <= 1e-6 and np.abs(t).max() <= 1e-6
    # tiny clouds
    for n1, n2 in [(1, 1), (1, 7), (5, 1), (2, 2)]:
        t1 = rng.uniform(-1, 1, size=(n1, 3)).astype(np.float32)
        t2 = rng.uniform(-1, 1, size=(n2, 3)).astype(np.float32)
        gv, gR, gt = native.icp(t1, t2, max_iter=4)
        rv, rR, rt = orc.icp(t1, t2, max_iter=4, nn_mode="brute")
        assert np.allclose(gv, rv, atol=TOL, equal_nan=True) and np.allclose(gR, rR, atol=TOL, equal_nan=True) and np.allclose(gt, rt, atol=TOL, equal_nan=True)
    # empty clouds: untouched, error reported, returns 1.0 (callers guard with nClientCount >= 2, MainWindowForm.cs:469-473)
    import ctypes as C
    L = native.lib()
    R = np.eye(3, dtype=np.float32).ravel(); tt = np.zeros(3, np.float32)
    ret = L.ICP(a.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p), 0, 10, R.ctypes.data_as(C.c_void_p), tt.ctypes.data_as(C.c_void_p), 10)
    assert ret == 1.0 and native.last_error() != "" and np.array_equal(R, np.eye(3, dtype=np.float32).ravel())


def test_refine_loop_matches_oracle(gpu, orc):
    """lsnRefine = refineWorker_DoWork (MainWindowForm.cs:330-410) with the clouds resident in HBM: Gauss-Seidel order,
    accumulated Rs/Ts, and the C#'s in-place pose composition -- against the oracle's mirror of the same loop."""
    import time
    clouds = _scene_clouds(orc, 4, 256, 212, seed=7)
    n = len(clouds)
    rng = np.random.default_rng(3)
    wR = np.stack([synth.rot_y(0.3 * i) @ synth.rot_x(0.1 * i) for i in range(n)]).astype(np.float32)
    wt = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    t0 = time.perf_counter()
    got_c, got_R, got_t, got_Rs, got_Ts = native.refine(clouds, wR, wt, n_refine_iters=2, n_icp_iters=5)
    t_gpu = time.perf_counter() - t0
    ref_c, ref_R, ref_t, ref_Rs, ref_Ts = orc.refine(clouds, wR, wt, n_refine_iters=2, n_icp_iters=5, n_threads=8)
    for i in range(n):
        assert np.abs(got_c[i] - ref_c[i]).max() <= TOL, i
        assert np.abs(got_c[i] - clouds[i]).max() > 1e-4          # the clouds did move
    assert np.abs(got_Rs - ref_Rs).max() <= TOL and np.abs(got_Ts - ref_Ts).max() <= TOL
    assert np.abs(got_R - ref_R).max() <= TOL and np.abs(got_t - ref_t).max() <= TOL
    # fewer than two sensors: nothing to refine against (MainWindowForm.cs:469-473 guards the same way)
    c1, R1, t1, Rs1, Ts1 = native.refine(clouds[:1], wR[:1], wt[:1])
    assert np.array_equal(c1[0], clouds[0]) and np.array_equal(Rs1[0], np.eye(3, dtype=np.float32)) and not Ts1.any()
    assert np.array_equal(R1[0], wR[0]) and np.array_equal(t1[0], wt[0])


_CONFIGS2_ORACLE = {}


@pytest.mark.parametrize("mode", [native.NN_GRID, native.NN_BRUTE], ids=["grid", "brute"])
def test_icp_at_the_callers_shape_configs2(gpu, orc, mode):
    """BASELINE configs[2] = the shape LiveScanServer's refine loop calls ICP with (MainWindowForm.cs:347-376): the source is one
    sensor's cloud (n2 ~ 108 k), the target all seven others' (n1 ~ 738 k), maxIter = 10 -- both NN modes against the oracle's run
    (kd-tree NN) within 1e-4 on every vertex, R and t."""
    import torch
    from tests import scene_cases
    if "run" not in _CONFIGS2_ORACLE:
        clouds = scene_cases.scene_clouds(orc, 8)
        tgt, src = np.concatenate(clouds[1:]), clouds[0]
        _CONFIGS2_ORACLE["run"] = (tgt, src, orc.icp(tgt, src, max_iter=10, n_threads=8))
    tgt, src, (ref_v, ref_R, ref_t) = _CONFIGS2_ORACLE["run"]
    assert len(tgt) > 700000 and len(src) > 100000
    v1 = torch.from_numpy(tgt).cuda()
    v2 = torch.from_numpy(src.copy()).cuda()
    Rt = torch.tensor([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=torch.float32, device="cuda")
    ws = native.IcpWorkspace(0, len(tgt), len(src))
    st = int(torch.cuda.current_stream().cuda_stream)
    ws.run(v1.data_ptr(), len(tgt), v2.data_ptr(), len(src), Rt.data_ptr(), Rt.data_ptr() + 36, 10, mode, st)
    torch.cuda.synchronize()
    got_v, got = v2.cpu().numpy(), Rt.cpu().numpy()
    ws.close()
    assert np.abs(got_v - ref_v).max() <= TOL
    assert np.abs(got[:9].reshape(3, 3) - ref_R).max() <= TOL and np.abs(got[9:] - ref_t).max() <= TOL
    assert np.abs(got_v - src).max() > 1e-3                        # the source did move


def test_refine_pass_at_full_size(gpu, orc):
    """The whole refine pass of LiveScanServer at its real size: 8 sensors x 512x424, nNumRefineIters = 2, nNumICPIterations = 10
    (160 ICP iterations, every call against the seven other clouds) through lsnRefine against the oracle's mirror of the loop."""
    from tests import scene_cases
    clouds = scene_cases.scene_clouds(orc, 8)
    n = len(clouds)
    wR = np.stack([synth.rot_y(2 * np.pi * i / n) for i in range(n)]).astype(np.float32)
    wt = np.tile(np.array([0, 0, -2.0], np.float32), (n, 1))
    got_c, got_R, got_t, got_Rs, got_Ts = native.refine(clouds, wR, wt, n_refine_iters=2, n_icp_iters=10)
    ref_c, ref_R, ref_t, ref_Rs, ref_Ts = orc.refine(clouds, wR, wt, n_refine_iters=2, n_icp_iters=10, n_threads=16)
    for i in range(n):
        assert np.abs(got_c[i] - ref_c[i]).max() <= TOL, i
    assert np.abs(got_Rs - ref_Rs).max() <= TOL and np.abs(got_Ts - ref_Ts).max() <= TOL
    assert np.abs(got_R - ref_R).max() <= TOL and np.abs(got_t - ref_t).max() <= TOL
