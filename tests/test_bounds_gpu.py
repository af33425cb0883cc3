"""Nothing is written outside the buffers the device-resident entry points are given.

Every output of lsnFusionRun / lsnFusionRunMesh / lsnFusionRadialCorrectTo / lsnIcpRun / lsnTransferPack / lsnPlyPack sits in the
middle of a larger allocation filled with a pattern; after the call the bytes either side of the documented extent (include/NativeUtils.h)
must still hold it.  The kernels store in 16-byte chunks aligned to the DESTINATION with ragged ends element by element (fusion.hip,
mesh.hip, radial.hip), so the cases are the ones where those ends exist: ragged rigs, widths that are not multiples of 8 (the scalar
paths), several ticks, vertex / triangle counts that are not multiples of anything.  Results are compared with the oracle as well, so a
store that went missing inside the buffer shows up too.  (GPU AddressSanitizer is not available on this pool.)"""
import numpy as np
import pytest

from livescan3d_amd import native, synth

pytestmark = pytest.mark.gpu

GUARD = 4096
PATTERN = 0xA5

UNIFORM = [(512, 424)] * 3
RAGGED = [(61, 37), (512, 424), (100, 3), (7, 5), (2049, 1), (64, 48), (129, 65)]


class Guarded:
    """`nbytes` of device memory with GUARD pattern bytes either side."""

    def __init__(self, torch, nbytes, dev):
        self.torch, self.n = torch, int(nbytes)
        self.buf = torch.full((GUARD + self.n + GUARD,), PATTERN, dtype=torch.uint8, device=dev)

    @property
    def ptr(self):
        return self.buf.data_ptr() + GUARD

    def body(self):
        return self.buf[GUARD:GUARD + self.n]

    def intact(self):
        return bool((self.buf[:GUARD] == PATTERN).all().item()) and bool((self.buf[GUARD + self.n:] == PATTERN).all().item())


def _rig(sizes, seed, kind):
    n = len(sizes)
    depths, rgbs, intr, wt = [], [], [], []
    for s, (w, h) in enumerate(sizes):
        d, c = synth.scene_frame(seed, 0, s, n, w, h) if kind == "scene" and w >= 64 and h >= 48 else synth.noise_frame(seed, 0, s, w, h)
        depths.append(d); rgbs.append(c)
        intr.append(synth.kinect_intrinsics(w, h))
        wt.append(synth.pack_pose(*synth.ring_pose(s, n)))
    return synth.Rig(depths, rgbs, np.concatenate(intr), np.concatenate(wt), synth.CROP_BOUNDS)


@pytest.fixture(scope="module")
def torch_dev(gpu):
    import torch
    return torch, torch.device("cuda", 0)


@pytest.mark.parametrize("sizes,kind,ticks", [(UNIFORM, "scene", 1), (UNIFORM, "noise", 3), (RAGGED, "scene", 1), (RAGGED, "noise", 2)],
                         ids=["uniform-scene-1", "uniform-noise-3", "ragged-scene-1", "ragged-noise-2"])
def test_fusion_mesh_and_radial_stay_inside_their_buffers(torch_dev, orc, sizes, kind, ticks):
    torch, dev = torch_dev
    rig = _rig(sizes, 21, kind)
    S = len(sizes)
    plan = native.FusionPlan(0, ticks, rig.widths, rig.heights)
    cap, P = plan.capacity, plan.pixels_per_tick
    plan.set_params(rig.intr, rig.wt, rig.bounds)
    depth = torch.from_numpy(rig.depth_maps.view(np.int16).copy()).to(dev).unsqueeze(0).repeat(ticks, 1).contiguous()
    rgb = torch.from_numpy(rig.depth_colors.copy()).to(dev).unsqueeze(0).repeat(ticks, 1).contiguous()
    want_v, _, want_t = orc.generate_mesh(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)

    # ---- lsnFusionRun ----
    verts, offs = Guarded(torch, ticks * cap * 16, dev), Guarded(torch, ticks * (S + 1) * 4, dev)
    plan.run(depth.data_ptr(), rgb.data_ptr(), verts.ptr, offs.ptr)
    torch.cuda.synchronize()
    assert verts.intact() and offs.intact(), "lsnFusionRun wrote outside vertices / offsets"
    o = offs.body().view(torch.int32).view(ticks, S + 1).cpu().numpy()
    for k in range(ticks):
        assert o[k, -1] == len(want_v)
        got = verts.body().view(ticks, cap, 16)[k, :len(want_v)].cpu().numpy().view(native.VERTEX_DTYPE).reshape(-1)
        assert got.tobytes() == want_v.tobytes(), f"tick {k}"

    # ---- lsnFusionRunMesh ----
    verts, offs = Guarded(torch, ticks * cap * 16, dev), Guarded(torch, ticks * (S + 1) * 4, dev)
    tris, toffs = Guarded(torch, ticks * 2 * cap * 12, dev), Guarded(torch, ticks * (S + 1) * 4, dev)
    plan.run_mesh(depth.data_ptr(), rgb.data_ptr(), verts.ptr, offs.ptr, tris.ptr, toffs.ptr)
    torch.cuda.synchronize()
    assert verts.intact() and offs.intact() and tris.intact() and toffs.intact(), "lsnFusionRunMesh wrote outside its outputs"
    to = toffs.body().view(torch.int32).view(ticks, S + 1).cpu().numpy()
    for k in range(ticks):
        assert to[k, -1] == len(want_t)
        got_t = tris.body().view(torch.int32).view(ticks, 2 * cap, 3)[k, :len(want_t)].cpu().numpy()
        assert np.array_equal(got_t, want_t), f"tick {k}"

    # ---- lsnFusionRadialCorrectTo (out of place) and in place ----
    cd, cc = orc.radial_correction(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr)
    cd = np.ascontiguousarray(np.asarray(cd)).view(np.uint8).ravel()
    cc = np.ascontiguousarray(np.asarray(cc)).ravel()
    out_d, out_c = Guarded(torch, ticks * P * 2, dev), Guarded(torch, ticks * P * 3, dev)
    plan.radial_correct_to(rig.intr, depth.data_ptr(), rgb.data_ptr(), out_d.ptr, out_c.ptr)
    torch.cuda.synchronize()
    assert out_d.intact() and out_c.intact(), "lsnFusionRadialCorrectTo wrote outside the corrected maps"
    for k in range(ticks):
        assert out_d.body().view(ticks, P * 2)[k].cpu().numpy().tobytes() == cd.tobytes(), f"tick {k}"
        assert out_c.body().view(ticks, P * 3)[k].cpu().numpy().tobytes() == cc.tobytes(), f"tick {k}"
    in_d, in_c = Guarded(torch, ticks * P * 2, dev), Guarded(torch, ticks * P * 3, dev)
    in_d.body().copy_(depth.view(torch.uint8).view(-1))
    in_c.body().copy_(rgb.view(-1))
    plan.radial_correct(rig.intr, in_d.ptr, in_c.ptr)
    torch.cuda.synchronize()
    assert in_d.intact() and in_c.intact(), "lsnFusionRadialCorrect (in place) wrote outside the maps"
    assert in_d.body().view(ticks, P * 2)[ticks - 1].cpu().numpy().tobytes() == cd.tobytes()
    assert in_c.body().view(ticks, P * 3)[ticks - 1].cpu().numpy().tobytes() == cc.tobytes()
    plan.close()


def test_wire_packers_stay_inside_their_buffers(torch_dev, orc):
    torch, dev = torch_dev
    rig = _rig([(160, 120)] * 3, 5, "scene")
    v, _, t = orc.generate_mesh(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    nv, nt = len(v), len(t)
    assert nv > 1000 and nt > 1000
    d_v = torch.from_numpy(v.view(np.uint8).reshape(-1).copy()).to(dev)
    d_t = torch.from_numpy(np.ascontiguousarray(t, dtype=np.int32).reshape(-1).copy()).to(dev)
    bound = native.transfer_frame_bound(nv, nt)
    out = Guarded(torch, bound, dev)
    packer = native.TransferPacker(0, nv, nt)
    n = packer.pack(d_v.data_ptr(), nv, d_t.data_ptr(), nt, out.ptr, bound, 0)
    torch.cuda.synchronize()
    assert 0 < n <= bound and out.intact(), "lsnTransferPack wrote outside the frame buffer"
    assert out.body()[:n].cpu().numpy().tobytes() == np.asarray(orc.transfer_frame(v, t)).tobytes()
    packer.close()
    pb = native.ply_binary_bytes(nv, nt)
    ply = Guarded(torch, pb, dev)
    native.ply_pack(0, d_v.data_ptr(), nv, d_t.data_ptr(), nt, ply.ptr, pb, 0)
    torch.cuda.synchronize()
    assert ply.intact(), "lsnPlyPack wrote outside the image"


def test_icp_stays_inside_its_buffers(torch_dev, orc):
    """lsnIcpRun moves verts2 in place and updates R, t: 3 * n2, 9 and 3 floats, nothing either side (n2 is not a multiple of the
    workgroup size, so the last workgroup of every per-query launch is ragged)."""
    torch, dev = torch_dev
    rig = synth.make_rig("scene", 2, 160, 128, seed=4, perturb=True)
    v, counts = orc.generate_mesh_vertices(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    xyz = np.stack([v["X"], v["Y"], v["Z"]], axis=1).astype(np.float32)
    c0, c1 = xyz[:counts[0]], xyz[counts[0]:counts[0] + counts[1]]
    c1 = c1[:len(c1) - (len(c1) % 256 == 0)]   # keep the count off a workgroup multiple
    n1, n2 = len(c0), len(c1)
    assert n1 > 500 and n2 > 500 and n2 % 256 != 0
    tgt = torch.from_numpy(c0.copy()).to(dev)
    src, R, t = Guarded(torch, n2 * 12, dev), Guarded(torch, 36, dev), Guarded(torch, 12, dev)
    src.body().copy_(torch.from_numpy(c1.view(np.uint8).reshape(-1).copy()).to(dev))
    R.body().copy_(torch.from_numpy(np.eye(3, dtype=np.float32).view(np.uint8).reshape(-1).copy()).to(dev))
    t.body().zero_()
    ws = native.IcpWorkspace(0, n1, n2)
    ws.run(tgt.data_ptr(), n1, src.ptr, n2, R.ptr, t.ptr, 5, native.NN_GRID, 0)
    torch.cuda.synchronize()
    assert src.intact() and R.intact() and t.intact(), "lsnIcpRun wrote outside verts2 / R / t"
    want_v, want_R, want_t = orc.icp(c0, c1, max_iter=5)
    got = src.body().cpu().numpy().view(np.float32).reshape(-1, 3)
    assert np.abs(got - want_v).max() < 1e-4          # the north-star's bar
    assert np.abs(R.body().cpu().numpy().view(np.float32).reshape(3, 3) - want_R).max() < 1e-4
    ws.close()
