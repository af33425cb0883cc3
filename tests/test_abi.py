"""The C-ABI library: it loads without a GPU, exports every symbol include/NativeUtils.h declares, keeps the
reference's struct layouts, and refuses to compute without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from livescan3d_amd import native, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "NativeUtils.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", text)))


def test_library_exports_every_declared_symbol():
    L = native.lib()
    names = _declared_functions()
    assert {"generateMeshFromDepthMaps", "generateVerticesFromDepthMap", "createMesh", "deleteMesh", "ICP"} <= set(names)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert set(native.EXPORTS) == set(names)


def _abi_fixture():
    """tests/golden/abi_reference.json: what the reference's own header gives (generator: tests/golden/make_abi_golden.py)."""
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "abi_reference.json")))


def _layout_here():
    """The same 17 numbers as oracle/ref_abi_harness.cpp::ref_abi_layout, from this package's ctypes / numpy mirrors."""
    v = native.VERTEX_DTYPE
    p3 = np.dtype([("X", "<f4"), ("Y", "<f4"), ("Z", "<f4")])     # Point3f as native.icp passes it: n x 3 float32, C order
    return ([v.itemsize] + [v.fields[k][1] for k in "RGBAXYZ"] +
            [C.sizeof(native.Mesh), native.Mesh.nVertices.offset, native.Mesh.vertices.offset, native.Mesh.nTriangles.offset,
             native.Mesh.triangles.offset] + [p3.itemsize] + [p3.fields[k][1] for k in "XYZ"])


def test_struct_layouts_match_the_reference():
    """VertexC4ubV3f / Mesh / Point3f against sizeof / offsetof from the reference's header (depthprocessing.h:29-33,42-48, icp.h:15-18)
    as its compiler lays them out -- the committed fixture, not literals."""
    fx = _abi_fixture()
    assert _layout_here() == fx["layout"]
    assert fx["layout"][0] == 16 and fx["layout"][8] == 32     # Utils.cs: SizeInBytes = 16; Mesh 32 bytes on LP64


def test_sensor_params_packing_matches_the_reference_constructors():
    """lsnPackSensorParams (= what lsnFusionSetParams uploads) against IntrinsicCameraParameters(float*) and WorldTranformation(float*)
    (depthprocessing.h:56-63,96-97) run on position-coded inputs: every float must land where the reference's members read it."""
    fx = _abi_fixture()
    intr = np.array(fx["intr_in"], np.float32)
    wt = np.array(fx["world_in"], np.float32)
    out = np.zeros(16, np.float32)
    assert native.lib().lsnPackSensorParams(intr.ctypes.data, wt.ctypes.data, out.ctypes.data) == 0
    cx, cy, fxx, fy = fx["intr_members_cx_cy_fx_fy_r2_r4_r6"][:4]
    assert out[:4].tolist() == [cx, cy, fxx, fy]
    assert out[4:7].tolist() == fx["world_t"]
    assert out[7:16].tolist() == fx["world_R_rowmajor"]
    # the oracle reads the same arrays the same way (lsn_oracle.c unpacks intr[0..3], wt[0..2], wt[3..11])
    assert fx["icp_default_maxIter"] == 10


def test_live_reference_header_agrees_with_the_fixture():
    """Where oracle/_ref/libref_abi.so exists (built from the reference's header in the build container; travels to the GPU box), the live
    values equal the committed ones, and random parameters unpack identically through the reference's constructors and this library."""
    path = os.path.join(ROOT, "oracle", "_ref", "libref_abi.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libref_abi.so not built (needs /root/reference)")
    R = C.CDLL(path)
    lay = (C.c_int * 32)()
    n = R.ref_abi_layout(lay, 32)
    fx = _abi_fixture()
    assert list(lay[:n]) == fx["layout"] == _layout_here()
    rng = np.random.default_rng(5)
    for _ in range(8):
        intr = rng.standard_normal(7).astype(np.float32)
        wt = rng.standard_normal(12).astype(np.float32)
        o7 = np.zeros(7, np.float32); t3 = np.zeros(3, np.float32); R9 = np.zeros(9, np.float32)
        R.ref_unpack_intrinsics(C.c_void_p(intr.ctypes.data), C.c_void_p(o7.ctypes.data))
        R.ref_unpack_world(C.c_void_p(wt.ctypes.data), 0, C.c_void_p(t3.ctypes.data), C.c_void_p(R9.ctypes.data))
        out = np.zeros(16, np.float32)
        assert native.lib().lsnPackSensorParams(intr.ctypes.data, wt.ctypes.data, out.ctypes.data) == 0
        assert out[:4].tobytes() == o7[:4].tobytes() and out[4:7].tobytes() == t3.tobytes() and out[7:].tobytes() == R9.tobytes()


def test_create_and_delete_mesh_without_gpu():
    L = native.lib()
    m = L.createMesh()
    assert m and m.contents.nVertices == 0 and m.contents.nTriangles == 0 and not m.contents.vertices and not m.contents.triangles
    L.deleteMesh(m)       # frees nothing, must not crash; the struct itself stays with the caller like in the reference
    L.deleteMesh(m)


@pytest.mark.skipif(native.device_count() > 0, reason="only meaningful on a machine without a GPU")
def test_compute_fails_loudly_without_gpu():
    rig = synth.make_rig("noise", 1, 16, 8)
    with pytest.raises(native.NativeUtilsError):
        native.generate_mesh_from_depth_maps(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    with pytest.raises(native.NativeUtilsError):
        native.icp(np.zeros((4, 3), np.float32), np.zeros((4, 3), np.float32))
    with pytest.raises(native.NativeUtilsError):
        native.FusionPlan(0, 1, [16], [8])
    # the outbound formats are built on the device as well: no GPU, no bytes
    with pytest.raises(native.NativeUtilsError):
        native.TransferPacker(0, 16, 16)
    with pytest.raises(native.NativeUtilsError):
        native.last_mesh_ply()
    with pytest.raises(native.NativeUtilsError):
        native.last_mesh_transfer_frame()
    # the raw export leaves an empty mesh and an error message instead of throwing across the ABI
    L = native.lib()
    mesh = native.Mesh()
    w = np.array([16], np.int32); h = np.array([8], np.int32)
    L.generateMeshFromDepthMaps(1, rig.depth_maps.ctypes.data_as(C.c_void_p), rig.depth_colors.ctypes.data_as(C.c_void_p),
                                w.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p), rig.intr.ctypes.data_as(C.c_void_p),
                                rig.wt.ctypes.data_as(C.c_void_p), C.byref(mesh), False, -1.0, -1.0, -1.0, 1.0, 1.0, 1.0, False)
    assert mesh.nVertices == 0 and mesh.nTriangles == 0 and mesh.triangles
    assert "no HIP device" in native.last_error()
    L.deleteMesh(C.byref(mesh))
    assert not mesh.triangles and not mesh.vertices


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under livescan3d_amd/ may import, link or call it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "livescan3d_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "lsn_oracle" not in text and "from oracle" not in text and "import oracle" not in text and "orc_" not in text, f


def test_public_header_is_plain_c_and_cpp():
    """include/NativeUtils.h is the drop-in boundary: it must compile as C99 and as C++11 on its own (no HIP, no torch types)."""
    import subprocess
    hdr = os.path.join(ROOT, "include", "NativeUtils.h")
    subprocess.check_call(["gcc", "-x", "c", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", hdr])
    subprocess.check_call(["g++", "-x", "c++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", hdr])
    includes = [l for l in open(hdr).read().splitlines() if l.strip().startswith("#include")]
    assert all(("<stdbool.h>" in l) or ("<stdint.h>" in l) for l in includes), f"the boundary header pulls in more than <stdbool.h>/<stdint.h>: {includes}"


@pytest.mark.gpu
@pytest.mark.parametrize("n_sensors", [4, 1])
def test_merge_and_single_sensor_calls_from_two_threads(gpu, orc, n_sensors):
    """LiveScanServer's updateWorker (merge calls) and refineWorker (single-sensor calls) run concurrently (MainWindowForm.cs:238,304):
    the two families have their own lanes (streams, buffers, lock, PLANS) inside the library; every result must be the oracle's.
    One sensor: generateMeshFromDepthMaps(n_maps = 1) and generateVerticesFromDepthMap(index 0) describe the same geometry -- a plan
    table shared between the lanes would hand both the same plan and run it on two streams at once."""
    import threading
    from livescan3d_amd import synth
    rig = synth.make_rig("scene", n_sensors, 256, 212, seed=41, bounds=synth.CROP_BOUNDS)
    want_v, counts, want_t = orc.generate_mesh(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    edges = np.concatenate([[0], np.cumsum(counts)])
    errors = []

    def merges():
        try:
            for _ in range(12):
                v, t = native.generate_mesh_from_depth_maps(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
                assert v.tobytes() == want_v.tobytes() and np.array_equal(t, want_t)
        except Exception as ex:  # noqa: BLE001
            errors.append(f"merge: {ex!r}")

    def singles():
        try:
            for rep in range(12):
                i = rep % n_sensors
                v = native.generate_vertices_from_depth_map(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds, i)
                assert v.tobytes() == want_v[edges[i]:edges[i + 1]].tobytes()
        except Exception as ex:  # noqa: BLE001
            errors.append(f"single: {ex!r}")

    def radials():
        try:
            want_d, want_c = orc.radial_correction(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr)
            for _ in range(6):
                d, c = native.radial_correction(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr)
                assert np.array_equal(d, np.asarray(want_d).view(np.uint8).ravel()) and np.array_equal(c, np.asarray(want_c).ravel())
        except Exception as ex:  # noqa: BLE001
            errors.append(f"radial: {ex!r}")

    threads = [threading.Thread(target=f) for f in (merges, singles, radials)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


# ---- nothing throws across the C-ABI (the reference itself lets nanoflann throw, include/nanoflann.h:904) ----------------------

_THROW_SCRIPT = r"""
import ctypes as C, json, sys
import numpy as np
sys.path.insert(0, {root!r})
from livescan3d_amd import native, synth
L = native.lib()
vp = C.c_void_p
rig = synth.make_rig("noise", 1, 16, 8)
w = np.array([16], np.int32); h = np.array([8], np.int32)
out = {{}}
mesh = native.Mesh()
mesh.nVertices = 77
L.generateMeshFromDepthMaps(1, rig.depth_maps.ctypes.data_as(vp), rig.depth_colors.ctypes.data_as(vp), w.ctypes.data_as(vp), h.ctypes.data_as(vp),
                            rig.intr.ctypes.data_as(vp), rig.wt.ctypes.data_as(vp), C.byref(mesh), False, -1.0, -1.0, -1.0, 1.0, 1.0, 1.0, False)
out["merge"] = [mesh.nVertices, mesh.nTriangles, bool(mesh.triangles), native.last_error()]
L.deleteMesh(C.byref(mesh))
L.generateVerticesFromDepthMap(rig.depth_maps.ctypes.data_as(vp), rig.depth_colors.ctypes.data_as(vp), w.ctypes.data_as(vp), h.ctypes.data_as(vp),
                               rig.intr.ctypes.data_as(vp), rig.wt.ctypes.data_as(vp), C.byref(mesh), -1.0, -1.0, -1.0, 1.0, 1.0, 1.0, 0)
out["single"] = [mesh.nVertices, mesh.nTriangles, bool(mesh.triangles), native.last_error()]
L.deleteMesh(C.byref(mesh))
R = np.eye(3, dtype=np.float32).ravel().copy(); t = np.array([1, 2, 3], np.float32)
v1 = np.zeros((4, 3), np.float32); v2 = np.ones((4, 3), np.float32)
L.ICP.restype = C.c_float
r = L.ICP(v1.ctypes.data_as(vp), v2.ctypes.data_as(vp), 4, 4, R.ctypes.data_as(vp), t.ctypes.data_as(vp), 3)
out["icp"] = [float(r), R.tolist(), t.tolist(), v2.ravel().tolist(), native.last_error()]
plan = L.lsnFusionCreate(0, 1, 1, w.ctypes.data_as(vp), h.ctypes.data_as(vp))
out["create"] = [bool(plan), native.last_error()]
L.depthMapAndColorSetRadialCorrection(1, rig.depth_maps.ctypes.data_as(vp), rig.depth_colors.ctypes.data_as(vp), w.ctypes.data_as(vp), h.ctypes.data_as(vp),
                                      rig.intr.ctypes.data_as(vp))
out["radial"] = [native.last_error()]
print(json.dumps(out))
"""


def _run_with_env(script, **env):
    import json
    import subprocess
    import sys
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=e, timeout=300)
    assert r.returncode == 0, f"the process died (rc {r.returncode}): an exception crossed the boundary?\n{r.stderr[-2000:]}"
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.skipif(native.device_count() > 0, reason="written for a machine without a GPU (every compute call ends in 'no HIP device'); "
                                                     "the GPU form is test_a_failed_allocation_leaves_an_empty_mesh_and_the_next_call_works")
@pytest.mark.parametrize("nth", [1, 2, 3, 4, 5])
def test_an_exception_inside_an_export_never_crosses_the_boundary(nth):
    """LSN_TEST_THROW=n makes the n-th guarded entry of the process throw std::bad_alloc from inside the export's body (what a
    growing std::vector / std::map would do under memory pressure).  The caller must see an empty mesh / untouched R, t / a null
    handle and a message -- never std::terminate.  Runs without a GPU: the trampoline sits in front of everything."""
    script = _THROW_SCRIPT.format(root=ROOT)
    base = _run_with_env(script, LSN_TEST_THROW="0")
    got = _run_with_env(script, LSN_TEST_THROW=str(nth))
    # every mesh export leaves an empty mesh with a valid, never-dereferenced triangle pointer, whatever happened inside
    for k in ("merge", "single"):
        assert got[k][:3] == [0, 0, True] and got[k][3], got[k]
    assert got["icp"][0] == 1.0 and got["icp"][1] == base["icp"][1] and got["icp"][2] == [1.0, 2.0, 3.0] and got["icp"][3] == [1.0] * 12
    assert got["create"][0] is False and got["create"][1]
    if nth == 1:   # the first guarded entry of the script is the merge call: it says what hit it
        assert "generateMeshFromDepthMaps: std::bad_alloc" in got["merge"][3], got["merge"]


_ALLOC_SCRIPT = r"""
import ctypes as C, hashlib, json, sys
import numpy as np
sys.path.insert(0, {root!r})
from livescan3d_amd import native, synth
rig = synth.make_rig("scene", 4, 256, 212, seed=41, bounds=synth.CROP_BOUNDS)
out = []
for _ in range(3):
    try:
        v, t = native.generate_mesh_from_depth_maps(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
        out.append([len(v), len(t), hashlib.sha256(v.tobytes() + t.tobytes()).hexdigest(), ""])
    except native.NativeUtilsError as ex:
        out.append([0, 0, "", str(ex)])
print(json.dumps(out))
"""


@pytest.mark.gpu
@pytest.mark.parametrize("nth", [1, 2, 4, 7, 11])
def test_a_failed_allocation_leaves_an_empty_mesh_and_the_next_call_works(gpu, orc, nth):
    """LSN_TEST_FAIL_ALLOC=n: the n-th device / pinned allocation of the process throws std::bad_alloc in the middle of a real merge call.
    That call ends in an empty mesh and a message (never std::terminate, nothing left in flight that could store into a recycled block);
    the calls after it return the oracle's mesh."""
    import hashlib
    from livescan3d_amd import synth
    rig = synth.make_rig("scene", 4, 256, 212, seed=41, bounds=synth.CROP_BOUNDS)
    want_v, _, want_t = orc.generate_mesh(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    want = [len(want_v), len(want_t), hashlib.sha256(want_v.tobytes() + want_t.tobytes()).hexdigest(), ""]
    got = _run_with_env(_ALLOC_SCRIPT.format(root=ROOT), LSN_TEST_FAIL_ALLOC=str(nth))
    failed = [g for g in got if g[3]]
    assert len(failed) <= 1 and all("bad_alloc" in g[3] for g in failed), got
    assert [g for g in got if not g[3]] == [want] * (3 - len(failed)), got
    assert got[-1] == want


_LATE_ALLOC_SCRIPT = r"""
import ctypes as C, hashlib, json, sys
import numpy as np
sys.path.insert(0, {root!r})
from livescan3d_amd import native, synth
rig = synth.make_rig("scene", 8, 256, 212, seed=43, bounds=synth.CROP_BOUNDS)
L = native.lib()
rows = []
for _ in range(5):
    before = int(L.lsnTestFaultPoints(1))
    try:
        if {flow!r} == "tick":
            v, t, _, _ = native.correct_and_generate_mesh(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
        else:
            v, t = native.generate_mesh_from_depth_maps(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
        row = [len(v), len(t), hashlib.sha256(v.tobytes() + t.tobytes()).hexdigest(), ""]
    except native.NativeUtilsError as ex:
        row = [0, 0, "", str(ex)]
    rows.append(row + [before, int(L.lsnTestFaultPoints(1))] + list(native.host_pool_stats()))
print(json.dumps(rows))
"""


@pytest.mark.gpu
@pytest.mark.parametrize("flow", ["tick", "merge"])
def test_an_allocation_that_fails_behind_the_launches_returns_every_block(gpu, flow):
    """A steady-state call allocates nothing but the mesh's pinned blocks -- and in the copy-engine flow (the tick as one call) the triangle
    block is taken AFTER every kernel and vertex copy of the call has been queued.  Each allocation of the third call is made to throw in
    turn: the call must end in an empty mesh with nothing in flight, the pool must hold no block out (a block the failed call kept would
    show as `live`), and the calls after it return the same mesh as before.  (The wrappers copy the mesh and call deleteMesh, so between
    calls nothing is out.)"""
    clean = _run_with_env(_LATE_ALLOC_SCRIPT.format(root=ROOT, flow=flow), LSN_TEST_FAIL_ALLOC="0")
    want = clean[0][:4]
    assert want[0] > 1000 and want[1] > 1000 and not want[3]
    assert all(r[:4] == want and r[6] == 0 for r in clean), clean          # live blocks == 0 after every clean call
    first, last = clean[2][4] + 1, clean[2][5]                             # the fault points the third call passes
    assert last - first + 1 >= 2, clean                                     # at least the vertex and the triangle block
    for nth in range(first, last + 1):
        got = _run_with_env(_LATE_ALLOC_SCRIPT.format(root=ROOT, flow=flow), LSN_TEST_FAIL_ALLOC=str(nth))
        assert [r[:4] for r in got[:2]] == [want, want], (nth, got)
        assert got[2][0] == 0 and "bad_alloc" in got[2][3], (nth, got)
        assert [r[:4] for r in got[3:]] == [want, want], (nth, got)
        assert all(r[6] == 0 for r in got), (nth, got)                      # nothing left out -- after the failed call either
        assert got[4][7] <= clean[4][7] + 1, (nth, got)                     # and the pool did not grow past its steady state


# ---- the upload schedule of the host exports: pure host logic, testable without a GPU ------------------------------------------------

def _parse_schedule(text):
    """'D[0-2] C[0-2] | D[3-7] C[3-5] | C[6-7]' -> list of groups, each a list of (kind, first, last)."""
    groups = []
    for part in text.split("|"):
        runs = []
        for item in part.split():
            m = re.fullmatch(r"([DC])\[(\d+)(?:-(\d+))?\]", item)
            assert m, item
            a = int(m.group(2))
            runs.append((m.group(1), a, int(m.group(3)) if m.group(3) else a))
        groups.append(runs)
    return groups


def test_host_upload_schedule_known_rigs():
    """lsnHostScheduleDescribe on the rigs the numbers in DESIGN.md section 5 were measured on (and a few corners)."""
    S = native.host_schedule
    assert S([512] * 8, [424] * 8) == (3, "D[0-2] C[0-2] | D[3-7] C[3-5] | C[6-7]")                  # merge call: groups follow the >= 1 MiB depth runs
    assert S([512] * 8, [424] * 8, radial=True) == (2, "D[0-2] C[0-2] | D[3-7] C[3-7]")              # tick as one call / radial export: first group >= 1.9 MB of colours, then >= 2.5 MB
    assert S([512] * 8, [424] * 8, radial=2) == (2, "D[0-3] C[0-3] | D[4-7] C[4-7]")                 # the radial export alone: equal groups of >= 2.5 MB of colours
    assert S([512] * 16, [424] * 16, radial=True) == (4, "D[0-2] C[0-2] | D[3-6] C[3-6] | D[7-10] C[7-10] | D[11-15] C[11-15]")
    assert S([512] * 8, [424] * 8, first=3, count=1) == (1, "D[3] C[3]")                             # generateVerticesFromDepthMap(index 3)
    assert S([1024] * 3, [768] * 3) == (3, "D[0] C[0] | D[1] C[1] | D[2] C[2]")                      # big frames: a group per sensor
    assert S([512] * 8, [424] * 8, sensors_per_group=1) == (8, "D[0-2] C[0] | C[1] | C[2] | D[3-7] C[3] | C[4] | C[5] | C[6] | C[7]")
    assert S([250] * 5, [121] * 5)[0] == 1                                                           # slices that break the wide-load alignment: one group
    g, text = S([1024] * 40, [1024] * 40)
    assert g <= 16 and text.startswith("D[0-2] C[0-2] |")                                            # more big sensors than group events: regrouped
    with pytest.raises(native.NativeUtilsError):
        S([512] * 2, [424] * 2, first=1, count=2)


def test_host_upload_schedule_properties():
    """For random rigs: every sensor's depth and colours go up exactly once and in order; a group is launched only after all of its own
    frames (a depth run may run ahead of its group, never behind); runs below 1 MiB only occur where nothing could be merged."""
    rng = np.random.default_rng(11)
    for _ in range(300):
        n = int(rng.integers(1, 25))
        if rng.random() < 0.5:
            w = [int(rng.choice([128, 256, 512, 640, 1024]))] * n
            h = [int(rng.choice([96, 212, 424, 560, 768]))] * n
        else:
            w = [int(rng.choice([61, 100, 250, 256, 512, 1024])) for _ in range(n)]
            h = [int(rng.choice([3, 37, 120, 212, 424])) for _ in range(n)]
        first = int(rng.integers(0, n))
        count = int(rng.integers(1, n - first + 1))
        radial = bool(rng.integers(0, 2))
        per = int(rng.choice([0, 0, 0, 1, 2, 3, 5]))
        g, text = native.host_schedule(w, h, first=first, count=count, radial=radial, sensors_per_group=per)
        groups = _parse_schedule(text)
        assert len(groups) == g >= 1
        seen = {"D": first, "C": first}           # next sensor expected per array
        done_c = first
        for runs in groups:
            for kind, a, b in runs:
                assert a == seen[kind] and b >= a, (text, kind, a)
                seen[kind] = b + 1
            # the group that becomes ready here: its sensors are those whose colours ended in this part; their depth must be there
            assert any(k == "C" for k, _, _ in runs), text
            assert seen["C"] > done_c and seen["D"] >= seen["C"], text
            done_c = seen["C"]
        assert seen == {"D": first + count, "C": first + count}, text
        if per == 0 and g > 1:
            for runs in groups[:-1]:
                for kind, a, b in runs:
                    nbytes = sum(w[i] * h[i] for i in range(a, b + 1)) * (2 if kind == "D" else 3)
                    assert nbytes >= (1 << 20) or kind == "C" and not radial, (text, kind, a, b, nbytes)


_NULL_ARGS_SCRIPT = r"""
import ctypes as C, re, sys
sys.path.insert(0, {root!r})
from livescan3d_amd import native
L = native.lib()
text = re.sub(r"/\*.*?\*/", "", open({header!r}).read(), flags=re.S)
n = 0
for ret, name, params in re.findall(r"\b([A-Za-z_][A-Za-z0-9_ \*]*?)\b([A-Za-z_][A-Za-z0-9_]*)\s*\(([^;{{}}]*)\)\s*;", text):
    ps = [p.strip() for p in params.replace("\n", " ").split(",") if p.strip()]
    if ps == ["void"]:
        ps = []
    kinds = ["p" if "*" in p else "f" if re.search(r"\bfloat\b", p) else "q" if re.search(r"\blong long\b", p) else "b" if re.search(r"\bbool\b", p) else "i" for p in ps]
    f = getattr(L, name)
    f.argtypes = [dict(p=C.c_void_p, f=C.c_float, q=C.c_longlong, b=C.c_bool, i=C.c_int)[k] for k in kinds]
    ret = ret.strip()
    f.restype = C.c_void_p if "*" in ret else C.c_float if "float" in ret else C.c_longlong if "long long" in ret else None if ret == "void" else C.c_int
    print(name, flush=True)
    f(*[None if k == "p" else 0.0 if k == "f" else 0 for k in kinds])
    n += 1
print("ALL", n)
"""


def test_every_export_survives_null_and_zero_arguments():
    """Each of the exports include/NativeUtils.h declares, called with NULL for every pointer and 0 for every number (no GPU needed):
    none may crash the process -- a P/Invoke caller that passes a wrong handle gets an error, not an access violation."""
    import subprocess
    import sys as _sys
    r = subprocess.run([_sys.executable, "-c", _NULL_ARGS_SCRIPT.format(root=ROOT, header=os.path.join(ROOT, "include", "NativeUtils.h"))],
                       capture_output=True, text=True, timeout=300)
    lines = r.stdout.strip().splitlines()
    assert r.returncode == 0 and lines and lines[-1].startswith("ALL"), f"crashed in {lines[-1] if lines else '?'} (rc {r.returncode})\n{r.stderr[-1500:]}"
    assert int(lines[-1].split()[1]) == len(native.EXPORTS)


# ---- merge calls sharded over the devices of $LSN_HOST_DEVICES: the split is pure host logic -------------------------------------------

def test_host_shard_split_known_cases():
    H = native.host_shards
    assert H(8, 2) == ([0, 4, 8], "0:[0-3] 1:[4-7]")                        # BASELINE configs[2]'s rig on two devices
    assert H(8, 8) == ([0, 1, 2, 3, 4, 5, 6, 7, 8], " ".join(f"{d}:[{d}]" for d in range(8)))   # configs[3]: one sensor per device
    assert H(16, 8)[0] == list(range(0, 17, 2))                              # configs[4]: two per device
    assert H(3, 8) == ([0, 1, 2, 3], "0:[0] 1:[1] 2:[2]")                     # more devices than sensors: the spare ones stay idle
    assert H(8, 3) == ([0, 2, 5, 8], "0:[0-1] 1:[2-4] 2:[5-7]")
    assert H(5, 1) == ([0, 5], "0:[0-4]")
    with pytest.raises(native.NativeUtilsError):
        H(0, 2)


def test_host_shard_split_properties():
    """For every sensor count and device count: the blocks are contiguous, in sensor order (= the order their vertices take in the Mesh:
    formMesh concatenates the sensors in index order, depthprocessing.cpp:1594-1608), cover every sensor once, none is empty, and their
    sizes differ by at most one; and the vertex / triangle bases the devices derive from each other's counts are the exclusive prefix sums
    in that order (restated here on random counts: what ShardedCall::base_of computes)."""
    rng = np.random.default_rng(11)
    for n in list(range(1, 40)) + [64, 100]:
        for dev in range(1, 20):
            first, text = native.host_shards(n, dev)
            D = len(first) - 1
            assert D == min(dev, n, 16) and first[0] == 0 and first[-1] == n
            sizes = np.diff(first)
            assert (sizes >= 1).all() and sizes.max() - sizes.min() <= 1
            assert len(text.split()) == D
            counts = rng.integers(0, 217088, size=n)
            per_dev = [int(counts[first[d]:first[d + 1]].sum()) for d in range(D)]
            bases = [sum(per_dev[:d]) for d in range(D)]
            for d in range(D):   # device d's first vertex = the vertices of every sensor before its block
                assert bases[d] == int(counts[:first[d]].sum())


def test_host_devices_environment_is_validated(tmp_path):
    """A device list that names a device the machine does not have (here: none at all, or one beyond the count) must end in an error text,
    not in a crash or a silent one-device run."""
    script = (f"import sys; sys.path.insert(0, {ROOT!r}); from livescan3d_amd import native\n"
              "print(native.host_shards(8, 0))\n")
    import subprocess, sys as _sys
    e = dict(os.environ, LSN_HOST_DEVICES="0,0,0")
    r = subprocess.run([_sys.executable, "-c", script], capture_output=True, text=True, env=e, timeout=120)
    assert r.returncode == 0 and "0:[0-1] 1:[2-4] 2:[5-7]" in r.stdout, r.stderr[-1000:]
    e = dict(os.environ, LSN_HOST_DEVICES="0,x")
    r = subprocess.run([_sys.executable, "-c", script], capture_output=True, text=True, env=e, timeout=120)
    assert r.returncode != 0 and "LSN_HOST_DEVICES" in r.stderr
