"""bench.py is the driver's measurement entry point and only runs to its end on a GPU: what CAN be checked here is that every leg it calls exists
(a leg that fails on the GPU box is recorded as {"error": ...} in the line instead of failing the run -- by design, so a missing function would
go unnoticed until somebody reads the line)."""
import importlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_leg_bench_py_calls_exists():
    src = open(os.path.join(ROOT, "bench.py")).read()
    aliases = {"dl": "bench_support.legs_device", "hl": "bench_support.legs_host", "il": "bench_support.legs_icp"}
    used = set(re.findall(r"\b(dl|hl|il)\.([A-Za-z_][A-Za-z_0-9]*)\(", src))
    assert len(used) >= 10
    for alias, name in sorted(used):
        mod = importlib.import_module(aliases[alias])
        assert callable(getattr(mod, name, None)), f"bench.py calls {alias}.{name} but {aliases[alias]} has no such function"
    for mod_name, names in re.findall(r"from (bench_support\.[a-z_]+) import ([A-Za-z_, ]+)", src):
        mod = importlib.import_module(mod_name)
        for name in [n.strip() for n in names.split(",") if n.strip()]:
            assert hasattr(mod, name), f"{mod_name} lacks {name}"


def test_bench_does_not_import_from_tests():
    for root, _, files in os.walk(os.path.join(ROOT, "bench_support")):
        for f in files:
            if f.endswith(".py"):
                assert not re.search(r"^\s*(from|import) tests\b", open(os.path.join(root, f)).read(), re.M), f
    assert not re.search(r"^\s*(from|import) tests\b", open(os.path.join(ROOT, "bench.py")).read(), re.M)


def test_tools_that_use_the_legs_still_find_them():
    for tool in ("host_path.py", "shard_parts.py"):
        src = open(os.path.join(ROOT, "tools", tool)).read()
        for name in re.findall(r"legs_host\.([A-Za-z_]+)\(", src):
            assert callable(getattr(importlib.import_module("bench_support.legs_host"), name, None)), (tool, name)


def test_call_with_timeout_reports_value_error_and_timeout():
    """bench_support.multi.call_with_timeout: what keeps an N > 1 bench run from waiting for ever inside the library's communicator set-up."""
    import time
    from bench_support.multi import call_with_timeout
    assert call_with_timeout(lambda: 41 + 1, 5.0) == ("ok", 42)
    status, err = call_with_timeout(lambda: (_ for _ in ()).throw(ValueError("no")), 5.0)
    assert status == "error" and isinstance(err, ValueError)
    t0 = time.time()
    assert call_with_timeout(lambda: time.sleep(30), 0.3) == ("timeout", None)
    assert time.time() - t0 < 5.0


def test_bench_line_carries_value_verified_and_fails_on_a_mismatch():
    """bench.py puts `value_verified` beside `value` (and into every `shapes` entry and `full_tick.scene`) and ends non-zero when one of them is not bit-exact."""
    from bench_support.verify import failures
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"value_verified": verified' in src and "failures(result)" in src and "sys.exit(status)" in src and "@checked" in src
    legs = open(os.path.join(ROOT, "bench_support", "legs_device.py")).read()
    assert legs.count('"value_verified": verified') >= 2
    line = {"value": 1.0, "value_verified": {"ticks": [0, 31, 63], "bitexact": True},
            "shapes": {"a": {"value_verified": {"ticks": [0], "bitexact": True}}, "b": {"value_verified": {"ticks": [0], "bitexact": False, "first_mismatch": "tick 0: x"}}},
            "full_tick": {"scene": {"value_verified": {"ticks": [0], "bitexact": True}}}}
    bad = failures(line)
    assert [p for p, _ in bad] == ["shapes.b.value_verified"]
    line["shapes"]["b"]["value_verified"]["bitexact"] = True
    assert failures(line) == []
    line["value_verified"] = {"bitexact": None, "not_checked": "OSError: liblsn_oracle.so"}       # a check that could not run is reported, not failed
    assert failures(line) == []
    from bench_support.verify import checked
    assert checked(lambda: 1 / 0)() == {"bitexact": None, "not_checked": "ZeroDivisionError: division by zero"}
    assert checked(lambda x: {"bitexact": True, "x": x})(3) == {"bitexact": True, "x": 3}


def test_value_verification_catches_a_corrupted_cloud(orc):
    """The checker itself (bench_support/verify.py) on the CPU: the oracle's own cloud passes; one flipped byte, one wrong offset, one missing
    vertex and one wrong triangle index do not."""
    import numpy as np
    from bench_support.verify import compare_cloud, compare_mesh
    from livescan3d_amd import synth
    rig = synth.make_rig("scene", 3, 96, 80, seed=5)
    args = (rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    v, counts = orc.generate_mesh_vertices(*args)
    cloud = v.view(np.uint8).reshape(-1, 16).copy()
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    assert compare_cloud(orc, *args, cloud, off)[0]
    for k, (r, c) in enumerate([(0, 0), (len(cloud) // 2, 7), (len(cloud) - 1, 15)]):
        bad = cloud.copy(); bad[r, c] ^= 1 << (k % 8)
        ok, why = compare_cloud(orc, *args, bad, off)
        assert not ok and "differ" in why
    bad_off = off.copy(); bad_off[1] += 1
    assert not compare_cloud(orc, *args, cloud, bad_off)[0]
    assert not compare_cloud(orc, *args, cloud[:-1], off)[0]
    # the chained tick's checker
    cd, cc = orc.radial_correction(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr)
    cd8 = np.ascontiguousarray(np.asarray(cd)).view(np.uint8).ravel(); cc8 = np.ascontiguousarray(np.asarray(cc)).ravel()
    mv, mcounts, mt = orc.generate_mesh(cd8, cc8, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    moff = np.concatenate([[0], np.cumsum(np.asarray(mcounts).ravel()[:3])]).astype(np.int32)
    mt = np.asarray(mt, np.int32).reshape(-1, 3)
    toff = np.array([0, 0, 0, len(mt)], np.int32)
    good = (cd8.view(np.uint16), cc8, mv.view(np.uint8).reshape(-1, 16), moff, mt, toff)
    assert compare_mesh(orc, *args, *good)[0]
    t2 = mt.copy(); t2[len(t2) // 3, 1] += 1
    assert not compare_mesh(orc, *args, good[0], good[1], good[2], good[3], t2, toff)[0]
    d2 = good[0].copy(); d2[1234] ^= 1
    assert not compare_mesh(orc, *args, d2, *good[1:])[0]
