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
