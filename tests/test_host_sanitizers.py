"""The HOST side of libNativeUtils under sanitizers, without a GPU ("race detection / sanitizers").

tests/fake_hip builds every .hip file of the product unchanged for the host alone (clang -x hip --cuda-host-only) against a double of the
HIP runtime -- malloc-backed device and pinned memory, synchronous streams, kernels that return plausible in-bounds counts -- once with
-fsanitize=address,undefined and once with -fsanitize=thread, and links tests/fake_hip/soak_main.cpp: every export with NULL / zero
arguments, random ragged rigs from arrays of exactly the needed length, the inbound parsers (frame messages and recordings: valid, truncated,
bit-flipped, with lying headers -- what arrives from the network and from disk), then LiveScanServer's call mix from four threads at once (merge + tick-as-one-call + last-mesh stream | single-sensor calls |
radial export | ICP), every mesh checked, the pool of pinned blocks empty at the end.  GPU sanitizers do not exist on the pool; the lanes,
the pinned pool, the plan tables and (new this round) the worker threads of the sharded flow are host code, and this is where a race
or a leaked block would be.

The round-3 race -- one plan shared by two lanes -- reintroduced on a scratch copy IS caught by the TSan run (8 reports, all in get_plan;
profiles/r05_tsan_race_demo.txt), which is what makes a clean run here mean something."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_hip")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def soaks():
    if not os.path.exists(CLANG):
        pytest.skip("ROCm's clang is not installed here")
    r = subprocess.run(["make", "-C", FAKE, "-j4", "asan", "tsan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return {k: os.path.join(FAKE, "build", k, "soak") for k in ("asan", "tsan")}


def _run(binary, iters, **env):
    e = {k: v for k, v in os.environ.items() if not k.startswith("LSN_")}
    e.update(ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="halt_on_error=0", **env)
    r = subprocess.run([binary, str(iters)], capture_output=True, text=True, env=e, timeout=600)
    text = r.stdout + r.stderr
    assert "ERROR: AddressSanitizer" not in text and "runtime error:" not in text and "WARNING: ThreadSanitizer" not in text, text[-4000:]
    assert "CHECK failed" not in text, text[-4000:]
    assert r.returncode == 0, text[-4000:]
    summary = [ln for ln in r.stdout.splitlines() if ln.startswith("soak:")]
    assert len(summary) == 1 and "pool 0 live" in summary[0] and "0 check(s) failed" in summary[0], text[-2000:]
    return summary[0]


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_call_mix_from_four_threads(soaks, kind):
    line = _run(soaks[kind], 3)
    assert "over 1 device part(s)" in line


@pytest.mark.parametrize("kind", ["asan", "tsan"])
@pytest.mark.parametrize("devices,parts", [("0,1", 2), ("1,0", 2), ("0,1,1", 3), ("1,0,1,0,1,0,1,0", 8)])
def test_merge_calls_sharded_over_devices(soaks, kind, devices, parts):
    """The sharded flow ($LSN_HOST_DEVICES): one worker thread per device part, counts exchanged through atomics, every part storing into
    the same pinned block at its base -- the double's two devices listed in any order and more than once."""
    line = _run(soaks[kind], 3, LSN_HOST_DEVICES=devices)
    assert f"over {parts} device part(s)" in line


@pytest.mark.parametrize("devices", ["", "0,1,0"])
def test_failed_allocations_leave_nothing_behind(soaks, devices):
    """$LSN_TEST_FAIL_ALLOC=n for a spread of n (first calls, steady state, inside the sharded flow's worker threads): the call that is hit
    returns an empty mesh, everything after it works, the pool ends empty, and neither sanitizer has anything to say."""
    env = {"LSN_HOST_DEVICES": devices} if devices else {}
    for n in (1, 2, 5, 9, 14, 23, 37, 38, 39, 40, 41, 55, 77, 120, 160, 200):
        _run(soaks["asan"], 2, LSN_TEST_FAIL_ALLOC=str(n), **env)
    for n in (3, 40, 90):
        _run(soaks["tsan"], 2, LSN_TEST_FAIL_ALLOC=str(n), **env)


def test_an_exception_at_any_guarded_entry(soaks):
    for n in (1, 5, 12, 30, 31, 32, 60):
        _run(soaks["asan"], 2, LSN_TEST_THROW=str(n))


def _scratch_soak(tmp_path, soaks, edit):
    """An ASan soak built from a scratch copy of the product's sources with `edit(text of host_flows.hip)` applied (only that object is rebuilt:
    the others are taken over from the regular build)."""
    import shutil
    root = tmp_path / "tree"
    shutil.copytree(os.path.join(ROOT, "livescan3d_amd", "csrc"), root / "livescan3d_amd" / "csrc", ignore=shutil.ignore_patterns("build"))
    shutil.copytree(os.path.join(ROOT, "include"), root / "include")
    b = tmp_path / "b"
    shutil.copytree(os.path.join(FAKE, "build", "asan"), b / "asan")              # copy2: the objects keep their times, newer than the copied sources
    f = root / "livescan3d_amd" / "csrc" / "host_flows.hip"
    text = f.read_text()
    edited = edit(text)
    assert edited != text, "the edit did not apply: the product's code has moved on, update this test"
    f.write_text(edited)
    r = subprocess.run(["make", "-C", FAKE, "asan", f"CSRC={root}/livescan3d_amd/csrc", f"B={b}"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return str(b / "asan" / "soak")


def _violations(binary, **env):
    e = {k: v for k, v in os.environ.items() if not k.startswith("LSN_")}
    e.update(ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", **env)
    r = subprocess.run([binary, "2"], capture_output=True, text=True, env=e, timeout=600)
    return [ln for ln in (r.stdout + r.stderr).splitlines() if ln.startswith("CHECK failed: fake hip:")]


def test_the_double_tells_its_two_devices_apart(soaks, tmp_path):
    """First contact of the sharded host flow with two REAL devices, as far as a box without them allows: the runtime double keeps the
    devices' allocations, pinned blocks and streams apart and checks every use against the calling thread's hipSetDevice (fake_hip.cpp,
    "device discipline").  The product passes with two devices in any order (test_merge_calls_sharded_over_devices, both sanitizers);
    here two scratch copies show that the check bites: (1) the mesh block allocated hipHostMallocDefault while shards exist
    (host_flows.hip pinned_get: every device's kernels store into that block), (2) a shard's part run without selecting its device."""
    assert _violations(soaks["asan"], LSN_HOST_DEVICES="0,1") == []
    not_portable = _scratch_soak(tmp_path / "a", soaks, lambda t: t.replace("c.shards.empty() ? hipHostMallocDefault : hipHostMallocPortable", "hipHostMallocDefault"))
    assert _violations(not_portable) == []                                   # one device: nothing to tell apart
    got = _violations(not_portable, LSN_HOST_DEVICES="0,1")
    assert got and all("not hipHostMallocPortable" in ln for ln in got), got[:5]
    import re
    def drop_set_device(t):
        # the first hipSetDevice of the sharded flow's per-part function
        i = t.index("LSN_HIP(hipSetDevice(l.device));", t.index("int shard_part("))
        return t[:i] + "(void)0;" + t[i + len("LSN_HIP(hipSetDevice(l.device));"):]
    wrong_device = _scratch_soak(tmp_path / "b", soaks, drop_set_device)
    # (with "0,1" the omission is harmless by accident: part 0 runs on the calling thread, whose device is 0 already, and device 1's worker
    # thread keeps the device it selected once; "1,0" asks the calling thread for device 1)
    assert _violations(soaks["asan"], LSN_HOST_DEVICES="1,0") == []
    got = _violations(wrong_device, LSN_HOST_DEVICES="1,0")
    assert got and any("of another device" in ln for ln in got), got[:5]
