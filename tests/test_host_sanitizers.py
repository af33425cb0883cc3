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
@pytest.mark.parametrize("devices,parts", [("0,1", 2), ("0,1,1", 3), ("1,0,1,0,1,0,1,0", 8)])
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
