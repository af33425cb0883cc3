"""bench_support.common -- constants and small helpers every leg of bench.py shares."""
import contextlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PCIE_GBS = 63.0             # MI355X_MICROARCH.md: PCIe 5.0 x16, per direction
VALU_F32_PEAK_TF = 157.3    # MI355X_MICROARCH.md: dense fp32 vector peak (FMA, packed)


def settle(fn, sync, seconds, agree=None):
    """Clock settling, separate from --warmup: runs fn back to back for `seconds` of wall time (a fresh box starts a run at idle clocks, and
    a 20-step timed region is over in 6 ms -- before the clocks have moved).  agree (N > 1): turns this rank's "go on" into rank 0's, so
    every rank runs the same number of (collective) steps.  Returns the milliseconds actually spent."""
    t0 = time.perf_counter()
    while True:
        go = time.perf_counter() - t0 < seconds
        if agree is not None:
            go = agree(go)
        if not go:
            break
        for _ in range(8):
            fn()
        sync()
    return 1e3 * (time.perf_counter() - t0)


@contextlib.contextmanager
def leg(result, name):
    """An extra leg of the bench line must never cost the headline: a failure is recorded under its name instead."""
    try:
        yield
    except Exception as ex:  # noqa: BLE001
        result[name] = {"error": f"{type(ex).__name__}: {ex}"}

def _kernel_sources_sha256():
    import hashlib
    hsh = hashlib.sha256()
    for f in ("fusion.hip", "fusion_shared.hpp"):
        hsh.update(open(os.path.join(ROOT, "livescan3d_amd", "csrc", f), "rb").read())
    return hsh.hexdigest()


def pmc_traffic(args, S_loc, B, w, h):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/pmc_traffic.json, written by
    tools/pmc.sh on the GPU box: separate --pmc runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
    None when no pass was recorded for this exact workload."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    key = f"mode{args.mode}-{S_loc}x{w}x{h}-ticks{B}"
    rec = json.load(open(path)).get(key)
    if rec is None or rec.get("kernel_sources_sha256") != _kernel_sources_sha256():
        return None          # no pass for this workload, or the kernel has changed since: a stale counter is not a measurement
    return rec["hbm_bytes_per_launch"]


def host_description():
    """CPU model, core count and how the CPU port was built (BASELINE.md asks for these beside every CPU figure)."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    flags = "unknown"
    try:
        mk = open(os.path.join(ROOT, "oracle", "Makefile")).read()
        flags = [l.split("=", 1)[1].strip() for l in mk.splitlines() if l.startswith("CFLAGS")][0]
    except (OSError, IndexError):
        pass
    return {"cpu_model": model, "nproc": os.cpu_count(), "compiler": "gcc " + flags}

