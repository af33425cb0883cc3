#!/usr/bin/env python3
"""Dev helper: one-tick plans (what a live caller holds), device resident: ms per call and the library's kernel time, single pass vs the
single pass (LSN_ONE_TICK_SINGLE_PASS=1), and the single pass on a 64-tick plan (mode 2).  usage: python3 tools/one_tick_driver.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from livescan3d_amd import synth
from livescan3d_amd.fusion import DeviceFusion

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda", 0)


def run(label, T, S, w, h, mode, single_pass=False, kind="noise"):
    os.environ["LSN_ONE_TICK_SINGLE_PASS"] = "1" if single_pass else "0"   # (left alone, a one-tick plan of up to 2048 tiles takes the single pass)
    try:
        fus = DeviceFusion(T, [w] * S, [h] * S, device=0, mode=mode)
    finally:
        os.environ.pop("LSN_ONE_TICK_SINGLE_PASS", None)
    rig = synth.make_rig("noise", S, w, h, seed=1, bounds=synth.CROP_BOUNDS)
    fus.set_params(rig.intr, rig.wt, rig.bounds)
    d, c = synth.noise_frames_torch(dev, 1, T, S, w, h)
    d, c = d.view(T, -1), c.view(T, -1)
    for _ in range(20):
        fus.run(d, c)
    torch.cuda.synchronize()
    n = max(20, reps // T)
    # the step time without the library's kernel timing (its two event records per call are 5-7 us of a one-tick call), best of three;
    # then the same calls with it for the kernel's own time
    dt = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            fus.run(d, c)
        torch.cuda.synchronize()
        dt = min(dt, (time.perf_counter() - t0) / n)
    fus.plan.profile(True)
    fus.plan.kernel_stats(reset=True)
    for _ in range(n):
        fus.run(d, c)
    torch.cuda.synchronize()
    ks = fus.plan.kernel_stats(reset=True)
    fus.plan.profile(False)
    print(f"{label:34s} {1e6 * dt:8.2f} us/step  {ks['kernel']:16s} {1e3 * ks['avg_ms']:8.2f} us  ({int(fus.offsets[0, -1])} vertices in tick 0)", flush=True)


run("8x512x424 x1 tick single pass", 1, 8, 512, 424, 0, single_pass=True)
run("8x512x424 x1 tick three launches", 1, 8, 512, 424, 0)
run("1x512x424 x1 tick single pass", 1, 1, 512, 424, 0, single_pass=True)
run("1x512x424 x1 tick three launches", 1, 1, 512, 424, 0)
run("2x1024x1024 x1 tick single pass", 1, 2, 1024, 1024, 0, single_pass=True)
run("2x1024x1024 x1 tick three launches", 1, 2, 1024, 1024, 0)
run("8x512x424 x1 tick single pass (scene)", 1, 8, 512, 424, 0, single_pass=True)
run("16x1024x1024 x1 tick single pass", 1, 16, 1024, 1024, 0, single_pass=True)
run("16x1024x1024 x1 tick three launches", 1, 16, 1024, 1024, 0)
run("8x512x424 x64 ticks mode 2", 64, 8, 512, 424, 2)
run("8x512x424 x64 ticks two-pass", 64, 8, 512, 424, 0)
