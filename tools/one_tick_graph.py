#!/usr/bin/env python3
"""Dev helper (A/B): the three launches of a one-tick plan (count -> scan -> write, 8 x 512x424, device resident) replayed from a captured
graph against the same launches issued eagerly; with the radial correction and the triangulation in front / behind (the whole live tick) as well.
usage: python3 tools/one_tick_graph.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from livescan3d_amd import synth
from livescan3d_amd.fusion import DeviceFusion

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda", 0)
S, w, h = 8, 512, 424
fus = DeviceFusion(1, [w] * S, [h] * S, device=0)
rig = synth.make_rig("scene", S, w, h, seed=3, bounds=synth.CROP_BOUNDS)
fus.set_params(rig.intr, rig.wt, rig.bounds)
d, c = synth.noise_frames_torch(dev, 1, 1, S, w, h)
d, c = d.view(1, -1), c.view(1, -1)
cd, cc = torch.empty_like(d), torch.empty_like(c)
tris = torch.empty((1, 2 * fus.capacity, 3), dtype=torch.int32, device=dev)
toff = torch.zeros((1, S + 1), dtype=torch.int32, device=dev)
side = torch.cuda.Stream()


def fusion(st):
    fus.plan.run(d.data_ptr(), c.data_ptr(), fus.vertices.data_ptr(), fus.offsets.data_ptr(), st)


def tick(st):
    fus.plan.radial_correct_to(rig.intr, d.data_ptr(), c.data_ptr(), cd.data_ptr(), cc.data_ptr(), st)
    fus.plan.run_mesh(cd.data_ptr(), cc.data_ptr(), fus.vertices.data_ptr(), fus.offsets.data_ptr(), tris.data_ptr(), toff.data_ptr(), st)


for name, fn in (("fusion (3 launches)", fusion), ("whole tick (radial -> fusion -> triangles)", tick)):
    with torch.cuda.stream(side):
        st = int(side.cuda_stream)
        for _ in range(20):
            fn(st)
        side.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(st)
        side.synchronize()
        eager = (time.perf_counter() - t0) / reps
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=side):
                fn(int(torch.cuda.current_stream().cuda_stream))
            for _ in range(20):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                g.replay()
            torch.cuda.synchronize()
            graph = (time.perf_counter() - t0) / reps
            print(f"{name:46s} eager {1e6 * eager:7.2f} us/call   graph replay {1e6 * graph:7.2f} us/call", flush=True)
        except Exception as ex:  # noqa: BLE001
            print(f"{name:46s} eager {1e6 * eager:7.2f} us/call   capture failed: {type(ex).__name__}: {str(ex)[:200]}", flush=True)
