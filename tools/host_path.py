#!/usr/bin/env python3
"""Only bench.py's host_path leg (the reference's exports on host arrays), for A/B runs of the library's host flow:

    LSN_HOST_PATH=copy python3 tools/host_path.py        # round 3's flow (device-resident output + copy engine)
    LSN_HOST_GROUP=4 python3 tools/host_path.py          # sensors per upload group of the direct flow (default: by size)
    LSN_HOST_DEVICES=0,0 python3 tools/host_path.py      # merge calls sharded over "two devices" (one GPU listed twice: a rehearsal, not a measurement)

Prints one compact line per variant and the JSON object."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench_support import legs_host  # noqa: E402
from livescan3d_amd import native, synth  # noqa: E402

torch.cuda.set_device(0)
native.require_gpu()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w = int(sys.argv[2]) if len(sys.argv) > 2 else 512
h = int(sys.argv[3]) if len(sys.argv) > 3 else 424
rows = os.environ.get("LSN_HOST_PATH_ROWS")      # e.g. "merge_noise,merge_scene,tick_one_call_scene": only these rows (bench.py's sharded rehearsal)
secs = float(os.environ.get("LSN_HOST_PATH_SECONDS", "1.5"))
out = legs_host.bench_host_path(native, synth, S, w, h, synth.CROP_BOUNDS, only=rows.split(",") if rows else None, seconds=secs, rehearse_sharded=False)
for k, v in out.items():
    if isinstance(v, dict) and "ms_per_call" in v:
        print(f"{k:24s} {v['ms_per_call']:.3f} ms  {v['calls_per_s']:8.1f}/s  up {v['bytes_up'] / 1e6:5.1f} MB down {v['bytes_down'] / 1e6:5.1f} MB  "
              f"half-duplex frac {v['frac_of_pcie_bound']:.2f}  full-duplex bound {v['pcie_full_duplex_bound_ms']:.3f} ms frac {v['frac_of_full_duplex_bound']:.2f}")
print(json.dumps(out))
