"""bench_support.launch -- how an N > 1 run of bench.py comes to have its ranks."""
import json
import os
import sys

BENCH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves -- as a CHILD process (never an exec:
    nothing in this process has touched the GPU yet, and nothing will), `python -m torch.distributed.run --nproc-per-node N bench.py <same
    arguments>` on 127.0.0.1 and a free port -- relay rank 0's JSON line and leave with the child's status.  Under an existing launcher
    (WORLD_SIZE set) this is never reached."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL between processes needs on these hosts
    print(f"[bench] --gpus {args.gpus} without a launcher: starting {' '.join(cmd)}", file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in child.stdout:                                # rank 0's line is the only thing the ranks put on stdout; anything else goes by
        if out.lstrip().startswith("{"):
            line = out
        else:
            sys.stderr.write(out)
    rc = child.wait()
    if line is not None:
        sys.stdout.write(line)
        sys.stdout.flush()
    raise SystemExit(rc if rc != 0 or line is not None else 1)


def launch_probe(args):
    """LSN_BENCH_LAUNCH_PROBE=1: the ranks meet over gloo, count each other and rank 0 prints one line -- the launch path of an N > 1 run
    (self_launch or an outer launcher, rendezvous, the relay of the line) without a GPU.  tests/test_sharding_gloo.py."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    seen = torch.ones(1, dtype=torch.int32)
    dist.all_reduce(seen)
    if dist.get_rank() == 0:
        print(json.dumps({"probe": True, "n_gpus": args.gpus, "n_ranks_seen": int(seen.item()), "world_size": dist.get_world_size()}), flush=True)
    dist.barrier()
    dist.destroy_process_group()

