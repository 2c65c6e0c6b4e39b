"""Rank launcher for the entry points (`bench.py`, `Train_Agent.py`, `Train_Geo.py`): `--gpus N` starts N ranks.

The path shards by batch: one process per GPU.  When `--gpus N > 1` is given and no launcher has set
WORLD_SIZE, the calling process becomes a pure PARENT: it starts ONE child,
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P <script> <argv>`,
relays what the ranks print (JSON lines to stdout, everything else to stderr) and exits with the child's return code.
It must be called BEFORE anything initialises the GPU in the parent (no torch.cuda call, no HIP call): a process that has
touched the device is never replaced or forked here -- the child is a fresh interpreter.

This module imports neither torch nor the HIP library.
"""
import os
import socket
import subprocess
import sys


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def requested_gpus(argv):
    """--gpus N / --gpus=N from an argument list (1 if absent); no argparse so that it can run before the heavy imports."""
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    return n


def rank_command(script, argv, gpus, port=None):
    """The child command line (also what a user would type to launch the ranks by hand)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script] + list(argv)


def is_rank_process():
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def spawn_ranks_if_needed(script, argv=None):
    """Call first thing in an entry point.  Returns normally in a rank process (or when --gpus is 1); otherwise runs the
    ranks as a child and exits with its return code."""
    argv = list(sys.argv[1:] if argv is None else argv)
    gpus = requested_gpus(argv)
    if gpus <= 1 or is_rank_process():
        return
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // gpus)))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL needs it on this host driver
    cmd = rank_command(os.path.abspath(script), argv, gpus)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, bufsize=1)
    for line in proc.stdout:                                 # rank 0's JSON line(s) -> stdout, anything else -> stderr
        s = line.strip()
        if s.startswith("{") and s.endswith("}"):
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            sys.stderr.write(line)
            sys.stderr.flush()
    sys.exit(proc.wait())
