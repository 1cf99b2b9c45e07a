"""One process per GPU, started from a parent that never touches a GPU (no reference counterpart: the reference is a single
process on one GPU, src/Main.cpp:9).

`python bench.py --gpus N` with no launcher around it lands here: the parent only spawns children -- it makes no HIP call and
replaces no running program -- each child gets the torch.distributed environment (RANK, LOCAL_RANK, WORLD_SIZE,
LOCAL_WORLD_SIZE, MASTER_ADDR = 127.0.0.1, MASTER_PORT = a free port) and runs the same command; rank 0's stdout is passed
through as the parent's own, the other ranks' stdout goes to stderr with a rank prefix.  The first child to fail ends the
job: the others are terminated by their exact PIDs and the parent exits with that child's code.
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import threading
import time


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def rank_env(rank: int, world: int, port: int, base=None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PT_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs between processes on this driver
    return env


def _pump(stream, sink, prefix):
    for line in iter(stream.readline, ""):
        sink.write(prefix + line)
        sink.flush()
    stream.close()


def launch(world: int, cmd, timeout_s: float = 3000.0, poll_s: float = 0.05, out=None, err=None) -> int:
    """Run `cmd` (argv list) as `world` ranks; returns the job's exit code (0 only if every rank returned 0)."""
    if world < 1:
        raise ValueError("world < 1")
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    port = free_port()
    procs, pumps = [], []
    for rank in range(world):
        p = subprocess.Popen(list(cmd), env=rank_env(rank, world, port), stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1)
        procs.append(p)
        t = threading.Thread(target=_pump, args=(p.stdout, out if rank == 0 else err, "" if rank == 0 else f"[rank {rank}] "), daemon=True)
        t.start()
        pumps.append(t)
    t0, code = time.monotonic(), 0
    live = set(range(world))
    while live and code == 0:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                code = rc if rc > 0 else 128 - rc        # a signal's number the way a shell reports it
                err.write(f"[launch] rank {r} exited with {rc}; stopping the other ranks\n")
                break
        if live and code == 0:
            if time.monotonic() - t0 > timeout_s:
                err.write(f"[launch] no result after {timeout_s:.0f} s; stopping all ranks\n")
                code = 124
                break
            time.sleep(poll_s)
    for r in sorted(live):                               # only after a failure or the time limit
        if procs[r].poll() is None:
            procs[r].send_signal(signal.SIGTERM)
    for r in sorted(live):
        try:
            procs[r].wait(timeout=10)
        except subprocess.TimeoutExpired:
            procs[r].kill()
            procs[r].wait()
    for t in pumps:
        t.join(timeout=5)
    return code
