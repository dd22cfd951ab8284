"""
One process per GPU, started from a parent that never touches the GPU.

``spawn_ranks(cmd, nprocs)`` starts ``nprocs`` copies of ``cmd`` with the rendezvous environment ``torch.distributed`` reads
(RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT), relays rank 0's stdout line by
line, and returns the first non-zero exit code (terminating the other ranks).  The parent must not have initialised HIP
(``torch.cuda.is_available()`` does): a process that has opened the GPU is never replaced or forked here — the ranks are
fresh interpreters, and the caller exits with the returned code.  This is what ``bench.py --gpus N`` uses when it is not
already running under ``torch.distributed.run``.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
from typing import Optional, Sequence


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank: int, world: int, port: int, base: Optional[dict] = None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL across processes needs dmabuf IPC on this driver
    return env


def spawn_ranks(cmd: Sequence[str], nprocs: int, env: Optional[dict] = None, timeout: Optional[float] = None) -> int:
    """Run ``cmd`` as ranks 0..nprocs-1 of one node; rank 0's stdout goes to ours, every rank's stderr to ours."""
    if nprocs < 1:
        raise ValueError("nprocs must be >= 1")
    port = free_port()
    procs = []
    for r in range(nprocs):
        procs.append(subprocess.Popen(list(cmd), env=rank_env(r, nprocs, port, env), stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=None, text=True))

    def relay():
        for line in procs[0].stdout:
            sys.stdout.write(line)
            sys.stdout.flush()

    t = threading.Thread(target=relay, daemon=True)
    t.start()
    import time

    rc = 0
    deadline = None if timeout is None else time.monotonic() + timeout
    while True:
        codes = [p.poll() for p in procs]
        bad = next((c for c in codes if c not in (None, 0)), None)
        if bad is not None or (deadline is not None and time.monotonic() > deadline):
            rc = bad if bad is not None else 124
            for q in procs:          # a failed rank leaves the others waiting in a collective: stop exactly these children
                if q.poll() is None:
                    q.terminate()
            for q in procs:
                try:
                    q.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    q.kill()
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.02)
    t.join(5)
    return rc
