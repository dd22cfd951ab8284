#!/usr/bin/env python3
"""gf_history_unroll against a plain device copy of the same bytes (development tool).   python tools/bench_unroll.py [N ...]
Both move 2 * N * O * H * 4 bytes; `flush` streams 1 GB through the caches between launches (what the rest of a step does)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genesis-forge_amd"))
import torch
from genesis_forge_amd import _native as nat, gs

gs.set_device("cuda:0")
be = nat.get_backend()
O, H = 62, 5
junk = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda")   # 1 GB


def timed(fn, iters, flush):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for i in range(iters + 3):
        if flush:
            junk.add_(1.0)
        if i >= 3:
            ev[i - 3][0].record()
        fn(i)
        if i >= 3:
            ev[i - 3][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2]


for n in [int(x) for x in sys.argv[1:]] or [65536, 262144]:
    ring = torch.randn(n, H * O, device="cuda")
    outs = [torch.empty_like(ring) for _ in range(3)]
    a = nat.GfHistoryUnrollArgs()
    a.ring, a.num_envs, a.frame_width, a.history_len = ring.data_ptr(), n, O, H

    def unroll(i):
        a.out, a.ring_slot = outs[i % 3].data_ptr(), i % H + 1
        be.call("history_unroll", a)

    def copy(i):
        outs[i % 3].copy_(ring)

    mb = 2 * n * O * H * 4 / 1e6
    for flush in (False, True):
        tu, tc = timed(unroll, 30, flush), timed(copy, 30, flush)
        print(f"N={n} flush={flush}: unroll {tu:7.1f} us ({mb / tu:5.2f} TB/s)   copy_ {tc:7.1f} us ({mb / tc:5.2f} TB/s)   [{mb:.0f} MB]")
