"""Env sharding + the single logging all-reduce, world_size 2 over gloo on CPU.

Property: a run sharded over ranks reproduces the unsharded run env for env (Philox is keyed by the global
env id) and every rank reports the same, global, log values."""
import os
import socket
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

from _dist_worker import run_probe_shard, run_rank0_reader, run_shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
@pytest.mark.parametrize("reduce_every,read_lag,mutate_at,user_term", [(1, 0, None, False), (8, 11, None, False), (16, 0, 20, False), (4, 3, 17, False),
                                                                       (1, 0, None, True), (8, 2, 19, True), (32, 5, None, False)])
def test_sharded_run_equals_unsharded(oracle_lib_path, reduce_every, read_lag, mutate_at, user_term):
    """reduce_every = K > 1: the statistics rows of K steps travel in one all-reduce (folded by the following step's action
    kernel, no pack launch).  Logs read later than K steps hit closed batches (no extra collective); logs read at once close
    the open batch on both ranks every step; a curriculum mutation drops and re-records the step on both ranks.  With a Python-level
    reward term the recorded step is cut around the call on every rank alike."""
    n_global, steps, sizes = 70, 40, [33, 37]
    with tempfile.TemporaryDirectory() as d1, tempfile.TemporaryDirectory() as d2:
        ctx = mp.get_context("spawn")
        p = ctx.Process(target=run_shard, args=(0, 1, _free_port(), d1, n_global, steps, [n_global], 1, 0, mutate_at, user_term))
        p.start(); p.join(240)
        assert p.exitcode == 0
        port = _free_port()
        procs = [ctx.Process(target=run_shard, args=(r, 2, port, d2, n_global, steps, sizes, reduce_every, read_lag, mutate_at, user_term)) for r in range(2)]
        for q in procs:
            q.start()
        for q in procs:
            q.join(240)
            assert q.exitcode == 0
        full = torch.load(os.path.join(d1, "rank0.pt"))
        shards = [torch.load(os.path.join(d2, f"rank{r}.pt")) for r in range(2)]
    assert all(s["traced"] for s in shards), "the sharded env should still record its step"
    assert all((s["cuts"] == 1) == user_term for s in shards)
    for t in range(steps):
        ref = full["outs"][t]
        for k in range(4):
            cat = torch.cat([s["outs"][t][k] for s in shards])
            assert torch.equal(cat, ref[k]), f"per-env output {k} differs from the unsharded run at step {t}"
        logs = [s["outs"][t][4] for s in shards]
        assert logs[0].keys() == logs[1].keys() == ref[4].keys(), f"log keys differ at step {t}"
        for key, want in ref[4].items():
            for lg in logs:
                assert abs(lg[key] - want) <= 1e-6 + 1e-6 * abs(want), (t, key, lg[key], want)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("reduce_every", [8, 32])
def test_last_episode_means_survive_ring_recycling(oracle_lib_path, reduce_every):
    """Batched group ring, logs unread: rows are recycled without a host copy, and the last-episode means a curriculum reads come
    from the newest ring row that reset something — or, when that row is more than a ring ago, from the device row
    gf_stats_last_reset carried it into.  Same values as the unsharded run (whose single-process ring folds ``last_reset`` in
    the action kernel)."""
    n_global, steps, sizes = 70, 190, [33, 37]
    probes = (60, 183, 189)   # (no probe between the resets of steps 90-110 and step 183: a read would cache the means on the host)
    with tempfile.TemporaryDirectory() as d1, tempfile.TemporaryDirectory() as d2:
        ctx = mp.get_context("spawn")
        p = ctx.Process(target=run_probe_shard, args=(0, 1, _free_port(), d1, n_global, steps, [n_global], 1, probes))
        p.start(); p.join(240)
        assert p.exitcode == 0
        port = _free_port()
        procs = [ctx.Process(target=run_probe_shard, args=(r, 2, port, d2, n_global, steps, sizes, reduce_every, probes)) for r in range(2)]
        for q in procs:
            q.start()
        for q in procs:
            q.join(240)
            assert q.exitcode == 0
        full = torch.load(os.path.join(d1, "rank0.pt"))
        shards = [torch.load(os.path.join(d2, f"rank{r}.pt")) for r in range(2)]
    assert all(s["traced"] for s in shards)
    total = [a + b for a, b in zip(shards[0]["resets"], shards[1]["resets"])]
    assert total == full["resets"]
    last = max(t for t in range(183) if total[t])
    assert 183 - last > 64, "the config should leave the newest reset row more than a ring behind probe 183"
    assert any(v == v and v != 0.0 for v in full["probes"][183][1].values())
    for t in probes:
        want_log, want_means = full["probes"][t]
        for s in shards:
            log, means = s["probes"][t]
            assert log.keys() == want_log.keys()
            for k, v in want_log.items():
                assert abs(log[k] - v) <= 1e-6 + 1e-6 * abs(v), (t, k, log[k], v)
            for k, v in want_means.items():
                assert (means[k] != means[k] and v != v) or abs(means[k] - v) <= 1e-6 + 1e-6 * abs(v), (t, k, means[k], v)   # (nan: a zero-weight term, as in the reference)


@pytest.mark.timeout(300)
def test_rank_local_read_of_an_open_batch_raises_instead_of_hanging(oracle_lib_path):
    """with K > 1 a rank-0-only logger that reads a fresh step's log would enter a collective alone.  Unless the caller
    promised lock-step reads (``attach(lockstep_reads=True)``) the read raises; logs of closed batches stay readable rank-locally."""
    with tempfile.TemporaryDirectory() as d:
        ctx = mp.get_context("spawn")
        port = _free_port()
        procs = [ctx.Process(target=run_rank0_reader, args=(r, 2, port, d, 8)) for r in range(2)]
        for q in procs:
            q.start()
        for q in procs:
            q.join(200)
            assert q.exitcode == 0, "a rank hung or died"
        r0 = torch.load(os.path.join(d, "rank0.pt"))
    assert r0["traced"] and r0["raised"] and r0["old_ok"]


def test_shard_partition():
    from genesis_forge_amd.distributed import shard

    for n, w in [(65536, 8), (70, 3), (5, 8), (8192, 4)]:
        parts = [shard(n, r, w) for r in range(w)]
        assert sum(c for _, c in parts) == n
        pos = 0
        for s, c in parts:
            assert s == pos
            pos += c
