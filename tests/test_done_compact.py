"""gf_done_compact through the raw C ABI: the ascending index list of `mask | mask2` — torch's `nonzero()` of
managed_env.py:308-310 — from two small launches; the oracle twin on the CPU, the HIP kernels on the GPU."""
import pytest
import torch

from genesis_forge_amd import _native as nat


def _args(n, mask, mask2, dev):
    ids = torch.full((max(n, 1),), -7, dtype=torch.int64, device=dev)
    count = torch.zeros(1, dtype=torch.int32)
    if dev != "cpu":
        count = count.pin_memory()
    scratch = torch.zeros((n + 4095) // 4096 + 1, dtype=torch.int32, device=dev)
    a = nat.GfCompactArgs()
    a.num_envs = n
    a.mask, a.mask2 = mask.data_ptr(), (None if mask2 is None else mask2.data_ptr())
    a.ids_out, a.count_out, a.block_counts = ids.data_ptr(), count.data_ptr(), scratch.data_ptr()
    return a, ids, count, scratch


def _cases():
    g = torch.Generator().manual_seed(3)
    for n in (1, 15, 16, 17, 63, 64, 65, 4095, 4096, 4097, 8191, 65536, 65573, 131072, 131073, 1048576 + 37):   # (≤ 131 072: one launch)
        for p in (0.0, 0.002, 0.3, 1.0):
            m1 = torch.rand(n, generator=g) < p
            m2 = (torch.rand(n, generator=g) < p / 2) if n % 2 else None
            yield n, m1, m2


def _check(backend, dev, max_n):
    seen = 0
    for n, m1, m2 in _cases():
        if n > max_n:
            continue
        a, ids, count, _s = _args(n, m1.to(dev), None if m2 is None else m2.to(dev), dev)
        keep = (m1.to(dev), None if m2 is None else m2.to(dev))
        a.mask, a.mask2 = keep[0].data_ptr(), (None if keep[1] is None else keep[1].data_ptr())
        backend.call("done_compact", a)
        if dev != "cpu":
            torch.cuda.synchronize()
        want = (m1 if m2 is None else (m1 | m2)).nonzero().reshape(-1)
        k = int(count[0])
        assert k == want.numel(), (n, k, want.numel())
        assert torch.equal(ids[:k].cpu(), want), f"indices differ at n = {n}"
        assert bool((ids[k:] == -7).all()), "wrote past the list"
        seen += 1
    assert seen >= 44


def test_done_compact_oracle(oracle_backend):
    _check(oracle_backend, "cpu", 70000)
    a = nat.GfCompactArgs()
    with pytest.raises(nat.GfError, match="GF_E_NULL"):
        oracle_backend.call("done_compact", a)


@pytest.mark.gpu
def test_done_compact_hip(hip_backend):
    _check(hip_backend, "cuda", 1 << 21)
    # an unaligned mask view (offset 3 bytes): the element path
    n = 5000
    base = torch.rand(n + 3, device="cuda") < 0.1
    m = base[3:]
    a, ids, count, _s = _args(n, m, None, "cuda")
    hip_backend.call("done_compact", a)
    torch.cuda.synchronize()
    want = m.nonzero().reshape(-1)
    assert int(count[0]) == want.numel() and torch.equal(ids[:want.numel()], want)
    with pytest.raises(nat.GfError, match="GF_E_NULL"):
        hip_backend.call("done_compact", nat.GfCompactArgs())


@pytest.mark.gpu
def test_done_compact_waits_when_asked_hip(hip_backend):
    """``wait = 1``: the call returns with the stream drained — the pinned count is valid without any further synchronisation (what
    genesis_env.DoneIds relies on); queued behind a long-running launch so that an early return would read the stale word."""
    import ctypes as C

    n = 1 << 20
    m = torch.rand(n, device="cuda") < 0.01
    want = m.nonzero().reshape(-1)
    a, ids, _count, _s = _args(n, m, None, "cuda")
    pinned = torch.full((1,), -1, dtype=torch.int32).pin_memory()
    a.count_out, a.wait = pinned.data_ptr(), 1
    big = torch.randn(1 << 26, device="cuda")
    for _ in range(8):
        big = big * 1.0001 + 0.5   # a queue of work in front of the compaction
    hip_backend.call("done_compact", a)
    k = C.c_int32.from_address(pinned.data_ptr()).value   # no torch.cuda.synchronize()
    assert k == want.numel()
    assert torch.equal(ids[:k], want)
