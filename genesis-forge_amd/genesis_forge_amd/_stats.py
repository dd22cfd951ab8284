"""
Per-step logging statistics without per-step host syncs.

The reference produces its ``extras["episode"]`` scalars with blocking calls every step
(``nonzero()`` per termination term, termination_manager.py:178; ``.item()`` per reward term on
reset, reward_manager.py:215).  Here the kernels accumulate into one small device block
(``GfStepStats``); at the end of a step it is copied asynchronously into a pinned host ring slot
and an event is recorded.  ``extras["episode"]`` is a :class:`LazyEpisodeLog` that only waits for
that event if somebody actually reads it.  At world_size > 1 the block is summed over ranks with a
single all-reduce (RCCL over xGMI) before it is interpreted — the only collective of the path.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

import numpy as np
import torch

from . import _native as nat

BLOCK_BYTES = C.sizeof(nat.GfStepStats)
STATS_BYTES = BLOCK_BYTES * nat.GF_STATS_SHARDS  # the device block is GF_STATS_SHARDS shards (see gf_step.h)
_RING = 64
_INTS = BLOCK_BYTES // 4


def sum_shards(buf: bytes) -> nat.GfStepStats:
    """Fold the shards of one statistics read-back: counts and sums add, flag words OR."""
    raw = np.frombuffer(buf, dtype=np.uint8).reshape(nat.GF_STATS_SHARDS, BLOCK_BYTES)
    ints = raw[:, :96].copy().view(np.int32)                       # term_fired[16], reset, action_flags, contact_flags, resample, pad[4]
    f64 = raw[:, 96:96 + 8 * nat.GF_MAX_TERMS].copy().view(np.float64)
    st = nat.GfStepStats()
    tot = ints.sum(axis=0, dtype=np.int64)
    for k in range(nat.GF_MAX_TERM_TERMS):
        st.term_fired[k] = int(tot[k])
    o = nat.GF_MAX_TERM_TERMS
    st.reset_count = int(tot[o])
    st.action_flags = int(np.bitwise_or.reduce(ints[:, o + 1]))
    st.contact_flags = int(np.bitwise_or.reduce(ints[:, o + 2]))
    st.resample_count = int(tot[o + 3])
    sums = f64.sum(axis=0)
    for t in range(nat.GF_MAX_TERMS):
        st.reward_episode_sum[t] = float(sums[t])
    return st



class StatsSnapshot:
    """One step's statistics; ``wait()`` returns a host GfStepStats once the copy has landed."""

    __slots__ = ("_host", "_event", "_value", "_is_vector", "_native")

    def __init__(self, host: torch.Tensor, event, is_vector: bool = False, native=None):
        self._host = host
        self._event = event
        self._value = None
        self._is_vector = is_vector
        self._native = native  # (backend, native event handle) for snapshots taken by a recorded step

    def wait(self) -> nat.GfStepStats:
        if self._value is None:
            if self._native is not None:
                self._native[0].event_synchronize(self._native[1])
            elif self._event is not None:
                self._event.synchronize()
            if self._is_vector:  # cross-rank reduced f64 vector
                st = vector_to_stats(self._host.numpy().copy())
            else:
                st = sum_shards(self._host.numpy().tobytes())
            self._value = st
            self._host = None
            self._event = None
        return self._value


class RingSnapshot:
    """Statistics of one recorded step, living in a device ring slot until somebody reads them."""

    __slots__ = ("_owner", "_slot", "_value")

    def __init__(self, owner, slot: int):
        self._owner, self._slot, self._value = owner, slot, None

    def wait(self) -> nat.GfStepStats:
        if self._value is None:
            self._owner.materialize_ring()
        return self._value


class VecRingSnapshot:
    """Cross-rank reduced statistics of one recorded step: a row of the device vector ring + the pending all-reduce."""

    __slots__ = ("_owner", "_slot", "_work", "_value")

    def __init__(self, owner, slot: int, work):
        self._owner, self._slot, self._work, self._value = owner, slot, work, None

    def wait(self) -> nat.GfStepStats:
        if self._value is None:
            self._owner.materialize_vec_ring()
        return self._value


class StepStats:
    """Owns the device stats block and the pinned read-back ring."""

    def __init__(self, device: torch.device):
        self.device = device
        self.dev = torch.zeros(STATS_BYTES, dtype=torch.uint8, device=device)
        pin = device.type == "cuda"
        self._ring = [torch.zeros(STATS_BYTES, dtype=torch.uint8, pin_memory=pin) for _ in range(_RING)]
        self._live: list[Optional[StatsSnapshot]] = [None] * _RING
        self._slot = 0
        self.group = None       # torch.distributed process group (distributed.attach); None = single process
        self._vring = None

    @property
    def ptr(self) -> int:
        return self.dev.data_ptr()

    def clear(self, backend: nat.Backend) -> None:
        backend.stats_clear(self.ptr)

    def snapshot(self) -> StatsSnapshot:
        """Enqueue the device->host copy of this step's block; never blocks."""
        i = self._slot
        self._slot = (i + 1) % _RING
        old = self._live[i]
        if old is not None and old._value is None:
            old.wait()  # ring wrapped around an unread snapshot: its copy finished long ago
        if self.group is not None:
            snap = self._snapshot_reduced(i)
            self._live[i] = snap
            return snap
        host = self._ring[i]
        if self.device.type == "cuda":
            host.copy_(self.dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            host = self.dev.clone()
            ev = None
        snap = StatsSnapshot(host, ev)
        self._live[i] = snap
        return snap

    # -- device ring used by recorded steps -------------------------------------------------------------
    # A recorded step writes its statistics into slot (step % _RING) of a device-resident ring and zeroes the next
    # slot from inside its first kernel, so it needs neither a memset nor a device→host copy (together they cost as
    # much stream time as all the kernels of a step).  Slots are copied out in one batch only when a log entry is
    # actually read, or just before an unread slot would be recycled.
    def ensure_ring(self) -> None:
        if getattr(self, "ring", None) is None:
            self.ring = torch.zeros(_RING, STATS_BYTES, dtype=torch.uint8, device=self.device)
            self.ring_pos = 0
            self._ring_snaps = [None] * _RING

    def ring_ptr(self, slot: int) -> int:
        return self.ring.data_ptr() + slot * STATS_BYTES

    def ring_next(self):
        """(pointer of this step's slot, pointer of the slot to zero for the next step, snapshot)."""
        i = self.ring_pos
        j = (i + 1) % _RING
        old = self._ring_snaps[j]
        if old is not None and old._value is None:
            self.materialize_ring()  # about to be recycled while still unread
        snap = RingSnapshot(self, i)
        self._ring_snaps[i] = snap
        self.ring_pos = j
        return self.ring_ptr(i), self.ring_ptr(j), snap

    def materialize_ring(self) -> None:
        host = self.ring.cpu().numpy()  # one blocking copy of every slot (synchronises the stream first)
        for snap in self._ring_snaps:
            if snap is not None and snap._value is None:
                snap._value = sum_shards(host[snap._slot].tobytes())

    # -- recorded steps with a process group: shards are folded into a row of a device f64 ring by gf_stats_pack (one op of
    # the recorded step), that row is all-reduced asynchronously (the single collective of the path), and rows are copied
    # out in a batch only when a log entry is read.
    def ensure_vec_ring(self) -> None:
        self.ensure_ring()
        if getattr(self, "vec_ring", None) is None:
            self.vec_ring = torch.zeros(_RING, STATS_VECTOR_LEN, dtype=torch.float64, device=self.device)
            self._vec_snaps = [None] * _RING

    def vec_ptr(self, slot: int) -> int:
        return self.vec_ring.data_ptr() + slot * STATS_VECTOR_LEN * 8

    def vec_ring_next(self):
        """(slot index, shard-slot pointer, next shard-slot pointer, vector-row pointer) for this step."""
        i = self.ring_pos
        j = (i + 1) % _RING
        old = self._vec_snaps[i]
        if old is not None and old._value is None:
            self.materialize_vec_ring()
        self.ring_pos = j
        return i, self.ring_ptr(i), self.ring_ptr(j), self.vec_ptr(i)

    def vec_ring_reduce(self, slot: int) -> "VecRingSnapshot":
        import torch.distributed as dist

        row = self.vec_ring[slot]
        work = dist.all_reduce(row, op=dist.ReduceOp.SUM, group=self.group, async_op=self.device.type == "cuda")
        snap = VecRingSnapshot(self, slot, work)
        self._vec_snaps[slot] = snap
        return snap

    def materialize_vec_ring(self) -> None:
        for snap in self._vec_snaps:
            if snap is not None and snap._value is None and snap._work is not None:
                snap._work.wait()
        host = self.vec_ring.cpu().numpy()
        for snap in self._vec_snaps:
            if snap is not None and snap._value is None:
                snap._value = vector_to_stats(host[snap._slot])
                snap._work = None

    def ensure_native_events(self, backend) -> None:
        if getattr(self, "_events", None) is None:
            self._events = [backend.event_create() for _ in range(_RING)]

    def native_slot(self, copy_args, backend) -> StatsSnapshot:
        """Point a recorded step's GF_OP_STATS_COPY at the next ring slot; returns the snapshot it will fill."""
        i = self._slot
        self._slot = (i + 1) % _RING
        old = self._live[i]
        if old is not None and old._value is None:
            old.wait()
        host = self._ring[i]
        copy_args.dst = host.data_ptr()
        ev = self._events[i]
        copy_args.event = ev
        snap = StatsSnapshot(host, None, native=(backend, ev) if ev is not None else None)
        self._live[i] = snap
        return snap

    def pack_vector(self) -> torch.Tensor:
        """The stats block as one f64 vector on the device (layout of ``stats_to_vector``): the all-reduce payload."""
        blocks = self.dev.view(nat.GF_STATS_SHARDS, BLOCK_BYTES)
        ints = blocks[:, :96].contiguous().view(torch.int32)
        f64 = blocks[:, 96:96 + 8 * nat.GF_MAX_TERMS].contiguous().view(torch.float64).sum(dim=0)
        tot = ints.sum(dim=0)
        flags = ints[:, _NT + 1]
        head = torch.stack([tot[_NT], (flags & 1).max(), ((flags >> 1) & 1).max(), (ints[:, _NT + 2] & 1).max(), tot[_NT + 3]]).to(torch.float64)
        return torch.cat([tot[:_NT].to(torch.float64), head, f64])

    def _snapshot_reduced(self, i: int) -> StatsSnapshot:
        """Sum the block over the ranks of ``self.group`` — the single collective of the path (RCCL over xGMI on
        GPUs; ~350 B, latency bound), enqueued asynchronously behind this step's kernels — then copy it out."""
        import torch.distributed as dist

        vec = self.pack_vector()
        if self._vring is None:
            pin = self.device.type == "cuda"
            self._vring = [torch.zeros(STATS_VECTOR_LEN, dtype=torch.float64, pin_memory=pin) for _ in range(_RING)]
        if self.device.type == "cuda":
            work = dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            work.wait()  # stream-level dependency only: the host does not block
            host = self._vring[i]
            host.copy_(vec, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=self.group)
            host, ev = vec.clone(), None
        return StatsSnapshot(host, ev, is_vector=True)


class LazyEpisodeLog(dict):
    """``extras["episode"]``: a dict whose manager-produced entries appear on first read.

    Producers register ``fill(stats, out_dict)`` callbacks; the first read waits for the step's
    snapshot, runs them once, and from then on this is a plain dict.  Keys written directly
    (custom managers do ``extras["episode"][k] = v``) are kept as they are.
    """

    def __init__(self):
        super().__init__()
        self._snap: Optional[StatsSnapshot] = None
        self._fillers: list[Callable] = []

    def attach(self, snap: StatsSnapshot) -> None:
        self._snap = snap

    def add_filler(self, fn: Callable) -> None:
        self._fillers.append(fn)

    def materialize(self) -> "LazyEpisodeLog":
        if self._fillers:
            fillers, self._fillers = self._fillers, []
            if self._snap is not None:
                st = self._snap.wait()
                tmp: dict = {}
                for fn in fillers:
                    fn(st, tmp)
                for k, v in tmp.items():
                    if not dict.__contains__(self, k):
                        dict.__setitem__(self, k, v)
        return self

    # every read path materialises first
    def __getitem__(self, k):
        return dict.__getitem__(self.materialize(), k)

    def __contains__(self, k):
        return dict.__contains__(self.materialize(), k)

    def __iter__(self):
        return dict.__iter__(self.materialize())

    def __len__(self):
        return dict.__len__(self.materialize())

    def __repr__(self):
        return dict.__repr__(self.materialize())

    def __eq__(self, other):
        return dict.__eq__(self.materialize(), other)

    def keys(self):
        return dict.keys(self.materialize())

    def values(self):
        return dict.values(self.materialize())

    def items(self):
        return dict.items(self.materialize())

    def get(self, k, default=None):
        return dict.get(self.materialize(), k, default)

    def copy(self):
        return dict(self.materialize())


_NT = nat.GF_MAX_TERM_TERMS
STATS_VECTOR_LEN = _NT + 5 + nat.GF_MAX_TERMS


def stats_to_vector(st: nat.GfStepStats) -> np.ndarray:
    """Flatten to f64 for the cross-rank sum (counts are exact in f64; flag bits become counts)."""
    v = np.zeros(STATS_VECTOR_LEN, dtype=np.float64)
    v[:_NT] = list(st.term_fired)
    v[_NT] = st.reset_count
    v[_NT + 1] = st.action_flags & 1
    v[_NT + 2] = (st.action_flags >> 1) & 1
    v[_NT + 3] = st.contact_flags & 1
    v[_NT + 4] = st.resample_count
    v[_NT + 5:] = list(st.reward_episode_sum)
    return v


def vector_to_stats(v: np.ndarray) -> nat.GfStepStats:
    st = nat.GfStepStats()
    for k in range(_NT):
        st.term_fired[k] = int(round(v[k]))
    st.reset_count = int(round(v[_NT]))
    st.action_flags = (1 if v[_NT + 1] > 0 else 0) | (2 if v[_NT + 2] > 0 else 0)
    st.contact_flags = 1 if v[_NT + 3] > 0 else 0
    st.resample_count = int(round(v[_NT + 4]))
    for t in range(nat.GF_MAX_TERMS):
        st.reward_episode_sum[t] = float(v[_NT + 5 + t])
    return st
