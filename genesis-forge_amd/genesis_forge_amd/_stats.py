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
import weakref
from typing import Callable, Optional

import numpy as np
import torch

from . import _native as nat

BLOCK_BYTES = C.sizeof(nat.GfStepStats)
STATS_BYTES = BLOCK_BYTES * nat.GF_STATS_SHARDS  # the device block is GF_STATS_SHARDS shards (see gf_step.h)
_RING = 64
_INTS = BLOCK_BYTES // 4


class HostStats:
    """One step's folded statistics on the host (same field names as GfStepStats)."""

    __slots__ = ("term_fired", "reset_count", "action_flags", "contact_flags", "resample_count", "reward_episode_sum", "gait_count")

    def __init__(self, term_fired, reset_count, action_flags, contact_flags, resample_count, reward_episode_sum, gait_count=None):
        self.term_fired, self.reset_count, self.action_flags = term_fired, reset_count, action_flags
        self.contact_flags, self.resample_count, self.reward_episode_sum = contact_flags, resample_count, reward_episode_sum
        self.gait_count = gait_count if gait_count is not None else np.zeros(nat.GF_MAX_GAITS, dtype=np.int64)


def fold_shards(raw: np.ndarray):
    """Fold the shards of a batch of statistics blocks, ``raw`` = uint8 [B, GF_STATS_SHARDS * BLOCK_BYTES]: counts and
    sums add, flag words OR.  Returns (int64 [B, 24], float64 [B, GF_MAX_TERMS]) — vectorised over the batch."""
    blocks = raw.reshape(raw.shape[0], nat.GF_STATS_SHARDS, BLOCK_BYTES)
    ints = np.ascontiguousarray(blocks[:, :, :96]).view(np.int32)  # term_fired[16], reset, action_flags, contact_flags, resample, gait_count[4]
    f64 = np.ascontiguousarray(blocks[:, :, 96:96 + 8 * nat.GF_MAX_TERMS]).view(np.float64)
    tot = ints.sum(axis=1, dtype=np.int64)
    o = nat.GF_MAX_TERM_TERMS
    tot[:, o + 1] = np.bitwise_or.reduce(ints[:, :, o + 1], axis=1)
    tot[:, o + 2] = np.bitwise_or.reduce(ints[:, :, o + 2], axis=1)
    return tot, f64.sum(axis=1)


def host_stats(tot_row: np.ndarray, sums_row: np.ndarray) -> HostStats:
    o = nat.GF_MAX_TERM_TERMS
    return HostStats(tot_row[:o], int(tot_row[o]), int(tot_row[o + 1]), int(tot_row[o + 2]), int(tot_row[o + 3]), sums_row,
                     tot_row[o + 4:o + 4 + nat.GF_MAX_GAITS])


def sum_shards(buf: bytes) -> HostStats:
    tot, sums = fold_shards(np.frombuffer(buf, dtype=np.uint8).reshape(1, -1))
    return host_stats(tot[0], sums[0])



class StatsSnapshot:
    """One step's statistics; ``wait()`` returns a host GfStepStats once the copy has landed."""

    __slots__ = ("_host", "_event", "_value", "_is_vector", "_native")

    def __init__(self, host: torch.Tensor, event, is_vector: bool = False, native=None):
        self._host = host
        self._event = event
        self._value = None
        self._is_vector = is_vector
        self._native = native  # (backend, native event handle) for snapshots taken by a recorded step

    def wait(self) -> HostStats:
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

    __slots__ = ("_owner", "_slot", "_value", "__weakref__")

    def __init__(self, owner, slot: int):
        self._owner, self._slot, self._value = owner, slot, None

    def wait(self) -> HostStats:
        if self._value is None:
            self._owner.materialize_ring()
        return self._value


class VecRingSnapshot:
    """Cross-rank reduced statistics of one recorded step: a row of the device vector ring + the pending all-reduce."""

    __slots__ = ("_owner", "_slot", "_work", "_value")

    def __init__(self, owner, slot: int, work):
        self._owner, self._slot, self._work, self._value = owner, slot, work, None

    def wait(self) -> HostStats:
        if self._value is None:
            self._owner.materialize_vec_ring()
        return self._value


class GroupRingSnapshot:
    """Statistics of one recorded step of the batched process-group ring (``reduce_every`` > 1): a row of the device vector
    ring that is folded by the following step and all-reduced with its batch.  Held weakly by the ring, like RingSnapshot: a
    log nobody kept is never copied out."""

    __slots__ = ("_owner", "_slot", "_value", "__weakref__")

    def __init__(self, owner, slot: int):
        self._owner, self._slot, self._value = owner, slot, None

    def wait(self) -> HostStats:
        if self._value is None:
            self._owner.materialize_group_ring(self._slot)
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
        #: a recorded step whose tail runs in Python (an env that overrides reset()) points the phase-by-phase launches of that
        #: tail at the step's ring slot, so that the whole step's statistics end up in one block
        self.ptr_override: Optional[int] = None

    @property
    def ptr(self) -> int:
        return self.ptr_override if self.ptr_override is not None else self.dev.data_ptr()

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
    # A recorded step writes its statistics into slot (step % _RING) of a device-resident ring of shard blocks.  Its
    # action kernel zeroes the next slot and folds the PREVIOUS slot into a 49-entry f64 row of ``vec_ring`` (and into
    # ``last_reset`` when that step reset something), so a step carries neither a memset, nor a pack launch, nor a
    # device→host copy — together those cost as much stream time as all the kernels of a step.  Rows are copied out,
    # 23 KB in one batch, only when a log entry is actually read or just before an unread slot is recycled.
    def ensure_ring(self) -> None:
        if getattr(self, "ring", None) is None:
            self.ring = torch.zeros(_RING, STATS_BYTES, dtype=torch.uint8, device=self.device)
            self.vec_ring = torch.zeros(_RING, STATS_VECTOR_LEN, dtype=torch.float64, device=self.device)
            self.last_reset = torch.zeros(STATS_VECTOR_LEN, dtype=torch.float64, device=self.device)
            # every slot's address, once (a recorded step asks for three or four of them: a data_ptr() call each otherwise)
            self._ring_ptrs = [self.ring.data_ptr() + k * STATS_BYTES for k in range(_RING)]
            self._vec_ptrs = [self.vec_ring.data_ptr() + k * STATS_VECTOR_LEN * 8 for k in range(_RING)]
            self.ring_pos = 0
            self._ring_prev = None      # slot written by the previous recorded step (not folded yet)
            self._ring_snaps = [None] * _RING
            self._vec_snaps = [None] * _RING
            self._grp_refs = [None] * _RING   # batched group ring: weak references to the rows' snapshots,
            self._grp_work = [None] * _RING   # the all-reduce that covered a row (True: finished synchronously), None = not reduced
            self._grp_open: list[int] = []    # slots of the open batch, oldest first
            self._fold_slot = None

    def ring_ptr(self, slot: int) -> int:
        return self._ring_ptrs[slot]

    def vec_ptr(self, slot: int) -> int:
        return self._vec_ptrs[slot]

    def ring_next(self):
        """(this step's slot pointer, slot pointer to zero, previous slot pointer or None, its vector row or None, snapshot)."""
        i = self.ring_pos
        ref = self._ring_snaps[i]
        old = ref() if ref is not None else None
        if old is not None and old._value is None:
            self.materialize_ring()  # still referenced by a live extras dict and about to be recycled unread
        prev = self._ring_prev
        snap = RingSnapshot(self, i)
        self._ring_snaps[i] = weakref.ref(snap)  # weak: a log nobody kept is never copied out
        self._ring_prev = i
        self.ring_pos = (i + 1) % _RING
        rp = self._ring_ptrs
        if prev is None:
            return rp[i], rp[self.ring_pos], None, None, snap
        return rp[i], rp[self.ring_pos], rp[prev], self._vec_ptrs[prev], snap

    def _fold_latest(self, backend) -> None:
        """The newest slot has not been folded by a following step yet: fold it explicitly (one tiny launch)."""
        if self._ring_prev is not None:
            backend.stats_pack(self.ring_ptr(self._ring_prev), self.vec_ptr(self._ring_prev))

    def materialize_ring(self) -> None:
        self._fold_latest(nat.get_backend())
        host = self.vec_ring.cpu().numpy()  # one blocking 23 KB copy (synchronises the stream first)
        for ref in self._ring_snaps:
            snap = ref() if ref is not None else None
            if snap is not None and snap._value is None:
                snap._value = vector_to_stats(host[snap._slot])

    def read_last_reset(self) -> Optional["HostStats"]:
        """Statistics of the most recent recorded step that reset at least one env, or None (blocking, on demand)."""
        if getattr(self, "ring", None) is None:
            return None
        if self.group is not None and self.reduce_every > 1:
            return self._group_last_reset()
        if self._ring_prev is None:
            return None
        self._fold_latest(nat.get_backend())
        both = torch.stack([self.vec_ring[self._ring_prev], self.last_reset]).cpu().numpy()
        for row in both:
            if row[_NT] > 0:
                return vector_to_stats(row)
        return None

    def end_recording(self) -> Optional["HostStats"]:
        """Called when a recorded step is dropped: hands back the last reset statistics and forgets the unfolded slot."""
        last = self.read_last_reset()
        if getattr(self, "ring", None) is not None:
            if self.group is not None and self.reduce_every > 1:
                self.materialize_group_ring(None)   # rows are all reduced by now (read_last_reset closed the open batch)
                self._grp_work = [None] * _RING
            else:
                self.materialize_ring()
                self._ring_prev = None
            self.last_reset.zero_()
        return last

    # -- recorded steps with a process group: shards are folded into a row of a device f64 ring by gf_stats_pack (one op of
    # the recorded step), that row is all-reduced asynchronously (the single collective of the path), and rows are copied
    # out in a batch only when a log entry is read.
    def ensure_vec_ring(self) -> None:
        self.ensure_ring()

    def vec_ring_next(self):
        """reduce_every = 1: (slot index, shard-slot pointer, next shard-slot pointer, vector-row pointer) for this step."""
        i = self.ring_pos
        j = (i + 1) % _RING
        old = self._vec_snaps[i]
        if old is not None and old._value is None:
            self.materialize_vec_ring()   # an unread, already reduced row is about to be recycled: copy the ring out (local)
        self.ring_pos = j
        return i, self.ring_ptr(i), self.ring_ptr(j), self.vec_ptr(i)

    #: Rows of the vector ring all-reduced per collective.  1 (default): one all-reduce per recorded step, enqueued behind the
    #: step's kernels — reading a log entry is then purely local, any rank may read any step at any time.  K > 1 (must divide
    #: the ring length, at most half of it): fewer, larger collectives — the rows of K consecutive steps go out as ONE
    #: all-reduce at every K-th step (the per-step host cost of enqueueing a collective dominates a 25 µs step).  Per-step
    #: values are unchanged, but reading a step whose batch is still open closes the batch early — a collective — so with
    #: K > 1 every rank must read the same steps' logs (or call ``flush_reduce()`` at the same step) — see distributed.attach.
    reduce_every = 1

    def vec_ring_reduce(self, slot: int) -> "VecRingSnapshot":
        import torch.distributed as dist

        row = self.vec_ring[slot]
        work = dist.all_reduce(row, op=dist.ReduceOp.SUM, group=self.group, async_op=self.device.type == "cuda")
        snap = VecRingSnapshot(self, slot, work)
        self._vec_snaps[slot] = snap
        return snap

    def materialize_vec_ring(self) -> None:
        for snap in self._vec_snaps:
            if snap is not None and snap._value is None and snap._work is not None and snap._work is not True:
                snap._work.wait()
        host = self.vec_ring.cpu().numpy()
        for snap in self._vec_snaps:
            if snap is not None and snap._value is None:
                snap._value = vector_to_stats(host[snap._slot])
                snap._work = None

    # -- reduce_every = K > 1: the batched group ring ---------------------------------------------------------------------
    # The step is the single-process ring's step (cur / next-to-zero / previous slot and its vector row arrive as call
    # parameters, the action kernel folds the previous slot) — the host adds an integer to a list per step.  Every K-th step
    # one all-reduce covers the K rows folded so far.  A batch is recycled 64 - K steps after its all-reduce was issued: at
    # that point the rows' unread live snapshots (normally none) are copied out and gf_stats_last_reset carries "the newest
    # row that reset something" into ``last_reset`` on the device, so ``last_episode_mean_reward`` needs no host copy of rows
    # nobody asked for.
    def group_ring_next(self):
        """(slot index, this step's slot pointer, slot pointer to zero, previous slot pointer or None, its vector row or None,
        snapshot)."""
        i = self.ring_pos
        if self._grp_work[i] is not None:
            self._recycle_batch(i)
        prev = self._fold_slot
        self._fold_slot = i
        self.ring_pos = j = (i + 1) % _RING
        snap = GroupRingSnapshot(self, i)
        self._grp_refs[i] = weakref.ref(snap)
        if prev is None:
            return i, self.ring_ptr(i), self.ring_ptr(j), None, None, snap
        return i, self.ring_ptr(i), self.ring_ptr(j), self.ring_ptr(prev), self.vec_ptr(prev), snap

    def group_ring_after(self, slot: int) -> None:
        """After step ``slot`` has been enqueued.  The rows of the steps already in the open batch were folded by the action
        kernels of the steps that followed them — the newest of them by the step just enqueued — so a full batch (or one that
        would not stay contiguous across the ring's wrap) goes out now; this step's own row is folded by the next step and
        joins the next batch."""
        pend = self._grp_open
        if pend and (len(pend) >= self.reduce_every or pend[-1] + 1 != slot):
            self._reduce_pending()
        self._grp_open.append(slot)

    def _grp_live(self, slot: int):
        ref = self._grp_refs[slot]
        snap = ref() if ref is not None else None
        return snap if snap is not None and snap._value is None else None

    def _recycle_batch(self, i: int) -> None:
        w = self._grp_work[i]
        n = 1
        while i + n < _RING and self._grp_work[i + n] is w:
            n += 1
        if any(self._grp_live(i + k) is not None for k in range(n)):
            self.materialize_group_ring(None)   # still referenced by a live extras dict and about to be recycled unread (local)
        if w is not True:
            w.wait()   # stream dependency: the fold that rewrites these rows and the pick below run behind the all-reduce
        nat.get_backend().stats_last_reset(self.vec_ptr(i), n, self.last_reset.data_ptr())
        for k in range(n):
            self._grp_work[i + k] = None

    #: reduce_every > 1: may a log read close an open batch by itself?  That is a COLLECTIVE, so it is only safe when every rank reads
    #: the same steps (``distributed.attach(..., lockstep_reads=True)``: training loops that log on every rank, curricula, bench.py).
    #: The default refuses instead of hanging a run whose rank 0 alone reads a fresh log.
    lockstep_reads = False

    def _early_close(self, what: str) -> None:
        if not self.lockstep_reads:
            raise RuntimeError(
                f"{what} is a collective with reduce_every = {self.reduce_every}: every rank would have to do it at the same step, and a "
                "rank-local read (a rank-0-only logger) would wait for the other ranks forever.  Read steps older than reduce_every, call "
                "env.stats.flush_reduce() on EVERY rank first, attach with lockstep_reads=True if all ranks read the same steps, or keep "
                "reduce_every = 1 (rank-local reads)")
        self.flush_reduce()

    def flush_reduce(self) -> None:
        """Close the open batch now (COLLECTIVE: every rank must call it at the same step): the newest pending step has not been
        followed by another recorded step yet, so its shards are folded explicitly, then every pending row is all-reduced."""
        pend = getattr(self, "_grp_open", None)
        if not pend:
            return
        newest = pend[-1]
        if self._fold_slot == newest:
            nat.get_backend().stats_pack(self.ring_ptr(newest), self.vec_ptr(newest))
            self._fold_slot = None   # folded and about to be reduced: the next step must not fold it again
        self._reduce_pending()

    def _reduce_pending(self) -> None:
        import torch.distributed as dist

        pend = self._grp_open
        if not pend:
            return
        i0, i1 = pend[0], pend[-1] + 1
        rows = self.vec_ring[i0:i1]   # contiguous rows of one batch
        work = dist.all_reduce(rows, op=dist.ReduceOp.SUM, group=self.group, async_op=self.device.type == "cuda")
        if work is None:
            work = True
        for s in pend:
            self._grp_work[s] = work
        self._grp_open = []

    def materialize_group_ring(self, want: Optional[int]) -> None:
        """Copy the reduced rows out and fill the snapshots somebody still holds.  ``want`` is the slot being read: if its batch
        is still open the batch is closed first (a collective — the documented cost of reading a fresh step with K > 1);
        whether that happens depends only on WHICH step is read, never on what else happens to be alive on this rank."""
        if want is not None and want in self._grp_open:
            self._early_close("reading the log of a step whose statistics batch is still open")
        seen = []
        for w in self._grp_work:
            if w is not None and w is not True and not any(w is x for x in seen):
                w.wait()
                seen.append(w)
        host = self.vec_ring.cpu().numpy()
        for slot in range(_RING):
            snap = self._grp_live(slot)
            if snap is not None and self._grp_work[slot] is not None:
                snap._value = vector_to_stats(host[slot])

    def _group_last_reset(self) -> Optional["HostStats"]:
        """Batched group ring: global statistics of the most recent recorded step that reset at least one env (COLLECTIVE when a
        batch is open — curricula read this on every rank at the same step)."""
        if getattr(self, "_grp_open", None):
            self._early_close("reading the last episode means while a statistics batch is open")
        seen = []
        for w in self._grp_work:
            if w is not None and w is not True and not any(w is x for x in seen):
                w.wait()
                seen.append(w)
        host = torch.cat([self.vec_ring, self.last_reset[None]]).cpu().numpy()
        for k in range(1, _RING + 1):   # newest first; rows already carried into ``last_reset`` are skipped
            slot = (self.ring_pos - k) % _RING
            if self._grp_work[slot] is not None and host[slot][_NT] > 0:
                return vector_to_stats(host[slot])
        if host[_RING][_NT] > 0:
            return vector_to_stats(host[_RING])
        return None

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
        return torch.cat([tot[:_NT].to(torch.float64), head, f64, tot[_NT + 4:_NT + 4 + nat.GF_MAX_GAITS].to(torch.float64)])

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
_NG = _NT + 5 + nat.GF_MAX_TERMS  # first gait_count entry
STATS_VECTOR_LEN = _NG + nat.GF_MAX_GAITS


def stats_to_vector(st) -> np.ndarray:
    """Flatten to f64 for the cross-rank sum (counts are exact in f64; flag bits become counts)."""
    v = np.zeros(STATS_VECTOR_LEN, dtype=np.float64)
    v[:_NT] = list(st.term_fired)
    v[_NT] = st.reset_count
    v[_NT + 1] = st.action_flags & 1
    v[_NT + 2] = (st.action_flags >> 1) & 1
    v[_NT + 3] = st.contact_flags & 1
    v[_NT + 4] = st.resample_count
    v[_NT + 5:_NG] = list(st.reward_episode_sum)
    v[_NG:] = list(st.gait_count)
    return v


def vector_to_stats(v: np.ndarray) -> HostStats:
    return HostStats(np.rint(v[:_NT]).astype(np.int64), int(round(v[_NT])), (1 if v[_NT + 1] > 0 else 0) | (2 if v[_NT + 2] > 0 else 0),
                     1 if v[_NT + 3] > 0 else 0, int(round(v[_NT + 4])), np.array(v[_NT + 5:_NG], dtype=np.float64),
                     np.rint(v[_NG:]).astype(np.int64))
