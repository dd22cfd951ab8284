"""
Term programs: the host-side compiler from the reference's operator schema
(``{name: {"fn": callable, "params": dict, "weight" | "time_out": …}}``; reward_manager.py:10-20,
termination_manager.py:9-19) to the native term tables of include/gf_step.h.

* every ``mdp.rewards.*`` / ``mdp.terminations.*`` callable carries a ``_gf_spec(env, **params)``
  that describes it as a :class:`TermSpec` (opcode + parameters + which buffers it reads);
* anything else (lambdas, bound methods of user managers, terms on a second entity) becomes an
  EXTERNAL column: the callable is evaluated by Python exactly as the reference would and its
  ``[N]`` result is handed to the kernel — "drop in unchanged" stays true for opaque terms;
* weights and params are re-read whenever the owning manager is marked dirty (curricula mutate
  ``cfg[name].weight`` / ``.params`` at run time, docs/guide/managers/reward.md:132-171).

The same programs back the direct calls of ``mdp.*`` functions (EVAL mode, one-term table).
"""
from __future__ import annotations

from typing import Any, Callable, Optional

import torch

from .. import _native as nat
from .. import gs


class TermSpec:
    """Symbolic description of one term; slots are resolved by the program that owns it."""

    def __init__(self, op: int, p=(), i=(), flags: int = 0, entity=None, action_manager=None, needs_actions: bool = False,
                 needs_terminated: bool = False, cmd: Optional[dict] = None, contact: Optional[dict] = None, ext: Optional[dict] = None,
                 state: Optional[dict] = None, link_vel: bool = False, link_pos: bool = False, after: Optional[Callable[[], None]] = None, terrain=None):
        self.op = op
        self.p = list(p) + [0.0] * (4 - len(p))
        self.i = list(i) + [0] * (4 - len(i))
        self.flags = flags
        self.entity = entity
        self.action_manager = action_manager
        self.needs_actions = needs_actions
        self.needs_terminated = needs_terminated
        self.cmd = cmd or {}          # {index into i[]: CommandManager | Tensor}
        self.contact = contact or {}  # {index into i[]: ContactManager}
        self.ext = ext or {}          # {index into i[]: () -> Tensor}
        self.state = state or {}      # {index into i[]: Tensor [N,6]}
        self.link_vel = link_vel
        self.link_pos = link_pos      # the term reads the tracked links' world positions (GfContactView.link_pos)
        self.after = after
        self.terrain = terrain        # TerrainManager whose map the term samples in-kernel (GfRewardArgs.terrain)


def _col(t: torch.Tensor, n: int, dtype) -> torch.Tensor:
    """Normalise a Python-evaluated term result to a contiguous device column."""
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t)
    if t.device != gs.device:
        t = t.to(gs.device)
    if t.dtype != dtype:
        t = t.to(dtype)
    t = t.reshape(n, -1) if t.dim() > 1 else t
    return t.contiguous()


class _Slots:
    """Deduplicating slot allocator for command views, contact views, ext columns and state buffers."""

    def __init__(self, env):
        self.env = env
        self.cmds: list = []
        self.contacts: list = []   # [manager, need_link_vel]
        self.exts: list = []
        self.states: list = []
        self.terrain = None        # the one TerrainManager a phase may sample
        #: True when a launch bound a per-step temporary the recorded step cannot refresh (link velocities fetched through a
        #: Genesis getter, an external command controller, a converted copy of a command tensor): such a descriptor must never
        #: be frozen into a recorded step.  Python-evaluated columns (``exts``) are NOT volatile in this sense: the recorded step
        #: re-evaluates them at the point of the step where the ordinary path does (rebind_exts, _trace.StepTrace.splits).
        self.volatile = False

    def cmd(self, src) -> int:
        for k, s in enumerate(self.cmds):
            if s is src:
                return k
        if len(self.cmds) >= nat.GF_MAX_COMMAND_VIEWS:
            raise RuntimeError(f"a fused phase can read at most {nat.GF_MAX_COMMAND_VIEWS} command buffers")
        self.cmds.append(src)
        return len(self.cmds) - 1

    def contact(self, mgr, link_vel: bool, link_pos: bool = False) -> int:
        for k, s in enumerate(self.contacts):
            if s[0] is mgr:
                s[1] = s[1] or link_vel
                s[2] = s[2] or link_pos
                return k
        if len(self.contacts) >= nat.GF_MAX_CONTACT_VIEWS:
            raise RuntimeError(f"a fused phase can read at most {nat.GF_MAX_CONTACT_VIEWS} ContactManagers")
        self.contacts.append([mgr, link_vel, link_pos])
        return len(self.contacts) - 1

    def ext(self, provider) -> int:
        if len(self.exts) >= nat.GF_MAX_EXT:
            raise RuntimeError(f"a fused phase can take at most {nat.GF_MAX_EXT} Python-evaluated terms")
        self.exts.append(provider)
        return len(self.exts) - 1

    def state(self, t) -> int:
        if len(self.states) >= 4:
            raise RuntimeError("at most 4 stateful terms per RewardManager")
        self.states.append(t)
        return len(self.states) - 1

    def resolve(self, spec: TermSpec, term: nat.GfTerm) -> None:
        for j in range(4):
            term.p[j] = float(spec.p[j])
            term.i[j] = int(spec.i[j])
        for j, src in spec.cmd.items():
            term.i[j] = -1 if src is None else self.cmd(src)
        for j, mgr in spec.contact.items():
            term.i[j] = self.contact(mgr, spec.link_vel, spec.link_pos)
        for j, prov in spec.ext.items():
            term.i[j] = self.ext(prov)
        for j, t in spec.state.items():
            term.i[j] = self.state(t)
        if spec.terrain is not None:
            if self.terrain is not None and self.terrain is not spec.terrain:
                raise RuntimeError("a fused phase can sample one TerrainManager")
            self.terrain = spec.terrain
        term.op = spec.op
        term.flags = spec.flags

    def bind(self, args, ext_dtype, keep: list) -> None:
        """Refresh the pointers of every slot for this launch."""
        n = self.env.num_envs
        self.volatile = False
        for k, src in enumerate(self.cmds):
            if getattr(src, "_external_controller", None) is not None:
                self.volatile = True
            if hasattr(src, "_gf_command_view"):  # a manager that owns strided state rows (GaitCommandManager)
                src._gf_command_view(args.command[k], args)
                continue
            t = src.command if hasattr(src, "command") else src
            if t.dim() == 1:
                t = t.unsqueeze(-1)
            if isinstance(t, torch.Tensor) and t.dim() == 2 and t.shape[0] == n and t.dtype == torch.float32 and t.device == gs.device \
                    and (t.shape[1] == 1 or t.stride(1) == 1) and t.stride(0) >= t.shape[1] and t.data_ptr() % 4 == 0:
                # read in place — also a column / slice VIEW of a wider tensor (examples/simple/environment.py:160,168 pass
                # `self.target_command[:, :2]` and `[:, 2]`): like the reference, later in-place edits of the base stay visible
                keep.append(t)
                args.command[k].command = t.data_ptr()
                args.command[k].width = t.shape[1]
                args.command[k].stride = 0 if t.stride(0) == t.shape[1] else t.stride(0)
                continue
            t = _col(t, n, torch.float32)   # an exotic layout or dtype: a converted copy, valid for this launch only
            self.volatile = True
            keep.append(t)
            args.command[k].command = t.data_ptr()
            args.command[k].width = t.shape[1]
            args.command[k].stride = 0
        if hasattr(args, "contact"):
            for k, (mgr, lv, lp) in enumerate(self.contacts):
                tmp = mgr.view(args.contact[k], need_link_vel=lv, need_link_pos=lp)
                if tmp:
                    self.volatile = True  # link velocities are a fresh tensor every step
                keep.extend(tmp)
        self.rebind_exts(args, ext_dtype, keep)
        if hasattr(args, "state"):
            for k, t in enumerate(self.states):
                args.state[k] = t.data_ptr()
        if self.terrain is not None and hasattr(args, "terrain"):
            self.terrain.gf_view(args.terrain)


def _rebind_exts(self, args, ext_dtype, keep: list) -> None:
    """Evaluate every Python-level term now (exactly the call the reference makes at this point of the step) and hand the
    columns to the descriptor."""
    n = self.env.num_envs
    for k, prov in enumerate(self.exts):
        t = _col(call_untraced(self.env, prov), n, ext_dtype)
        keep.append(t)
        args.ext[k] = t.data_ptr()


def call_untraced(env, fn):
    """Run a user callable with step recording suspended: native launches it makes itself (an EntityManager getter, a direct
    ``mdp.*`` call) belong to the callable — the recorded step re-runs the callable, not a frozen copy of its launches."""
    backend = env.backend
    tracer, backend.tracer = backend.tracer, None
    try:
        return fn()
    finally:
        backend.tracer = tracer


_Slots.rebind_exts = _rebind_exts


def spec_of(fn, env, params) -> Optional[TermSpec]:
    """TermSpec of a library term, or None for an opaque callable."""
    maker = getattr(fn, "_gf_spec", None)
    if maker is None:
        return None
    return maker(env, **params)


class RewardProgram:
    """Compiles reward terms into a GfRewardArgs and launches gf_reward_step."""

    def __init__(self, env):
        self.env = env
        self.args = nat.GfRewardArgs()
        self.slots = _Slots(env)
        self.entity = None
        self.action_manager = None
        self.needs_actions = False
        self.needs_terminated = False
        self.after: list = []
        self.n = 0

    def add(self, spec: Optional[TermSpec], fallback: Callable[[], torch.Tensor], w: float, row: int) -> None:
        if self.n >= nat.GF_MAX_TERMS:
            raise RuntimeError(f"at most {nat.GF_MAX_TERMS} reward terms")
        if spec is not None:
            if spec.entity is not None:
                if self.entity is None:
                    self.entity = spec.entity
                elif spec.entity is not self.entity:
                    spec = None  # second entity: evaluate through its own launch, feed as a column
            if spec is not None and spec.action_manager is not None:
                if self.action_manager is None:
                    self.action_manager = spec.action_manager
                elif spec.action_manager is not self.action_manager:
                    spec = None
        if spec is None:
            spec = TermSpec(nat.GF_R_EXTERNAL, ext={0: fallback})
        term = self.args.terms[self.n]
        self.slots.resolve(spec, term)
        term.w = w
        term.row = row
        self.needs_actions |= spec.needs_actions
        self.needs_terminated |= spec.needs_terminated
        if spec.after is not None:
            self.after.append(spec.after)
        self.n += 1
        self.args.num_terms = self.n

    def launch(self) -> None:
        env, a = self.env, self.args
        keep: list = []
        a.num_envs = env.num_envs
        a.dt = float(env.dt)
        if self.entity is not None:
            env.entity_views(self.entity).fill(a.entity)
        if self.action_manager is not None:
            am = self.action_manager
            dof = am._scene_dofs("position")
            keep.append(dof)
            a.num_dofs = am.num_actions
            a.dof_pos = dof.data_ptr()
            a.default_dof_pos = am._k_default.data_ptr()
        if self.needs_actions:
            if env.actions is None:
                raise RuntimeError("action_rate_l2 needs env.actions; call env.reset() / env.step() first")
            a.num_dofs = env.actions.shape[1]
            a.actions = env.actions.data_ptr()
            a.last_actions = env.last_actions.data_ptr()
        if self.needs_terminated:
            t = env.extras["terminations"]
            keep.append(t)
            a.terminated = t.data_ptr()
        self.slots.bind(a, torch.float32, keep)
        env.backend.call("reward_step", a, owner=self)
        for fn in self.after:
            fn()
        self._keep = keep


def same_structure(old, new) -> bool:
    """Two compiled programs (RewardProgram / TerminationProgram) that differ at most in NUMBERS — term weights and params: same terms in
    the same order with the same opcodes, flags, rows and view slots, the same entity / action manager, the same command / contact /
    state / terrain sources, as many Python-level terms.  Then ``old``'s descriptor — the one a recorded step froze — can take ``new``'s
    term table as it is (refresh_terms)."""
    if old is None or new is None or type(old) is not type(new) or old.n != new.n or old.entity is not new.entity:
        return False
    for name in ("action_manager",):
        if getattr(old, name, None) is not getattr(new, name, None):
            return False
    for name in ("needs_actions", "needs_terminated"):
        if getattr(old, name, False) != getattr(new, name, False):
            return False
    if len(getattr(old, "after", ())) != len(getattr(new, "after", ())):
        return False
    so, sn = old.slots, new.slots
    if so.volatile or sn.volatile or so.terrain is not sn.terrain or len(so.exts) != len(sn.exts):
        return False
    if len(so.cmds) != len(sn.cmds) or any(a is not b for a, b in zip(so.cmds, sn.cmds)):
        return False
    if len(so.contacts) != len(sn.contacts) or any(a[0] is not b[0] or a[1:] != b[1:] for a, b in zip(so.contacts, sn.contacts)):
        return False
    if len(so.states) != len(sn.states) or any(a is not b and a.data_ptr() != b.data_ptr() for a, b in zip(so.states, sn.states)):
        return False
    for k in range(old.n):
        a, b = old.args.terms[k], new.args.terms[k]
        if a.op != b.op or a.flags != b.flags or a.row != b.row or any(a.i[j] != b.i[j] for j in range(4)):
            return False
    return True


def refresh_terms(old, new) -> None:
    """``new``'s numbers into ``old``'s descriptor (same_structure holds): the term table, entry by entry."""
    for k in range(old.n):
        a, b = old.args.terms[k], new.args.terms[k]
        a.w = b.w
        for j in range(4):
            a.p[j] = b.p[j]


def _program_trace_pre(prog, ext_dtype):
    """Recorded step: what must run in Python right before this program's op — its Python-level terms."""
    if not prog.slots.exts:
        return None

    def pre(prog=prog, ext_dtype=ext_dtype):
        keep: list = []
        prog.slots.rebind_exts(prog.args, ext_dtype, keep)
        prog._keep_ext = keep

    return pre


RewardProgram._trace_pre = lambda self, args: _program_trace_pre(self, torch.float32)


def eval_reward_spec(env, spec: TermSpec) -> torch.Tensor:
    """Direct call of an ``mdp.rewards.*`` function: one-term program in EVAL mode → ``[N]`` tensor."""
    prog = RewardProgram(env)
    prog.add(spec, lambda: None, 1.0, 0)
    out = torch.empty(env.num_envs, device=gs.device, dtype=gs.tc_float)
    prog.args.mode = nat.GF_REWARD_MODE_EVAL
    prog.args.term_out = out.data_ptr()
    prog.launch()
    return out


class TerminationProgram:
    """Compiles termination terms into a GfTerminationArgs and launches gf_termination_step."""

    def __init__(self, env):
        self.env = env
        self.args = nat.GfTerminationArgs()
        self.slots = _Slots(env)
        self.entity = None
        self.n = 0

    def add(self, spec: Optional[TermSpec], fallback: Callable[[], torch.Tensor], time_out: bool) -> None:
        if self.n >= nat.GF_MAX_TERM_TERMS:
            raise RuntimeError(f"at most {nat.GF_MAX_TERM_TERMS} termination terms")
        if spec is not None and spec.entity is not None:
            if self.entity is None:
                self.entity = spec.entity
            elif spec.entity is not self.entity:
                spec = None
        if spec is None:
            spec = TermSpec(nat.GF_T_EXTERNAL, ext={0: fallback})
        term = self.args.terms[self.n]
        self.slots.resolve(spec, term)
        term.flags = spec.flags | (nat.GF_TERM_FLAG_TIME_OUT if time_out else 0)
        self.n += 1
        self.args.num_terms = self.n

    def launch(self) -> None:
        env, a = self.env, self.args
        keep: list = []
        a.num_envs = env.num_envs
        if self.entity is not None:
            env.entity_views(self.entity).fill(a.entity)
        a.episode_length = env.episode_length.data_ptr()
        a.max_episode_length = None if env.max_episode_length is None else env.max_episode_length.data_ptr()
        self.slots.bind(a, torch.bool, keep)
        env.backend.call("termination_step", a, owner=self)
        self._keep = keep


TerminationProgram._trace_pre = lambda self, args: _program_trace_pre(self, torch.bool)


def eval_termination_spec(env, spec: TermSpec) -> torch.Tensor:
    """Direct call of an ``mdp.terminations.*`` function → bool ``[N]`` tensor."""
    prog = TerminationProgram(env)
    prog.add(spec, lambda: None, False)
    n = env.num_envs
    scratch = torch.empty(3, n, device=gs.device, dtype=torch.bool)
    prog.args.terminated = scratch[0].data_ptr()
    prog.args.truncated = scratch[1].data_ptr()
    prog.args.term_out = scratch[2].data_ptr()
    prog.args.stats = None
    prog.launch()
    return scratch[2]
