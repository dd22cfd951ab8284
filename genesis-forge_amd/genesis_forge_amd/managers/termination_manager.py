"""
TerminationManager — API mirror of genesis_forge/managers/termination_manager.py.

``step`` (:151-190) is one ``gf_termination_step`` launch producing the persistent
``terminated`` / ``truncated`` masks; the per-term "fired" counts behind the
``"Terminations / <name>"`` log entries (:178-182, one ``nonzero()`` sync per term in the reference)
are counted on device and read lazily.
"""
from __future__ import annotations

from typing import Any, Callable, Optional, TypedDict

import torch

from .. import _native as nat
from .. import gs
from ._program import TerminationProgram, spec_of, refresh_terms, same_structure
from .base import BaseManager, LiveAttr
from .config import TerminationConfigItem


class TerminationConfig(TypedDict):
    fn: Callable[..., torch.Tensor]
    params: dict[str, Any]
    time_out: bool


class TerminationManager(BaseManager):
    logging_enabled = LiveAttr("logging_enabled")   # checked on every step in the reference (termination_manager.py:176)

    """Calculates termination / truncation signals (ctor as termination_manager.py:96-118)."""

    def __init__(self, env, term_cfg: dict[str, TerminationConfig], logging_enabled: bool = True, logging_tag: str = "Terminations"):
        super().__init__(env, type="termination")
        self.logging_enabled = logging_enabled
        self.logging_tag = logging_tag
        if len(term_cfg) > nat.GF_MAX_TERM_TERMS:
            raise ValueError(f"TerminationManager supports at most {nat.GF_MAX_TERM_TERMS} terms")
        self.term_cfg: dict[str, TerminationConfigItem] = {}
        for name, cfg in term_cfg.items():
            self.term_cfg[name] = TerminationConfigItem(cfg, env, on_dirty=self._mark_dirty)
        self._terminated_buf = torch.zeros(env.num_envs, device=gs.device, dtype=torch.bool)
        self._truncated_buf = torch.zeros_like(self._terminated_buf)
        self._program: Optional[TerminationProgram] = None
        self._dirty = True

    def _mark_dirty(self, soft: bool = False):
        """``soft``: a param VALUE was assigned — see RewardManager._mark_dirty."""
        self._dirty = True
        env = self.env
        if soft and hasattr(env, "_soft_dirty"):
            env._soft_dirty.add(self)
        elif hasattr(env, "invalidate_trace"):
            env.invalidate_trace()

    def _refresh_in_place(self) -> bool:
        """See RewardManager._refresh_in_place."""
        old = self._program
        if old is None or not self.enabled:
            return False
        self._compile()
        return self._program is old

    @property
    def dones(self) -> torch.Tensor:
        return self._terminated_buf | self._truncated_buf

    @property
    def terminated(self) -> torch.Tensor:
        return self._terminated_buf

    @property
    def truncated(self) -> torch.Tensor:
        return self._truncated_buf

    def build(self):
        for cfg in self.term_cfg.values():
            cfg.build()

    def _compile(self):
        env = self.env
        old = self._program
        prog = TerminationProgram(env)
        for name, item in self.term_cfg.items():
            fn, params = item.fn, item.params

            def fallback(fn=fn, params=params, name=name):
                try:
                    return fn(env, **params)
                except Exception as e:  # termination_manager.py:184-186
                    print(f"Error calculating termination for '{name}'")
                    raise e

            prog.add(spec_of(fn, env, params), fallback, bool(item.time_out))
        prog.args.terminated = self._terminated_buf.data_ptr()
        prog.args.truncated = self._truncated_buf.data_ptr()
        if old is not None and same_structure(old, prog):
            refresh_terms(old, prog)   # only numbers differ: the descriptor that exists takes them (RewardManager._compile)
            prog = old
        elif old is not None and getattr(env, "_trace", None) is not None:
            env.invalidate_trace()     # another structure under a recorded step
        self._program = prog
        self._dirty = False

    def step(self) -> tuple[torch.Tensor, torch.Tensor]:
        """termination_manager.py:151-190"""
        if not self.enabled:
            return self._terminated_buf, self._truncated_buf
        env = self.env
        if self._dirty:
            self._compile()
        self._program.manager = self
        self._program.args.stats = env.stats.ptr if self.logging_enabled else None
        self._program.launch()
        return self._publish()

    def _publish(self):
        """Post-launch bookkeeping: log filler and the extras keys (termination_manager.py:178-190)."""
        env = self.env
        extras = env._extras
        if self.logging_enabled:
            log = extras[env.extras_logging_key]
            if hasattr(log, "add_filler"):
                log.add_filler(self._fill_log)
        extras["terminations"] = self._terminated_buf
        extras["time_outs"] = self._truncated_buf
        return self._terminated_buf, self._truncated_buf

    def _fill_log(self, st, out):
        n = self._global_envs()
        tag = self.logging_tag
        for k, name in enumerate(self.term_cfg.keys()):
            if st.term_fired[k] > 0:  # key only present when the term fired (quirk q8)
                out[f"{tag} / {name}"] = torch.tensor(st.term_fired[k] / n, dtype=torch.float32)

    def _global_envs(self) -> int:
        return getattr(self.env, "global_num_envs", self.env.num_envs)
