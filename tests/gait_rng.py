"""Serve the RNG calls of a USER-LEVEL gait command manager (the reference's
examples/gait_trainer/gait_command_manager.py, run unchanged) from dense per-env draws.

The user manager draws with ``torch.multinomial`` (gait selection, :379-399) and ``Tensor.uniform_`` on fresh
``[len(env_ids)]`` tensors (foot clearance, gait period, :362-377).  torch's RNG stream cannot be reproduced by a
kernel, so — as for every other random input of the path — the draws are INPUTS: column 0 selects the gait by
inverse CDF over the manager's weights, column 1 is the foot-clearance draw, column 2 the gait-period draw, each
indexed by env id.  tools/gen_golden.py uses this to record the reference, tests use it to run the same user file on
this package; the native GaitCommandManager consumes the same ``[N,3]`` arrays through ``env.set_draws``.
"""
import numpy as np
import torch

FIXED_CLEARANCE_GAITS = ("pronk", "bound")  # gait_command_manager.py:366-368


def _ids_np(ids, n):
    if ids is None:
        return np.arange(n)
    return np.asarray(torch.as_tensor(ids).cpu().numpy(), dtype=np.int64).reshape(-1)


def install(gait_cls, get_draws):
    """Wrap ``gait_cls`` (in place, once).  ``get_draws(manager, phase) -> np.float32 [N,3]`` with phase "step"/"reset"."""
    if getattr(gait_cls, "_gf_rng_installed", False):
        gait_cls._gf_get_draws = staticmethod(get_draws)
        return
    gait_cls._gf_rng_installed = True
    gait_cls._gf_get_draws = staticmethod(get_draws)
    state = {}

    def uniform_(self, lo=0.0, hi=1.0):
        col = state["cols"].pop(0)
        u = state["u"][state["ids"], col].astype(np.float32)
        lo32, hi32 = np.float32(lo), np.float32(hi)
        val = (u * np.float32(hi32 - lo32) + lo32).astype(np.float32)
        self.copy_(torch.from_numpy(np.ascontiguousarray(val)).reshape(self.shape).to(self.device))
        return self

    def multinomial(weights, num_samples, *a, **k):
        assert num_samples == 1
        if weights.shape[0] == 0:
            return torch.zeros(0, 1, dtype=torch.int64, device=weights.device)
        w = weights[0].detach().cpu().to(torch.float32)
        cum = torch.cumsum(w, 0).numpy()
        u = state["u"][state["ids"], 0].astype(np.float32)
        idx = (u[:, None] >= cum[None, :-1]).sum(axis=1).astype(np.int64)
        return torch.from_numpy(idx).reshape(-1, 1).to(weights.device)

    orig_resample, orig_set, orig_step = gait_cls.resample_command, gait_cls._set_gait, gait_cls.step

    def resample_command(self, env_ids):
        n = self.env.num_envs
        phase = "step" if getattr(self, "_gf_phase", None) == "step" else "reset"
        state.update(u=type(self)._gf_get_draws(self, phase), ids=_ids_np(env_ids, n), cols=[])
        saved = (torch.Tensor.uniform_, torch.multinomial)
        torch.Tensor.uniform_, torch.multinomial = uniform_, multinomial
        try:
            return orig_resample(self, env_ids)
        finally:
            torch.Tensor.uniform_, torch.multinomial = saved

    def _set_gait(self, gait_name, env_ids=None):
        state["ids"] = _ids_np(env_ids, self.env.num_envs)
        state["cols"] = [2] if gait_name in FIXED_CLEARANCE_GAITS else [1, 2]
        return orig_set(self, gait_name, env_ids)

    def step(self):
        self._gf_phase = "step"
        try:
            return orig_step(self)
        finally:
            self._gf_phase = None

    gait_cls.resample_command, gait_cls._set_gait, gait_cls.step = resample_command, _set_gait, step
