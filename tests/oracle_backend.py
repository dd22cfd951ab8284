"""Test-only backend: routes the package's phase calls to the CPU oracle (host pointers).
Lives under tests/ on purpose — the product package never imports anything from oracle/."""
import ctypes as C

from genesis_forge_amd import _native as nat


class OracleBackend(nat.Backend):
    name = "oracle"
    device_type = "cpu"

    def __init__(self, path: str):
        self.lib = C.CDLL(path)
        nat.check_abi(self.lib, "gfo_")
        self._fn = {}
        for name, st in nat.PHASE_FUNCS.items():
            f = getattr(self.lib, "gfo_" + name)
            f.restype = C.c_int
            f.argtypes = [C.POINTER(st)]
            self._fn[name] = f
        self.lib.gfo_stats_clear.restype = C.c_int
        self.lib.gfo_stats_clear.argtypes = [C.c_void_p]
        self.lib.gfo_run_ops.restype = C.c_int
        self.lib.gfo_run_ops.argtypes = [C.POINTER(nat.GfOp), C.c_int, C.POINTER(C.c_int)]
        self.lib.gfo_replay_step.restype = C.c_int
        self.lib.gfo_replay_step.argtypes = [C.POINTER(nat.GfReplay), C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        self.calls = []
        self.replays = 0

    def call(self, fn, args, owner=None):
        self.calls.append(fn)
        if self.tracer is not None:
            self.tracer.record(fn, args, owner)
        self._note_call(args)
        rc = self._fn[fn](C.byref(args))
        if rc != 0:
            raise nat.GfError(f"gfo_{fn} failed: {nat.GF_ERRORS.get(rc, rc)}")

    def stats_clear(self, stats_ptr):
        rc = self.lib.gfo_stats_clear(stats_ptr)
        if rc != 0:
            raise nat.GfError(f"gfo_stats_clear failed: {rc}")

    def stats_pack(self, src_ptr, dst_ptr):
        a = nat.GfStatsPackArgs()
        a.src, a.dst = src_ptr, dst_ptr
        rc = self.lib.gfo_stats_pack(C.byref(a))
        if rc != 0:
            raise nat.GfError(f"gfo_stats_pack failed: {rc}")

    def stats_last_reset(self, rows_ptr, num_rows, dst_ptr):
        self.lib.gfo_stats_last_reset.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        rc = self.lib.gfo_stats_last_reset(rows_ptr, num_rows, dst_ptr)
        if rc != 0:
            raise nat.GfError(f"gfo_stats_last_reset failed: {rc}")

    def post_check(self, refs):
        return self.lib.gfo_post_physics_check(C.byref(refs)) == 0

    def run_ops(self, ops, n):
        self.replays += 1
        failed = C.c_int(-1)
        rc = self.lib.gfo_run_ops(ops, n, C.byref(failed))
        if rc != 0:
            raise nat.GfError(f"gfo_run_ops failed at op {failed.value}: {nat.GF_ERRORS.get(rc, rc)}")

    def replay_step(self, replay, actions_ptr, params, num_params):
        if replay.num_ops > 0:
            self.replays += 1
        failed = C.c_int(-1)
        rc = self.lib.gfo_replay_step(replay, actions_ptr, params, num_params, C.byref(failed))
        if rc != 0:
            raise nat.GfError(f"gfo_replay_step failed at op {failed.value}: {nat.GF_ERRORS.get(rc, rc)}")
