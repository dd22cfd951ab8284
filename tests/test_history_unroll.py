"""gf_history_unroll through the raw C ABI: in-place history ring → the reference's newest-first tensor.

The reference returns `torch.cat(self._history, dim=-1)` with the newest frame first (observation_manager.py:219-226).  A manager
that keeps its history as a ring ([N, H, O], newest frame in slot k, older ones upwards, wrapping) produces that tensor with one
gather launch.  Checked against numpy on the oracle (CPU) and against the oracle on the GPU, for frame widths that are and are not
multiples of four (units straddling frame and row edges), every ring slot, env counts with array tails, and the second destination."""
import ctypes as C

import numpy as np
import pytest
import torch

from genesis_forge_amd import _native as nat

CASES = [  # (N, O, H)
    (1, 1, 1), (3, 1, 2), (5, 3, 4), (70, 62, 5), (130, 16, 5), (64, 48, 3), (257, 7, 6), (1000, 45, 2), (33, 310, 2), (4099, 62, 5),
]


def _want(ring, k):
    n, H, O = ring.shape
    order = [(k + j) % H for j in range(H)]
    return ring[:, order, :].reshape(n, H * O)


def _args(ring_ptr, out_ptr, out2_ptr, n, O, H, slot):
    a = nat.GfHistoryUnrollArgs()
    a.ring, a.out, a.out2, a.num_envs, a.frame_width, a.history_len, a.ring_slot = ring_ptr, out_ptr, out2_ptr, n, O, H, slot
    return a


@pytest.mark.parametrize("n,O,H", CASES)
def test_oracle_unroll_is_the_newest_first_cat(oracle_lib_path, n, O, H):
    lib = C.CDLL(oracle_lib_path)
    lib.gfo_history_unroll.argtypes = [C.POINTER(nat.GfHistoryUnrollArgs)]
    rng = np.random.default_rng(n * 1000 + O * 10 + H)
    ring = rng.standard_normal((n, H, O)).astype(np.float32)
    for k in range(H):
        out = torch.full((n, H * O), -7.0)
        out2 = torch.full((n, H * O), -9.0)
        a = _args(ring.ctypes.data, out.data_ptr(), out2.data_ptr() if k % 2 else None, n, O, H, k + 1)
        assert lib.gfo_history_unroll(C.byref(a)) == 0
        assert np.array_equal(out.numpy(), _want(ring, k))
        assert np.array_equal(out2.numpy(), _want(ring, k)) if k % 2 else bool((out2 == -9.0).all())


def test_unroll_refusals(oracle_lib_path):
    hip = C.CDLL(nat.lib_path())
    orc = C.CDLL(oracle_lib_path)
    buf = torch.zeros(64)
    for call in (lambda a: hip.gf_history_unroll(C.byref(a), None), lambda a: orc.gfo_history_unroll(C.byref(a))):
        assert call(_args(None, buf.data_ptr(), None, 4, 4, 2, 1)) == -1          # GF_E_NULL
        assert call(_args(buf.data_ptr(), None, None, 4, 4, 2, 1)) == -1
        assert call(_args(buf.data_ptr(), buf.data_ptr(), None, 4, 4, 2, 0)) == -2  # GF_E_RANGE: slot is 1-based
        assert call(_args(buf.data_ptr(), buf.data_ptr(), None, 4, 4, 2, 3)) == -2
        assert call(_args(buf.data_ptr(), buf.data_ptr(), None, 4, 0, 2, 1)) == -2
        assert call(_args(buf.data_ptr(), buf.data_ptr(), None, -1, 4, 2, 1)) == -2
        assert call(_args(buf.data_ptr(), buf.data_ptr() + 4, None, 4, 4, 2, 1)) == -5  # GF_E_UNSUPPORTED: out must be 16-byte aligned
        assert call(_args(buf.data_ptr(), buf.data_ptr(), None, 0, 4, 2, 1)) == 0     # nothing to do


@pytest.mark.gpu
@pytest.mark.parametrize("n,O,H", CASES + [(65536 + 37, 62, 5)])
def test_hip_unroll_equals_oracle(hip_backend, oracle_lib_path, n, O, H):
    lib = C.CDLL(oracle_lib_path)
    lib.gfo_history_unroll.argtypes = [C.POINTER(nat.GfHistoryUnrollArgs)]
    g = torch.Generator().manual_seed(n + O + H)
    ring = torch.randn(n, H, O, generator=g)
    d_ring = ring.cuda()
    for k in range(H):
        want = torch.empty(n, H * O)
        assert lib.gfo_history_unroll(C.byref(_args(ring.data_ptr(), want.data_ptr(), None, n, O, H, k + 1))) == 0
        out = torch.full((n, H * O), -7.0, device="cuda")
        out2 = torch.full((n, H * O), -9.0, device="cuda")
        a = _args(d_ring.data_ptr(), out.data_ptr(), out2.data_ptr() if k % 2 == 0 else None, n, O, H, k + 1)
        hip_backend.call("history_unroll", a)
        assert torch.equal(out.cpu(), want), (n, O, H, k)
        assert torch.equal(out2.cpu(), want) if k % 2 == 0 else bool((out2 == -9.0).all())
