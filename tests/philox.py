"""numpy Philox4x32-10 — third implementation (besides oracle/gf_oracle.c and csrc/gf_device.h) used to
generate parity-mode draws identically in tools/gen_golden.py and in the tests."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint32) for x in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0.astype(np.uint64)
        p1 = M1 * c2.astype(np.uint64)
        n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ np.uint32(k0)
        n1 = p1.astype(np.uint32)
        n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ np.uint32(k1)
        n3 = p0.astype(np.uint32)
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def uniform(seed: int, stream: int, n_env: int, n_col: int) -> np.ndarray:
    """[n_env, n_col] float32 U[0,1): element (e, c) == philox_uniform(seed, stream, e, c) of the C/HIP code — word c & 3 of the
    block with counter (e, c >> 2, stream).  One block per FOUR columns (not one per column), in chunks of 8 192 envs so the u64
    temporaries stay in cache: 65 536 x 96 draws take 0.3 s instead of 4."""
    groups = (n_col + 3) // 4
    out = np.empty((n_env, groups * 4), dtype=np.float32)
    grp = np.arange(groups, dtype=np.uint32)[None, :]
    words = np.empty((min(n_env, 8192), groups, 4), dtype=np.uint32)   # one scratch block for every chunk (no per-chunk stack)
    for e0 in range(0, n_env, 8192):
        m = min(8192, n_env - e0)
        env = np.arange(e0, e0 + m, dtype=np.uint32)[:, None]
        x = philox4x32_10(env, grp, np.uint32(stream & 0xFFFFFFFF), np.uint32((stream >> 32) & 0xFFFFFFFF),
                          seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        w = words[:m]
        for k in range(4):
            w[:, :, k] = x[k]
        out[e0:e0 + m] = (w.reshape(m, groups * 4) >> np.uint32(8)).astype(np.float32) * np.float32(5.9604644775390625e-8)
    return np.ascontiguousarray(out[:, :n_col])


def draws(seed: int, step: int, kind: int, n_env: int, n_col: int) -> np.ndarray:
    """Parity-mode draws for (step, kind): kind 0 = command.step, 1 = command.reset, 2 = episode length, 3 = observation noise."""
    return uniform(seed, (int(kind) << 40) | int(step), n_env, n_col)
