#!/usr/bin/env python3
"""Where a fuzz seed's HIP run first leaves the oracle's, fused and with the phase chains (development tool).
    python tools/fuzz_debug.py <seed> [<seed> ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
ROOT = sys.argv[1]; seed = int(sys.argv[2])
sys.path.insert(0, os.path.join(ROOT, "genesis-forge_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from genesis_forge_amd import _native as nat, gs
from oracle_backend import OracleBackend
import test_fuzz_configs as F
gs.set_device("cuda:0"); nat.set_backend(None)
hip, info = F._run(seed, "cuda"); torch.cuda.synchronize()
gs.set_device("cpu"); nat.set_backend(OracleBackend(os.path.join(ROOT, "oracle", "libgf_oracle.so")))
ref, _ = F._run(seed, "cpu")
print("seed", seed, "NO_FUSE", os.environ.get("GF_NO_FUSE"), info)
for t, ((sa, la), (sb, lb)) in enumerate(zip(hip, ref)):
    bad = [(k, float((sa[k].float().cpu() - sb[k].float()).abs().max()), int(((sa[k].float().cpu() - sb[k].float()).abs() > 1e-5).reshape(sa[k].shape[0], -1).any(1).sum())) for k in sa
           if not torch.allclose(sa[k].float().cpu(), sb[k].float(), atol=1e-5, rtol=0, equal_nan=True)]
    if bad:
        print(" first difference at step", t, bad)
        rows = ((sa[bad[0][0]].float().cpu() - sb[bad[0][0]].float()).abs() > 1e-5).reshape(sa[bad[0][0]].shape[0], -1).any(1).nonzero().flatten().tolist()
        print(" envs", rows[:10], "terminated", [int(sa["terminated"][r]) for r in rows[:10]], "truncated", [int(sa["truncated"][r]) for r in rows[:10]])
        break
else:
    print(" equal over", len(hip), "steps")
'''
for seed in sys.argv[1:]:
    for nf in ("0", "1"):
        env = dict(os.environ, GF_NO_FUSE=nf)
        subprocess.run([sys.executable, "-c", CHILD, ROOT, seed], env=env)
