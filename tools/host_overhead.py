"""Where does the host time of a recorded step go?  (development tool; run on the GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genesis-forge_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from genesis_forge_amd import _native as nat, gs
from genesis_forge_amd.tasks import Go2CommandDirectionEnv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
gs.set_device("cuda:0")
env = Go2CommandDirectionEnv(num_envs=N, scene_kwargs=dict(ang_noise=0.05))
env.build(); env.reset()
act = torch.randn(N, 12, device="cuda")
for _ in range(50): env.step(act)
torch.cuda.synchronize()
tr = env._trace
assert tr is not None
def timeit(fn, n=2000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("N", N, "ops", tr.n_ops, "fused", tr.post_refs is not None)
print("env.step (replay)        %.1f us" % timeit(lambda: env.step(act)))
b = env.backend
print("run_ops only (all ops)   %.1f us" % timeit(lambda: b.run_ops(tr.ops, tr.n_ops)))
import ctypes as C
sub = (nat.GfOp * tr.n_ops)()
k = 0
for i in range(tr.n_ops):
    if tr.ops[i].phase not in (nat.GF_OP_STATS_CLEAR, nat.GF_OP_STATS_COPY, nat.GF_OP_POST_PHYSICS):
        sub[k].phase, sub[k].args = tr.ops[i].phase, tr.ops[i].args; k += 1
print("run_ops kernels only (%d)  %.1f us" % (k, timeit(lambda: b.run_ops(sub, k))))
one = (nat.GfOp * 1)()
for i in range(tr.n_ops):
    one[0].phase, one[0].args = tr.ops[i].phase, tr.ops[i].args
    print("   op phase %3d alone      %.1f us" % (tr.ops[i].phase, timeit(lambda: b.run_ops(one, 1), 1000)))
def py_only():
    env._begin_step_light()
    for p in tr.patches: p(act)
    cur, nxt, prev, prev_vec, snap = env.stats.ring_next()
    for _, f in tr.afters: f()
    env._finish_step_light(snap)
print("python bookkeeping only  %.1f us" % timeit(py_only))
# host-side enqueue cost alone: time a short burst before the queue can fill, without waiting for the GPU
def burst(fn, n=40, reps=30):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        best = min(best, (t1 - t0) / n * 1e6)
    return best
print("enqueue only: run_ops    %.1f us (host time per call, GPU not waited for)" % burst(lambda: b.run_ops(tr.ops, tr.n_ops)))
print("enqueue only: env.step   %.1f us" % burst(lambda: env.step(act)))
for i in range(tr.n_ops):
    one[0].phase, one[0].args = tr.ops[i].phase, tr.ops[i].args
    print("   enqueue only: op phase %3d  %.1f us" % (tr.ops[i].phase, burst(lambda: b.run_ops(one, 1))))
