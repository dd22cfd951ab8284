#!/usr/bin/env python3
"""Where the fused post-physics launch of a CONFIG spends its time: per-wave wall-clock stamps at the phase boundaries of
`post_ws_kernel` (GF_WSTAMP in csrc/gf_post_ws.h), taken in a DIAGNOSTIC build of the library (-DGF_STAMPS, built into
tools/_stamps/; no product build contains a stamp) while the env steps on the recorded path.

    python tools/stamp_config.py build                    # here or on the GPU box: hipcc, ~1 min (again after any change under csrc/ or include/)
    python tools/stamp_config.py run gait_8192 [steps]    # on the GPU box
"""
import ctypes as C
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "genesis-forge_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_stamps")
LIB = os.path.join(OUT, "libgf_step_stamps.so")
NAMES = ["args staged", "pre-barrier done", "past barrier A", "role stores", "obs starts", "tile built", "past barrier B", "end"]


def build():
    os.makedirs(OUT, exist_ok=True)
    flags = "--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -fvisibility=hidden -DGF_STAMPS".split()
    srcs = [f for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    with open(os.path.join(OUT, "stamps_var.cpp"), "w") as f:
        f.write('extern "C" __attribute__((visibility("default"))) unsigned long long* gf_debug_stamps = nullptr;\n')

    def cc(s):
        o = os.path.join(OUT, s[:-4] + ".o")
        subprocess.run(["hipcc", *flags, "-I", CSRC, "-c", os.path.join(CSRC, s), "-o", o], check=True)
        return o

    with ThreadPoolExecutor(6) as ex:
        objs = list(ex.map(cc, srcs))
    var = os.path.join(OUT, "stamps_var.o")
    subprocess.run(["g++", "-fPIC", "-c", os.path.join(OUT, "stamps_var.cpp"), "-o", var], check=True)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, var], check=True)
    print("built", LIB)


def run(config, steps):
    os.environ["GF_JIT"] = "off"   # a plugin is compiled against the product GfPostArgs layout
    sys.path.insert(0, os.path.join(ROOT, "genesis-forge_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    from genesis_forge_amd import _native, gs, tasks
    _native.lib_path = lambda: LIB
    gs.set_device("cuda:0")
    name, _, size = config.partition("@")   # "go2_cmd@1048576": the config at another size
    n, factory = tasks.BASELINE_CONFIGS[name]
    n = int(size) if size else n
    env = factory(n)
    env.build()
    env.seed(1)
    env.reset()
    lib = _native.get_backend().lib
    stamps = torch.zeros(128, dtype=torch.int64, device="cuda")
    D = env.action_space.shape[0]
    acts = [torch.randn(n, D, device="cuda") for _ in range(4)]
    for i in range(20):
        env.step(acts[i % 4])
    assert env._trace is not None, "not recorded"
    C.c_void_p.in_dll(lib, "gf_debug_stamps").value = stamps.data_ptr()
    acc = torch.zeros(4, 16, dtype=torch.float64)
    for i in range(steps):
        env.step(acts[i % 4])
        torch.cuda.synchronize()
        h = stamps.cpu()
        t0 = min(int(h[64 + 16 * w + 1]) for w in range(4))
        for w in range(4):
            for k in range(1, 16):
                acc[w, k] += (int(h[64 + 16 * w + k]) - t0) / 100.0
    acc /= steps
    print(f"{config}: {n} envs, fused={env._trace.post_refs is not None}, mean of {steps} launches, us since the first wave had its args (middle workgroup)")
    print("          " + " ".join(f"{s:>17s}" for s in NAMES))
    for w in range(4):
        print(f"  wave {w}: " + " ".join(f"{float(acc[w, k]):17.2f}" for k in range(1, 9)))
    print("  inside the pre-barrier block — wave 0: terminations evaluated | command draws | gait manager;  wave 1: rows reduced | (behind the barrier) fold starts | fold done;  wave 2: rows requested")
    for w in (0, 1, 2):
        print(f"  wave {w}: " + " ".join(f"{float(acc[w, k]):17.2f}" for k in range(9, 12 if w < 2 else 10)))
    print("  folded contact phase (0 when the launch has none): slot ids + masks in LDS | prefix done | contributions in LDS | links done")
    for w in range(4):
        print(f"  wave {w}: " + " ".join(f"{float(acc[w, k]):17.2f}" for k in range(12, 16)))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 200)
