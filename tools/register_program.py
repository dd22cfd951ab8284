#!/usr/bin/env python3
"""Register a task config as a static program of the fused post-physics kernel (csrc/gf_post_programs.h).

A static program pins a config's STRUCTURE (opcode sequences, command widths, observation layout) at compile time; numbers stay
run-time arguments.  Configs that match no program run the kernel's table interpreter (≈ 1.3 × the time at 65 536 envs).

    python tools/register_program.py --config humanoid28 --name humanoid28_stress            # print the struct
    python tools/register_program.py --env my_pkg.envs:make_env --name my_task --write       # … and add it to the library sources

`--env module:callable` names a callable(num_envs) returning an un-built ManagedEnvironment.  Needs the GPU (the signature comes from
gf_post_physics_describe on the recorded step's descriptors).  With --write: rebuild with `make -C genesis-forge_amd/csrc`."""
import argparse
import importlib
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genesis-forge_amd"))
CSRC = os.path.join(ROOT, "genesis-forge_amd", "csrc")


def describe(make_env, n):
    import torch
    from genesis_forge_amd import gs

    gs.set_device("cuda:0")
    env = make_env(n)
    env.build()
    env.reset()
    d = env.action_space.shape[0]
    for _ in range(6):
        env.step(torch.zeros(n, d, device=gs.device))
    tr = env._trace
    if tr is None or tr.post_refs is None:
        raise SystemExit("this config's step is not recorded with a fused post-physics launch (GF_POST_WHY=1 prints the reason)")
    return env.backend.post_describe(tr.post_refs)


from genesis_forge_amd._programs import parse, struct_text  # noqa: E402  (the generator the run-time compiler uses too)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", help="a key of genesis_forge_amd.tasks.BASELINE_CONFIGS")
    ap.add_argument("--env", help="module:callable(num_envs) -> ManagedEnvironment")
    ap.add_argument("--signature", help="a signature line printed earlier (this tool's stderr, tools/describe_config.py): no GPU needed")
    ap.add_argument("--name", required=True, help="program name (lower_snake_case), e.g. my_quadruped_flat")
    ap.add_argument("--num-envs", type=int, default=256)
    ap.add_argument("--write", action="store_true", help="insert the struct into gf_post_programs.h and register it in gf_post.hip")
    args = ap.parse_args()
    if sum(map(bool, (args.config, args.env, args.signature))) != 1:
        raise SystemExit("give exactly one of --config / --env / --signature")
    if args.signature:
        make, origin = None, "a recorded signature"
    elif args.config:
        from genesis_forge_amd.tasks import BASELINE_CONFIGS
        make = BASELINE_CONFIGS[args.config][1]
        origin = f"tasks.BASELINE_CONFIGS['{args.config}']"
    else:
        mod, fn = args.env.split(":")
        make = getattr(importlib.import_module(mod), fn)
        origin = args.env
    sig = args.signature if args.signature else describe(make, args.num_envs)
    print(sig, file=sys.stderr)
    if not sig.startswith("program 0 "):
        raise SystemExit(f"already a static program: {sig.split(':')[0]}")
    cls = "Prog" + "".join(w.capitalize() for w in args.name.split("_"))
    text = struct_text(parse(sig), args.name, cls, f"{origin} (registered with tools/register_program.py)")
    if not args.write:
        print(text)
        print(f"// and in gf_post.hip, GF_POST_PROGRAMS: X(<next id>, gf::{cls})")
        return
    hp = os.path.join(CSRC, "gf_post_programs.h")
    src = open(hp).read()
    anchor = "template <class P>\nbool program_matches"
    assert anchor in src and cls not in src
    open(hp, "w").write(src.replace(anchor, text.lstrip("\n") + "\n" + anchor, 1))
    pp = os.path.join(CSRC, "gf_post.hip")
    src = open(pp).read()
    m = re.search(r"(#define GF_POST_PROGRAMS\(X\).*?\n)((?:\s+X\(.*\\?\n)+)", src)
    ids = [int(x) for x in re.findall(r"X\((\d+),", m.group(2))]
    last = m.group(2).rstrip("\n")
    new = last + f" \\\n    X({max(ids) + 1}, gf::{cls})\n"
    open(pp, "w").write(src.replace(m.group(2), new, 1))
    print(f"registered {cls} as program {max(ids) + 1}; rebuild: make -C genesis-forge_amd/csrc", file=sys.stderr)


if __name__ == "__main__":
    main()
