"""Static programs compiled at run time (genesis_forge_amd/_programs.py, gf_post_program_register): a config the library was not
built with gets post_ws_kernel<its own structure> instead of the table interpreter.

The signature of a recorded step comes from gf_post_physics_describe, which is host-only — so the whole chain up to the launch
(signature → generated source → hipcc for gfx950 → dlopen → matcher) is tested without a GPU, for every fuzz config; the GPU leg
runs the compiled kernels against the oracle."""
import concurrent.futures
import os
import re

import pytest
import torch

import test_fuzz_configs as fz
from genesis_forge_amd import _native as nat
from genesis_forge_amd import _programs
from helpers import FLOAT_TOL

FUSED_SEEDS = None   # filled by signatures(): the fuzz seeds whose recorded step has a fused post-physics launch


def _host_describe(refs) -> str:
    """gf_post_physics_describe through the HIP library — no launch, no GPU: packing a descriptor is host work."""
    hip = _host_describe.hip = getattr(_host_describe, "hip", None) or nat.HipBackend()
    return hip.post_describe(refs)


@pytest.fixture(scope="session")
def signatures(oracle_lib_path):
    """{seed: signature} of every fuzz config's fused launch, from 4 steps on the CPU oracle; and the plugins of all of them,
    compiled in parallel (≈ 5 s each, one hipcc per core) unless the cache already holds them."""
    from genesis_forge_amd import gs
    from oracle_backend import OracleBackend

    if _programs.hipcc() is None:
        pytest.skip("no hipcc")
    old_dev, old_backend = gs.device, nat._backend
    gs.set_device("cpu")
    nat.set_backend(OracleBackend(oracle_lib_path))
    os.environ["GF_JIT"] = "off"
    sigs = {}
    try:
        for seed in fz.SEEDS:
            _out, info = fz._run(seed, "cpu", steps=4)
            if info["fused"]:
                sigs[seed] = _host_describe(info["post_refs"])
    finally:
        del os.environ["GF_JIT"]
        nat.set_backend(old_backend)
        gs.device = old_dev
    todo = {}
    for seed, sig in sigs.items():
        if sig.startswith("program 0 "):
            todo.setdefault(_programs.plugin_paths(sig)[2], sig)
    missing = [sig for so, sig in todo.items() if not os.path.exists(so)]
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        times = list(pool.map(lambda sig: _programs.compile_sync(sig)[1], missing))
    sigs["_compile_seconds"] = times
    return sigs


def test_every_fuzz_config_gets_a_program_without_a_gpu(signatures):
    """signature → source → hipcc (gfx950) → dlopen → the library's matcher picks the plugin for exactly that descriptor."""
    hip = nat.HipBackend()   # loads the library; nothing is launched
    seeds = [s for s in signatures if isinstance(s, int)]
    assert len(seeds) >= 10   # (configs with a reset() override, a user manager class or a manager reset(ids) override have no fused launch)
    ids = {}
    for seed in seeds:
        sig = signatures[seed]
        if not sig.startswith("program 0 "):
            continue   # one of the library's built-in structures
        so = _programs.plugin_paths(sig)[2]
        assert os.path.exists(so), f"seed {seed}: plugin was not built"
        ids[seed] = hip.register_program(so)
        assert ids[seed] >= 100
    assert len(ids) >= 10, "the fuzz configs should not match built-in programs"
    # every config now selects ITS program (not the interpreter, not another config's): re-run 4 steps on the oracle for the
    # descriptors and ask the library which kernel it would launch
    from genesis_forge_amd import gs
    from oracle_backend import OracleBackend

    old_dev, old_backend = gs.device, nat._backend
    gs.set_device("cpu")
    nat.set_backend(OracleBackend(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "libgf_oracle.so")))
    os.environ["GF_JIT"] = "off"
    try:
        for seed in list(ids)[:6]:
            _out, info = fz._run(seed, "cpu", steps=4)
            now = hip.post_describe(info["post_refs"])
            m = re.match(r"program (\d+) \((jit_[0-9a-f]+)\)", now)
            assert m and int(m.group(1)) == ids[seed], f"seed {seed}: {now[:80]}"
            assert now.split(": ", 1)[1] == signatures[seed].split(": ", 1)[1]
    finally:
        del os.environ["GF_JIT"]
        nat.set_backend(old_backend)
        gs.device = old_dev
    t = signatures["_compile_seconds"]
    if t:
        print(f"compiled {len(t)} programs, {min(t):.1f} … {max(t):.1f} s each (in parallel)")


def test_dry_run_signature_equals_the_recorded_steps(signatures):
    """``_programs.signature_of``: the structure signature from two steps through a backend that launches NOTHING (ahead-of-time
    compilation on a machine without a GPU; ``__graft_entry__.build()`` uses it) is the signature of the really recorded step."""
    import test_fuzz_configs as fz2

    for seed in [s for s in signatures if isinstance(s, int)][:8]:
        got = _programs.signature_of(lambda n, seed=seed: fz2.make_fuzz_env(seed), num_envs=0)
        assert got is not None and got.split(": ", 1)[1] == signatures[seed].split(": ", 1)[1], f"seed {seed}"


def test_a_stale_or_foreign_plugin_is_refused(tmp_path):
    hip = nat.HipBackend()
    bogus = tmp_path / "gfp_bogus.so"
    bogus.write_bytes(b"not an ELF file")
    with pytest.raises(nat.GfError, match="GF_E_UNSUPPORTED"):
        hip.register_program(str(bogus))
    with pytest.raises(nat.GfError, match="GF_E_UNSUPPORTED"):
        hip.register_program(nat.lib_path())   # a shared object without the gfp_* exports


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [s for s in fz.SEEDS])
def test_compiled_program_equals_oracle_hip(hip_backend, oracle_lib_path, signatures, seed, monkeypatch):
    """-m gpu: the fuzz config runs on ITS compiled program (GF_JIT=sync; the fixture has filled the cache) and equals the oracle."""
    from genesis_forge_amd import gs
    from oracle_backend import OracleBackend

    if seed not in signatures:
        pytest.skip("this config's step has no fused post-physics launch (reset() override)")
    monkeypatch.setenv("GF_JIT", "sync")
    hip, info = fz._run(seed, "cuda")
    torch.cuda.synchronize()
    if signatures[seed].startswith("program 0 "):
        assert info["program"] is not None and "plugin" in info["program"], info["program"]
        now = hip_backend.post_describe(info["post_refs"])
        assert re.match(r"program [1-3]\d\d \(jit_", now), now[:60]
    tr = info["env"]._trace
    if tr is not None and tr.tail_seg.get("obs", {}).get("fused_obs"):
        # a reset() override: the observation-only launch of the tail has a structure of its own — and a program of its own
        tail = info["env"]._program_info_tail
        assert tail is not None and "plugin" in tail, tail
        assert re.match(r"program [1-3]\d\d \(jit_", hip_backend.post_describe(tr._tail_refs))
    monkeypatch.setenv("GF_JIT", "off")
    gs.set_device("cpu")
    nat.set_backend(OracleBackend(oracle_lib_path))
    try:
        ref, _ = fz._run(seed, "cpu")
    finally:
        nat.set_backend(None)
        gs.set_device("cuda:0")
    fz._compare(hip, ref, FLOAT_TOL, f"seed {seed} compiled program")


def test_child_compiler_never_inherits_a_profilers_environment(monkeypatch):
    """under ``rocprofv3 --pmc`` LD_PRELOAD and the ROCP_* tool variables would reach hipcc, whose preloaded library
    initialises the GPU before it execs clang — a forbidden exec on this pool.  The child env is scrubbed and the default policy is
    ``off`` while such variables are present."""
    from genesis_forge_amd import _programs

    class _E:
        num_envs, jit_programs = 65536, None

    monkeypatch.delenv("GF_JIT", raising=False)
    for k in list(os.environ):
        if k in _programs._TOOL_ENV_NAMES or k.startswith(_programs._TOOL_ENV_PREFIXES):
            monkeypatch.delenv(k)
    assert _programs.mode_for(_E()) == "async" and not _programs.under_profiler()
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "x")
    monkeypatch.setenv("ROCPROFILER_PC_SAMPLING", "1")
    monkeypatch.setenv("HSA_TOOLS_LIB", "y")
    assert _programs.under_profiler() and _programs.mode_for(_E()) == "off"
    env = _programs.compiler_env()
    assert not {"LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_PC_SAMPLING", "HSA_TOOLS_LIB"} & set(env)
    assert env.get("PATH") == os.environ.get("PATH")
    monkeypatch.setenv("GF_JIT", "sync")   # an explicit choice stands (the child is scrubbed either way)
    assert _programs.mode_for(_E()) == "sync"


def test_concurrent_compiles_of_one_program_share_the_cache(tmp_path, monkeypatch):
    """ranks that record the same step at the same time compile from sources of their own and move the finished plugin
    into place atomically — no process ever reads a file another one is writing."""
    from genesis_forge_amd import _programs

    if _programs.hipcc() is None:
        pytest.skip("no hipcc")
    monkeypatch.setenv("GF_PROGRAM_CACHE", str(tmp_path))
    sig = ("program 0 (interpreter): DV = 3; n_term = 1; term = {{1, 1}}; n_rew = 1; rew = {{9, 0, 0, 0}}; n_cmd = 0; cmd_width = {}; n_obs = 1;"
           " obs[0]: width 12 history 1 items {{8, 12, 0, false, false}}; n_air = 0; n_gait = 0")
    so1, p1 = _programs.start_compile(sig)
    so2, p2 = _programs.start_compile(sig)   # the plugin is not there yet: a second compile starts, as in another rank
    assert so1 == so2 and p1 is not None and p2 is not None and p1._gf_src != p2._gf_src and p1._gf_tmp != p2._gf_tmp
    assert _programs.finish_compile(p1) == so1 and _programs.finish_compile(p2) == so1
    assert os.path.exists(so1) and not os.path.exists(p1._gf_src) and not os.path.exists(p2._gf_src)
    assert [n for n in os.listdir(tmp_path) if ".tmp" in n] == []
    assert _programs.start_compile(sig) == (so1, None)   # cache hit
