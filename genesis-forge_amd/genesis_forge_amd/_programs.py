"""
Static programs compiled at run time for configs the library was not built with.

The fused post-physics kernel exists in two forms (csrc/gf_post_ws.h): a table interpreter that runs any configuration, and
``post_ws_kernel<Program>`` — the same code with one configuration's STRUCTURE (opcode sequences, slots, widths, observation
layout) as compile-time constants, about 1.3 × faster.  The reference's configs are live Python dicts
(genesis_forge/managers/config/config_item.py:31-44): a user's task is not one of the structures compiled into the library.
So, once a step has been recorded and its fused launch turns out to be the interpreter's, this module

1. takes the signature of the recorded descriptors (``gf_post_physics_describe``, host-only),
2. writes the program struct it denotes + six ``extern "C"`` exports into a ~40-line ``.hip`` file that includes the kernel's
   headers from ``csrc/``,
3. compiles it for gfx950 with ``hipcc -shared`` (≈ 5 s) into ``<package>/../programs/gfp_<hash>.so`` — the hash covers the
   signature AND the kernel headers, so a changed kernel never meets a stale plugin — and
4. registers it (``gf_post_program_register``): every later launch whose descriptor matches runs the compiled kernel.

Only structure is compiled in.  Weights, parameters, thresholds, ranges, scales, noise levels stay run-time arguments, so the
reference's live mutation (``cfg[name].weight = …``, ``params[k] = v``) keeps working without recompiling; a mutation that changes
the structure (a term enabled / disabled, a scale from 1 to something else) stops matching and runs the interpreter until its own
program is there.

Policy (``GF_JIT``): ``off``; ``sync`` — compile when the step is recorded, blocking; ``async`` — compile in a child process,
register when it is done (polled every 32 steps), the interpreter runs meanwhile.  Default: ``async`` from 16 384 envs on
(below that a step is host-bound and the kernel's 2–3 µs do not show), ``off`` otherwise, and ``off`` under a profiler
(``LD_PRELOAD`` / ``ROCP*`` / ``HSA_TOOLS*`` in the environment; the child compiler never inherits those).  Missing ``hipcc`` = ``off``.
Plugins are keyed by signature, kernel headers, flags and ``hipcc --version``.
"""
from __future__ import annotations

import hashlib
import itertools
import os
import re
import shutil
import subprocess
import time
from typing import Optional

from . import _native as nat

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(os.path.dirname(PKG_DIR), "include")
_HEADERS = ("gf_post_args.h", "gf_post_ws.h", "gf_post_programs.h", "gf_terms.h", "gf_device.h", "gf_obs_hist.h", "gf_prefetch.h",
            "gf_launch.h", "gf_contact_tile.h")
_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-fvisibility=hidden", "-Wno-unused-value"]


def cache_dir() -> str:
    """``GF_PROGRAM_CACHE``, else ``<package>/../programs``; when that tree is not writable (an installed, read-only package) a
    per-user directory."""
    d = os.environ.get("GF_PROGRAM_CACHE")
    if d:
        return d
    d = os.path.join(PKG_DIR, "programs")
    probe = d if os.path.isdir(d) else PKG_DIR
    if os.access(probe, os.W_OK):
        return d
    return os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "genesis_forge_amd", "programs")


# What a profiler / tool injects into a process: the child compiler must not inherit it.  Under ``rocprofv3 --pmc`` the preloaded
# library initialises the GPU inside hipcc, which then execs clang / lld — an exec from a GPU-initialised process.
_TOOL_ENV_PREFIXES = ("ROCP", "ROCPROFILER", "ROCTRACER", "HSA_TOOLS", "RPD_")
_TOOL_ENV_NAMES = ("LD_PRELOAD", "HSA_TOOLS_LIB", "HSA_TOOLS_REPORT_LOAD_FAILURE", "ROCP_TOOL_LIBRARIES", "ROCP_TOOL_LIB")


def under_profiler() -> bool:
    return any(k in os.environ for k in _TOOL_ENV_NAMES) or any(k.startswith(_TOOL_ENV_PREFIXES) for k in os.environ)


def compiler_env() -> dict:
    return {k: v for k, v in os.environ.items() if k not in _TOOL_ENV_NAMES and not k.startswith(_TOOL_ENV_PREFIXES)}


def hipcc() -> Optional[str]:
    return shutil.which(os.environ.get("HIPCC", "hipcc")) or (os.path.exists("/opt/rocm/bin/hipcc") and "/opt/rocm/bin/hipcc") or None


# -- signature → program struct (the notation of csrc/gf_post_programs.h; gf_post_physics_describe prints it) -----------------
def parse(sig: str) -> dict:
    body = sig.split(": ", 1)[1]
    g = lambda pat: re.search(pat, body)
    out = {"DV": int(g(r"DV = (\d+)").group(1)), "n_term": int(g(r"n_term = (\d+)").group(1)), "n_rew": int(g(r"n_rew = (-?\d+)").group(1)),
           "n_cmd": int(g(r"n_cmd = (\d+)").group(1)), "n_obs": int(g(r"n_obs = (\d+)").group(1)), "n_air": int(g(r"n_air = (\d+)").group(1)),
           "n_gait": int(g(r"n_gait = (\d+)").group(1))}
    out["n_rew"] = max(0, out["n_rew"])   # (-1: no reward manager in the launch — as a program: no reward terms; the matcher agrees)
    out["term_done"] = g(r"term_done = 1") is not None
    out["term"] = g(r"term = (\{.*?\}); n_rew").group(1)
    out["rew"] = g(r"rew = (\{.*?\}); n_cmd").group(1)
    out["cmd_width"] = [int(x) for x in re.findall(r"\d+", g(r"cmd_width = \{(.*?)\}").group(1))]
    out["obs"] = [(int(w), int(h), items) for w, h, items in re.findall(r"obs\[\d+\]: width (\d+) history (\d+) items (\{.*?\});", body)]
    return out


def struct_text(p: dict, name: str, cls: str, comment: str) -> str:
    pad = lambda xs, n: ", ".join(str(x) for x in (list(xs) + [0] * n)[:n])
    items = ",\n".join("        " + (it if it != "{}" else "{}") for it in [o[2] for o in p["obs"]] + ["{}"] * (2 - len(p["obs"])))
    arr = lambda n: max(1, n)
    return f'''
// {comment}
struct {cls} {{
    static constexpr bool kStatic = true;
    static constexpr const char* name = "{name}";
    static constexpr int DV = {p["DV"]};
    static constexpr int n_term = {p["n_term"]};
    static constexpr TermSig term[{arr(p["n_term"])}] = {p["term"] if p["n_term"] else "{}"};
    static constexpr int n_rew = {p["n_rew"]};
    static constexpr RewSig rew[{arr(p["n_rew"])}] = {p["rew"] if p["n_rew"] > 0 else "{}"};
    static constexpr int n_cmd = {p["n_cmd"]};
    static constexpr int cmd_width[GF_POST_MAX_CMD] = {{{pad(p["cmd_width"], 2)}}};
    static constexpr int n_obs = {p["n_obs"]};
    static constexpr int obs_width[GF_POST_MAX_OBS] = {{{pad([o[0] for o in p["obs"]], 2)}}};
    static constexpr int obs_history[GF_POST_MAX_OBS] = {{{pad([o[1] for o in p["obs"]], 2)}}};
    static constexpr int obs_items[GF_POST_MAX_OBS] = {{{pad([o[2].count("{") - 1 for o in p["obs"]], 2)}}};
    static constexpr ItemSig item[GF_POST_MAX_OBS][kPostMaxItems] = {{
{items}}};
    static constexpr int n_air = {p["n_air"]};
    static constexpr int n_gait = {p["n_gait"]};{"""
    static constexpr bool term_done = true;   // the termination masks are inputs (Python-level terms ran behind a termination launch of its own)""" if p.get("term_done") else ""}
}};
'''


def _headers_digest() -> str:
    h = hashlib.sha1()
    for f in [os.path.join(CSRC, n) for n in _HEADERS] + [os.path.join(INCLUDE, "gf_step.h")]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def plugin_source(sig: str, key: str) -> str:
    body = sig.split(": ", 1)[1]
    struct = struct_text(parse(sig), "jit_" + key, "ProgJit", f"signature: {body}")
    return f'''// generated by genesis_forge_amd/_programs.py — a static program of the fused post-physics kernel, compiled at run time
#include "gf_post_args.h"
#include "gf_post_ws.h"
#include "gf_post_programs.h"

namespace gf {{
{struct}
}}  // namespace gf

#define GFP_EXPORT extern "C" __attribute__((visibility("default")))
GFP_EXPORT int gfp_abi_version(void) {{ return GF_ABI_VERSION; }}
GFP_EXPORT int gfp_args_size(void) {{ return (int)sizeof(gf::GfPostArgs); }}
GFP_EXPORT const char* gfp_name(void) {{ return gf::ProgJit::name; }}
GFP_EXPORT int gfp_matches(const gf::GfPostArgs* a) {{ return gf::program_matches<gf::ProgJit>(*a) ? 1 : 0; }}
GFP_EXPORT const void* gfp_kernel(void) {{ return (const void*)&gf::post_ws_kernel<gf::ProgJit>; }}
GFP_EXPORT size_t gfp_lds_bytes(int omax, int n_gait) {{ return gf::lds_ws_floats<gf::ProgJit>(omax, n_gait) * sizeof(float); }}
GFP_EXPORT int gfp_folds(void) {{ return gf::ws_prog_folds<gf::ProgJit>() ? 1 : 0; }}
'''


_COMPILER_ID: Optional[str] = None


def compiler_id() -> str:
    """``hipcc --version`` (one call per process): a plugin built by another ROCm release is not the one this library expects."""
    global _COMPILER_ID
    if _COMPILER_ID is None:
        cc = hipcc()
        out = ""
        if cc is not None:
            try:
                out = subprocess.run([cc, "--version"], capture_output=True, text=True, timeout=60, env=compiler_env()).stdout
            except Exception:
                out = ""
        _COMPILER_ID = hashlib.sha1(out.encode()).hexdigest()[:12] if out else "no-compiler"
    return _COMPILER_ID


def _build_tag() -> str:
    """What a plugin was built against: the kernel headers, the compiler flags and the compiler's version (8 hex digits, first part
    of its file name)."""
    return hashlib.sha1((_headers_digest() + "|" + " ".join(_FLAGS) + "|" + compiler_id()).encode()).hexdigest()[:8]


def plugin_key(sig: str) -> str:
    body = sig.split(": ", 1)[1]
    return _build_tag() + hashlib.sha1(body.encode()).hexdigest()[:12]


_STALE_SECONDS = 24 * 3600.0
_compile_counter = itertools.count()


def _sweep_stale(directory: str) -> None:
    """Plugins built against other kernel headers can never be loaded again by THIS checkout (their name no longer comes up): remove
    the ones nobody has touched for a day — another checkout or another rank sharing the cache may be compiling or about to load a
    younger one."""
    tag = _build_tag()
    now = time.time()
    try:
        for name in os.listdir(directory):
            if name.startswith("gfp_") and not name.startswith("gfp_" + tag) and (name.endswith((".so", ".hip", ".lock")) or ".tmp" in name):
                path = os.path.join(directory, name)
                try:
                    if now - os.path.getmtime(path) > _STALE_SECONDS:
                        os.unlink(path)
                except OSError:
                    pass
    except OSError:
        pass


def plugin_paths(sig: str) -> tuple:
    key = plugin_key(sig)
    d = cache_dir()
    return key, os.path.join(d, f"gfp_{key}.hip"), os.path.join(d, f"gfp_{key}.so")


def compile_command(src: str, out: str) -> list:
    return [hipcc()] + _FLAGS + ["-I", CSRC, "-I", INCLUDE, src, "-o", out]


def start_compile(sig: str):
    """Write the source and start ``hipcc`` as a child process.  Returns ``(so_path, Popen | None)`` — None when the plugin is
    already in the cache."""
    key, src, so = plugin_paths(sig)
    if os.path.exists(so):
        return so, None
    if hipcc() is None:
        raise RuntimeError("hipcc not found: static programs cannot be compiled at run time")
    os.makedirs(os.path.dirname(src), exist_ok=True)
    _sweep_stale(os.path.dirname(src))
    # Several ranks record the same step at the same time and share the cache: every process compiles from a source file of its OWN
    # (pid in the name, renamed into place: nobody ever reads a half-written file) into a plugin of its own, and the finished plugin
    # is moved into place atomically — whoever finishes last wins with identical bytes.
    text = plugin_source(sig, key)
    uid = f"{os.getpid()}_{next(_compile_counter)}"
    my_src = f"{src}.{uid}.tmp.hip"
    with open(my_src, "w") as fh:
        fh.write(text)
    try:
        if not os.path.exists(src):
            shutil.copyfile(my_src, src + f".{uid}.tmp.cp")
            os.replace(src + f".{uid}.tmp.cp", src)   # (kept for inspection: the source a cached plugin was built from)
    except OSError:
        pass
    tmp = f"{so}.{uid}.tmp"
    proc = subprocess.Popen(compile_command(my_src, tmp), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=compiler_env())
    proc._gf_tmp, proc._gf_so, proc._gf_t0, proc._gf_src = tmp, so, time.perf_counter(), my_src
    return so, proc


def finish_compile(proc) -> str:
    """Wait for a compile started by ``start_compile``; the finished plugin is moved into place atomically."""
    _out, err = proc.communicate()
    try:
        os.unlink(proc._gf_src)
    except (OSError, AttributeError):
        pass
    if proc.returncode != 0:
        try:
            os.unlink(proc._gf_tmp)
        except OSError:
            pass
        raise RuntimeError(f"hipcc failed on a generated program ({proc.returncode}):\n{err[-2000:]}")
    os.replace(proc._gf_tmp, proc._gf_so)
    proc._gf_seconds = time.perf_counter() - proc._gf_t0
    return proc._gf_so


def compile_sync(sig: str) -> tuple:
    """(plugin path, seconds spent compiling — 0.0 for a cache hit)."""
    so, proc = start_compile(sig)
    if proc is None:
        return so, 0.0
    finish_compile(proc)
    return so, proc._gf_seconds


def register(backend, so: str) -> int:
    return backend.register_program(so)


# -- ahead of time: a config's program without a GPU ---------------------------------------------------------------------------
class _DryRunBackend(nat.HipBackend):
    """Host logic only: phase calls are recorded and validated by nobody, nothing is launched.  Two steps through it are enough
    for a step to be recorded, and a recorded step's descriptors are all ``gf_post_physics_describe`` needs — packing a descriptor
    is host work, its signature is structure (opcodes, slots, widths), not values."""

    name = "dry-run"
    device_type = "cpu"

    def call(self, fn: str, args, owner=None) -> None:
        if self.tracer is not None:
            self.tracer.record(fn, args, owner)
        self._note_call(args)

    def stats_clear(self, stats_ptr: int) -> None:
        pass

    def stats_pack(self, src_ptr: int, dst_ptr: int) -> None:
        pass

    def stats_last_reset(self, rows_ptr: int, num_rows: int, dst_ptr: int) -> None:
        pass

    def run_ops(self, ops, n: int) -> None:
        pass

    def replay_step(self, replay, actions_ptr: int, params, num_params: int) -> None:
        pass

    def event_create(self):
        return None

    def event_synchronize(self, ev) -> None:
        pass


def signature_of(make_env, num_envs: int = 64) -> Optional[str]:
    """The structure signature of ``make_env(num_envs)``'s recorded step — no GPU, no kernels: the env is built on CPU tensors
    and stepped twice through a backend that launches nothing.  None when the config's step is not recorded with a fused
    post-physics launch (a ``reset()`` override, a user manager class in the middle of it)."""
    import torch

    from . import gs

    old_dev, old_backend, old_jit = gs.device, nat._backend, os.environ.get("GF_JIT")
    os.environ["GF_JIT"] = "off"
    try:
        gs.set_device("cpu")
        dry = _DryRunBackend()
        nat.set_backend(dry)
        env = make_env(num_envs)
        env.build()
        env.reset()
        width = env.action_space.shape[0]
        for _ in range(3):
            env.step(torch.zeros(env.num_envs, width))
        tr = env._trace
        if tr is None or tr.post_refs is None:
            return None
        return dry.post_describe(tr.post_refs)
    finally:
        nat.set_backend(old_backend)
        gs.device = old_dev
        if old_jit is None:
            os.environ.pop("GF_JIT", None)
        else:
            os.environ["GF_JIT"] = old_jit


def precompile(make_env, num_envs: int = 64) -> Optional[dict]:
    """Compile (or find in the cache) the static program of a config ahead of time — e.g. on a build machine without a GPU:
    ``python -m genesis_forge_amd._programs my_pkg.envs:make_env``.  Returns what happened, or None when the config runs a built-in
    program / has no fused launch."""
    sig = signature_of(make_env, num_envs)
    if sig is None or not sig.startswith("program 0 "):
        return None
    so, secs = compile_sync(sig)
    return {"signature": sig, "plugin": so, "compile_s": secs}


# -- the hook ManagedEnvironment.step calls when a step has just been recorded ---------------------------------------------------
def mode_for(env) -> str:
    m = os.environ.get("GF_JIT")
    if m is None:
        # (a run under a profiler compiles nothing by itself: set GF_JIT explicitly — the child compiler then runs with the tool's
        # variables scrubbed — or precompile the config's program beforehand)
        m = getattr(env, "jit_programs", None) or ("async" if env.num_envs >= 16384 and not under_profiler() else "off")
    return m if m in ("off", "sync", "async") else "off"


class Pending:
    """An ``async`` compile in flight for one env."""

    def __init__(self, env, sig: str, proc, so: str, slot: str = "_program_info"):
        self.env, self.sig, self.proc, self.so, self.slot = env, sig, proc, so, slot
        self.countdown = 32

    def __del__(self):
        # the env went away before the compile finished: stop the child, leave no half-written plugin behind
        proc = getattr(self, "proc", None)
        if proc is not None and proc.poll() is None:
            try:
                proc.kill()
                proc.wait(timeout=5)
            except Exception:
                pass
        my_src = getattr(proc, "_gf_src", None)
        if my_src and os.path.exists(my_src) and (proc is None or proc.poll() is not None):
            try:
                os.unlink(my_src)
            except OSError:
                pass
        tmp = getattr(proc, "_gf_tmp", None)
        if tmp and os.path.exists(tmp) and (proc is None or proc.returncode != 0 or not os.path.exists(getattr(proc, "_gf_so", ""))):
            try:
                os.unlink(tmp)
            except OSError:
                pass

    def poll(self) -> bool:
        """True when done (registered, or failed: the interpreter stays)."""
        self.countdown -= 1
        if self.countdown > 0:
            return False
        self.countdown = 32
        if self.proc.poll() is None:
            return False
        try:
            register(self.env.backend, finish_compile(self.proc))
            setattr(self.env, self.slot, {"signature": self.sig, "plugin": self.so, "compile_s": self.proc._gf_seconds, "mode": "async"})
        except Exception as e:   # a failed compile must never take the run down: the interpreter is correct, just slower
            setattr(self.env, self.slot, {"signature": self.sig, "error": str(e), "mode": "async"})
        return True


class PendingSet:
    """The async compiles in flight for one env (its fused launch; the observation-only launch of a ``reset()``-override tail)."""

    def __init__(self):
        self.items: list = []

    def add(self, p: "Pending") -> None:
        self.items.append(p)

    def poll(self) -> bool:
        self.items = [p for p in self.items if not p.poll()]
        return not self.items


def on_recorded(env, refs=None) -> None:
    """A step has just been recorded (``refs`` None: its fused launch) — or the tail of a ``reset()``-override env adopted, whose
    observation managers run as one GF_POST_OBSERVE_ONLY launch (``refs``).  If that launch is the interpreter's, get it its own
    program."""
    tr = env._trace
    backend = env.backend
    slot = "_program_info" if refs is None else "_program_info_tail"
    if refs is None:
        refs = tr.post_refs if tr is not None else None
    if refs is None or not hasattr(backend, "post_describe") or not hasattr(backend, "register_program"):
        return
    mode = mode_for(env)
    if mode == "off":
        return
    sig = backend.post_describe(refs)
    if not sig.startswith("program 0 "):
        return   # a built-in program, or one registered earlier
    if hipcc() is None and not os.path.exists(plugin_paths(sig)[2]):
        return
    try:
        so, proc = start_compile(sig)
        if proc is None or mode == "sync":
            secs = 0.0
            if proc is not None:
                finish_compile(proc)
                secs = proc._gf_seconds
            register(backend, so)
            setattr(env, slot, {"signature": sig, "plugin": so, "compile_s": secs, "mode": mode})
        else:
            if env._program_pending is None:
                env._program_pending = PendingSet()
            env._program_pending.add(Pending(env, sig, proc, so, slot))
    except Exception as e:
        setattr(env, slot, {"signature": sig, "error": str(e), "mode": mode})


if __name__ == "__main__":   # python -m genesis_forge_amd._programs module:callable [num_envs]
    import importlib
    import sys

    mod, fn = sys.argv[1].split(":")
    info = precompile(getattr(importlib.import_module(mod), fn), int(sys.argv[2]) if len(sys.argv) > 2 else 64)
    print(info if info is not None else "nothing to compile: a built-in program matches, or the step has no fused post-physics launch")
