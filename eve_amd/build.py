"""Build libeve_moe.so in-tree: hipcc --offload-arch=gfx950 (cross-compiles without a GPU).

    python eve_amd/build.py [--force]          eve_amd/libeve_moe.so
    python eve_amd/build.py --debug-knobs      eve_amd/libeve_moe_dbg.so: the same library with the EVE_FFN_* A/B
                                               environment knobs of ffn_mfma.hip compiled in (select it with
                                               EVE_MOE_LIB=eve_amd/libeve_moe_dbg.so; the shipped library reads no
                                               environment variable)

The library carries a digest of its sources (eve_moe_source_hash); `eve_amd/_lib.py` compares it with the
digest of the checkout, so a library built from other sources is refused instead of silently tested.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HEADER = os.path.join(HERE, "..", "include", "eve_moe.h")
OUT = os.path.join(HERE, "libeve_moe.so")
OUT_DBG = os.path.join(HERE, "libeve_moe_dbg.so")
SOURCES = ["capi.hip", "gate.hip", "permute.hip", "ffn.hip", "ffn_mfma.hip", "multiway.hip", "fused.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-Wno-unused-const-variable"]


def source_files():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) + [HEADER]


def source_hash() -> str:
    """sha256 over (file name, contents) of every kernel source and the public header, 16 hex digits."""
    h = hashlib.sha256()
    for path in source_files():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def _stale(out: str) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    if any(os.path.getmtime(d) > t for d in source_files()):
        return True
    # mtimes lie after a checkout or a copy: the digest recorded beside the library at link time must be the checkout's
    try:
        with open(out + ".digest") as f:
            return f.read().strip() != source_hash()
    except OSError:
        return True


def build(force: bool = False, verbose: bool = True, debug_knobs: bool = False, variant: str = "", defines=(),
          extra_flags=()) -> str:
    """`variant` + `defines`: an experimental build eve_amd/libeve_moe_<variant>.so with extra -D flags and the debug
    knobs (kernel A/B experiments; select it with EVE_MOE_LIB)."""
    out = OUT_DBG if debug_knobs else OUT
    if variant:
        out, debug_knobs = os.path.join(HERE, f"libeve_moe_{variant}.so"), True
    if not force and not _stale(out):
        return out
    digest = source_hash()
    objs = []
    bdir = os.path.join(HERE, "..", "build", f"obj_{variant}" if variant else ("obj_dbg" if debug_knobs else "obj"))
    os.makedirs(bdir, exist_ok=True)
    procs = []
    for src in SOURCES:
        if not os.path.exists(os.path.join(CSRC, src)):
            continue
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = ["hipcc", *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if src == "capi.hip":
            cmd.insert(-4, f'-DEVE_SOURCE_HASH="{digest}"')
        if debug_knobs:
            cmd.insert(-4, "-DEVE_FFN_DEBUG_KNOBS=1")
        for d in defines:
            cmd.insert(-4, "-D" + d)
        for f in extra_flags:  # (variant builds only: compiler-option experiments)
            cmd.insert(-4, f)
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(out + ".digest", "w") as f:
        f.write(digest + "\n")
    return out


if __name__ == "__main__":
    argv = sys.argv[1:]
    var = argv[argv.index("--variant") + 1] if "--variant" in argv else ""
    defs = [argv[i + 1] for i, a in enumerate(argv) if a == "--define"]
    flags = [argv[i + 1] for i, a in enumerate(argv) if a == "--flag"]
    if flags and not var:
        raise SystemExit("build.py: --flag is for --variant builds")
    build(force="--force" in argv, debug_knobs="--debug-knobs" in argv, variant=var, defines=defs, extra_flags=flags)
