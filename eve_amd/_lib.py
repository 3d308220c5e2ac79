"""ctypes binding of libeve_moe.so (C ABI declared in include/eve_moe.h).

The shared library is built in-tree by `__graft_entry__.build()` / `eve_amd/build.py`
(hipcc --offload-arch=gfx950).  There is NO fallback: if the library is missing or a call
fails, a RuntimeError is raised -- nothing in this package computes on the CPU.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_size_t, c_void_p, c_long

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libeve_moe.so")
LIB_PATH = os.environ.get("EVE_MOE_LIB", LIB_PATH)  # A/B experiments: an alternative build of the same ABI

F32, BF16 = 0, 1
MODE_NONE, MODE_VL, MODE_VISION, MODE_TEXT = 0, 1, 2, 3
ACT_QUICK_GELU, ACT_GELU, ACT_RELU, ACT_IDENTITY = 0, 1, 2, 3
ABI_VERSION = 11
FFN_SHARED_GPU = 1

_P, _I, _Z = c_void_p, c_int, c_size_t

# name -> (restype, argtypes); must list every symbol include/eve_moe.h declares
SIGNATURES = {
    "eve_moe_abi_version": (c_int, []),
    "eve_moe_last_error": (c_char_p, []),
    "eve_moe_source_hash": (c_char_p, []),
    "eve_capacity": (c_int, [_I, _I, c_double, _I]),
    "eve_gate_logits_fwd": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "eve_gate_workspace_bytes": (c_size_t, [_I, _I]),
    "eve_gate_top2_fwd": (c_int, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "eve_gate_top2_fwd_noalloc": (c_int, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "eve_gumbel_from_uniform": (c_int, [_P, _P, c_long, _P]),
    "eve_gate_top2_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "eve_gate_wgrad_workspace_bytes": (c_size_t, [_I, _I, _I]),
    "eve_gate_wgrad": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "eve_dispatch_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "eve_dispatch_bwd": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "eve_embed_grad_workspace_bytes": (c_size_t, [_I, _I]),
    "eve_embed_grad": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "eve_route_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "eve_combine_fwd": (c_int, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "eve_combine_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "eve_split_ln_fwd": (c_int, [_P] * 8 + [_I] * 5 + [c_float, _I, _P]),
    "eve_split_ln_workspace_bytes": (c_size_t, [_I, _I]),
    "eve_split_ln_bwd": (c_int, [_P] * 8 + [_I] * 6 + [_P, _Z, _P]),
    "eve_split_permute": (c_int, [_P, _P] + [_I] * 7 + [_P]),
    "eve_ffn_fwd": (c_int, [_P] * 9 + [_I] * 8 + [_P]),
    "eve_ffn_bwd": (c_int, [_P] * 13 + [_I] * 8 + [_P, _Z, _P]),
    "eve_ffn_bwd_workspace_bytes": (c_size_t, [_I] * 6),
    "eve_scale_residual_fwd": (c_int, [_P] * 5 + [_I] * 4 + [_P]),
    "eve_scale_residual_workspace_bytes": (c_size_t, [_I, _I]),
    "eve_scale_residual_bwd": (c_int, [_P] * 6 + [_I] * 4 + [_P, _Z, _P]),
    "eve_moe_saved_bytes": (c_size_t, [_P, _I]),
    "eve_moe_workspace_bytes": (c_size_t, [_P, _I]),
    "eve_moe_saved_offsets": (c_int, [_P, _I, _P, _I]),
    "eve_moe_fwd": (c_int, [_P] * 6 + [_I] + [_P] * 8 + [_Z, _I, _P, _Z, _P]),
    "eve_moe_bwd": (c_int, [_P] * 10 + [_Z] + [_P] * 9 + [_Z, _P]),
    "eve_moe_route_fwd": (c_int, [_P] * 6 + [_I] + [_P] * 3 + [_Z, _P, _Z, _P]),
    "eve_moe_finish_fwd": (c_int, [_P, _P, _P, _Z, _P, _P]),
    "eve_moe_finish_bwd": (c_int, [_P, _P, _P, _P, _Z, _P, _P, _P]),
    "eve_moe_route_bwd": (c_int, [_P] * 9 + [_Z] + [_P] * 5 + [_Z, _P]),
}
MOE_PASS_INFERENCE, MOE_PASS_FORWARD, MOE_PASS_BACKWARD = 0, 1, 2
MOE_PASS_ROUTE_FWD, MOE_PASS_ROUTE_BWD = 3, 4
MOE_SAVED_FIELDS = ("gates", "idx", "loc", "slot", "weight", "slot_token", "slot_weight", "rows", "exp_counts",
                    "xd", "z", "a", "y")


class MoEShape(ctypes.Structure):
    """`eve_moe_shape` of include/eve_moe.h."""
    _fields_ = [(n, c_int) for n in ("S", "T", "n_vis", "mode", "M", "I", "E", "C", "act", "dtype", "embed_dtype")]

_lib = None


def load():
    """Load the library once; raise loudly when it is absent or the ABI does not match."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"eve_amd: {LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU/eager fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    got = lib.eve_moe_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"eve_amd: libeve_moe.so ABI {got} != expected {ABI_VERSION}; rebuild")
    _check_source_hash(lib)
    _lib = lib
    return lib


def source_hash() -> str:
    """Digest of the sources the loaded library was built from (eve_moe_source_hash)."""
    return load().eve_moe_source_hash().decode()


def _check_source_hash(lib) -> None:
    """The library must have been built from THIS checkout's kernel sources: a stale .so would be tested (and
    benchmarked) in place of the committed code.  EVE_MOE_LIB builds (A/B experiments) are exempt."""
    if "EVE_MOE_LIB" in os.environ:
        return
    from . import build as _build
    want = _build.source_hash()
    got = lib.eve_moe_source_hash().decode()
    if got != want:
        raise RuntimeError(
            f"eve_amd: {LIB_PATH} was built from other sources (library {got}, checkout {want}); rebuild it with "
            "`python -c 'import __graft_entry__ as g; g.build()'`")


def call(name, *args):
    """Invoke an int-returning entry point; raise RuntimeError with the library's message on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.eve_moe_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{name} failed (code {rc}): {msg}")
