"""ctypes binding of libvitamd.so (C ABI declared in include/vitamd.h).

The library is built in-tree by `csrc/Makefile` (see __graft_entry__.build).  There is no
fallback: if it is missing, `load()` raises — the product path never routes around the HIP kernels.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvitamd.so")
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
ABI_VERSION = 8

_c = ctypes
_P, _I, _F, _L, _U64 = _c.c_void_p, _c.c_int, _c.c_float, _c.c_long, _c.c_ulonglong

# name -> argtypes ; every function returns int (VITAMD_OK == 0)
SIGNATURES = {
    "vitamd_abi_version": [],
    "vitamd_init": [_I, _P],
    "vitamd_gemm_nt_plan": [_I, _I, _I, _I, _I, _I],
    "vitamd_gemm_nt_bf16": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "vitamd_gemm_tn_bf16": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "vitamd_gemm_tn_bf16_ws": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _L, _I, _I, _P],
    "vitamd_gemm_tn_ws_bytes": [_I, _I, _I, _I],
    "vitamd_layernorm_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _F, _P],
    "vitamd_layernorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "vitamd_layernorm_affine_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _F, _P],
    "vitamd_layernorm_affine_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "vitamd_layernorm_affine_fwd_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _F, _P],
    "vitamd_layernorm_affine_bwd_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "vitamd_attention_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _U64, _P],
    "vitamd_attention_fwd_resid": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _U64, _P],
    "vitamd_attention_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _U64, _P],
    "vitamd_layernorm_bwd_dropout": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _U64, _P],
    "vitamd_layernorm_bwd_xhat": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _U64, _P],
    "vitamd_linear_dropout_resid_bf16": [_P, _P, _P, _P, _P, _I, _I, _I, _F, _U64, _I, _P],
    "vitamd_cast_f32_bf16_dropout": [_P, _P, _L, _F, _U64, _P],
    "vitamd_cast_f32_bf16": [_P, _P, _L, _P],
    "vitamd_dropout_bf16": [_P, _P, _L, _L, _F, _U64, _P],
    "vitamd_dropout_f32": [_P, _P, _L, _L, _F, _U64, _P],
    "vitamd_cast_transpose_weight": [_P, _P, _P, _I, _I, _P],
    "vitamd_cast_transpose_batched": [_P, _I, _I, _P],
    "vitamd_im2col_bf16": [_P, _P, _I, _I, _I, _I, _I, _P],
    "vitamd_colsum_bf16": [_P, _P, _I, _I, _I, _P],
    "vitamd_embed_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vitamd_vq_nearest": [_P, _P, _P, _I, _I, _I, _P],
    "vitamd_conv3x3_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vitamd_conv3x3_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vitamd_adamw_step": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _P],
}

ERRORS = {1: "unsupported shape", 2: "bad argument", 3: "HIP launch failure", 4: "vitamd_init has not run for this device"}

_lib = None


class VitamdError(RuntimeError):
    pass


EXP_LIB_PATH = os.path.join(_HERE, "libvitamd_exp.so")


def build(verbose: bool = False, experimental: bool = False) -> str:
    """Compile libvitamd.so for gfx950 with hipcc (cross-compiles without a GPU).  experimental=True also builds libvitamd_exp.so: the
    same sources with -DVITAMD_EXPERIMENTAL (measured alternative kernels + vitamd_set_debug) for the A/B tools under tools/ - those
    call build(experimental=True) themselves through use_experimental().  The product and its tests load libvitamd.so only."""
    for args in ([], ["EXPERIMENTAL=1"])[: 2 if experimental else 1]:
        r = subprocess.run(["make", "-C", CSRC, "-j8", *args], capture_output=True, text=True)
        if verbose or r.returncode != 0:
            print(r.stdout[-4000:], r.stderr[-4000:])
        if r.returncode != 0:
            raise VitamdError("building libvitamd%s.so failed" % ("_exp" if args else ""))
    return LIB_PATH


def use_experimental():
    """A/B tools only: make load() return libvitamd_exp.so (alternative kernels behind extra `tile` codes, vitamd_set_debug).
    Must be called before the first load()."""
    global LIB_PATH
    if _lib is not None:
        raise VitamdError("use_experimental() must come before the first load()")
    if not os.path.exists(EXP_LIB_PATH):
        build(experimental=True)
    LIB_PATH = EXP_LIB_PATH


def hip_runtimes(maps_path="/proc/self/maps"):
    """The distinct libamdhip64 files mapped into this process (by real path)."""
    found = set()
    with open(maps_path) as fh:
        for line in fh:
            parts = line.split(None, 5)
            if len(parts) == 6 and "libamdhip64" in os.path.basename(parts[5].strip()):
                found.add(os.path.realpath(parts[5].strip()))
    return sorted(found)


def check_single_hip_runtime(maps_path="/proc/self/maps"):
    """Two HIP runtimes in one process - e.g. /opt/rocm's libamdhip64 pulled in by an early dlopen of libvitamd.so AND the copy a PyTorch wheel
    bundles - register kernels with one and pass streams and pointers of the other: the first launch fails with `HIP launch failure` (round 3,
    gpurun_out/r3/smoke_dbg*.log).  Detected here instead of relying on import order: raises VitamdError naming both files."""
    found = hip_runtimes(maps_path)
    if len(found) > 1 and os.environ.get("VITAMD_ALLOW_TWO_HIP_RUNTIMES") != "1":      # (the variable: for a tool that maps a second copy it never launches through)
        raise VitamdError("two HIP runtimes are mapped into this process: " + " and ".join(found) + " - libvitamd.so must resolve libamdhip64 to "
                          "the runtime that owns the caller's streams and pointers (import torch / load the framework BEFORE dlopen-ing libvitamd.so)")
    return found


def load():
    """Load the library (once) and type every entry point.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VitamdError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(the HIP kernels are the only implementation of this path; there is no fallback)")
    # torch FIRST: its wheel bundles its own HIP runtime (libamdhip64.so of the ROCm it was built against).  Loaded before torch, libvitamd.so would pull
    # /opt/rocm's copy in and torch then its own: two HIP runtimes in one process - kernels registered with one, streams and pointers from the other -
    # and the first launch fails (seen on a GPU box with `build(); smoke()` in one process).  With torch imported first the library's libamdhip64
    # dependency resolves to the runtime already in the process.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    check_single_hip_runtime()
    if LIB_PATH == EXP_LIB_PATH:
        for dbg in (lib.vitamd_set_debug, lib.vitamd_set_debug2):
            dbg.argtypes = [ctypes.c_int]
            dbg.restype = ctypes.c_int
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.argtypes = argtypes
        fn.restype = ctypes.c_long if name.endswith("_bytes") else ctypes.c_int
    if lib.vitamd_abi_version() != ABI_VERSION:
        raise VitamdError("libvitamd.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        raise VitamdError(f"{what}: {ERRORS.get(code, code)}")
