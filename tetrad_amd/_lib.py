"""ctypes binding of libtetrad_hip.so (the C ABI declared in include/tetrad_hip.h).

There is no CPU fallback: if the HIP library is missing or cannot be loaded the
import-time helper raises, and every compute entry point raises TetradHipError
on a non-zero return code (message from tq_last_error).
"""
from __future__ import annotations

import ctypes
import os
import sys
import shutil
import subprocess
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB_PATH = CSRC / "libtetrad_hip.so"
SRC_PATH = CSRC / "tetrad_hip.hip"
HEADER = Path(__file__).resolve().parents[1] / "include" / "tetrad_hip.h"

TQ_OK = 0
ERR_NAMES = {
    -1: "TQ_ERR_INVALID_ARG", -2: "TQ_ERR_NO_DEVICE", -3: "TQ_ERR_HIP",
    -4: "TQ_ERR_NO_DATA", -5: "TQ_ERR_LOCUS_ORDER", -6: "TQ_ERR_OOM",
}
FLAG_ZERO_DATA, FLAG_DEGENERATE, FLAG_BAD_INDEX, FLAG_NO_CONVERGENCE, FLAG_INVALID_DIAGNOSTIC = 1, 2, 4, 8, 16

#: every symbol include/tetrad_hip.h declares (checked by tests/test_cabi_symbols.py)
SYMBOLS = [
    "tq_create", "tq_destroy", "tq_last_error", "tq_set_data", "tq_resolve",
    "tq_set_source", "tq_bootstrap", "tq_bootstrap_async", "tq_sample_quartets_dev", "tq_get_data", "tq_data_shape",
    "tq_host_alloc", "tq_host_free", "tq_resolve_to_host", "tq_scan_dev", "tq_svd_dev",
    "tq_resolve_dev", "tq_resolve_range_dev", "tq_unrank_dev", "tq_resolve_debug",
    "tq_timing_enable", "tq_timing_read", "tq_timing_read_split", "tq_timing_read_kernels",
    "tq_set_option", "tq_device_info", "tq_debug_fetch", "tq_debug_bdsqr",
    "tq_format_tsv", "tq_format_qmc", "tq_qmc_tree", "tq_qmc_splits", "tq_unrank", "tq_numpy_choice_tail",
]


class TetradHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        self.code = code
        super().__init__(f"{ERR_NAMES.get(code, code)}: {message}")


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    sources = [SRC_PATH, HEADER, *CSRC.glob("*.hpp")]
    newest_src = max(p.stat().st_mtime for p in sources)
    if not force and LIB_PATH.exists() and LIB_PATH.stat().st_mtime >= newest_src:
        return LIB_PATH
    # -Wno-inline-asm: scan.hpp's park stores set M0 (ds_write_addtid_b32) and say so in their clobber list; the
    # compiler warns that it will not preserve a reserved register across the statement (nothing else in the unit uses M0)
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-Wno-inline-asm",
           "-o", str(LIB_PATH), str(SRC_PATH)]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    _stamp_commit()
    return LIB_PATH


def _stamp_commit() -> None:
    """Remember which commit the library was built from (bench.py quotes it; the GPU box has no .git)."""
    repo = CSRC.parents[1]
    try:
        head = subprocess.check_output(["git", "-C", str(repo), "rev-parse", "--short", "HEAD"],
                                       stderr=subprocess.DEVNULL).decode().strip()
        dirty = subprocess.check_output(["git", "-C", str(repo), "status", "--porcelain", "--untracked-files=no"],
                                        stderr=subprocess.DEVNULL).decode().strip()
        (repo / ".build_commit").write_text(head + ("+uncommitted" if dirty else "") + "\n")
    except Exception:
        pass


_lib = None
LOADED_PATH = None      # the file load() mapped (bench.py records it: an inherited TQ_LIB_PATH must show in the line)


def load() -> ctypes.CDLL:
    """Load libtetrad_hip.so and declare every prototype.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc, gfx950).  tetrad_amd has no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.7 and must be the
    # first to load it, otherwise torch.cuda (device buffers, streams, RCCL) finds no GPU.
    try:
        import torch  # noqa: F401
    except ImportError:  # torch is plumbing, not a requirement of the host-buffer API
        pass
    # developer knob for A/B runs of alternative builds of the same source (tools/ab/*.so): never a fallback --
    # a path that does not exist fails exactly like a missing library
    alt = os.environ.get("TQ_LIB_PATH")
    if alt:
        print(f"tetrad_amd: TQ_LIB_PATH is set -- loading {alt} instead of {LIB_PATH}", file=sys.stderr)
    global LOADED_PATH
    LOADED_PATH = alt if alt else str(LIB_PATH)
    lib = ctypes.CDLL(LOADED_PATH)
    c = ctypes
    vp, i64, i32 = c.c_void_p, c.c_int64, c.c_int
    lib.tq_create.argtypes = [c.POINTER(vp), i32]
    lib.tq_create.restype = i32
    lib.tq_destroy.argtypes = [vp]
    lib.tq_destroy.restype = None
    lib.tq_last_error.argtypes = [vp]
    lib.tq_last_error.restype = c.c_char_p
    lib.tq_set_data.argtypes = [vp, vp, i64, i64, vp, i64]
    lib.tq_set_data.restype = i32
    lib.tq_set_source.argtypes = [vp, vp, i64, i64, vp, i64]
    lib.tq_set_source.restype = i32
    lib.tq_bootstrap.argtypes = [vp, vp, i64, c.c_uint64, c.c_uint64, c.POINTER(i64)]
    lib.tq_bootstrap.restype = i32
    lib.tq_bootstrap_async.argtypes = [vp, vp, i64, c.c_uint64, c.c_uint64, c.POINTER(i64), vp]
    lib.tq_bootstrap_async.restype = i32
    lib.tq_sample_quartets_dev.argtypes = [vp, c.c_uint64, i64, vp, vp, vp]
    lib.tq_sample_quartets_dev.restype = i32
    lib.tq_get_data.argtypes = [vp, vp, vp]
    lib.tq_get_data.restype = i32
    lib.tq_data_shape.argtypes = [vp, c.POINTER(i64), c.POINTER(i64)]
    lib.tq_data_shape.restype = i32
    lib.tq_resolve.argtypes = [vp, vp, i64, i32, vp, vp, vp]
    lib.tq_resolve.restype = i32
    lib.tq_host_alloc.argtypes = [i64, c.POINTER(vp)]
    lib.tq_host_alloc.restype = i32
    lib.tq_host_free.argtypes = [vp]
    lib.tq_host_free.restype = i32
    lib.tq_resolve_to_host.argtypes = [vp, vp, i64, i32, vp, vp, vp]
    lib.tq_resolve_to_host.restype = i32
    lib.tq_scan_dev.argtypes = [vp, vp, i64, i32, vp]
    lib.tq_scan_dev.restype = i32
    lib.tq_svd_dev.argtypes = [vp, i64, i64, vp, vp, vp, vp]
    lib.tq_svd_dev.restype = i32
    lib.tq_timing_read_kernels.argtypes = [vp, c.POINTER(c.c_double), i32, c.POINTER(i64)]
    lib.tq_timing_read_kernels.restype = i32
    lib.tq_resolve_dev.argtypes = [vp, vp, i64, i32, vp, vp, vp, vp]
    lib.tq_resolve_dev.restype = i32
    lib.tq_resolve_range_dev.argtypes = [vp, c.c_uint64, i64, i32, vp, vp, vp, vp, vp]
    lib.tq_resolve_range_dev.restype = i32
    lib.tq_unrank_dev.argtypes = [vp, vp, i64, vp, vp]
    lib.tq_unrank_dev.restype = i32
    lib.tq_resolve_debug.argtypes = [vp, vp, i64, i32, vp, vp, vp, vp, vp, vp]
    lib.tq_resolve_debug.restype = i32
    lib.tq_timing_enable.argtypes = [vp, i32]
    lib.tq_timing_enable.restype = i32
    lib.tq_timing_read.argtypes = [vp, c.POINTER(c.c_double), c.POINTER(i64)]
    lib.tq_timing_read.restype = i32
    lib.tq_timing_read_split.argtypes = [vp, c.POINTER(c.c_double), c.POINTER(c.c_double),
                                         c.POINTER(c.c_double), c.POINTER(i64)]
    lib.tq_timing_read_split.restype = i32
    lib.tq_debug_fetch.argtypes = [vp, i32, vp, i64]
    lib.tq_debug_fetch.restype = i32
    lib.tq_debug_bdsqr.argtypes = [vp, vp, i64, vp, vp, i32, c.POINTER(c.c_double)]
    lib.tq_debug_bdsqr.restype = i32
    lib.tq_set_option.argtypes = [vp, c.c_char_p, i64]
    lib.tq_set_option.restype = i32
    lib.tq_format_tsv.argtypes = [vp, vp, vp, i64, vp, i64, c.POINTER(i64)]
    lib.tq_format_tsv.restype = i32
    lib.tq_format_qmc.argtypes = [vp, vp, vp, i64, i32, i64, c.c_double, vp, i64, c.POINTER(i64), c.POINTER(i64)]
    lib.tq_format_qmc.restype = i32
    lib.tq_qmc_splits.argtypes = [vp, vp, vp, i64, i32, i64, c.c_double, vp, vp, c.POINTER(i64)]
    lib.tq_qmc_splits.restype = i32
    lib.tq_numpy_choice_tail.argtypes = [vp, c.c_uint64, i64, vp]
    lib.tq_numpy_choice_tail.restype = i32
    lib.tq_unrank.argtypes = [vp, c.c_uint64, i64, i64, vp]
    lib.tq_unrank.restype = i32
    lib.tq_qmc_tree.argtypes = [vp, vp, i64, i64, c.c_uint64, vp, i64, c.POINTER(i64)]
    lib.tq_qmc_tree.restype = i32
    lib.tq_device_info.argtypes = [vp, c.POINTER(c.c_int32), c.POINTER(c.c_int32), c.POINTER(i64)]
    lib.tq_device_info.restype = i32
    _lib = lib
    return lib
