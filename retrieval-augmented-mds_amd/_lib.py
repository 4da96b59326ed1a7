"""ctypes binding of the C ABI declared in include/mips_hip.h (libmips_hip.so, hipcc, gfx950).

There is NO CPU fallback: if the library cannot be built or loaded, or no GPU is visible when a
device call is made, the product raises.  (The CPU oracle under oracle/ is test infrastructure and
is never imported from here.)
"""
from __future__ import annotations

import ctypes
import os
import shutil
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmips_hip.so")
HEADER = os.path.join(_ROOT, "include", "mips_hip.h")
ABI_VERSION = 1

# constants of include/mips_hip.h
DTYPE_F32, DTYPE_BF16, DTYPE_FP8_E4M3, DTYPE_FP8_E4M3_DOCS = 0, 1, 2, 3
METRIC_IP, METRIC_L2 = 0, 1
Q_DEVICE, OUT_DEVICE, OUT_PACKED, FORCE_IP = 1, 2, 4, 8
SYNTH_LATTICE, SYNTH_GAUSS, SYNTH_LATTICE_FP8 = 0, 1, 2
SEED_DOCS, SEED_QUERIES = 0xD0C5, 0x0E21  # fixed seeds of the synthetic workloads
MAX_K = 29
IDX_POISON = -2  # MIPS_IDX_POISON: what a search whose scan kernel timed out returns in every slot

_lock = threading.Lock()
_lib = None


def _sources():
    out = [HEADER]
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".hpp", ".h")):
            out.append(os.path.join(CSRC, f))
    return out


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(s) > t for s in _sources())


# -amdgpu-mfma-vgpr-form: MFMA results in architectural VGPRs even in kernels that pin values in AGPRs.  The
# query-stationary kernels fill the AGPR half of the register file with stationary B fragments (inline-asm "a"
# constraints); without the option LLVM then selects the AGPR-destination MFMA forms for the whole kernel, the
# accumulators compete with the fragments for AGPRs and scan_kernel_v5 spills fragments (reloaded behind a vmcnt(0)
# that drains the LDS-DMA ring); with it: 254 VGPRs + 256 AGPRs, no scratch.  No other kernel changes its spill count.
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]

EXP_LIB_PATH = os.path.join(_ROOT, "tools", "_build", "libmips_hip_exp.so")


def build_experimental(verbose: bool = False) -> str:
    """The A/B build for tools/ab.py: the same source with -DMIPS_EXPERIMENTAL (the `sub` instances of the
    experiment logs under profiles/, two of which return wrong results by design).  Never loaded by the product
    unless MIPS_HIP_EXPERIMENTAL=1 is set in the environment."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(os.path.dirname(EXP_LIB_PATH), exist_ok=True)
    if os.path.exists(EXP_LIB_PATH) and all(os.path.getmtime(s) <= os.path.getmtime(EXP_LIB_PATH) for s in _sources()):
        return EXP_LIB_PATH
    cmd = [hipcc, *HIPCC_FLAGS, "-DMIPS_EXPERIMENTAL", "-o", EXP_LIB_PATH, os.path.join(CSRC, "mips_hip.hip")]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc failed ({proc.returncode}):\n{proc.stderr[-4000:]}")
    return EXP_LIB_PATH


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 csrc/mips_hip.hip -> lib/libmips_hip.so (in-tree)."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libmips_hip.so (no CPU fallback exists)")
    os.makedirs(LIB_DIR, exist_ok=True)
    import fcntl

    # one builder at a time (the ranks of a multi-GPU job import the package together)
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():  # another process built it while we waited
                return LIB_PATH
            tmp = LIB_PATH + f".tmp{os.getpid()}"
            cmd = [hipcc, *HIPCC_FLAGS, "-o", tmp, os.path.join(CSRC, "mips_hip.hip")]
            if verbose:
                print(" ".join(cmd), flush=True)
            proc = subprocess.run(cmd, capture_output=True, text=True)
            if proc.returncode != 0:
                raise RuntimeError(f"hipcc failed ({proc.returncode}):\n{proc.stderr[-4000:]}")
            os.replace(tmp, LIB_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


def _bind(lib):
    c = ctypes
    vp, i64, i32, u64 = c.c_void_p, c.c_int64, c.c_int, c.c_uint64
    sig = {
        "mips_abi_version": (i32, []),
        "mips_last_error": (c.c_char_p, []),
        "mips_index_create": (i32, [c.POINTER(vp), i32, i64, i32, i32]),
        "mips_index_destroy": (i32, [vp]),
        "mips_index_reserve": (i32, [vp, i64]),
        "mips_index_add": (i32, [vp, vp, i64, i32, i32, vp]),
        "mips_index_reset": (i32, [vp]),
        "mips_index_ntotal": (i64, [vp]),
        "mips_index_dim": (i64, [vp]),
        "mips_index_metric": (i32, [vp]),
        "mips_index_phi": (i32, [vp, c.POINTER(c.c_double), vp]),
        "mips_index_set_phi": (i32, [vp, c.c_double]),
        "mips_index_read_rows": (i32, [vp, i64, i64, vp, vp]),
        "mips_index_add_synthetic": (i32, [vp, i64, i64, u64, i32, vp]),
        "mips_synth_fill": (i32, [vp, i64, i64, i64, u64, i32, i32, i32, vp]),
        "mips_search": (i32, [vp, vp, i32, i64, i32, vp, vp, i64, i32, vp]),
        "mips_search_split": (i32, [vp, vp, i32, i64, i32, vp, vp, i64, i32, vp, vp]),
        "mips_search_fused": (i32, [vp, vp, i32, i64, i32, i32, vp, vp, vp, i64, vp]),
        "mips_merge_topk": (i32, [vp, vp, i64, i32, i32, i32, vp, vp, i32, vp]),
        "mips_merge_topk_packed": (i32, [vp, i64, i32, i32, i32, vp, vp, i32, vp]),
        "mips_filter_ignore": (i32, [vp, vp, vp, i64, i32, i32, vp, vp, i32, vp]),
        "mips_cosine_rescore": (i32, [vp, vp, i32, i64, i32, i64, vp, i32, vp]),
        "mips_cosine_rescore_bias": (i32, [vp, vp, i32, i64, i32, i64, vp, i64, vp, i32, vp]),
        "mips_cosine_rescore_backward": (i32, [vp, vp, i32, i64, i32, i64, vp, vp, i64, vp, vp, i32, vp]),
        "mips_l2_normalize": (i32, [vp, i64, i64, i32, vp]),
        "mips_rows_max_sumsq": (i32, [vp, i64, i64, c.POINTER(c.c_double), i32, vp]),
        "mips_rows_max_sumsq_device": (i32, [vp, i64, i64, vp, i32, vp]),
        "mips_index_set_param": (i32, [vp, c.c_char_p, i64]),
        "mips_scan_timing": (i32, [vp, c.POINTER(c.c_float), c.POINTER(c.c_int), i32]),
        "mips_index_check_error": (i32, [vp, i32, vp]),
        "mips_index_last_kernel": (c.c_char_p, [vp]),
        "mips_index_margin_stats": (i32, [vp, c.POINTER(i64), c.POINTER(i64), c.POINTER(i64), i32, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return sig


EXPORTS = (
    "mips_abi_version", "mips_last_error", "mips_index_create", "mips_index_destroy",
    "mips_index_reserve", "mips_index_add", "mips_index_reset", "mips_index_ntotal",
    "mips_index_dim", "mips_index_metric", "mips_index_phi", "mips_index_set_phi", "mips_index_read_rows",
    "mips_index_add_synthetic", "mips_synth_fill", "mips_search", "mips_merge_topk",
    "mips_merge_topk_packed", "mips_filter_ignore", "mips_cosine_rescore", "mips_cosine_rescore_bias", "mips_l2_normalize", "mips_rows_max_sumsq", "mips_index_set_param", "mips_scan_timing",
    "mips_index_check_error", "mips_index_last_kernel", "mips_cosine_rescore_backward", "mips_index_margin_stats", "mips_search_fused", "mips_search_split",
    "mips_rows_max_sumsq_device",
)


def load():
    """Load (building first if needed) libmips_hip.so.  torch is imported first on purpose: its
    bundled libamdhip64 has the same SONAME as /opt/rocm's, so the dynamic loader binds our
    library to the HIP runtime torch already brought in -- one runtime per process, and torch
    device pointers are valid in our calls."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        import torch  # noqa: F401  (must precede dlopen, see docstring)

        path = build_experimental() if os.environ.get("MIPS_HIP_EXPERIMENTAL") == "1" else build()
        lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        _bind(lib)
        ver = lib.mips_abi_version()
        if ver != ABI_VERSION:
            raise RuntimeError(f"libmips_hip.so ABI {ver} != expected {ABI_VERSION}; rebuild")
        _lib = lib
        return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().mips_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what or 'libmips_hip'} failed (code {rc}): {msg}")


def require_gpu(device: int | None = None) -> int:
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError(
            "retrieval-augmented-mds_amd: no AMD GPU visible; this backend has no CPU path "
            "(the reference's CPU FAISS search is what it replaces)")
    return torch.cuda.current_device() if device is None else int(device)
