"""Build recipe for libgrouped_cumprod_hip.so (gfx950 only, in-tree).

Replaces the reference's setup.py (reference: setup.py:1-47, a CUDAExtension of
cuda_kernel.cpp + three .cu files).  The product here is a C-ABI shared library
(no torch types, see include/grouped_cumprod_hip.h) compiled by one explicit
hipcc invocation; Python binds it with ctypes (_lib.py).
"""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
SRCS = [os.path.join(PKG_DIR, "csrc", f) for f in ("gcp_scan.hip", "gcp_raster.hip", "gcp_project.hip", "gcp_loss.hip", "gcp_optim.hip")]
HDRS = [os.path.join(PKG_DIR, "csrc", "gcp_device.hpp")]
INCLUDE = os.path.join(ROOT, "include")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libgrouped_cumprod_hip.so")

HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-ffp-contract=off",  # products and sums stay separate: same arithmetic as the reference's C++
    # hipcc's SLP pass packs neighbouring f32 adds into v_pk_add_f32, which un-fuses the DPP
    # row reductions (v_add_f32_dpp -> v_mov_b32_dpp + v_pk_add + moves): blend backward 2.1 -> 1.7 ms
    "-fno-slp-vectorize",
    "-fPIC",
    "-shared",
]


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [*SRCS, *HDRS, os.path.join(INCLUDE, "grouped_cumprod_hip.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip_library(force=False, verbose=False, extra_flags=()):
    """Compile the HIP library in-tree; returns its path."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [find_hipcc(), *HIPCC_FLAGS, *extra_flags, "-I", INCLUDE, "-o", LIB_PATH, *SRCS]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    if verbose and res.stderr:
        print(res.stderr)
    return LIB_PATH
