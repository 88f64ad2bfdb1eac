"""Loader for oracle/_ref/: the reference's own sources compiled in the build container by oracle/Makefile
(never copied, outputs only).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import importlib
import os
import sys

REF_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def load(name):
    """Import oracle/_ref/<name>.so (`grouped_cumprod_ref_host`: the two forward .cu files on rocThrust's CPP backend;
    `grouped_cumprod_ref_gfx950`: the whole extension for the GPU) or return None where it was not built."""
    if not os.path.exists(os.path.join(REF_DIR, name + ".so")):
        return None
    import torch  # noqa: F401  (the extensions link against libtorch)

    if REF_DIR not in sys.path:
        sys.path.insert(0, REF_DIR)
    return importlib.import_module(name)
