"""A plain C program (gcc, no C++/torch/Python types) consumes the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "abi_smoke.c")
EXE = os.path.join(ROOT, "tests", "c", "abi_smoke")


def _build():
    from simplegaussiansplat_tk71_amd import _build

    lib = _build.build_hip_library()
    libdir = os.path.dirname(lib)
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", EXE,
           "-L", libdir, "-lgrouped_cumprod_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return EXE


def test_c_program_compiles_and_links_against_the_abi():
    exe = _build()
    out = subprocess.run([exe, "link-only"], check=True, capture_output=True, text=True).stdout
    assert "link ok" in out


@pytest.mark.gpu
def test_c_program_runs_on_gpu(device):
    exe = _build()
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and "abi_smoke ok" in res.stdout, res.stdout + res.stderr
