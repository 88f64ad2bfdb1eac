"""Build / install.  `python setup.py build_ext --inplace` compiles the HIP library
in-tree with hipcc for gfx950 (reference counterpart: setup.py:1-47, a CUDAExtension)."""
from setuptools import Command, setup
from setuptools.command.build_ext import build_ext as _build_ext


class build_ext(_build_ext):
    """No Python C extension: one hipcc call that produces the C-ABI shared library."""

    def run(self):
        from simplegaussiansplat_tk71_amd._build import build_hip_library

        print("built", build_hip_library(force=True, verbose=True))


setup(
    name="simplegaussiansplat_tk71_amd",
    version="0.1.0",
    description="MI355X-native grouped cumprod/cumsum (alpha-compositing scan) drop-in",
    packages=["simplegaussiansplat_tk71_amd"],
    py_modules=["grouped_cumprod", "cuda_kernel"],
    package_data={"simplegaussiansplat_tk71_amd": ["lib/*.so", "csrc/*.hip"]},
    cmdclass={"build_ext": build_ext},
)
