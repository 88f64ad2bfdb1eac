"""Build a variant of the HIP library with extra -D flags into variants/<name>.so (for A/B runs: GCP_LIBRARY=variants/<name>.so).

  python tools/build_variant.py stage16 -DGCP_STAGE_BWD=16
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplegaussiansplat_tk71_amd import _build  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(_build.ROOT, "variants", name + ".so")
os.makedirs(os.path.dirname(out), exist_ok=True)
cmd = [_build.find_hipcc(), *_build.HIPCC_FLAGS, *flags, f'-DGCP_SOURCE_HASH="{_build.source_hash()}"', "-I", _build.INCLUDE, "-o", out,
       *_build.SRCS]
subprocess.run(cmd, check=True)
print(out)
