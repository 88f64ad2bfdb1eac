"""How long a device->host read of a few words takes behind a kernel: (a) tensor.tolist() of a device tensor (what the calls did),
(b) the kernel writing straight into pinned host memory + a stream synchronise.  One-off measurement."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplegaussiansplat_tk71_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()
x = torch.randint(0, 100, (50000, 2), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream(dev)


def run(out):
    _lib.check(lib.gcp_pixels_range(x.data_ptr(), 0, x.size(0), out.data_ptr(), st.cuda_stream), "range")


d = torch.empty(3, dtype=torch.int32, device=dev)
p = torch.empty(3, dtype=torch.int32, pin_memory=True)
for name, fn in (("device tensor .tolist()", lambda: (run(d), d.tolist())[1]),
                 ("pinned + stream.synchronize()", lambda: (run(p), st.synchronize(), p.tolist())[2])):
    for _ in range(20):
        r = fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        t0 = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"{name:32s} result {r}  median {ts[100] * 1e6:.1f} us  p10 {ts[20] * 1e6:.1f} us")
