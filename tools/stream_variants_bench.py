"""Which launch shape streams 2 reads + 1 write fastest?  (investigation tool, not the product)"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
V = ctypes.c_void_p


def main():
    dev = torch.device("cuda", 0)
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libstream_variants.so"))
    lib.stream_variant.argtypes = [ctypes.c_int, V, V, V, ctypes.c_longlong, ctypes.c_int, V]
    m = 166_335_639 // 4 * 4
    x = torch.rand(m, device=dev)
    k = torch.randint(0, 1 << 20, (m,), device=dev, dtype=torch.int32)
    y = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    names = {0: "rows4 nt-st", 1: "rows2 nt-st", 2: "rows8 nt-st", 3: "rows16 nt-st", 4: "rows4 nt-ld nt-st", 5: "rows8 nt-ld nt-st",
             6: "rows4 persistent", 7: "rows8 persistent", 8: "rows4 persistent nt-ld", 9: "rows1 nt-st"}
    grids = {6: [1024, 2048, 4096], 7: [1024, 2048], 8: [2048]}
    cases = []
    for vid in names:
        for g in grids.get(vid, [0]):
            cases.append((vid, g))
    res = {c: [] for c in cases}
    for c in cases:
        assert lib.stream_variant(c[0], x.data_ptr(), k.data_ptr(), y.data_ptr(), m, c[1], st) == 0
    torch.cuda.synchronize()
    for r in range(5):
        for c in cases:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                lib.stream_variant(c[0], x.data_ptr(), k.data_ptr(), y.data_ptr(), m, c[1], st)
            b.record(); torch.cuda.synchronize()
            res[c].append(a.elapsed_time(b) / 10 * 1e3)
    for c in cases:
        t = sorted(res[c]); med = t[len(t) // 2]
        print(f"{names[c[0]]:26s} grid {c[1]:5d}  median {med:7.1f} us  {12*m/med/1e3:6.0f} GB/s")


if __name__ == "__main__":
    main()
