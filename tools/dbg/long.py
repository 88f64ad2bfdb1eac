import sys, os
sys.path.insert(0, os.getcwd())
import torch
import grouped_cumprod as gc
from oracle import c_oracle as co
dev = torch.device("cuda", 0)
torch.manual_seed(0)
n = 5 * gc.tile_elems() + 123
kl = torch.zeros(n, dtype=torch.int32); kl[n - 700:] = 1
xl = 1.0 - 1e-4 * torch.rand(n)
yl = torch.empty(n, device=dev)
for mode in (200, -1):
    gc.set_lookback_wait_us(mode)
    gc.grouped_cumprod_forward(xl.to(dev), kl.to(dev), yl)
    want = co.cumprod_forward_f64(xl, kl)
    err = (yl.cpu().double() - want).abs()
    bad = torch.nonzero(err > 1e-5).flatten()
    print("mode", mode, "walked", gc.last_lookback_tiles(dev), "left", gc.last_fallback_tiles(dev), "max err", float(err.max()), "n bad", bad.numel(),
          "first bad", bad[:3].tolist(), "last bad", bad[-3:].tolist())
    for i in (4095, 4096, 8191, 8192, 12288, 16384, 19902, 19903, 20479, 20480, n - 1):
        print("  ", i, float(yl[i]), float(want[i]))
