import sys, os
sys.path.insert(0, os.getcwd())
import torch
import grouped_cumprod as gc
dev = torch.device("cuda", 0)
torch.manual_seed(0)
n = 4096 * 3
x = 1.0 - 1e-4 * torch.rand(n)
k = torch.zeros(n, dtype=torch.int32)
y = torch.empty(n, device=dev)
gc.grouped_cumprod_forward(x.to(dev), k.to(dev), y)
t64 = torch.cumprod(x.double(), 0)
t32c = torch.cumprod(x, 0)
t32g = torch.cumprod(x.to(dev), 0).cpu()
for i in (3, 63, 255, 1023, 4095, 8191, 12287):
    print(i, "ours", float(y[i]) - float(t64[i]), "torch cpu f32", float(t32c[i]) - float(t64[i]), "torch gpu f32", float(t32g[i]) - float(t64[i]))
# different data: random in [0.5, 1]
x2 = 0.5 + 0.5 * torch.rand(64)
y2 = torch.empty(64, device=dev)
gc.grouped_cumprod_forward(x2.to(dev), torch.zeros(64, dtype=torch.int32, device=dev), y2)
print("rel err 64 elems", float(((y2.cpu().double() - torch.cumprod(x2.double(), 0)) / torch.cumprod(x2.double(), 0)).abs().max()))
