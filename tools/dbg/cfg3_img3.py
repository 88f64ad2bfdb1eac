import sys, os
sys.path.insert(0, os.getcwd())
import torch
import grouped_cumprod as gc
from simplegaussiansplat_tk71_amd import raster, synthetic
dev = torch.device("cuda", 0)
sc = synthetic.make_scene_config("cfg3", seed=0, device=dev)
w, h = sc["width"], sc["height"]
bins = raster.bin_tiles(sc["start"], sc["end"], w, h)
img = raster.blend_forward(bins, sc["start"], sc["end"], sc["mean"], sc["vinv"], sc["opacity"], sc["l_d"])
pl = raster.pixel_lists(bins, sc["start"], sc["end"])
m = pl.pair_gauss.numel()
g = pl.pair_gauss.long()
pix = torch.repeat_interleave(torch.arange((h + 1) * (w + 1), device=dev), torch.diff(pl.pixel_off).long())
py, px = pix // (w + 1), pix % (w + 1)
dx = px.float() - sc["mean"][g, 0].float()
dy = py.float() - sc["mean"][g, 1].float()
v = sc["vinv"].reshape(-1, 4)[g]
q = (dx * v[:, 0] + dy * v[:, 2]) * dx + (dx * v[:, 1] + dy * v[:, 3]) * dy
gk = torch.exp(-0.5 * q)
og = sc["opacity"][g, 0] * gk
anti = 1.0 - og
incl = torch.empty_like(anti)
gc.grouped_cumprod_forward(anti.contiguous(), pl.pair_key, incl)
T = incl / anti
wgt = torch.where(incl != 0, T * og, torch.zeros_like(og))
p = 313969
lo, hi = int(pl.pixel_off[p]), int(pl.pixel_off[p + 1])
print("pixel", p, "pairs", hi - lo)
print("g", g[lo:hi].tolist()[:10])
print("q", q[lo:hi].tolist()[:10])
print("og", og[lo:hi].tolist()[:10])
print("incl", incl[lo:hi].tolist()[:10])
print("wgt sum", float(wgt[lo:hi].sum()))
col = (wgt[lo:hi, None] * sc["l_d"][g[lo:hi]]).sum(0)
print("col by hand", col.tolist(), "fused", img.reshape(-1, 3)[p].tolist())
want = torch.zeros((h + 1) * (w + 1), 3, device=dev)
src = wgt[:, None] * sc["l_d"][g]
want.index_add_(0, pix, src)
print("index_add", want[p].tolist())
want2 = torch.zeros((h + 1) * (w + 1), 3, device=dev)
for c in range(3):
    want2[:, c].index_add_(0, pix, src[:, c].contiguous())
print("index_add per channel", want2[p].tolist())
