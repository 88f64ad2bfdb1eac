import sys, os
sys.path.insert(0, os.getcwd())
import torch
import grouped_cumprod as gc
from simplegaussiansplat_tk71_amd import raster, synthetic
from oracle import dense_render as dr
dev = torch.device("cuda", 0)
sc = synthetic.make_scene_config("cfg3", seed=0, device=dev)
w, h = sc["width"], sc["height"]
bins = raster.bin_tiles(sc["start"], sc["end"], w, h)
img = raster.blend_forward(bins, sc["start"], sc["end"], sc["mean"], sc["vinv"], sc["opacity"], sc["l_d"])
cx, cy, cw, ch = 1000, 500, 48, 40
s, e = sc["start"].cpu(), sc["end"].cpu()
hit = (s[:, 0] <= cx + cw) & (e[:, 0] >= cx) & (s[:, 1] <= cy + ch) & (e[:, 1] >= cy)
idx = torch.nonzero(hit).flatten()
shift = torch.tensor([cx, cy], dtype=torch.int32); lim = torch.tensor([cw, ch], dtype=torch.int32)
s2 = (s[idx] - shift).clamp(min=0); e2 = torch.minimum(e[idx] - shift, lim); m2 = sc["mean"].cpu()[idx] - shift
i64 = dr.render(s2, e2, m2, sc["vinv"].cpu()[idx], sc["opacity"].cpu()[idx], sc["l_d"].cpu()[idx], cw, ch, torch.float64)
got = img.cpu()[cy:cy+ch+1, cx:cx+cw+1].double()
print("fused vs dense on crop:", (got - i64).abs().max().item(), "img max", i64.abs().max().item())
ts = bins.tile_start.cpu()
cnt = ts[1:] - ts[:-1]
print("tile entries: max", int(cnt.max()), "mean", float(cnt.float().mean()), "K", bins.n_tile_pairs)
print("opacity range", float(sc["opacity"].min()), float(sc["opacity"].max()))
