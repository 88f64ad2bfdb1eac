"""Developer benchmark of rows f1/f2 (tile binning + fused blend) on a Function-level synthetic scene
(N Gaussians, integer centres uniform in the image, boxes with mean area P*D/N, depth = index)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simplegaussiansplat_tk71_amd import raster, synthetic  # noqa: E402


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--no-cameras", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    sc = synthetic.make_scene_config(args.config, seed=0, device=dev)
    w, h, n = sc["width"], sc["height"], sc["start"].size(0)
    m = int(sc["boxsize"].sum().item())
    print(f"{args.config}: {w}x{h}, N={n} Gaussians, M={m} pairs ({m/((w+1)*(h+1)):.1f} per pixel)", flush=True)
    t0 = time.time()
    bins = raster.bin_tiles(sc["start"], sc["end"], w, h)
    torch.cuda.synchronize()
    print(f"tile pairs K={bins.n_tile_pairs} ({bins.n_tile_pairs/n:.2f} per Gaussian), tiles {bins.tiles_x}x{bins.tiles_y}, first bin {time.time()-t0:.3f}s")
    args_ = (sc["start"], sc["end"], sc["mean"], sc["vinv"], sc["opacity"], sc["l_d"])
    img, ck = raster.blend_forward(bins, *args_, with_checkpoints=True)
    gimg = torch.randn_like(img)
    res = {
        "bin_tiles": timeit(lambda: raster.bin_tiles(sc["start"], sc["end"], w, h)),
        "blend_forward (image only)": timeit(lambda: raster.blend_forward(bins, *args_)),
        "blend_forward": timeit(lambda: raster.blend_forward(bins, *args_, with_checkpoints=True)),
        "blend_backward": timeit(lambda: raster.blend_backward(bins, *args_, ck, gimg)),
    }
    tot = 0.0
    for k, (med, mn) in res.items():
        tot += med if "only" not in k else 0.0
        print(f"{k:28s} median {med*1e3:9.1f} us  min {mn*1e3:9.1f} us   {m/med/1e6:8.2f} Gpairs/s")
    print(f"bin+fwd+bwd {tot*1e3:.1f} us -> {m/tot/1e6:.2f} Gpairs/s end to end (whole Function)")
    # the same three steps with a caller-bounded entry count (no device->host read), eagerly and as ONE replayed HIP graph
    cap = int(1.25 * bins.n_tile_pairs) + 4096

    def whole(capacity):
        b = raster.bin_tiles(sc["start"], sc["end"], w, h, capacity=capacity)
        im, c = raster.blend_forward(b, *args_, with_checkpoints=True)
        return raster.blend_backward(b, *args_, c, gimg)

    t_sync = timeit(lambda: whole(None))
    t_cap = timeit(lambda: whole(cap))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        whole(cap)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        outs = whole(cap)
    t_graph = timeit(graph.replay)
    print(f"bin+fwd+bwd in one go: exact K (one read-back) {t_sync[0]*1e3:.1f} us | capacity {cap} (no read-back) {t_cap[0]*1e3:.1f} us | "
          f"the same as one replayed HIP graph {t_graph[0]*1e3:.1f} us")
    # a batch of cameras (the reference renders them one after the other, gs_model.py:402-449): 1 vs 2 streams
    cams = [sc] * 6
    for ns in (() if args.no_cameras else (1, 2, 3)):
        t, _ = timeit(lambda: raster.render_cameras(cams, n_streams=ns), iters=5, warmup=2)
        print(f"render 6 cameras (bin + forward) on {ns} stream(s): {t:.3f} ms")


if __name__ == "__main__":
    main()
