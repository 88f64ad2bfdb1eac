#!/usr/bin/env python3
"""Generate tests/golden/scan_golden.npz from the REFERENCE's own forward kernels.

Runs only in the build container (needs /root/reference and oracle/_ref/):
    make -C oracle ref_host && python tests/golden/make_scan_golden.py

The outputs are produced by the reference's cuda_kernel/grouped_cumprod_forward.cu and
grouped_cumsum_forward.cu, compiled unmodified for the host (rocThrust CPP backend) into
oracle/_ref/grouped_cumprod_ref_host.so — thrust::inclusive_scan_by_key itself, sequential
fp32.  Only data is stored: run lengths / run key values, the inputs on a uint16 lattice
(x = (q+1)/65536, built from 1 - alpha*g with alpha drawn from sigmoid(opacity.pt) of the
reference's trained scene) and the reference's outputs.  No reference source is copied.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
import grouped_cumprod_ref_host as ref  # noqa: E402

SIZES = [1, 5, 63, 64, 65, 4097, 8209]
DISTS = ["all1", "poisson8", "geo80", "one_run"]
BIG = 100003
STRIDE = 101


def run_lengths(n, dist, g):
    if dist == "all1":
        lens = torch.ones(n, dtype=torch.long)
    elif dist == "poisson8":
        lens = torch.poisson(torch.full((n // 4 + 8,), 8.0), generator=g).long()
    elif dist == "geo80":
        u = torch.rand(n // 8 + 8, generator=g).clamp_(min=1e-12)
        lens = (torch.floor(torch.log(u) / torch.log1p(torch.tensor(-1.0 / 80.0))) + 1).long()
    else:
        lens = torch.tensor([n])
    lens = torch.cat([lens[lens > 0], torch.tensor([n])])
    csum = torch.cumsum(lens, 0)
    k = int(torch.searchsorted(csum, torch.tensor(n)).item()) + 1
    lens = lens[:k].clone()
    lens[-1] -= int(csum[k - 1].item()) - n
    return lens


def make_case(n, dist, seed, alpha_pool):
    g = torch.Generator().manual_seed(seed)
    lens = run_lengths(n, dist, g)
    # pixel-key-like values y*10000+x, increasing, with gaps (gs_model.py:538-541)
    step = torch.randint(1, 4, (lens.numel(),), generator=g)
    pix = torch.cumsum(step, 0)
    vals = ((pix // 640) * 10000 + pix % 640).to(torch.int32)
    key = torch.repeat_interleave(vals, lens)
    a = alpha_pool[torch.randint(0, alpha_pool.numel(), (n,), generator=g)]
    x = 1.0 - a * torch.rand(n, generator=g)
    q = (x * 65536.0).round().clamp_(1, 65536).to(torch.int64) - 1
    xq = ((q + 1).to(torch.float32) / 65536.0).contiguous()
    cp = torch.zeros_like(xq)
    ref.grouped_cumprod_forward(xq, key, cp)
    cs = torch.zeros_like(xq)
    ref.grouped_cumsum_forward(xq, key, cs)
    return lens, vals, q.to(torch.uint16 if hasattr(torch, "uint16") else torch.int32), cp, cs


def main():
    opacity = torch.load("/root/reference/opacity.pt", weights_only=True, map_location="cpu")
    alpha_pool = torch.sigmoid(opacity.detach().flatten())[:: 37].contiguous()
    out = {}
    # known-answer vectors stated in the reference itself (cuda_test.py:19-22,27,34)
    x = torch.tensor([0.4, 0.2, 0.1, 0.8, 0.2])
    k = torch.tensor([0, 0, 1, 1, 2], dtype=torch.int32)
    y = torch.zeros_like(x)
    ref.grouped_cumprod_forward(x, k, y)
    out["kat/cumprod"] = y.numpy().copy()
    ref.grouped_cumsum_forward(x, k, y)
    out["kat/cumsum"] = y.numpy().copy()
    seed = 1000
    for n in SIZES:
        for d in DISTS:
            seed += 1
            lens, vals, q, cp, cs = make_case(n, d, seed, alpha_pool)
            name = f"n{n}_{d}"
            out[name + "/lens"] = lens.numpy().astype(np.int32)
            out[name + "/vals"] = vals.numpy()
            out[name + "/xq"] = q.numpy().astype(np.uint16)
            out[name + "/cumprod"] = cp.numpy()
            out[name + "/cumsum"] = cs.numpy()
    for d in ("geo80",):
        seed += 1
        lens, vals, q, cp, cs = make_case(BIG, d, seed, alpha_pool)
        name = f"n{BIG}_{d}_sampled"
        idx = np.arange(0, BIG, STRIDE)
        out[name + "/lens"] = lens.numpy().astype(np.int32)
        out[name + "/vals"] = vals.numpy()
        out[name + "/xq"] = q.numpy().astype(np.uint16)
        out[name + "/idx"] = idx.astype(np.int32)
        out[name + "/cumprod"] = cp.numpy()[idx]
        out[name + "/cumsum"] = cs.numpy()[idx]
        out[name + "/cumprod_sum64"] = np.array(cp.double().sum().item())
        out[name + "/cumsum_sum64"] = np.array(cs.double().sum().item())
    path = os.path.join(HERE, "scan_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
