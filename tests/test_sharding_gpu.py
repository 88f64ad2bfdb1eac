"""SURVEY.md §8(e) on the HIP kernels: every rank runs grouped_cumprod_forward / grouped_cumprod_backward (and the
suffix-sum scan) on ITS slice of the pair list — `sharding.local_arrays`: contiguous views with group ids and end offsets
rebased to the slice — and one gather of per-group rows assembles the frame.  Ranks are gloo processes sharing the one
GPU of the test box (RCCL needs a GPU per rank); the kernels, the slices and the rebased `inv` / `inv_len` are exactly
what an 8-GPU run uses.

Two kinds of input:
  * exact arithmetic (factors 2 and 1/2, small-integer gradients): every product and sum is exact in fp32 whatever the
    association, so the sharded result must equal the single-rank run over the whole list BIT FOR BIT — any slip in the
    slices, the rebasing or the gather shows as a wrong bit;
  * the BASELINE value distribution: a slice starts at another offset inside the scan's 4096-element tiles than the
    same pairs have in the whole list, so the fp32 association differs; held to the north-star's 1e-5 against the
    oracle, and bit for bit against a single-process run over the same slices (the result may not depend on the rank).
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from simplegaussiansplat_tk71_amd import sharding

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _pair_list(lens, exact, seed):
    """key / x / inv / inv_len / grad_out of groups with the given lengths (CPU tensors)."""
    lens = torch.as_tensor(lens, dtype=torch.long)
    g = torch.Generator().manual_seed(seed)
    n = int(lens.sum())
    gid = torch.repeat_interleave(torch.arange(lens.numel()), lens)
    key = ((gid * 7919) % 100003).to(torch.int32)  # equal keys only inside a group, not globally sorted
    inv = gid.to(torch.int32)
    inv_len = torch.cumsum(lens, 0).to(torch.int32)
    if exact:
        start = torch.repeat_interleave(inv_len.long() - lens, lens)
        pos = torch.arange(n) - start
        x = torch.where(pos % 2 == 0, torch.tensor(2.0), torch.tensor(0.5))  # inclusive products alternate 2, 1, 2, 1, ...
        grad_out = torch.randint(-4, 5, (n,), generator=g).float()
    else:
        a = torch.sigmoid(torch.randn(n, generator=g) * 2.0 + 1.7).clamp_(0.005, 0.995)
        x = 1.0 - a * torch.rand(n, generator=g)
        grad_out = torch.randn(n, generator=g)
    return key, x.contiguous(), inv, inv_len, grad_out.contiguous()


def _scan_slice(dev, key, x, inv, inv_len, go):
    """The three scans of one slice on the HIP library -> (y, grad_in, suffix) on the device."""
    import grouped_cumprod as gc

    key, x, inv, inv_len, go = (t.contiguous().to(dev) for t in (key, x, inv, inv_len, go))
    y, gin, suf = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    gc.grouped_cumprod_forward(x, key, y)
    gc.grouped_cumprod_backward(x, y, go, inv, gin, inv_len)
    gc.grouped_cumsum_reverse(go, key, suf)
    return y, gin, suf


def _group_rows(y, gin, inv_len):
    """Per-group rows of a slice: the transmittance behind the whole list (inclusive product at the group's last pair) and
    the gradient at its first pair — one row per pixel group, what a frame gather moves."""
    if inv_len.numel() == 0:
        return y.new_zeros((0, 2))
    last = inv_len.long().to(y.device) - 1
    first = torch.cat([last.new_zeros(1), last[:-1] + 1])
    return torch.stack([y[last], gin[first]], 1)


CASES = {
    # the cut k*M/N falls inside a long group's neighbourhood: groups of 5000-13000 pairs (longer than one scan tile)
    # between short ones, so a slice begins right behind / in front of a group that spans tiles
    "long_groups_at_the_cut": lambda: [7, 9000, 3, 12, 12011, 5, 6001, 40, 8191, 4097, 2, 4096, 13000, 1, 77, 5123],
    # one group holds nearly everything: with three ranks the first cut collapses to 0 and rank 0 gets an EMPTY shard
    "empty_shard": lambda: [20000, 10, 10],
    "many_short": lambda: torch.poisson(torch.full((6000,), 8.0), generator=torch.Generator().manual_seed(4)).long().clamp_(min=1).tolist(),
}


def _worker(rank, world, port, case, exact, out_dir, backend="gloo"):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    # gloo: the ranks share GPU 0 (the test box has one); nccl = RCCL over xGMI: one GPU per rank, as bench.py runs
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        key, x, inv, inv_len, go = _pair_list(CASES[case](), exact, seed=11)
        shards = sharding.partition_groups(inv_len, world)
        s = shards[rank]
        lk, lx, linv, llen, lgo = sharding.local_arrays(s, key, x, inv, inv_len, go)
        y, gin, suf = _scan_slice(dev, lk, lx, linv, llen, lgo)
        rows = _group_rows(y, gin, llen)
        frame_rows = sharding.gather_groups(rows, shards, dst=0)  # device rows; gloo stages them through the host
        back = sharding.scatter_groups(frame_rows, shards, like=rows, src=0)
        assert torch.equal(back, rows), "gather -> scatter round trip is not the identity"
        torch.save({"y": y.cpu(), "gin": gin.cpu(), "suf": suf.cpu(), "shard": (s.pair_start, s.pair_end, s.group_start, s.group_end),
                    "frame_rows": None if frame_rows is None else frame_rows.cpu()}, os.path.join(out_dir, f"rank{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("case,world", [("long_groups_at_the_cut", 2), ("long_groups_at_the_cut", 3), ("empty_shard", 3), ("many_short", 2)])
@pytest.mark.parametrize("exact", [True, False], ids=["exact_arithmetic", "baseline_values"])
def test_sharded_hip_scans_equal_the_single_rank_frame(device, tmp_path, case, world, exact):
    from oracle import c_oracle as co

    mp.spawn(_worker, args=(world, _free_port(), case, exact, str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(str(tmp_path / f"rank{r}.pt")) for r in range(world)]
    key, x, inv, inv_len, go = _pair_list(CASES[case](), exact, seed=11)
    shards = sharding.partition_groups(inv_len, world)
    if case == "empty_shard":
        assert shards[0].n_pairs == 0 and shards[0].n_groups == 0
    # the slices tile the list
    assert [g["shard"] for g in got] == [(s.pair_start, s.pair_end, s.group_start, s.group_end) for s in shards]
    y = torch.cat([g["y"] for g in got])
    gin = torch.cat([g["gin"] for g in got])
    suf = torch.cat([g["suf"] for g in got])
    assert y.numel() == x.numel()
    # single rank, whole list, same kernels
    y1, gin1, suf1 = (t.cpu() for t in _scan_slice(device, key, x, inv, inv_len, go))
    rows1 = _group_rows(y1, gin1, inv_len)
    if exact:
        assert torch.equal(y, y1) and torch.equal(gin, gin1) and torch.equal(suf, suf1)
        assert torch.equal(got[0]["frame_rows"], rows1)
    else:
        want_y = co.cumprod_forward(x, key)
        assert (y - want_y).abs().max().item() <= 1e-5 and (y1 - want_y).abs().max().item() <= 1e-5
        want_g = co.cumprod_backward(x, want_y, go, inv, inv_len)
        scale = co.cumprod_backward_f64(x, want_y, go.abs(), inv)
        assert bool(((gin.double() - want_g.double()).abs() <= 1e-5 * (1.0 + scale)).all())
        want_s = co.cumsum_reverse(go, key)
        scale_s = co.cumsum_reverse(go.abs(), key).double()
        assert bool(((suf.double() - want_s.double()).abs() <= 1e-5 * (1.0 + scale_s)).all())
        # the result of a slice does not depend on which rank (process) scanned it
        for s, g in zip(shards, got):
            lk, lx, linv, llen, lgo = sharding.local_arrays(s, key, x, inv, inv_len, go)
            ys, gs, ss = (t.cpu() for t in _scan_slice(device, lk, lx, linv, llen, lgo))
            assert torch.equal(ys, g["y"]) and torch.equal(gs, g["gin"]) and torch.equal(ss, g["suf"])
    # the gathered frame = the slices' rows in order, bit for bit
    rows = torch.cat([_group_rows(g["y"], g["gin"], sharding.local_arrays(s, key, x, inv, inv_len)[3]) for s, g in zip(shards, got)])
    assert torch.equal(got[0]["frame_rows"], rows)
    assert all(g["frame_rows"] is None for g in got[1:])


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_hip_scans_over_rccl(device, tmp_path, world):
    """The same on RCCL — one GPU per rank, the rows gathered and scattered GPU to GPU over xGMI — wherever the node has the
    GPUs (skipped on a one-GPU box; the driver's scaling run is the first place with more)."""
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, this node shows {torch.cuda.device_count()}")
    case = "long_groups_at_the_cut"
    mp.spawn(_worker, args=(world, _free_port(), case, True, str(tmp_path), "nccl"), nprocs=world, join=True)
    got = [torch.load(str(tmp_path / f"rank{r}.pt")) for r in range(world)]
    key, x, inv, inv_len, go = _pair_list(CASES[case](), True, seed=11)
    y1, gin1, suf1 = (t.cpu() for t in _scan_slice(device, key, x, inv, inv_len, go))
    assert torch.equal(torch.cat([g["y"] for g in got]), y1) and torch.equal(torch.cat([g["gin"] for g in got]), gin1)
    assert torch.equal(torch.cat([g["suf"] for g in got]), suf1)
    assert torch.equal(got[0]["frame_rows"], _group_rows(y1, gin1, inv_len))


@pytest.mark.timeout(900)
def test_bench_launches_its_own_ranks(device):
    """`python bench.py --gpus 2` with no launcher in the environment starts its two ranks itself (a fresh
    torch.distributed.run child; the parent never touches the GPU), relays rank 0's line and returns the child's exit
    code.  gloo here: the two ranks share the one GPU of the test box."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["GCP_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "cfg2"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["world_size_checked"] == 2 and out["steps"] == 3
    assert out["value"] > 0 and out["scaling"] == "weak"
    # a collective that never returns: the watchdog prints the headline and every rank leaves with a NON-ZERO code
    env["GCP_BENCH_EXTRAS_TIMEOUT"] = "0"
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    assert res.returncode != 0
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and "error" in out["sharded_frames"]
