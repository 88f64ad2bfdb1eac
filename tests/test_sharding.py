"""Multi-GPU path on CPU: world_size-2 gloo processes.  Each rank scans ITS slice (with the oracle —
this is a test) and one gather / one scatter move per-group rows; results must equal the unsharded
computation exactly (groups are independent, so sharding changes no arithmetic)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from simplegaussiansplat_tk71_amd import sharding, synthetic


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_partition_is_balanced_and_on_group_boundaries():
    p = synthetic.make_pairs(64, 96, 20.0, deep=True, seed=5)
    for world in (1, 2, 3, 8):
        shards = sharding.partition_groups(p.inv_len, world)
        assert len(shards) == world
        assert shards[0].pair_start == 0 and shards[-1].pair_end == p.n_pairs
        assert shards[0].group_start == 0 and shards[-1].group_end == p.n_groups
        for a, b in zip(shards, shards[1:]):
            assert a.pair_end == b.pair_start and a.group_end == b.group_start
        for s in shards:
            if s.n_groups:
                assert s.pair_end == int(p.inv_len[s.group_end - 1])
                assert s.pair_start == (int(p.inv_len[s.group_start - 1]) if s.group_start else 0)
            # balance: no slice is off its share by more than the longest group
            assert abs(s.n_pairs - p.n_pairs / world) <= int(p.run_len.max()) + 1
            key, x, inv, inv_len = sharding.local_arrays(s, p.key, p.x, p.inv, p.inv_len)
            assert key.numel() == s.n_pairs
            if s.n_groups:
                assert int(inv.min()) == 0 and int(inv.max()) == s.n_groups - 1 and int(inv_len[-1]) == s.n_pairs


def test_shards_from_counts():
    sh = sharding.shards_from_counts([3, 0, 5], [30, 0, 42])
    assert [(s.group_start, s.group_end, s.pair_start, s.pair_end) for s in sh] == [(0, 3, 0, 30), (3, 3, 30, 30), (3, 8, 30, 72)]
    assert [s.rank for s in sh] == [0, 1, 2]


def test_partition_degenerate():
    one = torch.tensor([10], dtype=torch.int32)
    shards = sharding.partition_groups(one, 4)
    assert sum(s.n_groups for s in shards) == 1 and sum(s.n_pairs for s in shards) == 10
    empty = sharding.partition_groups(torch.zeros(0, dtype=torch.int32), 2)
    assert all(s.n_pairs == 0 and s.n_groups == 0 for s in empty)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import c_oracle as co

        h, w = 40, 64
        p = synthetic.make_pairs(h, w, 16.0, deep=True, seed=9)
        colour = torch.rand(p.n_pairs, 3, generator=torch.Generator().manual_seed(1))
        shards = sharding.partition_groups(p.inv_len, world)
        s = shards[rank]
        key, x, inv, inv_len, col = sharding.local_arrays(s, p.key, p.x, p.inv, p.inv_len, colour)
        # forward on this rank's slice only: transmittance, then per-pixel colour sum
        cp = co.cumprod_forward(x.contiguous(), key.contiguous())
        T = cp / x  # exclusive (gs_model.py:562)
        pix = torch.zeros(s.n_groups, 3).index_add_(0, inv.long(), T[:, None] * col)
        group_key = torch.unique_consecutive(p.key)
        frame_rows = sharding.gather_groups(pix, shards, dst=0)
        # backward: rank 0 owns dL/dI, scatters the rows of every band to its owner
        gI = torch.randn(h + 1, w + 1, 3, generator=torch.Generator().manual_seed(2))
        full_rows = sharding.groups_from_frame(gI, group_key) if rank == 0 else None
        my_rows = sharding.scatter_groups(full_rows, shards, like=pix, src=0)
        pair_grad = (my_rows[inv.long()] * (T[:, None] * col)).sum(1)  # (g . p) per pair, gs_model.py:632
        suffix = co.cumsum_reverse(pair_grad.contiguous(), key.contiguous())
        local = {"suffix": suffix, "rows": my_rows}
        if rank == 0:
            # unsharded reference computation
            cp_all = co.cumprod_forward(p.x, p.key)
            T_all = cp_all / p.x
            pix_all = torch.zeros(p.n_groups, 3).index_add_(0, p.inv.long(), T_all[:, None] * colour)
            assert torch.equal(frame_rows, pix_all)
            frame = sharding.frame_from_groups(frame_rows, group_key, h, w)
            assert frame.shape == (h + 1, w + 1, 3)
            assert torch.equal(sharding.groups_from_frame(frame, group_key), pix_all)
            rows_all = sharding.groups_from_frame(gI, group_key)
            pg_all = (rows_all[p.inv.long()] * (T_all[:, None] * colour)).sum(1)
            suffix_all = co.cumsum_reverse(pg_all, p.key)
            local["suffix_all"] = suffix_all
            local["rows_all"] = rows_all
        q.put((rank, s.pair_start, s.pair_end, s.group_start, s.group_end, {k: v.clone() for k, v in local.items()}))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gloo_forward_gather_backward_scatter():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = [q.get(timeout=150) for _ in range(world)]
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    got.sort(key=lambda t: t[0])
    ref = got[0][5]
    for rank, p0, p1, g0, g1, loc in got:
        assert torch.equal(loc["suffix"], ref["suffix_all"][p0:p1])
        assert torch.equal(loc["rows"], ref["rows_all"][g0:g1])


def test_row_bands_cover_the_frame_on_tile_boundaries():
    for h, world in ((1079, 8), (1079, 3), (47, 2), (15, 4), (2159, 8)):
        bands = sharding.row_bands(h, world)
        rows = [y for (y0, y1) in bands for y in range(y0, y1 + 1)]
        assert rows == list(range(h + 1))
        assert all(y0 % 16 == 0 for (y0, y1) in bands if y1 >= y0)


def test_row_bands_by_pairs_balance_a_scene_that_crowds_one_region():
    """Bands cut by work: the same coverage rules as `row_bands` (whole tile rows, contiguous, the whole frame), and on a scene
    whose boxes crowd the middle of the frame the heaviest band holds little more than its share of the pairs where equal-row
    bands leave most of them to the middle ranks."""
    g = torch.Generator().manual_seed(5)
    h, w, n, world = 1079, 1919, 20000, 8
    cy = (torch.randn(n, generator=g) * (h / 8) + h / 2).round().clamp(0, h).long()
    cx = torch.randint(0, w + 1, (n,), generator=g)
    half = torch.randint(1, 9, (n, 2), generator=g)
    start = torch.stack([(cx - half[:, 0]).clamp(min=0), (cy - half[:, 1]).clamp(min=0)], 1).to(torch.int32)
    end = torch.stack([(cx + half[:, 0]).clamp(max=w), (cy + half[:, 1]).clamp(max=h)], 1).to(torch.int32)

    def pairs_in(band):
        y0, y1 = band
        if y1 < y0:
            return 0
        rows = (torch.minimum(end[:, 1].long(), torch.tensor(y1)) - torch.maximum(start[:, 1].long(), torch.tensor(y0)) + 1).clamp(min=0)
        return int((rows * (end[:, 0] - start[:, 0] + 1).long()).sum())

    total = pairs_in((0, h))
    for world in (8, 3, 2):
        bands = sharding.row_bands_by_pairs(start, end, h, world)
        rows = [y for (y0, y1) in bands for y in range(y0, y1 + 1)]
        assert rows == list(range(h + 1)) and all(y0 % 16 == 0 for (y0, y1) in bands if y1 >= y0)
        assert sum(pairs_in(b) for b in bands) == total
        heaviest = max(pairs_in(b) for b in bands)
        even = max(pairs_in(b) for b in sharding.row_bands(h, world))
        one_tile_row = max(pairs_in((t * 16, min(t * 16 + 15, h))) for t in range((h + 16) // 16))
        assert heaviest <= total / world + one_tile_row          # within one tile row of the ideal share
        if world == 8:
            assert even > 2.5 * total / world and heaviest < 1.5 * total / world
    # degenerate inputs: no boxes at all -> the equal-row bands; more ranks than tile rows -> empty bands, still a cover
    assert sharding.row_bands_by_pairs(start[:0], end[:0], 47, 3) == sharding.row_bands(47, 3)
    bands = sharding.row_bands_by_pairs(start, end.clamp(max=29), 29, 5)
    assert [y for (y0, y1) in bands for y in range(y0, y1 + 1)] == list(range(30))


def _band_worker(rank, world, port, q, w=40, h=45):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import dense_render as dr
        from tests.util import make_scene

        sc = make_scene(60, w, h, 9, seed=3)
        bands = sharding.row_bands(h, world)
        y0, y1 = bands[rank]
        gI_full = sc["wimg"].double() if rank == 0 else None
        like = torch.zeros(max(y1 - y0 + 1, 0), w + 1, 3, dtype=torch.float64)
        gI = sharding.scatter_bands(gI_full, bands, like, src=0)  # backward input: dL/dI rows of this band

        def dense(s, e, m, vinv, op, l_d, width, height, grad):  # the dense oracle stands in for the HIP kernels on CPU
            img, gv, go, gl = dr.render_with_grads(s, e, m, vinv, op, l_d, width, height, grad)
            return img, (gv, go, gl)

        img, (gv, go, gl) = sharding.blend_band(bands[rank], sc["start"], sc["end"], sc["mean"], sc["vinv"].double(),
                                                sc["opacity"].double(), sc["l_d"].double(), w, gI, blend=dense)
        frame = sharding.gather_bands(img, bands, dst=0)
        gv, go, gl = sharding.allreduce_gaussian_grads(gv, go, gl)
        if rank == 0:
            full = dr.render_with_grads(sc["start"], sc["end"], sc["mean"], sc["vinv"], sc["opacity"], sc["l_d"], w, h, sc["wimg"])
            ok = (torch.equal(frame, full[0]) and torch.allclose(gv, full[1], atol=1e-12) and torch.allclose(go, full[2], atol=1e-12)
                  and torch.allclose(gl, full[3], atol=1e-12))
            q.put(ok)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gloo_band_sharded_function():
    """Function-level sharding: each rank blends ITS row band (dense oracle here), one gather builds the frame,
    one scatter distributes dL/dI, one all-reduce sums the per-Gaussian gradients: equals the unsharded result."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_band_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    assert q.get(timeout=150) is True
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0


@pytest.mark.timeout(180)
def test_more_ranks_than_tile_rows_empty_bands_join_the_collectives():
    """A 30-row frame has two 16-pixel tile rows; with 3 ranks one band is empty.  That rank must contribute a
    zero-row band and zero gradients and still enter every collective (it used to raise on a negative height while
    the others waited)."""
    world = 3
    assert any(sharding.band_is_empty(b) for b in sharding.row_bands(29, world))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_band_worker, args=(r, world, port, q, 40, 29)) for r in range(world)]
    for pr in procs:
        pr.start()
    assert q.get(timeout=150) is True
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0


def _bench_worker(rank, world, port, q):
    """The multi-GPU block of bench.py (sharded_frame) with the oracle standing in for the HIP scans and gloo for RCCL."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from oracle import c_oracle as co

        seen = {}

        def scan_fwd_bwd(p):
            y = co.cumprod_forward(p.x, p.key)
            g = co.cumprod_backward_f64(p.x, y, p.grad_out, p.inv)
            seen["checksum"] = (float(y.double().sum()), float(g.sum()), p.n_pairs, p.n_groups)

        def sync():
            dist.barrier()

        out = bench.sharded_frame("cfg1", world, rank, torch.device("cpu"), scan_fwd_bwd, sync, steps=2, warmup=1)
        q.put((rank, out, seen["checksum"]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("world", [2, 3])
def test_bench_sharded_frame_block_gloo_rehearsal(world):
    """bench.py --gpus N adds a `sharded_frames` block (BASELINE.json config 5: ONE frame cut by partition_groups,
    pairs/s with and without the frame gather + gradient scatter).  Rehearsal of exactly that function on two gloo ranks:
    both ranks report the same numbers, the slices tile the frame, the collectives ran (no collective_error) and their
    round trip is the identity."""
    from simplegaussiansplat_tk71_amd import synthetic

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = sorted((q.get(timeout=200) for _ in range(world)), key=lambda t: t[0])
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    o0 = got[0][1]
    assert all(g[1] == o0 for g in got)  # every rank holds the same report (times are max-reduced)
    assert o0["collective_error"] is None
    assert o0["frame_gather_ms"] is not None and o0["grad_scatter_ms"] is not None
    assert o0["pairs_per_s_with_gather_scatter"] < o0["pairs_per_s_scan_only"]
    # SURVEY §8e reporting: per-GPU rate / GB/s and efficiency against N copies of rank 0's own slice rate
    assert abs(o0["per_gpu_pairs_per_s"] * world - o0["pairs_per_s_scan_only"]) <= 1e-6 * o0["pairs_per_s_scan_only"]
    assert abs(o0["per_gpu_GBps"] - 32 * o0["per_gpu_pairs_per_s"] / 1e9) <= 1e-9 * o0["per_gpu_GBps"] + 1e-12
    assert o0["rank0_slice_pairs_per_s"] > 0 and 0.0 < o0["efficiency_vs_single_gpu"] <= 1.5
    whole = synthetic.make_config("cfg1", seed=0)  # same seed: the slices are a cut of this frame's run lengths
    assert o0["total_pairs"] == whole.n_pairs == sum(o0["pairs_per_rank"])
    assert sum(o0["groups_per_rank"]) == whole.n_groups
    assert tuple(g[2][2] for g in got) == tuple(o0["pairs_per_rank"]) and tuple(g[2][3] for g in got) == tuple(o0["groups_per_rank"])
    assert max(o0["pairs_per_rank"]) - min(o0["pairs_per_rank"]) <= 64  # balanced by pairs, cut at a group boundary


def test_bench_sharded_frame_reports_a_collective_failure(monkeypatch):
    """A failing collective must end up in the JSON line (`collective_error`), not on stderr only."""
    import bench

    class Boom(RuntimeError):
        pass

    def broken(*a, **k):
        raise Boom("RCCL says no")

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        monkeypatch.setattr(sharding, "gather_groups", broken)
        # world = 1 skips the collectives; pretend to be one rank of two for the branch that times them
        monkeypatch.setattr(sharding, "partition_groups", lambda ends, w: [sharding.Shard(0, 0, ends.numel(), 0, int(ends[-1])),
                                                                            sharding.Shard(1, ends.numel(), ends.numel(), int(ends[-1]), int(ends[-1]))])
        out = bench.sharded_frame("cfg1", 2, 0, torch.device("cpu"), lambda p: None, lambda: None, steps=1, warmup=0)
    finally:
        dist.destroy_process_group()
    assert out["collective_error"] is not None and "RCCL says no" in out["collective_error"]
    assert out["frame_gather_ms"] is None and out["pairs_per_s_with_gather_scatter"] is None


@pytest.mark.timeout(300)
def test_bench_without_a_launcher_starts_its_ranks_and_returns_their_exit_code():
    """`python bench.py --gpus 2` with no RANK / WORLD_SIZE in the environment must not stop at a usage message: it starts
    `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2` as a child and returns the child's exit
    code.  On a box with fewer than two GPUs the ranks refuse (RCCL needs a GPU per rank) — which is what proves here
    that they were started and that their failure reaches the caller's return code."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                           "GCP_BENCH_BACKEND")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=280)
    assert "starting 2 ranks" in res.stderr and "torch.distributed.run" in res.stderr
    if torch.cuda.device_count() < 2:
        assert res.returncode != 0
        assert "needs 2 GPUs" in res.stderr
        assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
