"""The N>1 path on CPU: world_size-2 gloo processes shard the work items and gather heat maps."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conceptattention_amd.distributed import gather_heatmaps, shard_items


def test_shard_items_partition():
    for n in (0, 1, 5, 8, 17):
        for w in (1, 2, 3, 8):
            got = sorted(i for r in range(w) for i in shard_items(n, r, w))
            assert got == list(range(n))
    assert shard_items(5, 1, 2) == [1, 3]
    with pytest.raises(ValueError):
        shard_items(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_items, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_items(n_items, rank, world)
    local = torch.stack([torch.full((4, 8, 8), float(i)) + torch.arange(4.0)[:, None, None] for i in mine]) \
        if mine else torch.zeros(0, 4, 8, 8)
    full = gather_heatmaps(local, n_items, rank, world)
    ok = all(torch.equal(full[i], torch.full((4, 8, 8), float(i)) + torch.arange(4.0)[:, None, None])
             for i in range(n_items))
    q.put((rank, ok, tuple(full.shape)))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [4, 5])
def test_gather_heatmaps_gloo_world2(n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res), res
    assert all(shape == (n_items, 4, 8, 8) for _, _, shape in res)


def test_single_rank_is_identity():
    x = torch.randn(3, 4, 8, 8)
    assert gather_heatmaps(x, 3, 0, 1) is x


def _sweep_worker(rank, world, port, q):
    """Noise-level sharding of SURVEY.md §8e-2: each rank fills the rows of its own levels, one all_reduce(sum)
    completes the table; max_over_ranks / barrier are the bench's timing helpers."""
    from conceptattention_amd.distributed import allreduce_sum_, barrier, max_over_ranks
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_levels = 5
    table = torch.zeros(n_levels, 3, 4)
    for li in range(rank, n_levels, world):            # the loop of layer_noise_sweep_on_device
        table[li] = float(li + 1)
    allreduce_sum_(table)
    ok = all(torch.equal(table[li], torch.full((3, 4), float(li + 1))) for li in range(n_levels))
    slowest = max_over_ranks(1.0 + rank, "cpu")
    barrier()
    q.put((rank, ok, slowest))
    dist.destroy_process_group()


def test_sharded_sweep_allreduce_and_timing_helpers_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sweep_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res), res
    assert all(s == 2.0 for _, _, s in res), res
