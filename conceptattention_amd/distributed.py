"""Work-item sharding over the GPUs of one node and the final heat-map gather.

The reference has no multi-GPU layer; its own multi-GPU use is one process per
`--device cuda:N`.  The path shards across *independent work items* only (the
timesteps inside one generate_image call are a sequential recurrence and the concepts are coupled
through the joint softmax, so neither can be split exactly): item j runs on rank j % world_size
with a full weight replica per GPU (23.8 GB of 288 GB), and the only collective is one RCCL
all_gather of the small (C,h,w) fp32 heat maps at the end.  Contract: the N-GPU result equals the
1-GPU result item for item (same kernels, same seeds), returned in item order.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """One process per GPU as launched by torch.distributed.run: reads RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_*.  Returns (rank, world, local_rank).  backend "nccl" is RCCL on ROCm."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("CA_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_items(n_items: int, rank: int, world: int) -> list[int]:
    """Round-robin: rank r owns items r, r+world, ...  (SURVEY.md §8e partitioning 1)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return list(range(rank, n_items, world))


def gather_heatmaps(local_maps: torch.Tensor, n_items: int, rank: int, world: int,
                    force_collective: bool = False) -> torch.Tensor:
    """local_maps [n_local, ...] for items shard_items(n_items, rank, world) -> [n_items, ...] in item
    order on every rank.  One all_gather of the padded per-rank block (ranks may own one item fewer).
    ``force_collective``: run the collective even in a one-rank group (exercises the RCCL path on a 1-GPU box)."""
    mine = shard_items(n_items, rank, world)
    if local_maps.shape[0] != len(mine):
        raise ValueError(f"rank {rank} holds {local_maps.shape[0]} maps but owns {len(mine)} items")
    if world == 1 and not (force_collective and dist.is_initialized()):
        return local_maps
    per = (n_items + world - 1) // world
    # RCCL works on the device tensors directly; gloo (CPU rehearsals) goes through host memory
    dev = local_maps.device if dist.get_backend() == "nccl" else torch.device("cpu")
    pad = torch.zeros((per,) + tuple(local_maps.shape[1:]), dtype=local_maps.dtype, device=dev)
    pad[: len(mine)] = local_maps
    out = torch.empty((world, per) + tuple(local_maps.shape[1:]), dtype=local_maps.dtype, device=dev)
    dist.all_gather_into_tensor(out.view(world * per, *local_maps.shape[1:]), pad)
    full = torch.empty((n_items,) + tuple(local_maps.shape[1:]), dtype=local_maps.dtype, device=dev)
    for r in range(world):
        idx = shard_items(n_items, r, world)
        full[idx] = out[r, : len(idx)]
    return full.to(local_maps.device)


def allreduce_sum_(acc: torch.Tensor, force_collective: bool = False) -> torch.Tensor:
    """Partial sums of a sharded noise-level sweep (SURVEY.md §8e partitioning 2): one all_reduce of
    the fp32 (C, patches) accumulator."""
    if dist.is_initialized() and (dist.get_world_size() > 1 or force_collective):
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    return acc


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def collective_evidence(device, force_collective: bool = False) -> dict | None:
    """What a reader of the bench line needs to believe that N ranks took part: the backend, how many ranks an
    all_reduce(sum) of 1 saw, and the device every rank ran on (all_gather of its index and its PCI bus id), so that
    a job whose ranks all sat on one GPU, or whose group is smaller than --gpus, shows in its own JSON."""
    if not (dist.is_initialized() and (dist.get_world_size() > 1 or force_collective)):
        return None
    world, backend = dist.get_world_size(), dist.get_backend()
    on = torch.device(device) if backend == "nccl" else torch.device("cpu")
    one = torch.ones(1, dtype=torch.int64, device=on)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    dev = torch.device(device)
    idx = dev.index if dev.type == "cuda" and dev.index is not None else -1
    bus = -1
    if dev.type == "cuda" and torch.cuda.is_available():
        try:
            bus = int(torch.cuda.get_device_properties(dev).pci_bus_id)
        except (AttributeError, RuntimeError):
            bus = -1
    # a host identifier as well: ranks on different nodes may share a local index and a bus id
    import socket
    import zlib
    host = zlib.crc32(socket.gethostname().encode()) & 0x7FFFFFFF
    mine = torch.tensor([idx, bus, host], dtype=torch.int64, device=on)
    out = torch.empty(world * 3, dtype=torch.int64, device=on)
    dist.all_gather_into_tensor(out, mine)
    rows = out.view(world, 3).cpu().tolist()
    return {"backend": "rccl (torch.distributed 'nccl')" if backend == "nccl" else backend,
            "ranks_seen": int(one.item()), "devices": [r[0] for r in rows], "pci_bus_ids": [r[1] for r in rows],
            "hosts": len({r[2] for r in rows}), "distinct_devices": len({tuple(r) for r in rows})}


def check_collective_evidence(ev: dict | None, world: int, rehearsal: bool = False) -> None:
    """A multi-rank run whose group is not what --gpus asked for must not print a line that looks like one (VERDICT r04
    #7a): with the RCCL backend, ``ranks_seen`` and ``distinct_devices`` must both equal the world size.  gloo runs and
    CA_BENCH_DEVICE rehearsals (several ranks on one GPU on purpose) are exempt (``rehearsal``)."""
    if ev is None or world <= 1 or rehearsal or not str(ev.get("backend", "")).startswith("rccl"):
        return
    # (the device count is conclusive only where the PCI bus ids are known: a launcher that masks devices per rank gives
    # every rank local index 0, and without bus ids that cannot be told from ranks sharing one GPU)
    buses_known = all(b is not None and b >= 0 for b in ev.get("pci_bus_ids", [-1]))
    if ev.get("ranks_seen") != world or (ev.get("distinct_devices") != world and
                                         (buses_known or "pci_bus_ids" not in ev)):
        raise SystemExit(f"bench: --gpus {world} but the collectives saw {ev.get('ranks_seen')} ranks on "
                         f"{ev.get('distinct_devices')} distinct devices: {ev}")


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
