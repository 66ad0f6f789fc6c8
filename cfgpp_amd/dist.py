"""Multi-GPU layer: one process per GPU, prompts sharded over ranks, ONE broadcast
of the shared conditioning, independent sampling chains, optional gather.

The reference has no multi-GPU path at all (single process, single device:
examples/text_to_img.py:16).  Sampling chains are independent units - nothing
is exchanged inside the NFE loop - so the only collective is a broadcast of the
conditioning block from rank 0 (RCCL over xGMI on the GPU box,
``backend="nccl"``; ``gloo`` in the CPU tests) and, if the caller wants all
results on one rank, a gather of the final latents / images.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process if unset)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend: Optional[str] = None) -> Tuple[int, int, int]:
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition [lo, hi) of n_items chains for this rank (sizes differ by <= 1)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_conditioning(tensors: Sequence[Optional[torch.Tensor]], shapes_dtypes, device, src: int = 0):
    """Broadcast the conditioning block (null-prompt embeds, per-prompt embeds, pooled embeds,
    time ids ...) from ``src`` to every rank.  ``shapes_dtypes`` = [(shape, dtype), ...] is known
    on every rank (it only depends on the model and the number of prompts); non-src ranks pass
    ``None`` tensors.  Payloads are packed into ONE flat byte buffer -> a single collective."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        return [t.to(device) for t in tensors]
    sizes = []
    for shape, dtype in shapes_dtypes:
        n = 1
        for d in shape:
            n *= d
        sizes.append(n * torch.empty((), dtype=dtype).element_size())
    offs = [0]
    for s in sizes:
        offs.append(offs[-1] + ((s + 15) // 16) * 16)
    buf = torch.zeros(offs[-1], dtype=torch.uint8, device=device)
    if rank == src:
        for t, o, s in zip(tensors, offs, sizes):
            buf[o:o + s] = t.contiguous().to(device).view(torch.uint8).reshape(-1)
    dist.broadcast(buf, src=src)
    out = []
    for (shape, dtype), o, s in zip(shapes_dtypes, offs, sizes):
        out.append(buf[o:o + s].view(dtype).reshape(shape).clone())
    return out


def gather_rows(local: torch.Tensor, counts: List[int], dst: int = 0) -> Optional[torch.Tensor]:
    """Gather per-rank result rows (latents or images) to ``dst`` in prompt order.
    ``counts[r]`` = rows held by rank r (from :func:`shard_range`)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return local
    rank = dist.get_rank()
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    if rank != dst:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(x: float, device) -> float:
    if not dist.is_initialized():
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(xs: Sequence[float], device) -> List[List[float]]:
    """every rank's list of floats (equal length) on every rank: [[rank 0's ...], [rank 1's ...], ...]"""
    if not dist.is_initialized():
        return [list(xs)]
    t = torch.tensor(list(xs), dtype=torch.float64, device=device)
    bufs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(bufs, t)
    return [[float(v) for v in b.cpu()] for b in bufs]


def broadcast_ints(values: Optional[Sequence[int]], device, src: int = 0) -> List[int]:
    """a list of ints known on ``src`` only (e.g. the tile configs rank 0's in-situ tuning pinned) -> every rank.
    Two small collectives (length, payload); outside the sampling loop."""
    if not dist.is_initialized():
        return list(values or [])
    rank = dist.get_rank()
    n = torch.tensor([len(values) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src)
    buf = torch.zeros(int(n.item()), dtype=torch.int64, device=device)
    if rank == src:
        buf.copy_(torch.tensor(list(values), dtype=torch.int64))
    dist.broadcast(buf, src=src)
    return [int(v) for v in buf.cpu()]


def gather_strings(text: str, device) -> List[str]:
    """every rank's string on every rank (rank order): the sibling of :func:`gather_floats` for identity records
    (backend, device name, PCI bus id ...), padded to the longest and sent as bytes in one all_gather."""
    if not dist.is_initialized():
        return [text]
    raw = text.encode("utf-8")
    n = torch.tensor([len(raw)], dtype=torch.int64, device=device)
    dist.all_reduce(n, op=dist.ReduceOp.MAX)
    width = int(n.item())
    buf = torch.zeros(width + 8, dtype=torch.uint8, device=device)
    buf[:8] = torch.tensor(list(len(raw).to_bytes(8, "little")), dtype=torch.uint8)
    if raw:
        buf[8:8 + len(raw)] = torch.tensor(list(raw), dtype=torch.uint8)
    bufs = [torch.empty_like(buf) for _ in range(dist.get_world_size())]
    dist.all_gather(bufs, buf)
    out = []
    for b in bufs:
        b = b.cpu()
        ln = int.from_bytes(bytes(b[:8].tolist()), "little")
        out.append(bytes(b[8:8 + ln].tolist()).decode("utf-8"))
    return out


def rank_identity(device) -> str:
    """one line that says what THIS rank is running on: backend, world size, device name + PCI bus id, RCCL version - gathered
    into the bench line so that a multi-GPU record shows N distinct GPUs behind N ranks"""
    import json
    rec = {"rank": dist.get_rank() if dist.is_initialized() else 0,
           "world": dist.get_world_size() if dist.is_initialized() else 1,
           "backend": dist.get_backend() if dist.is_initialized() else "none",
           "pid": os.getpid(), "device": str(device)}
    dev = torch.device(device)
    if dev.type == "cuda":
        props = torch.cuda.get_device_properties(dev)
        rec["name"] = props.name
        rec["pci_bus_id"] = "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0))
        rec["gcn_arch"] = getattr(props, "gcnArchName", "")
        try:
            rec["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            rec["rccl_version"] = "unavailable"
    return json.dumps(rec, sort_keys=True)
