"""One process per GPU.  The VTC path shards at file granularity, so ranks never exchange data; the only
communication is the benchmark's barrier + max-over-ranks timing (RCCL via backend "nccl" on GPUs,
gloo on CPU for tests)."""

from __future__ import annotations

import os

import torch


def env_rank() -> tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend: str | None = None) -> tuple[int, int, int]:
    rank, world, local_rank = env_rank()
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            use_gpu = backend != "gloo" and torch.cuda.is_available()
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if use_gpu:
                dist.init_process_group(backend or "nccl", device_id=torch.device(f"cuda:{local_rank}"))
            else:
                dist.init_process_group(backend or "gloo")
    return rank, world, local_rank


def _dev():
    import torch.distributed as dist
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def barrier() -> None:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(x: float) -> float:
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(x)
    t = torch.tensor([x], dtype=torch.float64, device=_dev())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x: float) -> float:
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(x)
    t = torch.tensor([x], dtype=torch.float64, device=_dev())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def shutdown() -> None:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
