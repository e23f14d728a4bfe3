"""One process per GPU, data parallel over independent scenes.

The hot path has no data-path collective: every rank renders its own batch.  What lives
here is only the launch contract plumbing `bench.py` and training scripts share: rendezvous
from the torchrun environment (RCCL = backend "nccl" on ROCm; gloo on CPU for tests),
barrier, max-over-ranks timing, per-rank seeds (the reference seeds per rank too:
/root/reference/src/main.py:106, src/dataset/data_module.py:83-88).
"""
from __future__ import annotations

import os

import torch


def env_rank() -> tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """Joins the process group described by RANK / WORLD_SIZE / MASTER_* (no-op for 1 rank)."""
    rank, world, local = env_rank()
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    elif torch.cuda.is_available():
        torch.cuda.set_device(0)
    return rank, world, local


def barrier(world: int) -> None:
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(value: float, world: int, device: torch.device | str = "cpu") -> float:
    if world <= 1:
        return float(value)
    import torch.distributed as dist

    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def rank_seed(base: int, rank: int) -> int:
    """Distinct synthetic batches per rank (weak scaling: per-GPU work fixed)."""
    return base + rank


def aggregate_throughput(units_per_rank_step: int, steps: int, world: int, elapsed_max: float) -> float:
    """Whole-job units/s: all ranks' units over the slowest rank's time."""
    return world * units_per_rank_step * steps / elapsed_max


def shutdown(world: int) -> None:
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()
