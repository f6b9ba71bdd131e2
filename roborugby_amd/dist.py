"""Multi-GPU plumbing: one process per GPU, arenas sharded with NO data-path collective.

Arenas are independent (the reference is single-arena), so rank r simply owns global arenas
[r*n_local, (r+1)*n_local): the counter-based reset RNG is keyed by the GLOBAL arena id (rr_config.arena_offset),
which makes every arena's trajectory invariant to how the batch is sharded.  The only exchange is a logging-side
all-gather of finished-episode returns (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests)."""
import os

import torch
import torch.distributed as dist


def dist_env():
    """(rank, local_rank, world_size) from the launcher's environment (torch.distributed.run)."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)))


def init_process_group(backend=None):
    rank, local_rank, world = dist_env()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # RR_DIST_BACKEND=gloo lets a 1-GPU box rehearse the N>1 path (RCCL refuses two ranks on one device)
        backend = backend or os.environ.get("RR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_offset(rank, n_local):
    """Global id of this rank's arena 0 (contiguous shards of n_local arenas)."""
    return rank * n_local


def all_gather_returns(local, async_op=False):
    """All-gathers a per-arena tensor [n_local, ...] into [world*n_local, ...] in global arena order.
    256 KiB per rank at 65,536 fp32 returns; latency-bound, issued off the critical path."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return (local.clone(), None) if async_op else local.clone()
    world = dist.get_world_size()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(out, local.contiguous(), async_op=async_op)
    return (out, work) if async_op else out


def reduce_max(value, device):
    """MAX over ranks of a python float (the timing rule of bench.py)."""
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum(value, device):
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
