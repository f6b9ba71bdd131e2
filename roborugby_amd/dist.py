"""Multi-GPU plumbing: one process per GPU, arenas sharded with NO data-path collective.

Arenas are independent (the reference is single-arena), so rank r simply owns global arenas
[r*n_local, (r+1)*n_local): the counter-based reset RNG is keyed by the GLOBAL arena id (rr_config.arena_offset),
which makes every arena's trajectory invariant to how the batch is sharded.  The only exchange is a logging-side
all-gather of finished-episode returns (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests and in the
one-GPU rehearsal of the N > 1 path)."""
import os

import torch
import torch.distributed as dist


def dist_env():
    """(rank, local_rank, world_size) from the launcher's environment (torch.distributed.run)."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)))


def local_device_index(local_rank):
    """HIP device of this rank: its local rank -- or 0 for every rank when RR_SHARE_GPU (alias RR_BENCH_SHARE_GPU) asks
    for the rehearsal of the N > 1 path on a one-GPU box (then with RR_DIST_BACKEND=gloo: RCCL refuses two ranks on one
    device)."""
    if os.environ.get("RR_SHARE_GPU") or os.environ.get("RR_BENCH_SHARE_GPU"):
        return 0
    return local_rank


def _backend():
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else None


def backend_name():
    """'nccl' (= RCCL on ROCm), 'gloo', or None without a process group -- what bench.py records next to its collective count."""
    return _backend()


def init_process_group(backend=None):
    """Returns (rank, device_index, world).  The device is selected -- and, for RCCL, bound to the process group
    (device_id: eager communicator, no "guessing device ID" at the first barrier) -- BEFORE the group exists."""
    rank, local_rank, world = dist_env()
    dev = local_device_index(local_rank)
    # the backend is resolved FIRST: only RCCL needs a device bound per rank.  A gloo group (CPU tests, the one-GPU
    # rehearsal) binds a device only when that ordinal exists -- two gloo ranks with LOCAL_RANK 0/1 on a one-GPU box must not
    # die with "invalid device ordinal", and a CPU-only gloo run must not create a GPU context on every rank.
    backend = backend or os.environ.get("RR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(dev)
    elif (os.environ.get("RR_SHARE_GPU") or os.environ.get("RR_BENCH_SHARE_GPU")) and torch.cuda.is_available():
        torch.cuda.set_device(dev)  # the rehearsal: every rank steps its shard on cuda:0
    # (any other gloo group: no device is touched here; callers address `cuda:<dev>` explicitly if they use the GPU)
    # (RR_DIST_FORCE_INIT=1 builds the group even for a single rank: lets a one-GPU box exercise the RCCL code path itself)
    if (world > 1 or os.environ.get("RR_DIST_FORCE_INIT")) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", dev)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, dev, world


def shard_offset(rank, n_local):
    """Global id of this rank's arena 0 (contiguous shards of n_local arenas)."""
    return rank * n_local


def _comm_tensor(t):
    """gloo has no device collectives for all_gather: stage through the host (rehearsal / CPU tests only)."""
    return t.cpu() if (_backend() == "gloo" and t.is_cuda) else t


def all_gather_returns(local, async_op=False):
    """All-gathers a per-arena tensor [n_local, ...] into [world*n_local, ...] in global arena order.
    256 KiB per rank at 65,536 fp32 returns; latency-bound, issued off the critical path."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not os.environ.get("RR_DIST_FORCE_INIT")):
        return (local.clone(), None) if async_op else local.clone()
    world = dist.get_world_size()
    src = _comm_tensor(local.contiguous())
    out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    work = dist.all_gather_into_tensor(out, src, async_op=async_op)
    if src.device != local.device:  # host-staged: hand back a device tensor like the RCCL path does
        if async_op:
            work.wait()
            work = None
        out = out.to(local.device)
    return (out, work) if async_op else out


def _reduce(value, device, op):
    dev = torch.device("cpu") if _backend() == "gloo" else device
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("RR_DIST_FORCE_INIT")):
        dist.all_reduce(t, op=op)
    return float(t.item())


def reduce_max(value, device):
    """MAX over ranks of a python float (the timing rule of bench.py)."""
    return _reduce(value, device, dist.ReduceOp.MAX)


def reduce_sum(value, device):
    return _reduce(value, device, dist.ReduceOp.SUM)


def barrier():
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("RR_DIST_FORCE_INIT")):
        if _backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()
