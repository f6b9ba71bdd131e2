"""N>1 path on CPU: world_size-2 gloo processes exercise the sharding rule and the logging-side all-gather
(roborugby_amd/dist.py) -- the same code bench.py runs over RCCL on the 8-GPU node."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_local, q):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from roborugby_amd import dist as rrd
    import emu_lib as el
    r, lr, w = rrd.init_process_group(backend="gloo")
    assert (r, w) == (rank, world)
    off = rrd.shard_offset(rank, n_local)
    # each rank places its shard with the GLOBAL arena id, steps it, and reports per-arena returns
    rets = torch.zeros(n_local, dtype=torch.float32)
    first = np.zeros((n_local, 11))
    for a in range(n_local):
        e = el.EmuEnv("T", seed=5)
        e.reset(off + a, 0)
        first[a] = e.observe(1)
        tot = 0.0
        for s in range(3):
            tot += e.step([(off + a + s) % 8])["reward"]
        rets[a] = tot
    gathered = rrd.all_gather_returns(rets)
    out, work = rrd.all_gather_returns(rets, async_op=True)
    if work is not None:
        work.wait()
    assert torch.equal(out, gathered)
    mx = rrd.reduce_max(float(rank + 1), torch.device("cpu"))
    sm = rrd.reduce_sum(float(rank + 1), torch.device("cpu"))
    rrd.barrier()
    q.put((rank, gathered.numpy(), first, mx, sm))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_shards_equal_one_big_batch():
    world, n_local = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_local, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # every rank holds the same gathered vector, in global arena order
    assert np.array_equal(res[0][1], res[1][1]) and res[0][1].shape == (world * n_local,)
    assert res[0][3] == 2.0 and res[0][4] == 3.0
    # shard invariance: the same six arenas stepped in ONE process give the same returns / first observations
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import emu_lib as el
    for a in range(world * n_local):
        e = el.EmuEnv("T", seed=5)
        e.reset(a, 0)
        o = e.observe(1)
        tot = sum(e.step([(a + s) % 8])["reward"] for s in range(3))
        assert np.float32(tot) == res[0][1][a]
        assert np.array_equal(o, res[a // n_local][2][a % n_local])


def test_single_process_helpers_are_noops():
    from roborugby_amd import dist as rrd
    t = torch.arange(4, dtype=torch.float32)
    assert torch.equal(rrd.all_gather_returns(t), t)
    assert rrd.reduce_max(3.5, torch.device("cpu")) == 3.5
    assert rrd.shard_offset(3, 65536) == 196608
    rrd.barrier()
