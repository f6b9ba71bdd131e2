"""A single-rank RCCL process group on the one GPU there is (tests/test_gpu_dist.py starts it as a fresh process): the calls
bench.py makes at N > 1 -- device bound to the communicator before the group exists, all-gather of returns (sync + async),
max / sum reductions, barrier with device_ids -- through the nccl (= RCCL) backend itself rather than its gloo stand-in."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    from roborugby_amd import dist as rrd
    rank, dev, world = rrd.init_process_group()
    assert (rank, dev, world) == (0, 0, 1) and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
    import roborugby_amd as rr
    env = rr.BatchedRoboRugbyEnv(4096, preset="T", device="cuda:0", seed=1, arena_offset=rrd.shard_offset(rank, 4096))
    env.reset()
    a = torch.zeros(4096, 1, dtype=torch.int32, device="cuda:0")
    for _ in range(301):
        env.step(a)
    lr = env.episode_stats()[0]
    g1 = rrd.all_gather_returns(lr)
    g2, work = rrd.all_gather_returns(lr, async_op=True)
    if work is not None:
        work.wait()
    torch.cuda.synchronize()
    assert torch.equal(g1, lr) and torch.equal(g2, lr)
    assert rrd.reduce_max(2.5, torch.device("cuda:0")) == 2.5 and rrd.reduce_sum(1.5, torch.device("cuda:0")) == 1.5
    rrd.barrier()
    env.close()
    torch.distributed.destroy_process_group()
    print("RCCL single-rank group OK", flush=True)


if __name__ == "__main__":
    main()
