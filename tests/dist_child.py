"""One rank of the two-process rehearsal of the N > 1 path on a one-GPU box (tests/test_gpu_dist.py starts it twice, as
fresh processes, before anything in them has touched the GPU).  RANK / WORLD_SIZE / MASTER_* come from the parent;
RR_SHARE_GPU=1 puts every rank on cuda:0 and RR_DIST_BACKEND=gloo carries the logging all-gather (RCCL refuses two ranks
on one device).  Each rank loads libroborugby_amd.so itself, owns arenas [rank*n, (rank+1)*n) through arena_offset, steps
them with the slice of a shared action table, and all-gathers the finished-episode returns."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def action_table(steps, n_total, na, seed=4321):
    import torch
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 8, (steps, n_total, na), generator=g, dtype=torch.int32)


def run_shard(n, offset, steps, n_total, preset="T", seed=31):
    """Steps arenas [offset, offset + n) of a batch of n_total; returns (last finished returns [n], final obs [n,11], loaded .so paths)."""
    import torch
    import roborugby_amd as rr
    env = rr.BatchedRoboRugbyEnv(n, preset=preset, device="cuda:0", seed=seed, arena_offset=offset)
    env.reset()
    acts = action_table(steps, n_total, env.preset.nr)[:, offset:offset + n].cuda()
    for s in range(steps):
        obs, rew, done, info = env.step(acts[s])
    lr, _, ll, cnt = env.episode_stats()
    assert int(cnt.min()) >= 1  # every arena finished an episode
    return lr, obs, env


def main():
    out_dir, n, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    import torch
    from roborugby_amd import dist as rrd
    rank, dev, world = rrd.init_process_group()
    assert dev == 0 and torch.cuda.current_device() == 0
    lr, obs, env = run_shard(n, rrd.shard_offset(rank, n), steps, world * n)
    gathered, work = rrd.all_gather_returns(lr, async_op=True)
    if work is not None:
        work.wait()
    obs_all = rrd.all_gather_returns(obs)
    worst = rrd.reduce_max(float(rank), torch.device("cuda:0"))
    rrd.barrier()
    maps = [ln.split()[-1] for ln in open("/proc/self/maps") if "libroborugby_amd" in ln]
    torch.save(dict(rank=rank, world=world, returns=gathered.cpu(), obs=obs_all.cpu(), reduce_max=worst, lib=sorted(set(maps)),
                    backend=torch.distributed.get_backend()), os.path.join(out_dir, f"rank{rank}.pt"))
    env.close()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
