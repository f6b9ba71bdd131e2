"""Actor groups: one batch of arenas as P independent shard envs stepped on P HIP streams.

A `k_step` launch ends with its slowest wavefront, so stepping one big batch drains the chip at the end of every step
(G, 65,536 arenas: ~280 us of work in a ~340 us launch).  Arenas never interact, and a policy maps each arena's observation
to that arena's action -- so the batch can be cut into shards (`arena_offset`: the mechanism that spreads it over GPUs,
every arena's trajectory is the single batch's, bit for bit) whose step s+1 only waits for their OWN step s and their own
policy call.  With each shard on its own stream one shard's launch fills the chip while another's drains: +17 % at P = 2
(more shards cost more host launches than they gain).

This is the caller-side pattern `bench.py --pipeline P` measures; the reference's one-call-per-step gym surface cannot
express it (all observations are due before the next actions exist), which is why it is a helper and not the default."""
import torch

from .env import BatchedRoboRugbyEnv


class ShardedPipeline:
    def __init__(self, num_envs, shards=2, device=None, arena_offset=0, **env_kwargs):
        if num_envs % shards:
            raise ValueError(f"{num_envs} arenas do not split into {shards} equal shards")
        self.num_envs, self.shards, self.n = num_envs, shards, num_envs // shards
        self.device = torch.device(device if device is not None else "cuda:0")
        self.envs = [BatchedRoboRugbyEnv(self.n, device=self.device, arena_offset=arena_offset + i * self.n, **env_kwargs)
                     for i in range(shards)]
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(shards)]
        self.obs = [None] * shards
        self.last = [None] * shards

    @property
    def preset(self):
        return self.envs[0].preset

    def reset(self):
        """every shard's env.reset() on its stream; returns the list of per-shard observation tensors (each valid on its stream)."""
        cur = torch.cuda.current_stream(self.device)
        for i, e in enumerate(self.envs):
            self.streams[i].wait_stream(cur)
            with torch.cuda.stream(self.streams[i]):
                self.obs[i] = e.reset()
        return self.obs

    def run(self, policy, steps, on_step=None, outs=None):
        """`steps` env steps of every shard.  policy(shard_index, obs[n,11]) -> actions for that shard, called with the shard's
        stream current (everything it enqueues runs there); on_step(shard_index, step, obs, reward, done, info) likewise.
        `outs`: per-shard preallocated output tuples, or a callable (shard_index, step) -> tuple (e.g. a fresh status row per step).
        No host synchronisation happens here: call synchronize() (or wait on the streams) before reading results elsewhere."""
        for s in range(steps):
            for i, e in enumerate(self.envs):
                with torch.cuda.stream(self.streams[i]):
                    a = policy(i, self.obs[i])
                    res = e.step(a, out=(outs(i, s) if callable(outs) else outs[i]) if outs is not None else None)
                    self.obs[i], self.last[i] = res[0], res
                    if on_step is not None:
                        on_step(i, s, *res)
        return self.last

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def gather(self, per_shard):
        """concatenate per-shard tensors into batch order (after synchronize())."""
        return torch.cat(list(per_shard), 0)

    def close(self):
        for e in self.envs:
            e.close()
