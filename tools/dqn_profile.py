"""Where a batched DQN step spends its time (config 5): stage timings with CUDA events."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import roborugby_amd as rr
from roborugby_amd.dqn import BatchedDQNAgent
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = rr.make("RoboRugbySimpleDuel-v3", num_envs=n, preset="T", device="cuda:0")
agent = BatchedDQNAgent(batch_size=min(32768, max(2500, n // 2)), device="cuda:0", max_mem_size=max(500000, 32 * n))  # train()'s defaults
obs = env.reset()
acc = {}
def timed(name, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
    return r
for i in range(140):
    if i == 40: acc.clear()
    a = timed("choose_action", lambda: agent.choose_action(obs))
    o2, r, d, info = timed("env.step", lambda: env.step(a.view(-1, 1)))
    real = (info.status & 1024) == 0
    timed("store", lambda: agent.store_transition(obs, a, r, o2, d, valid=real))
    timed("learn x4", lambda: [agent.learn() for _ in range(4)])
    obs = o2
for k, v in acc.items(): print(f"{k:14s} {v / 100 * 1e3:8.3f} ms/step")
