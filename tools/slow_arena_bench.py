"""Latency of ONE pathological arena (tools/fixtures/squeezed_G.npz: a ball squeezed between two robots -- every
sub-step runs the push, all 10 resolve passes and the undo loop) -- the floor under any launch that contains it.
usage: [RR_LIB_PATH=...] [RR_VW=..] python tools/slow_arena_bench.py [n_copies]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import roborugby_amd as rr
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
d = np.load(os.path.join(ROOT, 'tools/fixtures/squeezed_G.npz'))
env = rr.BatchedRoboRugbyEnv(n, preset="G", seed=0, auto_reset=False, time_limit=False)
env.reset()
rep = lambda k: np.repeat(d[k][None], n, 0)
acts = torch.as_tensor(rep('actions'), dtype=torch.int32, device='cuda')
ts = []
for i in range(12):
    env.set_state(rep('robots'), rep('robots_i'), rep('balls'), np.full(n, int(d['step']), np.int32))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); o, r, dn, info = env.step(acts); e1.record(); torch.cuda.synchronize()
    if i >= 2: ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print(f"squeezed arena x{n}, VW={env.lanes_per_env()}: step latency median {ts[len(ts)//2]:.0f} us (min {ts[0]:.0f}), status {int(info.status[0]) & 0xffff:#x}")
