"""Steps ONE stuck arena of tests/data/stuck_chase_{T,G}.npz alone, K times from the same state (product library; run with RR_NO_MEMO=1
so that no exact shortcut fires: the never-repeating case).  Made to sit under rocprofv3 --pmc (tools/r04_stuck_sq.sh): every k_step
dispatch of this process is one wavefront with one working arena.  usage: python3 tools/stuck_arena_step.py [T|G] [index] [K]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import roborugby_amd as rr
preset = sys.argv[1] if len(sys.argv) > 1 else "G"
idx = int(sys.argv[2]) if len(sys.argv) > 2 else 0
K = int(sys.argv[3]) if len(sys.argv) > 3 else 12
d = np.load(os.path.join(ROOT, "tests", "data", f"stuck_chase_{preset}.npz"))
env = rr.BatchedRoboRugbyEnv(1, preset=preset, seed=0, auto_reset=False, time_limit=False)
env.reset()
acts = torch.as_tensor(d["actions"][idx][None], dtype=torch.int32, device="cuda")
ts = []
for i in range(K):
    env.set_state(d["robots"][idx][None], d["robots_i"][idx][None], d["balls"][idx][None], np.array([int(d["step"][idx])], np.int32))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); env.step(acts); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print(f"{preset} stuck arena #{idx}: step latency median {ts[len(ts) // 2]:.0f} us (min {ts[0]:.0f}) over {K} steps")
