"""Diagnostic (-DRR_PROFILE_PHASES build, RR_NO_MEMO=1): latency and phase split of ONE stuck arena of tests/data/stuck_chase_{T,G}.npz
stepped alone -- every sub-step runs the push, the ten resolve passes and the undo loop; lane 0 of block 0 IS that arena, so the
in-kernel stamps are exact here.  usage: RR_NO_MEMO=1 RR_LIB_PATH=<diag .so> python tools/stuck_arena_phases.py [T|G] [index]"""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import roborugby_amd as rr
from roborugby_amd import _lib
preset = sys.argv[1] if len(sys.argv) > 1 else "T"
idx = int(sys.argv[2]) if len(sys.argv) > 2 else 0
d = np.load(os.path.join(ROOT, "tests", "data", f"stuck_chase_{preset}.npz"))
env = rr.BatchedRoboRugbyEnv(1, preset=preset, seed=0, auto_reset=False, time_limit=False)
env.reset()
lib = _lib.load()
buf = (C.c_ulonglong * 32)()
acts = torch.as_tensor(d["actions"][idx][None], dtype=torch.int32, device="cuda")
ts = []
K = 10
for i in range(K + 2):
    env.set_state(d["robots"][idx][None], d["robots_i"][idx][None], d["balls"][idx][None], np.array([int(d["step"][idx])], np.int32))
    torch.cuda.synchronize()
    if i == 2: lib.rr_debug_phase_cycles(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); o, r, dn, info = env.step(acts); e1.record(); torch.cuda.synchronize()
    if i >= 2: ts.append(e0.elapsed_time(e1) * 1e3)
lib.rr_debug_phase_cycles(buf, 0)
ts.sort()
print(f"{preset} stuck arena #{idx}: step latency median {ts[len(ts)//2]:.0f} us (min {ts[0]:.0f}); contact work {(int(info.status[0]) >> 20) & 1023}")
names = ["(unused)", "hooks + moves + broad", "resolve_bot (if close)", "push (if close)", "roll + broad", "resolve loop (if close)", "undo (if failed)", "(unused)",
         "step_begin", "(12 substeps total)", "rewards", "obs+out", "load+derive", "store",
         "  resolve: ball-ball detect + bounces", "  resolve: ball-robot detect", "  resolve: bounce_ball_off_bot", "  resolve: wall detect + bounce",
         "  undo: detections", "  undo: undo lanes", "  push: apply_force_to_ball", "  push: bounce_ball_off_bot", "  push: ball-robot detect (no cache)",
         "    bounce_ball_off_bot: surface / corner search", "    bounce_ball_off_bot: prior-frame pose", "    bounce_ball_off_bot: response + write",
         "    detect (cached): broad phase + ballot", "    detect (cached): mask gather + cache checks", "    detect (cached): narrow-phase round(s)"]
v = list(buf)[:29]
tot = sum(v[i] for i in (8, 9, 10, 11, 12, 13))
for i, nm in enumerate(names):
    if v[i]: print(f"  {nm:40s} {v[i] / K:10.0f} ticks/step {100.0 * v[i] / tot:5.1f}%")
