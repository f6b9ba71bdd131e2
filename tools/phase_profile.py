"""Diagnostic: per-phase cycle shares of k_step from a -DRR_PROFILE_PHASES build (never the product library).
usage: RR_LIB_PATH=<diag .so> python tools/phase_profile.py [G|T] [random|still]"""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
import roborugby_amd as rr
from roborugby_amd import _lib
preset = sys.argv[1] if len(sys.argv) > 1 else "G"
mode = sys.argv[2] if len(sys.argv) > 2 else "random"
n = 65536
env = rr.BatchedRoboRugbyEnv(n, preset=preset, seed=0)
env.reset()
lib = _lib.load()
buf = (C.c_ulonglong * 32)()
na = env.preset.nr
g = torch.Generator(device='cuda'); g.manual_seed(1)
last = [env.get_game_state()]
def act():
    if mode == "random":
        return torch.randint(0, 8, (n, na), generator=g, device='cuda', dtype=torch.int32)
    if mode == "chase":
        o = last[0]
        d = (o[:, 1] - o[:, 0] + 540.0) % 360.0 - 180.0
        a0 = torch.where(d.abs() < 8, 0, torch.where(d > 0, 2, 3)).to(torch.int32).view(n, 1)
        return torch.cat([a0, torch.randint(0, 8, (n, na - 1), generator=g, device='cuda', dtype=torch.int32)], 1) if na > 1 else a0
    return torch.full((n, na), 8, device='cuda', dtype=torch.int32)
def step():
    last[0] = env.step(act())[0]
for _ in range(150 if mode == "chase" else 5): step()
torch.cuda.synchronize(); lib.rr_debug_phase_cycles(buf, 1)
K = 20
for _ in range(K): step()
torch.cuda.synchronize(); lib.rr_debug_phase_cycles(buf, 0)
names = ["(unused)", "hooks + moves + broad", "resolve_bot (if close)", "push (if close)", "roll + broad", "resolve loop (if close)", "undo (if failed)", "(unused)",
         "step_begin", "(12 substeps total)", "rewards", "obs+out", "load+derive", "store",
         "  resolve: ball-ball detect + bounces", "  resolve: ball-robot detect", "  resolve: bounce_ball_off_bot", "  resolve: wall detect + bounce",
         "  undo: detections", "  undo: undo lanes", "  push: apply_force_to_ball", "  push: bounce_ball_off_bot", "  push: ball-robot detect (no cache)"]
v = list(buf)[:23]
tot = sum(v[i] for i in (8, 9, 10, 11, 12, 13))
waves = (n // (64 // env.lanes_per_env()) + 63) // 64  # every 64th wavefront is stamped
print(f"{preset} {mode}: VW={env.lanes_per_env()} s_memtime ticks per wave per step: {tot / K / waves:.0f}")
for i, nm in enumerate(names):
    print(f"  {nm:26s} {v[i] / K / waves:10.0f}  {100.0 * v[i] / tot:5.1f}%")
