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
def act():
    return torch.randint(0, 8, (n, na), generator=g, device='cuda', dtype=torch.int32) if mode == "random" else torch.full((n, na), 8, device='cuda', dtype=torch.int32)
for _ in range(5): env.step(act())
torch.cuda.synchronize(); lib.rr_debug_phase_cycles(buf, 1)
K = 20
for _ in range(K): env.step(act())
torch.cuda.synchronize(); lib.rr_debug_phase_cycles(buf, 0)
names = ["frame_begin", "move_bots", "K1 robot pairs", "push (br detect+resp)", "roll", "resolve loop", "undo", "frame_end",
         "step_begin", "(12 substeps total)", "rewards", "obs+out", "load+derive", "store"]
v = list(buf)[:14]
tot = sum(v[i] for i in (8, 9, 10, 11, 12, 13))
waves = n // (64 // env.lanes_per_env())
print(f"{preset} {mode}: VW={env.lanes_per_env()} cycles(100MHz ticks?) per wave per step: {tot / K / waves:.0f}")
for i, nm in enumerate(names):
    print(f"  {nm:26s} {v[i] / K / waves:10.0f}  {100.0 * v[i] / tot:5.1f}%")
