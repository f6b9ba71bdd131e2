"""Diagnostic (-DRR_PROFILE_PHASES build): step a G batch and save the pre-step state of every arena group whose
wavefront ran longer than a threshold (gpurun_out/monsters_G.npz) -- material for analysing the slow path on CPU."""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import roborugby_amd as rr
from roborugby_amd import _lib
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
thr_us = float(sys.argv[2]) if len(sys.argv) > 2 else 450.0
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 20  # steps to run before collecting
n = 65536
env = rr.BatchedRoboRugbyEnv(n, preset="G", seed=0)
env.reset()
lib = _lib.load()
g = torch.Generator(device='cuda'); g.manual_seed(1)
waves = n // 8
buf = (C.c_ulonglong * (2 * waves))()
order_cap = None
found = []
for s in range(skip + steps):
    a = torch.randint(0, 8, (n, 4), generator=g, device='cuda', dtype=torch.int32)
    before = env.get_state() if s >= skip else None
    env.step(a); torch.cuda.synchronize()
    if before is None: continue
    assert lib.rr_debug_wave_times(buf, waves) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(waves, 2).astype(np.int64)
    d = (t[:, 1] - t[:, 0]) / 100.0
    # NOTE: wave times are indexed by dispatch slot; with slowest-first dispatch the slot -> group map is the order array,
    # so run this tool with RR_NO_ORDER=1
    for w in np.nonzero(d > thr_us)[0]:
        ar = np.arange(w * 8, w * 8 + 8)
        found.append(dict(step=s, wave=int(w), us=float(d[w]), actions=a[ar].cpu().numpy(),
                          robots=before['robots'][ar].cpu().numpy(), robots_i=before['robots_i'][ar].cpu().numpy(),
                          balls=before['balls'][ar].cpu().numpy(), stepc=before['step'][ar].cpu().numpy()))
print("monster groups found:", len(found), [(f['step'], f['wave'], int(f['us'])) for f in found][:60])
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
np.savez(os.path.join(ROOT, 'gpurun_out', 'monsters_G.npz'), **{f"{k}_{i}": v for i, f in enumerate(found[:80]) for k, v in f.items()})
