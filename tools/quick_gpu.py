import time, torch, sys
sys.path.insert(0, '.')
import roborugby_amd as rr
for preset, dtype in (("T","f64"),("G","f64"),("T","f32"),("G","f32")):
    n = 65536
    env = rr.BatchedRoboRugbyEnv(n, preset=preset, dtype=dtype, seed=1)
    env.reset()
    na = env.preset.nr
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    acts = torch.randint(0, 8, (n, na), generator=g, device='cuda', dtype=torch.int32)
    for _ in range(3): env.step(acts)
    torch.cuda.synchronize()
    t = time.time(); K = 20
    for _ in range(K): env.step(acts)
    torch.cuda.synchronize()
    dt = (time.time()-t)/K
    print(preset, dtype, "ms/step", dt*1e3, "steps/s", n/dt, flush=True)
