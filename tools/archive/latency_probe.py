"""Single-wave latency probe: kernel time for tiny batches (one or a few wavefronts) -- the floor of any launch."""
import sys, torch
sys.path.insert(0, '.')
import roborugby_amd as rr
preset = sys.argv[1] if len(sys.argv) > 1 else "G"
for n in (8, 64, 2048, 16384):
    env = rr.BatchedRoboRugbyEnv(n, preset=preset, seed=0, reset_on_fault=True)
    env.reset()
    na = env.preset.nr
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    for mode in ("still", "random"):
        ts = []
        for i in range(25):
            a = torch.randint(0, 8, (n, na), generator=g, device='cuda', dtype=torch.int32) if mode == "random" else torch.full((n, na), 8, device='cuda', dtype=torch.int32)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); env.step(a); e1.record(); torch.cuda.synchronize()
            if i >= 5: ts.append(e0.elapsed_time(e1))
        ts.sort()
        print(preset, "n", n, mode, "median ms %.4f min %.4f" % (ts[len(ts)//2], ts[0]), flush=True)
