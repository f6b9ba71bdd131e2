import sys, torch
sys.path.insert(0, '.')
import roborugby_amd as rr
for seed in (0, 1, 2):
    for mode in ("still", "forward", "spin", "random"):
        n = 8
        env = rr.BatchedRoboRugbyEnv(n, preset="G", seed=seed)
        env.reset()
        g = torch.Generator(device='cuda'); g.manual_seed(1)
        ts = []; bits = 0
        for i in range(25):
            if mode == "random": a = torch.randint(0, 8, (n, 4), generator=g, device='cuda', dtype=torch.int32)
            elif mode == "still": a = torch.full((n, 4), 8, device='cuda', dtype=torch.int32)
            elif mode == "forward": a = torch.zeros((n, 4), device='cuda', dtype=torch.int32)
            else: a = torch.full((n, 4), 2, device='cuda', dtype=torch.int32)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); o, r, d, info = env.step(a); e1.record(); torch.cuda.synchronize()
            bits |= int((info.status & 0xFFFF).max())
            if i >= 5: ts.append(e0.elapsed_time(e1))
        ts.sort()
        print("seed", seed, mode, "median ms %.4f min %.4f max %.4f" % (ts[len(ts)//2], ts[0], ts[-1]), "status bits", bits, flush=True)
