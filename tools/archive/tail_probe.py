import sys, time, torch
sys.path.insert(0, '.')
import roborugby_amd as rr
def run(preset, n, mode, steps=30):
    env = rr.BatchedRoboRugbyEnv(n, preset=preset, seed=0)
    env.reset()
    na = env.preset.nr
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    out = None
    ts = []
    for i in range(steps + 5):
        if mode == "random": a = torch.randint(0, 8, (n, na), generator=g, device='cuda', dtype=torch.int32)
        elif mode == "still": a = torch.full((n, na), 8, device='cuda', dtype=torch.int32)
        else: a = torch.zeros((n, na), device='cuda', dtype=torch.int32)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); o, r, d, info = env.step(a); e1.record(); torch.cuda.synchronize()
        if i >= 5: ts.append(e0.elapsed_time(e1))
    st = info.status
    print(preset, n, mode, "ms mean %.3f min %.3f max %.3f" % (sum(ts)/len(ts), min(ts), max(ts)), "faulted arenas", int(((st & 8) != 0).sum()), flush=True)
for n in (16384, 65536, 262144):
    for mode in ("still", "random"):
        run("G", n, mode)
run("T", 65536, "still"); run("T", 65536, "random")
