"""Probe: the 65,536 arenas of one step as TWO shard launches on two HIP streams that are JOINED after every step (what an rr_step that
forks internally could do without changing its one-call-per-step semantics) against the single launch and against the un-joined
pipeline.  usage: python tools/split_launch_probe.py [G|T] [steps]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import roborugby_amd as rr
preset = sys.argv[1] if len(sys.argv) > 1 else "G"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n = 65536
dev = torch.device("cuda:0")
def bench(parts, join):
    envs = [rr.BatchedRoboRugbyEnv(n // parts, preset=preset, device=dev, seed=0, arena_offset=i * (n // parts)) for i in range(parts)]
    na = envs[0].preset.nr
    g = torch.Generator(device=dev); g.manual_seed(1)
    acts = [torch.randint(0, 8, (K + 20, n // parts, na), generator=g, device=dev, dtype=torch.int32) for _ in range(parts)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    for e in envs: e.reset()
    torch.cuda.synchronize()
    main = torch.cuda.current_stream(dev)
    def step(s):
        if parts == 1:
            envs[0].step(acts[0][s]); return
        if join:
            ev = torch.cuda.Event(); ev.record(main)
        for i, e in enumerate(envs):
            with torch.cuda.stream(streams[i]):
                if join: streams[i].wait_event(ev)
                e.step(acts[i][s])
        if join:
            for st in streams:
                d = torch.cuda.Event(); d.record(st); main.wait_event(d)
    for s in range(20): step(s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(20, 20 + K): step(s)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    for e in envs: e.close()
    return n * K / dt / 1e6
print(preset, "single launch            %.1f M env-steps/s" % bench(1, False))
print(preset, "two launches, joined     %.1f M" % bench(2, True))
print(preset, "four launches, joined    %.1f M" % bench(4, True))
print(preset, "two launches, pipelined  %.1f M" % bench(2, False))
