"""Probe: 65,536 arenas as P independent shard envs (arena_offset) stepped on P HIP streams -- consecutive launches of different
shards overlap, so the chip does not drain at the end of every step.  Random policy (actions do not depend on observations).
usage: python tools/pipeline_probe.py [G|T] [P] [steps]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import roborugby_amd as rr
preset = sys.argv[1] if len(sys.argv) > 1 else "G"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2
K = int(sys.argv[3]) if len(sys.argv) > 3 else 200
N = 65536
n = N // P
dev = torch.device("cuda:0")
envs = [rr.BatchedRoboRugbyEnv(n, preset=preset, device=dev, seed=0, arena_offset=i * n) for i in range(P)]
na = envs[0].preset.nr
g = torch.Generator(device=dev); g.manual_seed(1234)
acts = torch.randint(0, 8, (K + 20, N, na), generator=g, device=dev, dtype=torch.int32)
streams = [torch.cuda.Stream(device=dev) for _ in range(P)]
outs = []
for e in envs:
    e.reset()
    outs.append(None)
torch.cuda.synchronize()
def run(k0, k1):
    for s in range(k0, k1):
        for i, e in enumerate(envs):
            with torch.cuda.stream(streams[i]):
                outs[i] = e.step(acts[s, i * n:(i + 1) * n], out=outs[i][:4] if False else None)
run(0, 20)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(20, 20 + K)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{preset}: {P} shard(s) x {n} arenas on {P} stream(s): {N * K / dt / 1e6:.1f} M env-steps/s ({dt / K * 1e3:.4f} ms per step of all {N} arenas)")
