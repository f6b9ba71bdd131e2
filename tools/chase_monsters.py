"""Diagnostic (-DRR_PROFILE_PHASES build, RR_NO_ORDER=1): roll a G (or T) batch under the chase policy and, for every step after
the warm-up, keep the pre-step state + actions of the arenas of the SLOWEST wavefront together with the distribution of the
wavefront run times -- what bounds a contact-rich launch.  Output: gpurun_out/chase_monsters_<preset>.npz
usage: RR_NO_ORDER=1 RR_LIB_PATH=<diag .so> python tools/chase_monsters.py [G|T] [steps] [warmup] [chase|random]"""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import roborugby_amd as rr
from roborugby_amd import _lib
preset = sys.argv[1] if len(sys.argv) > 1 else "G"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 150
policy = sys.argv[4] if len(sys.argv) > 4 else "chase"
budget = int(sys.argv[5]) if len(sys.argv) > 5 else 0  # the budgeted step (shader clocks), 0 = synchronous
n = 65536
env = rr.BatchedRoboRugbyEnv(n, preset=preset, seed=0, step_budget_clocks=budget)
obs = env.reset()
if os.environ.get("RR_STAGGER"):  # steady state of a long rollout: random episode phases + one whole episode of pre-roll (like bench.py)
    st = env.get_state()
    T = env.preset.game_len_steps
    gg = torch.Generator(device='cuda'); gg.manual_seed(7)
    env.set_state(st["robots"], st["robots_i"], st["balls"], torch.randint(0, T, (n,), generator=gg, device='cuda', dtype=torch.int32))
    for i in range(T + 1):
        obs = env.step(torch.randint(0, 8, (n, env.preset.nr), generator=gg, device='cuda', dtype=torch.int32))[0]
lib = _lib.load()
na = env.preset.nr
apw = 64 // env.lanes_per_env()
waves = n // apw
buf = (C.c_ulonglong * (2 * waves))()
g = torch.Generator(device='cuda'); g.manual_seed(1)
def act(o):
    if policy == "random":
        return torch.randint(0, 8, (n, na), generator=g, device='cuda', dtype=torch.int32)
    d = (o[:, 1] - o[:, 0] + 540.0) % 360.0 - 180.0
    a0 = torch.where(d.abs() < 8, 0, torch.where(d > 0, 2, 3)).to(torch.int32)
    r = torch.randint(0, 8, a0.shape, generator=g, device='cuda', dtype=torch.int32)
    m = torch.rand(a0.shape, generator=g, device='cuda') < 0.1
    a0 = torch.where(m, r, a0).view(n, 1)
    return torch.cat([a0, torch.randint(0, 8, (n, na - 1), generator=g, device='cuda', dtype=torch.int32)], 1) if na > 1 else a0
found, pct = [], []
for s in range(warm + steps):
    a = act(obs)
    before = env.get_state() if s >= warm else None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); res = env.step(a); obs = res[0]; e1.record(); torch.cuda.synchronize()
    stw = res[3].status  # diagnostic build: contact work of the step in bits 20-29, "began frozen" in bit 30
    if before is None: continue
    assert lib.rr_debug_wave_times(buf, waves) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(waves, 2).astype(np.int64)
    d = (t[:, 1] - t[:, 0]) / 100.0  # us (s_memtime ticks at 100 MHz)
    span = (t[:, 1].max() - t[:, 0].min()) / 100.0
    q = np.percentile(d, [50, 90, 99, 99.9, 100])
    pct.append(np.concatenate([q, [span, e0.elapsed_time(e1) * 1e3, d.sum() / 2048.0]]))
    w = int(np.argmax(d))
    ar = np.arange(w * apw, w * apw + apw)
    found.append(dict(step=s, wave=w, us=float(d[w]), actions=a[ar].cpu().numpy(), work=((stw[ar] >> 20) & 1023).cpu().numpy(),
                      frozen=((stw[ar] >> 30) & 1).cpu().numpy(),
                      robots=before['robots'][ar].cpu().numpy(), robots_i=before['robots_i'][ar].cpu().numpy(),
                      balls=before['balls'][ar].cpu().numpy(), stepc=before['step'][ar].cpu().numpy()))
    if s == warm + steps - 1:  # the last step: every wavefront above half the slowest one's time
        slow = np.nonzero(d > 0.5 * d.max())[0]
        ars = (slow[:, None] * apw + np.arange(apw)[None]).reshape(-1)
        slow_rec = dict(slow_waves=slow, slow_us=d[slow], slow_actions=a[ars].cpu().numpy(), slow_robots=before['robots'][ars].cpu().numpy(),
                        slow_robots_i=before['robots_i'][ars].cpu().numpy(), slow_balls=before['balls'][ars].cpu().numpy(),
                        slow_stepc=before['step'][ars].cpu().numpy(), all_us=d)
pct = np.array(pct)
print(f"{preset} {policy}: wavefront run time us  p50 {pct[:,0].mean():.0f}  p90 {pct[:,1].mean():.0f}  p99 {pct[:,2].mean():.0f}  p99.9 {pct[:,3].mean():.0f}  "
      f"max {pct[:,4].mean():.0f}   launch span {pct[:,5].mean():.0f}  event time {pct[:,6].mean():.0f}  "
      f"sum of wave times / 2048 slots {pct[:,7].mean():.0f}")
print("slowest wave per step (step, wave, us):", [(f['step'], f['wave'], int(f['us'])) for f in found])
for f in found[::4]:
    print("  step", f['step'], "wave", f['wave'], int(f['us']), "us: work", f['work'].tolist(), "began frozen", f['frozen'].tolist())
# how the wavefront time relates to the contact work of its arenas (last step): mean duration by the wavefront's total / max work
w_all = ((stw >> 20) & 1023).cpu().numpy().reshape(waves, apw)
tot, mx = w_all.sum(1), w_all.max(1)
for lo, hi in ((0, 0), (1, 3), (4, 11), (12, 47), (48, 10 ** 6)):
    m = (mx >= lo) & (mx <= hi)
    if m.any():
        print(f"  wavefronts whose busiest arena did {lo}..{hi} work units: {int(m.sum())} ({100.0 * m.mean():.1f} %), mean {d[m].mean():.1f} us, p90 {np.percentile(d[m], 90):.1f} us")
print("  arenas with work > 0: %.2f %%; wavefronts with any: %.1f %%" % (100.0 * (w_all > 0).mean(), 100.0 * (mx > 0).mean()))
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
np.savez(os.path.join(ROOT, 'gpurun_out', f'chase_monsters_{preset}.npz' if policy == 'chase' else f'{policy}_monsters_{preset}.npz'), pct=pct, **slow_rec,
         **{f"{k}_{i}": v for i, f in enumerate(found) for k, v in f.items()})
