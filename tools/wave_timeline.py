"""Diagnostic (-DRR_PROFILE_PHASES build): start/end clock of every wavefront of one k_step launch -> how long waves
live, how many are resident over time, and how much of the launch is the tail of its slowest wavefronts.
usage: RR_LIB_PATH=<diag .so> python tools/wave_timeline.py [G|T] [steps_before] [n_arenas]"""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, '.')
import roborugby_amd as rr
from roborugby_amd import _lib
preset = sys.argv[1] if len(sys.argv) > 1 else "G"
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
env = rr.BatchedRoboRugbyEnv(n, preset=preset, seed=0)
env.reset()
lib = _lib.load()
na = env.preset.nr
g = torch.Generator(device='cuda'); g.manual_seed(1)
waves = n // (64 // env.lanes_per_env())
buf = (C.c_ulonglong * (2 * waves))()
for rep in range(warm + 3):
    a = torch.randint(0, 8, (n, na), generator=g, device='cuda', dtype=torch.int32)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if rep >= warm: before = {k: v.cpu().numpy() for k, v in env.get_state().items()}
    e0.record(); o_, r_, d_, info = env.step(a); e1.record(); torch.cuda.synchronize()
    if rep < warm: continue
    assert lib.rr_debug_wave_times(buf, waves) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(waves, 2).astype(np.int64)
    t0 = t[:, 0].min(); s = (t[:, 0] - t0) / 100.0; e = (t[:, 1] - t0) / 100.0  # s_memrealtime: 100 MHz -> us
    d = e - s
    span = e.max()
    us = e0.elapsed_time(e1) * 1e3
    print(f"{preset} n={n} waves={waves} launch {us:.0f} us; first start -> last end {span:.0f} us")
    q = np.percentile(d, [50, 90, 99, 99.9, 100])
    print("  wave duration us: mean %.1f p50 %.1f p90 %.1f p99 %.1f p99.9 %.1f max %.1f" % ((d.mean(),) + tuple(q)))
    print("  sum(durations)/span = avg resident waves: %.0f" % (d.sum() / span))
    for frac in (0.5, 0.9, 0.99, 0.999):
        print("  time by which %5.1f%% of waves ended: %.2f of span" % (frac * 100, np.percentile(e, frac * 100) / span))
    grid = np.linspace(0, span, 21)
    res = [(int(((s <= x) & (e > x)).sum())) for x in grid]
    print("  resident waves over time:", res)
    apw = 64 // env.lanes_per_env()
    st = info.status.cpu().numpy()
    for w in np.argsort(-d)[:4]:
        ar = np.arange(w * apw, (w + 1) * apw)
        print(f"  slow wave {w}: {d[w]:.0f} us, status bits {[hex(int(x) & 0xffff) for x in st[ar]]} naughty {[int(x) >> 16 for x in st[ar]]} step {before['step'][ar].tolist()}")
        if w == np.argsort(-d)[0]:
            if d[w] > 1000: np.savez(f'gpurun_out/slow_wave_{preset}_{rep}.npz', actions=a.cpu().numpy()[ar], **{k: v[ar] for k, v in before.items()})
            for a_ in ar:
                rb = before['robots'][a_]; bl = before['balls'][a_]
                dmin = min(float(np.hypot(bl[b, 0] - rb[r, 0], bl[b, 1] - rb[r, 1])) for b in range(bl.shape[0]) for r in range(rb.shape[0]))
                wall = float(min(bl[:, 0].min(), bl[:, 1].min(), (env.preset.arena_w - bl[:, 0]).min(), (env.preset.arena_h - bl[:, 1]).min()))
                rr_ = min([float(np.hypot(rb[i, 0] - rb[j, 0], rb[i, 1] - rb[j, 1])) for i in range(rb.shape[0]) for j in range(i)] or [float('inf')])
                bb_ = min([float(np.hypot(bl[i, 0] - bl[j, 0], bl[i, 1] - bl[j, 1])) for i in range(bl.shape[0]) for j in range(i)] or [float('inf')])
                rw_ = float(min(rb[:, 2].min(), rb[:, 4].min(), (env.preset.arena_w - rb[:, 3]).min(), (env.preset.arena_h - rb[:, 5]).min()))
                print(f"    arena {a_}: min ball-robot {dmin:.1f} robot-robot {rr_:.1f} ball-ball {bb_:.1f} ball-wall {wall:.1f} robot-edge-wall {rw_:.2f} |v|max {np.abs(bl[:, 6:8]).max():.3f} actions {a[a_].tolist()}")
