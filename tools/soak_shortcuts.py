"""One-off soak: a long chase-policy rollout (several episode boundaries) with the exact shortcuts on and off -- every per-step output
and the final state must be bit-identical.  usage: python tools/soak_shortcuts.py [T|G|D] [arenas] [steps] [exact]
(`exact`: the parity build -- its scratch-rect carry has to survive the shortcuts and the in-kernel resets too)"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import roborugby_amd as rr
preset = sys.argv[1] if len(sys.argv) > 1 else "T"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
exact = len(sys.argv) > 4 and sys.argv[4] == "exact"
res = []
for sw in ({}, {"RR_NO_MEMO": "1", "RR_NO_ORDER": "1"}):
    for k, v in sw.items(): os.environ[k] = v
    env = rr.BatchedRoboRugbyEnv(n, preset=preset, seed=3, exact_trig=exact)
    for k in sw: os.environ.pop(k)
    na = env.preset.nr
    gen = torch.Generator(device="cuda"); gen.manual_seed(7)
    obs = env.reset()
    if preset in ("G", "D"):  # shorten the first episode so that the rollout crosses episode boundaries
        st = env.get_state(); st["step"][:] = env.preset.game_len_steps - 400 - (torch.arange(n, device="cuda") % 300).to(st["step"].dtype)
        env.set_state(st["robots"], st["robots_i"], st["balls"], st["step"])
    h = torch.zeros(4, device="cuda", dtype=torch.float64)
    nd = 0
    for s in range(S):
        d = (obs[:, 1] - obs[:, 0] + 540.0) % 360.0 - 180.0
        a = torch.where(d.abs() < 8, 0, torch.where(d > 0, 2, 3)).to(torch.int32)
        noise = torch.rand(n, generator=gen, device="cuda") < 0.1
        a = torch.where(noise, torch.randint(0, 8, (n,), generator=gen, device="cuda", dtype=torch.int32), a).view(n, 1)
        if na > 1: a = torch.cat([a, torch.randint(0, 8, (n, na - 1), generator=gen, device="cuda", dtype=torch.int32)], 1)
        obs, r, dn, info = env.step(a)
        w = torch.arange(1, n + 1, device="cuda", dtype=torch.float64)
        h += torch.stack([(obs.double().nan_to_num().sum(1) * w).sum(), (r.double() * w).sum(), (dn.double() * w).sum(), ((info.status & 0xFFFFF).double() * w).sum()])
        nd += int(dn.sum())
    st = env.get_state()
    res.append((h.cpu().numpy(), {k: v.cpu().numpy() for k, v in st.items()}, nd))
    print(preset, "parity build" if exact else "default build", "shortcuts", "off" if sw else "on", "episodes finished", nd, "checksums", h.cpu().numpy())
same = np.array_equal(res[0][0], res[1][0]) and all(np.array_equal(res[0][1][k], res[1][1][k], equal_nan=True) for k in res[0][1])
print("IDENTICAL" if same else "MISMATCH")
sys.exit(0 if same else 1)
