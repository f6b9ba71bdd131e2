"""Soak of the exact shortcuts under the RANDOM policy at the steady state (where stuck islands with bystander balls live: resting
neighbours, witness tests): arenas at random episode phases, one episode of pre-roll with the shortcuts ON in both runs' common prefix is not
possible (the switches are read at creation), so both envs run the same pre-roll + S compared steps; every per-step output checksum and the
final state must be identical.  usage: python tools/soak_random.py [G|T|D] [arenas] [preroll steps] [compared steps] [exact]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import roborugby_amd as rr
preset = sys.argv[1] if len(sys.argv) > 1 else "G"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
pre = int(sys.argv[3]) if len(sys.argv) > 3 else 4600
S = int(sys.argv[4]) if len(sys.argv) > 4 else 300
exact = len(sys.argv) > 5 and sys.argv[5] == "exact"
res = []
for sw in ({}, {"RR_NO_MEMO": "1", "RR_NO_ORDER": "1"}):
    old = {k: os.environ.get(k) for k in sw}
    os.environ.update(sw)
    try:
        env = rr.BatchedRoboRugbyEnv(n, preset=preset, seed=4, exact_trig=exact)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    env.reset()
    st = env.get_state()
    T = env.preset.game_len_steps
    env.set_state(st["robots"], st["robots_i"], st["balls"], torch.randint(0, T, (n,), generator=g, device="cuda", dtype=torch.int32))
    na = env.preset.nr
    chk = torch.zeros(4, dtype=torch.float64, device="cuda")
    w = torch.linspace(0.5, 1.5, 11, device="cuda", dtype=torch.float64)
    for s in range(pre + S):
        a = torch.randint(0, 8, (n, na), generator=g, device="cuda", dtype=torch.int32)
        o, r, d, info = env.step_f64(a)
        if s >= pre:
            chk[0] += (torch.nan_to_num(o) * w).sum() * (1 + (s % 7)); chk[1] += r.sum() * (1 + (s % 5)); chk[2] += d.sum(); chk[3] += (info.status & 0xFFFF).sum()
    fin = env.get_state()
    res.append((chk.cpu().numpy(), {k: v.cpu().numpy() for k, v in fin.items()}))
    print(preset, "parity build" if exact else "default build", "shortcuts", "off" if sw else "on", "checksums", res[-1][0], flush=True)
    env.close()
same = np.array_equal(res[0][0], res[1][0]) and all(np.array_equal(res[0][1][k], res[1][1][k], equal_nan=True) for k in res[0][1])
print("IDENTICAL" if same else "DIFFERENT")
sys.exit(0 if same else 1)
