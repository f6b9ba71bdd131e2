"""Soak of the budgeted step: a long chase-policy rollout (policy in the loop, auto-reset, episode ends, faults) stepped synchronously
and with a budget; every arena's stream of (obs, reward, done, status) over its first K ACCEPTED steps is folded into a per-arena
fp64 checksum (order-sensitive) and the two runs must agree exactly.  Exits non-zero on any difference.
usage: python tools/soak_budget.py [T|G] [arenas] [accepted steps K] [budget clocks]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import roborugby_amd as rr
from roborugby_amd import players
preset = sys.argv[1] if len(sys.argv) > 1 else "T"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
K = int(sys.argv[3]) if len(sys.argv) > 3 else 700
budget = int(sys.argv[4]) if len(sys.argv) > 4 else 100000
NOT_READY = 16384
w = torch.linspace(0.5, 1.5, 11, device="cuda", dtype=torch.float64)


def run(b):
    env = rr.BatchedRoboRugbyEnv(n, preset=preset, seed=11, step_budget_clocks=b)
    obs = env.reset()
    out = (obs.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, dtype=torch.uint8, device="cuda"),
           torch.zeros(n, 11, device="cuda") if env.has_grumpy else None, torch.zeros(n, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda"))
    acc = torch.zeros(n, dtype=torch.int32, device="cuda")
    chk = torch.zeros(n, dtype=torch.float64, device="cuda")
    acts = torch.zeros(n, env.preset.nr, dtype=torch.int32, device="cuda")
    calls = not_ready = episodes = faults = 0
    while int(acc.min()) < K:
        players.chase(env, out[0], step_of=acc, noise=0.1, seed=5, out=acts)
        env.step(acts, out=out)
        calls += 1
        ready = ((out[5] & NOT_READY) == 0) & (acc < K)
        k1 = (acc + 1).double()
        term = (out[0].double() * w).sum(1) * k1 + out[1].double() * k1 * 3.0 + out[2].double() * 7.0 + (out[5] & 0xFFFF).double()
        if out[3] is not None:
            term = term + (torch.nan_to_num(out[3].double()) * w).sum(1) * k1 * 0.5 + out[4].double() * k1
        chk += torch.where(ready, term, torch.zeros_like(term))
        not_ready += int(((out[5] & NOT_READY) != 0).sum())
        episodes += int((out[2].bool() & ready).sum())
        faults += int((((out[5] & 63) != 0) & ready).sum())
        acc += ready.to(torch.int32)
        assert calls < 400 * K, "no progress"
    env.close()
    return chk, calls, not_ready, episodes, faults


a, ca, _, ea, fa = run(0)
b, cb, nr, eb, fb = run(budget)
same = bool(torch.equal(a, b))
print(f"{preset}: {n} arenas x {K} accepted chase steps: synchronous {ca} calls ({ea} episode ends, {fa} faulted steps), budget {budget} clocks: "
      f"{cb} calls, {nr} NOT_READY rows ({100.0 * nr / (cb * n):.2f} %), {eb} episode ends; per-arena stream checksums identical: {same}")
sys.exit(0 if same and ea == eb and fa == fb else 1)
