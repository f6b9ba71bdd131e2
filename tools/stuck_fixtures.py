"""Builds tests/data/stuck_chase_{T,G}.npz from gpurun_out/chase_monsters_{T,G}.npz (tools/chase_monsters.py): the arenas of the
slowest wavefronts of a chase-policy rollout on the MI355X in which the resolve loop gives up -- robots driving balls into walls,
into each other, ball clusters.  Test material for the exact shortcuts (tests/test_fixed_point_memo.py).  Dev tool: uses the
host-emulated wave of the test harness to pick the arenas."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import emu_lib as el
from test_fixed_point_memo import _count_events

for preset, cap in (("T", 64), ("G", 40)):
    d = np.load(os.path.join(ROOT, "gpurun_out", f"chase_monsters_{preset}.npz"))
    cand = []
    nrec = sum(1 for k in d.files if k.startswith("robots_i_"))
    for i in range(nrec):
        for a in range(d[f"robots_{i}"].shape[0]):
            cand.append((d[f"robots_{i}"][a], d[f"robots_i_{i}"][a], d[f"balls_{i}"][a], int(d[f"stepc_{i}"][a]), d[f"actions_{i}"][a]))
    for a in range(d["slow_robots"].shape[0]):
        cand.append((d["slow_robots"][a], d["slow_robots_i"][a], d["slow_balls"][a], int(d["slow_stepc"][a]), d["slow_actions"][a]))
    keep, seen = [], set()
    for c in cand:
        key = c[0].tobytes() + c[2].tobytes()
        if key in seen:
            continue
        seen.add(key)
        def run():
            e = el.EmuEnv(preset, time_limit=1, auto_reset=1)
            e.set_state(c[0], c[1], c[2], c[3])
            e.step(c[4])
        n = _count_events(run, ("resolve gave up",))["resolve gave up"]
        if n >= 2:
            keep.append(c)
        if len(keep) >= cap:
            break
    out = os.path.join(ROOT, "tests", "data", f"stuck_chase_{preset}.npz")
    np.savez_compressed(out, robots=np.array([k[0] for k in keep]), robots_i=np.array([k[1] for k in keep]),
                        balls=np.array([k[2] for k in keep]), step=np.array([k[3] for k in keep], np.int32),
                        actions=np.array([k[4] for k in keep], np.int32))
    print(preset, "kept", len(keep), "of", len(seen), "->", out, os.path.getsize(out), "bytes")
