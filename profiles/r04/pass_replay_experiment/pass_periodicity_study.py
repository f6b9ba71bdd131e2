import os, sys, subprocess, collections
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np
preset = sys.argv[1]; idxs = [int(x) for x in sys.argv[2].split(",")]
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if os.environ.get("CHILD"):
    import emu_lib, ctypes
    L = emu_lib.lib()
    d = np.load(f"/root/repo/tests/data/stuck_chase_{preset}.npz")
    for idx in idxs:
        e = emu_lib.EmuEnv(preset)
        e.set_state(d["robots"][idx], d["robots_i"][idx], d["balls"][idx], int(d["step"][idx]))
        L.emu_debug_trace(2)
        for s in range(nsteps):
            sys.stderr.write(f"E STEP {idx} {s}\n"); sys.stderr.flush()
            e.step(d["actions"][idx])
        L.emu_debug_trace(0)
    sys.exit(0)
p = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=dict(os.environ, CHILD="1", RR_NO_MEMO="1"), stderr=subprocess.PIPE, text=True)
lines = p.stderr.splitlines()
loops = []; cur = []; ev = []
for ln in lines:
    if ln.startswith("E pass") and " state" in ln:
        vals = ln.split(" state ")[1].split()
        nb = len(vals) // 8
        pos = tuple(v for b in range(nb) for v in vals[8*b:8*b+6])
        cur.append((pos, tuple(ev))); ev = []
    elif ln.startswith("E resolve") or ln.startswith("E STEP"):
        if cur: loops.append(cur)
        cur = []; ev = []
    elif ln.startswith("E bounce") or ln.startswith("E pass") :
        ev.append(" ".join(ln.split()[1:7]).split(" v=")[0])
if cur: loops.append(cur)
stat = collections.Counter(); tot = 0; saved = 0
for lp in loops:
    n = len(lp); tot += n
    # first pass k (0-based idx) whose end positions equal the previous pass's end positions (=> geometry of pass k+1.. same as pass k, if events same)
    first = None
    for k in range(1, n):
        if lp[k][0] == lp[k-1][0]: first = k; break
    if first is None: stat[f"len{n} never"] += 1; continue
    # check that from `first` on, events are identical and positions stay
    ok = all(lp[k][0] == lp[first][0] and lp[k][1] == lp[first][1] for k in range(first, n))
    bbhit = any("bb" in e for e in lp[first][1])
    stat[f"len{n} pos-fixed from pass {first+1} stays={ok} bb={bbhit}"] += 1
    if ok: saved += n - (first + 1)
for k, v in sorted(stat.items()): print(k, v)
print("passes total", tot, "replayable", saved)
