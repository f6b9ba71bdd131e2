#!/usr/bin/env python3
"""Golden-vector generator -- TEST INFRASTRUCTURE ONLY (build container only).

Imports the *unmodified* reference from /root/reference under the stand-ins in
standins.py and writes small .npz fixtures into tests/golden/:

  kat_<preset>.npz    primitive known-answer tables (MyUtils.py / RR_TrashyPhysics.py)
  traj_<preset>.npz   full-state trajectories of SimpleDuel3.step()
                      (RR_EnvBase.py:260-297) under scripted policies
  reset_<preset>.npz  layouts produced by reference reset() (RR_EnvBase.py:155-216)
  mix_<preset>.npz    other reward / observer mixins (part "mix")
  thrust_<preset>.npz trajectories through the continuous-thrust entry GameEnv.step (part "thrust")

usage:  python oracle/refgen/gen_golden.py [G|T|all]
One preset per process (constants are module globals in the reference), so
`all` re-invokes this script once per preset.
"""
import json
import math
import os
import random
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
GOLD = os.environ.get("RR_GOLDEN_OUT") or os.path.join(REPO, "tests", "golden")  # (tests regenerate into a scratch directory)
sys.path.insert(0, HERE)

META = {
    "generator": "oracle/refgen/gen_golden.py",
    "reference": "harman097/RoboRugby @ /root/reference (unmodified, imported under stand-ins)",
    "standin_assumptions": [
        "pygame.Rect truncates each ctor arg toward zero; right=left+width; colliderect strict",
        "pygame Group iterates in insertion order",
        "no gym TimeLimit wrapper (env constructed directly)",
    ],
    "python": sys.version.split()[0],
}

NAN = float("nan")


# ------------------------------------------------------------------ state dump
def robot_state(rb):
    r = rb.rectDbl
    i = (rb.lngMoveCount - 1) % 360
    st = rb._lstStates[i]
    if st is None or st[3] != rb.lngMoveCount - 1:
        px, py, prot = NAN, NAN, NAN
    else:
        px, py, prot = st[0], st[1], st[2]
    return ([r._dblCenterX, r._dblCenterY, r._dblLeft, r._dblRight, r._dblTop, r._dblBottom,
             float(r._dblRotation), px, py, float(prot)],
            [rb.lngMoveCount, rb.lngLThrust, rb.lngRThrust])


def ball_state(b):
    r = b.rectDbl
    return [r._dblCenterX, r._dblCenterY, r._dblLeft, r._dblRight, r._dblTop, r._dblBottom,
            float(b.dbl_velocity_x), float(b.dbl_velocity_y)]


def dump_state(env, R):
    rob = [robot_state(rb) for rb in env.lstRobots]
    inner = R.tp._rectBallInner
    return dict(
        robots=np.array([x[0] for x in rob], dtype=np.float64),
        robots_i=np.array([x[1] for x in rob], dtype=np.int32),
        balls=np.array([ball_state(b) for b in env.lstBalls], dtype=np.float64),
        inner=np.array([inner._dblCenterX, inner._dblCenterY, float(inner._dblRotation)], dtype=np.float64),
        step=np.int32(env.lngStepCount),
    )


# ------------------------------------------------------------------ policies
def angdiff(a, b):
    return (a - b + 540.0) % 360.0 - 180.0


def chase_action(obs, rng, noise):
    if obs is None or rng.random() < noise:
        return rng.randint(0, 7)
    d = angdiff(obs[1], obs[0])
    if abs(d) < 8:
        return 0
    if abs(d) < 40:
        return 4 if d > 0 else 5
    return 2 if d > 0 else 3


def policy_actions(env, R, kind, rng, NA):
    acts = []
    for i in range(NA):
        rb = env.lstRobots[i]
        if kind == "random":
            acts.append(rng.randint(0, 7))
        elif kind == "forward":
            acts.append(0 if rng.random() < 0.8 else rng.randint(0, 7))
        else:
            noise = {"chase": 0.05, "chase_noisy": 0.3}[kind]
            # every robot chases a (per-robot) ball so that contacts of all kinds happen
            ball = env.lstBalls[i % len(env.lstBalls)]
            o = env.get_game_state(obj_robot=rb, obj_ball=ball)
            acts.append(chase_action(o, rng, noise))
    return acts


# ------------------------------------------------------------------ coverage counters
class Counters:
    NAMES = ["apply_force_to_ball", "bounce_ball_off_bot", "bounce_balls", "bounce_ball_off_wall"]

    def __init__(self, R):
        self.c = {k: 0 for k in self.NAMES}
        self.c["undo_naughty"] = 0
        self.c["robot_collision"] = 0
        for name in self.NAMES:
            orig = getattr(R.tp, name)
            setattr(R.tp, name, self._wrap(name, orig))
        cls = R.base.GameEnv
        orig_undo = cls._undo_naughty_movement
        me = self

        def undo(self_env, a, b):
            me.c["undo_naughty"] += 1
            return orig_undo(self_env, a, b)
        cls._undo_naughty_movement = undo
        nb = R.sk.NaughtyBots
        orig_col = nb.on_robot_collision

        def col(self_env, b1, b2):
            me.c["robot_collision"] += 1
            return orig_col(self_env, b1, b2)
        nb.on_robot_collision = col
        # GAME_MODE=True turns the undo-loop fault into print + loop-forever (RR_EnvBase.py:417-421).
        # Intercept that module's print so a hang becomes an exception the generator can record.
        self.warns = 0

        def guarded_print(*a, **k):
            if a and str(a[0]).startswith("WARNING: UNABLE TO RESOLVE ALL"):
                me.warns += 1
                if me.warns > 4:
                    raise RuntimeError("UNABLE TO RESOLVE ALL COLLISIONS FOR FRAME (reference spins forever)")
        R.base.print = guarded_print

    def _wrap(self, name, fn):
        def w(*a, **k):
            self.c[name] += 1
            return fn(*a, **k)
        return w


# ------------------------------------------------------------------ trajectories
EXC_CODES = {  # message prefix -> status bit (mirrors include/roborugby_amd.h RR_STATUS_*)
    "UNABLE TO RESOLVE BOT/BOT": 1,
    "ROBOTS STUCK": 2,
    "UNABLE TO UNDO MOVE": 4,
    "UNABLE TO RESOLVE ALL": 8,
    "Really tho": 16,
    "Numerator AND": 32,
    "Game is over": 64,
}


def exc_code(e):
    msg = str(e)
    for k, v in EXC_CODES.items():
        if msg.startswith(k):
            return v
    return 1 << 20


def run_episode(R, env, kind, NA, nsteps, rng, scramble, vmax, cnt=None):
    const = R.const
    random.seed(rng.randint(0, 1 << 30))
    env.reset()
    if scramble:
        for b in env.lstBalls:
            b.set_velocity(rng.uniform(-vmax, vmax), rng.uniform(-vmax, vmax))
    NR, NB = len(env.lstRobots), len(env.lstBalls)
    S = dict(robots=np.zeros((nsteps + 1, NR, 10)), robots_i=np.zeros((nsteps + 1, NR, 3), np.int32),
             balls=np.zeros((nsteps + 1, NB, 8)), inner=np.zeros((nsteps + 1, 3)),
             step=np.zeros(nsteps + 1, np.int32))
    out = dict(actions=np.full((nsteps, NR), -1, np.int32), obs=np.full((nsteps, 11), NAN),
               obs_g=np.full((nsteps, 11), NAN), reward=np.zeros(nsteps), reward_g=np.zeros(nsteps),
               done=np.zeros(nsteps, np.uint8), naughty=np.zeros(nsteps, np.int32), warn=np.zeros(nsteps, np.int32))

    def put(i):
        d = dump_state(env, R)
        for k in S:
            S[k][i] = d[k]
    put(0)
    obs0 = env.get_game_state()
    obs0_g = env.get_game_state(int_team=const.TEAM_GRUMPY)
    length, exc = 0, 0
    for s in range(nsteps):
        acts = policy_actions(env, R, kind, rng, NA)
        out["actions"][s, :NA] = acts
        try:
            if cnt is not None:
                cnt.warns = 0
            o, r, d, info = env.step(acts)
        except Exception as e:  # the reference raises from inside step
            exc = exc_code(e)
            break
        out["obs"][s] = o
        if info.adblGrumpyState is not None:
            out["obs_g"][s] = info.adblGrumpyState
        out["reward"][s] = r
        out["reward_g"][s] = info.dblGrumpyScore
        out["done"][s] = d
        nm = 0
        for i, rb in enumerate(env.lstRobots):
            if rb in env.set_naughty_bots:
                nm |= 1 << i
        out["naughty"][s] = nm
        out["warn"][s] = cnt.warns if cnt is not None else 0
        put(s + 1)
        length = s + 1
        if d:
            break
    out.update({"state_" + k: v for k, v in S.items()})
    out["length"] = np.int32(length)
    out["exc"] = np.int32(exc)
    out["obs0"] = np.asarray(obs0, dtype=np.float64)
    out["obs0_g"] = np.full(11, NAN) if obs0_g is None else np.asarray(obs0_g, dtype=np.float64)
    return out


def reset_scratch_rect(R):
    """The module-global scratch rect _rectBallInner (RR_TrashyPhysics.py:29-36) keeps whatever the last contact test of
    the process left in it; episode 0 must not inherit that from the KAT / reset generators that ran before."""
    R.tp._rectBallInner.center = (0.0, 0.0)
    R.tp._rectBallInner.rotation = 0


def gen_traj(R, preset):
    cnt = Counters(R)
    env = R.envs.SimpleDuel3()
    reset_scratch_rect(R)
    NR = len(env.lstRobots)
    rng = random.Random({"G": 20201, "T": 20202, "D": 20203, "X": 20204}[preset])
    if preset in ("D", "X"):  # two robots (one per team), two balls (one of each colour), G's arena and episode length
        plan = [("random", NR, 150, False, 0), ("chase", NR, 300, False, 0), ("chase_noisy", NR, 300, True, 6.0),
                ("chase", NR, 300, True, 9.0), ("forward", NR, 200, True, 3.0), ("chase", 1, 300, True, 5.0),
                ("chase_noisy", NR, 300, True, 12.0), ("chase", NR, 300, True, 2.0), ("random", NR, 200, True, 10.0),
                ("chase", NR, 300, False, 0), ("chase_noisy", NR, 300, True, 4.0), ("chase", NR, 300, True, 7.0)]
    elif preset == "G":
        plan = [("random", NR, 60, False, 0), ("chase", NR, 220, False, 0), ("chase_noisy", NR, 220, True, 6.0),
                ("chase", NR, 220, True, 9.0), ("forward", NR, 150, True, 3.0), ("chase", 1, 150, True, 5.0),
                ("chase", 2, 150, False, 0), ("chase_noisy", NR, 220, True, 12.0), ("chase", NR, 220, True, 2.0),
                ("random", NR, 100, True, 10.0), ("chase", NR, 220, False, 0), ("chase_noisy", NR, 220, True, 4.0),
                ("chase_noisy", NR, 220, True, 8.0), ("chase", NR, 220, True, 5.0), ("chase_noisy", 3, 220, True, 7.0),
                ("chase", NR, 220, True, 3.0), ("chase_noisy", NR, 220, False, 0), ("forward", NR, 220, True, 11.0)]
    else:
        plan = [("random", 1, 150, False, 0), ("chase", 1, 302, False, 0), ("chase_noisy", 1, 302, True, 6.0),
                ("chase", 1, 302, True, 9.0), ("forward", 1, 200, True, 3.0), ("chase", 1, 302, True, 12.0),
                ("chase", 1, 302, False, 0), ("chase_noisy", 1, 302, True, 2.0), ("random", 1, 302, True, 10.0),
                ("chase", 1, 302, True, 4.0), ("chase", 1, 302, False, 0), ("chase", 1, 302, True, 7.0),
                ("chase_noisy", 1, 302, True, 5.0), ("chase", 1, 302, True, 1.0), ("chase", 1, 302, False, 0),
                ("chase", 1, 302, True, 8.0)]
    eps = []
    for (kind, NA, n, scr, vmax) in plan:
        ep = run_episode(R, env, kind, NA, n, rng, scr, vmax, cnt)
        print(preset, kind, NA, "len", int(ep["length"]), "exc", int(ep["exc"]), cnt.c, flush=True)
        eps.append(ep)
    smax = max(e["actions"].shape[0] for e in eps)
    packed = {}
    for k in eps[0]:
        arrs = [e[k] for e in eps]
        if arrs[0].ndim == 0 or k in ("obs0", "obs0_g"):
            packed[k] = np.stack(arrs)
            continue
        tgt = smax + 1 if k.startswith("state_") else smax
        pad = []
        for a in arrs:
            p = np.zeros((tgt,) + a.shape[1:], a.dtype)
            p[:a.shape[0]] = a
            pad.append(p)
        packed[k] = np.stack(pad)
    meta = dict(META, preset=preset, plan=[list(p) for p in plan], coverage=cnt.c,
                state_layout=dict(robots="cx,cy,left,right,top,bottom,rot,prev_x,prev_y,prev_rot (prev = ring entry moveCount-1, NaN if absent)",
                                  robots_i="moveCount,lthrust,rthrust", balls="cx,cy,left,right,top,bottom,vx,vy",
                                  inner="_rectBallInner cx,cy,rot (module-global scratch rect, RR_TrashyPhysics.py:29-36)"))
    packed["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(GOLD, f"traj_{preset}.npz"), **packed)


# ------------------------------------------------------------------ continuous-thrust entry (RR_EnvBase.py:260-273)
THRUST_MENU = [-2.5, -1.5, -1.0, -0.51, -0.5, -0.49, 0.0, 0.49, 0.5, 0.51, 1.0, 1.5, 2.5]
THRUST_SCALE = [1.0, 1.0, 1.0, 0.51, 1.49, 0.5, 1.5, 2.5]  # of a Direction's (L, R): rounds to 1,1,1,1,1,0,2,2
DIR_LR = [(1, 1), (-1, -1), (-1, 1), (1, -1), (0, 1), (1, 0), (-1, 0), (0, -1)]  # RR_EnvBase.py:593-602


def run_thrust_episode(R, env, NK, nsteps, rng, scramble, vmax, cnt, dtype):
    """Like run_episode, but drives the reference through GameEnv.step (the Box-thrust entry, bypassing
    GameEnv_Simple.step's Direction table) with flat (L, R) float pairs for the first NK robots."""
    const = R.const
    random.seed(rng.randint(0, 1 << 30))
    env.reset()
    if scramble:
        for b in env.lstBalls:
            b.set_velocity(rng.uniform(-vmax, vmax), rng.uniform(-vmax, vmax))
    NR, NB = len(env.lstRobots), len(env.lstBalls)
    S = dict(robots=np.zeros((nsteps + 1, NR, 10)), robots_i=np.zeros((nsteps + 1, NR, 3), np.int32),
             balls=np.zeros((nsteps + 1, NB, 8)), inner=np.zeros((nsteps + 1, 3)), step=np.zeros(nsteps + 1, np.int32))
    out = dict(thrust=np.full((nsteps, NR, 2), NAN), obs=np.full((nsteps, 11), NAN), obs_g=np.full((nsteps, 11), NAN),
               reward=np.zeros(nsteps), reward_g=np.zeros(nsteps), done=np.zeros(nsteps, np.uint8),
               naughty=np.zeros(nsteps, np.int32), warn=np.zeros(nsteps, np.int32))

    def put(i):
        d = dump_state(env, R)
        for k in S:
            S[k][i] = d[k]
    put(0)
    length, exc = 0, 0
    for s in range(nsteps):
        pairs = []
        for i in range(NK):
            u = rng.random()
            if u < 0.55:  # a chasing Direction, scaled so that round() lands on 0 / 1 / 2 through the half-way cases
                ball = env.lstBalls[i % NB]
                a = chase_action(env.get_game_state(obj_robot=env.lstRobots[i], obj_ball=ball), rng, 0.1)
                k = rng.choice(THRUST_SCALE)
                pairs.append((DIR_LR[a][0] * k, DIR_LR[a][1] * k))
            elif u < 0.9:
                pairs.append((rng.choice(THRUST_MENU), rng.choice(THRUST_MENU)))
            else:
                pairs.append((rng.uniform(-2.6, 2.6), rng.uniform(-2.6, 2.6)))
        arr = np.asarray(pairs, dtype=dtype)
        out["thrust"][s, :NK] = arr.astype(np.float64)  # exactly the values the reference saw
        try:
            cnt.warns = 0
            # the form main.py / a Box policy use: a list of per-robot (L, R) tuples (RR_EnvBase.py:269 flattens it)
            o, r, d, info = R.base.GameEnv.step(env, [tuple(p) for p in arr])
        except Exception as e:
            exc = exc_code(e)
            break
        out["obs"][s] = o
        if info.adblGrumpyState is not None:
            out["obs_g"][s] = info.adblGrumpyState
        out["reward"][s] = r
        out["reward_g"][s] = info.dblGrumpyScore
        out["done"][s] = d
        nm = 0
        for i, rb in enumerate(env.lstRobots):
            if rb in env.set_naughty_bots:
                nm |= 1 << i
        out["naughty"][s] = nm
        out["warn"][s] = cnt.warns
        put(s + 1)
        length = s + 1
        if d:
            break
    out.update({"state_" + k: v for k, v in S.items()})
    out["length"] = np.int32(length)
    out["exc"] = np.int32(exc)
    return out


def gen_thrust(R, preset):
    cnt = Counters(R)
    env = R.envs.SimpleDuel3()
    reset_scratch_rect(R)
    NR = len(env.lstRobots)
    rng = random.Random(31001 if preset == "G" else 31002)
    if preset == "G":  # (robots driven, steps, scramble ball velocities, vmax, dtype the policy hands over)
        plan = [(NR, 120, False, 0, np.float64), (NR, 120, True, 6.0, np.float32), (2, 100, True, 4.0, np.float64),
                (1, 80, False, 0, np.float32), (NR, 120, True, 9.0, np.float64), (3, 100, True, 3.0, np.float64)]
    else:
        plan = [(1, 200, False, 0, np.float64), (1, 200, True, 6.0, np.float32), (1, 302, True, 3.0, np.float64),
                (1, 200, True, 9.0, np.float64), (1, 302, False, 0, np.float32)]
    eps = []
    for (NK, n, scr, vmax, dt) in plan:
        ep = run_thrust_episode(R, env, NK, n, rng, scr, vmax, cnt, dt)
        print(preset, "thrust", NK, "len", int(ep["length"]), "exc", int(ep["exc"]), cnt.c, flush=True)
        eps.append(ep)
    smax = max(e["thrust"].shape[0] for e in eps)
    packed = {}
    for k in eps[0]:
        arrs = [e[k] for e in eps]
        if arrs[0].ndim == 0:
            packed[k] = np.stack(arrs)
            continue
        tgt = smax + 1 if k.startswith("state_") else smax
        pad = []
        for a in arrs:
            p = np.full((tgt,) + a.shape[1:], NAN, a.dtype) if k == "thrust" else np.zeros((tgt,) + a.shape[1:], a.dtype)
            p[:a.shape[0]] = a
            pad.append(p)
        packed[k] = np.stack(pad)
    rounded = sorted({float(v) for v in np.unique(np.rint(packed["thrust"][~np.isnan(packed["thrust"])]))})
    meta = dict(META, preset=preset, plan=[[a, b, c, d, np.dtype(e).name] for (a, b, c, d, e) in plan], coverage=cnt.c,
                entry="GameEnv.step(env, [(L, R), ...]) -- RR_EnvBase.py:260-273, Robot.set_thrust RR_Robot.py:100-102",
                thrust_menu=THRUST_MENU, thrust_scales=THRUST_SCALE, rounded_values_seen=rounded,
                note="thrust[e, s, i] = the (L, R) floats robot i was given (NaN: robot got none and keeps its thrust)")
    packed["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(GOLD, f"thrust_{preset}.npz"), **packed)


# ------------------------------------------------------------------ KATs
def gen_kat(R, preset):
    MU, tp, const = R.MyUtils, R.tp, R.const
    rng = random.Random(777)
    out = {}
    # line intersection + within (MyUtils.py:61-94)
    segs, res, win = [], [], []
    specials = [((0, 0), (10, 0), (5, -5), (5, 5)), ((0, 0), (0, 10), (-5, 5), (5, 5)),
                ((0, 0), (10, 10), (0, 1), (10, 11)), ((0, 0), (0, 10), (3, 0), (3, 10)),
                ((0, 0), (10, 0), (0, 5), (10, 5)), ((1, 1), (4, 5), (4, 5), (9, 2)),
                ((0, 0), (800, 0), (100, 50), (300, 50.0000001)), ((800, 0), (800, 800), (10, 10), (10, 400))]
    for i in range(400):
        if i < len(specials):
            a, b, c, d = specials[i]
        else:
            sc = rng.choice([1.0, 30.0, 800.0])
            a, b, c, d = [(rng.uniform(-sc, sc), rng.uniform(-sc, sc)) for _ in range(4)]
            if i % 7 == 0:
                b = (a[0], b[1])  # vertical
            if i % 11 == 0:
                d = (d[0], c[1])  # horizontal
        p = MU.get_line_intersection((a, b), (c, d))
        segs.append([*a, *b, *c, *d])
        res.append([p[0], p[1]])
        win.append([MU.point_within_line(p, (a, b)), MU.point_within_line(p, (c, d)),
                    MU.point_within_line(p, (c, d), buffer=.5)])
    out["li_in"] = np.array(segs, dtype=np.float64)
    out["li_out"] = np.array(res, dtype=np.float64)
    out["li_within"] = np.array(win, dtype=np.uint8)
    # distance / angle_degrees (MyUtils.py:40-41,97-110)
    pts, dist, ang = [], [], []
    for i in range(300):
        a = (rng.uniform(0, 800), rng.uniform(0, 800))
        b = (rng.uniform(0, 800), rng.uniform(0, 800))
        if i % 13 == 1:
            b = (a[0], b[1])
        elif i % 17 == 2:
            b = (b[0], a[1])
        pts.append([*a, *b])
        dist.append(MU.distance(a, b))
        ang.append(MU.angle_degrees(a, b))
    out["pt_in"] = np.array(pts)
    out["pt_dist"] = np.array(dist)
    out["pt_angle"] = np.array(ang)
    # FloatRect rotation -> corners + edges, copy() (MyUtils.py:114-154,277-322)
    rin, rout, rcopy = [], [], []
    rots = [0, 90, 180, 270, 360, -90, 45, 0.6, 1.2, 359.4, 720.5, -0.6] + [rng.uniform(-400, 800) for _ in range(150)] \
        + [rng.randint(0, 360) for _ in range(40)]
    for j, rot in enumerate(rots):
        for (w, h) in ((20, 40), (14, 14)):
            cx, cy = rng.uniform(0, 800), rng.uniform(0, 800)
            fr = MU.FloatRect(0, w, 0, h)
            fr.center = (cx, cy)
            fr.rotation = rot
            cs = fr.corners
            rin.append([w, h, cx, cy, rot])
            rout.append([fr._dblCenterX, fr._dblCenterY, fr._dblLeft, fr._dblRight, fr._dblTop, fr._dblBottom,
                         float(fr._dblRotation)] + [v for c in cs for v in c])
            c2 = fr.copy()
            rcopy.append([c2._dblCenterX, c2._dblCenterY, c2._dblLeft, c2._dblRight, c2._dblTop, c2._dblBottom,
                          float(c2._dblRotation)])
    out["fr_in"] = np.array(rin, dtype=np.float64)
    out["fr_out"] = np.array(rout, dtype=np.float64)
    out["fr_copy"] = np.array(rcopy, dtype=np.float64)
    # two_way_lidar_rect against walls + one rect (RR_TrashyPhysics.py:365-391)
    lin, lout = [], []
    walls = MU.FloatRect(0, const.ARENA_WIDTH, 0, const.ARENA_HEIGHT)
    for i in range(200):
        o = MU.FloatRect(0, 20, 0, 40)
        o.center = (rng.uniform(50, 550), rng.uniform(50, 550))
        o.rotation = rng.choice([0, 90, rng.uniform(0, 360)])
        s = (rng.uniform(20, 580), rng.uniform(20, 580))
        th = rng.choice([0.0, math.pi / 2, rng.uniform(0, 2 * math.pi)])
        e = (s[0] + 40 * math.cos(th), s[1] + 40 * math.sin(th))
        if i % 9 == 0:
            e = (s[0] + 40.0, s[1])
        if i % 10 == 0:
            e = (s[0], s[1] + 40.0)
        f, b = tp.two_way_lidar_rect(s, e, [o, walls])
        lin.append([o._dblCenterX, o._dblCenterY, float(o._dblRotation), *s, *e])
        lout.append([f, b])
    out["lidar_in"] = np.array(lin)
    out["lidar_out"] = np.array(lout)
    # ball_robot_collided / robots_collided (RR_TrashyPhysics.py:18-69)
    class _S:
        pass
    bin_, bout = [], []
    inner = tp._rectBallInner
    for i in range(600):
        bot = _S()
        bot.rectDbl = MU.FloatRect(0, 20, 0, 40)
        bot.rectDbl.center = (300.0, 300.0)
        bot.rectDbl.rotation = rng.choice([0, 90, 37.2, rng.uniform(0, 360)])
        ball = _S()
        ball.rectDbl = MU.FloatRect(0, 14, 0, 14)
        r = rng.uniform(5, 32)
        th = rng.uniform(0, 2 * math.pi)
        ball.rectDbl.center = (300.0 + r * math.cos(th), 300.0 + r * math.sin(th))
        pre = [inner._dblCenterX, inner._dblCenterY, float(inner._dblRotation)]
        hit = tp.ball_robot_collided(ball, bot)
        bin_.append([float(bot.rectDbl._dblRotation), ball.rectDbl._dblCenterX, ball.rectDbl._dblCenterY] + pre)
        bout.append(hit)
    out["brc_in"] = np.array(bin_)
    out["brc_out"] = np.array(bout, dtype=np.uint8)
    rin2, rout2 = [], []
    for i in range(400):
        b1, b2 = _S(), _S()
        b1.rectDbl = MU.FloatRect(0, 20, 0, 40)
        b1.rectDbl.center = (300.0, 300.0)
        b1.rectDbl.rotation = rng.choice([0, 90, rng.uniform(0, 360)])
        b2.rectDbl = MU.FloatRect(0, 20, 0, 40)
        r = rng.uniform(0.5, 50)
        th = rng.uniform(0, 2 * math.pi)
        b2.rectDbl.center = (300.0 + r * math.cos(th), 300.0 + r * math.sin(th))
        b2.rectDbl.rotation = rng.choice([0, 270, rng.uniform(0, 360)])
        rin2.append([float(b1.rectDbl._dblRotation), b2.rectDbl._dblCenterX, b2.rectDbl._dblCenterY,
                     float(b2.rectDbl._dblRotation)])
        rout2.append(tp.robots_collided(b1, b2))
    out["rrc_in"] = np.array(rin2)
    out["rrc_out"] = np.array(rout2, dtype=np.uint8)
    out["consts"] = np.array([const.ARENA_WIDTH, const.ARENA_HEIGHT, const.GAME_LENGTH_STEPS,
                              const.POINTS_BALL_TRAVEL_MULT, const.POINTS_ROBOT_TRAVEL_MULT,
                              const.NUM_ROBOTS_HAPPY, const.NUM_ROBOTS_GRUMPY, const.NUM_BALL_POS, const.NUM_BALL_NEG,
                              const.MOVES_PER_FRAME, const.CALC_DIST_TRACK_CENTER_TO_ROBOT_CENTER], dtype=np.float64)
    out["meta"] = np.array(json.dumps(dict(META, preset=preset)))
    np.savez_compressed(os.path.join(GOLD, f"kat_{preset}.npz"), **out)


# ------------------------------------------------------------------ reset layouts
def gen_reset(R, preset, n=1000):
    env = R.envs.SimpleDuel3()
    rob, bal, obs = [], [], []
    for k in range(n):
        random.seed(1000 + k)
        o = env.reset()
        pos = env._get_positions()
        rob.append(pos[0])
        bal.append(pos[1])
        obs.append(o)
    np.savez_compressed(os.path.join(GOLD, f"reset_{preset}.npz"), robots=np.array(rob, dtype=np.float64),
                        balls=np.array(bal, dtype=np.float64), obs=np.array(obs, dtype=np.float64),
                        meta=np.array(json.dumps(dict(META, preset=preset, seeds="random.seed(1000+k) before reset()"))))


# ------------------------------------------------------------------ other mixins
KEEPER_IDS = {"NaughtyBots": 1, "ChasePosBall": 2, "PushPosBallsToGoal": 3, "DontDriveInGoals": 4, "KeepMovingGuys": 5,
              "BaseDestruction": 6, "PushNegBallsFromGoal": 7}


def exec_order(mro_names):
    """on_step_end execution order of a keeper MRO: every keeper calls super() first, except NaughtyBots which
    does not call it at all (RR_ScoreKeepers.py:130-135), so keepers after it in the MRO never run."""
    names = list(mro_names)
    if "NaughtyBots" in names:
        names = names[:names.index("NaughtyBots") + 1]
    return [KEEPER_IDS[n] for n in reversed(names)]


def alt_observations(R, env):
    """The other observer mixins evaluated on the same env state through their unbound methods (their helper
    methods are bound to the env for the duration of the call)."""
    import types
    O, c = R.obs, R.const

    def call(cls, **kw):
        added = []
        for name in ("_robot_state", "_ball_state"):
            if hasattr(cls, name) and name not in env.__dict__:
                setattr(env, name, types.MethodType(getattr(cls, name), env))
                added.append(name)
        try:
            return cls.get_game_state(env, **kw)
        finally:
            for name in added:
                delattr(env, name)
    out = {}
    for tag, team in (("h", c.TEAM_HAPPY), ("g", c.TEAM_GRUMPY)):
        has = (len(env.lstHappyBots) if team == c.TEAM_HAPPY else len(env.lstGrumpyBots)) > 0
        v1 = call(O.SingleBall_6wayLidar, int_team=team) if has else None
        bas = call(O.PosBall_BasicLidar, int_team=team) if has else None
        out["v1_" + tag] = np.full(11, NAN) if v1 is None else np.asarray(v1, dtype=np.float64)
        out["basic_" + tag] = np.full(5, NAN) if bas is None else np.asarray(bas, dtype=np.float64)
        out["all_" + tag] = np.asarray(call(O.AllCoords, int_team=team), dtype=np.float64)
        out["allp_" + tag] = np.asarray(call(O.AllCoords_WithPrior, int_team=team), dtype=np.float64)
    last = env.lstRobots[-1]
    out["basic_last"] = np.asarray(call(O.PosBall_BasicLidar, obj_robot=last), dtype=np.float64)
    out["v1_last_negball"] = np.asarray(call(O.SingleBall_6wayLidar, obj_robot=last, obj_ball=env.lstBalls[-1]), dtype=np.float64)
    return out


def gen_mix(R, preset):
    sk, O, base, c = R.sk, R.obs, R.base, R.const
    # The env constructors place sprites with Python's global `random`, and until the first step AllCoords_WithPrior reports the
    # stale pre-placement rectDblPriorStep of that construction (alt0_allp_*, row 0 of each mixin stack): seeded here, so that the
    # part reproduces its file whatever ran before it in the process (row 0 used to depend on the earlier parts).
    random.seed(90210 + (1 if preset == "G" else 0))
    cnt = Counters(R)  # also turns the reference's endless GAME_MODE warning loop into an exception
    R.envs.SimpleDuel3()  # first constructed env fixes the shared class-level observation_space (Obs:30-37)

    class MixA(sk.DontDriveInGoals, sk.KeepMovingGuys, sk.PushNegBallsFromGoal, sk.BaseDestruction,
               sk.PushPosBallsToGoal, sk.ChasePosBall, O.SingleBall_6wayLidar, base.GameEnv_Simple):
        pass

    class MixB(sk.KeepMovingGuys, sk.NaughtyBots, sk.DontDriveInGoals, O.SingleBall_6wayLidar_v2, base.GameEnv_Simple):
        pass

    progs = {"A": ["DontDriveInGoals", "KeepMovingGuys", "PushNegBallsFromGoal", "BaseDestruction", "PushPosBallsToGoal",
                   "ChasePosBall"],
             "B": ["KeepMovingGuys", "NaughtyBots", "DontDriveInGoals"]}
    rng = random.Random(4242 if preset == "G" else 4343)
    W, H = c.ARENA_WIDTH, c.ARENA_HEIGHT
    eps = []
    for which, cls in (("A", MixA), ("B", MixB)):
        env = cls()
        NR, NB = len(env.lstRobots), len(env.lstBalls)
        for e in range(4 if preset == "G" else 5):
            random.seed(rng.randint(0, 1 << 30))
            env.reset()
            # lure the chasers into the goal corners: park the balls inside the goal triangles
            spots = [(W - 70, H - 60), (60, 70), (W - 40, H - 150), (150, 40), (W - 150, H - 40), (40, 150), (W - 100, H - 100), (90, 90)]
            for i, b in enumerate(env.lstBalls):
                if e % 2 == 0:
                    b.rectDbl.center = spots[(i + e) % len(spots)]
            nsteps = 160 if preset == "G" else 260
            S = dict(robots=[], robots_i=[], balls=[], inner=[], step=[])
            rec = {k: [] for k in ("actions", "reward", "reward_g", "done", "v1_h", "v1_g", "basic_h", "basic_g", "all_h", "all_g",
                                   "allp_h", "allp_g", "basic_last", "v1_last_negball")}

            def put():
                d = dump_state(env, R)
                for k in S:
                    S[k].append(d[k])
            put()
            a0 = alt_observations(R, env)
            length = 0
            for s in range(nsteps):
                if s % 40 < 6:
                    acts = [8 if False else rng.choice([0, 1])] * NR if s % 80 < 3 else [rng.randint(0, 7) for _ in range(NR)]
                else:
                    acts = policy_actions(env, R, "chase", rng, NR)
                if s % 50 == 49 and NR > 1:
                    acts = acts[:1]  # only robot 0 gets a new action; the others keep their thrust
                try:
                    cnt.warns = 0
                    o, r, d, info = env.step(acts)
                except Exception:
                    break
                rec["actions"].append(list(acts) + [-1] * (NR - len(acts)))
                rec["reward"].append(r)
                rec["reward_g"].append(info.dblGrumpyScore)
                rec["done"].append(d)
                ao = alt_observations(R, env)
                for k, v in ao.items():
                    rec[k].append(v)
                put()
                length += 1
                if d:
                    break
            ep = {"state_" + k: np.array(v) for k, v in S.items()}
            ep.update({k: np.array(v) for k, v in rec.items()})
            ep["length"] = length
            ep["which"] = which
            ep["alt0"] = a0
            eps.append(ep)
            print(preset, "mix", which, "len", length, "sum reward", float(np.sum(rec["reward"])), flush=True)
    smax = max(e["length"] for e in eps)
    out = {}
    for k in eps[0]:
        if k in ("length", "which", "alt0"):
            continue
        arrs = []
        for e in eps:
            a = np.asarray(e[k])
            tgt = smax + 1 if k.startswith("state_") else smax
            padded = np.zeros((tgt,) + a.shape[1:], a.dtype)
            padded[:a.shape[0]] = a
            arrs.append(padded)
        out[k] = np.stack(arrs)
    out["length"] = np.array([e["length"] for e in eps], np.int32)
    out["which"] = np.array([0 if e["which"] == "A" else 1 for e in eps], np.int32)
    for k in eps[0]["alt0"]:
        out["alt0_" + k] = np.stack([e["alt0"][k] for e in eps])
    meta = dict(META, preset=preset, programs_mro=progs, programs_exec={k: exec_order(v) for k, v in progs.items()},
                keeper_ids=KEEPER_IDS, note="which: 0 = MixA, 1 = MixB; alt observations come from the other observer mixins "
                "called as unbound methods on the same env state",
                invocation=f"python oracle/refgen/gen_golden.py {preset} mix   (one fresh process per preset; the part seeds "
                           "Python's global random before it constructs its envs)")
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(GOLD, f"mix_{preset}.npz"), **out)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which == "all":
        for p in ("T", "G", "D", "X"):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), p])
        return
    from load_reference import load_reference
    R = load_reference(which)
    os.makedirs(GOLD, exist_ok=True)
    parts = sys.argv[2:] or ["kat", "reset", "traj"]
    if "kat" in parts:
        gen_kat(R, which)
    if "reset" in parts:
        gen_reset(R, which)
    if "traj" in parts:
        gen_traj(R, which)
    if "mix" in parts:
        gen_mix(R, which)
    if "thrust" in parts:
        gen_thrust(R, which)


if __name__ == "__main__":
    main()
