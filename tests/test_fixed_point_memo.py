"""The two exact shortcuts for stuck arenas (rr_sim.hpp: step_arena, substep):
  * whole-arena fixed point -- an expensive sub-step leaves the arena bit-identical to the sub-step before it: the
    remaining sub-steps are skipped;
  * island freeze -- the entities that took part in the hits of an expensive sub-step are bit-identical to a sub-step
    earlier: they are frozen (an obstacle with its recorded excursion) while the rest of the arena keeps stepping, and
    thawed the moment anything outside comes within a broad-phase bound of them.
Both must be EXACT: the kernel's phase source (host-emulated wave) with the shortcuts on and off has to produce
bit-identical states, observations, rewards and status words over multi-step rollouts of stuck / contact-dense arenas
-- and the shortcuts (freeze, both thaw points, fixed point) have to actually fire on them."""
import io
import os
import sys

import numpy as np
import pytest

import adversarial as adv
import emu_lib as el

HERE = os.path.dirname(os.path.abspath(__file__))


def _rollout(preset, robots, balls, actions, memo, steps, narrow=False, state=None):
    el.lib().emu_debug_memo(int(memo))
    try:
        env = el.EmuEnv(preset, narrow=narrow)
        if state is not None:
            env.set_state(*state)
        else:
            env.set_poses(robots, balls)
        out = []
        for k in range(steps):
            r = env.step(actions)  # same action every step: a robot keeps pushing
            st = env.get_state()
            out.append((r["obs"].tobytes(), r["obs_g"].tobytes(), r["reward"], r["reward_g"], r["done"], r["status"], r["naughty"],
                        st["robots"].tobytes(), st["robots_i"].tobytes(), st["balls"].tobytes(), st["step"]))
        return out
    finally:
        el.lib().emu_debug_memo(1)


def _count_events(fn, words=("fixed point",)):
    """runs fn() with the emulation trace on and counts trace lines on stderr (captured at fd level into a file)."""
    import tempfile
    sys.stderr.flush()
    saved = os.dup(2)
    tf = tempfile.TemporaryFile()
    os.dup2(tf.fileno(), 2)
    el.lib().emu_debug_trace(1)
    try:
        fn()
    finally:
        el.lib().emu_debug_trace(0)
        sys.stderr.flush()
        os.dup2(saved, 2)
        os.close(saved)
    tf.seek(0)
    out = tf.read()
    tf.close()
    return {w: out.count(w.encode()) for w in words}


def _count_fixed_points(fn):
    return _count_events(fn)["fixed point"]


@pytest.mark.parametrize("preset,n,narrow", [("T", 240, False), ("T", 120, True), ("G", 60, False), ("G", 60, True), ("X", 60, True), ("Y", 80, True)])
def test_shortcut_is_bit_exact_on_contact_dense_rollouts(preset, n, narrow):
    robots, balls, actions = adv.make_states(preset, n, seed=5 + int(narrow))
    for a in range(n):
        on = _rollout(preset, robots[a], balls[a], actions[a], True, 4, narrow)
        off = _rollout(preset, robots[a], balls[a], actions[a], False, 4, narrow)
        assert on == off, (preset, a)


def test_shortcut_fires_on_a_stuck_arena_T():
    # robot driving a ball into the left wall: every sub-step pushes, exhausts the resolve loop and is undone
    W = 600.0
    robots = np.array([[30.0, 300.0, 180.0]])   # facing -x, hugging the wall region
    balls = np.array([[9.0, 300.0, 0.0, 0.0]])
    act = np.array([0], np.int32)                # forward
    on = _rollout("T", robots, balls, act, True, 6)
    off = _rollout("T", robots, balls, act, False, 6)
    assert on == off
    fired = _count_fixed_points(lambda: _rollout("T", robots, balls, act, True, 6))
    assert fired >= 1, "the stuck arena never reached the fixed-point shortcut"


def test_shortcut_exact_on_the_squeezed_fixture_G():
    d = np.load(os.path.join(HERE, "..", "tools", "fixtures", "squeezed_G.npz"))
    state = (d["robots"], d["robots_i"], d["balls"], int(d["step"]))
    on = _rollout("G", None, None, d["actions"], True, 5, state=state)
    off = _rollout("G", None, None, d["actions"], False, 5, state=state)
    assert on == off


@pytest.mark.parametrize("fixture", ["stuck_islands_G.npz", "stuck_islands_wall_G.npz"])
def test_island_freeze_exact_on_stuck_islands_from_a_rollout_G(fixture):
    """Arenas taken from the slowest wavefronts of a 65,536-arena random rollout on the MI355X -- a ball squeezed between
    robots, robots pushing a ball into each other (step ~30-140), and, late in the episode (step ~2,800), robots within a
    few pixels of a wall pushing a ball into it: the freeze fires on them and changes nothing."""
    d = np.load(os.path.join(HERE, "data", fixture))
    fired = {"freeze": 0, "thaw in phase 1": 0}
    for a in range(len(d["step"])):
        state = (d["robots"][a], d["robots_i"][a], d["balls"][a], int(d["step"][a]))
        on = _rollout("G", None, None, d["actions"][a], True, 3, state=state)
        off = _rollout("G", None, None, d["actions"][a], False, 3, state=state)
        assert on == off, a
        c = _count_events(lambda: _rollout("G", None, None, d["actions"][a], True, 3, state=state), tuple(fired))
        for k in fired:
            fired[k] += c[k]
    # (the parity build also wants the balls its scratch-rect chain starts from to stand still: fewer islands qualify)
    assert fired["freeze"] >= (12 if el.DEFAULT_EXACT else 20), fired


def test_island_thaw_paths_are_exact_G():
    """The squeezed fixture with the other balls sent rolling at the island, at each other and at the walls, and the free
    robots driven at random: the island must thaw in phase 1 (a robot / ball comes close before the roll) and in phase 2
    (something turns up after the roll) and the results must not change by a bit."""
    d = np.load(os.path.join(HERE, "..", "tools", "fixtures", "squeezed_G.npz"))
    rng = np.random.RandomState(3)
    isl = np.array([611.7, 324.6])
    fired = {"freeze": 0, "thaw in phase 1": 0, "thaw in phase 2": 0, "witness 1": 0, "witness 2": 0, "witness 3": 0, "): hit": 0}
    for t in range(160):
        balls = d["balls"].copy()
        mode = t % 4
        for b in (0, 1, 2, 4, 5, 6, 7):
            if mode in (0, 1) and rng.rand() < 0.5:    # a free ball heading for / passing by the island
                ang, dist = rng.uniform(0, 2 * np.pi), rng.uniform(20, 90)
                pos = isl + dist * np.array([np.cos(ang), np.sin(ang)])
                v = -(pos - isl) / dist * rng.uniform(0.5, 6) + rng.uniform(-1, 1, 2)
            elif mode == 2 and rng.rand() < 0.5:        # free balls at the walls, far from the island
                pos = np.array([rng.choice([rng.uniform(8, 20), rng.uniform(780, 792)]), rng.uniform(20, 780)])
                v = rng.uniform(-5, 5, 2)
            elif mode == 3 and b in (0, 1):             # two free balls running into each other
                pos, v = np.array([200.0 + 10 * b, 600.0]), np.array([3.0 * (1 - 2 * b), 0.0])
            else:
                pos, v = balls[b, :2], rng.choice([0.0, 1.0]) * rng.uniform(-3, 3, 2)
            balls[b] = [pos[0], pos[1], pos[0] - 7, pos[0] + 7, pos[1] - 7, pos[1] + 7, v[0], v[1]]
        act = d["actions"].copy()
        act[:2] = rng.randint(0, 8, 2)
        state = (d["robots"], d["robots_i"], balls, int(d["step"]))
        on = _rollout("G", None, None, act, True, 2, state=state)
        off = _rollout("G", None, None, act, False, 2, state=state)
        assert on == off, (t, mode)
        c = _count_events(lambda: _rollout("G", None, None, act, True, 2, state=state), tuple(fired))
        for k in fired:
            fired[k] += c[k]
    # (default build: a ball that merely passes an island robot no longer thaws the island -- the narrow "witness" tests of substep()
    # decide: they run, most of them come out clear, some hit and thaw; the parity build thaws on the bound as before)
    assert fired["freeze"] > 100 and fired["thaw in phase 1"] > (20 if el.DEFAULT_EXACT else 15) and fired["thaw in phase 2"] > 10, fired
    if not el.DEFAULT_EXACT:
        assert fired["witness 1"] > 20 and fired["witness 2"] > 20 and fired["witness 3"] > 10 and fired["): hit"] > 3, fired


def _rollout_seq(preset, state, acts, memo, poke=None):
    """like _rollout, with one action per step; poke = (step index, fn(env)) mutates the env from outside before that step."""
    el.lib().emu_debug_memo(int(memo))
    try:
        env = el.EmuEnv(preset, time_limit=1, auto_reset=1)
        env.set_state(*state)
        out = []
        for k, a in enumerate(acts):
            if poke is not None and poke[0] == k:
                poke[1](env)
            r = env.step(a)
            st = env.get_state()
            out.append((r["obs"].tobytes(), r["obs_g"].tobytes(), r["reward"], r["reward_g"], r["done"], r["status"], r["naughty"],
                        st["robots"].tobytes(), st["robots_i"].tobytes(), st["balls"].tobytes(), st["step"]))
        return out
    finally:
        el.lib().emu_debug_memo(1)


@pytest.mark.parametrize("preset", ["T", "G"])
def test_freeze_carried_across_steps_is_exact_on_stuck_arenas_of_a_chase_rollout(preset):
    """Arenas of the slowest wavefronts of a chase-policy rollout on the MI355X (tools/chase_monsters.py, stuck_fixtures.py):
    a robot driving a ball into a wall (T: a one-robot arena whose robot edges keep drifting in the last bits -- frozen as an
    island, not as a whole-arena fixed point), ball clusters in front of a robot (G).  Eight steps each: the robot keeps its
    action most of the time (the island the step ended with is carried into the next one), changes it now and then (the island
    is recomputed), and once the state is rewritten from outside (nothing may be carried over).  Shortcuts on == off, bit for bit."""
    d = np.load(os.path.join(HERE, "data", f"stuck_chase_{preset}.npz"))
    n = len(d["step"]) if preset == "T" else 24
    fired = {"step begins frozen": 0, "freeze": 0, "thaw": 0}
    for a in range(n):
        rng = np.random.RandomState(100 + a)
        state = (d["robots"][a], d["robots_i"][a], d["balls"][a], int(d["step"][a]))
        acts = [d["actions"][a] if rng.rand() < 0.75 else rng.randint(0, 8, d["actions"][a].shape).astype(np.int32) for _ in range(8)]

        def rewrite(env):  # rr_set_state with the state it has: the same physics, but nothing frozen may survive it
            s = env.get_state()
            env.set_state(s["robots"], s["robots_i"], s["balls"], s["step"])
        poke = (5, rewrite) if a % 3 == 0 else None
        on = _rollout_seq(preset, state, acts, True, poke)
        off = _rollout_seq(preset, state, acts, False, poke)
        assert on == off, (preset, a)
        c = _count_events(lambda: _rollout_seq(preset, state, acts, True, poke), tuple(fired))
        for k in fired:
            fired[k] += c[k]
    assert fired["step begins frozen"] >= (150 if preset == "T" else 5) and fired["freeze"] >= (40 if preset == "T" else 5), fired


def test_resting_neighbours_join_the_island_on_slow_random_policy_arenas():
    """tests/data/stuck_random_G.npz: the busiest arena of the slowest wavefront of 40 consecutive random-policy launches at the steady
    state (MI355X, round 4).  Several of them are a robot squeezing a ball against a wall with a SECOND ball resting half a pixel off
    its flank: that ball fired the frozen variant's ball-robot bound in every sub-step -- freeze, thaw, twelve expensive sub-steps
    per step.  A ball at rest, untouched and unchanged, within the bound of an island robot now joins the island (rr_sim.hpp:
    resting_neighbours).  Shortcuts on == off bit for bit over six steps (actions kept, then changed), and the ping-pong is gone."""
    d = np.load(os.path.join(HERE, "data", "stuck_random_G.npz"))
    n = len(d["step"])
    ev = {"E freeze": 0, "thaw in phase": 0, "resolve gave up": 0}
    joined = 0
    for a in range(n):
        rng = np.random.RandomState(7 + a)
        state = (d["robots"][a], d["robots_i"][a], d["balls"][a], int(d["step"][a]))
        acts = [d["actions"][a]] * 3 + [rng.randint(0, 8, d["actions"][a].shape).astype(np.int32) for _ in range(3)]
        on = _rollout_seq("G", state, acts, True)
        off = _rollout_seq("G", state, acts, False)
        assert on == off, a
        if d["work"][a] > 150:  # the ~1 ms arenas
            c = _count_events(lambda: _rollout_seq("G", state, acts[:3], True), tuple(ev) + ("(hit",))
            for k in ev:
                ev[k] += c[k]
    # (with every shortcut but this one the three monsters of the fixture gave up 11-12 resolve loops per step: > 130 in 4 x 3 steps)
    assert ev["E freeze"] >= 4 and ev["resolve gave up"] < 80, ev


def test_witness_tests_are_exact_with_balls_drifting_past_frozen_islands():
    """substep()'s witness tests (default build): a free ball is put next to the robot of a stuck island -- 11 to 26 px from its centre in
    every direction, i.e. from overlapping its flank to just outside the frozen variant's bounds -- and sent drifting at 0 to 0.6 px per
    sub-step; the island's robot keeps its action (also turning ones: the put-back then changes the rotation and the restore has to
    rebuild the corner offsets).  Shortcuts on == off bit for bit over three steps; every witness outcome occurs."""
    words = ("witness 1 (before the roll, robots moved): hit", "witness 1 (before the roll, robots moved): clear",
             "witness 2 (after the roll, robots moved): hit", "witness 2 (after the roll, robots moved): clear",
             "witness 3 (after the roll, robots put back): hit", "witness 3 (after the roll, robots put back): clear")
    tot = {w: 0 for w in words}
    rng = np.random.RandomState(11)
    cases = 0
    for name in ("stuck_random_G.npz", "stuck_chase_G.npz"):
        d = np.load(os.path.join(HERE, "data", name))
        for a in range(len(d["step"])):
            robots, balls = d["robots"][a], d["balls"][a].copy()
            # the robot that is closest to a ball: the island's
            dist = np.hypot(balls[:, None, 0] - robots[None, :, 0], balls[:, None, 1] - robots[None, :, 1])
            b0, r0 = np.unravel_index(np.argmin(dist), dist.shape)
            free = [b for b in range(balls.shape[0]) if dist[b].min() > 60]
            if not free:
                continue
            for rep in range(3):
                b = free[rep % len(free)]
                ang, rad = rng.uniform(0, 2 * np.pi), rng.uniform(11, 26)
                pos = robots[r0, :2] + rad * np.array([np.cos(ang), np.sin(ang)])
                if np.hypot(*(pos - balls[b0, :2])) < 14.5 or not (8 < pos[0] < 792 and 8 < pos[1] < 792):
                    continue
                v = rng.uniform(-0.6, 0.6, 2) * rng.choice([0.0, 1.0, 1.0])
                bl = balls.copy()
                bl[b] = [pos[0], pos[1], pos[0] - 7, pos[0] + 7, pos[1] - 7, pos[1] + 7, v[0], v[1]]
                act = d["actions"][a].copy()
                if rep == 2:
                    act[r0] = rng.choice([2, 3, 4, 5, 6, 7])  # a turning action for the island's robot
                state = (robots, d["robots_i"][a], bl, int(d["step"][a]))
                on = _rollout("G", None, None, act, True, 3, state=state)
                off = _rollout("G", None, None, act, False, 3, state=state)
                assert on == off, (name, a, rep)
                cases += 1
                if not el.DEFAULT_EXACT and cases % 2 == 0:
                    c = _count_events(lambda: _rollout("G", None, None, act, True, 3, state=state), words)
                    for w in words:
                        tot[w] += c[w]
    assert cases > 120, cases
    if not el.DEFAULT_EXACT:
        assert all(tot[w] > 0 for w in words[1:]), tot
