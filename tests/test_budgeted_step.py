"""The budgeted step (rr_sim.hpp: ParkCtx, park_save / park_load; include/roborugby_amd.h: step_budget_clocks).

An arena whose step is over the budget at the end of a physics sub-step -- or between two passes of a sub-step's resolve loop --
parks there and the next call resumes it, ignoring the action it is given.  The promise: every arena's trajectory, as a function of the actions it ACCEPTED, is the synchronous
mode's bit for bit.  Checked here on the kernel's own phase source (host-emulated wave), which parks at pseudo-random
sub-step boundaries -- quiet ones included, at every boundary with park_mod = 1 -- with everything but the persistent record
overwritten with garbage before each call (what a GPU launch starts from: load_record + derive into an LDS slice that holds
another arena's leftovers), on the contact-dense / stuck fixtures whose steps freeze islands, thaw them, carry them across
steps, auto-reset and fault.  tests/test_gpu_budget.py repeats it through the C-ABI on the MI355X."""
import os

import numpy as np
import pytest

import adversarial as adv
import emu_lib as el

HERE = os.path.dirname(os.path.abspath(__file__))


def _record(env, r):
    st = env.get_state()
    return (r["obs"].tobytes(), r["obs_g"].tobytes(), r["reward"], r["reward_g"], r["done"], r["status"], r["naughty"],
            st["robots"].tobytes(), st["robots_i"].tobytes(), st["balls"].tobytes(), st["step"])


def _sync(preset, state, acts, narrow=False, poses=None, **kw):
    env = el.EmuEnv(preset, narrow=narrow, **kw)
    env.set_poses(*poses) if poses is not None else env.set_state(*state)
    return [_record(env, env.step(a)) for a in acts]


def _budgeted(preset, state, acts, park_mod, seed, narrow=False, poses=None, **kw):
    """feeds acts[k] to the k-th step the arena ACCEPTS; while a step is parked the calls carry a different action (ignored)."""
    env = el.EmuEnv(preset, narrow=narrow, **kw)
    env.set_poses(*poses) if poses is not None else env.set_state(*state)
    env.park_seed(seed)
    out, calls, parked = [], 0, 0
    for a in acts:
        r = env.step_budget(a, park_mod)
        calls += 1
        while r is None:
            parked += 1
            r = env.step_budget((np.asarray(a) + 3) % 8, park_mod)  # a parked step ignores the new action
            calls += 1
            assert calls < 200 * len(acts)  # (every sub-step boundary and every resolve pass can park: up to ~130 calls per step)
        out.append(_record(env, r))
    return out, parked


@pytest.fixture(autouse=True)
def _scrub():
    el.lib().emu_debug_scrub(1)
    yield
    el.lib().emu_debug_scrub(0)


@pytest.mark.parametrize("preset,n,narrow", [("T", 120, False), ("T", 60, True), ("G", 40, False), ("G", 30, True), ("D", 60, False), ("D", 40, True),
                                                  ("X", 30, True), ("Y", 40, True)])
def test_budgeted_equals_synchronous_on_contact_dense_states(preset, n, narrow):
    robots, balls, actions = adv.make_states(preset, n, seed=11 + int(narrow))
    parked = 0
    for a in range(n):
        rng = np.random.RandomState(a)
        acts = [actions[a] if rng.rand() < 0.6 else rng.randint(0, 8, actions[a].shape).astype(np.int32) for _ in range(4)]
        ref = _sync(preset, None, acts, narrow, poses=(robots[a], balls[a]))
        for mod in (1, 4):
            got, p = _budgeted(preset, None, acts, mod, 977 * a + mod, narrow, poses=(robots[a], balls[a]))
            assert got == ref, (preset, a, mod)
            parked += p
    assert parked > 10 * n  # park_mod = 1 parks at all eleven inner boundaries of every step


@pytest.mark.parametrize("preset", ["T", "G"])
def test_budgeted_equals_synchronous_on_stuck_arenas_with_carried_islands_and_auto_reset(preset):
    """The stuck arenas of the chase rollouts: islands freeze, are carried across steps, thaw when the action changes; the episode
    clock is set so that some arenas finish (done, then the auto-reset call) inside the eight steps."""
    d = np.load(os.path.join(HERE, "data", f"stuck_chase_{preset}.npz"))
    n = len(d["step"]) if preset == "T" else 20
    game_len = 300 if preset == "T" else 4500
    parked = 0
    for a in range(n):
        rng = np.random.RandomState(300 + a)
        step0 = int(d["step"][a]) if a % 4 else game_len - 3  # every fourth arena: done after three steps, then re-placed
        state = (d["robots"][a], d["robots_i"][a], d["balls"][a], step0)
        acts = [d["actions"][a] if rng.rand() < 0.75 else rng.randint(0, 8, d["actions"][a].shape).astype(np.int32) for _ in range(8)]
        ref = _sync(preset, state, acts, time_limit=1, auto_reset=1, reset_on_fault=1)
        for mod in (1, 2, 5):
            got, p = _budgeted(preset, state, acts, mod, 31 * a + mod, time_limit=1, auto_reset=1, reset_on_fault=1)
            assert got == ref, (preset, a, mod)
            parked += p
    assert parked > 20 * n


@pytest.mark.parametrize("fixture", ["stuck_islands_G.npz", "stuck_islands_wall_G.npz"])
def test_budgeted_equals_synchronous_on_frozen_islands_G(fixture):
    d = np.load(os.path.join(HERE, "data", fixture))
    for a in range(0, len(d["step"]), 2):
        state = (d["robots"][a], d["robots_i"][a], d["balls"][a], int(d["step"][a]))
        acts = [d["actions"][a]] * 3
        ref = _sync("G", state, acts)
        for mod in (1, 3):
            got, _ = _budgeted("G", state, acts, mod, 7 * a + mod)
            assert got == ref, (fixture, a, mod)


def test_parked_step_is_dropped_by_a_state_rewrite_and_survives_nothing_else():
    """rr_set_state / reset clear the parked mark (the record's fzp word): the next call starts a fresh step."""
    d = np.load(os.path.join(HERE, "data", "stuck_chase_T.npz"))
    state = (d["robots"][0], d["robots_i"][0], d["balls"][0], int(d["step"][0]))
    env = el.EmuEnv("T")
    env.set_state(*state)
    env.park_seed(1)
    assert env.step_budget(d["actions"][0], 1) is None          # parked after the first sub-step
    env.set_state(*state)                                        # rewritten from outside: the parked step is gone
    r = None
    calls = 0
    while r is None:
        r = env.step_budget(d["actions"][0], 1)
        calls += 1
    assert calls >= 12                                           # a whole step again: it parks at each of the eleven inner boundaries
                                                                 # (and between the passes of its resolve loops), then the result
    ref = el.EmuEnv("T")
    ref.set_state(*state)
    assert _record(ref, ref.step(d["actions"][0])) == _record(env, r)


@pytest.mark.parametrize("preset,goal", [("G", False), ("G", True), ("D", False), ("X", True)])
def test_budgeted_equals_synchronous_with_another_keeper_program_and_goal_scoring(preset, goal):
    """The other score keepers run as a program around the step (on_step_begin copies before it, the program after it:
    rr_extras.hpp).  Under a budget the copies belong to the call in which the arena's step BEGAN and the program (and the goal
    frame) to the call that completes it: a keeper stack that reads the prior-step copies (KeepMovingGuys) and one that ends in
    NaughtyBots, with and without the goal-scoring mode, parked at every boundary / at random ones == synchronous."""
    n = 30
    robots, balls, actions = adv.make_states(preset, n, seed=5)
    prog = [3, 2, 5, 4, 1]  # execution order: PushPos, Chase, KeepMoving, DontDrive, Naughty
    parked = 0

    def run(budget_mod, a, acts, seed=0):
        env = el.EmuEnv(preset, time_limit=1, auto_reset=1)
        env.set_poses(robots[a], balls[a])
        env.set_program(prog)
        if goal:
            env.set_goal_scoring(True)
        if budget_mod is None:
            return [_record(env, env.step(x)) + (tuple(env.goal_scores()),) for x in acts], 0
        env.park_seed(seed)
        out, p = [], 0
        for x in acts:
            r = env.step_budget(x, budget_mod)
            while r is None:
                p += 1
                r = env.step_budget((np.asarray(x) + 3) % 8, budget_mod)
            out.append(_record(env, r) + (tuple(env.goal_scores()),))
        return out, p

    for a in range(n):
        rng = np.random.RandomState(a)
        acts = [actions[a] if rng.rand() < 0.6 else rng.randint(0, 8, actions[a].shape).astype(np.int32) for _ in range(4)]
        ref, _ = run(None, a, acts)
        for mod in (1, 3):
            got, p = run(mod, a, acts, 31 * a + mod)
            assert got == ref, (preset, goal, a, mod)
            parked += p
    assert parked > 10 * n
