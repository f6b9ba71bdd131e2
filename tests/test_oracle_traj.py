"""Oracle step() vs full-state trajectories captured from the reference (tests/golden/traj_*.npz).

The oracle is started from the reference's dumped internal state at step 0 of every episode and then
free-runs on the recorded actions; state, observation, reward, done and the naughty set must equal the
reference's at EVERY step.  Bar: bit-exact."""
import json

import numpy as np
import pytest

import oracle_lib as ol


def _eq(a, b):
    return np.array_equal(a, b, equal_nan=True)


def _check_episode(preset, t, ep, resync=False):
    env = ol.OracleEnv(preset)
    n = int(t["length"][ep])
    S = {k: t["state_" + k][ep] for k in ("robots", "robots_i", "balls", "inner", "step")}
    env.set_state(S["robots"][0], S["robots_i"][0], S["balls"][0], S["inner"][0], S["step"][0])
    if "obs0" in t.files:
        o0 = env.observe(1)
        assert _eq(o0, t["obs0"][ep])
        og = env.observe(-1)
        if og is None:
            assert np.isnan(t["obs0_g"][ep]).all()
        else:
            assert _eq(og, t["obs0_g"][ep])
    for s in range(n):
        if resync:
            env.set_state(S["robots"][s], S["robots_i"][s], S["balls"][s], S["inner"][s], S["step"][s])
        r = _drive(env, t, ep, s)
        st = env.get_state()
        ctx = (preset, ep, s)
        assert _eq(st["robots"], S["robots"][s + 1]), (ctx, st["robots"] - S["robots"][s + 1])
        assert _eq(st["robots_i"], S["robots_i"][s + 1]), ctx
        assert _eq(st["balls"], S["balls"][s + 1]), (ctx, st["balls"] - S["balls"][s + 1])
        assert _eq(st["inner"], S["inner"][s + 1]), (ctx, st["inner"], S["inner"][s + 1])
        assert st["step"] == S["step"][s + 1]
        assert _eq(r["obs"], t["obs"][ep, s]), (ctx, r["obs"] - t["obs"][ep, s])
        assert _eq(r["obs_g"], t["obs_g"][ep, s]), (ctx, r["obs_g"], t["obs_g"][ep, s])
        assert r["reward"] == t["reward"][ep, s], ctx
        assert r["reward_g"] == t["reward_g"][ep, s], ctx
        assert r["done"] == bool(t["done"][ep, s]), ctx
        assert r["naughty"] == t["naughty"][ep, s], ctx
        # status: warning bit only when the reference printed its warning
        warned = int(t["warn"][ep, s]) > 0 if "warn" in t.files else False
        assert bool(r["status"] & 256) == warned, (ctx, r["status"])
        assert (r["status"] & ~256) == 0, (ctx, r["status"])
    exc = int(t["exc"][ep])
    if exc:
        # the reference raised inside step n: the oracle must flag the same fault on that step
        r = _drive(env, t, ep, n)
        assert r["status"] & exc, (preset, ep, n, r["status"], exc)
    return n


def _drive(env, t, ep, s):
    """One step with the recorded input: Direction actions (GameEnv_Simple.step) or (L, R) thrust pairs (GameEnv.step)."""
    if "thrust" in t.files:
        th = t["thrust"][ep, s]
        return env.step_thrust(th[~np.isnan(th[:, 0])])
    acts = t["actions"][ep, s]
    return env.step(acts[acts >= 0])


@pytest.mark.parametrize("preset", ["T", "G", "D", "X"])
def test_trajectories_bit_exact(golden_dir, preset):
    t = np.load(f"{golden_dir}/traj_{preset}.npz")
    total = 0
    for ep in range(t["length"].shape[0]):
        total += _check_episode(preset, t, ep)
    assert total > 1000
    cov = json.loads(str(t["meta"]))["coverage"]
    # the fixtures really exercise every response path of the hot loop
    for k in ("apply_force_to_ball", "bounce_ball_off_bot", "bounce_ball_off_wall", "undo_naughty"):
        assert cov[k] > 0, k
    if preset in ("G", "X"):
        assert cov["bounce_balls"] > 0 and cov["robot_collision"] > 0


@pytest.mark.parametrize("preset", ["T", "G", "D", "X"])
def test_thrust_entry_bit_exact(golden_dir, preset):
    """The continuous entry GameEnv.step(env, [(L, R), ...]) (RR_EnvBase.py:260-273) with Robot.set_thrust's
    int(round(x)) (RR_Robot.py:100-102), driven on the imported reference with the half-way cases +-0.5, +-1.5,
    +-2.5 and 0.49 / 0.51 (tests/golden/thrust_*.npz): the oracle's step_thrust free-runs every episode bit for bit."""
    t = np.load(f"{golden_dir}/thrust_{preset}.npz")
    total = sum(_check_episode(preset, t, ep) for ep in range(t["length"].shape[0]))
    assert total > 500
    th = t["thrust"][~np.isnan(t["thrust"])]
    for v in (0.5, -0.5, 1.5, -1.5, 2.5, -2.5, 0.49, -0.49, 0.51):  # the rounding cases SURVEY 8(f)-4 names are in the fixture
        assert (th == v).any(), v
    assert {-2.0, -1.0, 0.0, 1.0, 2.0} <= set(np.unique(np.rint(th)))  # thrust magnitudes 0, 1 and 2 (and 3) reach the kinematics
    cov = json.loads(str(t["meta"]))["coverage"]
    assert cov["apply_force_to_ball"] > 0 and cov["bounce_ball_off_bot"] > 0 and cov["bounce_ball_off_wall"] > 0


@pytest.mark.parametrize("preset", ["T", "G", "D", "X"])
def test_done_flag_is_step_counter(golden_dir, preset):
    """done = lngStepCount > GAME_LENGTH_STEPS (RR_EnvBase.py:555-559); stepping after done is flagged."""
    cfg = ol.PRESETS[preset]
    env = ol.OracleEnv(preset)
    env.reset(seed=3, arena=0, episode=0)
    st = env.get_state()
    env.set_state(st["robots"], st["robots_i"], st["balls"], None, cfg["game_len"] - 1)
    na = env.nr
    r = env.step([0] * na)
    assert not r["done"] and env.get_state()["step"] == cfg["game_len"]
    r = env.step([0] * na)
    assert r["done"] and r["status"] == 0
    r = env.step([0] * na)
    assert r["status"] & 64 and r["done"]
    assert env.get_state()["step"] == cfg["game_len"] + 1
