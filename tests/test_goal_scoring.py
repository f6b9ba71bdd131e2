"""Opt-in goal scoring -- an EXTENSION: the reference's goals never score on its live path (RR_Goal.py:58-91 is
only reached from the never-called GameEnv.__old_step and calls a property as a function), so there is no reference vector to
pin it to.  What is tested: the mechanism's own rules on hand-built scenarios (150 consecutive steps inside a goal triangle,
+-500 by ball colour and goal, a consumed ball is out of play, three negative balls destroy a goal, destroyed goal / empty
field ends the episode, BaseDestruction pays), and that the kernel source (host emulation here, the GPU in -m gpu) and the
oracle's restatement agree on them and on random rollouts.  With the mode OFF nothing changes (the golden tests cover that)."""
import numpy as np
import pytest

import emu_lib as el
import oracle_lib as ol


def _pair(preset, **kw):
    e, o = el.EmuEnv(preset, **kw), ol.OracleEnv(preset)
    e.set_goal_scoring(True); o.set_goal_scoring(True)
    return e, o


def _place(envs, robots_xyr, balls_xyv):
    envs[0].set_poses(robots_xyr, balls_xyv)
    envs[1].set_clean_state(robots_xyr, balls_xyv)


def _step_both(e, o, acts):
    re_, ro = e.step(acts), o.step(acts)
    se, so = e.get_state(), o.get_state()
    assert np.allclose(se["balls"], so["balls"], atol=1e-9, rtol=0), (se["balls"] - so["balls"])
    assert np.allclose(se["robots"][:, :7], so["robots"][:, :7], atol=1e-9, rtol=0)
    assert abs(re_["reward"] - ro["reward"]) < 1e-7 and abs(re_["reward_g"] - ro["reward_g"]) < 1e-7, (re_, ro)
    assert re_["done"] == ro["done"]
    assert (re_["status"] & (2048 | 4096 | 8192)) == (ro["status"] & (2048 | 4096 | 8192))
    assert np.array_equal(e.goal_scores(), o.goal_scores())
    return re_


def test_T_ball_resting_in_the_happy_goal_scores_after_150_steps_and_ends_the_game():
    e, o = _pair("T")
    # robot far away and idle (invalid action 8 keeps thrust 0); the one positive ball at rest inside the bottom-right triangle
    _place((e, o), [[100, 100, 0]], [[560, 570, 0, 0]])
    for s in range(1, 152):
        r = _step_both(e, o, [8])
        if s < 151:  # entered at step 1: consumed when frame - since >= 150, i.e. at step 151
            assert r["reward"] == 0.0 and not r["done"], s
    assert r["reward"] == 500.0 and r["reward_g"] == -500.0 and r["done"] and (r["status"] & 8192)  # no ball left
    assert list(e.goal_scores()) == [500, 0]
    st = e.get_state()
    assert st["balls"][0][0] == -1000.0 and st["balls"][0][1] == -1000.0 and (st["balls"][0][6:] == 0).all()  # parked
    r2 = e.step([8])
    assert r2["status"] & 64  # "Game is over" (auto_reset off)


def test_T_leaving_the_goal_restarts_the_count_and_the_grumpy_goal_costs_points():
    e, o = _pair("T")
    _place((e, o), [[300, 300, 0]], [[30, 40, 0, 0]])  # positive ball inside the top-left (grumpy) triangle
    for s in range(100):
        _step_both(e, o, [8])
    # knock it out and back in: re-place outside for one step, then inside again
    st = o.get_state()
    _place((e, o), [[300, 300, 0]], [[400, 400, 0, 0]])
    _step_both(e, o, [8])
    _place((e, o), [[300, 300, 0]], [[30, 40, 0, 0]])
    rewards = [_step_both(e, o, [8])["reward"] for _ in range(151)]
    assert rewards[:150] == [0.0] * 150 and rewards[150] == -500.0  # a positive ball in the grumpy goal
    assert list(e.goal_scores()) == [0, 500]


def test_G_consumed_balls_leave_play_three_negatives_destroy_a_goal_and_base_destruction_pays():
    e, o = _pair("G")
    for env in (e, o):
        env.set_program([6, 3, 2])  # execution order: BaseDestruction, PushPosBallsToGoal, ChasePosBall
    robots = [[400, 100, 0], [400, 200, 0], [400, 300, 0], [400, 400, 0]]
    # balls 4,5,6 (negative) at rest in the happy goal's corner, ball 0 (positive) there too; the rest far from everything
    balls = [[760, 770, 0, 0], [100, 400, 0, 0], [100, 500, 0, 0], [100, 600, 0, 0],
             [780, 700, 0, 0], [700, 780, 0, 0], [740, 740, 0, 0], [300, 700, 0, 0]]
    _place((e, o), robots, balls)
    for s in range(1, 152):
        r = _step_both(e, o, [8, 8, 8, 8])
        if s < 151:
            assert not r["done"] and r["reward"] == 0.0
    # one positive (+500) and three negative (-1500) balls consumed by the happy goal in the same step -> destroyed
    P = (500 + 200000) * 8
    assert r["done"] and (r["status"] & 2048) and not (r["status"] & 8192)
    assert r["reward"] == -1000.0 + P and r["reward_g"] == 1000.0 - P
    assert list(e.goal_scores()) == [500 - 1500, 0]
    st = e.get_state()["balls"]
    assert [st[b][0] for b in (0, 4, 5, 6)] == [-1000.0, -1160.0, -1200.0, -1240.0]  # parked, 40 px apart


def test_G_robot_drives_through_a_consumed_ball_and_gets_no_chase_reward_for_it():
    e, o = _pair("G")
    robots = [[300, 760, 0], [100, 100, 0], [100, 200, 0], [100, 300, 0]]
    balls = [[760, 770, 0, 0], [500, 100, 0, 0], [500, 200, 0, 0], [500, 300, 0, 0],
             [300, 400, 0, 0], [300, 500, 0, 0], [600, 400, 0, 0], [600, 500, 0, 0]]
    _place((e, o), robots, balls)
    for s in range(151):
        _step_both(e, o, [8, 8, 8, 8])
    assert list(e.goal_scores()) == [500, 0]
    # robot 0 now drives east along y = 760 straight through where ball 0 used to lie: nothing to collide with
    for s in range(40):
        r = _step_both(e, o, [0, 8, 8, 8])
        assert (r["status"] & 63) == 0
    assert e.get_state()["robots"][0][0] > 760  # it went through the spot


@pytest.mark.parametrize("preset", ["T", "G"])
def test_random_rollouts_with_goal_scoring_on_match_the_oracle(preset):
    """Balls scattered into both goal corners, robots driven at random for 200 steps: kernel source == oracle step by step."""
    rng = np.random.default_rng(5)
    cfg = ol.PRESETS[preset]
    W = cfg["W"]
    for trial in range(6 if preset == "T" else 3):
        e, o = _pair(preset)
        nr, nb = e.nr, e.nb
        robots = [[W / 2 + 60 * (i - nr / 2), W / 2, float(rng.integers(0, 360))] for i in range(nr)]
        balls = []
        for b in range(nb):
            corner = rng.integers(0, 2)
            x, y = (rng.uniform(W - 150, W - 20), rng.uniform(W - 60, W - 15)) if corner else (rng.uniform(15, 60), rng.uniform(20, 150))
            balls.append([x, y + (0 if corner else 0), 0, 0])
        # keep them apart
        for b in range(nb):
            balls[b][0] += 0.0
        _place((e, o), robots, balls)
        done = False
        for s in range(200):
            if done:
                break
            done = _step_both(e, o, rng.integers(0, 8, size=nr))["done"]
