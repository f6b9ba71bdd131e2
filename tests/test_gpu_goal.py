"""Opt-in goal scoring on the GPU (k_goal through the C-ABI) against the oracle's restatement of the same extension
(tests/test_goal_scoring.py has the rule-by-rule scenarios and explains why no reference vector exists for this mode)."""
import numpy as np
import pytest

import oracle_lib as ol

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _env(preset, n, **kw):
    import roborugby_amd as rr
    kw.setdefault("time_limit", False)
    kw.setdefault("auto_reset", False)
    return rr.BatchedRoboRugbyEnv(n, preset=preset, goal_scoring=True, **kw)


def test_T_batch_of_balls_in_and_around_the_goals_matches_the_oracle():
    n, steps = 256, 160
    rng = np.random.default_rng(3)
    robots = np.tile(np.array([[[300.0, 300.0, 0.0]]]), (n, 1, 1))
    balls = np.zeros((n, 1, 4))
    for a in range(n):  # around the hypotenuse of either goal triangle: some inside, some outside
        if a % 2:
            balls[a, 0, :2] = (rng.uniform(380, 590), rng.uniform(380, 590))
        else:
            balls[a, 0, :2] = (rng.uniform(10, 220), rng.uniform(10, 220))
    env = _env("T", n)
    env.set_poses(robots, balls)
    acts = torch.full((n, 1), 8, dtype=torch.int32, device="cuda")  # nobody moves
    total = torch.zeros(n, dtype=torch.float64, device="cuda")
    done_at = torch.full((n,), -1, dtype=torch.int64, device="cuda")
    obs_hist = []
    for s in range(steps):
        o, r, d, info = env.step_f64(acts)
        obs_hist.append(o.cpu().numpy())
        total += r
        done_at = torch.where((done_at < 0) & d, torch.full_like(done_at, s + 1), done_at)
    scores = env.goal_scores().cpu().numpy()
    st = env.get_state()
    consumed = 0
    for a in range(n):
        orc = ol.OracleEnv("T")
        orc.set_goal_scoring(True)
        orc.set_clean_state(robots[a], balls[a])
        tot, when = 0.0, -1
        for s in range(steps):
            res = orc.step([8])
            if res["status"] & 64:
                break
            # the observation too: the step that consumes the ball still sees it inside the goal, on both sides
            assert np.abs(res["obs"] - obs_hist[s][a]).max() < 1e-9, (a, s)
            tot += res["reward"]
            if res["done"] and when < 0:
                when = s + 1
        assert abs(tot - float(total[a])) < 1e-9 and when == int(done_at[a]), (a, tot, float(total[a]), when, int(done_at[a]))
        assert np.array_equal(orc.goal_scores(), scores[a])
        assert np.array_equal(orc.get_state()["balls"], st["balls"][a].cpu().numpy())
        consumed += when > 0
    assert 40 < consumed < n - 40  # both outcomes are exercised
    assert set(np.unique(scores[:, 0])) <= {0, 500} and set(np.unique(scores[:, 1])) <= {0, 500}


def test_G_destroyed_goal_base_destruction_and_auto_reset():
    import roborugby_amd as rr
    n = 64
    robots = np.tile(np.array([[[400, 100, 0], [400, 200, 0], [400, 300, 0], [400, 400, 0]]], dtype=np.float64), (n, 1, 1))
    one = np.array([[760, 770, 0, 0], [100, 400, 0, 0], [100, 500, 0, 0], [100, 600, 0, 0],
                    [780, 700, 0, 0], [700, 780, 0, 0], [740, 740, 0, 0], [300, 700, 0, 0]], dtype=np.float64)
    balls = np.tile(one[None], (n, 1, 1))
    balls[1::2, 6, :2] = (400, 600)  # odd arenas: only two negative balls in the goal -> not destroyed
    env = rr.BatchedRoboRugbyEnv(n, preset="G", goal_scoring=True, time_limit=True, auto_reset=True,
                                 rewards=("ChasePosBall", "PushPosBallsToGoal", "BaseDestruction"))
    env.set_poses(robots, balls)
    acts = torch.full((n, 4), 8, dtype=torch.int32, device="cuda")
    for s in range(151):
        o, r, d, info = env.step(acts)
    P = (500 + 200000) * 8
    r, d, status = r.cpu().numpy(), d.cpu().numpy(), info.status.cpu().numpy()
    assert d[0::2].all() and not d[1::2].any()
    assert ((status[0::2] & 2048) != 0).all() and ((status[1::2] & (2048 | 4096 | 8192)) == 0).all()
    assert np.allclose(r[0::2], -1000.0 + P) and np.allclose(r[1::2], 500.0 - 1000.0)
    sc = env.goal_scores().cpu().numpy()
    assert (sc[0::2] == [-1000, 0]).all() and (sc[1::2] == [-500, 0]).all()
    lr, _, ll, cnt = env.episode_stats()
    assert (cnt.cpu().numpy()[0::2] == 1).all() and (ll.cpu().numpy()[0::2] == 151).all() and np.allclose(lr.cpu().numpy()[0::2], -1000.0 + P)
    # the next call re-places the finished arenas: everything back in play, goal bookkeeping cleared
    o, r, d, info = env.step(acts)
    status = info.status.cpu().numpy()
    assert ((status[0::2] & 1024) != 0).all() and ((status[1::2] & 1024) == 0).all()
    st = env.get_state()["balls"].cpu().numpy()
    assert (st[0::2, :, 0] > 0).all() and (st[1::2, [0, 4, 5], 0] < -900).all()
    sc = env.goal_scores().cpu().numpy()
    assert (sc[0::2] == 0).all() and (sc[1::2] == [-500, 0]).all()


def test_goal_scoring_off_is_the_default_and_scores_stay_zero():
    import roborugby_amd as rr
    env = rr.BatchedRoboRugbyEnv(8, preset="T", time_limit=False, auto_reset=False)
    env.set_poses(np.tile([[[100.0, 100.0, 0.0]]], (8, 1, 1)), np.tile([[[560.0, 570.0, 0.0, 0.0]]], (8, 1, 1)))
    acts = torch.full((8, 1), 8, dtype=torch.int32, device="cuda")
    for _ in range(160):
        o, r, d, info = env.step(acts)
    assert not bool(d.any()) and float(r.abs().max()) == 0.0 and int(env.goal_scores().abs().max()) == 0
    assert float(env.get_state()["balls"][:, 0, 0].min()) == 560.0
