"""Reset placement (RR_EnvBase.py:155-216) against the 1,000 layouts per preset captured from the imported reference
(tests/golden/reset_*.npz): support, rejection rule, marginals, ball occupancy, first observations.  The CPU half
checks the oracle's reset; the -m gpu half (tests/test_gpu_parity.py) runs the same checks on the kernel's."""
import numpy as np
import pytest

import oracle_lib as ol
import reset_checks as rc


def oracle_layouts(preset, arenas, episodes, seed=2024):
    """arenas x episodes layouts: every arena is constructed (episode 0) and then reset `episodes` times in a row, like the
    fixture generator did with the reference -- robots not yet re-placed still block at their previous pose."""
    o = ol.OracleEnv(preset)
    R, B = [], []
    for a in range(arenas):
        o.reset(seed, a, 0)
        for e in range(1, episodes + 1):
            st = o.reset(seed, a, e)
            assert st == 0
            s = o.get_state()
            R.append(s["robots"])
            B.append(s["balls"])
    return np.array(R), np.array(B)


@pytest.mark.parametrize("preset", ["T", "G", "D", "X"])
def test_reference_layouts_satisfy_the_oracles_rejection_rule_and_first_obs(golden_dir, preset):
    """Every layout the REFERENCE produced, rebuilt in the oracle (clean rects at the reference's x, y, rot): no int-AABB
    overlap under the oracle's rect arithmetic -- i.e. oracle and reference agree on what `spritecollide` rejects -- and
    the oracle's observation of it equals the first observation the reference's reset() returned, bit for bit."""
    t = np.load(f"{golden_dir}/reset_{preset}.npz")
    cfg = ol.PRESETS[preset]
    o = ol.OracleEnv(preset)
    R, B = [], []
    worst = 0.0
    for k in range(t["robots"].shape[0]):
        balls = np.concatenate([t["balls"][k], np.zeros((t["balls"].shape[1], 2))], axis=1)
        o.set_clean_state(t["robots"][k], balls)
        s = o.get_state()
        R.append(s["robots"]); B.append(s["balls"])
        obs = o.observe(1)
        assert np.array_equal(obs, t["obs"][k]), (k, obs - t["obs"][k])
    R, B = np.array(R), np.array(B)
    assert rc.overlaps(R, B, cfg["W"], cfg["H"]) == dict(robot_robot=0, ball_goal=0, ball_robot=0, ball_ball=0)
    rc.check_support(R, B, cfg["W"], cfg["H"], need_endpoints=False)


@pytest.mark.parametrize("preset,arenas,episodes", [("T", 2000, 50), ("G", 1000, 100), ("D", 2000, 50), ("X", 1000, 100)])
def test_oracle_reset_distribution_matches_reference(golden_dir, preset, arenas, episodes):
    t = np.load(f"{golden_dir}/reset_{preset}.npz")
    cfg = ol.PRESETS[preset]
    R, B = oracle_layouts(preset, arenas, episodes)
    assert R.shape[0] >= 100000
    rc.check_support(R, B, cfg["W"], cfg["H"])
    assert rc.overlaps(R, B, cfg["W"], cfg["H"]) == dict(robot_robot=0, ball_goal=0, ball_robot=0, ball_ball=0)
    assert (B[:, :, 6:] == 0).all() and np.isnan(R[:, :, 7:]).all()  # balls at rest, no pose history (Ball.on_reset / Robot.on_reset)
    d = rc.ks_all(t["robots"], t["balls"], R, B, rot_col=6)
    p = rc.chi2_ball_occupancy(t["balls"], B, cfg["W"], cfg["H"])
    print(f"[{preset}] {R.shape[0]} oracle layouts vs {t['robots'].shape[0]} reference layouts: max KS D = {d:.4f}, occupancy chi2 p = {p:.3f}")
