"""The `robo_rugby` compat shim (compat/): the import lines and names of Training_DQN_pytorch.py:6-9,233-239,256-258 resolve,
constants equal the reference's (tests/golden/kat_*.npz `consts`, captured from the imported reference), and -- on the GPU --
a loop with the script's call pattern reproduces the reference's golden episodes through those names."""
import sys

import numpy as np
import pytest

import compat_paths


@pytest.mark.parametrize("preset", ["T", "G"])
def test_import_lines_of_the_dqn_script_resolve_and_constants_match(golden_dir, preset):
    compat_paths.use_compat(game_mode=(preset == "G"))
    import gym  # noqa: F401  (real gym if installed, else compat/gym_minimal)
    import robo_rugby
    import robo_rugby.gym_env.RR_Constants as const
    from robo_rugby.gym_env import GameEnv_Simple
    assert const.GAME_MODE == (preset == "G")
    c = np.load(f"{golden_dir}/kat_{preset}.npz")["consts"]
    got = [const.ARENA_WIDTH, const.ARENA_HEIGHT, const.GAME_LENGTH_STEPS, const.POINTS_BALL_TRAVEL_MULT, const.POINTS_ROBOT_TRAVEL_MULT,
           const.NUM_ROBOTS_HAPPY, const.NUM_ROBOTS_GRUMPY, const.NUM_BALL_POS, const.NUM_BALL_NEG, const.MOVES_PER_FRAME,
           const.CALC_DIST_TRACK_CENTER_TO_ROBOT_CENTER]
    assert np.array_equal(np.array(got, dtype=np.float64), c)
    assert const.NUM_ROBOTS_TOTAL == const.NUM_ROBOTS_HAPPY + const.NUM_ROBOTS_GRUMPY
    assert (const.TEAM_HAPPY, const.TEAM_GRUMPY, const.FRAMERATE) == (1, -1, 30)
    assert const.KEY_BOTH_MOTOR_FORWARD == ord("w") and const.KEY_BOTH_MOTOR_LEFT == ord("a")
    D = GameEnv_Simple.Direction  # Training_DQN_pytorch.py:208-229,258
    assert len(D) == 8 and [d.value for d in D] == list(range(8)) and D.F_L.value == 4 and D.B_R.value == 7
    assert [GameEnv_Simple.thrust_from_direction(d.value) for d in D] == \
        [(1, 1), (-1, -1), (-1, 1), (1, -1), (0, 1), (1, 0), (-1, 0), (0, -1)]  # RR_EnvBase.py:593-602
    spec = robo_rugby.REGISTRY["RoboRugbySimpleDuel-v3"]  # robo_rugby/__init__.py:28-34
    assert spec["max_episode_steps"] == const.GAME_LENGTH_STEPS and spec["nondeterministic"] and spec["reward_threshold"] == 1.0
    assert robo_rugby.gym_env.GameEnv.DebugInfo is not None  # the annotation at Training_DQN_pytorch.py:339
    if preset == "G":
        cs = robo_rugby.gym_env.GameEnv.CONFIG_STANDARD
        assert len(cs) == 2 and len(cs[0]) == 4 and len(cs[1]) == 8 and cs[0][0] == (560.0, 640.0, 135)


@pytest.mark.gpu
def test_dqn_script_call_pattern_reproduces_reference_episodes(golden_dir):
    """Training_DQN_pytorch.py:317-360, through the script's own names: gym.make -> reset -> unwrapped.get_game_state(int_team=)
    -> step([a]) -> info.adblGrumpyState / .dblGrumpyScore -> render(), fed the actions of the reference's golden T episodes
    (from the reference's start state, injected after reset): the (obs, reward, done) tuples are the reference's."""
    compat_paths.use_compat(game_mode=False)
    import gym
    import robo_rugby  # noqa: F401  (registers the id)
    import robo_rugby.gym_env.RR_Constants as const
    from robo_rugby.gym_env import GameEnv_Simple
    assert not const.GAME_MODE  # the script's guard, :233-234
    env = gym.make("RoboRugbySimpleDuel-v3")
    assert env.observation_space.shape == (11,) and env.spec.max_episode_steps == 300 and len(GameEnv_Simple.Direction) == 8
    assert env.metadata["video.frames_per_second"] == 30
    t = np.load(f"{golden_dir}/traj_T.npz")
    exact_to_end, total = 0, 0
    for ep in range(t["length"].shape[0]):
        if int(t["exc"][ep]):
            continue
        n = int(t["length"][ep])
        observation = env.reset()
        assert isinstance(observation, np.ndarray) and observation.shape == (11,) and observation.dtype == np.float64
        # test hook: start from the reference's dumped state instead of the random placement
        env.unwrapped._e._b.set_state(t["state_robots"][ep, 0], t["state_robots_i"][ep, 0], t["state_balls"][ep, 0], t["state_step"][ep, 0])
        observation = env.unwrapped.get_game_state()
        assert np.abs(observation - t["obs0"][ep]).max() < 1e-9
        obs_grumpy = env.unwrapped.get_game_state(int_team=const.TEAM_GRUMPY)
        assert obs_grumpy is None  # preset T has no grumpy robot (RR_Observers.py:304-307)
        score, done, s, diverged = 0.0, False, 0, None
        while not done:
            action = int(t["actions"][ep, s, 0])
            observation_, reward, done, info = env.step([action])
            # no grumpy robot on preset T, but PushPosBallsToGoal still books the opposite sign for that team (RR_ScoreKeepers.py:149-153)
            assert info.adblGrumpyState is None and isinstance(info.dblGrumpyScore, float)
            assert isinstance(reward, float) and isinstance(done, bool)
            score += reward
            if diverged is None and (np.abs(observation_ - t["obs"][ep, s]).max() > 1e-9 or abs(reward - t["reward"][ep, s]) > 1e-7):
                diverged = s  # chaotic after contacts: see test_free_running_episodes_vs_reference
            assert diverged is not None or abs(info.dblGrumpyScore - t["reward_g"][ep, s]) < 1e-7
            assert env.render() is None
            s += 1
            # gym's TimeLimit ends the episode at max_episode_steps; the raw env (what the golden recorded) one step later
            assert done == (s >= 300)
            if s >= n:
                break
        if n >= 300:
            assert done and s == 300 and info.get("TimeLimit.truncated") is True
        total += 1
        exact_to_end += diverged is None
        assert diverged is None or diverged >= 20, (ep, diverged)
    env.close()
    print(f"DQN call pattern through gym.make / robo_rugby names: {exact_to_end} of {total} golden episodes reproduced to their last step")
    assert exact_to_end >= 6
