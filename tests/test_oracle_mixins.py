"""The reference's other reward / observation mixins in the oracle, pinned bit-exactly
against tests/golden/mix_*.npz: episodes of two mixin stacks the reference can construct
  MixA = DontDriveInGoals, KeepMovingGuys, PushNegBallsFromGoal, BaseDestruction, PushPosBallsToGoal, ChasePosBall
  MixB = KeepMovingGuys, NaughtyBots, DontDriveInGoals   (NaughtyBots.on_step_end never calls super(): the keepers
         behind it in the MRO do not run -- RR_ScoreKeepers.py:130-135)
plus the SingleBall_6wayLidar / PosBall_BasicLidar / AllCoords / AllCoords_WithPrior observations of the same states."""
import json

import numpy as np
import pytest

import oracle_lib as ol


def _eq(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


@pytest.mark.parametrize("preset", ["T", "G", "D", "X"])
def test_mixin_rewards_and_observers_bit_exact(golden_dir, preset):
    t = np.load(f"{golden_dir}/mix_{preset}.npz")
    meta = json.loads(str(t["meta"]))
    progs = {0: meta["programs_exec"]["A"], 1: meta["programs_exec"]["B"]}
    assert progs[1] == [1, 5]  # NaughtyBots, then KeepMovingGuys; DontDriveInGoals is cut off by the missing super()
    nr = ol.PRESETS[preset]["nr_h"] + ol.PRESETS[preset]["nr_g"]
    has_g = ol.PRESETS[preset]["nr_g"] > 0
    total = pen = 0
    for ep in range(t["length"].shape[0]):
        env = ol.OracleEnv(preset)
        env.set_program(progs[int(t["which"][ep])])
        n = int(t["length"][ep])
        S = {k: t["state_" + k][ep] for k in ("robots", "robots_i", "balls", "inner", "step")}
        env.set_state(S["robots"][0], S["robots_i"][0], S["balls"][0], S["inner"][0], S["step"][0])
        for s in range(n):
            a = t["actions"][ep, s]
            r = env.step(a[a >= 0])
            st = env.get_state()
            assert _eq(st["robots"], S["robots"][s + 1]) and _eq(st["balls"], S["balls"][s + 1]), (preset, ep, s)
            assert r["reward"] == t["reward"][ep, s] and r["reward_g"] == t["reward_g"][ep, s], (preset, ep, s, r["reward"], t["reward"][ep, s])
            assert _eq(env.observe_kind(1, 1), t["v1_h"][ep, s]), (preset, ep, s)
            assert _eq(env.observe_kind(2, 1), t["basic_h"][ep, s])
            assert _eq(env.observe_kind(3, 1), t["all_h"][ep, s])
            assert _eq(env.observe_kind(3, -1), t["all_g"][ep, s])
            assert _eq(env.observe_kind(4, 1), t["allp_h"][ep, s]) and _eq(env.observe_kind(4, -1), t["allp_g"][ep, s])  # AllCoords_WithPrior
            if has_g:
                assert _eq(env.observe_kind(1, -1), t["v1_g"][ep, s])
                assert _eq(env.observe_kind(2, -1), t["basic_g"][ep, s])
            else:
                assert env.observe_kind(1, -1) is None and env.observe_kind(2, -1) is None
            last_team = -1 if has_g else 1
            assert _eq(env.observe_kind(2, last_team, robot=nr - 1), t["basic_last"][ep, s])
            nb = ol.PRESETS[preset]["nb_p"] + ol.PRESETS[preset]["nb_n"]
            assert _eq(env.observe_kind(1, last_team, robot=nr - 1, ball=nb - 1), t["v1_last_negball"][ep, s])
            total += 1
            pen += int(abs(t["reward"][ep, s] % 0.005) < 1e-12 and t["reward"][ep, s] < 0)
    assert total > 1000
    assert pen > 10  # the penalty keepers (DontDriveInGoals / KeepMovingGuys / NaughtyBots) really fired in stack B
