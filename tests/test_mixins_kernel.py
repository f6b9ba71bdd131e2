"""The reference's other mixin stacks through the product path: the side kernels' source on the host emulation (CPU) and
the HIP library through the C-ABI (GPU), against tests/golden/mix_*.npz (captured from the reference)."""
import json

import numpy as np
import pytest

import emu_lib as el

TOL = 1e-9


def _close(a, b, tol=TOL):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.array_equal(np.isnan(a), np.isnan(b)) and (np.nan_to_num(np.abs(a - b)).max() if a.size else 0.0) < tol


@pytest.mark.parametrize("preset", ["T", "G", "D", "X"])
def test_emulated_side_kernels_vs_reference_golden(golden_dir, preset):
    t = np.load(f"{golden_dir}/mix_{preset}.npz")
    meta = json.loads(str(t["meta"]))
    progs = {0: meta["programs_exec"]["A"], 1: meta["programs_exec"]["B"]}
    cfg = el.ol.PRESETS[preset]
    has_g = cfg["nr_g"] > 0
    nr, nb = cfg["nr_h"] + cfg["nr_g"], cfg["nb_p"] + cfg["nb_n"]
    n = 0
    for ep in range(t["length"].shape[0]):
        env = el.EmuEnv(preset)
        env.set_program(progs[int(t["which"][ep])])
        for s in range(0, int(t["length"][ep]), 3):
            env.set_state(t["state_robots"][ep, s], t["state_robots_i"][ep, s], t["state_balls"][ep, s], t["state_step"][ep, s])
            a = t["actions"][ep, s]
            r = env.step(a[a >= 0])
            assert abs(r["reward"] - t["reward"][ep, s]) < 1e-7 and abs(r["reward_g"] - t["reward_g"][ep, s]) < 1e-7, (preset, ep, s)
            assert _close(env.observe_kind(1, 1), t["v1_h"][ep, s]) and _close(env.observe_kind(2, 1), t["basic_h"][ep, s])
            assert _close(env.observe_kind(3, 1), t["all_h"][ep, s]) and _close(env.observe_kind(3, -1), t["all_g"][ep, s])
            assert _close(env.observe_kind(4, 1), t["allp_h"][ep, s]) and _close(env.observe_kind(4, -1), t["allp_g"][ep, s])
            if has_g:
                assert _close(env.observe_kind(1, -1), t["v1_g"][ep, s]) and _close(env.observe_kind(2, -1), t["basic_g"][ep, s])
            else:
                assert env.observe_kind(1, -1) is None and env.observe_kind(2, -1) is None
            lt = -1 if has_g else 1
            assert _close(env.observe_kind(2, lt, robot=nr - 1), t["basic_last"][ep, s])
            assert _close(env.observe_kind(1, lt, robot=nr - 1, ball=nb - 1), t["v1_last_negball"][ep, s])
            n += 1
    assert n > 300


def test_keeper_exec_order_rule():
    from roborugby_amd.env import keeper_exec_order
    assert keeper_exec_order(("PushPosBallsToGoal", "ChasePosBall", "NaughtyBots")) == [1, 2, 3]  # SimpleDuel3
    assert keeper_exec_order(("KeepMovingGuys", "NaughtyBots", "DontDriveInGoals")) == [1, 5]     # cut at NaughtyBots
    assert keeper_exec_order(("ChasePosBall", "PushPosBallsToGoal")) == [3, 2]                    # SimpleDuel
    with pytest.raises(KeyError):
        keeper_exec_order(("PushPosBallsInYourGoal",))  # an empty TODO class in the reference


@pytest.mark.gpu
@pytest.mark.parametrize("preset", ["T", "G", "D", "X"])
def test_gpu_mixin_stacks_vs_reference_golden(golden_dir, preset):
    import torch
    import roborugby_amd as rr
    t = np.load(f"{golden_dir}/mix_{preset}.npz")
    meta = json.loads(str(t["meta"]))
    nr = rr.PRESETS[preset].nr
    for which, stack in ((0, meta["programs_mro"]["A"]), (1, meta["programs_mro"]["B"])):
        eps = np.nonzero(t["which"] == which)[0]
        idx = [(ep, s) for ep in eps for s in range(int(t["length"][ep])) if (t["actions"][ep, s] >= 0).sum() == nr]
        ep = np.array([i[0] for i in idx]); st = np.array([i[1] for i in idx])
        env = rr.BatchedRoboRugbyEnv(len(idx), preset=preset, time_limit=False, auto_reset=False, rewards=tuple(stack),
                                     observer="SingleBall_6wayLidar")
        assert env.observation_space.shape == (11,)
        env.set_state(t["state_robots"][ep, st], t["state_robots_i"][ep, st], t["state_balls"][ep, st], t["state_step"][ep, st])
        o, r, d, info = env.step_f64(torch.as_tensor(t["actions"][ep, st].astype(np.int32)))
        assert np.abs(r.cpu().numpy() - t["reward"][ep, st]).max() < 1e-7
        assert np.abs(info.dblGrumpyScore.cpu().numpy() - t["reward_g"][ep, st]).max() < 1e-7
        for name, key, team in (("SingleBall_6wayLidar", "v1", 1), ("PosBall_BasicLidar", "basic", 1), ("AllCoords", "all", 1),
                                ("AllCoords", "all", -1)):
            got = env.get_game_state(team, f64=True, observer=name).cpu().numpy()
            assert _close(got, t[f"{key}_{'h' if team == 1 else 'g'}"][ep, st]), (preset, which, name, team)
        # AllCoords_WithPrior needs the on_step_begin snapshot: an env configured with that observer tracks it
        envp = rr.BatchedRoboRugbyEnv(len(idx), preset=preset, time_limit=False, auto_reset=False, rewards=tuple(stack),
                                      observer="AllCoords_WithPrior")
        assert envp.observation_space.shape == (6 * nr + 4 * rr.PRESETS[preset].nb,)
        envp.set_state(t["state_robots"][ep, st], t["state_robots_i"][ep, st], t["state_balls"][ep, st], t["state_step"][ep, st])
        envp.step_f64(torch.as_tensor(t["actions"][ep, st].astype(np.int32)))
        for team in (1, -1):
            got = envp.get_game_state(team, f64=True).cpu().numpy()
            assert _close(got, t[f"allp_{'h' if team == 1 else 'g'}"][ep, st]), (preset, which, "AllCoords_WithPrior", team)
        with pytest.raises(RuntimeError):
            env.get_game_state(1, observer="AllCoords_WithPrior")  # this env never asked for the snapshot
        if rr.PRESETS[preset].nr_grumpy > 0:
            got = env.get_game_state(-1, f64=True, observer="SingleBall_6wayLidar").cpu().numpy()
            assert _close(got, t["v1_g"][ep, st])
        else:
            assert env.get_game_state(-1, observer="PosBall_BasicLidar") is None
        # float32 step + configured observer: obs returned by step() is the v1 observation
        env.set_state(t["state_robots"][ep, st], t["state_robots_i"][ep, st], t["state_balls"][ep, st], t["state_step"][ep, st])
        o32, r32, _, _ = env.step(torch.as_tensor(t["actions"][ep, st].astype(np.int32)))
        assert np.abs(o32.cpu().numpy() - t["v1_h"][ep, st]).max() < 1e-3
        assert np.abs(r32.cpu().numpy() - t["reward"][ep, st]).max() < 1e-2
