"""The budgeted step on the MI355X, through the C-ABI (rr_config.step_budget_clocks / rr_set_step_budget; rr_sim.hpp: ParkCtx).

Promise: each arena's stream of (observation, reward, done, status), as a function of the actions it ACCEPTED, is the
synchronous mode's bit for bit -- whatever the budget, down to 1 clock (every expensive sub-step parks).  Checked on the stuck
arenas taken from the slowest wavefronts of chase-policy rollouts (tests/data/stuck_chase_*.npz) and on a 65,536-arena
chase-policy rollout with the policy in the loop.  tests/test_budgeted_step.py is the CPU twin (host-emulated wave)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
NOT_READY = 16384


def _chase(obs, cursor, table):
    """turn toward the ball, else forward; the noise of an arena's k-th accepted step comes from table[k, arena]: the action is a
    function of the arena's own observation and of how many steps it has accepted -- the same in both modes."""
    n = obs.shape[0]
    d = (obs[:, 1] - obs[:, 0] + 540.0) % 360.0 - 180.0
    a = torch.where(d.abs() < 8, 0, torch.where(d > 0, 2, 3)).to(torch.int32)
    row = table[cursor.clamp(max=table.shape[0] - 1), torch.arange(n, device=obs.device)]  # [n, na]: >= 8 keeps the chase action
    a1 = torch.where(row[:, 0] < 8, row[:, 0], a)
    return torch.cat([a1.view(n, 1), row[:, 1:] % 8], 1).contiguous()


def _streams(env, table, steps, budget_mode, max_calls):
    """runs until every arena has accepted `steps` steps; returns per-arena streams [steps, n, ...] of obs / reward / done / status"""
    n, dev = env.num_envs, env.device
    na = table.shape[2]
    rec_o = torch.zeros(steps, n, 11, device=dev); rec_r = torch.zeros(steps, n, device=dev)
    rec_d = torch.zeros(steps, n, dtype=torch.uint8, device=dev); rec_s = torch.zeros(steps, n, dtype=torch.int32, device=dev)
    rec_og = torch.zeros(steps, n, 11, device=dev) if env.has_grumpy else None
    out = (torch.zeros(n, 11, device=dev), torch.zeros(n, device=dev), torch.zeros(n, dtype=torch.uint8, device=dev),
           torch.zeros(n, 11, device=dev) if env.has_grumpy else None, torch.zeros(n, device=dev), torch.zeros(n, dtype=torch.int32, device=dev))
    out[0].copy_(env.get_game_state(1))
    cursor = torch.zeros(n, dtype=torch.long, device=dev)   # steps accepted AND completed
    parked = torch.zeros(n, dtype=torch.bool, device=dev)
    ar = torch.arange(n, device=dev)
    calls = not_ready_rows = 0
    while int(cursor.min()) < steps:
        a = _chase(out[0], cursor, table)
        if budget_mode:  # a parked arena must ignore what it is given: hand it something else
            a = torch.where(parked.view(n, 1), (a + 3) % 8, a)
        env.step(a[:, :na], out=out)
        calls += 1
        assert calls <= max_calls, "arenas do not make progress"
        ready = (out[5] & NOT_READY) == 0
        assert budget_mode or bool(ready.all())
        not_ready_rows += int((~ready).sum())
        idx = ar[ready & (cursor < steps)]
        c = cursor[idx]
        rec_o[c, idx] = out[0][idx]; rec_r[c, idx] = out[1][idx]; rec_d[c, idx] = out[2][idx]; rec_s[c, idx] = out[5][idx]
        if rec_og is not None:
            rec_og[c, idx] = out[3][idx]
        cursor += ready.long()
        parked = ~ready
    return (rec_o, rec_r, rec_d, rec_s, rec_og), calls, not_ready_rows


def _equal(a, b):
    return all(x is None or torch.equal(torch.nan_to_num(x.float(), nan=-7.0), torch.nan_to_num(y.float(), nan=-7.0)) for x, y in zip(a, b))


@pytest.mark.parametrize("preset", ["T", "G"])
def test_budgeted_equals_synchronous_on_stuck_chase_arenas(preset):
    import roborugby_amd as rr
    d = np.load(os.path.join(HERE, "data", f"stuck_chase_{preset}.npz"))
    n, na = len(d["step"]), d["actions"].shape[1]
    steps = 10
    g = torch.Generator(device="cuda").manual_seed(5)
    # mostly "keep the fixture's action" (>= 8 -> chase action; here replaced below), sometimes a random one
    table = torch.randint(0, 8, (steps, n, na), generator=g, device="cuda", dtype=torch.int32)
    keep = torch.rand(steps, n, generator=g, device="cuda") < 0.7
    table = torch.where(keep.unsqueeze(-1), torch.as_tensor(d["actions"], device="cuda").to(torch.int32).expand(steps, n, na), table)
    ref = None
    for budget in (0, 1, 50_000, 400_000):
        env = rr.BatchedRoboRugbyEnv(n, preset=preset, seed=3, time_limit=True, auto_reset=True, step_budget_clocks=budget)
        env.set_state(d["robots"], d["robots_i"], d["balls"], d["step"])
        got, calls, nr = _streams(env, table, steps, budget > 0, 2000)  # (a 1-clock budget parks after every sub-step and resolve pass)
        env.close()
        if ref is None:
            ref = got
            assert calls == steps and nr == 0
        else:
            assert _equal(got, ref), (preset, budget)
            if budget == 1:
                assert nr > n  # these arenas are the stuck ones: with a 1-clock budget they park over and over
        print(f"[{preset}] budget {budget}: {calls} calls for {steps} steps of {n} stuck arenas, {nr} NOT_READY rows")


@pytest.mark.parametrize("preset", ["T", "G"])
@pytest.mark.timeout(900)
def test_budgeted_equals_synchronous_on_a_65536_arena_chase_rollout(preset):
    """the contact-rich regime at full size, policy in the loop: 150 synchronous chase steps bring both envs to the same
    contact-rich state, then 30 accepted steps per arena are compared between the synchronous mode and two budgets."""
    import roborugby_amd as rr
    n, warm, steps = 65536, 150, 30
    na = 1 if preset == "T" else 4
    g = torch.Generator(device="cuda").manual_seed(9)
    table = torch.randint(0, 80, (warm + steps, n, na), generator=g, device="cuda", dtype=torch.int32)  # 10 % random, else chase
    if na > 1:
        table[:, :, 1:] %= 8
    ref = None
    for budget in (0, 150_000, 600_000):
        env = rr.BatchedRoboRugbyEnv(n, preset=preset, seed=12, time_limit=True, auto_reset=True)
        env.reset()
        _streams(env, table[:warm], warm, False, warm)          # synchronous warm-up: identical in every run
        env.set_step_budget(budget)
        got, calls, nr = _streams(env, table[warm:], steps, budget > 0, 40 * steps)
        if ref is None:
            ref = got
        else:
            assert _equal(got, ref), (preset, budget)
            assert nr > 0, "no arena ever parked: the budget is not exercised"
            # switching the budget off lets the parked arenas finish; nothing parks any more
            env.set_step_budget(0)
            a = torch.zeros(n, na, dtype=torch.int32, device="cuda")
            for _ in range(3):
                o, r, dn, info = env.step(a)
            assert int((info.status & NOT_READY).sum()) == 0
        print(f"[{preset}] budget {budget}: {calls} calls until every arena had accepted {steps} steps, {nr} NOT_READY rows "
              f"({100.0 * nr / (calls * n):.2f} % of the rows)")
        env.close()


def test_budget_rejects_what_it_cannot_bracket():
    """rr_rollout keeps the record in LDS across its steps: no boundary to park at."""
    import roborugby_amd as rr
    from roborugby_amd import _lib
    env = rr.BatchedRoboRugbyEnv(64, preset="G", step_budget_clocks=1000)
    a = torch.zeros(5, 64, 4, dtype=torch.int32, device="cuda")
    with pytest.raises(_lib.RRError, match="step budget"):
        env.rollout(a)
    env.close()


@pytest.mark.parametrize("preset,goal", [("G", False), ("G", True), ("D", False)])
def test_budgeted_equals_synchronous_with_another_reward_stack_and_goal_scoring(preset, goal):
    """The reference's other score keepers (RR_ScoreKeepers.py:69-179) run in side kernels that bracket a step: on_step_begin copies
    before it, the keeper program after it.  Under a budget a step spans several calls -- the copies are taken when the arena's step
    BEGINS (a parked arena keeps them), its on_step_end (and its goal frame) runs with the call that completes it.  Per-arena streams
    of a four-keeper stack (DontDriveInGoals first, KeepMovingGuys reading the prior-step copies), optionally with the goal-scoring
    mode, equal the synchronous mode's bit for bit at three budgets, on stuck arenas that park over and over."""
    import roborugby_amd as rr
    d = np.load(os.path.join(HERE, "data", "stuck_chase_G.npz")) if preset == "G" else None
    stack = ("DontDriveInGoals", "KeepMovingGuys", "PushPosBallsToGoal", "ChasePosBall", "NaughtyBots")
    n = len(d["step"]) if d is not None else 512
    na = 4 if preset == "G" else 2
    steps = 8
    g = torch.Generator(device="cuda").manual_seed(11)
    table = torch.randint(0, 80, (steps, n, na), generator=g, device="cuda", dtype=torch.int32)  # mostly "chase", sometimes random
    table[:, :, 1:] %= 8
    ref = None
    for budget in (0, 1, 30_000, 300_000):
        env = rr.BatchedRoboRugbyEnv(n, preset=preset, seed=3, time_limit=True, auto_reset=True, rewards=stack, goal_scoring=goal,
                                     step_budget_clocks=budget)
        if d is not None:
            env.set_state(d["robots"], d["robots_i"], d["balls"], d["step"])
        else:
            env.reset()
        got, calls, nr = _streams(env, table, steps, budget > 0, 3000)
        env.close()
        if ref is None:
            ref = got
            assert calls == steps and nr == 0
            assert float(got[1].abs().sum()) > 0.0  # the stack pays something
        else:
            assert _equal(got, ref), (preset, goal, budget)
            if budget == 1 and d is not None:  # (arenas fresh from a reset are mostly quiet: a quiet arena never reads the clock)
                assert nr > n // 2
        print(f"[{preset} goal={goal}] budget {budget}: {calls} calls for {steps} steps of {n} arenas, {nr} NOT_READY rows")


def test_budgeted_step_with_its_policy_can_be_captured_into_a_hip_graph():
    """the budgeted rr_step and the chase policy only enqueue kernels on the caller's stream: captured once, replayed, same streams
    of outputs as eager calls (the clock decides who parks, so the two envs are compared through their action-aligned streams)"""
    import roborugby_amd as rr
    from roborugby_amd import players
    n, steps = 8192, 12
    env = rr.BatchedRoboRugbyEnv(n, preset="T", seed=5, step_budget_clocks=60_000)
    obs = env.reset()
    out = (obs.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, dtype=torch.uint8, device="cuda"), None,
           torch.zeros(n, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda"))
    acts = torch.zeros(n, 1, dtype=torch.int32, device="cuda")
    accepted = torch.zeros(n, dtype=torch.int32, device="cuda")   # per-arena count of accepted steps: the policy's step index

    def body():
        players.chase(env, out[0], step_of=accepted, noise=0.1, seed=3, out=acts)
        env.step(acts, out=out)
        accepted.add_(((out[5] & NOT_READY) == 0).to(torch.int32))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    assert int(accepted.min()) > 20 and int(accepted.max()) <= 201  # (stuck arenas park dozens of times per step at this budget)
    # the same arenas stepped synchronously with the same policy reach the same observation after the same number of accepted steps
    ref = rr.BatchedRoboRugbyEnv(n, preset="T", seed=5)
    o = ref.reset()
    k = int(accepted.min())
    cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
    for s in range(k):
        a = players.chase(ref, o, step_of=cnt, noise=0.1, seed=3)
        o, r, d, info = ref.step(a)
        cnt += 1
    sel = accepted == k
    assert int(sel.sum()) > 0 and torch.equal(out[0][sel], o[sel])
    env.close(); ref.close()


def test_chase_policy_kernel_is_the_documented_rule():
    import roborugby_amd as rr
    from roborugby_amd import players
    n = 65536
    env = rr.BatchedRoboRugbyEnv(n, preset="G", seed=1)
    obs = env.reset()
    a = players.chase(env, obs, step=1, noise=0.0)
    d = (obs[:, 1] - obs[:, 0] + 540.0) % 360.0 - 180.0
    want = torch.where(d.abs() < 8, 0, torch.where(d > 0, 2, 3)).to(torch.int32)
    assert a.shape == (n, 4) and torch.equal(a[:, 0], want)
    hist = torch.bincount(a[:, 1:].reshape(-1).long(), minlength=8).float() / (3 * n)
    assert float((hist - 0.125).abs().max()) < 0.01                      # the other robots act uniformly at random
    b = players.chase(env, obs, step=1, noise=0.25)
    assert abs(float((b[:, 0] != want).float().mean()) - 0.25 * 7 / 8) < 0.01
    assert torch.equal(players.chase(env, obs, step=1, noise=0.25), b)   # a function of (seed, arena, step) ...
    assert not torch.equal(players.chase(env, obs, step=2, noise=0.25), b)  # ... and nothing else
    env.close()


def test_hbm_copy_probe_copies():
    import ctypes as C
    from roborugby_amd import _lib
    lib = _lib.load()
    src = torch.randn(1 << 22, device="cuda")
    dst = torch.zeros_like(src)
    _lib.check(lib.rr_probe_hbm_copy(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), C.c_size_t(src.numel() * 4),
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rr_probe_hbm_copy")
    torch.cuda.synchronize()
    assert torch.equal(dst, src)
    assert lib.rr_probe_hbm_copy(C.c_void_p(dst.data_ptr() + 4), C.c_void_p(src.data_ptr()), C.c_size_t(64), None) == -1


@pytest.mark.parametrize("preset", ["T", "G"])
def test_not_ready_rows_keep_the_previous_observation_through_the_default_outputs(preset):
    """step() with out=None hands out persistent buffers in the budgeted mode, and a row whose arena parks in the FIRST
    call after construction / reset() / set_state() / set_step_budget() must read as the arena's previous observation (both teams),
    not as zeros.  Budget = 1 clock on the stuck chase arenas: every one of them parks in its first call."""
    import roborugby_amd as rr
    d = np.load(os.path.join(HERE, "data", f"stuck_chase_{preset}.npz"))
    n = len(d["step"])
    acts = torch.as_tensor(d["actions"], device="cuda").to(torch.int32)

    def first_call(env, what):
        prev_h = env.get_game_state(1).clone()
        prev_g = env.get_game_state(-1).clone() if env.has_grumpy else None
        obs, rew, done, info = env.step(acts)
        nr = (info.status & NOT_READY) != 0
        assert int(nr.sum()) >= n // 2, (what, int(nr.sum()))  # the stuck arenas do park at a 1-clock budget
        assert torch.equal(obs[nr], prev_h[nr]), what
        assert float(obs[nr].abs().sum()) > 0.0
        assert not bool(done[nr].any()) and float(rew[nr].abs().sum()) == 0.0
        if prev_g is not None:
            assert torch.equal(info.adblGrumpyState[nr], prev_g[nr]), what
        # the parked arenas finish in later calls (at one resolve pass per call a squeezed G arena needs > 120 of them: switch the
        # budget off, the next call completes every parked step), with the observation of the step they accepted
        env.set_step_budget(0)
        obs, rew, done, info = env.step(acts)
        assert not bool(((info.status & NOT_READY) != 0).any())
        assert float(obs.abs().sum(1).min()) > 0.0
        env.set_step_budget(1)

    # (a) budget at construction, state written from outside
    env = rr.BatchedRoboRugbyEnv(n, preset=preset, seed=3, time_limit=False, auto_reset=False, step_budget_clocks=1)
    env.set_state(d["robots"], d["robots_i"], d["balls"], d["step"])
    first_call(env, "constructor + set_state")
    # (b) after a masked reset of half of the arenas followed by set_state of all (the rows not reset keep theirs)
    env.reset(mask=torch.arange(n, device="cuda") % 2 == 0)
    env.set_state(d["robots"], d["robots_i"], d["balls"], d["step"])
    first_call(env, "masked reset + set_state")
    env.close()
    # (c) budget switched on later
    env = rr.BatchedRoboRugbyEnv(n, preset=preset, seed=3, time_limit=False, auto_reset=False)
    env.set_state(d["robots"], d["robots_i"], d["balls"], d["step"])
    env.set_step_budget(1)
    first_call(env, "set_step_budget")
    env.close()
