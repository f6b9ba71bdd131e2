"""GPU tests of the exact shortcuts and of the dispatch order (whole-arena fixed point, island freeze, slowest-first
launch order): switching them off must not change one bit of any output, on stuck islands, on contact-dense synthetic
states, and on a full-size rollout; and the stuck arenas must still agree with the CPU oracle."""
import os

import numpy as np
import pytest

import adversarial as adv
import oracle_lib as ol

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


BUILDS = pytest.mark.parametrize("exact", [False, True], ids=["default", "parity_build"])  # the parity build's carry has to survive the shortcuts too


def _env(n, preset, exact=False, **env_vars):
    import roborugby_amd as rr
    old = {k: os.environ.get(k) for k in env_vars}
    os.environ.update({k: str(v) for k, v in env_vars.items()})
    try:
        return rr.BatchedRoboRugbyEnv(n, preset=preset, seed=3, time_limit=False, auto_reset=False, exact_trig=exact)  # switches are read at creation
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _run(env, state, actions, steps):
    env.set_state(*state)
    out = []
    for _ in range(steps):
        o, r, d, info = env.step_f64(actions)
        st = env.get_state()
        out.append([t.cpu().numpy().copy() for t in (o, r, d, info.adblGrumpyState, info.dblGrumpyScore, info.status,
                                                     st["robots"], st["robots_i"], st["balls"], st["step"]) if t is not None])
    return out


def _same(a, b):
    return all(np.array_equal(x, y, equal_nan=True) for sa, sb in zip(a, b) for x, y in zip(sa, sb))


@BUILDS
def test_stuck_islands_shortcuts_off_equals_on_and_match_oracle(exact):
    d = np.load(os.path.join(HERE, "data", "stuck_islands_G.npz"))
    f = np.load(os.path.join(HERE, "..", "tools", "fixtures", "squeezed_G.npz"))
    robots = np.concatenate([d["robots"], f["robots"][None]]); robots_i = np.concatenate([d["robots_i"], f["robots_i"][None]])
    balls = np.concatenate([d["balls"], f["balls"][None]]); step = np.concatenate([d["step"], [f["step"]]]).astype(np.int32)
    acts = np.concatenate([d["actions"], f["actions"][None]]).astype(np.int32)
    n = len(step)
    a_t = torch.as_tensor(acts, device="cuda")
    on = _run(_env(n, "G", exact), (robots, robots_i, balls, step), a_t, 4)
    off = _run(_env(n, "G", exact, RR_NO_MEMO=1, RR_NO_ORDER=1), (robots, robots_i, balls, step), a_t, 4)
    assert _same(on, off)
    # first step against the oracle (fp64, same tolerance as the parity tests)
    worst = 0.0
    for a in range(n):
        o = ol.OracleEnv("G")
        o.set_state(robots[a], robots_i[a], balls[a], None, int(step[a]))
        r = o.step(acts[a])
        s = o.get_state()
        if r["status"] & 63:
            continue
        worst = max(worst, float(np.nanmax(np.abs(s["robots"] - on[0][6][a]))), float(np.abs(s["balls"] - on[0][8][a]).max()),
                    float(np.abs(r["obs"] - on[0][0][a]).max()))
        assert np.array_equal(s["robots_i"], on[0][7][a])
    assert worst < 1e-9, worst


@BUILDS
@pytest.mark.parametrize("preset", ["T", "G"])
def test_contact_dense_states_shortcuts_off_equals_on(preset, exact):
    n = 1536
    robots, balls, actions = adv.make_states(preset, n, seed=21)
    import roborugby_amd as rr
    outs = []
    for sw in ({}, {"RR_NO_MEMO": 1, "RR_NO_ORDER": 1}):
        env = _env(n, preset, exact, **sw)
        env.set_poses(robots, balls)
        a_t = torch.as_tensor(actions, device="cuda")
        res = []
        for _ in range(5):  # the same action five times: robots keep pushing, islands form and freeze
            o, r, d, info = env.step_f64(a_t)
            st = env.get_state()
            res.append([t.cpu().numpy().copy() for t in (o, r, d, info.status, st["robots"], st["robots_i"], st["balls"])])
        outs.append(res)
    assert _same(outs[0], outs[1])


def test_full_size_rollout_independent_of_shortcuts_and_dispatch_order():
    import roborugby_amd as rr
    n = 65536
    gen = torch.Generator(device="cuda"); gen.manual_seed(9)
    acts = torch.randint(0, 8, (40, n, 4), generator=gen, device="cuda", dtype=torch.int32)
    finals = []
    for sw in ({}, {"RR_NO_MEMO": 1, "RR_NO_ORDER": 1}):
        old = {k: os.environ.get(k) for k in sw}
        os.environ.update({k: str(v) for k, v in sw.items()})
        try:
            env = rr.BatchedRoboRugbyEnv(n, preset="G", seed=0)
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        env.reset()
        rew = torch.zeros(n, device="cuda", dtype=torch.float64)
        for s in range(40):
            o, r, d, info = env.step(acts[s])
            rew += r.double()
        st = env.get_state()
        finals.append((o.cpu().numpy(), rew.cpu().numpy(), st["robots"].cpu().numpy(), st["balls"].cpu().numpy(), st["robots_i"].cpu().numpy()))
    for x, y in zip(*finals):
        assert np.array_equal(x, y, equal_nan=True)


@pytest.mark.parametrize("preset", ["T", "G"])
def test_rollout_in_one_launch_equals_step_by_step(preset):
    """rr_rollout (S steps per launch, the record stays in LDS) against S rr_step calls: every per-step output and the final
    state bit-identical, across an episode boundary (auto-reset inside the launch) and with action repeat."""
    import roborugby_amd as rr
    n, S = 4096, 9
    na = rr.PRESETS[preset].nr
    gen = torch.Generator(device="cuda"); gen.manual_seed(4)
    acts = torch.randint(0, 8, (S, n, na), generator=gen, device="cuda", dtype=torch.int32)
    envs = [rr.BatchedRoboRugbyEnv(n, preset=preset, seed=11) for _ in range(3)]
    for e in envs:
        e.reset()
        st = e.get_state()
        st["step"][: n // 2] = e.preset.game_len_steps - 4  # half of the arenas finish (and are re-placed) inside the rollout
        e.set_state(st["robots"], st["robots_i"], st["balls"], st["step"])
    ref = [envs[0].step(acts[s]) for s in range(S)]
    o, r, d, info = envs[1].rollout(acts)
    for s in range(S):
        assert torch.equal(o[s], ref[s][0]) and torch.equal(r[s], ref[s][1]) and torch.equal(d[s], ref[s][2]), (preset, s)
        assert torch.equal(info.status[s], ref[s][3].status) and torch.equal(info.dblGrumpyScore[s], ref[s][3].dblGrumpyScore)
        if info.adblGrumpyState is not None:
            assert torch.equal(info.adblGrumpyState[s], ref[s][3].adblGrumpyState)
    a, b = envs[0].get_state(), envs[1].get_state()
    assert all(torch.equal(a[k], b[k]) or (a[k].dtype.is_floating_point and torch.equal(a[k].nan_to_num(7e77), b[k].nan_to_num(7e77))) for k in a)
    assert int(d.sum()) >= n // 2  # the episode boundary really was inside
    # action repeat (frame skip): the same action S times
    rep = [envs[0].step(acts[0]) for _ in range(4)]
    o2, r2, d2, _ = envs[1].rollout(acts[0], repeat=4)
    assert all(torch.equal(o2[s], rep[s][0]) and torch.equal(r2[s], rep[s][1]) for s in range(4))
    del envs


def test_step_can_be_captured_into_a_hip_graph():
    """rr_step only enqueues kernels on the caller's stream (no synchronisation, allocation or copy), so a step with
    preallocated outputs can be captured once and replayed: same results as eager calls."""
    import roborugby_amd as rr
    n = 8192
    gen = torch.Generator(device="cuda"); gen.manual_seed(2)
    acts = torch.randint(0, 8, (12, n, 4), generator=gen, device="cuda", dtype=torch.int32)
    eager, graphed = rr.BatchedRoboRugbyEnv(n, preset="G", seed=5), rr.BatchedRoboRugbyEnv(n, preset="G", seed=5)
    eager.reset(); graphed.reset()
    a_buf = acts[0].clone()
    out = (torch.empty(n, 11, device="cuda"), torch.empty(n, device="cuda"), torch.empty(n, dtype=torch.uint8, device="cuda"),
           torch.empty(n, 11, device="cuda"), torch.empty(n, device="cuda"), torch.empty(n, dtype=torch.int32, device="cuda"))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graphed.step(a_buf, out=out)  # warm-up on the side stream, as torch's capture protocol asks
    torch.cuda.current_stream().wait_stream(side)
    ref0 = eager.step(acts[0])
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        graphed.step(a_buf, out=out)
    for s in range(1, 12):  # (the capture itself executed nothing: the warm-up was step 0)
        a_buf.copy_(acts[s])
        g.replay()
        o, r, d, info = eager.step(acts[s])
        torch.cuda.synchronize()
        assert torch.equal(out[0], o) and torch.equal(out[1], r) and torch.equal(out[2].bool(), d) and torch.equal(out[5], info.status), s


@BUILDS
@pytest.mark.parametrize("preset", ["T", "G", "Grandom"])
def test_islands_carried_across_steps_shortcuts_off_equals_on(preset, exact):
    """Stuck arenas from the slowest wavefronts of a chase-policy rollout (tests/data/stuck_chase_*.npz): ten steps in which
    every robot keeps its action most of the time (the frozen island is carried into the next step), changes it now and then
    (recomputed), with one rr_set_state of the state they have in between (nothing may be carried) -- shortcuts on == off."""
    # ("Grandom": the slowest arenas of RANDOM-policy launches at the steady state, tests/data/stuck_random_G.npz -- among them the ones
    # whose island gains a resting neighbour ball, rr_sim.hpp: resting_neighbours)
    d = np.load(os.path.join(HERE, "data", "stuck_random_G.npz" if preset == "Grandom" else f"stuck_chase_{preset}.npz"))
    preset = preset[0]
    n = len(d["step"])
    rng = np.random.RandomState(5)
    acts = []
    for s in range(10):
        a = d["actions"].copy()
        ch = rng.rand(n) < 0.25
        a[ch] = rng.randint(0, 8, a[ch].shape)
        acts.append(torch.as_tensor(a.astype(np.int32), device="cuda"))
    outs = []
    for sw in ({}, {"RR_NO_MEMO": 1, "RR_NO_ORDER": 1}):
        env = _env(n, preset, exact, **sw)
        env.set_state(d["robots"], d["robots_i"], d["balls"], d["step"].astype(np.int32))
        res = []
        for s in range(10):
            if s == 6:  # rewrite the arenas with the state they have
                st = env.get_state()
                env.set_state(st["robots"], st["robots_i"], st["balls"], st["step"])
            o, r, dn, info = env.step_f64(acts[s])
            st = env.get_state()
            res.append([t.cpu().numpy().copy() for t in (o, r, dn, info.status, st["robots"], st["robots_i"], st["balls"])])
        outs.append(res)
    assert _same(outs[0], outs[1])


def test_full_size_chase_rollout_independent_of_shortcuts():
    """65,536 T arenas, 120 steps of the chase policy (robots end up driving their ball into a wall: islands freeze, are
    carried across steps, thaw on the 10 % random actions, episodes end and arenas are re-placed inside the rollout)."""
    import roborugby_amd as rr
    n = 65536
    finals = []
    for sw in ({}, {"RR_NO_MEMO": 1, "RR_NO_ORDER": 1}):
        old = {k: os.environ.get(k) for k in sw}
        os.environ.update({k: str(v) for k, v in sw.items()})
        try:
            env = rr.BatchedRoboRugbyEnv(n, preset="T", seed=0)
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        gen = torch.Generator(device="cuda"); gen.manual_seed(17)
        obs = env.reset()
        st = env.get_state()
        st["step"][: n // 4] = env.preset.game_len_steps - 60  # a quarter of the arenas is re-placed half-way
        env.set_state(st["robots"], st["robots_i"], st["balls"], st["step"])
        rew = torch.zeros(n, device="cuda", dtype=torch.float64)
        nd = 0
        for s in range(120):
            dlt = (obs[:, 1] - obs[:, 0] + 540.0) % 360.0 - 180.0
            a = torch.where(dlt.abs() < 8, 0, torch.where(dlt > 0, 2, 3)).to(torch.int32)
            noise = torch.rand(n, generator=gen, device="cuda") < 0.1
            a = torch.where(noise, torch.randint(0, 8, (n,), generator=gen, device="cuda", dtype=torch.int32), a).view(n, 1)
            obs, r, d, info = env.step(a)
            rew += r.double(); nd += int(d.sum())
        st = env.get_state()
        finals.append((obs.cpu().numpy(), rew.cpu().numpy(), st["robots"].cpu().numpy(), st["balls"].cpu().numpy(), st["robots_i"].cpu().numpy(), nd))
    assert finals[0][-1] >= n // 4
    for x, y in zip(finals[0][:-1], finals[1][:-1]):
        assert np.array_equal(x, y, equal_nan=True)


def test_shard_envs_on_two_streams_equal_one_batch():
    """bench.py --pipeline 2: the batch as two shard envs (arena_offset) stepped on two HIP streams with overlapping launches --
    every arena's trajectory is the single batch's, bit for bit."""
    import roborugby_amd as rr
    n, S = 16384, 30
    gen = torch.Generator(device="cuda"); gen.manual_seed(2)
    acts = torch.randint(0, 8, (S, n, 4), generator=gen, device="cuda", dtype=torch.int32)
    whole = rr.BatchedRoboRugbyEnv(n, preset="G", seed=5)
    whole.reset()
    for s in range(S):
        ow, rw, dw, _ = whole.step(acts[s])
    halves = [rr.BatchedRoboRugbyEnv(n // 2, preset="G", seed=5, arena_offset=i * (n // 2)) for i in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    torch.cuda.synchronize()
    res = [None, None]
    for i, e in enumerate(halves):
        with torch.cuda.stream(streams[i]):
            e.reset()
    for s in range(S):
        for i, e in enumerate(halves):
            with torch.cuda.stream(streams[i]):
                res[i] = e.step(acts[s, i * (n // 2):(i + 1) * (n // 2)])
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([res[0][0], res[1][0]]), ow) and torch.equal(torch.cat([res[0][1], res[1][1]]), rw)
    a = whole.get_state()
    b = [h.get_state() for h in halves]
    for k in ("robots", "balls", "robots_i"):
        x, y = a[k], torch.cat([b[0][k], b[1][k]])
        assert torch.equal(x.nan_to_num(7e77), y.nan_to_num(7e77)) if x.dtype.is_floating_point else torch.equal(x, y)


def test_sharded_pipeline_with_a_policy_in_the_loop_equals_one_batch():
    """roborugby_amd.ShardedPipeline: two actor groups, each with its own (deterministic, per-arena) policy call on its own
    stream -- the trajectories are those of one batch stepped with the same policy."""
    import roborugby_amd as rr
    n, S = 8192, 40

    def chase(obs):
        dlt = (obs[:, 1] - obs[:, 0] + 540.0) % 360.0 - 180.0
        return torch.where(dlt.abs() < 8, 0, torch.where(dlt > 0, 2, 3)).to(torch.int32)
    whole = rr.BatchedRoboRugbyEnv(n, preset="T", seed=8)
    obs = whole.reset()
    rew = torch.zeros(n, device="cuda")
    for s in range(S):
        obs, r, d, _ = whole.step(chase(obs))
        rew += r
    pipe = rr.ShardedPipeline(n, shards=2, preset="T", seed=8)
    pipe.reset()
    rews = [torch.zeros(n // 2, device="cuda") for _ in range(2)]

    def on_step(i, s, o, r, d, info):
        rews[i] += r
    last = pipe.run(lambda i, o: chase(o), S, on_step=on_step)
    pipe.synchronize()
    assert torch.equal(pipe.gather([l[0] for l in last]), obs) and torch.equal(pipe.gather(rews), rew)
    a, b = whole.get_state(), [e.get_state() for e in pipe.envs]
    assert torch.equal(a["balls"], torch.cat([b[0]["balls"], b[1]["balls"]]))
