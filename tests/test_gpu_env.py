"""GPU tests of the boundary and of size-independent properties at BASELINE.json's full size (65,536 arenas)."""
import dataclasses

import numpy as np
import pytest

import oracle_lib as ol

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _rr():
    import roborugby_amd as rr
    return rr


def _chase(obs, gen, noise=0.1):
    d = (obs[:, 1] - obs[:, 0] + 540.0) % 360.0 - 180.0
    a = torch.where(d.abs() < 8, 0, torch.where(d > 0, 2, 3)).to(torch.int32)
    r = torch.randint(0, 8, a.shape, generator=gen, device=a.device, dtype=torch.int32)
    m = torch.rand(a.shape, generator=gen, device=a.device) < noise
    return torch.where(m, r, a)


@pytest.mark.parametrize("preset", ["T", "G"])
def test_fresh_contact_rich_states_match_oracle(preset):
    """States the fixtures never saw: GPU rollouts under a chase policy with kicked balls; at checkpoints a sample
    of arenas is copied into the oracle and both take the same step."""
    rr = _rr()
    n = 2048
    env = rr.BatchedRoboRugbyEnv(n, preset=preset, seed=77, time_limit=False, auto_reset=False)
    obs = env.reset()
    st = env.get_state()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    vel = (torch.rand(n, env.preset.nb, 2, generator=gen, device="cuda", dtype=torch.float64) - 0.5) * 16
    poses_r = st["robots"][:, :, [0, 1, 6]]
    poses_b = torch.cat([st["balls"][:, :, :2], vel], dim=2)
    env.set_poses(poses_r, poses_b)
    na = env.preset.nr
    worst, checked, contacts = 0.0, 0, 0
    obs = env.get_game_state()
    for step in range(36):
        a0 = _chase(obs, gen).view(n, 1)
        acts = torch.cat([a0, torch.randint(0, 8, (n, na - 1), generator=gen, device="cuda", dtype=torch.int32)], 1) if na > 1 else a0
        check = step % 6 == 5
        if check:
            pre = {k: v.cpu().numpy() for k, v in env.get_state().items()}
        o64, r64, d, info = env.step_f64(acts)
        obs = o64.float()
        if check:
            post = {k: v.cpu().numpy() for k, v in env.get_state().items()}
            a_np, o_np, r_np = acts.cpu().numpy(), o64.cpu().numpy(), r64.cpu().numpy()
            rg_np = info.dblGrumpyScore.cpu().numpy()
            st_np = info.status.cpu().numpy()
            for a in range(0, n, 16):
                o = ol.OracleEnv(preset)
                o.set_state(pre["robots"][a], pre["robots_i"][a], pre["balls"][a], None, pre["step"][a])
                r = o.step(a_np[a])
                os_ = o.get_state()
                if r["status"] & ~256:  # the reference would have raised: only the flag is comparable
                    assert st_np[a] & r["status"] & ~256
                    continue
                assert np.array_equal(os_["robots_i"], post["robots_i"][a])
                dd = max(float(np.nanmax(np.abs(os_["robots"] - post["robots"][a]))),
                         float(np.abs(os_["balls"] - post["balls"][a]).max()),
                         float(np.abs(r["obs"] - o_np[a]).max()))
                assert dd < 1e-8 and abs(r["reward"] - r_np[a]) < 1e-6 and abs(r["reward_g"] - rg_np[a]) < 1e-6, (preset, step, a, dd)
                worst = max(worst, dd)
                checked += 1
                contacts += int(np.abs(pre["balls"][a][:, 6:]).sum() > 0)
    assert checked > 500 and contacts > 100
    print(f"[{preset}] {checked} fresh states vs oracle, worst {worst:.2e}")


def test_grumpy_and_per_robot_observations_match_oracle():
    rr = _rr()
    env = rr.BatchedRoboRugbyEnv(128, preset="G", seed=3, time_limit=False, auto_reset=False)
    env.reset()
    st = {k: v.cpu().numpy() for k, v in env.get_state().items()}
    cases = [(1, -1, -1), (-1, -1, -1), (1, 1, 2), (-1, 3, 5), (1, 0, 6), (-1, 2, 0)]
    got = {c: env.get_game_state(c[0], c[1], c[2], f64=True).cpu().numpy() for c in cases}
    for a in range(0, 128, 7):
        o = ol.OracleEnv("G")
        o.set_state(st["robots"][a], st["robots_i"][a], st["balls"][a], None, 0)
        for c in cases:
            assert np.allclose(o.observe(*c), got[c][a], atol=1e-9, rtol=0), (a, c)
    envT = rr.BatchedRoboRugbyEnv(4, preset="T")
    assert envT.get_game_state(int_team=-1) is None  # no grumpy robot: the reference returns None (Obs:306-307)


def test_thrust_entry_matches_oracle():
    rr = _rr()
    n = 64
    env = rr.BatchedRoboRugbyEnv(n, preset="G", seed=8, time_limit=False, auto_reset=False)
    env.reset()
    st = {k: v.cpu().numpy() for k, v in env.get_state().items()}
    gen = torch.Generator(device="cuda")
    gen.manual_seed(2)
    thrust = (torch.rand(n, 8, generator=gen, device="cuda") * 4 - 2).round(decimals=1)
    thrust[0] = torch.tensor([0.5, 1.5, -0.5, 2.5, 0.49, -1.5, 1.0, 0.0])
    obs, rew, done, info = env.step_thrust(thrust)
    post = {k: v.cpu().numpy() for k, v in env.get_state().items()}
    th = thrust.cpu().numpy().astype(np.float64)
    for a in range(n):
        o = ol.OracleEnv("G")
        o.set_state(st["robots"][a], st["robots_i"][a], st["balls"][a], None, 0)
        r = o.step_thrust(th[a])
        assert np.array_equal(o.get_state()["robots_i"], post["robots_i"][a]), a
        assert np.allclose(o.get_state()["robots"][:, :7], post["robots"][a][:, :7], atol=1e-9, rtol=0)
        assert np.allclose(r["obs"], obs[a].cpu().numpy(), atol=1e-3, rtol=0)


def test_time_limit_auto_reset_and_episode_stats_full_size():
    """Preset T at 65,536 arenas through a whole episode: done fires for every arena exactly at step T
    (TimeLimit rule), the logged return equals the sum of rewards, the next call re-places every arena."""
    rr = _rr()
    n = 65536
    p = dataclasses.replace(rr.PRESETS["T"], game_len_steps=40)
    env = rr.BatchedRoboRugbyEnv(n, preset=p, seed=1, time_limit=True, auto_reset=True)
    env.reset()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0)
    total = torch.zeros(n, device="cuda", dtype=torch.float64)
    for s in range(40):
        a = torch.randint(0, 8, (n,), generator=gen, device="cuda", dtype=torch.int32)
        obs, rew, done, info = env.step(a)
        total += rew.double()
        assert bool(done.all()) == (s == 39) and bool(done.any()) == (s == 39)
    lr, lrg, ll, cnt = env.episode_stats()
    assert int(cnt.min()) == 1 and int(cnt.max()) == 1 and int(ll.min()) == 40 and int(ll.max()) == 40
    assert torch.allclose(lr.double(), total, atol=1e-2, rtol=1e-5)
    obs, rew, done, info = env.step(torch.zeros(n, dtype=torch.int32, device="cuda"))
    assert bool((info.status & 1024).bool().all()) and not bool(done.any()) and float(rew.abs().max()) == 0.0
    st = env.get_state()
    assert int(st["step"].max()) == 0
    assert torch.isnan(st["robots"][:, :, 7:]).all()  # pose history cleared like Robot.on_reset
    obs2, _, _, info2 = env.step(torch.zeros(n, dtype=torch.int32, device="cuda"))
    assert int(env.get_state()["step"].min()) == 1 and not bool((info2.status & 1024).any())


def test_determinism_and_shard_invariance_full_size():
    """Same seed -> bit-identical rollouts; a shard created with arena_offset reproduces the matching slice of the
    big batch (what the 8-GPU run relies on: no cross-arena state, RNG keyed by the global arena id)."""
    rr = _rr()
    n, half = 65536, 32768
    gen = torch.Generator(device="cuda")
    gen.manual_seed(9)
    acts = torch.randint(0, 8, (6, n, 4), generator=gen, device="cuda", dtype=torch.int32)

    def run(num, offset, sl):
        env = rr.BatchedRoboRugbyEnv(num, preset="G", seed=123, arena_offset=offset, reset_on_fault=False)
        outs = [env.reset()]
        for k in range(6):
            o, r, d, info = env.step(acts[k][sl])
            outs += [o, r, info.adblGrumpyState, info.dblGrumpyScore]
        st = env.get_state()
        env.close()
        return outs, st

    big, st_big = run(n, 0, slice(0, n))
    big2, _ = run(n, 0, slice(0, n))
    for x, y in zip(big, big2):
        assert torch.equal(x, y)
    shard, st_sh = run(half, half, slice(half, n))
    for x, y in zip(big, shard):
        assert torch.equal(x[half:], y)
    assert torch.equal(st_big["balls"][half:], st_sh["balls"])
    # sanity invariants at full size
    rb = st_big["robots"]
    assert torch.isfinite(rb[:, :, :7]).all() and torch.isfinite(st_big["balls"]).all()
    assert float(rb[:, :, 2].min()) >= 0 and float(rb[:, :, 3].max()) <= 800 and float(rb[:, :, 4].min()) > 0 and float(rb[:, :, 5].max()) < 800
    assert int(st_big["step"].min()) == 6 and int(st_big["step"].max()) == 6
    obs = big[-4]
    assert float(obs.abs().max()) <= 800 and float(obs[:, 5:].max()) <= 150 and float(obs[:, 5:].min()) >= 0


def test_single_env_wrapper_behaves_like_the_reference():
    rr = _rr()
    env = rr.make("RoboRugbySimpleDuel-v3", preset="T", time_limit=False)
    assert env.observation_space.shape == (11,) and env.action_space.n == 8
    assert env.spec.max_episode_steps == 300 and env.metadata["video.frames_per_second"] == 30
    obs = env.reset()
    assert isinstance(obs, np.ndarray) and obs.shape == (11,)
    o, r, d, info = env.step([rr.Direction.FORWARD])
    assert isinstance(r, float) and isinstance(d, bool) and info.adblGrumpyState is None and info.dblGrumpyScore == 0.0
    assert np.allclose(env.unwrapped.get_game_state(), o)
    with pytest.raises(Exception, match="commands but only 1 robots"):
        env.step([0, 1])
    for _ in range(300):
        o, r, d, info = env.step([0])
    assert d
    with pytest.raises(Exception, match="Game is over"):
        env.step([0])


def test_f32_handle_refuses_f64_outputs():
    """(fp32-mode parity lives in tests/test_gpu_fp32.py: reference goldens + the fp64 oracle.)"""
    rr = _rr()
    e32 = rr.BatchedRoboRugbyEnv(64, preset="G", seed=4, dtype="f32")
    e32.reset()
    with pytest.raises(Exception, match="RR_DTYPE_F64"):
        e32.step_f64(torch.zeros(64, 4, dtype=torch.int32, device="cuda"))


def test_reset_on_fault_replaces_faulted_arenas():
    """Arenas whose spawn leaves a ball jammed against a robot fault on every step in the reference (it raises /
    spins forever).  With reset_on_fault they are reported done once and re-placed; without it they linger."""
    rr = _rr()
    n = 65536
    a = torch.full((n, 4), 8, dtype=torch.int32, device="cuda")  # invalid action: nobody moves
    keep = rr.BatchedRoboRugbyEnv(n, preset="G", seed=0, reset_on_fault=False)
    keep.reset()
    for _ in range(3):
        _, _, d_keep, i_keep = keep.step(a)
    lingering = int(((i_keep.status & 63) != 0).sum())
    assert lingering > 0 and not bool(d_keep.any())
    env = rr.BatchedRoboRugbyEnv(n, preset="G", seed=0)  # default: reset_on_fault follows auto_reset
    env.reset()
    _, _, d1, i1 = env.step(a)
    f1 = (i1.status & 63) != 0  # fatal bits: the reference raised or hangs
    assert int(f1.sum()) == lingering and torch.equal(d1, f1)
    _, _, d2, i2 = env.step(a)
    assert bool(((i2.status & 1024) != 0)[f1].all()) and not bool(d2[f1].any())
    for _ in range(4):
        _, _, d3, i3 = env.step(a)
    assert int(((i3.status & 63) != 0).sum()) < max(2, lingering // 4)


@pytest.mark.parametrize("preset,n", [("T", 1), ("T", 33), ("T", 1001), ("G", 1), ("G", 7), ("G", 9), ("G", 1001)])
def test_ragged_batch_sizes_match_oracle(preset, n):
    """Batch sizes that leave the last wavefront partly empty (several arenas share a wavefront): every arena still
    matches the oracle and nothing outside the batch is touched."""
    rr = _rr()
    env = rr.BatchedRoboRugbyEnv(n, preset=preset, seed=21, time_limit=False, auto_reset=False)
    env.reset()
    na = env.preset.nr
    gen = torch.Generator(device="cuda")
    gen.manual_seed(n)
    for _ in range(3):
        pre = {k: v.cpu().numpy() for k, v in env.get_state().items()}
        acts = torch.randint(0, 8, (n, na), generator=gen, device="cuda", dtype=torch.int32)
        o, r, d, info = env.step_f64(acts)
        post = {k: v.cpu().numpy() for k, v in env.get_state().items()}
        for a in sorted(set([0, n - 1, n // 2] + list(range(0, n, 97)))):
            orc = ol.OracleEnv(preset)
            orc.set_state(pre["robots"][a], pre["robots_i"][a], pre["balls"][a], None, pre["step"][a])
            res = orc.step(acts[a].cpu().numpy())
            if res["status"] & 63:
                continue
            os_ = orc.get_state()
            assert np.allclose(os_["balls"], post["balls"][a], atol=1e-9, rtol=0)
            assert np.allclose(os_["robots"][:, :7], post["robots"][a][:, :7], atol=1e-9, rtol=0)
            assert np.allclose(res["obs"], o[a].cpu().numpy(), atol=1e-9, rtol=0)


def test_abi_rejects_misuse_without_crashing():
    rr = _rr()
    from roborugby_amd._lib import RRError
    env = rr.BatchedRoboRugbyEnv(8, preset="T")
    with pytest.raises(Exception, match="commands but only 1 robots"):
        env.step(torch.zeros(8, 2, dtype=torch.int32, device="cuda"))
    with pytest.raises(RRError, match="bad team/robot/ball index"):
        env.get_game_state(int_team=1, robot_idx=3)
    import ctypes as C
    import dataclasses
    from roborugby_amd import _lib
    # entity counts the contact masks cannot hold are refused before anything is compiled (other counts outside the built shapes get a
    # one-shape library on demand: tests/test_custom_shape.py) ...
    with pytest.raises(ValueError, match="contact masks"):
        rr.BatchedRoboRugbyEnv(8, preset=dataclasses.replace(rr.PRESETS["G"], nr_happy=5, nr_grumpy=4))
    # ... and the main library itself refuses counts it was not built for
    p = rr.PRESETS["T"]
    cfg = _lib.RRConfig(struct_size=C.sizeof(_lib.RRConfig), num_envs=8, nr_happy=1, nr_grumpy=0, nb_pos=2, nb_neg=0, arena_w=p.arena_w,
                        arena_h=p.arena_h, game_len_steps=p.game_len_steps, game_mode=0, time_limit=1, auto_reset=1, reset_on_fault=1,
                        dtype=0, device=0, seed=1, arena_offset=0, step_budget_clocks=0, reserved_=0)
    h = C.c_void_p()
    with pytest.raises(RRError, match="unsupported entity counts"):
        _lib.check(_lib.load().rr_create(C.byref(cfg), C.byref(h)), "rr_create", _lib.load())
    with pytest.raises(RRError, match="num_envs must be positive"):
        rr.BatchedRoboRugbyEnv(0, preset="T")
    # bad actions are a per-arena status, not an error: the arena keeps its previous thrust (KeyError in the reference)
    obs, rew, done, info = env.step(torch.full((8,), 9, dtype=torch.int32, device="cuda"))
    assert bool(((info.status & 128) != 0).all())
