"""The kernel's phase source (roborugby_amd/csrc/rr_sim.hpp) compiled by g++ as a lane loop, against the golden
vectors of the reference and against the oracle.  This exercises on CPU exactly the code hipcc compiles for
gfx950 -- lane->task maps, ballot masks, list-order responses, undo bookkeeping -- so a logic error shows up
without a GPU.  (The emulation is a test harness, never a product path.)"""
import numpy as np
import pytest

import emu_lib as el
import oracle_lib as ol

TOL = 1e-9


def _diff(st, ref_r, ref_b):
    assert np.array_equal(np.isnan(st["robots"]), np.isnan(ref_r))
    return max(float(np.nanmax(np.abs(st["robots"] - ref_r))), float(np.abs(st["balls"] - ref_b).max()))


@pytest.mark.parametrize("preset,stride,narrow", [("T", 3, False), ("G", 4, False), ("T", 4, True), ("G", 5, True), ("D", 3, False), ("D", 4, True),
                                                       ("X", 5, False), ("X", 6, True)])
def test_emulated_kernel_vs_reference_golden(golden_dir, preset, stride, narrow):
    """narrow=True runs the phases with VW = 4 (T) / 16 (G) lanes per arena, i.e. in several rounds -- the
    lane->task maps of the packed GPU builds (several arenas per wavefront)."""
    t = np.load(f"{golden_dir}/traj_{preset}.npz")
    env = el.EmuEnv(preset, narrow=narrow)
    worst, n = 0.0, 0
    for ep in range(t["length"].shape[0]):
        for s in range(ep % stride, int(t["length"][ep]), stride):
            env.set_state(t["state_robots"][ep, s], t["state_robots_i"][ep, s], t["state_balls"][ep, s], t["state_step"][ep, s])
            a = t["actions"][ep, s]
            r = env.step(a[a >= 0])
            st = env.get_state()
            d = _diff(st, t["state_robots"][ep, s + 1], t["state_balls"][ep, s + 1])
            d = max(d, float(np.abs(r["obs"] - t["obs"][ep, s]).max()))
            assert np.array_equal(st["robots_i"], t["state_robots_i"][ep, s + 1]), (ep, s)
            assert d < TOL and abs(r["reward"] - t["reward"][ep, s]) < 1e-7, (preset, ep, s, d)
            assert r["done"] == bool(t["done"][ep, s])
            assert (r["status"] & ~256) == 0
            worst = max(worst, d)
            n += 1
    assert n > 500
    print(f"[{preset}] {n} golden steps through the emulated wave, worst {worst:.2e}")


@pytest.mark.parametrize("preset", ["T", "G", "D", "X"])
def test_emulated_kernel_thrust_entry_vs_reference_golden(golden_dir, preset):
    """The kernel's continuous-thrust entry (rr_step_thrust's path in step_arena) against the reference's own
    GameEnv.step vectors (tests/golden/thrust_*.npz: half-way rounding cases, |thrust| up to 3, fewer pairs than robots)."""
    t = np.load(f"{golden_dir}/thrust_{preset}.npz")
    env = el.EmuEnv(preset)
    worst, n = 0.0, 0
    for ep in range(t["length"].shape[0]):
        for s in range(ep % 2, int(t["length"][ep]), 2):
            env.set_state(t["state_robots"][ep, s], t["state_robots_i"][ep, s], t["state_balls"][ep, s], t["state_step"][ep, s])
            th = t["thrust"][ep, s]
            th = th[~np.isnan(th[:, 0])]
            assert np.array_equal(np.rint(th.astype(np.float32)), np.rint(th))  # the C-ABI takes float32 thrust
            r = env.step_thrust(th)
            st = env.get_state()
            d = _diff(st, t["state_robots"][ep, s + 1], t["state_balls"][ep, s + 1])
            d = max(d, float(np.abs(r["obs"] - t["obs"][ep, s]).max()))
            assert np.array_equal(st["robots_i"], t["state_robots_i"][ep, s + 1]), (ep, s)
            assert d < TOL and abs(r["reward"] - t["reward"][ep, s]) < 1e-7, (preset, ep, s, d)
            assert r["naughty"] == t["naughty"][ep, s] and r["done"] == bool(t["done"][ep, s])
            assert (r["status"] & ~256) == 0
            worst = max(worst, d)
            n += 1
    assert n > 250
    print(f"[{preset}] {n} golden thrust steps through the emulated wave, worst {worst:.2e}")


@pytest.mark.parametrize("preset", ["T", "G", "D"])
def test_emulated_reset_equals_oracle_reset(preset):
    for arena in (0, 5, 123456789):
        e, o = el.EmuEnv(preset, seed=42), ol.OracleEnv(preset)
        for episode in (0, 1, 2):
            e.reset(arena, episode)
            o.reset(42, arena, episode)
            se, so = e.get_state(), o.get_state()
            # integer draws -> centres/rotations identical; edges go through sin/cos/sqrt (libm vs pow ulps)
            assert np.array_equal(se["robots"][:, [0, 1, 6]], so["robots"][:, [0, 1, 6]])
            assert np.allclose(se["robots"][:, :7], so["robots"][:, :7], atol=1e-11, rtol=0)
            assert np.array_equal(se["balls"], so["balls"])
            assert np.isnan(se["robots"][:, 7:]).all()
            assert np.allclose(e.observe(1), o.observe(1), atol=1e-10, rtol=0)


def test_fault_paths_match_oracle_flags(golden_dir):
    """Where the reference raised (exc != 0 in the fixtures) the kernel source flags the same status bit."""
    seen = 0
    for preset in ("T", "G"):
        t = np.load(f"{golden_dir}/traj_{preset}.npz")
        for ep in range(t["length"].shape[0]):
            exc, n = int(t["exc"][ep]), int(t["length"][ep])
            if not exc:
                continue
            env = el.EmuEnv(preset)
            env.set_state(t["state_robots"][ep, n], t["state_robots_i"][ep, n], t["state_balls"][ep, n], t["state_step"][ep, n])
            a = t["actions"][ep, n]
            r = env.step(a[a >= 0])
            assert r["status"] & exc, (preset, ep, r["status"], exc)
            seen += 1
    assert seen >= 2


def test_thrust_rounding_is_bankers(golden_dir):
    """set_thrust = int(round(x)) (RR_Robot.py:100-102): 0.5 -> 0, 1.5 -> 2, -0.5 -> 0, 0.51 -> 1."""
    e, o = el.EmuEnv("T"), ol.OracleEnv("T")
    for thrust in ([0.5, 1.0], [1.5, 1.49], [-0.5, 0.51], [2.5, -2.5], [0.49, -0.49]):
        e.set_poses([[300, 300, 30]], [[100, 100, 0, 0]])
        o.set_clean_state([[300, 300, 30]], [[100, 100, 0, 0]])
        re_, ro = e.step_thrust(thrust), o.step_thrust(thrust)
        assert np.array_equal(e.get_state()["robots_i"], o.get_state()["robots_i"]), thrust
        assert np.allclose(e.get_state()["robots"][:, :7], o.get_state()["robots"][:, :7], atol=1e-10, rtol=0)
        assert np.allclose(re_["obs"], ro["obs"], atol=1e-9, rtol=0)


def test_time_limit_and_auto_reset_semantics():
    cfg = ol.PRESETS["T"]
    # raw rule: done on step T+1; TimeLimit rule: done on step T
    for tl, first_done in ((0, cfg["game_len"] + 1), (1, cfg["game_len"])):
        e = el.EmuEnv("T", time_limit=tl, auto_reset=1, seed=9)
        e.reset(0, 0)
        st = e.get_state()
        e.set_state(st["robots"], st["robots_i"], st["balls"], first_done - 2)
        r = e.step([0])
        assert not r["done"]
        r = e.step([0])
        assert r["done"] and e.get_state()["step"] == first_done
        r = e.step([0])  # auto-reset: this call re-places the arena instead of stepping it
        assert r["status"] & 1024 and not r["done"] and r["reward"] == 0.0
        assert e.get_state()["step"] == 0
        r = e.step([0])
        assert e.get_state()["step"] == 1 and not (r["status"] & 1024)
    e = el.EmuEnv("T", time_limit=0, auto_reset=0)
    e.reset(0, 0)
    st = e.get_state()
    e.set_state(st["robots"], st["robots_i"], st["balls"], cfg["game_len"] + 1)
    r = e.step([0])
    assert r["status"] & 64 and r["done"]  # "Game is over. Go home." (RR_EnvBase.py:261-262)


@pytest.mark.parametrize("preset,stride", [("T", 2), ("G", 3), ("D", 3)])
def test_f32_mode_single_step_vs_reference_golden(golden_dir, preset, stride):
    """fp32 fast mode, one step from synchronised state on the reference's golden steps, scored like the GPU test
    (tests/fp32_checks.py, tests/test_gpu_fp32.py): quiet steps within 1e-5 relative with integer state exact, contact
    steps inside the documented distribution.  (Host emulation: glibc's sinf/cosf; the GPU test runs ocml's.)"""
    import fp32_checks as fc
    t = np.load(f"{golden_dir}/traj_{preset}.npz")
    cfg = ol.PRESETS[preset]
    idx = [(ep, s) for ep in range(t["length"].shape[0]) for s in range(ep % stride, int(t["length"][ep]), stride)]
    ep = np.array([i[0] for i in idx]); s = np.array([i[1] for i in idx])
    pre = {k: t["state_" + k][ep, s] for k in ("robots", "robots_i", "balls", "step")}
    post = {k: t["state_" + k][ep, s + 1] for k in ("robots", "robots_i", "balls")}
    env = el.EmuEnv(preset, f32=True)
    got = {k: [] for k in post}
    for i in range(len(idx)):
        env.set_state(pre["robots"][i], pre["robots_i"][i], pre["balls"][i], pre["step"][i])
        a = t["actions"][ep[i], s[i]]
        r = env.step(a[a >= 0])
        assert r["done"] == bool(t["done"][ep[i], s[i]])
        st = env.get_state()
        for k in got:
            got[k].append(st[k])
    q, e, ints = fc.score(pre, post, {k: np.array(v) for k, v in got.items()}, cfg["W"], cfg["H"])
    c = ~q
    assert q.sum() > 100 and c.sum() > 500
    assert e[q].max() <= 1e-5 and ints[q].all(), float(e[q].max())
    assert np.median(e[c]) < 1e-5 and np.percentile(e[c], 90) < 3e-4 and (e[c] > 1e-2).mean() < 0.025 and ints[c].mean() > 0.995
    print(f"[{preset}] fp32 emulation: {int(q.sum())} quiet steps max {e[q].max():.2e}; {int(c.sum())} contact steps median "
          f"{np.median(e[c]):.2e} p90 {np.percentile(e[c], 90):.2e}, {100 * (e[c] > 1e-2).mean():.2f} % > 1e-2")


def test_reset_on_fault_ends_the_episode(golden_dir):
    """Where the reference raised inside step(), reset_on_fault reports done=True and the next call re-places the arena."""
    t = np.load(f"{golden_dir}/traj_T.npz")
    eps = [ep for ep in range(t["length"].shape[0]) if int(t["exc"][ep])]
    tg = np.load(f"{golden_dir}/traj_G.npz")
    epg = [ep for ep in range(tg["length"].shape[0]) if int(tg["exc"][ep])]
    assert epg
    ep = epg[0]
    n = int(tg["length"][ep])
    a = tg["actions"][ep, n]
    a = a[a >= 0]
    for rof, want_done in ((0, False), (1, True)):
        e = el.EmuEnv("G", auto_reset=1, reset_on_fault=rof, seed=3)
        e.set_state(tg["state_robots"][ep, n], tg["state_robots_i"][ep, n], tg["state_balls"][ep, n], tg["state_step"][ep, n])
        r = e.step(a)
        assert r["status"] & int(tg["exc"][ep]) and r["done"] == want_done
        r2 = e.step(a)
        assert bool(r2["status"] & 1024) == want_done  # re-placed only under the policy
