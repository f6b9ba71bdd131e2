"""Entity counts outside the library's built shapes (the reference's counts are free integers, RR_Constants.py:30-34): a one-shape
library of the same sources and ABI, compiled on demand (roborugby_amd/build.py: build_shape_library).  Preset X = 2 + 1 robots,
2 + 3 balls is the one pinned to reference vectors (tests/golden/{traj,reset}_X.npz: oracle bit for bit in test_oracle_traj.py, the
kernel's phase source in test_emulated_wave.py / test_differential_adversarial.py, the GPU in test_gpu_parity.py); here: the build
helper itself (CPU) and what is specific to such a library on the GPU."""
import os

import numpy as np
import pytest

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shape_limits_and_lane_widths():
    from roborugby_amd import build
    assert build.shape_lanes(2, 1, 2, 3) == 8 and build.shape_lanes(1, 0, 1, 0) == 2 and build.shape_lanes(2, 2, 4, 4) == 8
    assert build.shape_lanes(1, 1, 2, 1) == 4 and build.shape_lanes(4, 4, 2, 2) == 8 and build.shape_lanes(2, 0, 6, 5) == 16
    for bad in ((0, 1, 1, 1), (1, 0, 0, 1), (5, 4, 1, 1), (2, 2, 6, 6), (3, 3, 3, 3)):  # no happy robot / no positive ball / > 8 robots / > 11 balls / > 32 pairs
        with pytest.raises(ValueError):
            build.shape_lanes(*bad)
    assert (2, 2, 4, 4) in build.BUILT_SHAPES and (2, 1, 2, 3) not in build.BUILT_SHAPES


def test_one_shape_library_builds_and_exports_the_whole_abi():
    """hipcc cross-compiles it without a GPU (what __graft_entry__.build() does for preset X); every declared symbol is there."""
    import ctypes as C
    from roborugby_amd import _lib, build
    path = build.build_shape_library(2, 1, 2, 3)
    assert os.path.exists(path) and not build.shape_is_stale(2, 1, 2, 3)
    lib = C.CDLL(path)
    for name in _lib.SYMBOLS:
        assert hasattr(lib, name), name
    assert lib.rr_abi_version() == C.CDLL(build.LIB).rr_abi_version()


@pytest.mark.gpu
def test_custom_shape_surface_on_the_gpu():
    """Preset X through the gym mirror: spaces follow the counts, the library refuses other counts and the all-fp32 dtype, the exact
    shortcuts are exact (on == off over a contact-dense batch), both teams observe, fp32-state records work, rollout == steps."""
    import torch
    import roborugby_amd as rr
    import adversarial as adv
    from test_gpu_shortcuts import _env, _run, _same
    p = rr.PRESETS["X"]
    env = rr.BatchedRoboRugbyEnv(512, preset="X", seed=5)
    assert env.lanes_per_env() == 8 and env.preset.nr == 3 and env.preset.nb == 5
    obs = env.reset()
    assert obs.shape == (512, 11)
    a = torch.randint(0, 8, (512, 3), dtype=torch.int32, device="cuda")
    o, r, d, info = env.step(a)
    assert info.adblGrumpyState is not None and bool(torch.isfinite(o).all())
    # S steps in one launch == S single steps
    st = env.get_state()
    acts = torch.randint(0, 8, (6, 512, 3), dtype=torch.int32, device="cuda")
    ro, rr_, rd, _ = env.rollout(acts)
    env2 = rr.BatchedRoboRugbyEnv(512, preset="X", seed=5)
    env2.set_state(st["robots"], st["robots_i"], st["balls"], st["step"])
    for k in range(6):
        o2, r2, d2, _ = env2.step(acts[k])
        assert torch.equal(o2, ro[k]) and torch.equal(r2, rr_[k])
    # the one-shape library is for THIS shape and the two fp64-arithmetic precisions
    with pytest.raises(ValueError):
        rr.BatchedRoboRugbyEnv(64, preset="X", dtype="f32")
    e32 = rr.BatchedRoboRugbyEnv(256, preset="X", dtype="f32_state", seed=1)
    e32.reset()
    o32, _, _, _ = e32.step_f64(torch.randint(0, 8, (256, 3), dtype=torch.int32, device="cuda"))
    assert bool(torch.isfinite(o32[:, :6]).all()) and e32.state_bytes_per_env() < env.state_bytes_per_env()
    # exact shortcuts on == off on contact-dense states of this shape
    n = 1500
    robots, balls, actions = adv.make_states("X", n, seed=21)
    e = rr.BatchedRoboRugbyEnv(n, preset="X", time_limit=False, auto_reset=False)
    e.set_poses(robots, balls)
    s0 = e.get_state()
    state = tuple(s0[k].cpu().numpy() for k in ("robots", "robots_i", "balls", "step"))
    a_t = torch.as_tensor(actions, device="cuda")
    on = _run(_env(n, "X"), state, a_t, 4)
    off = _run(_env(n, "X", RR_NO_MEMO=1, RR_NO_ORDER=1), state, a_t, 4)
    assert _same(on, off)
    # a custom preset object with other counts resolves to its own library name (not built here: that takes a minute of hipcc)
    from roborugby_amd import build
    from roborugby_amd.config import custom_preset
    q = custom_preset(1, 1, 2, 1)
    assert (q.nr, q.nb) == (2, 3) and build.shape_lib_path(1, 1, 2, 1).endswith("libroborugby_amd_1x1_2x1.so")


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_on_demand_compile_of_a_four_lane_shape_and_parity_with_the_oracle():
    """1 + 1 robots, 2 + 1 balls: no library for it ships -- BatchedRoboRugbyEnv compiles one on the spot (hipcc on the GPU box, about
    half a minute), four lanes per arena: the narrow phases' "fewer than eight lanes" variants with three balls, a combination none
    of the built shapes has.  Contact-dense states against the oracle (1e-9, equal fault bits), reset draw for draw."""
    import torch
    import roborugby_amd as rr
    import adversarial as adv
    from roborugby_amd.config import custom_preset
    from test_differential_adversarial import _compare
    p = custom_preset(1, 1, 2, 1, name="Y")
    n = 1500
    robots, balls, actions = adv.make_states("Y", n, seed=9)
    env = rr.BatchedRoboRugbyEnv(n, preset=p, time_limit=False, auto_reset=False)
    assert env.lanes_per_env() == 4
    env.set_poses(robots, balls)
    o, r, d, info = env.step_f64(torch.as_tensor(actions))
    st = {k: v.cpu().numpy() for k, v in env.get_state().items()}
    o, r, d, status = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy(), info.status.cpu().numpy()
    ok = faults = knife = 0
    for a in range(0, n, 3):
        res, ost = adv.oracle_step("Y", robots[a], balls[a], actions[a])
        g_res = dict(status=int(status[a]) & 0xFFFF, obs=o[a], reward=float(r[a]), done=bool(d[a]))
        g_st = {k: st[k][a] for k in ("robots", "robots_i", "balls")}
        v = _compare("Y", res, ost, g_res, g_st, ("Y", a), (robots[a], balls[a], actions[a]))
        ok += v == "ok"; faults += v == "fault"; knife += v == "knife"
    assert ok > 0.5 * (n // 3) and knife <= 0.05 * (n // 3), (ok, faults, knife)
    env2 = rr.BatchedRoboRugbyEnv(65, preset=p, seed=1234, arena_offset=1000)
    env2.reset()
    s2 = {k: v.cpu().numpy() for k, v in env2.get_state().items()}
    for a in (0, 1, 64):
        orc = ol.OracleEnv("Y")
        orc.reset(1234, 1000 + a, 0); orc.reset(1234, 1000 + a, 1)
        os_ = orc.get_state()
        assert np.array_equal(os_["robots"][:, [0, 1, 6]], s2["robots"][a][:, [0, 1, 6]]) and np.array_equal(os_["balls"], s2["balls"][a]), a
    print(f"[Y = 1+1 / 2+1, compiled on demand] {ok} adversarial states match the oracle, {faults} faulted identically, {knife} knife-edge")
