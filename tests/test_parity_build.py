"""The parity build (-DRR_EXACT_TRIG=1 -> RR_CARRY: libroborugby_amd_exact.so; here the host-emulated wave compiled the same way):
exact trigonometry AND the scratch-rect carry.

The reference keeps one module-global FloatRect, `_rectBallInner` (RR_TrashyPhysics.py:26-36), and puts it on a ball with
`_rectBallInner.center = ball.center` -- relative moves (MyUtils.py:266-275), so the centre its diameters are built from is
fl(c_old + fl(x - c_old)): state carried from pair to pair, sweep to sweep, step to step.  The default build puts the rect exactly on
the ball; the parity build reproduces the carry (rr_sim.hpp "scratch-rect carry"; the golden trajectories dump the rect's centre as
`state_inner`, rr_set_scratch_rect seeds it).  With both, what is left between a free-running episode and the reference is the
reference's libm: glibc's sin / cos / pow are < 1 ulp but not always the nearest double.  Asserted here:
  * against the golden episodes (glibc): whole episodes bit for bit -- T 14 of 16, G 12 of 15, D 9 of 11 (default build: 4 / 2 / 1);
  * against the oracle evaluated with a correctly rounded libm (rro_debug_attribution(4)): ALL 42 episodes bit for bit to their end;
  * on stuck arenas -- frozen islands, fixed points, thaws: where the carry has to survive the shortcuts -- the same oracle, bit for
    bit, with the shortcuts firing; and the shortcut-equivalence suites re-run against the parity build."""
import os
import subprocess
import sys

import numpy as np
import pytest

import emu_lib as el
import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, "tools"))
import attribute_divergence as ad  # noqa: E402


def _same(a, b):
    return (np.array_equal(a["robots"], b["robots"], equal_nan=True) and np.array_equal(a["balls"], b["balls"])
            and np.array_equal(a["robots_i"], b["robots_i"]))


def _full_episodes(golden_dir, preset):
    t = dict(np.load(f"{golden_dir}/traj_{preset}.npz"))
    t["_preset"] = preset
    na_used = (t["actions"][:, 0, :] >= 0).sum(1)
    return t, np.nonzero(na_used == na_used.max())[0], int(na_used.max())


@pytest.mark.parametrize("preset,want", [("T", (4, 14)), ("G", (2, 12)), ("D", (1, 9))])
def test_parity_build_follows_whole_golden_episodes_bit_for_bit(golden_dir, preset, want):
    t, full, na = _full_episodes(golden_dir, preset)
    fast = [ad.free_run(lambda: el.EmuEnv(preset, exact=False), t, ep, na) for ep in full]
    exact = [ad.free_run(lambda: el.EmuEnv(preset, exact=True), t, ep, na) for ep in full]
    n_fast, n_exact = sum(x is None for x in fast), sum(x is None for x in exact)
    print(f"[{preset}] free-running golden episodes bit-identical to the reference to their last step: default build {n_fast}, parity build "
          f"{n_exact} of {len(full)}; first departures (parity build): {[x for x in exact if x is not None]}")
    assert (n_fast, n_exact) == want  # the counts of round 3, on the emulation and on the MI355X alike: exact


@pytest.mark.parametrize("preset", ["T", "G", "D"])
def test_parity_build_equals_the_reference_arithmetic_under_a_correctly_rounded_libm(golden_dir, preset):
    """every free-running golden episode, parity build and oracle side by side from the reference's step-0 state (scratch rect
    included), the oracle's sin / cos / pow correctly rounded: bit-identical state after every step of every episode"""
    t, full, na = _full_episodes(golden_dir, preset)
    steps = 0
    for ep in full:
        ol.lib().rro_debug_attribution(4)
        try:
            o = ol.OracleEnv(preset)
            o.set_state(t["state_robots"][ep, 0], t["state_robots_i"][ep, 0], t["state_balls"][ep, 0], t["state_inner"][ep, 0], int(t["state_step"][ep, 0]))
            e = el.EmuEnv(preset, exact=True)
            e.set_state(t["state_robots"][ep, 0], t["state_robots_i"][ep, 0], t["state_balls"][ep, 0], step=int(t["state_step"][ep, 0]))
            assert e.set_scratch_rect(t["state_inner"][ep, 0])
            for s in range(int(t["length"][ep])):
                a = np.clip(t["actions"][ep, s, :na], 0, 7).astype(np.int32)
                o.step(a)
                e.step(a)
                assert _same(o.get_state(), e.get_state()), (preset, ep, s)
                steps += 1
            assert np.array_equal(o.get_state()["inner"][:2], e.get_scratch_rect()), (preset, ep)
        finally:
            ol.lib().rro_debug_attribution(0)
    print(f"[{preset}] {len(full)} episodes, {steps} free-running steps: parity build == reference arithmetic with a correctly rounded libm, bit for bit")


@pytest.mark.parametrize("fixture", ["stuck_islands_G.npz", "stuck_islands_wall_G.npz"])
def test_parity_build_carry_survives_the_shortcuts_on_stuck_arenas(fixture):
    """Stuck arenas from the slowest wavefronts of a GPU rollout (tests/data): 4 steps each with the shortcuts on, against the
    oracle (correctly rounded libm) started from the same state with the scratch rect on the last ball."""
    import test_fixed_point_memo as fpm
    d = np.load(os.path.join(HERE, "data", fixture))
    fired = {"E freeze": 0, "fixed point": 0, "thaw": 0}
    for a in range(len(d["step"])):
        state = (d["robots"][a], d["robots_i"][a], d["balls"][a])
        inner = np.array([d["balls"][a][-1, 0], d["balls"][a][-1, 1], 0.0])

        def run(count=False):
            ol.lib().rro_debug_attribution(4)
            try:
                o = ol.OracleEnv("G")
                o.set_state(*state, inner, int(d["step"][a]))
                e = el.EmuEnv("G", exact=True)
                e.set_state(*state, step=int(d["step"][a]))
                for k in range(4):
                    o.step(d["actions"][a])
                    e.step(d["actions"][a])
                    if not count:
                        assert _same(o.get_state(), e.get_state()), (fixture, a, k)
            finally:
                ol.lib().rro_debug_attribution(0)
        run()
        el_default, el.DEFAULT_EXACT = el.DEFAULT_EXACT, True  # (the trace switch of _count_events goes to the parity build's emulator)
        try:
            c = fpm._count_events(lambda: run(True), tuple(fired))
        finally:
            el.DEFAULT_EXACT = el_default
        for k in fired:
            fired[k] += c[k]
    print(fixture, fired)
    assert fired["E freeze"] >= 10, fired


def test_shortcut_equivalence_suites_pass_against_the_parity_build():
    """tests/test_fixed_point_memo.py and tests/test_budgeted_step.py (shortcuts on == off, budgeted == synchronous, bit for bit)
    with every emulated env built as the parity build"""
    env = dict(os.environ, RR_EMU_EXACT="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", os.path.join(HERE, "test_fixed_point_memo.py"),
                        os.path.join(HERE, "test_budgeted_step.py")], env=env, cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("preset", ["T", "G", "D"])
def test_parity_build_carries_the_scratch_rect_across_resets(preset):
    """A reset moves the balls, not the scratch rect (RR_EnvBase.py:125-216 never touches RR_TrashyPhysics): three episodes of 40 random
    steps each with a reset in between, the parity build and the oracle (correctly rounded libm) side by side from the constructor
    placement on -- bit-identical state after every step, and the rect where the oracle has it."""
    rng = np.random.RandomState(11)
    ol.lib().rro_debug_attribution(4)
    try:
        for arena in (3, 77):
            e, o = el.EmuEnv(preset, seed=9, exact=True), ol.OracleEnv(preset)
            na = e.nr
            for episode in (0, 1, 2):
                e.reset(arena, episode)
                o.reset(9, arena, episode)
                so, se = o.get_state(), e.get_state()
                assert _same(so, se), (preset, arena, episode, "placement")
                for s in range(40):
                    a = rng.randint(0, 8, na).astype(np.int32)
                    o.step(a)
                    e.step(a)
                    assert _same(o.get_state(), e.get_state()), (preset, arena, episode, s)
                assert np.array_equal(o.get_state()["inner"][:2], e.get_scratch_rect()), (preset, arena, episode)
    finally:
        ol.lib().rro_debug_attribution(0)
