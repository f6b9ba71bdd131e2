"""The fixed-point shortcut of the sub-step loop (rr_sim.hpp, step_arena): when an expensive sub-step leaves the whole
arena bit-identical to the sub-step before it, the remaining sub-steps are skipped.  It must be EXACT: the kernel's phase
source (host-emulated wave) with the shortcut on and off has to produce bit-identical states, observations, rewards and
status words over multi-step rollouts of stuck / contact-dense arenas -- and the shortcut has to fire on them."""
import io
import os
import sys

import numpy as np
import pytest

import adversarial as adv
import emu_lib as el

HERE = os.path.dirname(os.path.abspath(__file__))


def _rollout(preset, robots, balls, actions, memo, steps, narrow=False, state=None):
    el.lib().emu_debug_memo(int(memo))
    try:
        env = el.EmuEnv(preset, narrow=narrow)
        if state is not None:
            env.set_state(*state)
        else:
            env.set_poses(robots, balls)
        out = []
        for k in range(steps):
            r = env.step(actions)  # same action every step: a robot keeps pushing
            st = env.get_state()
            out.append((r["obs"].tobytes(), r["obs_g"].tobytes(), r["reward"], r["reward_g"], r["done"], r["status"], r["naughty"],
                        st["robots"].tobytes(), st["robots_i"].tobytes(), st["balls"].tobytes(), st["step"]))
        return out
    finally:
        el.lib().emu_debug_memo(1)


def _count_fixed_points(fn):
    """runs fn() with the emulation trace on and counts 'fixed point' lines on stderr (fd-level capture)."""
    sys.stderr.flush()
    saved = os.dup(2)
    r, w = os.pipe()
    os.dup2(w, 2)
    os.close(w)
    el.lib().emu_debug_trace(1)
    try:
        fn()
    finally:
        el.lib().emu_debug_trace(0)
        sys.stderr.flush()
        os.dup2(saved, 2)
        os.close(saved)
    chunks = []
    while True:
        b = os.read(r, 1 << 16)
        if not b:
            break
        chunks.append(b)
    os.close(r)
    return b"".join(chunks).count(b"fixed point")


@pytest.mark.parametrize("preset,n,narrow", [("T", 240, False), ("T", 120, True), ("G", 60, False), ("G", 60, True)])
def test_shortcut_is_bit_exact_on_contact_dense_rollouts(preset, n, narrow):
    robots, balls, actions = adv.make_states(preset, n, seed=5 + int(narrow))
    for a in range(n):
        on = _rollout(preset, robots[a], balls[a], actions[a], True, 4, narrow)
        off = _rollout(preset, robots[a], balls[a], actions[a], False, 4, narrow)
        assert on == off, (preset, a)


def test_shortcut_fires_on_a_stuck_arena_T():
    # robot driving a ball into the left wall: every sub-step pushes, exhausts the resolve loop and is undone
    W = 600.0
    robots = np.array([[30.0, 300.0, 180.0]])   # facing -x, hugging the wall region
    balls = np.array([[9.0, 300.0, 0.0, 0.0]])
    act = np.array([0], np.int32)                # forward
    on = _rollout("T", robots, balls, act, True, 6)
    off = _rollout("T", robots, balls, act, False, 6)
    assert on == off
    fired = _count_fixed_points(lambda: _rollout("T", robots, balls, act, True, 6))
    assert fired >= 1, "the stuck arena never reached the fixed-point shortcut"


def test_shortcut_exact_on_the_squeezed_fixture_G():
    d = np.load(os.path.join(HERE, "..", "tools", "fixtures", "squeezed_G.npz"))
    state = (d["robots"], d["robots_i"], d["balls"], int(d["step"]))
    on = _rollout("G", None, None, d["actions"], True, 5, state=state)
    off = _rollout("G", None, None, d["actions"], False, 5, state=state)
    assert on == off
