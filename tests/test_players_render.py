"""SURVEY section 8(f)-4 pieces: the scripted OG_Twitchy opponent as a batched policy, the CPU debug render, the
continuous-thrust (SAC-facing) action surface."""
import numpy as np
import pytest
import torch

from roborugby_amd.players import og_twitchy
from roborugby_amd.render import draw_arena, robot_corners
from roborugby_amd.config import PRESETS


def test_og_twitchy_distribution():
    g = torch.Generator().manual_seed(0)
    t = og_twitchy(200000, 2, generator=g, device="cpu").view(-1, 2)
    kinds = {(-1.0, 1.0): 0.05, (1.0, 1.0): 0.45, (-1.0, -1.0): 0.45, (1.0, -1.0): 0.05}
    for (l, r), p in kinds.items():
        f = float(((t[:, 0] == l) & (t[:, 1] == r)).float().mean())
        assert abs(f - p) < 0.005, ((l, r), f)


def test_debug_render_shapes_and_marks():
    p = PRESETS["G"]
    robots = np.zeros((4, 10))
    robots[:, 0] = [100, 300, 500, 700]
    robots[:, 1] = 400
    robots[:, 6] = [0, 90, 45, 270]
    balls = np.zeros((8, 8))
    balls[:, 0] = np.arange(8) * 80 + 60
    balls[:, 1] = 200
    img = draw_arena(p, robots, balls)
    assert img.shape == (800, 1100, 3) and img.dtype == np.uint8
    assert tuple(img[200, 60]) == (80, 220, 100) and tuple(img[200, 60 + 80 * 5]) == (60, 16, 83)
    assert tuple(img[790, 790]) == (43, 146, 228) and tuple(img[5, 5]) == (242, 53, 87)
    c = robot_corners(0, 0, 90)  # rot 90: the 20x40 rect lies on its side (40 wide)
    assert abs(max(x for x, _ in c) - 20) < 1e-9 and abs(max(y for _, y in c) - 10) < 1e-9


@pytest.mark.gpu
def test_thrust_action_mode_and_render_on_gpu():
    import roborugby_amd as rr
    env = rr.make("RoboRugbySimpleDuel-v3", num_envs=256, preset="G", action_mode="thrust")
    assert env.action_space.shape == (4,) and float(env.action_space.high[0]) == 1.0 and env.reward_range[0] == -float("inf")
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    happy = torch.rand(256, 4, generator=g, device="cuda") * 2 - 1          # the agent's two robots (SAC-style Box action)
    grumpy = og_twitchy(256, 2, generator=g)                                 # scripted opponents
    obs, rew, done, info = env.step(torch.cat([happy, grumpy], dim=1))
    st = env.get_state()
    thr = st["robots_i"][:, :, 1:].float()
    assert torch.equal(thr[:, 2:].reshape(256, 4), grumpy) and int(thr[:, :2].abs().max()) <= 1
    img = env.render("rgb_array", arena=3)
    assert img.shape == (800, 1100, 3) and env.render() is None
    single = rr.make("RoboRugbySimpleDuel-v3", preset="T", action_mode="thrust")
    single.reset()
    o, r, d, i = single.step([(1.0, 0.4)])
    assert o.shape == (11,) and single.render("rgb_array").shape == (600, 900, 3)
    with pytest.raises(Exception, match="robot engines"):
        single.step([(1, 1), (1, 1)])
