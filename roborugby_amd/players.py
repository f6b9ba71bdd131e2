"""Scripted opponents as on-device batched policies.

`og_twitchy` is the reference's OG_Twitchy (robo_rugby/gym_env/RR_Players.py:14-30): per robot, 5 % turn left (-1, 1),
45 % straight (1, 1), 45 % back (-1, -1), 5 % turn right (1, -1) -- as (L, R) thrust pairs for `step_thrust`."""
import ctypes as C

import torch

_TABLE = ((-1.0, 1.0), (1.0, 1.0), (-1.0, -1.0), (1.0, -1.0))


def og_twitchy(num_envs, num_robots, generator=None, device="cuda"):
    """float32 [num_envs, 2*num_robots] thrust pairs drawn like OG_Twitchy.get_action() for every robot."""
    u = torch.rand(num_envs, num_robots, generator=generator, device=device)
    idx = (u > 0.05).long() + (u > 0.5).long() + (u >= 0.95).long()  # <=.05 left, <=.5 straight, <.95 back, else right
    table = torch.tensor(_TABLE, dtype=torch.float32, device=device)
    return table[idx].reshape(num_envs, 2 * num_robots)


def chase(env, obs, step=0, noise=0.1, seed=0, na=None, step_of=None, out=None):
    """The scripted policy of the contact-rich benchmark stream: robot 0 of every arena turns toward its
    ball (or drives forward within 8 degrees), `noise` of the arenas act at random, the other robots act at random.  ONE kernel
    launch on the current stream (rr_policy_chase); the draws are a function of (seed, global arena id, step index) -- `step`,
    or per arena `step_of` (int32 [N]).  Returns int32 [N, na]."""
    from . import _lib
    na = env.preset.nr if na is None else int(na)
    if out is None:
        out = torch.empty(env.num_envs, na, dtype=torch.int32, device=env.device)
    obs = obs.contiguous()
    assert obs.dtype == torch.float32 and obs.shape == (env.num_envs, 11)
    so = C.c_void_p(step_of.data_ptr()) if step_of is not None else None
    _lib.check(env._lib.rr_policy_chase(env._h, C.c_void_p(obs.data_ptr()), so, int(step) & 0xFFFFFFFF, float(noise), int(seed),
                                        C.c_void_p(out.data_ptr()), na, C.c_void_p(torch.cuda.current_stream(env.device).cuda_stream)),
               "rr_policy_chase", env._lib)
    return out
