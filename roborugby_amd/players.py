"""Scripted opponents as on-device batched policies.

`og_twitchy` is the reference's OG_Twitchy (robo_rugby/gym_env/RR_Players.py:14-30): per robot, 5 % turn left (-1, 1),
45 % straight (1, 1), 45 % back (-1, -1), 5 % turn right (1, -1) -- as (L, R) thrust pairs for `step_thrust`."""
import torch

_TABLE = ((-1.0, 1.0), (1.0, 1.0), (-1.0, -1.0), (1.0, -1.0))


def og_twitchy(num_envs, num_robots, generator=None, device="cuda"):
    """float32 [num_envs, 2*num_robots] thrust pairs drawn like OG_Twitchy.get_action() for every robot."""
    u = torch.rand(num_envs, num_robots, generator=generator, device=device)
    idx = (u > 0.05).long() + (u > 0.5).long() + (u >= 0.95).long()  # <=.05 left, <=.5 straight, <.95 back, else right
    table = torch.tensor(_TABLE, dtype=torch.float32, device=device)
    return table[idx].reshape(num_envs, 2 * num_robots)
