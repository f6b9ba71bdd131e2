"""A tiny `gym` stand-in for boxes without gym (there is no network to install it): exactly the surface
Training_DQN_pytorch.py and the robo_rugby shim touch -- gym.Env, gym.make, gym.spaces.{Box,Discrete},
gym.envs.registration.register -- with gym <= 0.21's TimeLimit rule.  Put <repo>/compat/gym_minimal on PYTHONPATH only when
the real gym is absent; it is never imported by roborugby_amd itself."""
import importlib
import sys
import types

from roborugby_amd.spaces import Box, Discrete


class Env:
    metadata = {"render.modes": []}
    reward_range = (-float("inf"), float("inf"))
    spec = None
    action_space = None
    observation_space = None

    @property
    def unwrapped(self):
        return self


_registry = {}


def register(id, entry_point=None, max_episode_steps=None, **kw):
    if id in _registry:
        raise ValueError(f"Cannot re-register id: {id}")
    _registry[id] = dict(entry_point=entry_point, max_episode_steps=max_episode_steps, **kw)


def make(id, **kw):
    if id not in _registry:
        raise KeyError(f"No registered env with id: {id}")
    spec = _registry[id]
    mod, cls = spec["entry_point"].split(":")
    env = getattr(importlib.import_module(mod), cls)(**kw)
    from robo_rugby import _TimeLimit  # the one TimeLimit implementation of this repo
    return _TimeLimit(env, spec["max_episode_steps"], id) if spec["max_episode_steps"] else env


spaces = types.ModuleType("gym.spaces")
spaces.Box, spaces.Discrete = Box, Discrete
envs = types.ModuleType("gym.envs")
registration = types.ModuleType("gym.envs.registration")
registration.register = register
envs.registration = registration
for _m in (spaces, envs, registration):
    sys.modules[_m.__name__] = _m
