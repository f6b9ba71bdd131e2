"""`robo_rugby` for callers of the reference, executed by the MI355X simulator (roborugby_amd).

    PYTHONPATH=<repo>:<repo>/compat ROBO_RUGBY_GAME_MODE=0 python Training_DQN_pytorch.py

`import robo_rugby` registers the env id with gym the way the reference's package does (robo_rugby/__init__.py:28-34:
max_episode_steps=GAME_LENGTH_STEPS, nondeterministic=True, reward_threshold=1.0), so the script's own
`gym.make('RoboRugbySimpleDuel-v3')` resolves to the HIP-backed SimpleDuel3.  Without gym installed the same id is served by
`robo_rugby.make(id)` (TimeLimit rule included); `<repo>/compat/gym_minimal` holds a tiny `gym` stand-in for boxes that
have none.  Builder-authored shim: nothing here is copied from the reference."""
import robo_rugby.gym_env.RR_Constants as const

REGISTRY = {
    "RoboRugbySimpleDuel-v3": dict(entry_point="robo_rugby.gym_env.RR_Environments:SimpleDuel3",
                                   max_episode_steps=const.GAME_LENGTH_STEPS, nondeterministic=True, reward_threshold=1.0),
}


def _register_with_gym():
    try:
        from gym.envs.registration import register
    except Exception:
        return False
    for env_id, kw in REGISTRY.items():
        try:
            register(id=env_id, **kw)
        except Exception:  # gym raises when an id is registered twice (re-import under another name)
            pass
    return True


HAVE_GYM = _register_with_gym()


class _TimeLimit:
    """What gym's TimeLimit wrapper does for an id registered with max_episode_steps (gym <= 0.21 semantics):
    done at elapsed == max_episode_steps, info['TimeLimit.truncated'] = not already done."""

    def __init__(self, env, max_episode_steps, env_id):
        import types
        self.env, self._max, self._elapsed = env, max_episode_steps, None
        self.spec = types.SimpleNamespace(id=env_id, max_episode_steps=max_episode_steps, nondeterministic=True, reward_threshold=1.0)

    def __getattr__(self, name):
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def reset(self, **kw):
        self._elapsed = 0
        return self.env.reset(**kw)

    def step(self, action):
        assert self._elapsed is not None, "Cannot call env.step() before calling reset()"
        obs, reward, done, info = self.env.step(action)
        self._elapsed += 1
        if self._elapsed >= self._max:
            info["TimeLimit.truncated"] = not done
            done = True
        return obs, reward, done, info


def make(env_id, **kw):
    """gym.make for the ids above, usable without gym."""
    import importlib
    spec = REGISTRY[env_id]
    mod, cls = spec["entry_point"].split(":")
    env = getattr(importlib.import_module(mod), cls)(**kw)
    return _TimeLimit(env, spec["max_episode_steps"], env_id)
