"""`robo_rugby.gym_env.RR_EnvBase` names (GameEnv, GameEnv_Simple) over the MI355X simulator.

One arena, numpy in/out, the reference's exceptions -- the shape Training_DQN_pytorch.py:317-360 drives -- executed by the
HIP library through roborugby_amd.RoboRugbyEnv (one arena of a batch of one; for throughput use
roborugby_amd.BatchedRoboRugbyEnv directly).  The env's own `done` is the raw rule step_count > T
(RR_EnvBase.py:555-559); gym.make adds the TimeLimit wrapper exactly as it does for the reference."""
from enum import Enum

from roborugby_amd import env as _env

from . import RR_Constants as const

try:  # a gym.Env subclass when gym is there, so gym.make's wrappers and isinstance checks see what they expect
    import gym as _gym
    _Base = _gym.Env
except Exception:  # gym is optional: robo_rugby.make() works without it
    _Base = object


def _preset():
    return "G" if const.GAME_MODE else "T"


class GameEnv(_Base):
    """Bare env with the continuous entry: step([(L, R), ...]) (RR_EnvBase.py:260-273), Box(-1, 1, (2 * happy robots,))."""
    metadata = {"render.modes": ["human", "rgb_array"], "video.frames_per_second": const.FRAMERATE}
    reward_range = (-float("inf"), float("inf"))
    DebugInfo = _env.DebugInfo
    CONFIG_RANDOM = None
    _w5, _h5 = const.ARENA_WIDTH / 5, const.ARENA_HEIGHT / 5
    CONFIG_STANDARD = [  # RR_EnvBase.py:35-52: two robots per team on the anti-diagonal, eight balls on a cross
        [(const.ARENA_WIDTH / 2 + 1 * _w5, const.ARENA_HEIGHT - 1 * _h5, 135), (const.ARENA_WIDTH / 2 + 2 * _w5, const.ARENA_HEIGHT - 2 * _h5, 135),
         (const.ARENA_WIDTH / 2 - 1 * _w5, 1 * _h5, 315), (const.ARENA_WIDTH / 2 - 2 * _w5, 2 * _h5, 315)],
        [(_w5 * 1, const.ARENA_HEIGHT - _h5 * 1), (_w5 * 2, const.ARENA_HEIGHT - _h5 * 2), (_w5 * 3, const.ARENA_HEIGHT - _h5 * 3),
         (_w5 * 4, const.ARENA_HEIGHT - _h5 * 4), (const.ARENA_WIDTH / 2, _h5), (const.ARENA_WIDTH / 2, const.ARENA_HEIGHT - _h5),
         (_w5, const.ARENA_HEIGHT / 2), (const.ARENA_WIDTH - _w5, const.ARENA_HEIGHT / 2)]]
    _ACTION_MODE = "thrust"
    _REWARDS = ()                          # the bare env has no score keepers: reward 0
    _OBSERVER = "SingleBall_6wayLidar_v2"  # (the reference's bare GameEnv observes None; an observer is needed to be usable)

    def __init__(self, lst_starting_config=CONFIG_RANDOM, **kw):
        kw.setdefault("time_limit", False)
        kw.setdefault("rewards", self._REWARDS)
        kw.setdefault("observer", self._OBSERVER)
        self._e = _env.RoboRugbyEnv(preset=_preset(), action_mode=self._ACTION_MODE, lst_starting_config=lst_starting_config, **kw)
        self.observation_space = self._e.observation_space
        self.action_space = self._e.action_space

    @property
    def unwrapped(self):
        return self

    @property
    def lngStepCount(self):
        return int(self._e._b.get_state()["step"][0])

    def reset(self, bln_randomize_pos=True):
        return self._e.reset(bln_randomize_pos)

    def step(self, lstArgs):
        return self._e.step(lstArgs)

    def get_game_state(self, int_team=const.TEAM_HAPPY, obj_robot=None, obj_ball=None):
        return self._e.get_game_state(int_team=int_team, obj_robot=obj_robot, obj_ball=obj_ball)

    def game_is_done(self):
        return self.lngStepCount > const.GAME_LENGTH_STEPS

    def render(self, mode="human"):
        return self._e.render(mode)

    def seed(self, seed=None):
        return self._e.seed(seed)

    def close(self):
        self._e.close()


class GameEnv_Simple(GameEnv):
    """Discrete(8) entry: step([direction, ...]) (RR_EnvBase.py:580-626)."""

    class Direction(Enum):
        FORWARD = 0
        BACKWARD = 1
        LEFT = 2
        RIGHT = 3
        F_L = 4
        F_R = 5
        B_L = 6
        B_R = 7

    _ACTION_MODE = "discrete"
    _LR = {0: (1, 1), 1: (-1, -1), 2: (-1, 1), 3: (1, -1), 4: (0, 1), 5: (1, 0), 6: (-1, 0), 7: (0, -1)}

    @staticmethod
    def thrust_from_direction(direction):
        return GameEnv_Simple._LR[int(direction)]
