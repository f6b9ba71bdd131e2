"""`robo_rugby.gym_env`: the names callers of the reference import (Training_DQN_pytorch.py:8-9)."""
from . import RR_Constants, RR_EnvBase, RR_Environments  # noqa: F401
from .RR_EnvBase import GameEnv, GameEnv_Simple  # noqa: F401
