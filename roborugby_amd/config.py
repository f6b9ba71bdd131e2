"""Constant presets of the reference (robo_rugby/gym_env/RR_Constants.py).

The reference has one source-level switch, GAME_MODE (RR_Constants.py:4); both settings are first-class here:
  G  GAME_MODE=True  (as checked in): 800x800, 2+2 robots, 4+4 balls, 4500-step games   (main.py)
  T  GAME_MODE=False: 600x600, 1+0 robots, 1+0 balls, 300-step games  (required by Training_DQN_pytorch.py:233-234)
A third compiled shape, D (the two-team "duel" the SimpleDuel* classes are named after): GAME_MODE=True with the four entity
counts of RR_Constants.py:30-34 set to 1 -- 800x800, 1+1 robots, 1+1 balls, 4500-step games.  The reference reaches it by editing
those four integers; its golden vectors are generated with exactly that edit (oracle/refgen/load_reference.py).
"""
import math
from dataclasses import dataclass


@dataclass(frozen=True)
class Preset:
    name: str
    game_mode: bool
    arena_w: float
    arena_h: float
    nr_happy: int
    nr_grumpy: int
    nb_pos: int
    nb_neg: int
    framerate: int
    game_len_steps: int

    @property
    def nr(self):
        return self.nr_happy + self.nr_grumpy

    @property
    def nb(self):
        return self.nb_pos + self.nb_neg

    @property
    def points_ball_travel_mult(self):  # RR_Constants.py:44-46
        return 200000 / math.pow(self.arena_w ** 2 + self.arena_h ** 2, .5)

    @property
    def points_robot_travel_mult(self):  # RR_Constants.py:50
        return self.points_ball_travel_mult / 100

    def algorithmic_bytes_per_step(self, na=None, real_bytes=4):
        """read+write of the persistent state (robot 7 words, ball 4 words, step 1 word),
        read actions, write obs/reward per reported team and done.  Quoted at 4-byte words."""
        g = 1 if self.nr_grumpy > 0 else 0
        na = self.nr if na is None else na
        w = real_bytes
        return 2 * (7 * w * self.nr + 4 * w * self.nb + 4) + 4 * na + 44 * (1 + g) + 4 * (1 + g) + 1


PRESETS = {
    "G": Preset("G", True, 800.0, 800.0, 2, 2, 4, 4, 30, int(2.5 * 60 * 30)),
    "T": Preset("T", False, 600.0, 600.0, 1, 0, 1, 0, 30, int(10 / 60 * 60 * 30)),
    "D": Preset("D", True, 800.0, 800.0, 1, 1, 1, 1, 30, int(2.5 * 60 * 30)),
    # NOT one of the shapes inside libroborugby_amd.so: the reference's entity counts are free integers (RR_Constants.py:30-34), and any
    # other counts run through a one-shape library compiled on demand (build.build_shape_library).  X -- 2 + 1 robots, 2 + 3 balls, G's
    # other constants -- is the one the tests pin to reference vectors (tests/golden/{traj,reset}_X.npz); custom_preset() makes more.
    "X": Preset("X", True, 800.0, 800.0, 2, 1, 2, 3, 30, int(2.5 * 60 * 30)),
}


def custom_preset(nr_happy, nr_grumpy, nb_pos, nb_neg, game_mode=True, name=None):
    """A preset with the reference's other constants of GAME_MODE=True / False (RR_Constants.py:6-7,24-25) and these entity counts
    (RR_Constants.py:30-34): pass it as `preset=`; the env loads (or compiles, hipcc permitting) the one-shape library for it."""
    base = PRESETS["G" if game_mode else "T"]
    return Preset(name or f"{nr_happy}+{nr_grumpy}/{nb_pos}+{nb_neg}", bool(game_mode), base.arena_w, base.arena_h, int(nr_happy), int(nr_grumpy),
                  int(nb_pos), int(nb_neg), base.framerate, base.game_len_steps)


ROBOT_LENGTH, ROBOT_WIDTH = 20, 40  # RR_Constants.py:8-9
BALL_RADIUS = 7
MOVES_PER_FRAME = 12
TEAM_HAPPY, TEAM_GRUMPY = 1, -1
GOAL_WIDTH = GOAL_HEIGHT = 240

ENV_IDS = {  # robo_rugby/__init__.py:28-34 -- the one registered id whose class constructs at this commit
    "RoboRugbySimpleDuel-v3": dict(entry="SimpleDuel3"),
}
