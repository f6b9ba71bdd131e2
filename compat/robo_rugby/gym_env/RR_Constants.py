"""`robo_rugby.gym_env.RR_Constants` for callers of the reference (Training_DQN_pytorch.py:8,233-234,258,322,339,361), backed by
roborugby_amd's presets.  Builder-authored: the names are the reference's (its callers import them), the values come from
roborugby_amd.config, which tests/test_abi_and_host.py pins against the constants captured from the reference.

The reference switches presets by editing `GAME_MODE = True` in its source (RR_Constants.py:4).  Here the same switch is
the environment variable ROBO_RUGBY_GAME_MODE (default "1" = the value the reference has checked in); the DQN script
needs 0 (it refuses to run otherwise, Training_DQN_pytorch.py:233-234)."""
import os

from roborugby_amd import config as _cfg

GAME_MODE = os.environ.get("ROBO_RUGBY_GAME_MODE", "1").strip().lower() not in ("0", "false", "no", "")
_p = _cfg.PRESETS["G" if GAME_MODE else "T"]

ARENA_WIDTH, ARENA_HEIGHT = int(_p.arena_w), int(_p.arena_h)
ROBOT_LENGTH, ROBOT_WIDTH = _cfg.ROBOT_LENGTH, _cfg.ROBOT_WIDTH
ROBOT_WIDTH_BODY, ROBOT_WIDTH_TRACKS = 32, 8
ROBOT_VEL = MOVES_PER_FRAME = _cfg.MOVES_PER_FRAME
ROBOT_ANGULAR_VEL_ONE, ROBOT_ANGULAR_VEL_BOTH = .6, 1.2
GOAL_WIDTH, GOAL_HEIGHT = _cfg.GOAL_WIDTH, _cfg.GOAL_HEIGHT
BALL_RADIUS = _cfg.BALL_RADIUS
BALL_SLOWDOWN = PUSH_FACTOR = .995
BALL_MIN_SPEED = 0.005
FRAMERATE = _p.framerate
GAME_LENGTH_STEPS = _p.game_len_steps
GAME_LENGTH_MINS = GAME_LENGTH_STEPS / 60 / FRAMERATE
TIME_BALL_IN_GOAL_SECONDS = 5
TIME_BALL_IN_GOAL_STEPS = TIME_BALL_IN_GOAL_SECONDS * FRAMERATE
NUM_BALL_POS, NUM_BALL_NEG = _p.nb_pos, _p.nb_neg
NUM_ROBOTS_HAPPY, NUM_ROBOTS_GRUMPY = _p.nr_happy, _p.nr_grumpy
NUM_ROBOTS_TOTAL = _p.nr
MAX_NEG_BALLS = 3
POINTS_BALL_SCORED = 500
POINTS_TIME_PENALTY = .1
POINTS_ROBOT_CRASH_PENALTY = POINTS_ROBOT_IN_GOAL_PENALTY = POINTS_NO_MOVE_PENALTY = .005
POINTS_BALL_TRAVEL_MAX = 200000
POINTS_BALL_TRAVEL_MULT = _p.points_ball_travel_mult
POINTS_GOAL_DESTROYED = (POINTS_BALL_SCORED + POINTS_BALL_TRAVEL_MAX) * (NUM_BALL_POS + NUM_BALL_NEG)
POINTS_ROBOT_TRAVEL_MULT = _p.points_robot_travel_mult
TEAM_HAPPY, TEAM_GRUMPY = _cfg.TEAM_HAPPY, _cfg.TEAM_GRUMPY

# key bindings: pygame's K_<letter> constants are the ASCII codes of the letters
KEY_LEFT_MOTOR_FORWARD, KEY_LEFT_MOTOR_BACKWARD = ord("i"), ord("k")
KEY_RIGHT_MOTOR_FORWARD, KEY_RIGHT_MOTOR_BACKWARD = ord("o"), ord("l")
KEY_BOTH_MOTOR_FORWARD, KEY_BOTH_MOTOR_BACKWARD = ord("w"), ord("s")
KEY_BOTH_MOTOR_LEFT, KEY_BOTH_MOTOR_RIGHT = ord("a"), ord("d")

CALC_DIST_TRACK_CENTER_TO_ROBOT_CENTER = (ROBOT_WIDTH / 2) - (ROBOT_WIDTH_TRACKS / 2)
BOUNCE_K_WALL, BOUNCE_K_ROBOT, BOUNCE_K_BALL = .8, .8, .995
MASS_BALL, MASS_ROBOT, MASS_WALL = 1, 2, 3
DASHBOARD_WIDTH = 300
