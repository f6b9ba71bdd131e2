"""ctypes binding of the CPU oracle (oracle/rr_oracle.c) -- test infrastructure only.

Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
SO = os.path.join(ORACLE_DIR, "librr_oracle.so")

PRESETS = {  # RR_Constants.py:6-7,24-25,30-34
    "G": dict(nr_h=2, nr_g=2, nb_p=4, nb_n=4, W=800.0, H=800.0, game_len=4500, game_mode=1),
    "T": dict(nr_h=1, nr_g=0, nb_p=1, nb_n=0, W=600.0, H=600.0, game_len=300, game_mode=0),
    "D": dict(nr_h=1, nr_g=1, nb_p=1, nb_n=1, W=800.0, H=800.0, game_len=4500, game_mode=1),  # the duel: G's constants, one entity each
    # a shape outside the library's built list (compiled on demand: roborugby_amd.build.build_shape_library): odd counts, unequal teams
    "X": dict(nr_h=2, nr_g=1, nb_p=2, nb_n=3, W=800.0, H=800.0, game_len=4500, game_mode=1),
    "Y": dict(nr_h=1, nr_g=1, nb_p=2, nb_n=1, W=800.0, H=800.0, game_len=4500, game_mode=1),  # four lanes per arena, three balls (emulation vs oracle only)
}


def build(force=False):
    src = os.path.join(ORACLE_DIR, "rr_oracle.c")
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", "librr_oracle.so"], stdout=subprocess.DEVNULL)
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        L = _lib
        dp, ip, u8p = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
        L.rro_create.restype = C.c_void_p
        L.rro_create.argtypes = [C.c_int] * 4 + [C.c_double] * 2 + [C.c_int] * 2
        L.rro_destroy.argtypes = [C.c_void_p]
        L.rro_set_state.argtypes = [C.c_void_p, dp, ip, dp, dp, C.c_int32]
        L.rro_get_state.argtypes = [C.c_void_p, dp, ip, dp, dp, ip]
        L.rro_set_clean_state.argtypes = [C.c_void_p, dp, dp, C.c_int32]
        L.rro_observe.argtypes = [C.c_void_p, C.c_int, dp]
        L.rro_observe_for.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, dp]
        L.rro_observe_kind.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, dp]
        L.rro_set_program.argtypes = [C.c_void_p, ip, C.c_int]
        L.rro_step.argtypes = [C.c_void_p, ip, C.c_int, dp, dp, dp, dp, u8p, ip]
        L.rro_step_thrust.argtypes = [C.c_void_p, dp, C.c_int, dp, dp, dp, dp, u8p, ip]
        L.rro_reset.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
        L.rro_status.argtypes = [C.c_void_p]
        L.rro_set_goal_scoring.argtypes = [C.c_void_p, C.c_int]
        L.rro_goal_scores.argtypes = [C.c_void_p, ip]
        L.rro_kat_line_intersection.argtypes = [dp, dp, u8p]
        L.rro_kat_dist_angle.argtypes = [dp, dp]
        L.rro_kat_floatrect.argtypes = [dp, dp, dp]
        L.rro_kat_lidar.argtypes = [C.c_double, C.c_double, dp, dp]
        L.rro_kat_ball_robot.argtypes = [dp]
        L.rro_kat_robots.argtypes = [dp]
        L.rro_rollout.restype = C.c_long
        L.rro_rollout.argtypes = [C.c_int] * 4 + [C.c_double] * 2 + [C.c_int] * 4 + [C.c_uint64, dp]
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class OracleEnv:
    """One arena of the CPU oracle."""

    def __init__(self, preset="T", **over):
        cfg = dict(PRESETS[preset]) if isinstance(preset, str) else dict(preset)
        cfg.update(over)
        self.cfg = cfg
        self.nr = cfg["nr_h"] + cfg["nr_g"]
        self.nb = cfg["nb_p"] + cfg["nb_n"]
        self.h = lib().rro_create(cfg["nr_h"], cfg["nr_g"], cfg["nb_p"], cfg["nb_n"], cfg["W"], cfg["H"],
                                  cfg["game_len"], cfg["game_mode"])
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            lib().rro_destroy(self.h)
            self.h = None

    def set_state(self, robots, robots_i, balls, inner=None, step=0):
        robots = np.ascontiguousarray(robots, np.float64)
        robots_i = np.ascontiguousarray(robots_i, np.int32)
        balls = np.ascontiguousarray(balls, np.float64)
        assert robots.shape == (self.nr, 10) and robots_i.shape == (self.nr, 3) and balls.shape == (self.nb, 8)
        ip = None
        if inner is not None:
            inner = np.ascontiguousarray(inner, np.float64)
            ip = _dp(inner)
        lib().rro_set_state(self.h, _dp(robots), _ip(robots_i), _dp(balls), ip, int(step))

    def get_state(self):
        robots = np.zeros((self.nr, 10))
        robots_i = np.zeros((self.nr, 3), np.int32)
        balls = np.zeros((self.nb, 8))
        inner = np.zeros(3)
        step = np.zeros(1, np.int32)
        lib().rro_get_state(self.h, _dp(robots), _ip(robots_i), _dp(balls), _dp(inner), _ip(step))
        return dict(robots=robots, robots_i=robots_i, balls=balls, inner=inner, step=int(step[0]))

    def set_clean_state(self, robots_xyr, balls_xyv, step=0):
        r = np.ascontiguousarray(robots_xyr, np.float64)
        b = np.ascontiguousarray(balls_xyv, np.float64)
        assert r.shape == (self.nr, 3) and b.shape == (self.nb, 4)
        lib().rro_set_clean_state(self.h, _dp(r), _dp(b), int(step))

    def observe(self, team=1, robot=-1, ball=-1):
        o = np.zeros(11)
        ok = lib().rro_observe_for(self.h, team, robot, ball, _dp(o))
        return o if ok else None

    def set_program(self, ids):
        """Reward keepers in on_step_end execution order (1 Naughty, 2 Chase, 3 PushPos, 4 DontDrive, 5 KeepMoving,
        6 BaseDestruction, 7 PushNeg); default = SimpleDuel3's (1, 2, 3)."""
        a = np.ascontiguousarray(ids, np.int32)
        lib().rro_set_program(self.h, _ip(a), len(a))

    def observe_kind(self, kind, team=1, robot=-1, ball=-1):
        """kind 0 = SingleBall_6wayLidar_v2, 1 = SingleBall_6wayLidar, 2 = PosBall_BasicLidar, 3 = AllCoords,
        4 = AllCoords_WithPrior."""
        o = np.zeros(64)
        n = lib().rro_observe_kind(self.h, kind, team, robot, ball, _dp(o))
        return o[:n].copy() if n > 0 else None

    def step(self, actions):
        a = np.ascontiguousarray(np.asarray(actions).reshape(-1), np.int32)
        obs, obs_g = np.zeros(11), np.zeros(11)
        rew, rew_g = np.zeros(1), np.zeros(1)
        done = np.zeros(1, np.uint8)
        naughty = np.zeros(1, np.int32)
        st = lib().rro_step(self.h, _ip(a), len(a), _dp(obs), _dp(obs_g), _dp(rew), _dp(rew_g),
                            done.ctypes.data_as(C.POINTER(C.c_uint8)), _ip(naughty))
        return dict(obs=obs, obs_g=obs_g, reward=float(rew[0]), reward_g=float(rew_g[0]), done=bool(done[0]),
                    naughty=int(naughty[0]), status=int(st))

    def step_thrust(self, thrust):
        t = np.ascontiguousarray(np.asarray(thrust, np.float64).reshape(-1))
        obs, obs_g = np.zeros(11), np.zeros(11)
        rew, rew_g = np.zeros(1), np.zeros(1)
        done = np.zeros(1, np.uint8)
        naughty = np.zeros(1, np.int32)
        st = lib().rro_step_thrust(self.h, _dp(t), len(t) // 2, _dp(obs), _dp(obs_g), _dp(rew), _dp(rew_g),
                                   done.ctypes.data_as(C.POINTER(C.c_uint8)), _ip(naughty))
        return dict(obs=obs, obs_g=obs_g, reward=float(rew[0]), reward_g=float(rew_g[0]), done=bool(done[0]),
                    naughty=int(naughty[0]), status=int(st))

    def reset(self, seed, arena, episode):
        lib().rro_reset(self.h, int(seed), int(arena), int(episode))
        return lib().rro_status(self.h)

    def set_goal_scoring(self, on=True):
        """Opt-in goal scoring: an extension with no reference behaviour behind it (oracle/rr_oracle.c: goal_step)."""
        lib().rro_set_goal_scoring(self.h, int(on))

    def goal_scores(self):
        s = np.zeros(2, np.int32)
        lib().rro_goal_scores(self.h, _ip(s))
        return s


def rollout(preset, n_env, n_steps, seed=0):
    cfg = PRESETS[preset]
    cs = np.zeros(1)
    n = lib().rro_rollout(cfg["nr_h"], cfg["nr_g"], cfg["nb_p"], cfg["nb_n"], cfg["W"], cfg["H"], cfg["game_len"],
                          cfg["game_mode"], n_env, n_steps, seed, _dp(cs))
    return n, float(cs[0])
