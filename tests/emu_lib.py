"""ctypes binding of the host-emulated wave (tests/emu/rr_emu.cpp) -- test harness only.

The emulation compiles the kernel's own source (roborugby_amd/csrc/rr_sim.hpp) with g++, running every
lane-parallel phase as a 64-iteration loop.  It checks phase logic on CPU; it is never a product path."""
import ctypes as C
import os
import subprocess

import numpy as np

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "emu", "librr_emu.so")
SO_EXACT = os.path.join(HERE, "emu", "librr_emu_exact.so")  # the same source with -DRR_EXACT_TRIG=1 (the exact-trig parity build)
SRC = [os.path.join(HERE, "emu", "rr_emu.cpp"), os.path.join(ol.REPO, "roborugby_amd", "csrc", "rr_sim.hpp")]


def build(exact=False):
    so = SO_EXACT if exact else SO
    if not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(s) for s in SRC):
        subprocess.check_call(["g++", "-O2", "-fPIC", "-ffp-contract=off", "-std=c++17", "-shared"] + (["-DRR_EXACT_TRIG=1"] if exact else [])
                              + ["-o", so, SRC[0]])
    return so


_lib = None
_libs = {}


# RR_EMU_EXACT=1: every test that does not say otherwise runs against the parity build (tests/test_parity_build.py re-runs the
# shortcut-equivalence suites that way)
DEFAULT_EXACT = bool(int(os.environ.get("RR_EMU_EXACT", "0")))


def lib(exact=None):
    exact = DEFAULT_EXACT if exact is None else exact
    global _lib
    if exact:
        if True not in _libs:
            _libs[True] = _bind(C.CDLL(build(True)))
        return _libs[True]
    if _lib is None:
        _lib = _bind(C.CDLL(build()))
    return _lib


def _bind(L):
    dp, ip, u8p, fp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.POINTER(C.c_float)
    L.emu_create.restype = C.c_void_p
    L.emu_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64]
    L.emu_destroy.argtypes = [C.c_void_p]
    L.emu_set_state.argtypes = [C.c_void_p, dp, ip, dp, C.c_int]
    L.emu_get_state.argtypes = [C.c_void_p, dp, ip, dp, ip]
    L.emu_set_poses.argtypes = [C.c_void_p, dp, dp]
    L.emu_set_scratch_rect.argtypes = [C.c_void_p, dp]
    L.emu_get_scratch_rect.argtypes = [C.c_void_p, dp]
    L.emu_step.argtypes = [C.c_void_p, ip, C.c_int, dp, dp, dp, dp, u8p]
    L.emu_step_budget.argtypes = [C.c_void_p, ip, C.c_int, dp, dp, dp, dp, u8p, C.c_int]
    L.emu_park_seed.argtypes = [C.c_void_p, C.c_uint32]
    L.emu_step_thrust.argtypes = [C.c_void_p, fp, C.c_int, dp, dp, dp, dp, u8p]
    L.emu_observe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, dp]
    L.emu_reset.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    L.emu_set_program.argtypes = [C.c_void_p, ip, C.c_int]
    L.emu_observe_kind.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, dp]
    L.emu_set_goal_scoring.argtypes = [C.c_void_p, C.c_int]
    L.emu_goal_scores.argtypes = [C.c_void_p, ip]
    L.emu_py_mod360.restype = C.c_double
    L.emu_py_mod360.argtypes = [C.c_double]
    L.emu_sincos.argtypes = [C.c_double, dp, dp]
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class EmuEnv:
    def __init__(self, preset="T", f32=False, time_limit=0, auto_reset=0, seed=0, narrow=False, reset_on_fault=0, exact=None):
        self._exact = DEFAULT_EXACT if exact is None else bool(exact)
        cfg = ol.PRESETS[preset]
        self.cfg = cfg
        self.nr = cfg["nr_h"] + cfg["nr_g"]
        self.nb = cfg["nb_p"] + cfg["nb_n"]
        # narrow=True: fp64 with VW = 4 (T, D) / 16 (G) lanes per arena, i.e. multi-round phases as in the packed builds
        self.h = lib(self._exact).emu_create({"T": 0, "G": 1, "D": 2, "X": 3, "Y": 4}[preset], 2 if narrow else int(f32), cfg["W"], cfg["H"], cfg["game_len"],
                                  cfg["game_mode"], time_limit, int(auto_reset) | (int(reset_on_fault) << 1), seed)

    def __del__(self):
        if getattr(self, "h", None):
            lib(getattr(self, "_exact", False)).emu_destroy(self.h)
            self.h = None

    def set_state(self, robots, robots_i, balls, step=0):
        r = np.ascontiguousarray(robots, np.float64)
        ri = np.ascontiguousarray(robots_i, np.int32)
        b = np.ascontiguousarray(balls, np.float64)
        lib(self._exact).emu_set_state(self.h, _dp(r), _ip(ri), _dp(b), int(step))

    def set_scratch_rect(self, xy):
        """centre of the reference's scratch rect (golden `state_inner[..., :2]`); False if the build does not carry it"""
        v = np.ascontiguousarray(np.asarray(xy, np.float64)[:2])
        return bool(lib(self._exact).emu_set_scratch_rect(self.h, _dp(v)))

    def get_scratch_rect(self):
        v = np.zeros(2)
        return v if lib(self._exact).emu_get_scratch_rect(self.h, _dp(v)) else None

    def get_state(self):
        r = np.zeros((self.nr, 10))
        ri = np.zeros((self.nr, 3), np.int32)
        b = np.zeros((self.nb, 8))
        st = np.zeros(1, np.int32)
        lib(self._exact).emu_get_state(self.h, _dp(r), _ip(ri), _dp(b), _ip(st))
        return dict(robots=r, robots_i=ri, balls=b, step=int(st[0]))

    def set_poses(self, rxyr, bxyv):
        r = np.ascontiguousarray(rxyr, np.float64)
        b = np.ascontiguousarray(bxyv, np.float64)
        lib(self._exact).emu_set_poses(self.h, _dp(r), _dp(b))

    def step(self, actions):
        a = np.ascontiguousarray(np.asarray(actions).reshape(-1), np.int32)
        obs, obs_g, rew, rew_g = np.zeros(11), np.zeros(11), np.zeros(1), np.zeros(1)
        done = np.zeros(1, np.uint8)
        st = lib(self._exact).emu_step(self.h, _ip(a), len(a), _dp(obs), _dp(obs_g), _dp(rew), _dp(rew_g),
                            done.ctypes.data_as(C.POINTER(C.c_uint8)))
        # status word: flags in bits 0-15, NaughtyBots' robot set in bits 16+
        return dict(obs=obs, obs_g=obs_g, reward=float(rew[0]), reward_g=float(rew_g[0]), done=bool(done[0]),
                    status=int(st) & 0xFFFF, naughty=(int(st) >> 16) & 0xFF)

    def step_budget(self, actions, park_mod=3):
        """The budgeted step of the kernel source (rr_sim.hpp: ParkCtx): parks at pseudo-random sub-step boundaries (1 in park_mod).
        Returns None while the step is parked (status NOT_READY: nothing was written), else the usual dict."""
        a = np.ascontiguousarray(np.asarray(actions).reshape(-1), np.int32)
        obs, obs_g, rew, rew_g = np.zeros(11), np.zeros(11), np.zeros(1), np.zeros(1)
        done = np.zeros(1, np.uint8)
        st = lib(self._exact).emu_step_budget(self.h, _ip(a), len(a), _dp(obs), _dp(obs_g), _dp(rew), _dp(rew_g),
                                   done.ctypes.data_as(C.POINTER(C.c_uint8)), int(park_mod))
        if int(st) & 16384:
            return None
        return dict(obs=obs, obs_g=obs_g, reward=float(rew[0]), reward_g=float(rew_g[0]), done=bool(done[0]),
                    status=int(st) & 0xFFFF, naughty=(int(st) >> 16) & 0xFF)

    def park_seed(self, seed):
        lib(self._exact).emu_park_seed(self.h, int(seed) & 0xFFFFFFFF)

    def set_program(self, ids):
        a = np.ascontiguousarray(ids, np.int32)
        lib(self._exact).emu_set_program(self.h, _ip(a), len(a))

    def observe_kind(self, kind, team=1, robot=-1, ball=-1):
        o = np.zeros(64)
        n = lib(self._exact).emu_observe_kind(self.h, kind, team, robot, ball, _dp(o))
        return o[:n].copy() if n > 0 else None

    def step_thrust(self, thrust):
        t = np.ascontiguousarray(np.asarray(thrust, np.float32).reshape(-1))
        obs, obs_g, rew, rew_g = np.zeros(11), np.zeros(11), np.zeros(1), np.zeros(1)
        done = np.zeros(1, np.uint8)
        st = lib(self._exact).emu_step_thrust(self.h, t.ctypes.data_as(C.POINTER(C.c_float)), len(t) // 2, _dp(obs), _dp(obs_g),
                                   _dp(rew), _dp(rew_g), done.ctypes.data_as(C.POINTER(C.c_uint8)))
        return dict(obs=obs, obs_g=obs_g, reward=float(rew[0]), reward_g=float(rew_g[0]), done=bool(done[0]),
                    status=int(st) & 0xFFFF, naughty=(int(st) >> 16) & 0xFF)

    def set_goal_scoring(self, on=True):
        lib(self._exact).emu_set_goal_scoring(self.h, int(on))

    def goal_scores(self):
        s = np.zeros(2, np.int32)
        lib(self._exact).emu_goal_scores(self.h, _ip(s))
        return s

    def observe(self, team=1, robot=-1, ball=-1):
        o = np.zeros(11)
        return o if lib(self._exact).emu_observe(self.h, team, robot, ball, _dp(o)) else None

    def reset(self, arena, episode):
        return lib(self._exact).emu_reset(self.h, int(arena), int(episode))
