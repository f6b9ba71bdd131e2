"""Host-side mirror of the reference's gym.Env surface for the batched MI355X simulator.

Reference boundary (harman097/RoboRugby): `gym.make('RoboRugbySimpleDuel-v3')` -> `SimpleDuel3`
(robo_rugby/gym_env/RR_Environments.py:27-32) whose `reset/step/get_game_state/observation_space/action_space/
spec/metadata/unwrapped/render/seed/close` are what Training_DQN_pytorch.py:239-360 touches.  Here the same
names drive N arenas at once; obs/reward/done are PyTorch-ROCm tensors that never leave the device, and the step
itself is one HIP kernel launch through the C-ABI (include/roborugby_amd.h).  PyTorch is plumbing only (device
memory + streams).  No CPU fallback exists: without the HIP library / a GPU this module raises.
"""
import ctypes as C
import math
import types

import numpy as np
import torch

from . import _lib
from .config import PRESETS, ENV_IDS, Preset
from .spaces import Box, Discrete

STATUS_BITS = {
    1: "UNABLE TO RESOLVE BOT/BOT COLLISIONS",            # RR_EnvBase.py:313
    2: "ROBOTS STUCK FROM PRIOR FRAME.",                  # RR_EnvBase.py:328
    4: "UNABLE TO UNDO MOVE FOR ROBOT",                   # RR_EnvBase.py:325
    8: "UNABLE TO RESOLVE ALL COLLISIONS FOR FRAME",      # RR_EnvBase.py:421
    16: "Really tho?? The balls are in the EXACT same spot????",  # RR_TrashyPhysics.py:250
    32: "Numerator AND Denominator are both 0.",          # MyUtils.py:25
    64: "Game is over. Go home.",                         # RR_EnvBase.py:262
    128: "action outside Direction 0..7",                 # KeyError at RR_EnvBase.py:606
}
STATUS_WARN, STATUS_RESET_GAVE_UP, STATUS_WAS_RESET = 256, 512, 1024
STATUS_GOAL_H_DESTROYED, STATUS_GOAL_G_DESTROYED, STATUS_NO_BALLS = 2048, 4096, 8192  # opt-in goal scoring only
STATUS_NOT_READY = 16384  # budgeted step only (step_budget_clocks > 0): the arena's step is still in progress
STATUS_FLAG_MASK, STATUS_NAUGHTY_SHIFT = 0xFFFF, 16  # info.status: flags in bits 0-15, NaughtyBots' robots in bits 16+

# reward mixins of RR_ScoreKeepers.py (ids of the C-ABI's keeper program) and observer mixins of RR_Observers.py
KEEPERS = {"NaughtyBots": 1, "ChasePosBall": 2, "PushPosBallsToGoal": 3, "DontDriveInGoals": 4, "KeepMovingGuys": 5,
           "BaseDestruction": 6, "PushNegBallsFromGoal": 7}
OBSERVERS = {"SingleBall_6wayLidar_v2": 0, "SingleBall_6wayLidar": 1, "PosBall_BasicLidar": 2, "AllCoords": 3,
             "AllCoords_WithPrior": 4}


def observer_dim(kind, nr, nb):
    return {0: 11, 1: 11, 2: 5, 3: 3 * nr + 2 * nb, 4: 6 * nr + 4 * nb}[kind]
SIMPLE_DUEL3_REWARDS = ("PushPosBallsToGoal", "ChasePosBall", "NaughtyBots")  # RR_Environments.py:27-32


def keeper_exec_order(mro_names):
    """on_step_end execution order of a keeper stack given in class (MRO) order: every keeper calls super() first --
    so the LAST one listed runs first -- except NaughtyBots, whose on_step_end never calls super()
    (RR_ScoreKeepers.py:130-135): keepers listed after it do not run at all."""
    names = list(mro_names)
    for n in names:
        if n not in KEEPERS:
            raise KeyError(f"unknown score keeper {n!r}; available: {sorted(KEEPERS)}")
    if "NaughtyBots" in names:
        names = names[:names.index("NaughtyBots") + 1]
    return [KEEPERS[n] for n in reversed(names)]


class Direction:  # GameEnv_Simple.Direction (RR_EnvBase.py:583-591)
    FORWARD, BACKWARD, LEFT, RIGHT, F_L, F_R, B_L, B_R = range(8)


class DebugInfo(dict):
    """RR_EnvBase.py:562-566: a dict (so wrappers can add keys) with the grumpy team's view as attributes."""

    def __init__(self, adblGrumpyState, dblGrumpyScore, status=None):
        super().__init__()
        self.adblGrumpyState = adblGrumpyState
        self.dblGrumpyScore = dblGrumpyScore
        self.status = status


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


# rr_config.dtype (include/roborugby_amd.h): "f64" = the reference's arithmetic, state and arithmetic in fp64 (parity mode, default);
# "f32_state" = BASELINE config 2's "fp32 state": the arenas' records in HBM are fp32, a step computes in fp64 between loading a record
# and writing it back (single steps within 1e-5 of the fp64 reference, contact steps included: tests/test_gpu_fp32.py); "f32" = state
# and arithmetic in fp32, the fast mode (quiet steps within 1e-5, contact steps statistically).
DTYPES = {"f64": 0, "f32": 1, "f32_state": 2}


class BatchedRoboRugbyEnv:
    """N lockstep arenas of SimpleDuel3 on one MI355X.

    reset() -> obs float32[N,11];  step(actions) -> (obs float32[N,11], reward float32[N], done bool[N], info)
    with info.adblGrumpyState float32[N,11] | None, info.dblGrumpyScore float32[N], info.status int32[N] (fault / reset flags
    in bits 0-15, the robots NaughtyBots flagged this step in bits 16+).

    time_limit=True reports done when step_count == max_episode_steps like gym's TimeLimit wrapper does for the
    DQN script; False is the raw env rule (step_count > T, RR_EnvBase.py:555-559).  With auto_reset=True a step
    on a finished arena re-places it (status bit 1024, reward 0, done False, obs = first obs of the new episode)
    instead of raising "Game is over" -- the policy's action for that arena is ignored on that call.  With
    reset_on_fault (default = auto_reset) a step in which the reference would have raised or hung
    (info.status bits 1|2|4|8|16|32) also reports done=True, so faulted arenas are re-placed instead of lingering.

    step_budget_clocks > 0 selects the BUDGETED step (opt-in extension, include/roborugby_amd.h): a call no longer waits for its
    slowest arena -- an arena whose wavefront is over the budget (shader clocks) at the end of an expensive physics sub-step
    parks there, the call reports STATUS_NOT_READY for it (reward 0, done False, its observation row keeps the previous
    values: step() then hands out persistent buffers) and the next call resumes it, ignoring the action it is given.  Each
    arena's trajectory as a function of the actions it accepted is bit-identical to the synchronous mode.
    """
    metadata = {"render.modes": ["human", "rgb_array"], "video.frames_per_second": 30}
    reward_range = (-float("inf"), float("inf"))

    def __init__(self, num_envs, preset="T", device=None, seed=0, time_limit=True, auto_reset=True, dtype="f64",
                 arena_offset=0, env_id="RoboRugbySimpleDuel-v3", reset_on_fault=None, action_mode="discrete",
                 rewards=SIMPLE_DUEL3_REWARDS, observer="SingleBall_6wayLidar_v2", lst_starting_config=None,
                 goal_scoring=False, step_budget_clocks=0, exact_trig=False):
        self.preset = PRESETS[preset] if isinstance(preset, str) else preset
        assert isinstance(self.preset, Preset)
        if not torch.cuda.is_available():
            raise RuntimeError("roborugby_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.type != "cuda":
            raise ValueError("device must be a ROCm/HIP device ('cuda:N')")
        self.num_envs = int(num_envs)
        if dtype not in DTYPES:
            raise ValueError(f"dtype must be one of {sorted(DTYPES)}")
        if dtype == "f32_state" and step_budget_clocks:
            raise ValueError("dtype 'f32_state' has no budgeted step (a parked arena's record would be rounded in the middle of its step)")
        self.dtype = dtype
        self.time_limit, self.auto_reset = bool(time_limit), bool(auto_reset)
        # where the reference would raise / hang inside step() the episode ends (done=True + status bit) so that
        # auto-reset re-places the arena; default: on exactly when auto_reset is
        self.reset_on_fault = bool(auto_reset if reset_on_fault is None else reset_on_fault)
        p = self.preset
        # exact_trig=True: the parity build (libroborugby_amd_exact.so, same ABI): sin / cos of the robot kinematics ~correctly rounded and
        # the reference's scratch-rect carry (set_scratch_rect), so free-running episodes follow the reference bit for bit -- most of them
        # to their last step; ~80 % of the default library's speed, fp64 only
        self.exact_trig = bool(exact_trig)
        if self.exact_trig and dtype != "f64":
            raise ValueError("exact_trig is a property of the fp64 parity mode")
        counts = (p.nr_happy, p.nr_grumpy, p.nb_pos, p.nb_neg)
        from . import build as _build
        if counts in _build.BUILT_SHAPES:
            self._lib = _lib.load(exact=self.exact_trig)
        else:  # the reference's entity counts are free integers (RR_Constants.py:30-34): a library for this one shape, compiled on demand
            if dtype == "f32":
                raise ValueError("entity counts outside the built shapes: dtype 'f64' or 'f32_state'")
            self._lib = _lib.load_shape(counts, exact=self.exact_trig)
        cfg = _lib.RRConfig(
            struct_size=C.sizeof(_lib.RRConfig), num_envs=self.num_envs, nr_happy=p.nr_happy, nr_grumpy=p.nr_grumpy,
            nb_pos=p.nb_pos, nb_neg=p.nb_neg, arena_w=p.arena_w, arena_h=p.arena_h, game_len_steps=p.game_len_steps,
            game_mode=int(p.game_mode), time_limit=int(self.time_limit), auto_reset=int(self.auto_reset),
            reset_on_fault=int(self.reset_on_fault),
            dtype=DTYPES[dtype], device=self.device.index or 0, seed=int(seed),
            arena_offset=int(arena_offset), step_budget_clocks=int(step_budget_clocks), reserved_=0)
        self.step_budget_clocks = int(step_budget_clocks)
        self._bout = None  # persistent step outputs of the budgeted mode (NOT_READY rows keep their previous observation)
        h = C.c_void_p()
        _lib.check(self._lib.rr_create(C.byref(cfg), C.byref(h)), "rr_create", self._lib)
        self._h = h
        # other mixin stacks of the reference: `rewards` in class order like the reference's env classes
        # (RR_Environments.py), `observer` one of OBSERVERS.  SimpleDuel3's own stack is the default and stays fused in
        # the step kernel; anything else runs in light side kernels around it.
        if observer not in OBSERVERS:
            raise KeyError(f"unknown observer {observer!r}; available: {sorted(OBSERVERS)}")
        self.observer, self.obs_kind = observer, OBSERVERS[observer]
        self.obs_dim = observer_dim(self.obs_kind, p.nr, p.nb)
        if self.obs_kind == 4:  # the prior-step copies have to be snapshotted at every on_step_begin from now on
            _lib.check(self._lib.rr_track_prior_step(self._h, 1, self._stream()), "rr_track_prior_step", self._lib)
        self.rewards = tuple(rewards)
        prog = np.asarray(keeper_exec_order(self.rewards), np.int32)
        _lib.check(self._lib.rr_set_reward_program(self._h, prog.ctypes.data_as(C.c_void_p), len(prog)),
                   "rr_set_reward_program", self._lib)
        # Opt-in goal scoring -- an EXTENSION: the reference's goals never score on its live path
        # (RR_Goal.py:58-91 is only reached from the never-called __old_step and is broken), so this has no reference behaviour
        # to match.  A ball inside a goal triangle for 150 consecutive steps is consumed (out of play), +-500 points, three
        # negative balls destroy a goal, a destroyed goal / an empty field ends the episode (include/roborugby_amd.h).
        self.goal_scoring = bool(goal_scoring)
        if self.goal_scoring:
            _lib.check(self._lib.rr_set_goal_scoring(self._h, 1, self._stream()), "rr_set_goal_scoring", self._lib)
        m = max(p.arena_w, p.arena_h, 360)  # RR_Observers.py:30-37
        self.observation_space = Box(-m, m, (self.obs_dim,), np.float32)
        # GameEnv_Simple: Discrete(8) (RR_EnvBase.py:610); bare GameEnv: Box(-1, 1, (2*happy robots,)) (RR_EnvBase.py:118-123),
        # the surface Training_SAC_pytorch.py:270,423 reads (`action_space.high`, `.shape[0]`)
        if action_mode not in ("discrete", "thrust"):
            raise ValueError("action_mode must be 'discrete' or 'thrust'")
        self.action_mode = action_mode
        self.action_space = Discrete(8) if action_mode == "discrete" else Box(-1.0, 1.0, (2 * p.nr_happy,), np.float32)
        self.spec = types.SimpleNamespace(id=env_id, max_episode_steps=p.game_len_steps, nondeterministic=True,
                                          reward_threshold=1.0)  # robo_rugby/__init__.py:28-34
        self.has_grumpy = p.nr_grumpy > 0
        # GameEnv.__init__ (RR_EnvBase.py:111-116): CONFIG_RANDOM keeps what _set_random_positions returned at
        # construction, a given lst_starting_config ([[(x, y, rot) x NR], [(x, y) x NB]], one layout for every arena or a
        # pair of [N, ...] arrays) is applied at once; either way reset(bln_randomize_pos=False) goes back to it.
        if lst_starting_config is None:
            st = self.get_state()
            self._start_robots = st["robots"][:, :, [0, 1, 6]].contiguous()
            self._start_balls = torch.cat([st["balls"][:, :, :2], torch.zeros_like(st["balls"][:, :, :2])], dim=2).contiguous()
        else:
            if len(lst_starting_config) != 2:  # RR_EnvBase.py:135-136
                raise Exception(f"Expected list of 2 lists. Not whatever this is: {lst_starting_config}")
            r = torch.as_tensor(np.asarray(lst_starting_config[0], dtype=np.float64), device=self.device)
            b = torch.as_tensor(np.asarray(lst_starting_config[1], dtype=np.float64), device=self.device)
            if r.shape[-2:] != (p.nr, 3):  # RR_EnvBase.py:138-139
                raise Exception(f"Robot count mismatch. {p.nr} != {r.shape[-2] if r.dim() >= 2 else r.shape}.")
            if b.shape[-2:] != (p.nb, 2):  # RR_EnvBase.py:141-142
                raise Exception(f"Ball count mismatch. {p.nb} != {b.shape[-2] if b.dim() >= 2 else b.shape}.")
            self._start_robots = r.expand(self.num_envs, p.nr, 3).contiguous()
            b = b.expand(self.num_envs, p.nb, 2)
            self._start_balls = torch.cat([b, torch.zeros_like(b)], dim=2).contiguous()
            self._reset_to_start(None, None)
        if self.step_budget_clocks:
            self._seed_bout()

    # ---------------------------------------------------------------- helpers
    @property
    def unwrapped(self):
        return self

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _new(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def _seed_bout(self):
        """Budgeted mode: the persistent step outputs, with both teams' observation rows holding the CURRENT observation -- a row
        that is NOT_READY in the next step() is not written by the kernel and must read as the arena's previous observation
        (called wherever a budget is switched on or the arenas are rewritten from outside: reset, set_state, set_poses)."""
        N = self.num_envs
        if self._bout is None:
            self._bout = (self._new((N, 11), torch.float32), torch.zeros(N, dtype=torch.float32, device=self.device),
                          torch.zeros(N, dtype=torch.uint8, device=self.device),
                          self._new((N, 11), torch.float32) if self.has_grumpy else None,
                          torch.zeros(N, dtype=torch.float32, device=self.device), torch.zeros(N, dtype=torch.int32, device=self.device))
        _lib.check(self._lib.rr_observe(self._h, 1, -1, -1, _ptr(self._bout[0]), self._stream()), "rr_observe", self._lib)
        if self.has_grumpy:
            _lib.check(self._lib.rr_observe(self._h, -1, -1, -1, _ptr(self._bout[3]), self._stream()), "rr_observe", self._lib)

    # ---------------------------------------------------------------- gym surface
    def _reset_to_start(self, mask, obs):
        _lib.check(self._lib.rr_reset_to_poses(self._h, _ptr(mask), _ptr(self._start_robots), _ptr(self._start_balls), _ptr(obs),
                                               None, self._stream()), "rr_reset_to_poses", self._lib)

    def _mask(self, mask):
        """uint8 [N] device mask; scalars / wrong sizes are rejected (the kernels index mask[arena] for every arena)."""
        if mask is None:
            return None
        if isinstance(mask, (bool, int, float)) or (hasattr(mask, "ndim") and mask.ndim == 0):
            raise TypeError("reset(mask=...): a per-arena mask of num_envs booleans is required, not a scalar "
                            "(the reference's reset(False) is reset(bln_randomize_pos=False) here)")
        m = torch.as_tensor(mask).to(device=self.device, dtype=torch.uint8).contiguous().view(-1)
        if m.numel() != self.num_envs:
            raise ValueError(f"reset(mask=...): {m.numel()} entries for {self.num_envs} arenas")
        return m

    def reset(self, mask=None, *, bln_randomize_pos=True):
        """env.reset(bln_randomize_pos) (RR_EnvBase.py:202-216) for all arenas, or those where mask is True: a fresh random
        placement (_set_random_positions), or with bln_randomize_pos=False the start configuration kept since construction
        (_set_starting_positions, RR_EnvBase.py:131-153; main.py:107 replays its layout that way)."""
        N = self.num_envs
        obs = self._new((N, 11), torch.float32)
        mask = self._mask(mask)
        if mask is not None:
            # rows that are not reset keep their current observation
            _lib.check(self._lib.rr_observe(self._h, 1, -1, -1, _ptr(obs), self._stream()), "rr_observe", self._lib)
        if bln_randomize_pos:
            _lib.check(self._lib.rr_reset(self._h, _ptr(mask), _ptr(obs), None, self._stream()), "rr_reset", self._lib)
        else:
            self._reset_to_start(mask, obs)
        if self._bout is not None or self.step_budget_clocks:  # budgeted mode: a NOT_READY row of the next step keeps these observations
            self._seed_bout()
        return obs if self.obs_kind == 0 else self.get_game_state(1)

    def starting_positions(self):
        """The retained start configuration in the reference's format per arena (GameEnv._lst_starting_positions):
        (robots [N,NR,3] x, y, rot ; balls [N,NB,2] x, y)."""
        return self._start_robots.clone(), self._start_balls[:, :, :2].clone()

    def step(self, actions, out=None):
        """GameEnv_Simple.step for every arena.  actions: int tensor [N] or [N,NA] (NA <= robots; action i drives
        robot i, happy robots first -- Training_DQN_pytorch.py:341 passes NA=1).  `out` may carry preallocated
        (obs, reward, done_u8, obs_g, reward_g, status) tensors to reuse."""
        N = self.num_envs
        a = torch.as_tensor(actions, device=self.device)
        if self.action_mode == "thrust" or a.is_floating_point():
            return self.step_thrust(a)
        if a.dim() == 1:
            a = a.view(N, 1)
        if a.shape[0] != N or a.dim() != 2:
            raise ValueError(f"actions must be [N] or [N,NA]; got {tuple(a.shape)}")
        if a.shape[1] > self.preset.nr:  # RR_EnvBase.py:621-622
            raise Exception(f"{a.shape[1]} commands but only {self.preset.nr} robots.")
        a = a.to(torch.int32).contiguous()
        own = out is None and bool(self.step_budget_clocks)
        if own:
            if self._bout is None:  # rows of parked arenas are not written: they must find their previous observation here
                self._seed_bout()
            out = self._bout
        if out is None:
            obs, rew = self._new((N, 11), torch.float32), self._new((N,), torch.float32)
            done = self._new((N,), torch.uint8)
            obs_g = self._new((N, 11), torch.float32) if self.has_grumpy else None
            rew_g = self._new((N,), torch.float32)
            status = self._new((N,), torch.int32)
        else:
            obs, rew, done, obs_g, rew_g, status = out
        _lib.check(self._lib.rr_step(self._h, _ptr(a), a.shape[1], _ptr(obs), _ptr(rew), _ptr(done), _ptr(obs_g),
                                     _ptr(rew_g), _ptr(status), self._stream()), "rr_step", self._lib)
        if own:  # hand out copies: the persistent buffers are overwritten by the next call
            obs, rew, done, rew_g, status = obs.clone(), rew.clone(), done.clone(), rew_g.clone(), status.clone()
            obs_g = obs_g.clone() if obs_g is not None else None
        if self.obs_kind != 0:
            obs, obs_g = self.get_game_state(1), (self.get_game_state(-1) if self.has_grumpy else None)
        return obs, rew, done.view(torch.bool), DebugInfo(obs_g, rew_g, status)

    def set_step_budget(self, clocks):
        """rr_set_step_budget: switch the budgeted step on (clocks > 0) or off (0: parked arenas finish in the next calls)."""
        _lib.check(self._lib.rr_set_step_budget(self._h, int(clocks)), "rr_set_step_budget", self._lib)
        self.step_budget_clocks = int(clocks)
        if self.step_budget_clocks:
            self._seed_bout()

    def rollout(self, actions, repeat=None, out=None):
        """Open-loop rollout in ONE launch: `actions` int [S, N] / [S, N, NA] steps the batch S times (or, with
        `repeat=S`, int [N] / [N, NA] is applied S times: action repeat / frame skip).  Returns (obs [S,N,11],
        reward [S,N], done [S,N], DebugInfo with [S,...] members): bit-identical to S step() calls, but no arena waits
        for the slowest arena of the batch between steps.  Build-side extension (rr_rollout); the reference's gym API steps
        one call at a time."""
        N = self.num_envs
        a = torch.as_tensor(actions, device=self.device).to(torch.int32)
        if repeat is not None:
            S = int(repeat)
            a = (a.view(N, 1) if a.dim() == 1 else a).contiguous()
            na, rep = a.shape[1], 1
        else:
            if a.dim() == 2:
                a = a.unsqueeze(-1)
            a = a.contiguous()
            S, na, rep = a.shape[0], a.shape[2], 0
        if a.shape[-2] != N or self.action_mode != "discrete" or self.obs_kind != 0:
            raise ValueError("rollout: discrete actions [S,N(,NA)] and the default observer only")
        if out is None:
            out = (self._new((S, N, 11), torch.float32), self._new((S, N), torch.float32), self._new((S, N), torch.uint8),
                   self._new((S, N, 11), torch.float32) if self.has_grumpy else None, self._new((S, N), torch.float32),
                   self._new((S, N), torch.int32))
        obs, rew, done, obs_g, rew_g, status = out
        _lib.check(self._lib.rr_rollout(self._h, _ptr(a), na, S, rep, _ptr(obs), _ptr(rew), _ptr(done), _ptr(obs_g), _ptr(rew_g),
                                        _ptr(status), self._stream()), "rr_rollout", self._lib)
        return obs, rew, done.view(torch.bool), DebugInfo(obs_g, rew_g, status)

    def step_thrust(self, thrust, f64=False):
        """GameEnv.step with continuous (L,R) thrust pairs (RR_EnvBase.py:260-273): float tensor [N, 2*k].  f64=True returns the
        observation / reward in fp64 (rr_step_thrust_f64: the entry's parity checks)."""
        N = self.num_envs
        t = torch.as_tensor(thrust, device=self.device, dtype=torch.float32).contiguous().view(N, -1)
        if t.shape[1] % 2 or t.shape[1] > 2 * self.preset.nr:  # RR_EnvBase.py:270-271
            raise Exception(f"{t.shape[1]} commands but only {self.preset.nr * 2} robot engines.")
        od = torch.float64 if f64 else torch.float32
        obs, rew = self._new((N, 11), od), self._new((N,), od)
        done = self._new((N,), torch.uint8)
        obs_g = self._new((N, 11), od) if self.has_grumpy else None
        rew_g, status = self._new((N,), od), self._new((N,), torch.int32)
        fn = self._lib.rr_step_thrust_f64 if f64 else self._lib.rr_step_thrust
        _lib.check(fn(self._h, _ptr(t), t.shape[1] // 2, _ptr(obs), _ptr(rew), _ptr(done),
                      _ptr(obs_g), _ptr(rew_g), _ptr(status), self._stream()), "rr_step_thrust", self._lib)
        if self.obs_kind != 0:
            obs, obs_g = self.get_game_state(1), (self.get_game_state(-1) if self.has_grumpy else None)
        return obs, rew, done.view(torch.bool), DebugInfo(obs_g, rew_g, status)

    def step_f64(self, actions):
        """Same step with fp64 outputs (parity checks against the fp64 reference arithmetic)."""
        N = self.num_envs
        a = torch.as_tensor(actions, device=self.device)
        a = (a.view(N, 1) if a.dim() == 1 else a).to(torch.int32).contiguous()
        obs, rew = self._new((N, 11), torch.float64), self._new((N,), torch.float64)
        done = self._new((N,), torch.uint8)
        obs_g = self._new((N, 11), torch.float64) if self.has_grumpy else None
        rew_g, status = self._new((N,), torch.float64), self._new((N,), torch.int32)
        _lib.check(self._lib.rr_step_f64(self._h, _ptr(a), a.shape[1], _ptr(obs), _ptr(rew), _ptr(done), _ptr(obs_g),
                                         _ptr(rew_g), _ptr(status), self._stream()), "rr_step_f64", self._lib)
        return obs, rew, done.view(torch.bool), DebugInfo(obs_g, rew_g, status)

    def get_game_state(self, int_team=None, robot_idx=-1, ball_idx=-1, f64=False, observer=None):
        """get_game_state of the configured (or named) observer mixin (RR_Observers.py); None when the reference
        returns None (the team has no robot)."""
        team = 1 if int_team is None else int(int_team)
        kind = self.obs_kind if observer is None else OBSERVERS[observer]
        if kind not in (3, 4) and robot_idx < 0 and ((team == 1 and self.preset.nr_happy == 0) or (team == -1 and self.preset.nr_grumpy == 0)):
            return None
        dim = observer_dim(kind, self.preset.nr, self.preset.nb)
        obs = self._new((self.num_envs, dim), torch.float64 if f64 else torch.float32)
        fn = self._lib.rr_observe_kind_f64 if f64 else self._lib.rr_observe_kind
        _lib.check(fn(self._h, kind, team, int(robot_idx), int(ball_idx), _ptr(obs), dim, self._stream()), "rr_observe_kind", self._lib)
        return obs

    def render(self, mode="human", arena=0):
        """The pygame window (RR_EnvBase.py:218-258) is UI and out of scope: 'human' is a no-op so callers' render()
        stays harmless; 'rgb_array' returns a CPU debug picture of ONE arena (arena width + 300-px dashboard strip like
        the reference's surface) drawn with PIL from the device state."""
        if mode == "human":
            return None
        if mode != "rgb_array":
            raise NotImplementedError("Other mode types not supported.")
        from .render import draw_arena
        st = self.get_state()
        return draw_arena(self.preset, st["robots"][arena].cpu().numpy(), st["balls"][arena].cpu().numpy())

    def seed(self, seed=None):
        """Like the reference (RR_EnvBase.py:568-570) this does not re-seed placement; the reset RNG is keyed at
        construction (`seed=`)."""
        return [seed]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------------------------------------------------------- state exchange / logging
    def get_state(self):
        """Canonical fp64 state (layout in include/roborugby_amd.h): dict of device tensors."""
        p, N = self.preset, self.num_envs
        robots = self._new((N, p.nr, 10), torch.float64)
        robots_i = self._new((N, p.nr, 3), torch.int32)
        balls = self._new((N, p.nb, 8), torch.float64)
        step = self._new((N,), torch.int32)
        _lib.check(self._lib.rr_get_state(self._h, _ptr(robots), _ptr(robots_i), _ptr(balls), _ptr(step), self._stream()),
                   "rr_get_state", self._lib)
        return dict(robots=robots, robots_i=robots_i, balls=balls, step=step)

    def set_scratch_rect(self, xy):
        """Parity build only (exact_trig=True): centre of the reference's module-global scratch rect, [N,2] fp64 -- e.g. the
        `state_inner[..., :2]` a golden trajectory dumped (include/roborugby_amd.h: rr_set_scratch_rect)."""
        xy = torch.as_tensor(xy, dtype=torch.float64, device=self.device).contiguous().view(self.num_envs, 2)
        _lib.check(self._lib.rr_set_scratch_rect(self._h, _ptr(xy), self._stream()), "rr_set_scratch_rect", self._lib)
        torch.cuda.current_stream(self.device).synchronize()

    def get_scratch_rect(self):
        xy = self._new((self.num_envs, 2), torch.float64)
        _lib.check(self._lib.rr_get_scratch_rect(self._h, _ptr(xy), self._stream()), "rr_get_scratch_rect", self._lib)
        return xy

    def get_episode_state(self):
        """Build-side bookkeeping that a checkpoint has to carry next to get_state(): `ints` [N,5] = episode index (keys the
        reset RNG: without it a resumed run would replay the placements of episodes 1, 2, ...), steps in the running
        episode, finished episodes, last episode's length, fault flag; `acc` [N,4] = running / last finished returns."""
        ints, acc = self._new((self.num_envs, 5), torch.int32), self._new((self.num_envs, 4), torch.float64)
        _lib.check(self._lib.rr_get_episode_state(self._h, _ptr(ints), _ptr(acc), self._stream()), "rr_get_episode_state", self._lib)
        return dict(ints=ints, acc=acc)

    def set_episode_state(self, ints, acc):
        ints = torch.as_tensor(ints, dtype=torch.int32, device=self.device).contiguous().view(self.num_envs, 5)
        acc = torch.as_tensor(acc, dtype=torch.float64, device=self.device).contiguous().view(self.num_envs, 4)
        _lib.check(self._lib.rr_set_episode_state(self._h, _ptr(ints), _ptr(acc), self._stream()), "rr_set_episode_state", self._lib)
        torch.cuda.current_stream(self.device).synchronize()

    def checkpoint_state(self):
        """Everything a resumed run needs besides the agent: `env_state` (get_state), `episode` (get_episode_state: the episode index
        keys the reset RNG) and, for the parity build, `scratch_rect` (the reference's module-global rect, part of its state).
        Budgeted mode: take it only after a call in which no row was NOT_READY (set_step_budget(0) + one step() guarantees that):
        rr_set_state on resume clears a parked step, i.e. would drop the rest of that step and the action the arena accepted."""
        ck = dict(env_state=self.get_state(), episode=self.get_episode_state())
        if self.exact_trig:
            ck["scratch_rect"] = self.get_scratch_rect()
        return ck

    def load_checkpoint_state(self, ck):
        st = ck["env_state"]
        self.set_state(st["robots"], st["robots_i"], st["balls"], st["step"])
        if "episode" in ck:
            self.set_episode_state(ck["episode"]["ints"], ck["episode"]["acc"])
        if self.exact_trig and "scratch_rect" in ck:
            self.set_scratch_rect(ck["scratch_rect"])

    def set_state(self, robots, robots_i, balls, step):
        p, N = self.preset, self.num_envs
        robots = torch.as_tensor(robots, dtype=torch.float64, device=self.device).contiguous().view(N, p.nr, 10)
        robots_i = torch.as_tensor(robots_i, dtype=torch.int32, device=self.device).contiguous().view(N, p.nr, 3)
        balls = torch.as_tensor(balls, dtype=torch.float64, device=self.device).contiguous().view(N, p.nb, 8)
        step = torch.as_tensor(step, dtype=torch.int32, device=self.device).contiguous().view(N)
        _lib.check(self._lib.rr_set_state(self._h, _ptr(robots), _ptr(robots_i), _ptr(balls), _ptr(step), self._stream()),
                   "rr_set_state", self._lib)
        torch.cuda.current_stream(self.device).synchronize()  # inputs may be temporaries
        if self._bout is not None:
            self._seed_bout()

    def set_poses(self, robots_xyr, balls_xyv):
        """The reference's lst_starting_config (RR_EnvBase.py:35-52) per arena, plus ball velocities."""
        p, N = self.preset, self.num_envs
        r = torch.as_tensor(robots_xyr, dtype=torch.float64, device=self.device).contiguous().view(N, p.nr, 3)
        b = torch.as_tensor(balls_xyv, dtype=torch.float64, device=self.device).contiguous().view(N, p.nb, 4)
        _lib.check(self._lib.rr_set_poses(self._h, _ptr(r), _ptr(b), self._stream()), "rr_set_poses", self._lib)
        torch.cuda.current_stream(self.device).synchronize()
        if self._bout is not None:
            self._seed_bout()

    def goal_scores(self):
        """Goal.get_score() of (happy, grumpy) goal (RR_Goal.py:87-88): identically 0 on the live path -- the reference
        never feeds its goal bookkeeping, so `done` is purely the step counter -- unless the env was
        built with goal_scoring=True (the opt-in extension): then 500 x (positive - negative balls the goal has consumed)."""
        s = self._new((self.num_envs, 2), torch.int32)
        _lib.check(self._lib.rr_goal_scores(self._h, _ptr(s), self._stream()), "rr_goal_scores", self._lib)
        return s

    def episode_stats(self):
        """(last finished episode return happy, grumpy, its length, number of finished episodes) per arena."""
        N = self.num_envs
        lr, lrg = self._new((N,), torch.float32), self._new((N,), torch.float32)
        ll, cnt = self._new((N,), torch.int32), self._new((N,), torch.int32)
        _lib.check(self._lib.rr_episode_stats(self._h, _ptr(lr), _ptr(lrg), _ptr(ll), _ptr(cnt), self._stream()),
                   "rr_episode_stats", self._lib)
        return lr, lrg, ll, cnt

    def lanes_per_env(self):
        v = C.c_int32()
        _lib.check(self._lib.rr_lanes_per_env(self._h, C.byref(v)), "rr_lanes_per_env", self._lib)
        return v.value

    def state_bytes_per_env(self):
        b = C.c_int64()
        _lib.check(self._lib.rr_state_bytes_per_env(self._h, C.byref(b)), "rr_state_bytes_per_env", self._lib)
        return b.value


class RoboRugbyEnv:
    """Single-arena, reference-shaped view: numpy in/out, Python exceptions where the reference raises.

    `env.step([action]) -> (ndarray[11], float, bool, DebugInfo)` exactly as Training_DQN_pytorch.py:341-343
    consumes it.  Backed by a BatchedRoboRugbyEnv with num_envs=1 on the GPU (no CPU path)."""
    metadata = BatchedRoboRugbyEnv.metadata
    reward_range = BatchedRoboRugbyEnv.reward_range

    def __init__(self, preset="T", device=None, seed=0, time_limit=True, dtype="f64", action_mode="discrete",
                 lst_starting_config=None, **kw):
        self._b = BatchedRoboRugbyEnv(1, preset=preset, device=device, seed=seed, time_limit=time_limit,
                                      auto_reset=False, dtype=dtype, action_mode=action_mode,
                                      lst_starting_config=lst_starting_config, **kw)
        self.observation_space = self._b.observation_space
        self.action_space = self._b.action_space
        self.spec = self._b.spec
        self.preset = self._b.preset
        self._f64 = dtype == "f64" and self._b.obs_kind == 0  # the reference hands out float64 ndarrays

    @property
    def unwrapped(self):
        return self

    def reset(self, bln_randomize_pos=True):
        """RR_EnvBase.py:202-216; reset(False) replays the layout kept since construction (main.py:107)."""
        obs = self._b.reset(bln_randomize_pos=bln_randomize_pos)
        if self._f64:
            obs = self._b.get_game_state(1, f64=True)
        return obs[0].double().cpu().numpy()

    def step(self, lstArgs):
        arr = np.concatenate([np.asarray(a).reshape(-1) for a in lstArgs], axis=None) if len(lstArgs) else np.zeros(0)
        if self._b.action_mode == "thrust":  # GameEnv.step: flat (L, R) thrust pairs (RR_EnvBase.py:269-273)
            if len(arr) > self.preset.nr * 2:
                raise Exception(f"{len(arr)} commands but only {self.preset.nr * 2} robot engines.")
            obs, rew, done, info = self._b.step_thrust(torch.as_tensor(arr.astype(np.float32)).view(1, -1))
        else:
            if len(arr) > self.preset.nr:
                raise Exception(f"{len(arr)} commands but only {self.preset.nr} robots.")
            a = torch.as_tensor(arr.astype(np.int64)).view(1, -1)
            obs, rew, done, info = self._b.step_f64(a) if self._f64 else self._b.step(a)
        st = int(info.status[0]) & STATUS_FLAG_MASK
        for bit, msg in STATUS_BITS.items():
            if st & bit:
                raise ZeroDivisionError(msg) if bit == 32 else Exception(msg)
        g = info.adblGrumpyState[0].double().cpu().numpy() if info.adblGrumpyState is not None else None
        return (obs[0].double().cpu().numpy(), float(rew[0]), bool(done[0]),
                DebugInfo(g, float(info.dblGrumpyScore[0]), st))

    def get_game_state(self, int_team=None, obj_robot=None, obj_ball=None):
        o = self._b.get_game_state(int_team, -1 if obj_robot is None else int(obj_robot),
                                   -1 if obj_ball is None else int(obj_ball), f64=self._f64)
        return None if o is None else o[0].double().cpu().numpy()

    def render(self, mode="human"):
        return self._b.render(mode)

    def seed(self, seed=None):
        return [seed]

    def close(self):
        self._b.close()


def make(env_id="RoboRugbySimpleDuel-v3", num_envs=None, preset="T", **kw):
    """gym.make look-alike (robo_rugby/__init__.py:4-34).  num_envs=None -> reference-shaped single env."""
    if env_id not in ENV_IDS:
        raise KeyError(f"unknown env id {env_id!r}; registered and constructible in the reference: {list(ENV_IDS)}")
    if num_envs is None:
        return RoboRugbyEnv(preset=preset, **kw)
    return BatchedRoboRugbyEnv(num_envs, preset=preset, env_id=env_id, **kw)
