/*
 * roborugby_amd.h -- C-ABI of the MI355X-native batched RoboRugby simulator.
 *
 * The reference (harman097/RoboRugby) has no FFI: its boundary is the gym.Env Python API
 * (`reset/step/get_game_state`, robo_rugby/gym_env/RR_EnvBase.py:202-216,260-297,550-559).  This
 * header is the thin C layer beneath the Python mirror of that API (roborugby_amd/env.py): plain
 * pointers and sizes, no torch types.  Every data pointer is a DEVICE pointer owned by the caller
 * (PyTorch-ROCm allocations); every call is enqueued on `stream` (a hipStream_t, NULL = default
 * stream) and returns without synchronising.  One handle per (process, device); a handle is not
 * re-entrant.  Return value: 0 = OK, <0 = API misuse (see rr_last_error()).  Physics faults -- the
 * places where the reference raises from inside step() -- never cross the ABI as errors: they are
 * reported per arena in `status` (RR_STATUS_* bits).
 *
 * Shapes use N = num_envs, NR = nr_happy + nr_grumpy, NB = nb_pos + nb_neg.
 */
#ifndef ROBORUGBY_AMD_H
#define ROBORUGBY_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RR_ABI_VERSION 4

/* per-arena status bits; each mirrors one exception site of the reference */
#define RR_STATUS_BOT_RESOLVE_FAIL 1   /* RR_EnvBase.py:313  "UNABLE TO RESOLVE BOT/BOT COLLISIONS"   */
#define RR_STATUS_BOT_STUCK 2          /* RR_EnvBase.py:328  "ROBOTS STUCK FROM PRIOR FRAME"          */
#define RR_STATUS_UNDO_MOVE_FAIL 4     /* RR_EnvBase.py:325  "UNABLE TO UNDO MOVE FOR ROBOT"          */
#define RR_STATUS_UNDO_FAIL 8          /* RR_EnvBase.py:421  "UNABLE TO RESOLVE ALL COLLISIONS..."    */
#define RR_STATUS_SAME_SPOT 16         /* RR_TrashyPhysics.py:250 balls in the exact same spot        */
#define RR_STATUS_DIV0 32              /* MyUtils.py:25      Div0(0, 0)                               */
#define RR_STATUS_STEP_AFTER_DONE 64   /* RR_EnvBase.py:262  "Game is over" (auto_reset = 0 only)     */
#define RR_STATUS_BAD_ACTION 128       /* action outside 0..7 (KeyError in RR_EnvBase.py:606)         */
#define RR_STATUS_UNDO_WARN 256        /* RR_EnvBase.py:419  GAME_MODE warning, step continued        */
#define RR_STATUS_RESET_GAVE_UP 512    /* spawn rejection sampling hit its attempt cap                */
#define RR_STATUS_WAS_RESET 1024       /* this call reset the arena instead of stepping it            */
#define RR_STATUS_NOT_READY 16384       /* budgeted step only: this arena's step is still in progress (see step_budget_clocks) */
#define RR_STATUS_FLAG_MASK 0xFFFF     /* bits 0-15 are the flags above ...                            */
#define RR_STATUS_NAUGHTY_SHIFT 16     /* ... bits 16+r: robot r joined NaughtyBots' set this step (RR_ScoreKeepers.py:123-128) */

#define RR_DTYPE_F64 0 /* state + arithmetic in fp64: the parity mode (the reference is fp64)         */
#define RR_DTYPE_F32 1 /* state + arithmetic in fp32: the fast mode                                   */
#define RR_DTYPE_F32_STATE 2 /* "fp32 state" (BASELINE config 2): the arenas' records in HBM are fp32, a step computes in fp64 between
                                loading a record and writing it back -- one rounding per stored value and step, single-step results
                                within 1e-5 of the fp64 reference on contact steps too.  Default lane widths, rr_step / rr_step_f64 /
                                rr_step_thrust(_f64) and every side entry; no rr_rollout, no step budget (both would keep unrounded
                                state across steps) */

#define RR_TEAM_HAPPY 1   /* RR_Constants.py:52 */
#define RR_TEAM_GRUMPY (-1) /* RR_Constants.py:53 */

#define RR_OBS_DIM 11 /* SingleBall_6wayLidar_v2, RR_Observers.py:394-406 */

typedef struct rr_env rr_env; /* opaque */

typedef struct rr_config {
    int32_t struct_size;    /* = sizeof(rr_config), for forward compatibility                          */
    int32_t num_envs;       /* N arenas stepped in lockstep by this handle                             */
    int32_t nr_happy, nr_grumpy, nb_pos, nb_neg; /* RR_Constants.py:30-34; shapes inside libroborugby_amd.so: (1,0,1,0), (2,2,4,4), (1,1,1,1);
                                                    any other counts (<= 8 robots, <= 11 balls, <= 32 ball-robot pairs): a one-shape library of the same
                                                    sources and ABI, hipcc -DRR_CUSTOM_SHAPE ... (roborugby_amd/build.py: build_shape_library) */
    double arena_w, arena_h;                     /* RR_Constants.py:6-7                                */
    int32_t game_len_steps; /* RR_Constants.py:25                                                      */
    int32_t game_mode;      /* RR_Constants.py:4: only selects the undo-loop fault rule (EnvBase:417-421) */
    int32_t time_limit;     /* 1: done = step_count >= T (what gym's TimeLimit wrapper reports to
                               Training_DQN_pytorch.py); 0: done = step_count > T (raw RR_EnvBase.py:555-559) */
    int32_t auto_reset;     /* 1: rr_step on a finished arena resets it (status WAS_RESET, reward 0, done 0)
                               instead of flagging STEP_AFTER_DONE                                     */
    int32_t reset_on_fault; /* 1: a step that ends with a fatal status (the places where the reference raises or
                               spins forever: bits 1|2|4|8|16|32) reports done = 1 and ends the episode, so that
                               auto_reset re-places the arena; 0: the arena keeps stepping with the bit set     */
    int32_t dtype;          /* RR_DTYPE_F64 | RR_DTYPE_F32 | RR_DTYPE_F32_STATE                        */
    int32_t device;         /* HIP device ordinal                                                      */
    uint64_t seed;          /* keys the counter-based reset RNG                                        */
    uint64_t arena_offset;  /* global id of local arena 0: makes results invariant to sharding         */
    uint32_t step_budget_clocks; /* 0 (default): every rr_step call completes every arena's step -- the reference's semantics
                               (RR_EnvBase.py:260-297), and the call lasts as long as its slowest arena.  > 0: the BUDGETED step,
                               an opt-in extension for contact-rich policies: an arena whose wavefront has run for more than this
                               many shader clocks at the end of an expensive physics sub-step (RR_EnvBase.py:275-287 runs 12 per
                               step) parks there; the call reports RR_STATUS_NOT_READY for it -- reward 0, done 0, its obs / obs_g
                               rows NOT written (reuse the buffers to keep the previous ones) -- and the next call resumes it where
                               it stopped, IGNORING the action it is given.  Each arena's trajectory as a function of the actions
                               it accepted is bit-identical to step_budget_clocks = 0.  Works with every reward program, observer, prior-step
                               tracking and the goal-scoring mode: their side kernels take the on_step_begin copies when an arena's step
                               BEGINS and run its on_step_end with the call that completes it (switch prior-step tracking on before the
                               budget, or after a call without NOT_READY rows). */
    uint32_t reserved_;     /* 0                                                                       */
} rr_config;

int rr_abi_version(void);
/* 1 in the parity build (libroborugby_amd_exact.so: the same sources with -DRR_EXACT_TRIG=1), 0 in the default library.  The parity
 * build (a) evaluates sin / cos of the robot kinematics in double-double, ~correctly rounded, so that they agree with the reference's
 * math.sin / math.cos (glibc) on 99.85 % of the evaluations instead of 97 %, and (b) carries the centre of the reference's scratch rect
 * (rr_get/set_scratch_rect below): free-running episodes then follow the reference bit for bit -- most of them to their last step --
 * at ~80 % of the default library's speed. */
int rr_exact_trig(void);
const char *rr_last_error(void); /* thread-local description of the last <0 return */

/* Allocates the per-arena state (SoA records in HBM) on cfg->device and places every arena like the
 * reference constructor does (RR_EnvBase.py:111-116 -> _set_random_positions). */
int rr_create(const rr_config *cfg, rr_env **out);
int rr_destroy(rr_env *env);

/* env.reset() (RR_EnvBase.py:202-216) for the arenas whose mask byte is non-zero (mask == NULL: all).
 * obs [N,11] float32, obs_g [N,11] float32 (nullable; NaN rows when there is no grumpy robot). */
int rr_reset(rr_env *env, const uint8_t *mask, float *obs, float *obs_g, void *stream);

/* GameEnv_Simple.step (RR_EnvBase.py:617-626 -> :260-297): actions [N,na] int32 in 0..7, action i
 * drives robot i (happy robots first); robots without an action keep their thrust.  Outputs (all
 * nullable except obs/reward/done): obs [N,11] f32, reward [N] f32, done [N] u8, obs_g [N,11] f32,
 * reward_g [N] f32 (info.adblGrumpyState / info.dblGrumpyScore, RR_EnvBase.py:562-566),
 * status [N] i32. */
int rr_step(rr_env *env, const int32_t *actions, int32_t na, float *obs, float *reward, uint8_t *done,
            float *obs_g, float *reward_g, int32_t *status, void *stream);
/* Same step with fp64 outputs (full-precision parity checks). */
int rr_step_f64(rr_env *env, const int32_t *actions, int32_t na, double *obs, double *reward, uint8_t *done,
                double *obs_g, double *reward_g, int32_t *status, void *stream);
/* Changes rr_config.step_budget_clocks of a live handle (host-side; takes effect with the next rr_step).  0 switches parking
 * off -- arenas that are parked at that moment finish their step in the next call(s) as usual. */
int rr_set_step_budget(rr_env *env, uint32_t clocks);
/* Open-loop rollout: nsteps consecutive rr_step calls per arena in ONE launch -- the record stays in LDS, and no arena
 * waits for the slowest arena of the batch between steps (a launch per step ends when its slowest wavefront does).
 * actions [nsteps, N, na] (repeat == 0) or [N, na] applied at every step (repeat != 0: the action-repeat / frame-skip
 * wrapper of RL practice); outputs are stacked per step: obs [nsteps, N, 11], reward / done / status [nsteps, N]
 * (obs_g, reward_g, status nullable).  Bit-identical to calling rr_step nsteps times (auto-reset included).  The reference
 * has no counterpart (gym steps one call at a time); policies that need step s's observation to choose step s+1's
 * action use rr_step. */
int rr_rollout(rr_env *env, const int32_t *actions, int32_t na, int32_t nsteps, int32_t repeat, float *obs, float *reward,
               uint8_t *done, float *obs_g, float *reward_g, int32_t *status, void *stream);

/* Continuous entry, GameEnv.step (RR_EnvBase.py:260-273): thrust [N,2*nk] float32 (L,R per robot),
 * rounded half-to-even like Python's round() (RR_Robot.py:100-102). */
int rr_step_thrust(rr_env *env, const float *thrust, int32_t nk, float *obs, float *reward, uint8_t *done,
                   float *obs_g, float *reward_g, int32_t *status, void *stream);
/* Same entry with fp64 outputs (the thrust entry's observation / reward parity checks; RR_DTYPE_F64 / RR_DTYPE_F32_STATE handles only). */
int rr_step_thrust_f64(rr_env *env, const float *thrust, int32_t nk, double *obs, double *reward, uint8_t *done,
                       double *obs_g, double *reward_g, int32_t *status, void *stream);

/* get_game_state(int_team=team) (RR_Observers.py:301-406) of the current state; robot_idx/ball_idx
 * = -1 selects the team's first robot / positive ball 0 like the reference defaults. */
int rr_observe(rr_env *env, int32_t team, int32_t robot_idx, int32_t ball_idx, float *obs, void *stream);
int rr_observe_f64(rr_env *env, int32_t team, int32_t robot_idx, int32_t ball_idx, double *obs, void *stream);

/* Full state exchange in the canonical fp64 layout (same as tests/golden/traj_*.npz):
 *   robots   [N,NR,10] cx,cy,left,right,top,bottom,rot,prev_x,prev_y,prev_rot (prev = pose ring entry
 *            moveCount-1, RR_Robot.py:43-58; NaN = absent)
 *   robots_i [N,NR,3]  moveCount,lthrust,rthrust
 *   balls    [N,NB,8]  cx,cy,left,right,top,bottom,vx,vy
 *   step     [N]       lngStepCount */
int rr_set_state(rr_env *env, const double *robots, const int32_t *robots_i, const double *balls,
                 const int32_t *step, void *stream);
int rr_get_state(rr_env *env, double *robots, int32_t *robots_i, double *balls, int32_t *step, void *stream);
/* Poses only -- the reference's lst_starting_config format (RR_EnvBase.py:35-52,131-153) plus ball
 * velocities: robots_xyr [N,NR,3], balls_xyv [N,NB,4]; edges re-derived, history cleared. */
int rr_set_poses(rr_env *env, const double *robots_xyr, const double *balls_xyv, void *stream);
/* env.reset(bln_randomize_pos=False) (RR_EnvBase.py:202-216 -> _set_starting_positions :131-153): the arenas whose mask
 * byte is non-zero (mask == NULL: all) restart at the given start configuration -- robots_xyr [N,NR,3], balls_xyv [N,NB,4]
 * (the reference's lst_starting_config, zero ball velocities) -- with step count 0, thrust 0, no pose history; obs / obs_g
 * (nullable) receive the first observations of the masked arenas.  The caller keeps the layout (the Python mirror retains
 * the construction placement exactly like GameEnv.__init__ keeps self._lst_starting_positions, RR_EnvBase.py:111-116). */
int rr_reset_to_poses(rr_env *env, const uint8_t *mask, const double *robots_xyr, const double *balls_xyv, float *obs,
                      float *obs_g, void *stream);
/* Episode bookkeeping for checkpoint / resume (build-side; the reference pickles the agent and never the env):
 * ints [N,5] = episode index (keys the reset RNG), steps in the running episode, finished episodes, length of the last
 * finished episode, fault flag; acc [N,4] = running return happy / grumpy, last finished return happy / grumpy. */
/* Parity build only (rr_exact_trig() == 1; the default library returns -3): the centre of the reference's module-global scratch
 * rect `_rectBallInner` (RR_TrashyPhysics.py:26-36), xy [N,2] fp64, device pointers.  Its centre setters are relative moves
 * (MyUtils.py:266-275), so the centre the next ball diameter is built from depends, in the last bits, on where the previous user left
 * it -- state that survives resets and, in a process that runs several envs, episodes.  The parity build keeps it in each arena's
 * record; rr_set_state puts it on the last ball (where every sub-step leaves it), rr_set_scratch_rect seeds it from a dumped
 * reference state (tests/golden/traj_*.npz `state_inner`), rr_reset / rr_set_poses leave it alone like the reference does. */
int rr_get_scratch_rect(rr_env *env, double *xy, void *stream);
int rr_set_scratch_rect(rr_env *env, const double *xy, void *stream);
int rr_get_episode_state(rr_env *env, int32_t *ints, double *acc, void *stream);
int rr_set_episode_state(rr_env *env, const int32_t *ints, const double *acc, void *stream);

/* ---- the reference's other mixins; SimpleDuel3's own stack is the default and is fused
 * into the step kernel, anything else is evaluated by light side kernels around it.
 * Reward keepers, given in on_step_end EXECUTION order (each keeper calls super() first, except NaughtyBots which
 * never does, so keepers behind it in the MRO do not run): 1 NaughtyBots, 2 ChasePosBall, 3 PushPosBallsToGoal,
 * 4 DontDriveInGoals, 5 KeepMovingGuys, 6 BaseDestruction, 7 PushNegBallsFromGoal (RR_ScoreKeepers.py:46-179).
 * Default {1,2,3}.  n <= 8. */
int rr_set_reward_program(rr_env *env, const int32_t *keeper_ids, int32_t n);
/* Observers: kind 0 SingleBall_6wayLidar_v2 (11 values), 1 SingleBall_6wayLidar (11, RR_Observers.py:168-285),
 * 2 PosBall_BasicLidar (5, :116-166), 3 AllCoords (3*NR + 2*NB, :47-83), 4 AllCoords_WithPrior (6*NR + 4*NB, :86-110:
 * every robot x, y, rot + its rectDblPriorStep x, y, rot, every ball x, y + prior-step x, y; needs
 * rr_track_prior_step).  obs [N, out_dim]; rows are NaN where the reference returns None.  out_dim must equal the
 * kind's size. */
int rr_observe_kind(rr_env *env, int32_t kind, int32_t team, int32_t robot_idx, int32_t ball_idx, float *obs,
                    int32_t out_dim, void *stream);
int rr_observe_kind_f64(rr_env *env, int32_t kind, int32_t team, int32_t robot_idx, int32_t ball_idx, double *obs,
                        int32_t out_dim, void *stream);

/* on != 0: every rr_step / rr_step_thrust first snapshots what the sprites' on_step_begin copies (rectDblPriorStep,
 * RR_Robot.py:116-117, RR_Ball.py:60-61) so that observer kind 4 can report it; the call itself (and rr_reset) seeds the
 * copies from the current poses (the reference holds the stale pre-placement pose until the first step). */
int rr_track_prior_step(rr_env *env, int32_t on, void *stream);

/* Opt-in goal scoring -- an EXTENSION: on the reference's live path the goals never score (Goal.track_balls / update_score are
 * only reached from the never-called GameEnv.__old_step, RR_EnvBase.py:458-520; RR_Goal.py:80 calls a property as a function),
 * so there is no reference behaviour to match; this is the mechanism of RR_Goal.py:54-91 + the commented block at
 * RR_EnvBase.py:494-512 made to work as SURVEY 8(f)-3 words the intent: a ball inside a goal triangle for 150 consecutive steps
 * is consumed (out of play from then on: parked at x <= -1000), +-500 points (happy team +500 for a positive ball in the happy
 * goal / a negative one in the grumpy goal, the grumpy team the opposite), 3 negative balls destroy a goal, and a destroyed goal
 * or an empty field ends the episode (done = 1, status bits below; BaseDestruction pays when it is in the keeper program).
 * Ordering: the step that consumes a ball still OBSERVES it inside the goal (the goal bookkeeping runs after the step's
 * observation is taken); the ball is out of play from the next observation on.
 * rr_goal_scores: Goal.get_score() of (happy, grumpy) goal per arena, int32 [N,2]; all zero while the mode is off. */
#define RR_STATUS_GOAL_H_DESTROYED 2048
#define RR_STATUS_GOAL_G_DESTROYED 4096
#define RR_STATUS_NO_BALLS 8192
int rr_set_goal_scoring(rr_env *env, int32_t on, void *stream);
int rr_goal_scores(rr_env *env, int32_t *scores, void *stream);

/* Logging: return/length of the last finished episode and the number of finished episodes per arena
 * (the caller accumulates `score` the same way, Training_DQN_pytorch.py:345-346). */
int rr_episode_stats(rr_env *env, float *last_return, float *last_return_g, int32_t *last_len,
                     int32_t *episodes_done, void *stream);

/* Scripted on-device policy of the contact-rich benchmark stream (the reference's scripted players
 * live in robo_rugby/gym_env/RR_Players.py): robot 0 of every arena turns toward its ball (observation values 0 and 1:
 * bot angle, ball angle) or drives forward when within 8 degrees, and with probability `noise` takes a random action; robots
 * 1..na-1 act at random.  obs [N,11] f32 (the rows rr_step / rr_reset wrote), actions [N,na] i32.  The random draws are a
 * function of (seed, global arena id, step index) only; the step index is step_of[arena] when step_of != NULL (per-arena
 * counters, e.g. "steps accepted so far" under the budgeted step), else `step`.  One launch, nothing else. */
int rr_policy_chase(rr_env *env, const float *obs, const int32_t *step_of, uint32_t step, float noise, uint64_t seed,
                    int32_t *actions, int32_t na, void *stream);

/* ---- config 5 (BASELINE.json): the learn step of the reference's DQN agent (Training_DQN_pytorch.py:25-67 DeepQNetwork
 * Linear 11 -> 256 -> 256 -> 8 + ReLU, MSELoss, Adam; :151-191 DQNAgent.learn) fused into two launches for the wide batches of the
 * batched trainer: forward of Q_eval(s) and Q_target(s'), TD target, loss gradient, backward and the weight-gradient reduction in
 * one kernel (fp32 matrix cores, activations resident in LDS), partial-gradient reduction + Adam in the second
 * (roborugby_amd/csrc/rr_dqn.hip).  Every pointer is a device pointer; the parameters stay PyTorch's (updated in place), the
 * handle owns the Adam moments and the scratch.  Same math as torch autograd up to fp32 summation order. */
typedef struct rr_dqn rr_dqn; /* opaque */
typedef struct rr_dqn_args {
    int32_t struct_size;            /* = sizeof(rr_dqn_args) */
    int32_t batch;                  /* samples per update: a positive multiple of 64 */
    float *eval_params[6];          /* fc1.weight [256,11], fc1.bias [256], fc2.weight [256,256], fc2.bias [256], fc3.weight [8,256],
                                       fc3.bias [8] of Q_eval (row-major, torch.nn.Linear layout): updated in place */
    const float *target_params[6];  /* the same six tensors of Q_target */
    const float *state_memory;      /* replay memory (Training_DQN_pytorch.py:95-104): [M,11] */
    const float *new_state_memory;  /* [M,11] */
    const int64_t *action_memory;   /* [M] */
    const float *reward_memory;     /* [M] */
    const uint8_t *terminal_memory; /* [M] (torch.bool) */
    const int64_t *batch_index;     /* [batch] sampled rows of the memory (NULL: rows 0 .. batch-1) */
    float gamma, lr, beta1, beta2, eps; /* :79 gamma; torch.optim.Adam defaults 1e-3 / .9 / .999 / 1e-8 unless set (:42 lr) */
    float *loss_out;                /* nullable: the batch's mean squared TD error (MSELoss) */
} rr_dqn_args;
int rr_dqn_create(int32_t device, rr_dqn **out);
int rr_dqn_destroy(rr_dqn *dqn);
const char *rr_dqn_last_error(void);
/* One gradient step: Q_eval's parameters <- Adam(grad of MSE(r + gamma max_a' Q_target(s'), Q_eval(s)[a])). */
int rr_dqn_update(rr_dqn *dqn, const rr_dqn_args *args, void *stream);
/* The same gradient without the update: grads [rr_dqn_param_count()] in the order fc2.weight, fc1.weight, fc3.weight, fc1.bias,
 * fc2.bias, fc3.bias (the parity tests' window). */
int rr_dqn_grads(rr_dqn *dqn, const rr_dqn_args *args, float *grads, void *stream);
int32_t rr_dqn_param_count(void);
/* DQNAgent.choose_action (Training_DQN_pytorch.py:138-149) for n observations (a multiple of 64) in one launch: actions[i] =
 * argmax_a Q(obs[i]) (first maximum), or with probability epsilon a uniform action -- the draw is a function of (seed, i, call),
 * so pass a fresh `call` counter every time.  params = the six tensors of the network to act with (rr_dqn_args.eval_params
 * order); qvalues [n,8] is optional (NULL: not written). */
int rr_dqn_act(rr_dqn *dqn, const float *const params[6], const float *obs, int32_t n, float epsilon, uint64_t seed, uint32_t call,
               int32_t *actions, float *qvalues, void *stream);
/* DQNAgent.store_transition (Training_DQN_pytorch.py:126-136) for n transitions in one launch: the rows whose `valid` byte is set
 * (NULL: all) are appended to the replay ring at mem_cntr, mem_cntr + 1, ... (mod mem_size) in row order; count_out (device int32,
 * nullable) receives how many.  state / new_state [n,11] f32, action [n] i32, reward [n] f32, done [n] u8 (torch.bool). */
int rr_dqn_store(rr_dqn *dqn, const float *state, const int32_t *action, const float *reward, const float *new_state,
                 const uint8_t *done, const uint8_t *valid, int32_t n, int64_t mem_cntr, int64_t mem_size, float *state_memory,
                 float *new_state_memory, int64_t *action_memory, float *reward_memory, uint8_t *terminal_memory,
                 int32_t *count_out, void *stream);
/* Adam moments (same order as rr_dqn_grads) and step count out of / into the handle: checkpoint / resume. */
int rr_dqn_adam_state(rr_dqn *dqn, float *exp_avg, float *exp_avg_sq, int64_t *step, int32_t set, void *stream);

/* Introspection used by bench.py for the roofline line: bytes of the per-arena HBM record, and how many lanes of
 * a wavefront work on one arena (64 = one wavefront per arena; smaller = several arenas packed per wavefront). */
int rr_state_bytes_per_env(const rr_env *env, int64_t *bytes);
/* HBM copy probe for the roofline line: copies `bytes` (a multiple of 16, 16-byte aligned device buffers) from
 * src to dst with a 16-B-per-lane grid-stride kernel on the current device; the caller times it (HIP events) and quotes
 * 2 * bytes / time next to the 8 TB/s spec. */
int rr_probe_hbm_copy(void *dst, const void *src, size_t bytes, void *stream);
int rr_lanes_per_env(const rr_env *env, int32_t *lanes);

#ifdef __cplusplus
}
#endif
#endif /* ROBORUGBY_AMD_H */
