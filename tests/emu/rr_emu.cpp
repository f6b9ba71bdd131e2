// rr_emu.cpp -- host-emulated wave: compiles roborugby_amd/csrc/rr_sim.hpp with g++ where each
// "lane-parallel" phase is a 64-iteration loop.  TEST HARNESS ONLY: it lets the CPU test-suite
// check the kernel's phase logic (lane->task maps, ballot masks, response ordering) against the
// oracle without a GPU.  The product library never links or loads this.
static int g_dbg_substeps = 12;
static int g_dbg_trace = 0;
static int g_dbg_memo = 1; // fixed-point check of the sub-step loop on/off (tests compare both)
static int g_dbg_scrub = 0; // overwrite everything but the persistent record with garbage before each step, then derive(): what a
                            // GPU launch starts from (load_record + derive into an LDS slice that holds another arena's leftovers)
#define RR_NUM_SUBSTEPS g_dbg_substeps
#define RR_EMU_TRACE g_dbg_trace
#include "../../roborugby_amd/csrc/rr_sim.hpp"
#include "../../roborugby_amd/csrc/rr_extras.hpp"
#include <cmath>
#include <cstdlib>
#include <cstring>

using namespace rr;

template <class C> struct Emu {
    Arena<C> A;
    SimParams<typename C::Real> sp;
    Program prog;     // reward keepers in execution order (default = SimpleDuel3)
    bool custom_prog;
    uint32_t snap[Arena<C>::SNAP_WORDS];
    typename C::Real xs[3 * C::NR + 1 + 2 * C::NB]; // on_step_begin snapshot (always taken here: AllCoords_WithPrior reads it)
    int32_t isnap[2 * C::NR];
    int goal_scoring;              // opt-in goal scoring (rr_extras.hpp: goal_step)
    int32_t gs[1 + 2 * C::NB + 4];
    uint32_t park[Arena<C>::PARK_WORDS]; // budgeted step: the parked mid-step state (what the GPU keeps in its side buffer)
    uint32_t park_rng;
};

template <typename R> static void fill_params(SimParams<R> &sp, double W, double H, int game_len, int game_mode,
                                              int time_limit, int auto_reset, uint64_t seed) {
    sp.W = (R)W; sp.H = (R)H;
    double mb = 200000.0 / std::pow(W * W + H * H, .5);
    sp.mult_ball = (R)mb; sp.mult_robot = (R)(mb / 100);
    sp.rob_cdist = (R)std::pow(10.0 * 10.0 + 20.0 * 20.0, .5);
    double hr = 7 * std::pow(2.0, .5) / 2;
    sp.inner_h = (R)hr;
    sp.inner_cdist = (R)std::pow(hr * hr + hr * hr, .5);
    sp.game_len = game_len; sp.game_mode = game_mode; sp.time_limit = time_limit; sp.auto_reset = auto_reset & 1;
    sp.reset_on_fault = (auto_reset >> 1) & 1; // bit 1 of the flag word
    sp.seed = seed; sp.arena_offset = 0; sp.memo = 1; sp.acc_external = 0;
}

template <class C> static void set_state(Emu<C> *e, const double *robots, const int32_t *ri, const double *balls, int step) {
    using R = typename C::Real;
    Arena<C> &A = e->A;
    for (int r = 0; r < C::NR; r++) {
        const double *q = robots + 10 * r;
        A.p.rcx[r] = (R)q[0]; A.p.rcy[r] = (R)q[1]; A.p.rl[r] = (R)q[2]; A.p.rrt[r] = (R)q[3]; A.p.rt[r] = (R)q[4];
        A.p.rb[r] = (R)q[5]; A.p.rrot[r] = (R)q[6]; A.p.px[r] = (R)q[7]; A.p.py[r] = (R)q[8]; A.p.prot[r] = (R)q[9];
        A.i.mc[r] = ri[3 * r]; A.i.thl[r] = ri[3 * r + 1]; A.i.thr[r] = ri[3 * r + 2];
    }
    for (int b = 0; b < C::NB; b++) {
        const double *q = balls + 8 * b;
        A.p.bcx[b] = (R)q[0]; A.p.bcy[b] = (R)q[1]; A.p.bl[b] = (R)q[2]; A.p.brt[b] = (R)q[3]; A.p.bt[b] = (R)q[4];
        A.p.bb[b] = (R)q[5]; A.p.bvx[b] = (R)q[6]; A.p.bvy[b] = (R)q[7];
    }
    A.i.step = step;
    A.i.fault = 0;
    A.i.fzp = 0;
#if RR_CARRY
    A.p.ic[0] = A.p.bcx[C::NB - 1]; A.p.ic[1] = A.p.bcy[C::NB - 1]; // where a sub-step leaves the scratch rect (emu_set_scratch_rect overrides)
#endif
    derive(A, e->sp);
}
template <class C> static void get_state(Emu<C> *e, double *robots, int32_t *ri, double *balls, int32_t *step) {
    Arena<C> &A = e->A;
    for (int r = 0; r < C::NR; r++) {
        double *q = robots + 10 * r;
        q[0] = A.p.rcx[r]; q[1] = A.p.rcy[r]; q[2] = A.p.rl[r]; q[3] = A.p.rrt[r]; q[4] = A.p.rt[r]; q[5] = A.p.rb[r];
        q[6] = A.p.rrot[r]; q[7] = A.p.px[r]; q[8] = A.p.py[r]; q[9] = A.p.prot[r];
        ri[3 * r] = A.i.mc[r]; ri[3 * r + 1] = A.i.thl[r]; ri[3 * r + 2] = A.i.thr[r];
    }
    for (int b = 0; b < C::NB; b++) {
        double *q = balls + 8 * b;
        q[0] = A.p.bcx[b]; q[1] = A.p.bcy[b]; q[2] = A.p.bl[b]; q[3] = A.p.brt[b]; q[4] = A.p.bt[b]; q[5] = A.p.bb[b];
        q[6] = A.p.bvx[b]; q[7] = A.p.bvy[b];
    }
    *step = A.i.step;
}

typedef Cfg<1, 0, 1, 0, double, 64> CT64;
typedef Cfg<2, 2, 4, 4, double, 64> CG64;
typedef Cfg<1, 0, 1, 0, float, 64> CT32;
typedef Cfg<2, 2, 4, 4, float, 64> CG32;
// narrow virtual waves: the same phases run in several rounds of VW lanes (what the packed GPU builds execute)
typedef Cfg<1, 0, 1, 0, double, 4> CT64n;
typedef Cfg<2, 2, 4, 4, double, 16> CG64n;
// the duel shape (1+1 robots, 1+1 balls)
typedef Cfg<1, 1, 1, 1, double, 64> CD64;
typedef Cfg<1, 1, 1, 1, float, 64> CD32;
typedef Cfg<1, 1, 1, 1, double, 4> CD64n;
// a shape that is NOT one of the library's built ones (2+1 robots, 2+3 balls: odd counts, unequal teams): what build_shape_library
// compiles on demand (roborugby_amd/build.py); one lane per entity needs 8 lanes
typedef Cfg<2, 1, 2, 3, double, 64> CX64;
typedef Cfg<2, 1, 2, 3, double, 8> CX64n;
// ... and one that packs at FOUR lanes per arena (1 + 1 robots, 2 + 1 balls): the narrow phases' "fewer than eight lanes" variants with
// more than one ball, a combination none of the built shapes has
typedef Cfg<1, 1, 2, 1, double, 64> CY64;
typedef Cfg<1, 1, 2, 1, double, 4> CY64n;

struct Handle { int kind; void *p; };

#define DISPATCH(h, ...)                                                   \
    switch ((h)->kind) {                                                   \
    case 0: { auto *e = (Emu<CT64> *)(h)->p; typedef CT64 CC; __VA_ARGS__; } break; \
    case 1: { auto *e = (Emu<CG64> *)(h)->p; typedef CG64 CC; __VA_ARGS__; } break; \
    case 2: { auto *e = (Emu<CT32> *)(h)->p; typedef CT32 CC; __VA_ARGS__; } break; \
    case 3: { auto *e = (Emu<CG32> *)(h)->p; typedef CG32 CC; __VA_ARGS__; } break; \
    case 4: { auto *e = (Emu<CT64n> *)(h)->p; typedef CT64n CC; __VA_ARGS__; } break; \
    case 5: { auto *e = (Emu<CG64n> *)(h)->p; typedef CG64n CC; __VA_ARGS__; } break; \
    case 6: { auto *e = (Emu<CD64> *)(h)->p; typedef CD64 CC; __VA_ARGS__; } break; \
    case 7: { auto *e = (Emu<CD32> *)(h)->p; typedef CD32 CC; __VA_ARGS__; } break; \
    case 8: { auto *e = (Emu<CD64n> *)(h)->p; typedef CD64n CC; __VA_ARGS__; } break; \
    case 9: { auto *e = (Emu<CX64> *)(h)->p; typedef CX64 CC; __VA_ARGS__; } break; \
    case 10: { auto *e = (Emu<CX64n> *)(h)->p; typedef CX64n CC; __VA_ARGS__; } break; \
    case 11: { auto *e = (Emu<CY64> *)(h)->p; typedef CY64 CC; __VA_ARGS__; } break; \
    case 12: { auto *e = (Emu<CY64n> *)(h)->p; typedef CY64n CC; __VA_ARGS__; } break; \
    }

extern "C" {
void emu_debug_set_substeps(int k) { g_dbg_substeps = k; }
void emu_set_goal_scoring(Handle *h, int on) { DISPATCH(h, (e->goal_scoring = on, goal_state_clear<CC>(e->gs))); }
void emu_goal_scores(Handle *h, int32_t *s2) {
    DISPATCH(h, { const int32_t *g = e->gs + 1 + 2 * CC::NB; for (int k = 0; k < 2; k++) s2[k] = 500 * (popcount8(g[k]) - popcount8(g[2 + k])); });
}
void emu_debug_trace(int on) { g_dbg_trace = on; }
void emu_debug_memo(int on) { g_dbg_memo = on; }
void emu_debug_scrub(int on) { g_dbg_scrub = on; }
// primitives of the kernel source, for unit tests
double emu_py_mod360(double a) { return py_mod<double>(a, 360.0); }
void emu_sincos(double x, double *s, double *c) { m_sincos(x, *s, *c); }
// preset: 0 = T, 1 = G, 2 = D ; f32: 0/1 ; f32 == 2 selects the narrow-virtual-wave fp64 build
Handle *emu_create(int preset, int f32, double W, double H, int game_len, int game_mode, int time_limit, int auto_reset,
                   uint64_t seed) {
    Handle *h = new Handle;
    h->kind = preset == 4 ? (f32 == 2 ? 12 : 11) : preset == 3 ? (f32 == 2 ? 10 : 9) : preset == 2 ? 6 + f32 : preset + 2 * f32; // presets 3, 4 = X, Y (fp64 only)
    switch (h->kind) {
    case 11: h->p = calloc(1, sizeof(Emu<CY64>)); break;
    case 12: h->p = calloc(1, sizeof(Emu<CY64n>)); break;
    case 9: h->p = calloc(1, sizeof(Emu<CX64>)); break;
    case 10: h->p = calloc(1, sizeof(Emu<CX64n>)); break;
    case 6: h->p = calloc(1, sizeof(Emu<CD64>)); break;
    case 7: h->p = calloc(1, sizeof(Emu<CD32>)); break;
    case 8: h->p = calloc(1, sizeof(Emu<CD64n>)); break;
    case 0: h->p = calloc(1, sizeof(Emu<CT64>)); break;
    case 1: h->p = calloc(1, sizeof(Emu<CG64>)); break;
    case 2: h->p = calloc(1, sizeof(Emu<CT32>)); break;
    case 3: h->p = calloc(1, sizeof(Emu<CG32>)); break;
    case 4: h->p = calloc(1, sizeof(Emu<CT64n>)); break;
    default: h->p = calloc(1, sizeof(Emu<CG64n>)); break;
    }
    DISPATCH(h, fill_params(e->sp, W, H, game_len, game_mode, time_limit, auto_reset, seed); (void)sizeof(CC);
             e->prog.n = 3; e->prog.id[0] = 1; e->prog.id[1] = 2; e->prog.id[2] = 3; e->custom_prog = false;
             for (int r = 0; r < CC::NR; r++) robot_set_clean_lane(e->A, e->sp, r, (typename CC::Real)0, (typename CC::Real)0,
                                                                   r < CC::NRH ? (typename CC::Real)90 : (typename CC::Real)-90);
             for (int b = 0; b < CC::NB; b++) ball_set_clean_lane(e->A, b, (typename CC::Real)0, (typename CC::Real)0,
                                                                  (typename CC::Real)0, (typename CC::Real)0);
             derive(e->A, e->sp));
    return h;
}
void emu_destroy(Handle *h) { free(h->p); delete h; }
void emu_set_state(Handle *h, const double *robots, const int32_t *ri, const double *balls, int step) {
    DISPATCH(h, set_state<CC>(e, robots, ri, balls, step));
}
void emu_get_state(Handle *h, double *robots, int32_t *ri, double *balls, int32_t *step) {
    DISPATCH(h, get_state<CC>(e, robots, ri, balls, step));
}
// centre of the reference's scratch rect (_rectBallInner): 1 if this build carries it (the parity build), 0 otherwise
int emu_set_scratch_rect(Handle *h, const double *xy) {
#if RR_CARRY
    DISPATCH(h, e->A.p.ic[0] = (typename CC::Real)xy[0]; e->A.p.ic[1] = (typename CC::Real)xy[1]);
    return 1;
#else
    (void)h; (void)xy;
    return 0;
#endif
}
int emu_get_scratch_rect(Handle *h, double *xy) {
#if RR_CARRY
    DISPATCH(h, xy[0] = (double)e->A.p.ic[0]; xy[1] = (double)e->A.p.ic[1]);
    return 1;
#else
    (void)h; (void)xy;
    return 0;
#endif
}
void emu_set_poses(Handle *h, const double *rxyr, const double *bxyv) {
    DISPATCH(h, for (int r = 0; r < CC::NR; r++) robot_set_clean_lane(e->A, e->sp, r, (typename CC::Real)rxyr[3 * r],
                                                                   (typename CC::Real)rxyr[3 * r + 1], (typename CC::Real)rxyr[3 * r + 2]);
             for (int b = 0; b < CC::NB; b++) ball_set_clean_lane(e->A, b, (typename CC::Real)bxyv[4 * b], (typename CC::Real)bxyv[4 * b + 1],
                                                                  (typename CC::Real)bxyv[4 * b + 2], (typename CC::Real)bxyv[4 * b + 3]);
             e->A.i.step = 0; e->A.i.fzp = 0; derive(e->A, e->sp));
}
// same bracketing as rr_step: snapshot -> step kernel phases -> keeper program (only for a non-default program)
extern "C++" {
template <class CC> static void scrub_scratch(Emu<CC> *e) {
    // keep P and I (the HBM record), trash the rest of the LDS image, rebuild what the kernel derives after load_record
    typename Arena<CC>::P p = e->A.p;
    typename Arena<CC>::I i = e->A.i;
    memset(&e->A, 0xA5, sizeof(e->A));
    e->A.p = p; e->A.i = i;
    derive(e->A, e->sp);
}
// park_mod < 0: the synchronous step; >= 0: the budgeted step, parking at pseudo-random sub-step boundaries (1 in park_mod; 0 / 1: all)
template <class CC> static int emu_step_t(Emu<CC> *e, const int32_t *actions, const float *thrust, int na, double *obs,
                                          double *obs_g, double *reward, double *reward_g, uint8_t *done, int park_mod = -1) {
    using RR = typename CC::Real;
    int32_t status = 0;
    RR *xs = e->xs;
    Rec<CC> q = { reinterpret_cast<const RR *>(&e->A.p) };
    if (g_dbg_scrub) scrub_scratch(e);
    const bool was_parked = e->A.i.fzp < 0;
    if (!was_parked) extras_begin<CC>(q, xs);
    StepOut<double> o = { obs, obs_g, reward, reward_g, done, &status, g_dbg_memo ? e->snap : nullptr, e->isnap, 0, 0, 0 };
    if (park_mod >= 0) {
        ParkCtx pk;
        pk.buf = e->park; pk.host_rng = &e->park_rng; pk.host_mod = (uint32_t)park_mod;
        step_arena<CC, double, true>(e->A, e->sp, 0, actions, thrust, na, o, pk);
        if (status & ST_NOT_READY) return status;
    } else
    step_arena<CC, double>(e->A, e->sp, 0, actions, thrust, na, o);
    if (e->custom_prog && !(status & (ST_WAS_RESET | ST_STEP_AFTER_DONE))) {
        RR rh, rg;
        extras_end<CC, double>(q, e->sp, xs, e->prog, (uint32_t)status >> 16, reward, reward_g, &status, rh, rg);
    }
    if (e->goal_scoring) { // the arena's LDS image has the record's layout: P, then I
        bool bd = false;
        for (int k = 0; k < e->prog.n; k++) bd |= e->prog.id[k] == KEEPER_BASEDESTRUCTION;
        goal_step<CC, double>(reinterpret_cast<RR *>(&e->A.p), reinterpret_cast<int32_t *>(&e->A.i), e->sp, e->gs, bd, reward, reward_g, done, &status);
    }
    return status;
}
} // extern "C++"
int emu_step(Handle *h, const int32_t *actions, int na, double *obs, double *obs_g, double *reward, double *reward_g,
             uint8_t *done) {
    int32_t status = 0;
    DISPATCH(h, status = emu_step_t<CC>(e, actions, nullptr, na, obs, obs_g, reward, reward_g, done));
    return status;
}
// the budgeted step (rr_sim.hpp: ParkCtx): returns ST_NOT_READY while the step is parked; outputs are written when it completes
int emu_step_budget(Handle *h, const int32_t *actions, int na, double *obs, double *obs_g, double *reward, double *reward_g,
                    uint8_t *done, int park_mod) {
    int32_t status = 0;
    DISPATCH(h, status = emu_step_t<CC>(e, actions, nullptr, na, obs, obs_g, reward, reward_g, done, park_mod < 0 ? 0 : park_mod));
    return status;
}
void emu_park_seed(Handle *h, uint32_t seed) { DISPATCH(h, e->park_rng = seed); }
int emu_step_thrust(Handle *h, const float *thrust, int nk, double *obs, double *obs_g, double *reward, double *reward_g,
                    uint8_t *done) {
    int32_t status = 0;
    DISPATCH(h, status = emu_step_t<CC>(e, nullptr, thrust, nk, obs, obs_g, reward, reward_g, done));
    return status;
}
void emu_set_program(Handle *h, const int32_t *ids, int n) {
    DISPATCH(h, e->prog.n = n; for (int i = 0; i < n; i++) e->prog.id[i] = ids[i];
             e->custom_prog = !(n == 3 && ids[0] == 1 && ids[1] == 2 && ids[2] == 3));
}
int emu_observe_kind(Handle *h, int kind, int team, int ridx, int bidx, double *out) {
    int m = 0;
    DISPATCH(h, if (kind == 0) { int st = 0; m = observe<CC, double>(e->A, e->sp, team, ridx, bidx, out, st) ? 11 : 0; }
                else { Rec<CC> q = { reinterpret_cast<const typename CC::Real *>(&e->A.p) };
                       m = observe_kind<CC, double>(q, e->sp, kind, team, ridx, bidx, out, e->xs); });
    return m;
}
int emu_observe(Handle *h, int team, int ridx, int bidx, double *obs) {
    int st = 0, ok = 0;
    DISPATCH(h, ok = observe<CC, double>(e->A, e->sp, team, ridx, bidx, obs, st) ? 1 : 0);
    return ok;
}
int emu_reset(Handle *h, uint64_t arena, uint64_t episode) {
    int st = 0;
    DISPATCH(h, (reset_arena(e->A, e->sp, arena, episode, st), goal_state_clear<CC>(e->gs)));
    return st;
}
}
