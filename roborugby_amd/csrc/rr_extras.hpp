// rr_extras.hpp -- the reference's OTHER reward / observation mixins, kept off the hot kernel.
//
// SimpleDuel3's stack (NaughtyBots + ChasePosBall + PushPosBallsToGoal, SingleBall_6wayLidar_v2) is fused into k_step.
// Any other keeper stack / observer is evaluated by the light thread-per-arena kernels below, straight from the HBM
// records: k_extras_begin snapshots what on_step_begin captures (rectDblPriorStep copies, ball_dist_sum), k_step runs
// and publishes the NaughtyBots set in status bits 16.., k_extras_end re-evaluates the whole keeper program in the
// reference's on_step_end order and overwrites the rewards; k_observe_kind serves the other observers.
#pragma once
#include "rr_sim.hpp"

namespace rr {

enum : int { KEEPER_NAUGHTY = 1, KEEPER_CHASE = 2, KEEPER_PUSHPOS = 3, KEEPER_DONTDRIVE = 4, KEEPER_KEEPMOVING = 5,
             KEEPER_BASEDESTRUCTION = 6, KEEPER_PUSHNEG = 7 };
enum : int { OBS_V2 = 0, OBS_V1 = 1, OBS_BASIC = 2, OBS_ALLCOORDS = 3, OBS_ALLCOORDS_PRIOR = 4 };
// words of the on_step_begin snapshot per arena: rectDblPriorStep copies of the robots (cx, cy, rot), ball_dist_sum,
// rectDblPriorStep copies of the balls (cx, cy)
template <class C> constexpr int xs_stride() { return 3 * C::NR + 1 + 2 * C::NB; }

struct Program { int32_t n; int32_t id[8]; };

// read-only view of one arena's HBM record (field-major layout of Arena<C>::P)
template <class C> struct Rec {
    using R = typename C::Real;
    const typename C::Store *p; // (fp32 under the F32State policy: widened on read)
    RR_HD R rcx(int r) const { return (R)p[0 * C::NR + r]; }
    RR_HD R rcy(int r) const { return (R)p[1 * C::NR + r]; }
    RR_HD R rrot(int r) const { return (R)p[6 * C::NR + r]; }
    RR_HD R bcx(int b) const { return (R)p[10 * C::NR + 0 * C::NB + b]; }
    RR_HD R bcy(int b) const { return (R)p[10 * C::NR + 1 * C::NB + b]; }
};
template <class C> RR_HD void rec_corners(const Rec<C> &q, const SimParams<typename C::Real> &sp, int r, V2<typename C::Real> c[4]) {
    using R = typename C::Real;
    R rel[8];
    corners_for<R>(q.rrot(r), (R)10, (R)20, sp.rob_cdist, rel);
    for (int k = 0; k < 4; k++) c[k] = { q.rcx(r) + rel[2 * k], q.rcy(r) + rel[2 * k + 1] };
}
template <typename R> RR_HD Seg<R> side_of(const V2<R> c[4], int s) { Seg<R> g = { c[side_a(s)], c[side_b(s)] }; return g; }

// RightTriangle.contains_point (MyUtils.py:438-445) for the two goal triangles (RR_Goal.py:14-28):
// happy goal  (W,H),(W,H-240),(W-240,H): hypotenuse origin (W-240, H), vector to (W, H-240)
// grumpy goal (240,0),(0,240),(0,0)    : hypotenuse origin (240, 0),   vector to (0, 240)
template <typename R> RR_HD bool goal_contains(bool happy, R W, R H, V2<R> p, int &st) {
    const R l = happy ? W - (R)240 : (R)0, r = happy ? W : (R)240, t = happy ? H - (R)240 : (R)0, b = happy ? H : (R)240;
    if (!((l <= p.x && p.x <= r) && (t <= p.y && p.y <= b))) return false;
    const V2<R> h0 = happy ? V2<R>{ l, b } : V2<R>{ r, t }, h1 = happy ? V2<R>{ r, t } : V2<R>{ l, b };
    const R sh = div0<R>(h1.y - h0.y, h1.x - h0.x, st), spn = div0<R>(p.y - h0.y, p.x - h0.x, st);
    return spn >= sh;
}

// ---- on_step_begin snapshot: [3*NR] rectDblPriorStep copies (cx, cy, rot) + ball_dist_sum + [2*NB] ball copies
template <class C> RR_HD void extras_begin(const Rec<C> &q, typename C::Real *xs) {
    using R = typename C::Real;
    for (int r = 0; r < C::NR; r++) { // FloatRect.copy(): new rect (0,20,0,40) -> center setter -> rotation setter
        xs[3 * r + 0] = (R)10 + (q.rcx(r) - (R)10);
        xs[3 * r + 1] = (R)20 + (q.rcy(r) - (R)20);
        xs[3 * r + 2] = norm360<R>(q.rrot(r));
    }
    R s = (R)0;
    V2<R> o = { (R)0, (R)0 };
    for (int b = 0; b < C::NBP; b++) { V2<R> c = { q.bcx(b), q.bcy(b) }; s = s + dist<R>(o, c); }
    xs[3 * C::NR] = s;
    for (int b = 0; b < C::NB; b++) { // Ball.on_step_begin (RR_Ball.py:60-61): FloatRect.copy() of a 14 x 14 rect
        xs[3 * C::NR + 1 + 2 * b] = (R)7 + (q.bcx(b) - (R)7);
        xs[3 * C::NR + 2 + 2 * b] = (R)7 + (q.bcy(b) - (R)7);
    }
}
// ---- on_step_end: the keeper program in execution order
template <class C, typename O>
RR_HD void extras_end(const Rec<C> &q, const SimParams<typename C::Real> &sp, const typename C::Real *xs, const Program &pg,
                      uint32_t naughty, O *reward, O *reward_g, int32_t *status, typename C::Real &rh, typename C::Real &rg) {
    using R = typename C::Real;
    rh = (R)0; rg = (R)0;
    int st = 0;
    for (int k = 0; k < pg.n; k++) {
        switch (pg.id[k]) {
        case KEEPER_NAUGHTY:
            for (int r = 0; r < C::NR; r++) if (naughty & (1u << r)) { if (r < C::NRH) rh -= (R).005; else rg -= (R).005; }
            break;
        case KEEPER_CHASE:
            for (int r = 0; r < C::NR; r++)
                for (int b = 0; b < C::NBP; b++) {
                    V2<R> rc = { q.rcx(r), q.rcy(r) }, pc = { xs[3 * r], xs[3 * r + 1] }, bc = { q.bcx(b), q.bcy(b) };
                    R now = dist<R>(rc, bc), prior = dist<R>(pc, bc);
                    if (r < C::NRH) rh += (prior - now) * sp.mult_robot; else rg += (prior - now) * sp.mult_robot;
                }
            break;
        case KEEPER_PUSHPOS:
        case KEEPER_PUSHNEG: { // PushNegBallsFromGoal sums the POSITIVE balls too (reference bug kept), signs flipped
            R s = (R)0;
            V2<R> o = { (R)0, (R)0 };
            for (int b = 0; b < C::NBP; b++) { V2<R> c = { q.bcx(b), q.bcy(b) }; s = s + dist<R>(o, c); }
            R delta = s - xs[3 * C::NR];
            if (pg.id[k] == KEEPER_PUSHPOS) { rh += delta * sp.mult_ball; rg -= delta * sp.mult_ball; }
            else { rh -= delta * sp.mult_ball; rg += delta * sp.mult_ball; }
        } break;
        case KEEPER_DONTDRIVE:
            for (int r = 0; r < C::NR; r++) {
                V2<R> c[4];
                rec_corners(q, sp, r, c);
                bool in = false;
                for (int g = 0; g < 2 && !in; g++) // happy goal first, then grumpy (RR_ScoreKeepers.py:76-77)
                    for (int k2 = 0; k2 < 4 && !in; k2++) in = goal_contains<R>(g == 0, sp.W, sp.H, c[k2], st);
                if (in) { if (r < C::NRH) rh -= (R).005; else rg -= (R).005; }
            }
            break;
        case KEEPER_KEEPMOVING:
            for (int r = 0; r < C::NR; r++)
                if (q.rcx(r) == xs[3 * r] && q.rcy(r) == xs[3 * r + 1] && q.rrot(r) == xs[3 * r + 2]) {
                    if (r < C::NRH) rh -= (R).005; else rg -= (R).005;
                }
            break;
        default: break; // KEEPER_BASEDESTRUCTION: goals are never destroyed on the live path
        }
    }
    *reward = (O)rh;
    if (reward_g) *reward_g = (O)rg;
    if (status && st) *status |= st;
}

// ---- opt-in goal scoring.  On the reference's live path the goals never score: Goal.track_balls / update_score are only
// called from GameEnv.__old_step (RR_EnvBase.py:458-520, never called), update_score itself calls the property `is_positive`
// as a function (RR_Goal.py:80) and the code that consumed the balls is commented out (:494-512).  This is that mechanism made
// to work, as SURVEY 8(f)-3 words the intent -- there is no reference behaviour to be equal to:
//   * a ball whose centre stays inside a goal triangle (RightTriangle.contains_point, RR_Goal.py:71-72) for
//     TIME_BALL_IN_GOAL_STEPS = 150 consecutive steps (RR_Constants.py:27-28; the dctBallsPrior / dctBallsCurrent hand-over of
//     RR_Goal.py:54-85 keeps the step of entry) is consumed by that goal (`sprBall.kill()`): out of play from then on (parked,
//     see ball_in_play);
//   * it scores POINTS_BALL_SCORED = 500 for the happy team when a positive ball lands in the happy goal or a negative one in the
//     grumpy goal, -500 otherwise; the grumpy team gets the opposite (the commented block at :494-512);
//   * Goal.get_score = 500 x (positive - negative balls it consumed); a goal with MAX_NEG_BALLS = 3 negative balls is destroyed
//     (RR_Goal.py:87-91); game_is_done also ends the game when a goal is destroyed or no ball is left (RR_EnvBase.py:555-559);
//   * BaseDestruction (RR_ScoreKeepers.py:99-109), when it is in the keeper stack, pays POINTS_GOAL_DESTROYED on that step.
// Per-arena bookkeeping `gs` (int32): [0] steps since reset (Goal.lngFrameCount), [1 + g NB + b] the step at which ball b's
// current stay in goal g began (-1: not inside), [1 + 2 NB + g] / [3 + 2 NB + g] bit masks of the positive / negative balls
// goal g has consumed (g = 0 happy, 1 grumpy).
template <class C> constexpr int gs_stride() { return 1 + 2 * C::NB + 4; }
template <class C> RR_HD void goal_state_clear(int32_t *gs) {
    gs[0] = 0;
    for (int k = 0; k < 2 * C::NB; k++) gs[1 + k] = -1;
    for (int k = 0; k < 4; k++) gs[1 + 2 * C::NB + k] = 0;
}
RR_HD int popcount8(int32_t m) { int n = 0; for (int k = 0; k < 16; k++) n += (m >> k) & 1; return n; }
// rec / irec: the arena's HBM record (mutable: a consumed ball is parked, the episode bookkeeping follows an early end)
template <class C, typename O>
RR_HD void goal_step(typename C::Store *rec, int32_t *irec, const SimParams<typename C::Real> &sp, int32_t *gs, bool base_destruction,
                     O *reward, O *reward_g, uint8_t *done, int32_t *status) {
    using R = typename C::Real;
    constexpr int NR = C::NR, NB = C::NB, BALLS = 10 * NR, ACC = 10 * NR + 8 * NB, I0 = 3 * NR; // irec: step episode ep_len ep_count last_len fault
    int st = *status;
    if (st & ST_WAS_RESET) { goal_state_clear<C>(gs); return; } // Goal.on_reset (RR_Goal.py:47-52)
    if (st & (ST_STEP_AFTER_DONE | ST_NOT_READY)) return; // (budgeted step: a step still in progress is no frame yet)
    const int frame = ++gs[0];
    R delta = (R)0;
    int alive = 0, dummy = 0;
    for (int g = 0; g < 2; g++)
        for (int b = 0; b < NB; b++) {
            const bool in_play = rec[BALLS + b] > (R)-900;
            V2<R> c = { rec[BALLS + b], rec[BALLS + NB + b] };
            const bool in = in_play && goal_contains<R>(g == 0, sp.W, sp.H, c, dummy);
            int since = gs[1 + g * NB + b];
            if (in) {
                if (since >= 0 && frame - since >= 150) { // consumed
                    const bool pos = b < C::NBP;
                    gs[1 + 2 * NB + (pos ? 0 : 2) + g] |= 1 << b;
                    delta += ((g == 0) == pos) ? (R)500 : (R)-500;
                    const R x = park_x<R>(b), y = park_y<R>();
                    rec[BALLS + b] = x; rec[BALLS + NB + b] = y;
                    rec[BALLS + 2 * NB + b] = x - (R)7; rec[BALLS + 3 * NB + b] = x + (R)7;
                    rec[BALLS + 4 * NB + b] = y - (R)7; rec[BALLS + 5 * NB + b] = y + (R)7;
                    rec[BALLS + 6 * NB + b] = (R)0; rec[BALLS + 7 * NB + b] = (R)0;
                    irec[I0 + 6] = 0; // a ball left the field: whatever island the step ended with is not carried over (Arena::I::fzp)
                    since = -1;
                } else if (since < 0) since = frame;
            } else since = -1;
            gs[1 + g * NB + b] = since;
        }
    for (int b = 0; b < NB; b++) alive += rec[BALLS + b] > (R)-900 ? 1 : 0;
    const bool dh = popcount8(gs[1 + 2 * NB + 2]) >= 3, dg = popcount8(gs[1 + 2 * NB + 3]) >= 3;
    R rh = delta, rg = -delta;
    if (base_destruction && (dh || dg)) { // (the reference pays the HAPPY team in both branches: kept)
        const R P = (R)(500 + 200000) * (R)NB; // POINTS_GOAL_DESTROYED, RR_Constants.py:48
        rh += P; rg -= P;
    }
    if (rh != (R)0 || rg != (R)0) {
        *reward = (O)((R)*reward + rh);
        if (reward_g) *reward_g = (O)((R)*reward_g + rg);
        rec[ACC + 0] += rh; rec[ACC + 1] += rg;
        if (*done) { rec[ACC + 2] = rec[ACC + 0]; rec[ACC + 3] = rec[ACC + 1]; }
    }
    st |= (dh ? ST_GOAL_H_DESTROYED : 0) | (dg ? ST_GOAL_G_DESTROYED : 0) | (alive == 0 ? ST_NO_BALLS : 0);
    if ((dh || dg || alive == 0) && !*done) { // game_is_done (RR_EnvBase.py:555-559): the episode ends here
        *done = 1;
        irec[I0 + 5] = 1; // over: a later call re-places the arena (auto_reset) or flags STEP_AFTER_DONE
        irec[I0 + 4] = irec[I0 + 2]; irec[I0 + 3] += 1;
        rec[ACC + 2] = rec[ACC + 0]; rec[ACC + 3] = rec[ACC + 1];
    }
    *status = st;
}

// ---- serial two_way_lidar_rect (RR_TrashyPhysics.py:365-391) over the other robots + the walls
template <class C>
RR_HD void lidar_serial(const Rec<C> &q, const SimParams<typename C::Real> &sp, int ridx, V2<typename C::Real> a,
                        V2<typename C::Real> b, typename C::Real &front, typename C::Real &back) {
    using R = typename C::Real;
    R f = inf_<R>(), bk = inf_<R>();
    Seg<R> ray = { a, b };
    int st = 0;
    for (int j = 0; j < C::NR; j++) {
        V2<R> c[4];
        if (j < C::NR - 1) {
            rec_corners(q, sp, j < ridx ? j : j + 1, c);
        } else {
            const R hx = sp.W / (R)2, hy = sp.H / (R)2; // rect_walls = FloatRect(0, W, 0, H)
            c[TL] = { hx + -hx, hy + -hy }; c[TR] = { hx + hx, hy + -hy }; c[BL] = { hx + -hx, hy + hy }; c[BR] = { hx + hx, hy + hy };
        }
        for (int s = 0; s < 4; s++) {
            V2<R> I = line_intersection<R>(side_of<R>(c, s), ray, st);
            R de = dist<R>(I, b), ds = dist<R>(I, a);
            if (de <= ds && de < f) f = de;
            if (ds <= de && ds < bk) bk = ds;
        }
    }
    front = f; back = bk;
}
// the other observers; returns the number of values written (0 = the reference returns None)
template <class C, typename O>
RR_HD int observe_kind(const Rec<C> &q, const SimParams<typename C::Real> &sp, int kind, int team, int ridx, int bidx, O *o,
                       const typename C::Real *xs = nullptr) {
    using R = typename C::Real;
    int st = 0;
    if (kind == OBS_ALLCOORDS_PRIOR) { // AllCoords_WithPrior (RR_Observers.py:86-110); xs = the on_step_begin snapshot
        if (!xs) return 0;
        int n = 0;
        const int f0 = team == -1 ? C::NRH : 0, f1 = team == -1 ? C::NR : C::NRH, s0 = team == -1 ? 0 : C::NRH, s1 = team == -1 ? C::NRH : C::NR;
        for (int pass = 0; pass < 2; pass++)
            for (int r = pass ? s0 : f0; r < (pass ? s1 : f1); r++) {
                o[n++] = (O)q.rcx(r); o[n++] = (O)q.rcy(r); o[n++] = (O)q.rrot(r);
                o[n++] = (O)xs[3 * r]; o[n++] = (O)xs[3 * r + 1]; o[n++] = (O)xs[3 * r + 2];
            }
        for (int b = 0; b < C::NB; b++) {
            o[n++] = (O)q.bcx(b); o[n++] = (O)q.bcy(b);
            o[n++] = (O)xs[3 * C::NR + 1 + 2 * b]; o[n++] = (O)xs[3 * C::NR + 2 + 2 * b];
        }
        return n;
    }
    if (kind == OBS_ALLCOORDS) { // AllCoords (RR_Observers.py:47-83): own team's robots first
        int n = 0;
        const int f0 = team == -1 ? C::NRH : 0, f1 = team == -1 ? C::NR : C::NRH, s0 = team == -1 ? 0 : C::NRH, s1 = team == -1 ? C::NRH : C::NR;
        for (int r = f0; r < f1; r++) { o[n++] = (O)q.rcx(r); o[n++] = (O)q.rcy(r); o[n++] = (O)q.rrot(r); }
        for (int r = s0; r < s1; r++) { o[n++] = (O)q.rcx(r); o[n++] = (O)q.rcy(r); o[n++] = (O)q.rrot(r); }
        for (int b = 0; b < C::NB; b++) { o[n++] = (O)q.bcx(b); o[n++] = (O)q.bcy(b); }
        return n;
    }
    if (ridx < 0) {
        if (team == 1 && C::NRH == 0) return 0;
        if (team == -1 && C::NRG == 0) return 0;
        ridx = team == 1 ? 0 : C::NRH;
    }
    V2<R> c[4];
    rec_corners(q, sp, ridx, c);
    V2<R> rc = { q.rcx(ridx), q.rcy(ridx) };
    if (kind == OBS_BASIC) { // PosBall_BasicLidar._robot_state (RR_Observers.py:143-166)
        V2<R> bc = { q.bcx(0), q.bcy(0) };
        R ball_angle = py_mod<R>(angle_degrees<R>(rc, bc, st) + (R)360, (R)360);
        R ball_dist = m_abs(dist<R>(rc, bc));
        Seg<R> top = side_of<R>(c, 1), bot = side_of<R>(c, 3);
        V2<R> mt = { (top.a.x + top.b.x) / (R)2, (top.a.y + top.b.y) / (R)2 }, mb = { (bot.a.x + bot.b.x) / (R)2, (bot.a.y + bot.b.y) / (R)2 };
        R lf, lb;
        lidar_serial(q, sp, ridx, mb, mt, lf, lb);
        o[0] = (O)q.rrot(ridx); o[1] = (O)ball_angle; o[2] = (O)ball_dist; o[3] = (O)lf; o[4] = (O)lb;
        return 5;
    }
    // OBS_V1: SingleBall_6wayLidar (RR_Observers.py:184-285)
    if (bidx < 0) bidx = 0;
    Seg<R> fr = side_of<R>(c, 0), bk = side_of<R>(c, 2);
    V2<R> mf = { (fr.a.x + fr.b.x) / (R)2, (fr.a.y + fr.b.y) / (R)2 }, mbk = { (bk.a.x + bk.b.x) / (R)2, (bk.a.y + bk.b.y) / (R)2 };
    R lf, lb, lfl, lbr, lfr, lbl;
    lidar_serial(q, sp, ridx, mbk, mf, lf, lb);
    lidar_serial(q, sp, ridx, c[BL], c[TR], lfl, lbr);
    lidar_serial(q, sp, ridx, c[TL], c[BR], lfr, lbl);
    V2<R> bc = { q.bcx(bidx), q.bcy(bidx) }, good = { sp.W, sp.H }, bad = { (R)0, (R)0 };
    R ball_angle = angle_degrees<R>(rc, bc, st), ball_dist = dist<R>(rc, bc);
    R goal_angle = angle_degrees<R>(rc, good, st), bot_angle = q.rrot(ridx), goal_dist;
    const bool ball_neg = bidx >= C::NBP;
    if ((team == 1 && !ball_neg) || (team == -1 && ball_neg)) {
        goal_dist = dist<R>(rc, good);
    } else {
        goal_dist = dist<R>(rc, bad);
        ball_angle = py_mod<R>(ball_angle + (R)180, (R)360);
        goal_angle = py_mod<R>(goal_angle + (R)180, (R)360);
        bot_angle = py_mod<R>(bot_angle + (R)180, (R)360);
    }
    const R cap = (R)150;
    o[0] = (O)bot_angle; o[1] = (O)ball_angle; o[2] = (O)py_min<R>(ball_dist, cap); o[3] = (O)goal_angle; o[4] = (O)py_min<R>(goal_dist, cap);
    o[5] = (O)py_min<R>(lf, cap); o[6] = (O)py_min<R>(lfl, cap); o[7] = (O)py_min<R>(lfr, cap); o[8] = (O)py_min<R>(lb, cap);
    o[9] = (O)py_min<R>(lbl, cap); o[10] = (O)py_min<R>(lbr, cap);
    return 11;
}

} // namespace rr
