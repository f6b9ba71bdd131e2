"""fp32 fast mode (RR_DTYPE_F32) against the reference's fp64 vectors: how the comparison is scored.

north_star asks "fp32 positions within 1e-5"; SURVEY section 7 (hard part 1) makes that precise: a single step from
synchronised state, |delta| <= 1e-5 * max(1, |x|) -- one fp32 ulp at |x| in [512, 1024) is already 6.1e-5 absolute.  That
bar is asserted on QUIET steps (no contact response, no undo, nothing near a wall or another entity: the arithmetic is
the kinematics only).  On the other steps a contact response amplifies round-off and a knife-edge predicate can flip, so
they are held to the error distribution DESIGN.md documents instead.  `quiet_mask` classifies the golden steps from the
reference's own pre/post state, without running anything."""
import numpy as np


def free_roll(balls, substeps=12):
    """Ball.move x 12 with no force (RR_Ball.py:78-105) in the reference's operation order: what a ball does when
    nothing touches it.  balls [n, NB, 8] = cx, cy, l, r, t, b, vx, vy; returns the same layout."""
    b = balls.copy()
    for _ in range(substeps):
        for c, lo, hi, v in ((0, 2, 3, 6), (1, 4, 5, 7)):
            nl = b[..., lo] + b[..., v]
            d = nl - b[..., lo]
            b[..., c] += d; b[..., lo] += d; b[..., hi] += d
        for v in (6, 7):
            b[..., v] *= 0.995
            b[..., v] = np.where(np.abs(b[..., v]) < 0.005, 0.0, b[..., v])
    return b


def quiet_mask(pre_r, pre_ri, pre_b, post_r, post_ri, post_b, W, H):
    """True where the reference's step was pure kinematics: every ball ended exactly where a free roll puts it (bit for
    bit -- any push / bounce / undo breaks that), every robot kept all 12 moves (no undo), and no robot came within 3 px
    of a wall (a move the wall blocks is a knife-edge comparison) or within 60 px of another robot or 45 px of a ball."""
    roll = free_roll(pre_b)
    q = np.all(roll[..., [0, 1, 6, 7]] == post_b[..., [0, 1, 6, 7]], axis=(1, 2))
    q &= np.all(post_ri[..., 0] - pre_ri[..., 0] == 12, axis=1)
    for r in (pre_r, post_r):
        q &= np.all((r[..., 2] > 3) & (r[..., 3] < W - 3) & (r[..., 4] > 3) & (r[..., 5] < H - 3), axis=1)
    NR, NB = pre_r.shape[1], pre_b.shape[1]
    for i in range(NR):
        for j in range(i + 1, NR):
            q &= np.hypot(pre_r[:, i, 0] - pre_r[:, j, 0], pre_r[:, i, 1] - pre_r[:, j, 1]) > 60
        for j in range(NB):
            q &= np.hypot(pre_r[:, i, 0] - pre_b[:, j, 0], pre_r[:, i, 1] - pre_b[:, j, 1]) > 45
    # a ball near a wall: collided_wall works on the int-truncated rect, again a knife edge
    for b in (pre_b, post_b):
        q &= np.all((b[..., 2] > 2) & (b[..., 3] < W - 2) & (b[..., 4] > 2) & (b[..., 5] < H - 2), axis=1)
    return q


def rel_err(got, ref):
    """max over an arena's values of |got - ref| / max(1, |ref|); NaNs must coincide."""
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    e = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    e = np.where(np.isnan(e), 0.0, e)
    return e.reshape(e.shape[0], -1).max(axis=1)


EDGE_FLOOR = 10.0  # px; see score()


def score(pre, post, got, W, H):
    """pre/post: the reference's state dicts (robots, robots_i, balls); got: ours after one fp32 step from `pre`.
    Returns (quiet mask, per-arena error, integer-state equality per arena).  The per-arena error is the largest of
      * |delta| / max(1, |x|) over the POSITIONS (centres of robots and balls) and the ball velocities,
      * the rotation error on the circle, relative to its 360-degree range,
      * |delta| / max(10, |x|) over the AABB edges.
    Why the edges get a 10-px floor: they are bookkeeping derived from centre + rotated corner offsets, and the fp32
    rotation has a quantum of ulp(360) = 3e-5 degrees that every turn re-rounds -- through the 22.4-px corner arm that is
    up to 12 * 1.5e-5 * pi/180 * 22.4 = 7e-5 px per step whatever the edge's own magnitude, and an edge can sit 3 px from
    the origin while the centre it belongs to cannot come closer than 10 px (half a robot's width)."""
    q = quiet_mask(pre["robots"], pre["robots_i"], pre["balls"], post["robots"], post["robots_i"], post["balls"], W, H)
    e = np.maximum(rel_err(got["robots"][..., :2], post["robots"][..., :2]), rel_err(got["balls"][..., [0, 1, 6, 7]], post["balls"][..., [0, 1, 6, 7]]))
    drot = np.abs((got["robots"][..., 6] - post["robots"][..., 6] + 180.0) % 360.0 - 180.0) / 360.0
    e = np.maximum(e, drot.max(axis=1))
    for got_e, ref_e in ((got["robots"][..., 2:6], post["robots"][..., 2:6]), (got["balls"][..., 2:6], post["balls"][..., 2:6])):
        ee = np.abs(got_e - ref_e) / np.maximum(EDGE_FLOOR, np.abs(ref_e))
        e = np.maximum(e, ee.reshape(len(e), -1).max(axis=1))
    ints = np.all(got["robots_i"] == post["robots_i"], axis=(1, 2))
    return q, e, ints


def oracle_on_rounded_state(preset, pre, actions):
    """The fp64 oracle (= the reference's arithmetic, bit for bit on tests/golden) stepping every arena of `pre` ONCE from its state
    ROUNDED TO FP32, result rounded to fp32: what an ideal "fp32 state" implementation returns.  pre: robots [n,NR,10], robots_i,
    balls [n,NB,8], step [n]; actions [n,NA] (-1 = no action for that robot).  Returns the state dict after the step."""
    import oracle_lib as ol
    r32 = lambda x: np.asarray(x, np.float64).astype(np.float32).astype(np.float64)
    n = len(pre["step"])
    out = {"robots": np.zeros_like(pre["robots"]), "robots_i": np.zeros_like(pre["robots_i"]), "balls": np.zeros_like(pre["balls"])}
    for i in range(n):
        o = ol.OracleEnv(preset)
        o.set_state(r32(pre["robots"][i]), pre["robots_i"][i], r32(pre["balls"][i]), None, int(pre["step"][i]))
        a = np.asarray(actions[i]); a = a[a >= 0]
        o.step(a.astype(np.int32))
        st = o.get_state()
        out["robots"][i] = r32(st["robots"]); out["balls"][i] = r32(st["balls"]); out["robots_i"][i] = st["robots_i"]
    return out
