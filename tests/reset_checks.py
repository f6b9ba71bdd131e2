"""Shared checks of reset layouts against the reference's `_set_random_positions` (RR_EnvBase.py:155-200).

The reference places sprites with Python's global Mersenne Twister; the build uses a counter-based Philox stream, so
placements cannot be compared draw for draw.  What CAN be pinned to the 1,000 layouts per preset that the imported
reference produced (tests/golden/reset_*.npz):
  * the support: per-coordinate integer ranges of RR_EnvBase.py:163-167,188-189, end points included;
  * the rejection rule: no int-AABB (pygame.Rect) overlap between robots, and between a ball and the goal boxes /
    robots / other balls (RR_EnvBase.py:168-197) -- evaluated with the SAME predicate on the reference's layouts;
  * the distribution: two-sample Kolmogorov-Smirnov on every marginal and a chi-square on the balls' 2-D occupancy
    (which shows the goal-box hole), reference sample vs ours;
  * the first observation the reference returned for its layouts.
Used by the CPU test of the oracle and by the -m gpu test of the kernel's reset."""
import numpy as np
from scipy import stats


def int_rect(l, t, r, b):
    """pygame.Rect(l, t, r - l, b - t): every argument truncated toward zero (C int); arrays in, (l, t, w, h) out."""
    return np.trunc(l).astype(np.int64), np.trunc(t).astype(np.int64), np.trunc(r - l).astype(np.int64), np.trunc(b - t).astype(np.int64)


def collide(a, b):
    """pygame colliderect: strict overlap."""
    al, at, aw, ah = a
    bl, bt, bw, bh = b
    return (al < bl + bw) & (at < bt + bh) & (al + aw > bl) & (at + ah > bt)


def overlaps(robots, balls, W, H):
    """Counts of int-AABB overlaps in layouts given as canonical state arrays robots [n,NR,>=6], balls [n,NB,>=6]
    (cx, cy, left, right, top, bottom, ...)."""
    n, NR = robots.shape[:2]
    NB = balls.shape[1]
    rr = [int_rect(robots[:, i, 2], robots[:, i, 4], robots[:, i, 3], robots[:, i, 5]) for i in range(NR)]
    bb = [int_rect(balls[:, i, 2], balls[:, i, 4], balls[:, i, 3], balls[:, i, 5]) for i in range(NB)]
    z = np.zeros(n, np.int64)
    goal_h = (z + int(W - 240), z + int(H - 240), z + 240, z + 240)  # RR_Goal.py:30-35 via get_rect(center=...)
    goal_g = (z, z, z + 240, z + 240)
    out = dict(robot_robot=0, ball_goal=0, ball_robot=0, ball_ball=0)
    for i in range(NR):
        for j in range(i + 1, NR):
            out["robot_robot"] += int(collide(rr[i], rr[j]).sum())
    for i in range(NB):
        out["ball_goal"] += int(collide(bb[i], goal_h).sum()) + int(collide(bb[i], goal_g).sum())
        for j in range(NR):
            out["ball_robot"] += int(collide(bb[i], rr[j]).sum())
        for j in range(i + 1, NB):
            out["ball_ball"] += int(collide(bb[i], bb[j]).sum())
    return out


def check_support(robots, balls, W, H, need_endpoints=True):
    """Integer draws inside the reference's ranges, and (big samples) reaching both ends of each range."""
    rx, ry, rrot = robots[:, :, 0], robots[:, :, 1], robots[:, :, 6 if robots.shape[2] > 3 else 2]
    bx, by = balls[:, :, 0], balls[:, :, 1]
    for name, v, lo, hi in (("robot x", rx, 80, W - 80), ("robot y", ry, 40, H - 40), ("robot rot", rrot, 0, 360),
                            ("ball x", bx, 40, W - 40), ("ball y", by, 40, H - 40)):
        assert np.array_equal(v, np.rint(v)), name
        # random.randint(0, 360) followed by the rotation setter's (rot+720)%360: 360 comes back as 0
        hi_seen = hi if name != "robot rot" else 359
        assert v.min() >= lo and v.max() <= hi_seen, (name, v.min(), v.max())
        if need_endpoints:
            assert v.min() == lo and v.max() == hi_seen, (name, v.min(), v.max())


def ks_all(ref_r, ref_b, our_r, our_b, rot_col, alpha=0.01):
    """Two-sample KS on every (entity, coordinate) marginal, Bonferroni-corrected."""
    tests = []
    for i in range(ref_r.shape[1]):
        for c_ref, c_our, nm in ((0, 0, "x"), (1, 1, "y"), (2, rot_col, "rot")):
            a, b = ref_r[:, i, c_ref], our_r[:, i, c_our]
            if nm == "rot":
                a = np.mod(a, 360.0)
            tests.append((f"robot{i}.{nm}", stats.ks_2samp(a, b)))
    for i in range(ref_b.shape[1]):
        for c, nm in ((0, "x"), (1, "y")):
            tests.append((f"ball{i}.{nm}", stats.ks_2samp(ref_b[:, i, c], our_b[:, i, c])))
    thr = alpha / len(tests)
    bad = [(n, float(t.statistic), float(t.pvalue)) for n, t in tests if t.pvalue < thr]
    assert not bad, bad
    return max(float(t.statistic) for _, t in tests)


def chi2_ball_occupancy(ref_b, our_b, W, H, cells=6):
    """Chi-square of the reference's ball positions against the cell probabilities estimated from our (much larger)
    sample; the goal boxes (240 x 240 in two corners) carve a hole both must show."""
    def hist(b):
        h, _, _ = np.histogram2d(b[:, :, 0].ravel(), b[:, :, 1].ravel(), bins=cells, range=[[40, W - 40 + 1e-9], [40, H - 40 + 1e-9]])
        return h.ravel()
    ho, hr = hist(our_b), hist(ref_b)
    p = ho / ho.sum()
    keep = p > 0
    assert hr[~keep].sum() == 0  # the reference never puts a ball where we never do
    chi = stats.chisquare(hr[keep], p[keep] * hr.sum())
    assert chi.pvalue > 1e-3, (float(chi.statistic), float(chi.pvalue))
    return float(chi.pvalue)
