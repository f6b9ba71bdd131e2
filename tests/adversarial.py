"""Synthetic contact-dense states (shared by the CPU and GPU differential tests): entities are deliberately placed
touching / overlapping each other and the walls so that every response path, the resolve-loop exhaustion, the undo
fallback and the fault branches fire far more often than in rollouts from reset."""
import numpy as np

import oracle_lib as ol


def make_states(preset, n, seed):
    cfg = ol.PRESETS[preset]
    nr, nb = cfg["nr_h"] + cfg["nr_g"], cfg["nb_p"] + cfg["nb_n"]
    W, H = cfg["W"], cfg["H"]
    rng = np.random.RandomState(seed)
    robots = np.zeros((n, nr, 3))
    balls = np.zeros((n, nb, 4))
    for a in range(n):
        mode = a % 6
        for r in range(nr):
            if mode == 3 and r > 0:  # robots crowding each other
                robots[a, r, :2] = robots[a, r - 1, :2] + rng.uniform(-42, 42, 2)
            elif mode == 4:           # robots hugging a wall
                robots[a, r] = [rng.choice([rng.uniform(21, 30), W - rng.uniform(21, 30)]), rng.uniform(40, H - 40), 0]
            else:
                robots[a, r, :2] = [rng.uniform(60, W - 60), rng.uniform(60, H - 60)]
            robots[a, r, :2] = np.clip(robots[a, r, :2], 23, W - 23)
            robots[a, r, 2] = rng.choice([0, 90, 180, 270, rng.uniform(0, 360), float(rng.randint(0, 361))])
        for b in range(nb):
            if mode in (0, 3):      # ball against / inside reach of a robot
                r = rng.randint(nr)
                d, th = rng.uniform(4, 34), rng.uniform(0, 2 * np.pi)
                balls[a, b, :2] = robots[a, r, :2] + d * np.array([np.cos(th), np.sin(th)])
            elif mode == 1:         # ball near a wall or a corner
                balls[a, b, :2] = [rng.choice([rng.uniform(-2, 12), W - rng.uniform(-2, 12), rng.uniform(20, W - 20)]),
                                   rng.choice([rng.uniform(-2, 12), H - rng.uniform(-2, 12), rng.uniform(20, H - 20)])]
            elif mode == 2 and b > 0:  # balls touching balls
                d, th = rng.uniform(0.5, 16), rng.uniform(0, 2 * np.pi)
                balls[a, b, :2] = balls[a, b - 1, :2] + d * np.array([np.cos(th), np.sin(th)])
            elif mode == 5:         # ball squeezed between a wall-hugging robot and the wall
                r = rng.randint(nr)
                balls[a, b, :2] = [robots[a, r, 0] + rng.choice([-1, 1]) * rng.uniform(12, 22), robots[a, r, 1] + rng.uniform(-15, 15)]
            else:
                balls[a, b, :2] = [rng.uniform(10, W - 10), rng.uniform(10, H - 10)]
            balls[a, b, 2:] = rng.choice([0.0, 1.0]) * rng.uniform(-9, 9, 2)
    actions = rng.randint(0, 8, (n, nr)).astype(np.int32)
    return robots, balls, actions


def oracle_step(preset, robots_xyr, balls_xyv, actions):
    o = ol.OracleEnv(preset)
    o.set_clean_state(robots_xyr, balls_xyv)
    res = o.step(actions)
    return res, o.get_state()


def make_corner_states(preset, n, seed):
    """Balls around robot corners: at 6.4-7.7 px from a corner, in and next to the 7 x 7 square beyond BOTH side lines where only
    the corner's radius test can hit (rr_sim.hpp: ball_near_robot's corner-zone bound) -- touching, grazing and just clear."""
    cfg = ol.PRESETS[preset]
    nr, nb = cfg["nr_h"] + cfg["nr_g"], cfg["nb_p"] + cfg["nb_n"]
    W, H = cfg["W"], cfg["H"]
    rng = np.random.RandomState(seed)
    robots = np.zeros((n, nr, 3))
    balls = np.zeros((n, nb, 4))
    for a in range(n):
        for r in range(nr):
            robots[a, r] = [rng.uniform(80, W - 80), rng.uniform(80, H - 80),
                            rng.choice([0, 90, 180, 270, rng.uniform(0, 360), float(rng.randint(0, 361))])]
        for b in range(nb):
            r = rng.randint(nr)
            rot = np.radians(robots[a, r, 2])
            ux, uy, vx, vy = np.cos(rot), -np.sin(rot), np.sin(rot), np.cos(rot)  # the 20-px and the 40-px axis on the screen (y down)
            sx, sy = rng.choice([-1, 1]), rng.choice([-1, 1])
            d, th = rng.uniform(6.4, 7.7), rng.uniform(-0.25, np.pi / 2 + 0.25)   # mostly outward of both sides, some just across
            lx, ly = sx * (10 + d * np.cos(th)), sy * (20 + d * np.sin(th))
            balls[a, b, :2] = robots[a, r, :2] + lx * np.array([ux, uy]) + ly * np.array([vx, vy])
            balls[a, b, 2:] = rng.choice([0.0, 0.0, 1.0]) * rng.uniform(-1.5, 1.5, 2)
    actions = rng.randint(0, 8, (n, nr)).astype(np.int32)
    return robots, balls, actions
