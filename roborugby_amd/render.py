"""CPU debug picture of one arena (PIL).  Not the reference's pygame renderer (UI is out of scope): just enough to look
at a state -- goals, robots as rotated 20x40 rectangles with their front edge marked, balls."""
import math

import numpy as np

from .config import GOAL_HEIGHT, GOAL_WIDTH

COLOR_BACKGROUND = (255, 255, 255)   # RR_Constants.py:69-79
COLOR_BALL_POS = (80, 220, 100)
COLOR_BALL_NEG = (60, 16, 83)
COLOR_GOAL_HAPPY = (43, 146, 228)
COLOR_GOAL_GRUMPY = (242, 53, 87)
COLOR_DASHBOARD_FILL = (200, 200, 200)
DASHBOARD_WIDTH = 300


def robot_corners(cx, cy, rot):
    """TL, TR, BL, BR of a 20 (x) by 40 (y) rect rotated by rot degrees, y down (MyUtils.py:277-322)."""
    th = math.radians(360 - rot)
    c, s = math.cos(th), math.sin(th)
    out = []
    for (x, y) in ((-10, -20), (10, -20), (-10, 20), (10, 20)):
        out.append((cx + x * c - y * s, cy + x * s + y * c))
    return out


def draw_arena(preset, robots, balls):
    from PIL import Image, ImageDraw
    W, H = int(preset.arena_w), int(preset.arena_h)
    img = Image.new("RGB", (W + DASHBOARD_WIDTH, H), COLOR_BACKGROUND)
    d = ImageDraw.Draw(img)
    d.rectangle([W, 0, W + DASHBOARD_WIDTH - 1, H - 1], fill=COLOR_DASHBOARD_FILL)
    d.polygon([(W, H), (W, H - GOAL_HEIGHT), (W - GOAL_WIDTH, H)], fill=COLOR_GOAL_HAPPY)     # RR_Goal.py:14-28
    d.polygon([(GOAL_WIDTH, 0), (0, GOAL_HEIGHT), (0, 0)], fill=COLOR_GOAL_GRUMPY)
    for i, r in enumerate(robots):
        tl, tr, bl, br = robot_corners(r[0], r[1], r[6])
        col = (40, 90, 200) if i < preset.nr_happy else (200, 60, 60)
        d.polygon([tl, tr, br, bl], fill=col, outline=(0, 0, 0))
        d.line([tr, br], fill=(255, 255, 0), width=2)  # "front" of the robot = RIGHT side (RR_Observers.py:322-324)
    for i, b in enumerate(balls):
        col = COLOR_BALL_POS if i < preset.nb_pos else COLOR_BALL_NEG
        d.ellipse([b[0] - 7, b[1] - 7, b[0] + 7, b[1] + 7], fill=col, outline=(0, 0, 0))
    return np.asarray(img)
