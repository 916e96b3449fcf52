"""Single-game state export for eyeballing (the reference renders with pygame,
envs/battle_env.py:498-560, envs/sprites.py:155-192,265-278,353-360 -- out of scope here).

`frame(env, index)` copies ONE game's unpacked state to the host (`bsx_export_state`) and rasterises it with numpy into
an RGB image of the 1200 x 800 field: bases as 62 x 62 squares, planes as their un-rotated 50 x 48 hit boxes with a
heading tick, bullets as 6 x 3 marks; dead planes hollow.  `save_ppm` writes it without any imaging dependency.
Purely diagnostic: nothing here is on the step() path."""
import math

import numpy as np

RED, BLUE, BLACK, WHITE = (138, 24, 26), (0, 93, 135), (0, 0, 0), (255, 255, 255)   # envs/sprites.py:5-8
W, H = 1200, 800


def _rect(img, cx, cy, w, h, colour, filled=True):
    x0, x1 = max(0, cx - (w >> 1)), min(W, cx - (w >> 1) + w)
    y0, y1 = max(0, cy - (h >> 1)), min(H, cy - (h >> 1) + h)
    if x0 >= x1 or y0 >= y1:
        return
    if filled:
        img[y0:y1, x0:x1] = colour
    else:
        img[y0:y1, [x0, x1 - 1]] = colour
        img[[y0, y1 - 1], x0:x1] = colour


def _line(img, x0, y0, x1, y1, colour):
    n = int(max(abs(x1 - x0), abs(y1 - y0))) + 1
    xs = np.clip(np.linspace(x0, x1, n).round().astype(int), 0, W - 1)
    ys = np.clip(np.linspace(y0, y1, n).round().astype(int), 0, H - 1)
    img[ys, xs] = colour


def frame_from_state(st, n_agents):
    """st: dict of numpy arrays for ONE game in the bsx_export_state schema (px, py, pdir, php [A]; base_xy [4];
    bhp [2]; bl_live, bl_x, bl_y [A, 12]).  Returns uint8 [800, 1200, 3]."""
    img = np.full((H, W, 3), WHITE, np.uint8)
    n = n_agents
    bx = st["base_xy"]
    for t, col in ((0, RED), (1, BLUE)):
        _rect(img, int(bx[2 * t]), int(bx[2 * t + 1]), 62, 62, col, filled=int(st["bhp"][t]) > 0)
    for a in range(2 * n):
        col = RED if a < n else BLUE
        x, y, d = int(st["px"][a]), int(st["py"][a]), float(st["pdir"][a])
        _rect(img, x, y, 50, 48, col, filled=int(st["php"][a]) > 0)
        ang = -math.radians(d)                          # heading convention of calc_new_xy (sprites.py:35-42)
        _line(img, x, y, x + 30 * math.cos(ang), y + 30 * math.sin(ang), BLACK)
        for k in range(st["bl_live"].shape[1]):
            if st["bl_live"][a, k]:
                _rect(img, int(st["bl_x"][a, k]), int(st["bl_y"][a, k]), 6, 3, col)
    return img


def frame(env, index=0):
    """RGB image of game `index` of a parallel_env (one device -> host copy of the exported state)."""
    st = {k: v[index].cpu().numpy() for k, v in env.export_state(("px", "py", "pdir", "php", "base_xy", "bhp", "bl_live", "bl_x", "bl_y")).items()}
    return frame_from_state(st, env.n_agents)


def save_ppm(path, img):
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img, np.uint8).tobytes())
