"""Checks that pin the PRODUCTION random draws of reset / auto-reset / the shot (in-kernel Philox on the GPU, the same construction in
the C oracle) to what the REFERENCE draws: fixture g7_spawn_stats.npz (20 000 resets of the unmodified reference at 2v2: per-column
minimum, maximum and mean, and the support of the red and blue headings; sprites.py:82-91,246-252) and sprites.py:314 for the jitter
(`angle + (random.random() * 8 - 4)`: uniform on [-4, 4)).  Used by tests/test_hip_rng_pin.py (GPU) and tests/test_oracle_c_golden.py (C)."""
import os

import numpy as np
from scipy import stats

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# g7's columns: base_red x,y, base_blue x,y, plane0..3 x,y,dir (2v2)
P_FLOOR = 1e-6            # a fixed-seed chi-square p-value below this is a distribution error, not bad luck


def spawn_table(st):
    """[E, 16] in g7's column order from an exported state (numpy arrays px, py, pdir [E, 4], base_xy [E, 4])."""
    cols = [st["base_xy"][:, 0], st["base_xy"][:, 1], st["base_xy"][:, 2], st["base_xy"][:, 3]]
    for i in range(4):
        cols += [st["px"][:, i], st["py"][:, i], np.rint(st["pdir"][:, i]).astype(np.int64)]
        assert np.array_equal(st["pdir"][:, i], np.rint(st["pdir"][:, i])), "spawn headings are whole degrees"
    return np.stack([np.asarray(c, np.int64) for c in cols], 1)


def check_spawn_table(d, what):
    """d [E >= 1e5, 16]: per-column extremes EQUAL the reference's, heading support equal, every column uniform over its support
    (chi-square), column means within 1 % of the reference's sample means, columns pairwise uncorrelated."""
    z = np.load(os.path.join(GOLDEN, "g7_spawn_stats.npz"))
    E = d.shape[0]
    assert E >= 100_000, "too few games for the extremes to be certain"
    assert np.array_equal(d.min(0), z["lo"]), (what, d.min(0).tolist(), z["lo"].tolist())
    assert np.array_equal(d.max(0), z["hi"]), (what, d.max(0).tolist(), z["hi"].tolist())
    red_support, blue_support = z["red_dir_hist"] > 0, z["blue_dir_hist"] > 0
    assert red_support.sum() == 181 and blue_support.sum() == 181            # {270..359} U {0..90}; 90..270
    for c in range(16):
        col = d[:, c]
        is_dir = c >= 6 and (c - 6) % 3 == 0
        if is_dir:
            support = red_support if c in (6, 9) else blue_support
            hist = np.bincount(col, minlength=361)
            assert np.array_equal(hist > 0, support), (what, c, "heading support differs from the reference's")
            obs = hist[support]
        else:
            lo, hi = int(z["lo"][c]), int(z["hi"][c])
            obs = np.bincount(col - lo, minlength=hi - lo + 1)
            assert obs.size == hi - lo + 1
        p = stats.chisquare(obs).pvalue
        assert p > P_FLOOR, (what, c, "not uniform over the reference's support", p)
        # the reference's own 20 000-sample mean carries ~0.4 % (positions) ... 0.6 % (headings) of sampling error
        assert abs(col.mean() - z["mean"][c]) <= 0.01 * z["mean"][c] + 3.0 * col.std() / np.sqrt(int(z["n"])), (what, c, col.mean(), z["mean"][c])
    cc = np.corrcoef(d.T.astype(np.float64))
    off = np.abs(cc - np.eye(16)).max()
    assert off < 6.0 / np.sqrt(E), (what, "columns are correlated", off)


def check_jitter(bl_dir, shooter_dir, what):
    """bl_dir, shooter_dir [M >= 1e5]: heading of a fresh bullet and its shooter's pre-move heading.  The difference is the reference's
    `random.random() * 8 - 4` (sprites.py:314): in [-4, 4), uniform, 53-bit resolution (u = (d + 4) / 8 is a multiple of 2^-53)."""
    M = bl_dir.shape[0]
    assert M >= 100_000
    # the reference adds in binary64: (u * 8 - 4) is exact for u a multiple of 2^-53, the sum with the heading rounds once.  Recover
    # the jitter only where the heading is small enough for the sum to have been exact to 2^-44 (|heading| < 512)
    j = bl_dir - shooter_dir
    assert j.min() >= -4.0 and j.max() < 4.0 + 1e-12, (what, j.min(), j.max())
    u = (j + 4.0) / 8.0
    obs = np.bincount(np.minimum((u * 256).astype(np.int64), 255), minlength=256)
    p = stats.chisquare(obs).pvalue
    assert p > P_FLOOR, (what, "jitter is not uniform on [-4, 4)", p)
    assert abs(j.mean()) < 6.0 * (8.0 / np.sqrt(12.0)) / np.sqrt(M), (what, j.mean())
    assert abs(j.std() - 8.0 / np.sqrt(12.0)) < 0.01
    # more than float32 resolution: a 53-bit uniform has ~all distinct values at this sample size
    assert np.unique(j).size > 0.999 * M, (what, "jitter values collide: fewer random bits than random.random()")
    # independent of the shooter's heading and between neighbouring shooters
    assert abs(np.corrcoef(j, shooter_dir)[0, 1]) < 6.0 / np.sqrt(M)
    assert abs(np.corrcoef(j[:-1], j[1:])[0, 1]) < 6.0 / np.sqrt(M)
