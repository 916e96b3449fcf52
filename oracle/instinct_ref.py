"""CPU oracle for the scripted "instinct" opponent (reference instinct/agent.py:10-62).  TEST INFRASTRUCTURE ONLY.

Pinned on tests/golden/g8_instinct_pairs.npz (outputs of the reference agent itself, tests/test_oracle_golden.py).
Arithmetic is binary64 on the float32 observation values -- what the reference computes under its pinned numpy 1.23.1
(a float32 scalar is promoted to float64 by its first operation with a Python number)."""
import math

import numpy as np

DIAG = math.sqrt(math.pow(1200, 2) + math.pow(800, 2))
SHOT_DIST = 500
MAX_TURN = 35


def choose_target(obs, n):
    """agent.py:12-39 -> (dist, angle) of the chosen target.  Score = dist * |angle|; the base wins ties, then the
    first enemy with the minimum; a dead enemy scores 1e6."""
    o = [float(v) for v in obs]
    info = [((o[0] + 1) / 2 * DIAG, o[1] * 360)]
    scores = [info[0][0] * abs(info[0][1])]
    for j in range(n):
        d, a = (o[3 + 3 * j] + 1) / 2 * DIAG, o[4 + 3 * j] * 360
        info.append((d, a))
        scores.append(d * abs(a) if o[2 + 3 * j] == 1 else 1000000)
    m = min(scores)
    return info[0] if m == scores[0] else info[scores.index(m)]


def discrete_action(obs, n):
    """agent.py:56-62."""
    d, a = choose_target(obs, n)
    if d < SHOT_DIST / 2 and abs(a) < 20:
        return 1
    return 3 if a > 0 else 2


def continuous_action(obs, n, rand, noise):
    """agent.py:41-54; `rand` = the np.random.rand() value (used only when in range), `noise` = the 3 uniform(-0.15, 0.15)."""
    d, a = choose_target(obs, n)
    act = [0, 0, 0]
    if d < SHOT_DIST / 3 * 2 and abs(a) < 20:
        act[2] = 1 if rand < 0.6 else -1
    act[0] = d / DIAG * 2 - 1
    act[1] = max(-a / MAX_TURN, -1) if a > 0 else min(-a / MAX_TURN, 1)
    return np.clip(np.asarray(act, np.float64) + np.asarray(noise, np.float64), -1, 1)
