"""A short fixed-seed pass of the randomised differential run (tests/fuzz_util.py; tools/fuzz_parity.py is its long form): random team sizes 1 ... 16, ragged batch sizes,
every action encoding, reward constants, env_offset, auto-reset or masked resets, host-drawn jitter or Philox, narrow / wide offset
kernels, and every launch form (per-step calls, K-tick launches, captured graphs, graphs of chains over game ranges) against the
C oracle: rewards and flags equal on every call, observations within 1e-5, the complete game state bit-identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_random_configurations_against_the_c_oracle(seed):
    import fuzz_util as fz
    rng = np.random.default_rng(seed)
    forms, vals, exact = set(), 0, 0
    for _ in range(14):
        case = fz.draw_case(rng, max_envs=1500)
        bad, st = fz.run_case(case)
        assert bad is None, (bad, case)
        forms.add(case["form"]); vals += st["vals"]; exact += st["exact"]
    assert len(forms) >= 3 and exact >= vals * (1 - 1e-6)


def test_random_single_games_behind_the_reference_surface():
    """The drop-in surface (one game, dict actions, reference return types, stdlib `random` in the reference's draw order) against the
    Python oracle on the same random stream: 1v1 ... 6v6, discrete ints / score vectors / continuous float32 and float64, integer and
    float reward constants, actions for all or only the live agents, `step({})`, resets after every finished game."""
    import fuzz_util as fz
    rng = np.random.default_rng(21)
    vals = exact = 0
    for _ in range(16):
        case = fz.draw_dropin_case(rng)
        bad, st = fz.run_dropin_case(case)
        assert bad is None, (bad, case)
        vals += st["vals"]; exact += st["exact"]
    assert exact >= vals * (1 - 1e-6)


@pytest.mark.parametrize("seed", [31, 32])
def test_random_policy_rollouts_replayed_by_the_c_oracle(seed):
    """PolicyRollout in random forms (one launch / two-kernel graph / chains; 1v1 ... 8v8; discrete, continuous; Gaussian, OU,
    categorical + value head; scripted opponents; f32 and split-bf16 actors): the C oracle replays the games from the recorded score
    rows -- rewards, flags and the final state identical, observations within 1e-5."""
    import fuzz_util as fz
    rng = np.random.default_rng(seed)
    forms = set()
    for _ in range(12):
        case = fz.draw_rollout_case(rng)
        bad, st = fz.run_rollout_case(case)
        assert bad is None, (bad, case)
        forms.add(case["form"])
    assert len(forms) >= 2
