"""The oracle-pinning tests, selected by `-m gpu` too: the parity chain "HIP path == C oracle == reference fixtures" has its SECOND
link checked on the very box whose gcc / libm build and run the oracle that the full-size, fuzz and soak GPU tests lean on
(envs/battle_env.py:38-58,281-381 as recorded in tests/golden/g1 ... g12 by the unmodified reference).

Nothing here touches the card; the marker is what makes the GPU run execute them (the CPU run has them under their own names in
test_oracle_c_golden.py / test_oracle_golden.py / test_oracle_properties.py).  The first test runs first in this file and says WHICH
oracle binary the rest of the GPU suite will load."""
import ctypes.util
import os
import platform
import subprocess

import pytest

pytestmark = pytest.mark.gpu

from test_oracle_c_golden import (                                                   # noqa: E402,F401
    test_c_oracle_equals_python_oracle_on_random_play,
    test_c_oracle_production_draws_have_the_reference_distribution,
    test_c_oracle_reproduces_reference_trace,
)
from test_oracle_golden import (                                                     # noqa: E402,F401
    test_instinct_oracle_reproduces_reference_agent,
    test_oracle_reproduces_reference_trace,
    test_oracle_same_seed_same_game_as_reference,
    test_rel_angle_and_dist_table,
    test_spawn_ranges_match_reference_draws,
    test_tie_tick_follows_float_accumulation,
)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_oracle_binary_is_the_one_this_box_runs(record_property):
    """The C oracle the GPU tests compare against: the file conftest's `make` left in oracle/, loaded here, resolved against THIS
    host's libm (it is linked dynamically: whatever built the file, the cos / sin / atan2 / sqrt it calls are this box's)."""
    from oracle import cref
    so = os.path.join(ROOT, "oracle", "libbattlespace_ref.so")
    assert os.path.exists(so)
    assert os.path.samefile(cref.load()._name, so)           # the library every CRefBatch of this process calls into
    needed = subprocess.run(["ldd", so], capture_output=True, text=True).stdout
    assert "libm.so" in needed, needed                       # libm is a run-time dependency, not baked in
    record_property("c_oracle", f"{so} on {platform.node()} ({platform.platform()}), libm = {ctypes.util.find_library('m')}")
