#!/usr/bin/env python3
"""GPU box: the reference's evaluation workload (evaluate.py:32-76) on device, with and without the script's stale first observation
(bench.evaluation_line) -> one JSON line: both tallies beside the unmodified evaluate.main()'s (tests/golden/g12_evaluation.npz)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench
E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
r = bench.evaluation_line(torch.device("cuda", 0), E, "f32")
keep = ("games", "ties", "red_wins", "blue_wins", "win_rate_red", "sigmas_from_reference_tally", "win_rate_red_stale_first_obs",
        "with_the_scripts_stale_first_observation", "reference_tally_evaluate_py", "us_per_tick")
print(json.dumps({k: r.get(k) for k in keep}))
