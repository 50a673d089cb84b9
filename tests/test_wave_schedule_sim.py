"""CPU check of the wavefront kernel's schedule (tools/sim_wave_sssp.py): windows of 64 advance nodes + 500 bp look-ahead,
A/B phases, step-back when a close node near the window start improves.  Run over the oracle's graph with python integers, it
must reproduce the oracle's distance and path — the argument for dropping the verification pass."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("seed,length,adv", [(11, 12000, 64), (12, 12000, 32), (13, 20000, 64), (14, 6000, 96)])
def test_windowed_schedule_reaches_the_fixed_point(seed, length, adv):
    import sim_wave_sssp as sim

    out = sim.run(seed, L=length, ADV=adv)
    assert out["viol"] == 0          # every edge constraint holds: the distances are the fixed point
    assert out["dist_ok"] and out["path_ok"]
    assert out["rollbacks"] <= 4
