"""CPU: the oracle's LM loop (blur_aware_direct_tracker.cpp:544-924 restated) on a synthetic blurred pair:
it must run coarse-to-fine, accept steps, reduce the cost and move the trajectory toward the ground truth."""
import numpy as np
import pytest

import tracking


@pytest.mark.parametrize("k", [4, 2])
def test_oracle_tracker_converges(orc, mbavo, k):
    sc = tracking.make_tracking_scene(orc, H=120, W=160, levels=3, S=8, k=k, seed=1)
    r = tracking.run_oracle_tracker(orc, sc)
    tr = r["trace"]
    assert [t[0] for t in tr if t[1] == 0] == [2, 1, 0]          # levels L-1 .. 0, one initial evaluation each
    assert any(t[2] == 1 for t in tr)                            # at least one accepted step
    first, last = tr[0][5], r["cost"]
    assert last < first
    assert list(r["start"]) == [0]
    assert tracking.flow_error(orc, sc, r["kt"], r["kR"]) < 0.5 * tracking.flow_error(orc, sc, sc["kt0"], sc["kR0"])
    # every accepted step obeys the acceptance rule quality > 0.5 and candidate < eval (A17)
    for t in tr:
        if t[2] == 1:
            assert t[8] > 0.5
