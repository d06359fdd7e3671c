"""CPU: the emulation behind DESIGN section 2's precision ladder (tools/precision_ladder_sim.py) on a small case -- the facts the
table rests on hold in miniature: the all-pair / split-weight / pair-weight-gradient rung IS the float64 evaluation, single-f16
storage costs the forward pass 1e-4..1e-3, and dropping the lo halves from the weight gradients alone costs ~2^-12 per tensor."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_ladder_emulation_orders_the_rungs():
    import precision_ladder_sim as P
    torch.manual_seed(0)
    rows = P.run(seed=11, n_blocks=2, size=12, only=["fast", "exact16", "stream+tail pair, dense f16, W split"])
    fast = rows["fast (all f16)"]
    exact3 = rows["exact16x3 (all pair, W split, wgrad pairs)"]
    hi_only = rows["exact16 (all pair, W split, wgrad hi-only)"]
    mixed = rows["stream+tail pair, dense f16, W split"]
    assert exact3["fwd_max_abs"] == 0.0 and exact3["grad_worst"] == 0.0            # nothing rounded: the reference evaluation itself
    assert 1e-5 < fast["fwd_max_abs"] < 2e-2 and fast["grad_worst"] > 1e-3           # every operand one f16
    assert hi_only["fwd_max_abs"] == 0.0 and 1e-5 < hi_only["grad_worst"] < 1e-3     # only the weight gradients' operands rounded: ~2^-12
    assert mixed["fwd_max_abs"] < 0.1 * fast["fwd_max_abs"]                          # f16 growth planes alone barely move the forward pass
