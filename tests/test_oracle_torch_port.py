"""The PyTorch-CPU port used as bench.py's cpu_baseline must itself agree with the reference fixtures."""
import numpy as np
import pytest

import cases
from conftest import GRAD_KEYS, RENDER_KEYS, load_golden, relerr
from oracle import vmap_oracle_torch as vt


@pytest.mark.parametrize("name", ["tiny", "drop_depth", "drop_colour", "h64", "bg_h128_s14"])
def test_torch_port_matches_reference(name):
    c = cases.build_case(name)
    g = load_golden(name)
    tr = vt.CpuTrainer(c["fc"], c["B"], c["scale"])
    loss, rend, grads = tr.step(c["batch"], update=False)
    loss = float(loss)
    assert abs(loss - float(g["loss"])) <= 2e-5 * abs(float(g["loss"]))
    for k in RENDER_KEYS:
        assert relerr(rend[k].detach().numpy(), g[k]) < 2e-5, k
    for k, gr in zip(GRAD_KEYS, grads):
        assert relerr(gr.numpy(), g[k]) < 1e-4, k
