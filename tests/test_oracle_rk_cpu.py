"""The restated fixed-step Runge-Kutta driver (oracle/fm_ref.py) and the host-side time grid / tableau of the product."""
import math

import pytest
import torch

from oracle.fm_ref import RK_TABLEAUS, midpoint_trajectory_end, rk_trajectory_end
from particle_fm_amd import hip_ops


def _f(t, x):
    return -x * (1.0 + t)  # dx/dt = -(1 + t) x  ->  x(t) = x(1) exp((1 + t)^2 / -2 + 2)


def test_midpoint_tableau_is_the_midpoint_driver_bit_for_bit():
    x = torch.randn(7, 5)
    t_span = torch.linspace(1.0, 0.0, 23)
    assert torch.equal(rk_trajectory_end(_f, x, t_span, "midpoint"), midpoint_trajectory_end(_f, x, t_span))


@pytest.mark.parametrize("solver,order", [("euler", 1), ("midpoint", 2), ("rk4", 4)])
def test_convergence_order(solver, order):
    x = torch.ones(3, dtype=torch.float64)
    exact = math.exp(((1 + 0.0) ** 2 - (1 + 1.0) ** 2) / -2)  # integrate from t=1 to t=0
    errs = []
    for n in (9, 17, 33):
        xe = rk_trajectory_end(_f, x, torch.linspace(1.0, 0.0, n, dtype=torch.float64), solver)
        errs.append(abs(float(xe[0]) - exact))
    for a, b in zip(errs, errs[1:]):
        assert order - 0.4 < math.log2(a / b) < order + 0.6, (solver, errs)


def test_tableau_consistency_and_product_mirror():
    for name, (c, a, b) in RK_TABLEAUS.items():
        assert abs(sum(b) - 1.0) < 1e-12
        for s in range(1, len(c)):
            assert abs(sum(a[s]) - c[s]) < 1e-12  # row-sum condition
        assert hip_ops.RK_TABLEAUS[name] == (c, a, b)
        t = hip_ops.rk_tableau(name)
        assert t.stages == len(b) and [t.b[i] for i in range(len(b))] == [torch.tensor(v, dtype=torch.float32).item() for v in b]
    with pytest.raises(NotImplementedError):
        hip_ops.rk_tableau("dopri5")


@pytest.mark.parametrize("solver", ["euler", "midpoint", "rk4"])
def test_grid_is_what_the_driver_visits(solver):
    seen = []

    def f(t, x):
        seen.append(t.clone())
        return -x

    rk_trajectory_end(f, torch.ones(2), torch.linspace(1.0, 0.0, 12), solver)
    ts, dts = hip_ops.rk_grid(12, solver)
    assert torch.equal(ts, torch.stack(seen)) and dts.numel() == 11
    if solver == "midpoint":
        ts2, dts2 = hip_ops.midpoint_grid(12)
        assert torch.equal(ts, ts2) and torch.equal(dts, dts2)
    # encode direction
    ts, dts = hip_ops.rk_grid(100, "rk4", 0.0, 1.0)
    assert float(ts[0]) == 0.0 and float(dts.sum()) == pytest.approx(1.0, abs=1e-6) and ts.numel() == 99 * 4
