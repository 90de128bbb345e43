"""particle_fm_amd/ode.py: the two embedded 5(4) pairs (Dormand-Prince, Tsitouras) against their order conditions, and the controller
on an ODE with a known solution (plain torch on the CPU: the integrator itself has no kernel, only its right-hand side does)."""
import math

import numpy as np
import pytest
import torch

from particle_fm_amd.ode import DOPRI5, TSIT5, dopri5, tsit5


def _arrays(tab):
    c, a, b, e = tab
    n = len(c)
    A = np.zeros((n, n))
    for i, row in enumerate(a):
        A[i, : len(row)] = row
    return np.array(c), A, np.array(b), np.array(e)


@pytest.mark.parametrize("tab", [DOPRI5, TSIT5], ids=["dopri5", "tsit5"])
def test_order_conditions(tab):
    c, A, b, e = _arrays(tab)
    np.testing.assert_allclose(A.sum(1), c, atol=1e-13)  # row sums
    np.testing.assert_allclose(A[6], b, atol=0)          # FSAL: the 7th stage's input is the new state
    Ac, Ac2, AAc = A @ c, A @ c**2, A @ A @ c
    fifth = {  # all 17 conditions of order 5
        "1": (b.sum(), 1), "c": (b @ c, 1 / 2), "c2": (b @ c**2, 1 / 3), "Ac": (b @ Ac, 1 / 6), "c3": (b @ c**3, 1 / 4),
        "cAc": (b @ (c * Ac), 1 / 8), "Ac2": (b @ Ac2, 1 / 12), "AAc": (b @ AAc, 1 / 24), "c4": (b @ c**4, 1 / 5),
        "c2Ac": (b @ (c**2 * Ac), 1 / 10), "cAc2": (b @ (c * Ac2), 1 / 15), "cAAc": (b @ (c * AAc), 1 / 30),
        "(Ac)2": (b @ (Ac * Ac), 1 / 20), "Ac3": (b @ (A @ c**3), 1 / 20), "A(cAc)": (b @ (A @ (c * Ac)), 1 / 40),
        "AAc2": (b @ (A @ Ac2), 1 / 60), "AAAc": (b @ (A @ AAc), 1 / 120),
    }
    for name, (got, want) in fifth.items():
        assert abs(got - want) < 5e-7, (name, got, want)  # (Tsitouras publishes 16-17 digits of coefficients that satisfy them to ~1e-8)
    bh = b - e  # the embedded 4th-order weights
    for got, want in ((bh.sum(), 1), (bh @ c, 1 / 2), (bh @ c**2, 1 / 3), (bh @ Ac, 1 / 6), (bh @ c**3, 1 / 4), (bh @ (c * Ac), 1 / 8),
                      (bh @ Ac2, 1 / 12), (bh @ AAc, 1 / 24)):
        assert abs(got - want) < 5e-7
    assert abs(bh @ c**4 - 1 / 5) > 1e-4  # ... and not of order 5: the difference is an error estimate


@pytest.mark.parametrize("solve", [dopri5, tsit5], ids=["dopri5", "tsit5"])
def test_rotation_backwards_in_time_with_checkpoints(solve):
    w = 3.0
    f = lambda t, x: torch.stack([-w * x[..., 1], w * x[..., 0]], dim=-1) * (1 + t)  # angle(t) = w (t + t^2 / 2)
    x0 = torch.tensor([[1.0, 0.0], [0.0, 2.0]])
    ang = lambda t: w * (t + t * t / 2)
    d = ang(0.0) - ang(1.0)
    rot = torch.tensor([[math.cos(d), -math.sin(d)], [math.sin(d), math.cos(d)]])
    want = x0 @ rot.T
    calls = []
    g = lambda t, x: (calls.append(float(t)), f(t, x))[1]
    got = solve(g, x0, 1.0, 0.0, atol=1e-6, rtol=1e-5, checkpoints=[0.75, 0.5, 0.25])
    torch.testing.assert_close(got, want, atol=2e-4, rtol=2e-4)
    assert min(calls) >= -1e-6 and max(calls) <= 1.0 + 1e-6  # never evaluated outside [t1, t0]
    coarse = solve(f, x0, 1.0, 0.0, atol=1e-3, rtol=1e-3)
    assert (coarse - want).abs().max() < 2e-2
