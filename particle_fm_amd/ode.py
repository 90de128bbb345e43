"""Adaptive Dormand-Prince 5(4) integration of dx/dt = f(t, x) where f is one HIP evaluation of the vector field: what the reference's
DEFAULT solver ``ode_solver="dopri5_zuko"`` (flow_matching_module.py:260-261: ``zuko.utils.odeint(f, z, 1.0, 0.0)``, adaptive dopri5 at
atol 1e-6 / rtol 1e-5) and torchdyn's adaptive "dopri5" (:267-277: atol = rtol = 1e-4 over linspace(1, 0, ode_steps)) ask for.

PARITY UNPINNED: neither zuko nor torchdyn is in this image and the reference holds no fixture of these solvers, so the step
controller below is the textbook one (Hairer, Noersett, Wanner II.4: error norm = rms of err / (atol + rtol max(|x|, |x_new|)),
factor = 0.9 err^(-1/5) clamped to [0.2, 10], their starting-step rule), not a restatement of either library's.  Two adaptive
integrators at the same tolerances agree to about those tolerances, not bit for bit: tests/test_hip_adaptive.py checks the result
against a far finer fixed-step rk4 solution of the oracle.

The stages, the error estimate and the state update are element-wise device ops between the field evaluations (like the diffusion
samplers of CNF._decode_diffusion_rows); ONE scalar per step -- the error norm -- travels to the host for the accept / reject decision.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch

# Dormand-Prince 5(4), FSAL (k7 of an accepted step is k1 of the next)
_C = (0.0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0)
_A = (
    (),
    (1 / 5,),
    (3 / 40, 9 / 40),
    (44 / 45, -56 / 15, 32 / 9),
    (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
    (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656),
    (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84),
)
_B5 = _A[6] + (0.0,)
_B4 = (5179 / 57600, 0.0, 7571 / 16695, 393 / 640, -92097 / 339200, 187 / 2100, 1 / 40)
_E = tuple(b5 - b4 for b5, b4 in zip(_B5, _B4))
DOPRI5 = (_C, _A, _B5, _E)

# Tsitouras 5(4) (Ch. Tsitouras, Comput. Math. Appl. 62 (2011) 770-775), FSAL: torchdyn's "tsit5" (flow_matching_module.py:288-292).  The
# published coefficients, held to their order conditions by tests/test_ode_cpu.py.
_TC = (0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0)
_TA = (
    (),
    (0.161,),
    (-0.008480655492356989, 0.335480655492357),
    (2.8971530571054935, -6.359448489975075, 4.3622954328695815),
    (5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525),
    (5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383),
    (0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774),
)
_TE = (-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995, -0.1447110071732629, 0.5823571654525552,
       -0.45808210592918697, 1 / 66)  # b - b_hat
TSIT5 = (_TC, _TA, _TA[6] + (0.0,), _TE)


def _rms(v: torch.Tensor) -> torch.Tensor:
    return v.square().mean().sqrt()


def dopri5(f: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], x0: torch.Tensor, t0: float, t1: float, atol: float = 1e-6,
           rtol: float = 1e-5, checkpoints: Optional[Sequence[float]] = None, max_steps: int = 2000, tableau=DOPRI5) -> torch.Tensor:
    """x(t1) from x(t0) = x0.  f(t, x): t a 0-dim float32 device tensor.  checkpoints: times the integrator lands on exactly on its way
    (torchdyn's adaptive odeint does with its t_span); the state at t1 is returned either way.  tableau: a 7-stage FSAL 5(4) pair."""
    _C, _A, _B5, _E = tableau
    dev = x0.device
    sign = 1.0 if t1 >= t0 else -1.0
    stops = [float(c) for c in (checkpoints or []) if (c - t0) * sign > 0 and (t1 - c) * sign > 0] + [float(t1)]
    tt = lambda v: torch.tensor(v, dtype=torch.float32, device=dev)
    x, t = x0.to(torch.float32), float(t0)
    k1 = f(tt(t), x)
    # starting step (HNW II.4): h0 = 0.01 |x| / |f|, one Euler probe, h1 from the second-derivative estimate
    sc = atol + rtol * x.abs()
    d0, d1 = float(_rms(x / sc)), float(_rms(k1 / sc))
    h0 = 1e-6 if d0 < 1e-5 or d1 < 1e-5 else 0.01 * d0 / d1
    d2 = float(_rms((f(tt(t + sign * h0), x + sign * h0 * k1) - k1) / sc)) / h0
    h1 = max(1e-6, 1e-3 * h0) if max(d1, d2) <= 1e-15 else (0.01 / max(d1, d2)) ** 0.2
    h = min(100 * h0, h1, abs(t1 - t0))
    steps = 0
    for stop in stops:
        while (stop - t) * sign > 1e-9:
            if steps >= max_steps:
                raise RuntimeError(f"dopri5: no convergence within {max_steps} steps (t = {t}, h = {h})")
            steps += 1
            hh = sign * min(h, abs(stop - t))
            ks = [k1]
            for s in range(1, 7):
                acc = _A[s][0] * ks[0]
                for j in range(1, s):
                    if _A[s][j]:
                        acc = acc + _A[s][j] * ks[j]
                ks.append(f(tt(t + _C[s] * hh), x + hh * acc))
            x_new = x + hh * sum(b * k for b, k in zip(_B5, ks) if b)  # (= the input of the 7th stage: FSAL)
            err = hh * sum(e * k for e, k in zip(_E, ks) if e)
            ratio = float(_rms(err / (atol + rtol * torch.maximum(x.abs(), x_new.abs()))))  # the step's one host round trip
            if ratio <= 1.0:
                t, x, k1 = t + hh, x_new, ks[6]
            h = abs(hh) * min(10.0, max(0.2, 0.9 * (ratio if ratio > 1e-10 else 1e-10) ** -0.2))
    return x


def tsit5(f, x0, t0, t1, atol: float = 1e-3, rtol: float = 1e-3, checkpoints=None, max_steps: int = 2000) -> torch.Tensor:
    """The same controller around Tsitouras' 5(4) pair: ``ode_solver="tsit5"`` (flow_matching_module.py:288-292: torchdyn's NeuralODE with
    its own default tolerances -- [recalled] atol = rtol = 1e-3 in torchdyn 1.0; unpinned like everything at that boundary)."""
    return dopri5(f, x0, t0, t1, atol=atol, rtol=rtol, checkpoints=checkpoints, max_steps=max_steps, tableau=TSIT5)
