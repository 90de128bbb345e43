"""ode_solver="dopri5_zuko" (the reference's DEFAULT, flow_matching_module.py:250, 260-261), "dopri5" (:267-277) and "tsit5" (:288-292):
adaptive Dormand-Prince / Tsitouras 5(4) around HIP evaluations of the field (particle_fm_amd/ode.py).  PARITY UNPINNED -- neither zuko nor torchdyn is in
the image and the reference holds no vector of these solvers -- so the bar is the ODE itself: the result agrees with a far finer
fixed-step rk4 solution of the ORACLE's field to the solver's tolerance, on every model family and under loss_type="diffusion"."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _fine(vf, z, cond, mask, steps=160):
    from oracle.fm_ref import sample_fixed_step
    return sample_fixed_step(vf, z, cond, mask, ode_steps=steps, solver="rk4")


# Fixtures with a SMOOTH time embedding (sincos: frequencies up to 32 pi; gaussian with the seeded projection): the "cosine" table goes up
# to e^31, a right-hand side no integrator converges on (the fixed-step tests compare like with like on one grid for that reason).
def _module(g):
    from particle_fm_amd.models import SetFlowMatchingLitModule
    m = SetFlowMatchingLitModule(optimizer=None, sigma=1e-4, **copy.deepcopy(g.hp))
    full = dict(g.state)
    full.update({"loss." + k: v for k, v in g.state.items()})
    m.load_state_dict(full, strict=False)
    return m.cuda()


def _epic():
    from oracle.fm_ref import EpicVectorField
    from tests.conftest import load_wide_golden
    g = load_wide_golden("sincos")  # hidden 128, 24 particles: the jet-resident kernels
    m = _module(g)
    assert not m.flows[0].net.is_wide(g.hp["num_particles"])
    return g, m, EpicVectorField(g.state, "flows.0.net", g.hp, freqs=g.freqs)


def _epicw():
    from oracle.fm_ref import EpicVectorField
    from tests.conftest import load_wide_golden
    g = load_wide_golden("gauss")  # hidden 300: the row-matrix kernels
    return g, _module(g), EpicVectorField(g.state, "flows.0.net", g.hp, freqs=g.freqs)


def _tf():
    from oracle.tf_ref import TransformerVectorField
    from tests.conftest import load_tf_golden
    g = load_tf_golden("gauss")
    return g, _module(g), TransformerVectorField(g.state, "flows.0.", g.hp, freqs=g.freqs)


def _mdma():
    from oracle.mdma_ref import MdmaVectorField, broadcast_field
    from tests.conftest import load_mdma_golden
    g = load_mdma_golden("tglob")  # sincos
    return g, _module(g), broadcast_field(MdmaVectorField(g.state, "flows.0.", g.hp, freqs=g.freqs))


@pytest.mark.parametrize("family", ["epic", "epicw", "tf", "mdma"])
def test_default_solver_reaches_the_fine_solution(family):
    g, m, vf = {"epic": _epic, "epicw": _epicw, "tf": _tf, "mdma": _mdma}[family]()
    tag = "midpoint_10/"
    z, mask, cond = (None if g.get(tag + k) is None else g.get(tag + k)[:2] for k in ("z", "mask", "cond"))  # (two jets: the CPU oracle sets the run time)
    dev = lambda a: None if a is None else a.cuda()
    want = _fine(vf, z, cond, mask)
    keep = mask.squeeze(-1) != 0
    assert (want - _fine(vf, z, cond, mask, steps=80))[keep].abs().max() < 5e-4  # the yardstick itself has converged
    out = m((z * mask).cuda(), cond=dev(cond), mask=mask.cuda(), reverse=True).cpu()  # forward(reverse=True)'s default: "dopri5_zuko"
    # local tolerances atol 1e-6 / rtol 1e-5 per step; the global error after the ~100 accepted steps is a few hundred times that
    torch.testing.assert_close(out[keep], want[keep], atol=2e-3, rtol=1e-2)
    out = m((z * mask).cuda(), cond=dev(cond), mask=mask.cuda(), reverse=True, ode_solver="dopri5", ode_steps=20).cpu()
    torch.testing.assert_close(out[keep], want[keep], atol=2e-2, rtol=5e-2)  # (atol = rtol = 1e-4 per step)
    out = m((z * mask).cuda(), cond=dev(cond), mask=mask.cuda(), reverse=True, ode_solver="tsit5", ode_steps=20).cpu()
    torch.testing.assert_close(out[keep], want[keep], atol=2e-2, rtol=5e-2)  # (Tsitouras 5(4) at 1e-3 per step; measured <= 6e-3)


def test_default_solver_under_the_diffusion_loss():
    from oracle import diffusion_ref as dr
    from oracle.fm_ref import EpicVectorField, rk_trajectory_end
    from tests.conftest import load_epic_seeded_golden
    g = load_epic_seeded_golden("diffusion_gauss")
    m = _module(g)
    vf = EpicVectorField(g.state, "flows.0.net", g.hp, freqs=g.freqs)
    z, mask, cond = (g.get("midpoint_10/" + k) for k in ("z", "mask", "cond"))
    with torch.no_grad():
        want = rk_trajectory_end(lambda tt, xx: dr.diffusion_rhs(vf, tt, xx, cond, mask, g.hp["diff_config"]), z * mask,
                                 torch.linspace(1.0, 0.0, 240), "rk4")
    out = m((z * mask).cuda(), cond=cond.cuda(), mask=mask.cuda(), reverse=True, ode_solver="dopri5_zuko").cpu()
    torch.testing.assert_close(out, want, atol=5e-3, rtol=2e-2)
