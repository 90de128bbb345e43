"""Fixed-step euler / rk4 sampling (ode_solver "euler", "rk4" of CNF.decode, the rk4 of CNF.encode) of all four vector-field
paths against the oracle's restated torchdyn driver; the midpoint tableau against the tuned midpoint sampler."""
import copy

import pytest
import torch

from oracle.ca_ref import CrossAttentionVectorField
from oracle.fm_ref import EpicVectorField, sample_fixed_step
from oracle.tf_ref import TransformerVectorField
from tests.conftest import load_ca_golden, load_golden, load_tf_golden, load_wide_golden
from tests.test_layout_cpu import cfg_of

pytestmark = pytest.mark.gpu


def _dev(t):
    return None if t is None else t.cuda()


def _family(name):
    """-> (golden, layout, blob, oracle field, rk sampler, tuned midpoint sampler)"""
    if name == "epic":
        from particle_fm_amd import hip_ops as ops
        from particle_fm_amd.layout import EpicLayout
        g = load_golden("cond_gl")
        lay = EpicLayout(cfg_of(g.hp))
        blob = lay.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
        return g, lay, blob, EpicVectorField(g.state, "flows.0.net", g.hp, freqs=g.freqs), ops.epic_sample_rk, ops.epic_sample_midpoint
    if name == "wide":
        from particle_fm_amd import hip_ops_wide as ops
        from particle_fm_amd.layout_wide import EpicWideLayout
        g = load_wide_golden("small")
        lay = EpicWideLayout(cfg_of(g.hp))
        blob = lay.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
        return g, lay, blob, EpicVectorField(g.state, "flows.0.net", g.hp, freqs=g.freqs), ops.ew_sample_rk, ops.ew_sample_midpoint
    if name == "tf":
        from particle_fm_amd import hip_ops_tf as ops
        from particle_fm_amd.layout_tf import TfConfig, TfLayout
        g = load_tf_golden("small")
        lay = TfLayout(TfConfig.from_hparams(g.hp))
        blob = lay.pack_blob(g.state, "flows.0.", freqs=g.freqs).cuda()
        return g, lay, blob, TransformerVectorField(g.state, "flows.0.", g.hp, freqs=g.freqs), ops.tf_sample_rk, ops.tf_sample_midpoint
    from particle_fm_amd import hip_ops_ca as ops
    from particle_fm_amd.layout_ca import CaConfig, CaLayout
    g = load_ca_golden("small")
    lay = CaLayout(CaConfig.from_hparams(g.hp))
    blob = lay.pack_blob(g.state, "flows.0.", freqs=g.freqs).cuda()
    return g, lay, blob, CrossAttentionVectorField(g.state, "flows.0.", g.hp, freqs=g.freqs), ops.ca_sample_rk, ops.ca_sample_midpoint


@pytest.mark.parametrize("family", ["epic", "wide", "tf", "ca"])
def test_euler_rk4_and_midpoint_tableau(family):
    g, lay, blob, vf, sample_rk, sample_mid = _family(family)
    tag = "midpoint_10/"
    z, mask, cond = g.get(tag + "z"), g.get(tag + "mask"), g.get(tag + "cond")
    for solver, steps in (("euler", 7), ("rk4", 5)):
        ref = sample_fixed_step(vf, z, cond, mask, ode_steps=steps, solver=solver)
        got = sample_rk(lay, blob, _dev(z), _dev(cond), _dev(mask), ode_steps=steps, solver=solver).cpu()
        torch.testing.assert_close(got, ref, atol=1e-4, rtol=1e-3)
    # the generic scheme with the midpoint tableau reproduces the tuned midpoint sampler bit for bit
    a = sample_rk(lay, blob, _dev(z), _dev(cond), _dev(mask), ode_steps=10, solver="midpoint").cpu()
    kw = {}
    if family == "epic":
        import ctypes
        from particle_fm_amd import _lib
        # unconditioned jets run both samplers on the lean evaluation with the time-term table (csrc/epic_fast.h); otherwise the Runge-
        # Kutta kernel has no table and the tuned midpoint sampler's table changes the sum order
        # (conditioned jets: only the midpoint sampler has the lean path, the Runge-Kutta kernel stays generic)
        if not _lib.load().pfm_epic_sample_is_fast(ctypes.byref(lay.desc)) or g.hp["global_cond_dim"] > 0:
            kw = dict(time_table=False)
    b = sample_mid(lay, blob, _dev(z), _dev(cond), _dev(mask), ode_steps=10, **kw).cpu()
    assert torch.equal(a, b)
    if kw:
        torch.testing.assert_close(sample_mid(lay, blob, _dev(z), _dev(cond), _dev(mask), ode_steps=10).cpu(), a, atol=5e-6, rtol=1e-5)
    torch.testing.assert_close(a, g.get(tag + "x_end"), atol=2e-4, rtol=1e-3)


def test_decode_and_encode_through_the_module():
    """CNF.decode(ode_solver="rk4" / "euler") and CNF.encode (rk4, t: 0 -> 1, 100 points, no conditioning) on an
    unconditioned EPiC model; encode followed by decode with the same scheme returns to the start (rk4 is accurate to
    ~1e-6 here)."""
    from particle_fm_amd.models import SetFlowMatchingLitModule
    g = load_golden("jetnet30")
    m = SetFlowMatchingLitModule(optimizer=None, **copy.deepcopy(g.hp))
    full = dict(g.state)
    full.update({"loss." + k: v for k, v in g.state.items()})
    m.load_state_dict(full)
    m = m.cuda()
    from particle_fm_amd.layout import EpicLayout
    vf = EpicVectorField(g.state, "flows.0.net", g.hp, freqs=EpicLayout(cfg_of(g.hp)).default_freqs())
    tag = "midpoint_10/"
    z, mask = g.get(tag + "z"), g.get(tag + "mask")
    zc = (z * mask).cuda() if mask is not None else z.cuda()
    for solver in ("euler", "rk4"):
        out = m(zc, cond=None, mask=_dev(mask), reverse=True, ode_solver=solver, ode_steps=8).cpu()
        torch.testing.assert_close(out, sample_fixed_step(vf, z, None, mask, ode_steps=8, solver=solver), atol=1e-4, rtol=1e-3)
    x0 = m(zc, cond=None, mask=_dev(mask), reverse=True, ode_solver="rk4", ode_steps=100)
    lat = m(x0, mask=_dev(mask), reverse=False)  # encode: rk4, linspace(0, 1, 100)
    ref = sample_fixed_step(vf, x0.cpu(), None, mask, ode_steps=100, solver="rk4", t0=0.0, t1=1.0)
    torch.testing.assert_close(lat.cpu(), ref, atol=2e-4, rtol=1e-3)
    torch.testing.assert_close(lat.cpu(), zc.cpu(), atol=5e-3, rtol=5e-3)
