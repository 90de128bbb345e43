"""The EPiC oracle at the JetClass width (hidden 300, 20 layers) against vectors recorded from the reference."""
import torch

from oracle.fm_ref import EpicVectorField, fm_ot_loss, sample_midpoint


def _vf(g, state=None):
    return EpicVectorField(state or g.state, "flows.0.net", g.hp, freqs=g.freqs)


def test_nfe(wide_golden):
    g = wide_golden
    vf = _vf(g)
    for mk in ("f32", "int64", "none"):
        tag = f"nfe_{mk}/"
        x, t, mask, cond = (g.get(tag + k) for k in ("x", "t", "mask", "cond"))
        tt = t.unsqueeze(-1).repeat_interleave(x.shape[1], dim=1)
        with torch.no_grad():
            v = vf(tt, x, cond=cond, mask=mask)
        ref = g.get(tag + "v_vec_t")
        torch.testing.assert_close(v, ref, rtol=1e-4, atol=1e-5 * float(ref.abs().max()))
        assert float(ref.abs().max()) < 1e4  # the seeded weights keep 20 residual layers in a sane range


def test_loss_and_grads(wide_golden):
    g = wide_golden
    state = {k: v.clone().requires_grad_(True) for k, v in g.state.items() if k != "flows.0.frequencies"}
    tag = "loss_f32/"
    x, t, z, mask, cond = (g.get(tag + k) for k in ("x", "t", "z", "mask", "cond"))
    loss, *_ = fm_ot_loss(_vf(g, state), x, mask, cond, t, z, 1e-4)
    torch.testing.assert_close(loss.detach(), g.get(tag + "loss"), rtol=1e-5, atol=1e-6)
    loss.backward()
    ref = g.grads(tag)
    for k, p in state.items():
        want = ref[k]
        got = g.pick(p.grad)
        assert float((got - want).norm()) <= 2e-4 * float(want.norm()) + 1e-7, k


def test_midpoint(wide_golden):
    g = wide_golden
    for steps in (3, 10):
        tag = f"midpoint_{steps}/"
        z, mask, cond = (g.get(tag + k) for k in ("z", "mask", "cond"))
        xe = sample_midpoint(_vf(g), z, cond, mask, steps)
        ref = g.get(tag + "x_end")
        torch.testing.assert_close(xe, ref, rtol=1e-3, atol=1e-4 * max(1.0, float(ref.abs().max())))
