"""DroidLoss (losses.py:304-342) in the oracle against the reference's recorded loss and gradients."""
import torch

from oracle.fm_ref import EpicVectorField, droid_loss
from oracle.seeded import subsample
from oracle.tf_ref import TransformerVectorField


def _check(g, vf_cls, prefix):
    state = {k: v.clone().requires_grad_(True) for k, v in g.state.items() if k != "flows.0.frequencies"}
    tag = "droid/"
    x, t, z, mask, cond = (g.get(tag + k) for k in ("x", "t", "z", "mask", "cond"))
    loss, *_ = droid_loss(vf_cls(state, prefix, g.hp, freqs=g.freqs), x, mask, cond, t, z)
    torch.testing.assert_close(loss.detach(), g.get(tag + "loss"), rtol=1e-5, atol=1e-6)
    loss.backward()
    ref = g.grads(tag)
    assert len(ref) == 6
    for k, want in ref.items():
        got = torch.from_numpy(subsample(state[k].grad.numpy()))
        assert float((got - want).norm()) <= 2e-4 * float(want.norm()) + 1e-7, k


def test_droid_epic(wide_golden):
    _check(wide_golden, EpicVectorField, "flows.0.net")


def test_droid_transformer(tf_golden):
    _check(tf_golden, TransformerVectorField, "flows.0.")
