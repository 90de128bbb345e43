"""HIP loss forward + backward of the Full-Transformer field against the reference's recorded loss and gradients."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(g):
    from particle_fm_amd.layout_tf import TfConfig, TfLayout
    lay = TfLayout(TfConfig.from_hparams(g.hp))
    flat = torch.cat([g.state[k].reshape(-1) for k in lay.keys("flows.0.")]).cuda().requires_grad_(True)
    return lay, flat


def _check_grads(g, lay, flat_grad, tag):
    ref = g.grads(tag)
    o = 0
    bad = []
    for k, shp in lay.shapes:
        n = int(torch.tensor(shp).prod())
        got = g.pick(flat_grad[o:o + n].reshape(shp))
        o += n
        want = ref["flows.0." + k]
        # fp32 re-association can flip LeakyReLU'(pre-activation ~ 0) of single elements (a 0.9 g jump in one term
        # of a sum), so the bound is on the tensor's relative L2 error plus a looser element-wise one
        scale = max(float(want.abs().max()), 1e-6)
        err = float((got - want).abs().max()) / scale
        l2 = float((got - want).norm() / want.norm().clamp_min(1e-12))
        if not (err < 1e-2 and l2 < 2e-3):
            bad.append((k, err, l2, scale))
    assert not bad, "gradient mismatch (key, max err / max |ref|, rel L2, max |ref|): " + str(bad[:8])


@pytest.mark.parametrize("kind", ["FM-OT", "CFM"])
def test_loss_and_all_parameter_gradients(tf_golden, kind):
    from particle_fm_amd.fm_loss_tf import tf_fm_loss
    g = tf_golden
    lay, flat = _setup(g)
    tag = "loss_f32/" if kind == "FM-OT" else "cfm/"
    x, t, mask, cond = (g.get(tag + k).cuda() for k in ("x", "t", "mask", "cond"))
    if kind == "FM-OT":
        a, eps = g.get(tag + "z").cuda(), None
    else:
        a, eps = g.get(tag + "x0").cuda(), g.get(tag + "eps").cuda()
    loss = tf_fm_loss(lay, flat, x, t, a, cond, mask, 1e-4, kind, eps, freqs=g.freqs)
    torch.testing.assert_close(loss.detach().cpu(), g.get(tag + "loss"), rtol=2e-5, atol=1e-6)
    loss.backward()
    _check_grads(g, lay, flat.grad.cpu(), tag)


def test_second_forward_before_the_first_backward_keeps_the_first_gradients(tf_golden):
    """The saved activations belong to the call: a second loss forward of the same batch size (another micro-batch, a no_grad
    validation loss) between a forward and its backward must not change that backward's gradients."""
    from particle_fm_amd.fm_loss_tf import tf_fm_loss
    g = tf_golden
    lay, flat = _setup(g)
    tag = "loss_f32/"
    x, t, mask, cond, z = (g.get(tag + k).cuda() for k in ("x", "t", "mask", "cond", "z"))
    loss = tf_fm_loss(lay, flat, x, t, z, cond, mask, 1e-4, "FM-OT", None, freqs=g.freqs)
    gen = torch.Generator(device="cuda").manual_seed(5)
    with torch.no_grad():  # same shapes, different numbers
        tf_fm_loss(lay, flat, torch.randn(x.shape, device="cuda", generator=gen), 1.0 - t, torch.randn(z.shape, device="cuda", generator=gen),
                   cond, mask, 1e-4, "FM-OT", None, freqs=g.freqs)
    other = tf_fm_loss(lay, flat, x.flip(0), t.flip(0), z, cond.flip(0), mask.flip(0), 1e-4, "FM-OT", None, freqs=g.freqs)
    loss.backward()
    _check_grads(g, lay, flat.grad.cpu(), tag)
    del other
