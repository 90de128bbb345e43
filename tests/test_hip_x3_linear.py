"""PFM_TF_F_F16X3 / PFM_EW_F_F16X3: the row-matrix GEMM paths (transformer, wide EPiC) with split-fp16 Linears must pass
the same fp32 parity bars as their fp32-MFMA builds."""
import pytest
import torch

from tests.test_layout_cpu import cfg_of

pytestmark = pytest.mark.gpu


def _dev(t):
    return None if t is None else t.cuda()


def test_transformer_forward_and_sample(tf_golden):
    from particle_fm_amd import hip_ops_tf as ops
    from particle_fm_amd.layout_tf import TfConfig, TfLayout
    g = tf_golden
    lay = TfLayout(TfConfig.from_hparams(g.hp), flags=1)
    blob = lay.pack_blob(g.state, "flows.0.", freqs=g.freqs).cuda()
    for mk in ("f32", "ones"):
        tag = f"nfe_{mk}/"
        x, t, mask, cond = (g.get(tag + k) for k in ("x", "t", "mask", "cond"))
        v = ops.tf_forward(lay, blob, _dev(t), _dev(x), _dev(cond), _dev(mask)).cpu()
        torch.testing.assert_close(v, g.get(tag + "v_vec_t"), atol=2e-5, rtol=2e-4)
    tag = "midpoint_10/"
    z, mask, cond = (g.get(tag + k) for k in ("z", "mask", "cond"))
    xe = ops.tf_sample_midpoint(lay, blob, _dev(z), _dev(cond), _dev(mask), ode_steps=10).cpu()
    torch.testing.assert_close(xe, g.get(tag + "x_end"), atol=2e-4, rtol=1e-3)


def test_transformer_training_gradients(tf_golden):
    from particle_fm_amd.fm_loss_tf import tf_fm_loss
    from particle_fm_amd.layout_tf import TfConfig, TfLayout
    from tests.test_hip_tf_train import _check_grads
    g = tf_golden
    lay = TfLayout(TfConfig.from_hparams(g.hp), flags=1)
    flat = torch.cat([g.state[k].reshape(-1) for k in lay.keys("flows.0.")]).cuda().requires_grad_(True)
    tag = "loss_f32/"
    x, t, mask, cond, z = (g.get(tag + k).cuda() for k in ("x", "t", "mask", "cond", "z"))
    loss = tf_fm_loss(lay, flat, x, t, z, cond, mask, 1e-4, "FM-OT", None, freqs=g.freqs)
    torch.testing.assert_close(loss.detach().cpu(), g.get(tag + "loss"), rtol=2e-5, atol=1e-6)
    loss.backward()
    _check_grads(g, lay, flat.grad.cpu(), tag)


def test_wide_epic_forward_and_sample(wide_golden):
    from particle_fm_amd import hip_ops_wide as ops
    from particle_fm_amd.layout_wide import EpicWideLayout
    g = wide_golden
    lay = EpicWideLayout(cfg_of(g.hp), flags=1)
    blob = lay.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda()
    for mk in ("f32", "none"):
        tag = f"nfe_{mk}/"
        x, t, mask, cond = (g.get(tag + k) for k in ("x", "t", "mask", "cond"))
        v = ops.ew_forward(lay, blob, _dev(t), _dev(x), _dev(cond), _dev(mask)).cpu()
        torch.testing.assert_close(v, g.get(tag + "v_vec_t"), atol=2e-5, rtol=2e-4)
    tag = "midpoint_10/"
    z, mask, cond = (g.get(tag + k) for k in ("z", "mask", "cond"))
    xe = ops.ew_sample_midpoint(lay, blob, _dev(z), _dev(cond), _dev(mask), ode_steps=10).cpu()
    torch.testing.assert_close(xe, g.get(tag + "x_end"), atol=2e-4, rtol=1e-3)
