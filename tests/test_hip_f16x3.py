"""PFM_F_F16X3_MFMA: the jet-resident inference kernels on split-fp16 operands must pass the SAME fp32 parity bars as
the fp32-MFMA kernels (tests/test_hip_forward.py): the split keeps 22 significant bits per operand."""
import pytest
import torch

from tests.test_layout_cpu import cfg_of

pytestmark = pytest.mark.gpu

ATOL, RTOL = 1e-5, 1e-4  # identical to tests/test_hip_forward.py


def _dev(t):
    return None if t is None else t.cuda()


def _setup(golden):
    from particle_fm_amd.layout import EpicLayout
    lay = EpicLayout(cfg_of(golden.hp), flags=1 | 4)
    return lay, lay.pack_blob(golden.state, "flows.0.net.", freqs=golden.freqs).cuda()


@pytest.mark.parametrize("mk", ["f32", "int64", "none"])
def test_forward_matches_reference_vectors(golden, mk):
    from particle_fm_amd import hip_ops
    lay, blob = _setup(golden)
    tag = f"nfe_{mk}/"
    x, t, mask, cond = (golden.get(tag + k) for k in ("x", "t", "mask", "cond"))
    v = hip_ops.epic_forward(lay, blob, _dev(t), _dev(x), _dev(cond), _dev(mask)).cpu()
    ref = golden.get(tag + "v_vec_t")
    torch.testing.assert_close(v, ref, atol=ATOL, rtol=RTOL)
    assert (v - ref).abs().max() < 5e-6  # in fact at the level of fp32 re-association noise
    if mask is not None:
        assert torch.all(v[mask.squeeze(-1) == 0] == 0)


@pytest.mark.parametrize("steps", [3, 10, 100])
def test_midpoint_matches_reference_vectors(golden, steps):
    from particle_fm_amd import hip_ops
    lay, blob = _setup(golden)
    tag = f"midpoint_{steps}/"
    z, mask, cond = golden.get(tag + "z"), golden.get(tag + "mask"), golden.get(tag + "cond")
    xe = hip_ops.epic_sample_midpoint(lay, blob, _dev(z), _dev(cond), _dev(mask), ode_steps=steps).cpu()
    torch.testing.assert_close(xe, golden.get(tag + "x_end"), atol=5e-5, rtol=1e-4)
