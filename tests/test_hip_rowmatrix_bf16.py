"""PFM_TF_F_BF16 / PFM_EW_F_BF16 / PFM_CA_F_BF16 / PFM_MDMA_F_BF16: the Linears of the row-matrix paths (Full-Transformer = BASELINE cfg 4,
wide EPiC = cfg 5, cross-attention, MDMA) on bf16 operands with fp32 accumulate -- what Lightning's trainer.precision="bf16-mixed"
(configs/trainer/default.yaml:11-12: autocast around the same modules) asks of nn.Linear.

As for the jet-resident kernels (tests/test_hip_bf16.py) there is no bit-level reference for reduced precision; the bar is the reference's
own bf16 path: the oracle under torch.autocast(bfloat16) against the reference's fp32 vectors.  The kernels keep activations, LayerNorm,
softmax, attention products and the dW GEMMs in fp32 and only round the Linear operands, so they must not be further from the fp32 result
than autocast is (x 1.5 + a small absolute term)."""
import pytest
import torch

from oracle.fm_ref import EpicVectorField, fm_ot_loss
from tests.test_layout_cpu import cfg_of

pytestmark = pytest.mark.gpu

BF16 = 32  # PFM_*_F_BF16


def _dev(t):
    return None if t is None else t.cuda()


def _paths():
    """name -> (golden, fp32 layout, bf16 layout, packed-blob maker, forward op, oracle field)"""
    from oracle.ca_ref import CrossAttentionVectorField
    from oracle.tf_ref import TransformerVectorField
    from particle_fm_amd import hip_ops_ca, hip_ops_tf, hip_ops_wide
    from particle_fm_amd.layout_ca import CaConfig, CaLayout
    from particle_fm_amd.layout_tf import TfConfig, TfLayout
    from particle_fm_amd.layout_wide import EpicWideLayout
    from tests.conftest import load_ca_golden, load_tf_golden, load_wide_golden

    def tf():
        g = load_tf_golden("small")
        mk = lambda fl: TfLayout(TfConfig.from_hparams(g.hp), flags=fl)
        return g, mk, lambda lay: lay.pack_blob(g.state, "flows.0.", freqs=g.freqs).cuda(), hip_ops_tf.tf_forward, \
            TransformerVectorField(g.state, "flows.0.", g.hp, freqs=g.freqs)

    def ca():
        g = load_ca_golden("small")
        mk = lambda fl: CaLayout(CaConfig.from_hparams(g.hp), flags=fl)
        return g, mk, lambda lay: lay.pack_blob(g.state, "flows.0.", freqs=g.freqs).cuda(), hip_ops_ca.ca_forward, \
            CrossAttentionVectorField(g.state, "flows.0.", g.hp, freqs=g.freqs)

    def ew():
        g = load_wide_golden("small")
        mk = lambda fl: EpicWideLayout(cfg_of(g.hp), flags=fl)
        return g, mk, lambda lay: lay.pack_blob(g.state, "flows.0.net.", freqs=g.freqs).cuda(), hip_ops_wide.ew_forward, \
            EpicVectorField(g.state, "flows.0.net", g.hp, freqs=g.freqs)

    def mdma():
        from oracle.mdma_ref import MdmaVectorField, broadcast_field
        from particle_fm_amd import hip_ops_mdma
        from particle_fm_amd.layout_mdma import MdmaConfig, MdmaLayout
        from tests.conftest import load_mdma_golden
        g = load_mdma_golden("small")
        mk = lambda fl: MdmaLayout(MdmaConfig.from_hparams(g.hp), flags=fl)
        fwd = lambda lay, blob, t, x, cond, mask: hip_ops_mdma.mdma_forward(lay, blob, t, x, mask)  # (B, N, F): the (B, N, 1) field broadcast
        return g, mk, lambda lay: lay.pack_blob(g.state, "flows.0.", freqs=g.freqs).cuda(), fwd, \
            broadcast_field(MdmaVectorField(g.state, "flows.0.", g.hp, freqs=g.freqs))

    return {"tf": tf, "ca": ca, "ew": ew, "mdma": mdma}


@pytest.mark.parametrize("path", ["tf", "ca", "ew", "mdma"])
def test_bf16_forward_within_the_reference_bf16_error(path):
    g, mk, pack, fwd, vf = _paths()[path]()
    lay32, lay16 = mk(0), mk(BF16)
    tag = "nfe_f32/"
    x, t, mask, cond = (g.get(tag + k) for k in ("x", "t", "mask", "cond"))
    ref = g.get(tag + "v_vec_t").expand_as(x)  # (MDMA records its (B, N, 1) field)
    v32 = fwd(lay32, pack(lay32), _dev(t), _dev(x), _dev(cond), _dev(mask)).cpu()
    v16 = fwd(lay16, pack(lay16), _dev(t), _dev(x), _dev(cond), _dev(mask)).cpu()
    N = x.shape[1]
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        vac = vf(t[:, None].expand(-1, N), x, cond=cond, mask=mask).float()
    keep = mask.squeeze(-1) != 0  # (the transformer paths leave meaningless values at padded rows, like the reference)
    e16, eac = (v16 - ref).abs()[keep], (vac - ref).abs()[keep]
    assert (v32 - ref).abs()[keep].max() < 5e-5                # the flag really selects another kernel ...
    assert e16.max() > 1e-5                                     # ... whose operands are rounded
    assert e16.max() <= 1.5 * eac.max() + 1e-3, (e16.max(), eac.max())
    assert e16.mean() <= 1.5 * eac.mean() + 1e-4, (e16.mean(), eac.mean())


def _grad_errors(got: dict, ref: dict):
    tot = n = nref = 0.0
    for k, gref in ref.items():
        scale = max(gref.abs().max().item(), 1e-8)
        tot += (got[k] - gref).abs().max().item() / scale
        n += float((got[k] - gref).double().pow(2).sum())
        nref += float(gref.double().pow(2).sum())
    return tot, (n / nref) ** 0.5


@pytest.mark.parametrize("path", ["tf", "ew"])
def test_bf16_training_within_the_reference_bf16_error(path):
    """Loss forward + backward under the flag (Linears and dX products on bf16 operands, dW GEMMs fp32): the loss and the whole
    gradient (relative L2 error, sum of the per-tensor max errors) are no further from the reference's fp32 vectors than the oracle
    under torch.autocast(bfloat16) is."""
    from particle_fm_amd.fm_loss_tf import tf_fm_loss
    from particle_fm_amd.fm_loss_wide import epic_wide_fm_loss
    g, mk, pack, fwd, _ = _paths()[path]()
    tag = "loss_f32/"
    x, t, z, mask, cond = (g.get(tag + k) for k in ("x", "t", "z", "mask", "cond"))
    ref_loss, ref = g.get(tag + "loss"), g.grads(tag)
    pre = "flows.0." if path == "tf" else "flows.0.net."

    def hip(flags):
        lay = mk(flags)
        if path == "tf":
            keys = lay.keys("flows.0.")
            flat = torch.cat([g.state[k].reshape(-1) for k in keys]).cuda().requires_grad_(True)
            loss = tf_fm_loss(lay, flat, _dev(x), _dev(t), _dev(z), _dev(cond), _dev(mask), 1e-4, "FM-OT", None, freqs=g.freqs)
            loss.backward()
            out, o = {}, 0
            for (k, shp), full in zip(lay.shapes, keys):
                n = int(torch.tensor(shp).prod())
                out[full] = flat.grad[o:o + n].reshape(shp).cpu()
                o += n
            return loss.detach().cpu(), out
        state = {k[len(pre):]: v.cuda().requires_grad_(True) for k, v in g.state.items() if k.startswith(pre)}
        src = lay.source_vector(state, "", freqs=g.freqs)
        loss = epic_wide_fm_loss(lay, src, _dev(x), _dev(t), _dev(z), _dev(cond), _dev(mask), 1e-4, "FM-OT")
        loss.backward()
        return loss.detach().cpu(), {pre + k: v.grad.cpu() for k, v in state.items()}

    l32, g32 = hip(0)
    l16, g16 = hip(BF16)
    st = {k: v.clone().requires_grad_(v.is_floating_point() and "frequencies" not in k) for k, v in g.state.items()}
    if path == "tf":
        from oracle.tf_ref import TransformerVectorField
        vf = TransformerVectorField(st, "flows.0.", g.hp, freqs=g.freqs)
    else:
        vf = EpicVectorField(st, "flows.0.net", g.hp, freqs=g.freqs)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        lac, *_ = fm_ot_loss(vf, x, mask, cond, t, z, sigma=1e-4)
    lac.float().backward()
    gac = {k: st[k].grad.float() for k in ref}
    ref = {k: v for k, v in ref.items()}
    pick = lambda d: {k: g.pick(d[k]) for k in ref}  # (fixtures with sub-sampled gradients)
    assert abs(l32 - ref_loss) < 2e-5 * max(1.0, abs(ref_loss))
    e16, eac = abs(float(l16 - ref_loss)), abs(float(lac.float().detach() - ref_loss))
    assert e16 > 1e-7, "the flag must select the bf16-operand kernels"
    assert e16 <= 1.5 * eac + 2e-3 * abs(float(ref_loss)), (e16, eac)
    t32, n32 = _grad_errors(pick(g32), ref)
    t16, n16 = _grad_errors(pick(g16), ref)
    tac, nac = _grad_errors(pick(gac), ref)
    assert n32 < 2e-3, n32
    assert n16 > 2 * n32, "the dX products must run on rounded operands"
    assert t16 <= 1.1 * tac, (t16, tac)
    assert n16 <= 1.05 * nac, (n16, nac)


def test_bf16_module_switch_transformer():
    """set_precision("bf16-mixed") on the mirror module: sample() and training_step() follow it (another descriptor, same parameters)."""
    from tests.conftest import load_tf_golden
    from tests.test_hip_tf_modules import _module
    g = load_tf_golden("small")
    m = _module(g)
    net = m.flows[0].net
    mask = g.get("nfe_f32/mask").cuda()
    cond = g.get("nfe_f32/cond").cuda()
    B = mask.shape[0]
    torch.manual_seed(9999)
    x32 = m.sample(B, cond=cond.cpu(), mask=mask.cpu(), ode_solver="midpoint", ode_steps=10).cpu()
    net.set_precision("bf16-mixed")
    assert net.layout().desc.flags & BF16
    torch.manual_seed(9999)
    x16 = m.sample(B, cond=cond.cpu(), mask=mask.cpu(), ode_solver="midpoint", ode_steps=10).cpu()
    keep = (mask.squeeze(-1) != 0).cpu()
    d = (x16 - x32).abs()[keep]
    assert torch.isfinite(x16).all() and 1e-6 < d.max() < 0.25, d.max()
    loss = m.training_step((g.get("loss_f32/x").cuda(), g.get("loss_f32/mask").cuda(), g.get("loss_f32/cond").cuda()), 0)["loss"]
    assert torch.isfinite(loss)
    net.set_precision("32")
    assert not (net.layout().desc.flags & BF16)
