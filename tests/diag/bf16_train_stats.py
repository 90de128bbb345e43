import sys, torch
sys.path.insert(0, "/root/repo")
from oracle.fm_ref import EpicVectorField, fm_ot_loss
from particle_fm_amd.fm_loss import epic_fm_loss
from particle_fm_amd.layout import EpicLayout
from tests.conftest import load_golden
from tests.test_layout_cpu import cfg_of
for name in ("jetnet30", "cond_gl", "jetnet150"):
    g = load_golden(name)
    tag = "loss_f32/"
    x, t, z, mask, cond = (g.get(tag + k) for k in ("x", "t", "z", "mask", "cond"))
    ref = g.grads(tag)
    dev = lambda a: None if a is None else a.cuda()
    lay = EpicLayout(cfg_of(g.hp), flags=3)
    state = {k: v.clone().cuda().requires_grad_(v.is_floating_point() and "frequencies" not in k) for k, v in g.state.items()}
    src = lay.source_vector(state, "flows.0.net.", freqs=g.freqs)
    loss = epic_fm_loss(lay, src, dev(x), dev(t), dev(z), dev(cond), dev(mask), sigma=1e-4); loss.backward()
    st = {k: v.clone().requires_grad_(v.is_floating_point() and "frequencies" not in k) for k, v in g.state.items()}
    vf = EpicVectorField(st, "flows.0.net", g.hp, freqs=g.freqs)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        lac, *_ = fm_ot_loss(vf, x, mask, cond, t, z, sigma=1e-4)
    lac.float().backward()
    rows = []
    n16 = nac = nref = 0.0
    for k, gref in ref.items():
        scale = max(gref.abs().max().item(), 1e-8)
        d16 = (state[k].grad.cpu() - gref).abs().max().item() / scale
        dac = (st[k].grad.float() - gref).abs().max().item() / scale
        rows.append((d16 / max(dac, 1e-9), d16, dac, k))
        n16 += float((state[k].grad.cpu() - gref).double().pow(2).sum()); nac += float((st[k].grad.float() - gref).double().pow(2).sum()); nref += float(gref.double().pow(2).sum())
    rows.sort(reverse=True)
    print(name, "loss err", abs(float(loss) - float(g.get(tag+"loss"))), abs(float(lac) - float(g.get(tag+"loss"))))
    print("  whole-gradient rel L2: hip %.4f autocast %.4f" % ((n16 / nref) ** 0.5, (nac / nref) ** 0.5))
    print("  sum d16 %.3f sum dac %.3f" % (sum(r[1] for r in rows), sum(r[2] for r in rows)))
    for r in rows[:6]: print("   %.2f  d16 %.4f dac %.4f %s" % r)
