"""ad-hoc GPU debug: compare every saved activation of the HIP forward with the CPU blob interpreter."""
import sys, torch
sys.path.insert(0, ".")  # run from the repository root
from tests.conftest import load_golden
from tests.test_layout_cpu import cfg_of
from tests.blob_interp import interp_forward
from oracle.fm_ref import fm_ot_targets
from particle_fm_amd.layout import EpicLayout, saved_layout
from particle_fm_amd import hip_ops

name = sys.argv[1] if len(sys.argv) > 1 else "jetnet30"
g = load_golden(name)
lay = EpicLayout(cfg_of(g.hp))
blob = lay.pack_blob(g.state, "flows.0.net.", freqs=g.freqs)
tag = "loss_f32/"
x, t, z, mask, cond = g.get(tag+"x"), g.get(tag+"t"), g.get(tag+"z"), g.get(tag+"mask"), g.get(tag+"cond")
tt, y, u, m = fm_ot_targets(x, mask, t, z, 1e-4)
tr = {}
v = interp_forward(lay, blob, t, y, cond, mask, trace=tr)
parts, cnt, saved = hip_ops.epic_fm_loss_forward(lay, blob.cuda(), x.cuda(), t.cuda(), z.cuda(), None if cond is None else cond.cuda(), mask.cuda())
saved = saved.cpu(); B, N, F = x.shape; H = 128
sl = saved_layout(N, F, lay.cfg.layers)
def sv(off, shape):
    n = 1
    for s_ in shape: n *= s_
    return saved[:, off:off+n].reshape(B, *shape)
def cmp(nm, a, b, mrows=True):
    if mrows and a.dim() == 3:
        d = ((a - b).abs() ).max().item()
    else:
        d = (a - b).abs().max().item()
    print(f"{nm:12s} maxdiff {d:.3e}  ref absmax {b.abs().max().item():.3e}")
cmp("y", sv(sl["y"], (N, F)), y); cmp("u", sv(sl["u"], (N, F)), u)
cmp("temb", saved[:, sl["temb"]:sl["temb"]+lay.cfg.t_dim], tr["temb"])
cmp("x1", sv(sl["x1"], (N, H)), tr["x1"]); cmp("x2", sv(sl["x2"], (N, H)), tr["x2"])
cmp("pool0", saved[:, sl["pool"]:sl["pool"]+H], tr["pool0"])
cmp("gstem1", saved[:, sl["gstem1"]:sl["gstem1"]+H], tr["gstem1"])
cmp("gstem", saved[:, sl["gstem"]:sl["gstem"]+lay.cfg.latent], tr["gstem"])
for k in range(lay.cfg.layers):
    go = sl["glayer"] + k*sl["gstride"]
    cmp(f"g1_{k}", saved[:, go:go+H], tr[f"g1_{k}"]); cmp(f"g_{k}", saved[:, go+H:go+H+lay.cfg.latent], tr[f"g_{k}"])
    cmp(f"l1_{k}", sv(sl["l1"]+k*sl["lstride"], (N, H)), tr[f"l1_{k}"]); cmp(f"xo_{k}", sv(sl["xo"]+k*sl["lstride"], (N, H)), tr[f"xo_{k}"])
cmp("v", sv(sl["v"], (N, F)), v)
print("loss hip", (parts.sum()/cnt.sum()).item(), "golden", g.get(tag+"loss").item())
vh = sv(sl["v"], (N, F)); uh = sv(sl["u"], (N, F))
print("parts hip     ", parts.cpu().tolist())
print("parts from v,u", (vh-uh).square().sum((1,2)).tolist())
print("parts cpu     ", (v-u).square().sum((1,2)).tolist())
print("cnt", cnt.cpu().tolist(), m.sum((1,2)).tolist())
