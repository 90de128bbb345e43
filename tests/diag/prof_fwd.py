"""profiling driver: a few forward launches (B=256, dense)"""
import sys, torch
sys.path.insert(0, ".")
from tests.conftest import load_golden
from tests.test_layout_cpu import cfg_of
from particle_fm_amd.layout import EpicLayout
from particle_fm_amd import hip_ops
g = load_golden("jetnet150")
lay = EpicLayout(cfg_of(g.hp), flags=0)
blob = lay.pack_blob(g.state, "flows.0.net.").cuda()
B = 256
gen = torch.Generator().manual_seed(0)
x = torch.randn(B, 150, 3, generator=gen).cuda(); t = torch.rand(B, generator=gen).cuda()
for _ in range(5):
    hip_ops.epic_forward(lay, blob, t, x, None, None)
torch.cuda.synchronize()
