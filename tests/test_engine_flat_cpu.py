"""FlatParams with an early group (CPU): the parameters named first lie in front, everything stays a view."""
import torch
import torch.nn as nn

from particle_fm_amd.engine import FlatParams, early_linear


def test_early_group_lies_in_front():
    m = nn.ModuleDict({"fc_l2": nn.Linear(5, 3), "fc_g1": nn.Linear(4, 2), "fc_local1": nn.Linear(3, 3), "fc_global2": nn.Linear(2, 7)})
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    early = {id(p) for n, p in m.named_parameters() if early_linear(n.rsplit(".", 1)[0])}
    fp = FlatParams(m.parameters(), first=early)
    by_id = {id(p): n for n, p in m.named_parameters()}
    names = [by_id[id(p)] for p in fp.params]
    assert names == ["fc_g1.weight", "fc_g1.bias", "fc_global2.weight", "fc_global2.bias",
                     "fc_l2.weight", "fc_l2.bias", "fc_local1.weight", "fc_local1.bias"]
    assert fp.n_first == fp.offsets[4] == 8 + 4 + 16 + 8 and fp.n_first % 4 == 0
    assert fp.is_intact()
    for n, p in m.named_parameters():
        assert torch.equal(p.detach(), before[n])
    fp.grad[: fp.n_first].fill_(1.0)
    assert float(m["fc_g1"].weight.grad.sum()) == 8.0 and float(m["fc_l2"].weight.grad.abs().sum()) == 0.0
    assert FlatParams(m.parameters()).n_first == 0


def test_early_linear_names():
    assert all(early_linear(n) for n in ("fc_l1", "fc_l3", "fc_g1", "fc_g2", "nn_list.3.fc_global1", "flows.0.net.nn_list.0.fc_global2"))
    assert not any(early_linear(n) for n in ("fc_l2", "nn_list.0.fc_local1", "nn_list.5.fc_local2"))
