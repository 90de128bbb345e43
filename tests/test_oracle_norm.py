"""oracle/norm_ref.py against vectors recorded from the reference's IterativeNormLayer (tests/golden/norm_layer.npz)."""
import os

import numpy as np
import torch

from oracle import norm_ref

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "norm_layer.npz"))


def t(k):
    return torch.from_numpy(G[k])


def test_running_statistics_forward_reverse():
    st = norm_ref.new_state(3)
    frozen = False
    for k in range(4):
        tag = f"step{k}/"
        x, mask = t(tag + "x"), t(tag + "mask")
        if not frozen:
            frozen = norm_ref.update(st, x, mask, int(G["max_n"]))
        y = norm_ref.forward(st, x, mask)
        torch.testing.assert_close(y, t(tag + "y"), rtol=1e-5, atol=1e-6)
        for b in ("means", "vars", "m2"):
            torch.testing.assert_close(st[b], t(tag + b), rtol=1e-5, atol=1e-6)
        assert int(st["n"]) == int(G[tag + "n"]) and frozen == bool(G[tag + "frozen"])
    assert [bool(G[f"step{k}/frozen"]) for k in range(4)] == [False, False, True, True]  # the freeze at max_n is exercised
    assert int(G["step3/n"]) == int(G["step2/n"])
    torch.testing.assert_close(norm_ref.reverse(st, t("step3/y"), t("step3/mask")), t("rev/y"), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(t("rev/y"), t("step3/x"), rtol=1e-4, atol=1e-5)
    sc = norm_ref.new_state(2)
    norm_ref.update(sc, t("cond/x"))
    torch.testing.assert_close(norm_ref.forward(sc, t("cond/x")), t("cond/y"), rtol=1e-5, atol=1e-6)
