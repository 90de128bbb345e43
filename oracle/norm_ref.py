"""CPU restatement of IterativeNormLayer (particle_fm/models/components/norm_layer.py:84-155) as pure functions on a state
dict {means, vars, n, m2} (shapes (1,F), (1,F), (), (1,F)).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  PINNED: tests/golden/norm_layer.npz, recorded from the reference's class by
oracle/make_golden.py."""
from __future__ import annotations

import torch


def new_state(features: int):
    return {"means": torch.zeros(1, features), "vars": torch.ones(1, features), "n": torch.tensor(0), "m2": torch.ones(1, features)}


def _sel(x, mask):
    return x if mask is None else x[mask]


def fit(state, x, mask=None):
    """:98-104: unbiased variance and mean over the valid rows; m2 = vars * n."""
    s = _sel(x, mask)
    state["vars"], state["means"] = torch.var_mean(s, dim=(0,), keepdim=True)
    state["n"] = torch.tensor(len(s))
    state["m2"] = state["vars"] * state["n"]


def update(state, x, mask=None, max_n: int = 500_000):
    """:137-155 (batched Welford); returns the new ``frozen`` flag."""
    s = _sel(x, mask)
    if state["n"] == 0:
        fit(state, s)
        return False  # fit(freeze=False)
    state["n"] = state["n"] + len(s)
    delta = s - state["means"]
    state["means"] = state["means"] + (delta / state["n"]).mean(dim=(0,), keepdim=True) * len(s)
    delta2 = s - state["means"]
    state["m2"] = state["m2"] + (delta * delta2).mean(dim=(0,), keepdim=True) * len(s)
    state["vars"] = state["m2"] / state["n"]
    return bool(state["n"] >= max_n)


def forward(state, x, mask=None):
    """:116-126 (the mapping only; the caller decides about update())."""
    s = _sel(x, mask)
    normed = (s - state["means"]) / (state["vars"].sqrt() + 1e-8)
    if mask is None:
        return normed
    out = x.clone()
    out[mask] = normed
    return out


def reverse(state, x, mask=None):
    """:128-139."""
    s = _sel(x, mask)
    un = s * state["vars"].sqrt() + state["means"]
    if mask is None:
        return un
    out = x.clone()
    out[mask] = un
    return out
